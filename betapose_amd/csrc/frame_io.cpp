// Host-side frame input of the hot path: the reference reads every frame with cv2.imread (BGR u8,
// dataloader.py:155) on a Python thread.  Here: a PNG decoder (LineMod frames are 8-bit RGB PNGs; inflate comes from
// zlib, everything else -- chunk walk, un-filtering, colour conversion to cv2's IMREAD_COLOR convention -- is below)
// and a threaded read-ahead loader that decodes straight into pinned host slots the pipeline uploads from.
//
// cv2.imread(IMREAD_COLOR) semantics kept: 3 channels B,G,R; alpha dropped (no compositing); grey replicated;
// palette expanded; 16-bit samples reduced to their high byte; bit depths 1/2/4 scaled to 0..255.
#include <zlib.h>

#include <atomic>
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "frame_io.h"

namespace bp {

namespace {

inline uint32_t be32(const uint8_t* p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }

struct PngHeader {
    uint32_t w = 0, h = 0;
    int depth = 0, ctype = 0, interlace = 0;
    int channels() const { return ctype == 0 ? 1 : ctype == 2 ? 3 : ctype == 3 ? 1 : ctype == 4 ? 2 : ctype == 6 ? 4 : 0; }
    size_t row_bytes() const { return ((size_t)w * channels() * depth + 7) / 8; }
    int bpp() const { const int b = channels() * depth / 8; return b < 1 ? 1 : b; }
};

const uint8_t kSig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};

void parse_header(const uint8_t* d, size_t n, PngHeader* hd) {
    if (n < 8 + 25 || std::memcmp(d, kSig, 8) != 0) throw IoError("not a PNG file");
    if (be32(d + 8) != 13 || std::memcmp(d + 12, "IHDR", 4) != 0) throw IoError("PNG: IHDR is not the first chunk");
    const uint8_t* p = d + 16;
    hd->w = be32(p); hd->h = be32(p + 4);
    hd->depth = p[8]; hd->ctype = p[9]; hd->interlace = p[12];
    if (p[10] != 0 || p[11] != 0) throw IoError("PNG: unknown compression / filter method");
    if (hd->w == 0 || hd->h == 0 || hd->w > 32768 || hd->h > 32768 || (uint64_t)hd->w * hd->h > (1ull << 26))
        throw IoError("PNG: bad dimensions (limit: 32768 per side, 64 Mpixel)");
    const int dep = hd->depth;
    bool ok = false;
    switch (hd->ctype) {
        case 0: ok = dep == 1 || dep == 2 || dep == 4 || dep == 8 || dep == 16; break;
        case 3: ok = dep == 1 || dep == 2 || dep == 4 || dep == 8; break;
        case 2: case 4: case 6: ok = dep == 8 || dep == 16; break;
        default: break;
    }
    if (!ok) throw IoError("PNG: invalid colour type / bit depth");
    if (hd->interlace != 0) throw IoError("PNG: Adam7-interlaced files are not supported");
}

inline uint8_t paeth(int a, int b, int c) {
    const int p = a + b - c;
    const int pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c);
    return (uint8_t)((pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c));
}

// in place: raw = h rows of (filter byte + row_bytes); afterwards every row holds reconstructed samples
void unfilter(uint8_t* raw, const PngHeader& hd) {
    const size_t rb = hd.row_bytes(), stride = rb + 1;
    const int bpp = hd.bpp();
    const uint8_t* prev = nullptr;
    for (uint32_t y = 0; y < hd.h; ++y) {
        uint8_t* row = raw + y * stride + 1;
        const int ft = row[-1];
        switch (ft) {
            case 0: break;
            case 1:
                for (size_t i = bpp; i < rb; ++i) row[i] = (uint8_t)(row[i] + row[i - bpp]);
                break;
            case 2:
                if (prev) for (size_t i = 0; i < rb; ++i) row[i] = (uint8_t)(row[i] + prev[i]);
                break;
            case 3:
                for (size_t i = 0; i < rb; ++i) {
                    const int a = i >= (size_t)bpp ? row[i - bpp] : 0, b = prev ? prev[i] : 0;
                    row[i] = (uint8_t)(row[i] + ((a + b) >> 1));
                }
                break;
            case 4:
                for (size_t i = 0; i < rb; ++i) {
                    const int a = i >= (size_t)bpp ? row[i - bpp] : 0, b = prev ? prev[i] : 0;
                    const int c = (prev && i >= (size_t)bpp) ? prev[i - bpp] : 0;
                    row[i] = (uint8_t)(row[i] + paeth(a, b, c));
                }
                break;
            default: throw IoError("PNG: bad row filter type");
        }
        prev = row;
    }
}

void to_bgr(const uint8_t* raw, const PngHeader& hd, const uint8_t* plte, int nplte, uint8_t* out) {
    const size_t stride = hd.row_bytes() + 1;
    const int step = hd.depth == 16 ? 2 : 1;   // 16-bit: the high (first) byte
    for (uint32_t y = 0; y < hd.h; ++y) {
        const uint8_t* r = raw + y * stride + 1;
        uint8_t* o = out + (size_t)y * hd.w * 3;
        switch (hd.ctype) {
            case 2: case 6: {
                const int px = hd.channels() * step;
                for (uint32_t x = 0; x < hd.w; ++x, r += px, o += 3) { o[0] = r[2 * step]; o[1] = r[step]; o[2] = r[0]; }
            } break;
            case 4: {
                for (uint32_t x = 0; x < hd.w; ++x, r += 2 * step, o += 3) o[0] = o[1] = o[2] = r[0];
            } break;
            case 0: case 3: {
                for (uint32_t x = 0; x < hd.w; ++x, o += 3) {
                    int v;
                    if (hd.depth >= 8) v = r[x * step];
                    else {
                        const int per = 8 / hd.depth, sh = (per - 1 - (int)(x % per)) * hd.depth;
                        v = (r[x / per] >> sh) & ((1 << hd.depth) - 1);
                    }
                    if (hd.ctype == 3) {
                        if (v >= nplte) throw IoError("PNG: palette index out of range");
                        o[0] = plte[3 * v + 2]; o[1] = plte[3 * v + 1]; o[2] = plte[3 * v];
                    } else {
                        if (hd.depth < 8) v = v * 255 / ((1 << hd.depth) - 1);
                        o[0] = o[1] = o[2] = (uint8_t)v;
                    }
                }
            } break;
        }
    }
}

}  // namespace

void png_info(const uint8_t* data, size_t n, int* h, int* w, int* channels) {
    PngHeader hd;
    parse_header(data, n, &hd);
    if (h) *h = (int)hd.h;
    if (w) *w = (int)hd.w;
    if (channels) *channels = hd.ctype == 3 ? 3 : hd.channels();
}

void png_decode_bgr(const uint8_t* data, size_t n, uint8_t* out, size_t cap, int* h, int* w, std::vector<uint8_t>& scratch) {
    PngHeader hd;
    parse_header(data, n, &hd);
    if ((size_t)hd.w * hd.h * 3 > cap) throw IoError("PNG: output buffer too small");
    const size_t raw_size = (hd.row_bytes() + 1) * hd.h;
    scratch.resize(raw_size);
    z_stream zs;
    std::memset(&zs, 0, sizeof(zs));
    if (inflateInit(&zs) != Z_OK) throw IoError("zlib: inflateInit failed");
    zs.next_out = scratch.data();
    zs.avail_out = (uInt)raw_size;
    const uint8_t* plte = nullptr;
    int nplte = 0;
    bool done = false, seen_idat = false, seen_end = false;
    size_t pos = 8;
    try {
        while (pos + 12 <= n) {
            const uint32_t len = be32(data + pos);
            const uint8_t* type = data + pos + 4;
            if ((size_t)len > n - pos - 12) throw IoError("PNG: truncated chunk");
            const uint8_t* body = data + pos + 8;
            if (std::memcmp(type, "PLTE", 4) == 0) {
                if (len % 3 || len > 768) throw IoError("PNG: bad PLTE");
                plte = body; nplte = (int)(len / 3);
            } else if (std::memcmp(type, "IDAT", 4) == 0) {
                seen_idat = true;
                if (!done && len) {
                    zs.next_in = const_cast<Bytef*>(body);
                    zs.avail_in = len;
                    const int rc = inflate(&zs, Z_NO_FLUSH);
                    if (rc == Z_STREAM_END) done = true;
                    else if (rc != Z_OK && !(rc == Z_BUF_ERROR && zs.avail_out == 0)) throw IoError("PNG: corrupt image data");
                }
            } else if (std::memcmp(type, "IEND", 4) == 0) {
                seen_end = true;
                break;
            }
            pos += 12 + (size_t)len;
        }
    } catch (...) {
        inflateEnd(&zs);
        throw;
    }
    const size_t produced = raw_size - zs.avail_out;
    inflateEnd(&zs);
    if (!seen_idat || !seen_end) throw IoError("PNG: truncated file");
    if (produced != raw_size) throw IoError("PNG: image data shorter than the header says");
    if (hd.ctype == 3 && !plte) throw IoError("PNG: palette image without PLTE");
    unfilter(scratch.data(), hd);
    to_bgr(scratch.data(), hd, plte, nplte, out);
    if (h) *h = (int)hd.h;
    if (w) *w = (int)hd.w;
}

std::vector<uint8_t> read_file(const std::string& path) {
    FILE* f = std::fopen(path.c_str(), "rb");
    if (!f) throw IoError("cannot open " + path);
    std::vector<uint8_t> buf;
    if (std::fseek(f, 0, SEEK_END) == 0) {
        const long sz = std::ftell(f);
        std::rewind(f);
        if (sz > 0) {
            buf.resize((size_t)sz);
            if (std::fread(buf.data(), 1, buf.size(), f) != buf.size()) { std::fclose(f); throw IoError("short read on " + path); }
        }
    }
    std::fclose(f);
    if (buf.empty()) throw IoError("empty file " + path);
    return buf;
}

// ---------------------------------------------------------------------------------------------------------------
// FrameLoader: `threads` workers decode files ahead of the consumer into a ring of `depth` host slots (pinned when an
// allocator is supplied); frames come out in list order.  Slot i % depth serves frames i, i+depth, ...; a worker
// waits until the consumer has released the previous occupant.
// ---------------------------------------------------------------------------------------------------------------
struct FrameLoader::Slot {
    uint8_t* buf = nullptr;
    long long expect = 0;     // frame index this slot accepts next
    int state = 0;            // 0 free, 1 decoding, 2 ready
    std::string err;
};

FrameLoader::FrameLoader(std::vector<std::string> paths, int H, int W, int threads, int depth, HostAlloc alloc, HostFree free_fn)
    : paths_(std::move(paths)), H_(H), W_(W), free_(free_fn) {
    if (H <= 0 || W <= 0) throw IoError("frame size");
    threads = threads < 1 ? 1 : threads;
    depth = depth < 2 ? 2 : depth;
    slots_.resize(depth);
    const size_t bytes = (size_t)H * W * 3;
    for (int i = 0; i < depth; ++i) {
        slots_[i].buf = alloc ? (uint8_t*)alloc(bytes) : (uint8_t*)std::malloc(bytes);
        if (!slots_[i].buf) throw IoError("host slot allocation failed");
        slots_[i].expect = i;
    }
    if (!alloc) free_ = nullptr;
    for (int t = 0; t < threads; ++t) workers_.emplace_back([this] { work(); });
}

FrameLoader::~FrameLoader() {
    {
        std::lock_guard<std::mutex> lk(m_);
        stop_ = true;
    }
    cv_free_.notify_all();
    cv_ready_.notify_all();
    for (auto& t : workers_) t.join();
    for (auto& s : slots_) {
        if (!s.buf) continue;
        if (free_) free_(s.buf); else std::free(s.buf);
    }
}

void FrameLoader::work() {
    std::vector<uint8_t> scratch;
    const long long n = (long long)paths_.size();
    for (;;) {
        const long long i = next_job_.fetch_add(1);
        if (i >= n) return;
        Slot& s = slots_[(size_t)(i % (long long)slots_.size())];
        {
            std::unique_lock<std::mutex> lk(m_);
            cv_free_.wait(lk, [&] { return stop_ || (s.state == 0 && s.expect == i); });
            if (stop_) return;
            s.state = 1;
        }
        std::string err;
        try {
            const std::vector<uint8_t> file = read_file(paths_[(size_t)i]);
            int h = 0, w = 0;
            png_info(file.data(), file.size(), &h, &w, nullptr);
            if (h != H_ || w != W_)
                throw IoError(paths_[(size_t)i] + ": frame is " + std::to_string(w) + "x" + std::to_string(h) + ", expected " +
                              std::to_string(W_) + "x" + std::to_string(H_));
            png_decode_bgr(file.data(), file.size(), s.buf, (size_t)H_ * W_ * 3, &h, &w, scratch);
        } catch (const std::exception& e) {
            err = e.what();
            if (err.find(paths_[(size_t)i]) == std::string::npos) err = paths_[(size_t)i] + ": " + err;
        }
        {
            std::lock_guard<std::mutex> lk(m_);
            s.err = err;
            s.state = 2;
        }
        cv_ready_.notify_all();
    }
}

int FrameLoader::next(long long* index, const uint8_t** bgr, std::string* err) {
    const long long i = next_out_;
    if (i >= (long long)paths_.size()) return 1;
    Slot& s = slots_[(size_t)(i % (long long)slots_.size())];
    std::unique_lock<std::mutex> lk(m_);
    cv_ready_.wait(lk, [&] { return stop_ || (s.state == 2 && s.expect == i); });
    if (stop_) return 1;
    ++next_out_;
    if (index) *index = i;
    if (!s.err.empty()) {
        if (err) *err = s.err;
        if (bgr) *bgr = nullptr;
        return -1;
    }
    if (bgr) *bgr = s.buf;
    return 0;
}

void FrameLoader::release(long long index) {
    Slot& s = slots_[(size_t)(index % (long long)slots_.size())];
    {
        std::lock_guard<std::mutex> lk(m_);
        if (s.state != 2 || s.expect != index) throw IoError("release of a frame that is not checked out");
        s.state = 0;
        s.err.clear();
        s.expect += (long long)slots_.size();
    }
    cv_free_.notify_all();
}

}  // namespace bp
