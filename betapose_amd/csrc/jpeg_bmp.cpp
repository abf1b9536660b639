// JPEG (baseline sequential and progressive, Huffman, 8-bit) and BMP (uncompressed) decoding to interleaved RGB u8 for the
// Darknet-API-compatible detector: the reference's Detector::detect(image file) loads images through Darknet's
// load_image_color -> stb_image v2.16 (train_YOLO/src/image.c:1820-1875, the vendored stb_image.h), and a detector fed
// different pixels is a different detector, so the arithmetic that decides the pixels follows stb_image's published
// choices exactly:
//   * inverse DCT = IJG "islow" with 12-bit constants, column pass kept at 2 extra bits (>> 10), row pass >> 17 with the
//     +128 level shift folded into the rounding term;
//   * chroma up-sampling = the "fancy" triangle filters: (3 near + far + 2) >> 2 along one axis, (9,3,3,1) / 16 for 2x2;
//     other ratios replicate;
//   * YCbCr -> RGB in 20-bit fixed point with the coefficients rounded to 12 bits and the cb term of G masked to its
//     upper 16 bits.
// Own structure (bit reader, canonical-code tables indexed by length, MCU walk); pinned bit-for-bit against the
// reference's compiled Darknet-C (oracle/_ref, load_image_color) in tests/test_image_codecs.py.
// Progressive frames (SOF2) keep every component's coefficients across the scans -- DC first / refinement, AC spectral
// bands with successive approximation and end-of-band runs -- and are de-quantised (in 16-bit arithmetic, as stb does)
// and inverse-transformed once, when the stream ends.
// Arithmetic-coded, 12-bit and CMYK files are rejected with a clear error (the reference's LineMod frames are PNG).
#include <cstdint>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

namespace bp {

namespace {

struct HuffTable {
    // canonical code: for every length 1..16 the first code, the index of its first symbol and the symbol count
    int first_code[17], first_sym[17], count[17];
    uint8_t sym[256];
    bool present = false;
};

struct Component {
    int id = 0, h = 1, v = 1, tq = 0, td = 0, ta = 0;
    int w2 = 0, h2 = 0;          // plane size padded to whole MCUs
    int dc_pred = 0;
    std::vector<uint8_t> plane;
    std::vector<short> coeff;    // progressive only: [block row][block column][64], natural order, not yet de-quantised
};

struct BitReader {
    const uint8_t* p;
    const uint8_t* end;
    uint32_t acc = 0;
    int nbits = 0;
    bool hit_marker = false;
    void reset() { acc = 0; nbits = 0; hit_marker = false; }
    void fill() {
        while (nbits <= 24) {
            int b = 0;
            if (!hit_marker && p < end) {
                b = *p++;
                if (b == 0xFF) {
                    int c = p < end ? *p : 0;
                    if (c == 0) ++p;                       // stuffed zero
                    else { hit_marker = true; --p; b = 0; }   // a marker: feed zeros, leave it for the caller
                }
            }
            acc |= (uint32_t)b << (24 - nbits);
            nbits += 8;
        }
    }
    int bits(int n) {
        if (n == 0) return 0;
        if (nbits < n) fill();
        const int v = (int)(acc >> (32 - n));
        acc <<= n;
        nbits -= n;
        return v;
    }
};

inline uint8_t clamp8(int x) { return (uint8_t)(x < 0 ? 0 : (x > 255 ? 255 : x)); }

const uint8_t kZigzag[64] = {0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48,
                             41, 34, 27, 20, 13, 6, 7, 14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22,
                             15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

// ---- inverse DCT.  stb_image's "islow" 8-point pass is a fixed LINEAR map of its eight inputs with integer
// coefficients (every constant is a float rounded to 12 fractional bits by (int)(x * 4096 + 0.5), truncation toward zero
// included), and integer arithmetic is exact, so the factored butterfly it is usually written as and the plain
// matrix form below give the same 32-bit results as long as nothing overflows (nothing does: |coefficients| < 2^14,
// |inputs| < 2^17).  Even half  E = A_even * (s0, s4, s2, s6), odd half  O = A_odd * (s1, s3, s5, s7);
// outputs  y[i] = E[i] + O[i],  y[7 - i] = E[i] - O[i].
struct Islow8 {
    int even26[4][2];    // contributions of (s2, s6) to E[0..3]
    int odd[4][4];       // O[i] from (s1, s3, s5, s7)
    static int q12(double x) { return (int)(x * 4096 + 0.5); }
    Islow8() {
        const int r = q12(0.5411961f), r6 = q12(-1.847759065f), r2 = q12(0.765366865f);
        const int hi2 = r + r2, hi6 = r, lo2 = r, lo6 = r + r6;      // t3 = s2 (r + r2) + s6 r ;  t2 = s2 r + s6 (r + r6)
        const int e[4][2] = {{hi2, hi6}, {lo2, lo6}, {-lo2, -lo6}, {-hi2, -hi6}};
        for (int i = 0; i < 4; ++i) { even26[i][0] = e[i][0]; even26[i][1] = e[i][1]; }
        const int k = q12(1.175875602f);
        const int d7 = q12(0.298631336f), d5 = q12(2.053119869f), d3 = q12(3.072711026f), d1 = q12(1.501321110f);
        const int m71 = q12(-0.899976223f), m53 = q12(-2.562915447f), m73 = q12(-1.961570560f), m51 = q12(-0.390180644f);
        // rows: O[0] pairs with (y0, y7), ..., O[3] with (y3, y4); columns s1, s3, s5, s7
        const int o[4][4] = {
            {d1 + k + m71 + m51, k,                 k + m51,            k + m71},
            {k,                  d3 + k + m53 + m73, k + m53,            k + m73},
            {k + m51,            k + m53,            d5 + k + m53 + m51, k},
            {k + m71,            k + m73,            k,                  d7 + k + m71 + m73}};
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 4; ++j) odd[i][j] = o[i][j];
    }
    // y[0..7] (before the caller's rounding shift); `bias` is added to the even half (rounding term / level shift)
    // (unsigned arithmetic: a corrupt stream can push the sums past 32 bits, and wrapping must stay defined)
    void apply(const int s[8], int bias, int y[8]) const {
        typedef unsigned U;
        const U sum = ((U)s[0] + (U)s[4]) * 4096u + (U)bias, dif = ((U)s[0] - (U)s[4]) * 4096u + (U)bias;
        const U base[4] = {sum, dif, dif, sum};
        for (int i = 0; i < 4; ++i) {
            const U E = base[i] + (U)s[2] * (U)even26[i][0] + (U)s[6] * (U)even26[i][1];
            const U O = (U)s[1] * (U)odd[i][0] + (U)s[3] * (U)odd[i][1] + (U)s[5] * (U)odd[i][2] + (U)s[7] * (U)odd[i][3];
            y[i] = (int)(E + O);
            y[7 - i] = (int)(E - O);
        }
    }
};
const Islow8 kIslow;

// 8x8 block: columns first, kept at 2 extra fractional bits (>> 10 with rounding), then rows (>> 17, the +128 level
// shift folded into the rounding term), clamped to u8.  A column whose AC terms are all zero is its DC term * 4.
void idct_block(uint8_t* out, int stride, const short* d) {
    int mid[8][8];                                   // [row][column] after the column pass
    for (int col = 0; col < 8; ++col) {
        int s[8], ac = 0;
        for (int r = 0; r < 8; ++r) { s[r] = d[8 * r + col]; if (r) ac |= s[r]; }
        if (ac == 0) {
            for (int r = 0; r < 8; ++r) mid[r][col] = (int)((unsigned)s[0] * 4u);
            continue;
        }
        int y[8];
        kIslow.apply(s, 512, y);
        for (int r = 0; r < 8; ++r) mid[r][col] = y[r] >> 10;
    }
    for (int row = 0; row < 8; ++row) {
        int y[8];
        kIslow.apply(mid[row], 65536 + (128 << 17), y);
        uint8_t* o = out + (size_t)row * stride;
        for (int c = 0; c < 8; ++c) o[c] = clamp8(y[c] >> 17);
    }
}

int huff_decode(BitReader& br, const HuffTable& h) {
    int code = 0;
    for (int len = 1; len <= 16; ++len) {
        code = (code << 1) | br.bits(1);
        if (h.count[len] && code - h.first_code[len] < h.count[len] && code >= h.first_code[len])
            return h.sym[h.first_sym[len] + code - h.first_code[len]];
    }
    throw std::runtime_error("JPEG: bad Huffman code");
}

inline int extend(int v, int n) { return v < (1 << (n - 1)) ? v - (1 << n) + 1 : v; }

struct Decoder {
    const uint8_t* data;
    size_t size;
    int W = 0, H = 0, ncomp = 0, hmax = 1, vmax = 1, restart = 0;
    bool progressive = false;
    int ss = 0, se = 63, ah = 0, al = 0;     // the current scan's spectral band and successive-approximation bits
    int eob_run = 0;                          // blocks still covered by the last end-of-band run
    uint16_t quant[4][64];
    bool have_q[4] = {false, false, false, false};
    HuffTable dc[4], ac[4];
    Component comp[3];

    static int be16(const uint8_t* p) { return (p[0] << 8) | p[1]; }

    void parse_dqt(const uint8_t* p, int len) {
        while (len > 0) {
            const int pq = p[0] >> 4, tq = p[0] & 15;
            if (tq > 3 || pq > 1) throw std::runtime_error("JPEG: bad DQT");
            if (len < 1 + (pq ? 128 : 64)) throw std::runtime_error("JPEG: truncated segment (DQT)");
            ++p; --len;
            for (int i = 0; i < 64; ++i) {
                quant[tq][kZigzag[i]] = pq ? (uint16_t)be16(p + 2 * i) : p[i];
            }
            p += pq ? 128 : 64;
            len -= pq ? 128 : 64;
            have_q[tq] = true;
        }
    }
    void parse_dht(const uint8_t* p, int len) {
        while (len > 0) {
            if (len < 17) throw std::runtime_error("JPEG: truncated segment (DHT)");
            const int tc = p[0] >> 4, th = p[0] & 15;
            if (tc > 1 || th > 3) throw std::runtime_error("JPEG: bad DHT");
            HuffTable& h = tc ? ac[th] : dc[th];
            int total = 0, code = 0;
            for (int l = 1; l <= 16; ++l) {
                h.count[l] = p[l];
                h.first_sym[l] = total;
                h.first_code[l] = code;
                total += p[l];
                code = (code + p[l]) << 1;
            }
            if (total > 256) throw std::runtime_error("JPEG: bad DHT");
            if (len < 17 + total) throw std::runtime_error("JPEG: truncated segment (DHT)");
            std::memcpy(h.sym, p + 17, total);
            h.present = true;
            p += 17 + total;
            len -= 17 + total;
        }
    }
    void parse_sof(const uint8_t* p, int len) {
        if (len < 6) throw std::runtime_error("JPEG: truncated segment (SOF)");
        if (p[0] != 8) throw std::runtime_error("JPEG: only 8-bit samples are supported");
        H = be16(p + 1); W = be16(p + 3); ncomp = p[5];
        if (H <= 0 || W <= 0) throw std::runtime_error("JPEG: bad dimensions");
        if (ncomp != 1 && ncomp != 3) throw std::runtime_error("JPEG: only grey and YCbCr images are supported");
        if (len < 6 + 3 * ncomp) throw std::runtime_error("JPEG: bad SOF");
        if ((long long)W * H > (1ll << 28)) throw std::runtime_error("JPEG: image too large");
        for (int i = 0; i < ncomp; ++i) {
            comp[i].id = p[6 + 3 * i];
            comp[i].h = p[7 + 3 * i] >> 4;
            comp[i].v = p[7 + 3 * i] & 15;
            comp[i].tq = p[8 + 3 * i];
            if (comp[i].h < 1 || comp[i].h > 4 || comp[i].v < 1 || comp[i].v > 4 || comp[i].tq > 3) throw std::runtime_error("JPEG: bad SOF");
            hmax = std::max(hmax, comp[i].h);
            vmax = std::max(vmax, comp[i].v);
        }
        for (int i = 0; i < ncomp; ++i)
            if (hmax % comp[i].h || vmax % comp[i].v) throw std::runtime_error("JPEG: unsupported sampling factors");
    }

    void decode_block(BitReader& br, Component& c, short* blk) {
        std::memset(blk, 0, 64 * sizeof(short));
        const HuffTable& hd = dc[c.td];
        const HuffTable& ha = ac[c.ta];
        const uint16_t* q = quant[c.tq];
        const int t = huff_decode(br, hd);
        if (t > 11) throw std::runtime_error("JPEG: corrupt block (DC category)");   // 8-bit baseline: at most 11 bits
        const int diff = t ? extend(br.bits(t), t) : 0;
        // (unsigned arithmetic: a corrupt stream can run the predictor up block after block; it then wraps instead of being
        // signed-overflow UB -- round-4 advisor finding.  Valid streams never leave 16 bits.)
        c.dc_pred = (int)((unsigned)c.dc_pred + (unsigned)diff);
        blk[0] = (short)((unsigned)c.dc_pred * (unsigned)q[0]);
        for (int k = 1; k < 64;) {
            const int rs = huff_decode(br, ha);
            const int r = rs >> 4, s = rs & 15;
            if (s == 0) {
                if (rs != 0xF0) break;
                k += 16;
            } else {
                k += r;
                if (k > 63) throw std::runtime_error("JPEG: corrupt block");
                const int z = kZigzag[k++];
                blk[z] = (short)((unsigned)extend(br.bits(s), s) * (unsigned)q[z]);
            }
        }
    }

    // ---- progressive scans (ITU T.81 annex G, in the form stb_image decodes them) ----
    // DC: the first scan carries the predicted difference, scaled by 2^Al; a refinement scan adds one bit per block
    void prog_dc(BitReader& br, Component& c, short* d) {
        if (se != 0) throw std::runtime_error("JPEG: corrupt progressive scan (DC scan with an AC band)");
        if (ah == 0) {
            std::memset(d, 0, 64 * sizeof(short));
            if (!dc[c.td].present) throw std::runtime_error("JPEG: scan refers to a missing table");
            const int t = huff_decode(br, dc[c.td]);
            if (t > 15) throw std::runtime_error("JPEG: corrupt block (DC category)");
            const int diff = t ? extend(br.bits(t), t) : 0;
            c.dc_pred = (int)((unsigned)c.dc_pred + (unsigned)diff);
            d[0] = (short)((unsigned)c.dc_pred << al);
        } else if (br.bits(1)) {
            d[0] = (short)(d[0] + (short)(1 << al));
        }
    }
    // one correction bit for a coefficient that is already non-zero: move it away from zero by `bit` unless it has it
    static void refine(BitReader& br, short* p, short bit) {
        if (br.bits(1) && (*p & bit) == 0) *p = (short)(*p > 0 ? *p + bit : *p - bit);
    }
    // AC band ss..se of one block
    void prog_ac(BitReader& br, Component& c, short* d) {
        if (ss == 0) throw std::runtime_error("JPEG: corrupt progressive scan (AC scan starting at DC)");
        if (!ac[c.ta].present) throw std::runtime_error("JPEG: scan refers to a missing table");
        const HuffTable& ha = ac[c.ta];
        if (ah == 0) {
            if (eob_run) { --eob_run; return; }
            int k = ss;
            do {
                const int rs = huff_decode(br, ha);
                const int r = rs >> 4, s = rs & 15;
                if (s == 0) {
                    if (r < 15) {                          // end of band for 2^r (+ r more bits) blocks, this one included
                        eob_run = 1 << r;
                        if (r) eob_run += br.bits(r);
                        --eob_run;
                        break;
                    }
                    k += 16;
                } else {
                    k += r;
                    if (k > 63) throw std::runtime_error("JPEG: corrupt block");
                    d[kZigzag[k++]] = (short)((unsigned)extend(br.bits(s), s) << al);
                }
            } while (k <= se);
            return;
        }
        const short bit = (short)(1 << al);
        if (eob_run) {
            --eob_run;
            for (int k = ss; k <= se; ++k) {
                short* p = d + kZigzag[k];
                if (*p != 0) refine(br, p, bit);
            }
            return;
        }
        int k = ss;
        do {
            const int rs = huff_decode(br, ha);
            int r = rs >> 4, s = rs & 15;
            if (s == 0) {
                if (r < 15) {
                    eob_run = (1 << r) - 1;
                    if (r) eob_run += br.bits(r);
                    r = 64;                                // nothing new in this block: only corrections to the end of the band
                }                                          // r == 15: sixteen zero-history coefficients to skip
            } else {
                if (s != 1) throw std::runtime_error("JPEG: corrupt block (refinement)");
                s = br.bits(1) ? bit : -bit;
            }
            while (k <= se) {
                short* p = d + kZigzag[k++];
                if (*p != 0) refine(br, p, bit);
                else {
                    if (r == 0) { *p = (short)s; break; }
                    --r;
                }
            }
        } while (k <= se);
    }

    void decode_scan(const uint8_t* p, const uint8_t* end, int ns, const int* order) {
        BitReader br{p, end};
        const int mcux = (W + 8 * hmax - 1) / (8 * hmax), mcuy = (H + 8 * vmax - 1) / (8 * vmax);
        short blk[64];
        int todo = restart ? restart : 0x7fffffff;
        eob_run = 0;
        if (ns == 1) {
            // non-interleaved: the component's own 8x8 blocks in raster order
            Component& c = comp[order[0]];
            const int bw = (((W * c.h + hmax - 1) / hmax) + 7) / 8, bh = (((H * c.v + vmax - 1) / vmax) + 7) / 8;
            for (int by = 0; by < bh; ++by)
                for (int bx = 0; bx < bw; ++bx) {
                    if (progressive) {
                        short* d = c.coeff.data() + 64 * ((size_t)by * (c.w2 / 8) + bx);
                        if (ss == 0) prog_dc(br, c, d); else prog_ac(br, c, d);
                    } else {
                        decode_block(br, c, blk);
                        idct_block(c.plane.data() + (size_t)by * 8 * c.w2 + bx * 8, c.w2, blk);
                    }
                    if (--todo <= 0) { next_restart(br); todo = restart; }
                }
            return;
        }
        for (int my = 0; my < mcuy; ++my)
            for (int mx = 0; mx < mcux; ++mx) {
                for (int k = 0; k < ns; ++k) {
                    Component& c = comp[order[k]];
                    for (int y = 0; y < c.v; ++y)
                        for (int x = 0; x < c.h; ++x) {
                            if (progressive) {             // interleaved progressive scans carry DC only
                                prog_dc(br, c, c.coeff.data() + 64 * ((size_t)(my * c.v + y) * (c.w2 / 8) + mx * c.h + x));
                                continue;
                            }
                            decode_block(br, c, blk);
                            idct_block(c.plane.data() + (size_t)(my * c.v + y) * 8 * c.w2 + (mx * c.h + x) * 8, c.w2, blk);
                        }
                }
                if (--todo <= 0) { next_restart(br); todo = restart; }
            }
    }
    // progressive: every scan has been merged into the coefficients -- de-quantise (16-bit products, as the reference's
    // loader does) and inverse-transform the blocks that carry image data
    void finish_progressive() {
        for (int i = 0; i < ncomp; ++i) {
            Component& c = comp[i];
            if (!have_q[c.tq]) throw std::runtime_error("JPEG: scan refers to a missing table");
            const uint16_t* q = quant[c.tq];
            const int bw = (((W * c.h + hmax - 1) / hmax) + 7) / 8, bh = (((H * c.v + vmax - 1) / vmax) + 7) / 8;
            for (int by = 0; by < bh; ++by)
                for (int bx = 0; bx < bw; ++bx) {
                    short* d = c.coeff.data() + 64 * ((size_t)by * (c.w2 / 8) + bx);
                    for (int k = 0; k < 64; ++k) d[k] = (short)(d[k] * q[k]);
                    idct_block(c.plane.data() + (size_t)by * 8 * c.w2 + bx * 8, c.w2, d);
                }
        }
    }
    void next_restart(BitReader& br) {
        // byte-align, expect RSTn, reset predictors
        const uint8_t* p = br.p;
        while (p + 1 < br.end && !(p[0] == 0xFF && p[1] >= 0xD0 && p[1] <= 0xD7)) {
            if (p[0] == 0xFF && p[1] != 0 && !(p[1] >= 0xD0 && p[1] <= 0xD7)) return;   // another marker: end of scan data
            ++p;
        }
        if (p + 1 >= br.end) return;
        br.p = p + 2;
        br.reset();
        eob_run = 0;
        for (int i = 0; i < ncomp; ++i) comp[i].dc_pred = 0;
    }

    void run(std::vector<uint8_t>& rgb, int* oh, int* ow) {
        if (size < 4 || data[0] != 0xFF || data[1] != 0xD8) throw std::runtime_error("not a JPEG stream");
        const uint8_t* p = data + 2;
        const uint8_t* end = data + size;
        bool have_sof = false, scanned = false;
        while (p + 4 <= end) {
            if (p[0] != 0xFF) { ++p; continue; }
            const int m = p[1];
            if (m == 0xFF) { ++p; continue; }
            if (m == 0xD9) break;
            if (m == 0x01 || (m >= 0xD0 && m <= 0xD7)) { p += 2; continue; }
            const int len = be16(p + 2);
            if (len < 2 || p + 2 + len > end) throw std::runtime_error("JPEG: truncated segment");
            const uint8_t* body = p + 4;
            if (m == 0xDB) parse_dqt(body, len - 2);
            else if (m == 0xC4) parse_dht(body, len - 2);
            else if (m == 0xDD) {
                if (len < 4) throw std::runtime_error("JPEG: truncated segment (DRI)");
                restart = be16(body);
            }
            else if (m == 0xC0 || m == 0xC1 || m == 0xC2) {
                if (have_sof) throw std::runtime_error("JPEG: more than one frame header");
                parse_sof(body, len - 2);
                have_sof = true;
                progressive = m == 0xC2;
                for (int i = 0; i < ncomp; ++i) {
                    Component& c = comp[i];
                    const int mcux = (W + 8 * hmax - 1) / (8 * hmax), mcuy = (H + 8 * vmax - 1) / (8 * vmax);
                    c.w2 = mcux * c.h * 8;
                    c.h2 = mcuy * c.v * 8;
                    c.plane.assign((size_t)c.w2 * c.h2, 0);
                    if (progressive) c.coeff.assign((size_t)c.w2 * c.h2, 0);
                }
            } else if (m == 0xC9 || m == 0xCA || m == 0xCB) throw std::runtime_error("JPEG: arithmetic coding is not supported");
            else if (m >= 0xC3 && m <= 0xCF && m != 0xC4 && m != 0xC8 && m != 0xCC) throw std::runtime_error("JPEG: unsupported frame type");
            else if (m == 0xDA) {
                if (!have_sof) throw std::runtime_error("JPEG: scan before frame header");
                if (len < 3) throw std::runtime_error("JPEG: truncated segment (SOS)");
                const int ns = body[0];
                if (ns < 1 || ns > ncomp || len < 6 + 2 * ns) throw std::runtime_error("JPEG: bad SOS");
                int order[3];
                for (int k = 0; k < ns; ++k) {
                    int ci = -1;
                    for (int i = 0; i < ncomp; ++i)
                        if (comp[i].id == body[1 + 2 * k]) ci = i;
                    if (ci < 0) throw std::runtime_error("JPEG: bad SOS component");
                    comp[ci].td = body[2 + 2 * k] >> 4;
                    comp[ci].ta = body[2 + 2 * k] & 15;
                    if (comp[ci].td > 3 || comp[ci].ta > 3) throw std::runtime_error("JPEG: bad SOS");
                    // a progressive scan needs only the table of its own band (checked where it is used); the
                    // quantisation tables are needed when the coefficients are finished
                    if (!progressive && (!dc[comp[ci].td].present || !ac[comp[ci].ta].present || !have_q[comp[ci].tq]))
                        throw std::runtime_error("JPEG: scan refers to a missing table");
                    comp[ci].dc_pred = 0;
                    order[k] = ci;
                }
                ss = body[1 + 2 * ns]; se = body[2 + 2 * ns];
                ah = body[3 + 2 * ns] >> 4; al = body[3 + 2 * ns] & 15;
                if (progressive) {
                    if (ss > 63 || se > 63 || ss > se || ah > 13 || al > 13) throw std::runtime_error("JPEG: bad SOS");
                    if (ns > 1 && ss != 0) throw std::runtime_error("JPEG: corrupt progressive scan (interleaved AC scan)");
                } else if (ss != 0 || ah != 0 || al != 0) throw std::runtime_error("JPEG: bad SOS");
                const uint8_t* sp = p + 2 + len;
                decode_scan(sp, end, ns, order);
                scanned = true;
                // skip the entropy-coded data to the next real marker
                const uint8_t* q = sp;
                while (q + 1 < end && !(q[0] == 0xFF && q[1] != 0 && !(q[1] >= 0xD0 && q[1] <= 0xD7))) ++q;
                p = q;
                continue;
            }
            p += 2 + len;
        }
        if (!have_sof || !scanned) throw std::runtime_error("JPEG: no image data");
        if (progressive) finish_progressive();
        assemble(rgb);
        *oh = H; *ow = W;
    }

    // up-sampling (per output row, from the component's own rows) + colour conversion
    void assemble(std::vector<uint8_t>& rgb) {
        rgb.assign((size_t)W * H * 3, 0);
        std::vector<uint8_t> line[3];
        const uint8_t* cur[3] = {nullptr, nullptr, nullptr};
        for (int k = 0; k < ncomp; ++k) line[k].resize((size_t)W + 8 * hmax + 3);
        for (int j = 0; j < H; ++j) {
            for (int k = 0; k < ncomp; ++k) {
                const Component& c = comp[k];
                const int hs = hmax / c.h, vs = vmax / c.v;
                const int rows = (H * c.v + vmax - 1) / vmax;      // rows that carry image data
                const int wl = (W + hs - 1) / hs;
                const uint8_t* near;
                const uint8_t* far;
                if (vs == 2) {
                    const int n = j >> 1;
                    const int f = (j & 1) ? std::min(n + 1, rows - 1) : std::max(n - 1, 0);
                    near = c.plane.data() + (size_t)std::min(n, rows - 1) * c.w2;
                    far = c.plane.data() + (size_t)f * c.w2;
                } else {
                    const int n = std::min(j / vs, rows - 1);
                    near = far = c.plane.data() + (size_t)n * c.w2;
                }
                uint8_t* o = line[k].data();
                if (hs == 1 && vs == 1) { cur[k] = near; continue; }
                if (hs == 1 && vs == 2) {
                    for (int i = 0; i < wl; ++i) o[i] = (uint8_t)((3 * near[i] + far[i] + 2) >> 2);
                } else if (hs == 2 && vs == 1) {
                    if (wl == 1) { o[0] = o[1] = near[0]; }
                    else {
                        o[0] = near[0];
                        o[1] = (uint8_t)((near[0] * 3 + near[1] + 2) >> 2);
                        int i = 1;
                        for (; i < wl - 1; ++i) {
                            const int n3 = 3 * near[i] + 2;
                            o[2 * i] = (uint8_t)((n3 + near[i - 1]) >> 2);
                            o[2 * i + 1] = (uint8_t)((n3 + near[i + 1]) >> 2);
                        }
                        o[2 * i] = (uint8_t)((near[wl - 2] * 3 + near[wl - 1] + 2) >> 2);
                        o[2 * i + 1] = near[wl - 1];
                    }
                } else if (hs == 2 && vs == 2) {
                    if (wl == 1) { o[0] = o[1] = (uint8_t)((3 * near[0] + far[0] + 2) >> 2); }
                    else {
                        int t1 = 3 * near[0] + far[0];
                        o[0] = (uint8_t)((t1 + 2) >> 2);
                        for (int i = 1; i < wl; ++i) {
                            const int t0 = t1;
                            t1 = 3 * near[i] + far[i];
                            o[2 * i - 1] = (uint8_t)((3 * t0 + t1 + 8) >> 4);
                            o[2 * i] = (uint8_t)((3 * t1 + t0 + 8) >> 4);
                        }
                        o[2 * wl - 1] = (uint8_t)((t1 + 2) >> 2);
                    }
                } else {
                    for (int i = 0; i < wl; ++i)
                        for (int r = 0; r < hs; ++r) o[i * hs + r] = near[i];
                }
                cur[k] = o;
            }
            uint8_t* out = rgb.data() + (size_t)j * W * 3;
            if (ncomp == 1) {
                for (int i = 0; i < W; ++i) { out[3 * i] = out[3 * i + 1] = out[3 * i + 2] = cur[0][i]; }
            } else {
                auto f2f = [](float x) { return ((int)(x * 4096.0f + 0.5f)) << 8; };
                const int kr = f2f(1.40200f), kg1 = f2f(0.71414f), kg2 = f2f(0.34414f), kb = f2f(1.77200f);
                for (int i = 0; i < W; ++i) {
                    const int yf = (cur[0][i] << 20) + (1 << 19);
                    const int cr = cur[2][i] - 128, cb = cur[1][i] - 128;
                    int r = yf + cr * kr;
                    int g = yf + (cr * -kg1) + (int)(((unsigned)(cb * -kg2)) & 0xffff0000u);
                    int b = yf + cb * kb;
                    r >>= 20; g >>= 20; b >>= 20;
                    out[3 * i] = clamp8(r); out[3 * i + 1] = clamp8(g); out[3 * i + 2] = clamp8(b);
                }
            }
        }
    }
};

}  // namespace

void jpeg_decode_rgb(const uint8_t* data, size_t n, std::vector<uint8_t>& rgb, int* h, int* w) {
    Decoder d;
    d.data = data;
    d.size = n;
    std::memset(d.quant, 0, sizeof(d.quant));
    d.run(rgb, h, w);
}

// uncompressed BMP (BITMAPINFOHEADER and later; 8-bit palette, 24-bit, 32-bit; bottom-up or top-down) -> RGB
void bmp_decode_rgb(const uint8_t* d, size_t n, std::vector<uint8_t>& rgb, int* oh, int* ow) {
    auto le32 = [&](size_t o) { return (uint32_t)d[o] | ((uint32_t)d[o + 1] << 8) | ((uint32_t)d[o + 2] << 16) | ((uint32_t)d[o + 3] << 24); };
    auto le16 = [&](size_t o) { return (uint32_t)d[o] | ((uint32_t)d[o + 1] << 8); };
    if (n < 54 || d[0] != 'B' || d[1] != 'M') throw std::runtime_error("not a BMP file");
    const uint32_t off = le32(10), hsz = le32(14);
    if (hsz < 40) throw std::runtime_error("BMP: OS/2 headers are not supported");
    const int W = (int)le32(18);
    int H = (int)le32(22);
    const bool flip = H > 0;
    if (H < 0) H = -H;
    const int bpp = (int)le16(28);
    const uint32_t compr = le32(30);
    if (W <= 0 || H <= 0 || (long long)W * H > (1ll << 28)) throw std::runtime_error("BMP: bad dimensions");
    if (!(compr == 0 || (compr == 3 && bpp == 32))) throw std::runtime_error("BMP: compressed files are not supported");
    if (bpp != 8 && bpp != 24 && bpp != 32) throw std::runtime_error("BMP: only 8 / 24 / 32 bits per pixel are supported");
    const size_t stride = (((size_t)W * bpp + 31) / 32) * 4;
    if (off + stride * H > n) throw std::runtime_error("BMP: truncated");
    const size_t pal = 14 + hsz;
    rgb.assign((size_t)W * H * 3, 0);
    for (int y = 0; y < H; ++y) {
        const uint8_t* row = d + off + stride * (flip ? H - 1 - y : y);
        uint8_t* o = rgb.data() + (size_t)y * W * 3;
        for (int x = 0; x < W; ++x) {
            if (bpp == 8) {
                const uint8_t* e = d + pal + 4 * row[x];
                if (pal + 4 * row[x] + 3 > n) throw std::runtime_error("BMP: palette out of range");
                o[3 * x] = e[2]; o[3 * x + 1] = e[1]; o[3 * x + 2] = e[0];
            } else {
                const uint8_t* e = row + (size_t)x * (bpp / 8);
                o[3 * x] = e[2]; o[3 * x + 1] = e[1]; o[3 * x + 2] = e[0];
            }
        }
    }
    *oh = H; *ow = W;
}

}  // namespace bp
