// Darknet-API-compatible detector on the HIP engine (SURVEY §8b "existing native ABI precedent"):
//   * bp_darknet_*            -- cfg with a [net] block + .weights -> detections the way the reference's
//                                Detector::detect produces them (train_YOLO/src/yolo_v2_class.cpp:239-317);
//   * init / detect_image / detect_mat / dispose / get_device_count / get_device_name
//                             -- the six extern "C" symbols of train_YOLO/src/yolo_v2_class.hpp:49-54.
// Chain restated from the reference: load image as planar RGB float /255 (image.c load_image_stb) -> two-pass bilinear
// resize_image (image.c:1725-1767) -> network forward with BatchNorm folded as scale/(sqrt(var)+1e-6) -> per [yolo]
// layer, cell-major / anchor-minor candidates with objectness > thresh, prob = objectness*class (0 when <= thresh),
// boxes relative to the image (yolo_layer.c:84-92,365-392) -> per-class sort + IoU suppression, nms 0.4
// (box.c:331-363) -> bbox_t in image pixels.
// Image decoding: PNG (csrc/frame_io.cpp), baseline JPEG and uncompressed BMP (csrc/jpeg_bmp.cpp, stb_image's pixel
// arithmetic restated and pinned against the reference's compiled load_image_color).
#include "../../include/betapose_hip.h"
#include "../../include/yolo_v2_class_compat.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <fstream>
#include <memory>
#include <mutex>
#include <sstream>
#include <string>
#include <vector>

#include <hip/hip_runtime.h>

#include "frame_io.h"

namespace {

thread_local std::string g_cerr;

struct Candidate {
    float x, y, w, h;          // relative to the image (0..1), centre-size
    float objectness;
    std::vector<float> prob;   // per class
};

// image.c:1725-1767 -- planar [c][h][w] float in, planar [c][oh][ow] out
std::vector<float> darknet_resize(const float* im, int w, int h, int c, int ow, int oh) {
    std::vector<float> part((size_t)c * h * ow), out((size_t)c * oh * ow);
    const float w_scale = (float)(w - 1) / (ow - 1);
    const float h_scale = (float)(h - 1) / (oh - 1);
    for (int k = 0; k < c; ++k)
        for (int r = 0; r < h; ++r) {
            const float* src = im + ((size_t)k * h + r) * w;
            float* dst = part.data() + ((size_t)k * h + r) * ow;
            for (int x = 0; x < ow; ++x) {
                if (x == ow - 1 || w == 1) {
                    dst[x] = src[w - 1];
                } else {
                    const float sx = x * w_scale;
                    const int ix = (int)sx;
                    const float dx = sx - ix;
                    dst[x] = (1 - dx) * src[ix] + dx * src[ix + 1];
                }
            }
        }
    for (int k = 0; k < c; ++k)
        for (int r = 0; r < oh; ++r) {
            const float sy = r * h_scale;
            const int iy = (int)sy;
            const float dy = sy - iy;
            float* dst = out.data() + ((size_t)k * oh + r) * ow;
            const float* p0 = part.data() + ((size_t)k * h + iy) * ow;
            for (int x = 0; x < ow; ++x) dst[x] = (1 - dy) * p0[x];
            if (r == oh - 1 || h == 1) continue;
            const float* p1 = p0 + ow;
            for (int x = 0; x < ow; ++x) dst[x] += dy * p1[x];
        }
    return out;
}

float overlap1d(float x1, float w1, float x2, float w2) {
    const float l = std::max(x1 - w1 / 2, x2 - w2 / 2), r = std::min(x1 + w1 / 2, x2 + w2 / 2);
    return r - l;
}
float box_iou(const Candidate& a, const Candidate& b) {      // box.c:67-97
    const float w = overlap1d(a.x, a.w, b.x, b.w), h = overlap1d(a.y, a.h, b.y, b.h);
    const float inter = (w < 0 || h < 0) ? 0.f : w * h;
    return inter / (a.w * a.h + b.w * b.h - inter);
}

}  // namespace

struct bp_darknet {
    bp_yolo* net = nullptr;
    int netw = 0, neth = 0, classes = 0, rows = 0, attrs = 0, device = 0;
    std::vector<int> head_grid;   // grid size per [yolo] layer, in network order
    float* d_img = nullptr;
    float* d_pred = nullptr;
    std::vector<float> h_pred;
    ~bp_darknet() {
        if (d_img) (void)hipFree(d_img);
        if (d_pred) (void)hipFree(d_pred);
        if (net) bp_yolo_destroy(net);
    }
};

#define DK_TRY try {
#define DK_CATCH                                   \
    }                                              \
    catch (const std::exception& e) {              \
        g_cerr = e.what();                         \
        return -1;                                 \
    }                                              \
    catch (...) {                                  \
        g_cerr = "unknown error";                  \
        return -1;                                 \
    }

static void fail_if(int rc) {
    if (rc != 0) throw std::runtime_error(bp_last_error());
}

extern "C" {

const char* bp_darknet_last_error(void) { return g_cerr.c_str(); }

int bp_darknet_create(const char* cfg_path, const char* weights_path, int device, bp_darknet** out) {
    DK_TRY
    if (!cfg_path || !weights_path || !out) throw std::runtime_error("null argument");
    std::ifstream f(cfg_path);
    if (!f) throw std::runtime_error(std::string("cannot open ") + cfg_path);
    std::stringstream ss;
    ss << f.rdbuf();
    const std::string cfg = ss.str();
    // [net] width / height (parser.c: parse_net_options); everything else is the engine's cfg parser
    int w = 0, h = 0;
    {
        std::istringstream in(cfg);
        std::string line, section;
        while (std::getline(in, line)) {
            line.erase(std::remove_if(line.begin(), line.end(), [](unsigned char ch) { return std::isspace(ch); }), line.end());
            if (line.empty() || line[0] == '#' || line[0] == ';') continue;
            if (line[0] == '[') { section = line; continue; }
            if (section != "[net]" && section != "[network]") continue;
            const size_t eq = line.find('=');
            if (eq == std::string::npos) continue;
            const std::string key = line.substr(0, eq), val = line.substr(eq + 1);
            if (key == "width") w = std::atoi(val.c_str());
            if (key == "height") h = std::atoi(val.c_str());
        }
    }
    if (w <= 0 || h <= 0) throw std::runtime_error("cfg has no [net] width/height");
    if (w != h) throw std::runtime_error("only square network inputs are supported");
    std::unique_ptr<bp_darknet> d(new bp_darknet);
    d->device = device;
    fail_if(bp_yolo_create_darknet(cfg_path, weights_path, w, 1, device, &d->net));
    d->netw = w; d->neth = h;
    d->rows = bp_yolo_rows(d->net);
    d->attrs = bp_yolo_attrs(d->net);
    d->classes = d->attrs - 5;
    // rows = sum over heads of 3*g*g, heads in network order with strides 32, 16, 8 ... (cfg order)
    {
        int left = d->rows;
        for (int stride = 32; stride >= 8 && left > 0; stride /= 2) {
            const int g = w / stride;
            d->head_grid.push_back(g);
            left -= 3 * g * g;
        }
        if (left != 0) throw std::runtime_error("unexpected [yolo] head layout (expected 3 anchors at strides 32/16/8)");
    }
    if (hipSetDevice(device) != hipSuccess) throw std::runtime_error("hipSetDevice failed");
    if (hipMalloc(&d->d_img, (size_t)3 * w * h * sizeof(float)) != hipSuccess ||
        hipMalloc(&d->d_pred, (size_t)d->rows * d->attrs * sizeof(float)) != hipSuccess)
        throw std::runtime_error("device allocation failed");
    d->h_pred.resize((size_t)d->rows * d->attrs);
    *out = d.release();
    return 0;
    DK_CATCH
}

void bp_darknet_destroy(bp_darknet* d) { delete d; }
int bp_darknet_width(const bp_darknet* d) { return d ? d->netw : -1; }
int bp_darknet_height(const bp_darknet* d) { return d ? d->neth : -1; }
int bp_darknet_classes(const bp_darknet* d) { return d ? d->classes : -1; }

int bp_darknet_detect_rgb(bp_darknet* d, const float* planar_rgb, int w, int h, float thresh, float nms, bp_bbox* out,
                          int cap) {
    DK_TRY
    if (!d || !planar_rgb || (!out && cap > 0)) throw std::runtime_error("null argument");
    if (w <= 0 || h <= 0) throw std::runtime_error("bad image size");
    if (hipSetDevice(d->device) != hipSuccess) throw std::runtime_error("hipSetDevice failed");
    std::vector<float> sized;
    const float* X = planar_rgb;
    if (w != d->netw || h != d->neth) {           // yolo_v2_class.cpp:263-268
        sized = darknet_resize(planar_rgb, w, h, 3, d->netw, d->neth);
        X = sized.data();
    }
    if (hipMemcpy(d->d_img, X, (size_t)3 * d->netw * d->neth * sizeof(float), hipMemcpyHostToDevice) != hipSuccess)
        throw std::runtime_error("upload failed");
    fail_if(bp_yolo_forward(d->net, d->d_img, 1, d->d_pred, nullptr));
    if (hipMemcpy(d->h_pred.data(), d->d_pred, d->h_pred.size() * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess)
        throw std::runtime_error("download failed");

    // candidates in Darknet's order: head -> cell -> anchor (rows are stored head -> anchor -> cell)
    std::vector<Candidate> dets;
    const int A = d->attrs, C = d->classes;
    size_t head_off = 0;
    for (int g : d->head_grid) {
        for (int cell = 0; cell < g * g; ++cell)
            for (int a = 0; a < 3; ++a) {
                const float* r = d->h_pred.data() + (head_off + (size_t)a * g * g + cell) * A;
                const float obj = r[4];
                if (!(obj > thresh)) continue;                     // yolo_layer.c:377
                Candidate c;
                c.x = r[0] / d->netw; c.y = r[1] / d->neth; c.w = r[2] / d->netw; c.h = r[3] / d->neth;
                c.objectness = obj;
                c.prob.resize(C);
                for (int j = 0; j < C; ++j) {
                    const float p = obj * r[5 + j];
                    c.prob[j] = p > thresh ? p : 0.f;              // yolo_layer.c:384-385
                }
                dets.push_back(std::move(c));
            }
        head_off += (size_t)3 * g * g;
    }
    // box.c:331-363 (entries with objectness 0 cannot occur here: they were filtered above)
    if (nms > 0) {
        for (int k = 0; k < C; ++k) {
            std::stable_sort(dets.begin(), dets.end(), [k](const Candidate& a, const Candidate& b) { return a.prob[k] > b.prob[k]; });
            for (size_t i = 0; i < dets.size(); ++i) {
                if (dets[i].prob[k] == 0) continue;
                for (size_t j = i + 1; j < dets.size(); ++j)
                    if (box_iou(dets[i], dets[j]) > nms) dets[j].prob[k] = 0;
            }
        }
    }
    int n = 0;
    for (const Candidate& c : dets) {                              // yolo_v2_class.cpp:293-311
        int best = 0;
        for (int j = 1; j < C; ++j)
            if (c.prob[j] > c.prob[best]) best = j;
        const float prob = c.prob[best];
        if (!(prob > thresh)) continue;
        if (n < cap) {
            bp_bbox b;
            b.x = (unsigned int)std::max(0.0, ((double)c.x - c.w / 2.) * w);
            b.y = (unsigned int)std::max(0.0, ((double)c.y - c.h / 2.) * h);
            b.w = (unsigned int)(c.w * w);
            b.h = (unsigned int)(c.h * h);
            b.prob = prob;
            b.obj_id = (unsigned int)best;
            b.track_id = 0;
            b.frames_counter = 0;
            out[n] = b;
        }
        ++n;
    }
    return n;
    DK_CATCH
}

// encoded image (PNG, baseline JPEG, uncompressed BMP: sniffed from the first bytes) -> interleaved RGB u8
static void decode_image_rgb(const unsigned char* data, size_t n, std::vector<uint8_t>& rgb, int* h, int* w) {
    if (n >= 8 && std::memcmp(data, "\x89PNG\r\n\x1a\n", 8) == 0) {
        bp::png_info(data, n, h, w, nullptr);
        std::vector<uint8_t> bgr((size_t)*h * *w * 3), scratch;
        bp::png_decode_bgr(data, n, bgr.data(), bgr.size(), h, w, scratch);
        rgb.resize(bgr.size());
        for (size_t i = 0; i + 2 < bgr.size(); i += 3) { rgb[i] = bgr[i + 2]; rgb[i + 1] = bgr[i + 1]; rgb[i + 2] = bgr[i]; }
    } else if (n >= 3 && data[0] == 0xFF && data[1] == 0xD8) {
        bp::jpeg_decode_rgb(data, n, rgb, h, w);
    } else if (n >= 2 && data[0] == 'B' && data[1] == 'M') {
        bp::bmp_decode_rgb(data, n, rgb, h, w);
    } else {
        throw std::runtime_error("unsupported image format (PNG, baseline JPEG and uncompressed BMP are decoded)");
    }
}

int bp_darknet_detect_image(bp_darknet* d, const unsigned char* data, size_t n, float thresh, float nms, bp_bbox* out, int cap) {
    DK_TRY
    if (!d || !data) throw std::runtime_error("null argument");
    int h = 0, w = 0;
    std::vector<uint8_t> rgb;
    decode_image_rgb(data, n, rgb, &h, &w);
    std::vector<float> im((size_t)3 * h * w);                      // image.c load_image_stb: planar R,G,B / 255
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x)
            for (int k = 0; k < 3; ++k) im[((size_t)k * h + y) * w + x] = (float)(rgb[((size_t)y * w + x) * 3 + k] / 255.);
    return bp_darknet_detect_rgb(d, im.data(), w, h, thresh, nms, out, cap);
    DK_CATCH
}

int bp_darknet_detect_png(bp_darknet* d, const unsigned char* png, size_t n, float thresh, float nms, bp_bbox* out, int cap) {
    return bp_darknet_detect_image(d, png, n, thresh, nms, out, cap);      // kept name: any supported format
}

// decoded pixels of an encoded image, interleaved RGB u8 (test / tool hook for the decoders above): *h, *w receive the
// size; out may be null to query it
int bp_image_decode_rgb(const unsigned char* data, size_t n, unsigned char* out, size_t cap, int* h, int* w) {
    DK_TRY
    if (!data || !h || !w) throw std::runtime_error("null argument");
    std::vector<uint8_t> rgb;
    decode_image_rgb(data, n, rgb, h, w);
    if (out) {
        if (cap < rgb.size()) throw std::runtime_error("output buffer too small");
        std::memcpy(out, rgb.data(), rgb.size());
    }
    return 0;
    DK_CATCH
}

int bp_darknet_detect_file(bp_darknet* d, const char* path, float thresh, float nms, bp_bbox* out, int cap) {
    DK_TRY
    if (!path) throw std::runtime_error("null argument");
    const std::vector<uint8_t> file = bp::read_file(path);
    return bp_darknet_detect_image(d, file.data(), file.size(), thresh, nms, out, cap);
    DK_CATCH
}

}  // extern "C"

// ------------------------------------------------------------------ yolo_v2_class.hpp:49-54
static std::unique_ptr<bp_darknet, void (*)(bp_darknet*)> g_detector(nullptr, bp_darknet_destroy);
static std::mutex g_detector_mutex;
static const float kThresh = 0.2f, kNms = 0.4f;   // Detector::detect default thresh (hpp:75), Detector::nms (hpp:63)

static int to_container(int n, const std::vector<bp_bbox>& v, bbox_t_container& container) {
    for (int i = 0; i < n && i < C_SHARP_MAX_OBJECTS; ++i) {
        bbox_t& o = container.candidates[i];
        o.x = v[i].x; o.y = v[i].y; o.w = v[i].w; o.h = v[i].h;
        o.prob = v[i].prob; o.obj_id = v[i].obj_id; o.track_id = v[i].track_id; o.frames_counter = v[i].frames_counter;
    }
    return n;
}

extern "C" int init(const char* configurationFilename, const char* weightsFilename, int gpu) {
    std::lock_guard<std::mutex> lk(g_detector_mutex);
    bp_darknet* d = nullptr;
    if (bp_darknet_create(configurationFilename, weightsFilename, gpu, &d) != 0) return -1;
    g_detector.reset(d);
    return 1;
}

extern "C" int detect_image(const char* filename, bbox_t_container& container) {
    std::lock_guard<std::mutex> lk(g_detector_mutex);
    if (!g_detector) { g_cerr = "init() has not been called"; return -1; }
    std::vector<bp_bbox> v(C_SHARP_MAX_OBJECTS);
    const int n = bp_darknet_detect_file(g_detector.get(), filename, kThresh, kNms, v.data(), (int)v.size());
    return n < 0 ? n : to_container(n, v, container);
}

extern "C" int detect_mat(const uint8_t* data, const size_t data_length, bbox_t_container& container) {
    std::lock_guard<std::mutex> lk(g_detector_mutex);
    if (!g_detector) { g_cerr = "init() has not been called"; return -1; }
    std::vector<bp_bbox> v(C_SHARP_MAX_OBJECTS);
    const int n = bp_darknet_detect_image(g_detector.get(), data, data_length, kThresh, kNms, v.data(), (int)v.size());
    return n < 0 ? n : to_container(n, v, container);
}

extern "C" int dispose() {
    std::lock_guard<std::mutex> lk(g_detector_mutex);
    g_detector.reset();
    return 1;
}

extern "C" int get_device_count() { return bp_device_count(); }

extern "C" int get_device_name(int gpu, char* deviceName) {      // deviceName: caller's buffer, >= 256 bytes in the reference's callers
    return bp_device_name(gpu, deviceName, 256) == 0 ? 1 : 0;
}
