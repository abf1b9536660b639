// extern "C" boundary of libbetapose_hip.so -- see include/betapose_hip.h.
#include "../../include/betapose_hip.h"

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <memory>
#include <sstream>
#include <vector>

#include "engine.h"
#include "frame_io.h"

namespace bp {   // host_post.cpp
int solve_pnp(const double* P, const double* U, int n, const double* K, double* R, double* t);
int solve_pnp_refined(const double* P, const double* U, int n, const double* K, double* R, double* t);
int solve_pnp_ransac(const double* P, const double* U, int n, const double* K, double reproj_err, int max_trials,
                     double confidence, double* R, double* t, unsigned char* inlier_mask);
int pose_nms(const float* bboxes, const float* bbox_scores, const float* preds, const float* scores, int n, int K,
             int* out_pick, float* out_pose, float* out_score, float* out_prop);
}

static thread_local std::string g_err;

#define BP_TRY try {
#define BP_CATCH                                   \
    }                                              \
    catch (const std::exception& e) {              \
        g_err = e.what();                          \
        return -1;                                 \
    }                                              \
    catch (...) {                                  \
        g_err = "unknown error";                   \
        return -1;                                 \
    }

struct bp_yolo {
    std::unique_ptr<bp::YoloNet> net;
    int device;
};
struct bp_kpd {
    std::unique_ptr<bp::KpdNet> net;
    int device;
};

struct bp_pipeline {
    bp_yolo* y;
    bp_kpd* k;
    int H, W, batch;
    float conf;
    int num_classes;
    bp::Arena arena;
    uint8_t* frames = nullptr;
    uint8_t* tmp = nullptr;
    float* results = nullptr;    // [batch][316] = sel[8] | pts[8] | kp[50][6], each written by its producer
    float* hm = nullptr;
    float* fixed_box = nullptr;  // [batch][4] or null
    bool use_fixed = false;
    int *hb = nullptr, *hk = nullptr, *vb = nullptr, *vk = nullptr;
    int ksh = 0, ksv = 0;
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    hipStream_t cap_stream = nullptr;
    unsigned ver_y = 0, ver_k = 0;   // engine plan versions the graph was captured with
    int latency_faults = 0;          // frames re-run because the latency mode's placement check failed (bp_pipeline_latency_faults)
    ~bp_pipeline() {
        if (exec) (void)hipGraphExecDestroy(exec);
        if (graph) (void)hipGraphDestroy(graph);
        if (cap_stream) (void)hipStreamDestroy(cap_stream);
    }
};

static std::string read_text(const char* path) {
    std::ifstream f(path);
    if (!f) throw bp::Error(std::string("cannot open ") + path);
    std::stringstream ss;
    ss << f.rdbuf();
    return ss.str();
}

// Built-in default of newly created engines: bf16x3 = fp32-accurate convolution on the bf16 matrix pipe (exact 3-way
// operand split, passes the whole parity suite at the fp32 bars).  BP_PRECISION=f32|bf16x3|f16|f16r overrides
// it (bp_*_set_precision still wins).
static int default_precision() {
    const char* e = std::getenv("BP_PRECISION");
    if (!e || !*e) return bp::PREC_BF16X3;
    const std::string v(e);
    if (v == "f32") return bp::PREC_F32;
    if (v == "f16") return bp::PREC_F16;
    if (v == "f16r") return bp::PREC_F16_RES;
    if (v == "bf16x3") return bp::PREC_BF16X3;
    throw bp::Error("BP_PRECISION must be f32, bf16x3, f16 or f16r, not '" + v + "'");
}

extern "C" {

const char* bp_last_error(void) { return g_err.c_str(); }
int bp_version(void) { return 100; }

int bp_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}
int bp_device_name(int device, char* out, int cap) {
    BP_TRY
    hipDeviceProp_t p;
    BP_HIP(hipGetDeviceProperties(&p, device));
    std::snprintf(out, cap, "%s (%s)", p.name, p.gcnArchName);
    return 0;
    BP_CATCH
}

// ------------------------------------------------------------------ streams bound to a CU subset
int bp_stream_create_masked(const uint32_t* cu_mask, int words, void** out) {
    BP_TRY
    BP_CHECK(cu_mask && words > 0 && out, "null argument");
    hipStream_t s = nullptr;
    BP_HIP(hipExtStreamCreateWithCUMask(&s, (uint32_t)words, cu_mask));
    *out = (void*)s;
    return 0;
    BP_CATCH
}

int bp_stream_destroy(void* stream) {
    BP_TRY
    BP_HIP(hipStreamDestroy((hipStream_t)stream));
    return 0;
    BP_CATCH
}

int bp_probe_placement(int blocks, int* h_xcc, int* h_hw_id, void* stream) {
    BP_TRY
    BP_CHECK(blocks > 0 && h_xcc, "bad argument");
    int* d = nullptr;
    BP_HIP(hipMalloc(&d, (size_t)blocks * 2 * sizeof(int)));
    bp::launch_probe_placement(d, blocks, (hipStream_t)stream);
    std::vector<int> h((size_t)blocks * 2);
    hipError_t e = hipMemcpyAsync(h.data(), d, h.size() * sizeof(int), hipMemcpyDeviceToHost, (hipStream_t)stream);
    if (e == hipSuccess) e = hipStreamSynchronize((hipStream_t)stream);
    (void)hipFree(d);
    BP_HIP(e);
    for (int i = 0; i < blocks; ++i) {
        h_xcc[i] = h[2 * i];
        if (h_hw_id) h_hw_id[i] = h[2 * i + 1];
    }
    return 0;
    BP_CATCH
}

// ------------------------------------------------------------------ detector
int bp_yolo_create_from_memory(const char* cfg_text, const float* stream, size_t n_floats, int reso, int max_batch,
                               int device, bp_yolo** out) {
    BP_TRY
    BP_CHECK(cfg_text && stream && out, "null argument");
    BP_HIP(hipSetDevice(device));
    std::unique_ptr<bp_yolo> y(new bp_yolo);
    y->device = device;
    y->net.reset(new bp::YoloNet(cfg_text, stream, n_floats, reso, max_batch));
    if (default_precision() != bp::PREC_F32) y->net->set_precision(default_precision());
    *out = y.release();
    return 0;
    BP_CATCH
}

static std::vector<float> read_weights_file(const char* weights_path) {
    std::ifstream f(weights_path, std::ios::binary);
    if (!f) throw bp::Error(std::string("cannot open ") + weights_path);
    f.seekg(0, std::ios::end);
    const size_t bytes = (size_t)f.tellg();
    f.seekg(0);
    BP_CHECK(bytes >= 16, "truncated .weights header");
    int32_t hdr[3];
    f.read(reinterpret_cast<char*>(hdr), 12);
    // parser.c:1161-1174: major*10+minor >= 2 -> 64-bit "seen", else 32-bit
    const size_t off = (hdr[0] * 10 + hdr[1] >= 2) ? 20 : 16;
    BP_CHECK(bytes >= off && (bytes - off) % 4 == 0, "malformed .weights payload");
    std::vector<float> stream((bytes - off) / 4);
    f.seekg(off);
    f.read(reinterpret_cast<char*>(stream.data()), bytes - off);
    return stream;
}

int bp_yolo_create(const char* cfg_path, const char* weights_path, int reso, int max_batch, int device, bp_yolo** out) {
    BP_TRY
    BP_CHECK(cfg_path && weights_path && out, "null argument");
    const std::string cfg = read_text(cfg_path);
    const std::vector<float> stream = read_weights_file(weights_path);
    return bp_yolo_create_from_memory(cfg.c_str(), stream.data(), stream.size(), reso, max_batch, device, out);
    BP_CATCH
}

int bp_yolo_create_darknet(const char* cfg_path, const char* weights_path, int reso, int max_batch, int device,
                           bp_yolo** out) {
    BP_TRY
    BP_CHECK(cfg_path && weights_path && out, "null argument");
    const std::string cfg = read_text(cfg_path);
    const std::vector<float> stream = read_weights_file(weights_path);
    BP_HIP(hipSetDevice(device));
    std::unique_ptr<bp_yolo> y(new bp_yolo);
    y->device = device;
    y->net.reset(new bp::YoloNet(cfg, stream.data(), stream.size(), reso, max_batch, nullptr, /*darknet_bn=*/true));
    if (default_precision() != bp::PREC_F32) y->net->set_precision(default_precision());
    *out = y.release();
    return 0;
    BP_CATCH
}

int bp_yolo_clone(const bp_yolo* y, bp_yolo** out) {
    BP_TRY
    BP_CHECK(y && out, "null argument");
    BP_HIP(hipSetDevice(y->device));
    std::unique_ptr<bp_yolo> c(new bp_yolo);
    c->device = y->device;
    c->net.reset(y->net->clone());
    if (y->net->precision() != bp::PREC_F32) c->net->set_precision(y->net->precision_id());
    c->net->set_fusion(y->net->fusion());
    *out = c.release();
    return 0;
    BP_CATCH
}
void bp_yolo_destroy(bp_yolo* y) { delete y; }
int bp_yolo_rows(const bp_yolo* y) { return y ? y->net->rows() : -1; }
int bp_yolo_attrs(const bp_yolo* y) { return y ? y->net->attrs() : -1; }

int bp_yolo_forward(bp_yolo* y, const float* d_img, int batch, float* d_pred, void* stream) {
    BP_TRY
    BP_CHECK(y && d_img && d_pred, "null argument");
    y->net->forward(d_img, false, batch, d_pred, 0.f, 0, nullptr, (hipStream_t)stream);
    return 0;
    BP_CATCH
}
int bp_yolo_forward_select(bp_yolo* y, const float* d_img, int batch, float conf, int num_classes, float* d_pred,
                           float* d_sel, void* stream) {
    BP_TRY
    BP_CHECK(y && d_img && d_sel, "null argument");
    y->net->forward(d_img, false, batch, d_pred, conf, num_classes, d_sel, (hipStream_t)stream);
    return 0;
    BP_CATCH
}

int bp_yolo_select(const float* d_pred, int batch, int rows, int attrs, float conf, int num_classes, float* d_sel,
                   void* stream) {
    BP_TRY
    BP_CHECK(d_pred && d_sel && batch >= 1 && rows >= 1 && attrs >= 6, "bad argument");
    bp::launch_yolo_select(d_pred, batch, rows, attrs, conf, num_classes, d_sel, (hipStream_t)stream);
    BP_HIP(hipGetLastError());
    return 0;
    BP_CATCH
}

static int tap_info(const bp::Net& n, int i, char* name, int cap, int* C, int* H, int* W) {
    if (i < 0 || i >= n.tap_count()) { g_err = "tap index"; return -1; }
    if (name && cap > 0) std::snprintf(name, cap, "%s", n.tap_name(i));
    n.tap_shape(i, C, H, W);
    return 0;
}
int bp_yolo_tap_count(const bp_yolo* y) { return y ? y->net->tap_count() : -1; }
int bp_yolo_tap_info(const bp_yolo* y, int i, char* name, int cap, int* C, int* H, int* W) {
    if (!y || !C || !H || !W) { g_err = "null argument"; return -1; }
    return tap_info(*y->net, i, name, cap, C, H, W);
}
int bp_yolo_tap_copy(bp_yolo* y, int i, int batch, float* d_out, void* stream) {
    BP_TRY
    BP_CHECK(y && d_out, "null argument");
    y->net->tap_copy(i, batch, d_out, (hipStream_t)stream);
    return 0;
    BP_CATCH
}

// ------------------------------------------------------------------ key-point detector
int bp_kpd_create(const float* stream, size_t n_floats, int n_classes, int max_batch, int device, bp_kpd** out) {
    BP_TRY
    BP_CHECK(stream && out, "null argument");
    BP_HIP(hipSetDevice(device));
    std::unique_ptr<bp_kpd> k(new bp_kpd);
    k->device = device;
    k->net.reset(new bp::KpdNet(stream, n_floats, n_classes, max_batch));
    if (default_precision() != bp::PREC_F32) k->net->set_precision(default_precision());
    *out = k.release();
    return 0;
    BP_CATCH
}
int bp_kpd_clone(const bp_kpd* k, bp_kpd** out) {
    BP_TRY
    BP_CHECK(k && out, "null argument");
    BP_HIP(hipSetDevice(k->device));
    std::unique_ptr<bp_kpd> c(new bp_kpd);
    c->device = k->device;
    c->net.reset(k->net->clone());
    if (k->net->precision() != bp::PREC_F32) c->net->set_precision(k->net->precision_id());
    c->net->set_fusion(k->net->fusion());
    *out = c.release();
    return 0;
    BP_CATCH
}
void bp_kpd_destroy(bp_kpd* k) { delete k; }
int bp_kpd_forward(bp_kpd* k, const float* d_inps, int batch, float* d_hm, void* stream) {
    BP_TRY
    BP_CHECK(k && d_inps && d_hm, "null argument");
    k->net->forward(d_inps, false, batch, d_hm, nullptr, (hipStream_t)stream);
    return 0;
    BP_CATCH
}
int bp_kpd_forward_argmax(bp_kpd* k, const float* d_inps, int batch, float* d_hm, float* d_kp, void* stream) {
    BP_TRY
    BP_CHECK(k && d_inps && d_kp, "null argument");
    k->net->forward(d_inps, false, batch, d_hm, d_kp, (hipStream_t)stream);
    return 0;
    BP_CATCH
}
int bp_kpd_tap_count(const bp_kpd* k) { return k ? k->net->tap_count() : -1; }
int bp_kpd_tap_info(const bp_kpd* k, int i, char* name, int cap, int* C, int* H, int* W) {
    if (!k || !C || !H || !W) { g_err = "null argument"; return -1; }
    return tap_info(*k->net, i, name, cap, C, H, W);
}
int bp_kpd_tap_copy(bp_kpd* k, int i, int batch, float* d_out, void* stream) {
    BP_TRY
    BP_CHECK(k && d_out, "null argument");
    k->net->tap_copy(i, batch, d_out, (hipStream_t)stream);
    return 0;
    BP_CATCH
}

int bp_yolo_set_policy(bp_yolo* y, int t, int mc, int ms, int ft) {
    BP_TRY
    BP_CHECK(y, "null argument");
    BP_CHECK(t >= 1 && mc >= 1 && ms >= 1 && ft >= -1 && ft <= bp::TILE_LAST, "policy values out of range");
    y->net->set_splitk_policy(t, mc);
    y->net->set_max_splits(ms);
    y->net->set_force_tile(ft);
    return 0;
    BP_CATCH
}
int bp_yolo_set_precision(bp_yolo* y, int prec) {
    BP_TRY
    BP_CHECK(y, "null argument");
    BP_HIP(hipSetDevice(y->device));
    y->net->set_precision(prec);
    return 0;
    BP_CATCH
}
int bp_kpd_set_precision(bp_kpd* k, int prec) {
    BP_TRY
    BP_CHECK(k, "null argument");
    BP_HIP(hipSetDevice(k->device));
    k->net->set_precision(prec);
    return 0;
    BP_CATCH
}
int bp_kpd_set_policy(bp_kpd* k, int t, int mc, int ms, int ft) {
    BP_TRY
    BP_CHECK(k, "null argument");
    BP_CHECK(t >= 1 && mc >= 1 && ms >= 1 && ft >= -1 && ft <= bp::TILE_LAST, "policy values out of range");
    k->net->set_splitk_policy(t, mc);
    k->net->set_max_splits(ms);
    k->net->set_force_tile(ft);
    return 0;
    BP_CATCH
}
static int op_stats(const bp::Net& n, double* flops, double* bytes, int cap) {
    const auto& ops = n.ops();
    for (int i = 0; i < (int)ops.size() && i < cap; ++i) {
        if (flops) flops[i] = ops[i].flops;
        if (bytes) bytes[i] = ops[i].bytes;
    }
    return (int)ops.size();
}
int bp_yolo_op_stats(const bp_yolo* y, double* flops, double* bytes, int cap) { return op_stats(*y->net, flops, bytes, cap); }
int bp_kpd_op_stats(const bp_kpd* k, double* flops, double* bytes, int cap) { return op_stats(*k->net, flops, bytes, cap); }
int bp_yolo_profile(bp_yolo* y, int batch, int iters, float* ms, int* info, int cap, void* stream) {
    BP_TRY
    return y->net->profile(batch, iters, ms, info, cap, (hipStream_t)stream);
    BP_CATCH
}
int bp_kpd_profile(bp_kpd* k, int batch, int iters, float* ms, int* info, int cap, void* stream) {
    BP_TRY
    return k->net->profile(batch, iters, ms, info, cap, (hipStream_t)stream);
    BP_CATCH
}
int bp_calibrate_ticks(long long ticks, float* ms, void* stream) {
    BP_TRY
    BP_CHECK(ms && ticks > 0, "arguments");
    hipStream_t s = (hipStream_t)stream;
    hipEvent_t e0, e1;
    BP_HIP(hipEventCreate(&e0));
    BP_HIP(hipEventCreate(&e1));
    bp::launch_spin_ticks(1000, s);            // warm
    BP_HIP(hipEventRecord(e0, s));
    bp::launch_spin_ticks(ticks, s);
    BP_HIP(hipEventRecord(e1, s));
    BP_HIP(hipEventSynchronize(e1));
    BP_HIP(hipEventElapsedTime(ms, e0, e1));
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    return 0;
    BP_CATCH
}
int bp_yolo_set_prefetch(bp_yolo* y, int on) { y->net->set_prefetch(on != 0); return 0; }
int bp_kpd_set_prefetch(bp_kpd* k, int on) { k->net->set_prefetch(on != 0); return 0; }
// conv -> conv fusion of residual / bottleneck blocks (conv_fused.hip): on by default; *launches = groups that run as ONE launch at `batch`
int bp_yolo_set_fusion(bp_yolo* y, int on) { y->net->set_fusion(on != 0); return 0; }
int bp_kpd_set_fusion(bp_kpd* k, int on) { k->net->set_fusion(on != 0); return 0; }
int bp_yolo_fused_launches(bp_yolo* y, int batch, int* launches) {
    BP_TRY
    BP_CHECK(y && launches && batch >= 1 && batch <= y->net->max_batch(), "arguments");
    BP_HIP(hipSetDevice(y->device));
    *launches = y->net->fused_launches(batch);
    return 0;
    BP_CATCH
}
int bp_kpd_fused_launches(bp_kpd* k, int batch, int* launches) {
    BP_TRY
    BP_CHECK(k && launches && batch >= 1 && batch <= k->net->max_batch(), "arguments");
    BP_HIP(hipSetDevice(k->device));
    *launches = k->net->fused_launches(batch);
    return 0;
    BP_CATCH
}
int bp_yolo_xcd_errors(bp_yolo* y, int* count, void* stream) {
    BP_TRY
    BP_CHECK(y && count, "null argument");
    BP_HIP(hipSetDevice(y->device));
    *count = y->net->take_xcd_errors((hipStream_t)stream);
    return 0;
    BP_CATCH
}
int bp_kpd_xcd_errors(bp_kpd* k, int* count, void* stream) {
    BP_TRY
    BP_CHECK(k && count, "null argument");
    BP_HIP(hipSetDevice(k->device));
    *count = k->net->take_xcd_errors((hipStream_t)stream);
    return 0;
    BP_CATCH
}
int bp_yolo_set_stamps(bp_yolo* y, unsigned long long* d_buf, int slots) { y->net->set_stamps(d_buf, slots); return 0; }
int bp_kpd_set_stamps(bp_kpd* k, unsigned long long* d_buf, int slots) { k->net->set_stamps(d_buf, slots); return 0; }
static int op_name(const bp::Net& net, int i, char* out, int cap) {
    if (i < 0 || i >= (int)net.ops().size() || !out || cap <= 0) return -1;
    const bp::Op& op = net.ops()[i];
    if (op.type == bp::OP_CONV)   // "<name> k<ksize> <OH>x<OW> <Cin>-><Cout> s<stride>"
        std::snprintf(out, (size_t)cap, "%s k%d %dx%d %d->%d s%d", net.op_name(i), op.conv.ksize, op.conv.OH, op.conv.OW, op.conv.Cin,
                      op.conv.Cout, op.conv.stride);
    else
        std::snprintf(out, (size_t)cap, "%s", net.op_name(i));
    return op.type == bp::OP_CONV ? 1 : 0;
}
int bp_yolo_op_name(const bp_yolo* y, int i, char* out, int cap) { return op_name(*y->net, i, out, cap); }
int bp_kpd_op_name(const bp_kpd* k, int i, char* out, int cap) { return op_name(*k->net, i, out, cap); }
size_t bp_yolo_device_bytes(const bp_yolo* y) { return y->net->device_bytes(); }
size_t bp_kpd_device_bytes(const bp_kpd* k) { return k->net->device_bytes(); }

// ------------------------------------------------------------------ stand-alone stages
int bp_crop(const uint8_t* d_frames, int batch, int H, int W, const float* d_sel, int reso, const float* d_boxes,
            float* d_out_nchw, float* d_out_nhwc, float* d_pts, int oh, int ow, void* stream) {
    BP_TRY
    BP_CHECK(d_frames && (d_sel || d_boxes) && (d_out_nchw || d_out_nhwc), "null argument");
    bp::launch_crop(d_frames, batch, H, W, d_sel, reso, d_boxes, d_out_nhwc, d_out_nchw, d_pts, oh, ow, (hipStream_t)stream);
    BP_HIP(hipGetLastError());
    return 0;
    BP_CATCH
}

int bp_resize_bicubic(const uint8_t* d_in, int batch, int H, int W, int oh, int ow, int swap_rb, uint8_t* d_out_u8,
                      float* d_out_nhwc, void* stream) {
    BP_TRY
    BP_CHECK(d_in && (d_out_u8 || d_out_nhwc), "null argument");
    hipStream_t s = (hipStream_t)stream;
    const bp::ResizePlan ph = bp::make_bicubic_plan(W, ow), pv = bp::make_bicubic_plan(H, oh);
    bp::Arena a;
    int* hb = (int*)a.alloc_bytes(ph.bounds.size() * 4);
    int* hk = (int*)a.alloc_bytes(ph.coeffs.size() * 4);
    int* vb = (int*)a.alloc_bytes(pv.bounds.size() * 4);
    int* vk = (int*)a.alloc_bytes(pv.coeffs.size() * 4);
    uint8_t* tmp = (uint8_t*)a.alloc_bytes((size_t)batch * H * ow * 3);
    BP_HIP(hipMemcpyAsync(hb, ph.bounds.data(), ph.bounds.size() * 4, hipMemcpyHostToDevice, s));
    BP_HIP(hipMemcpyAsync(hk, ph.coeffs.data(), ph.coeffs.size() * 4, hipMemcpyHostToDevice, s));
    BP_HIP(hipMemcpyAsync(vb, pv.bounds.data(), pv.bounds.size() * 4, hipMemcpyHostToDevice, s));
    BP_HIP(hipMemcpyAsync(vk, pv.coeffs.data(), pv.coeffs.size() * 4, hipMemcpyHostToDevice, s));
    bp::ResizeTables t{hb, hk, ph.ksize, vb, vk, pv.ksize};
    bp::launch_resize_bicubic(d_in, batch, H, W, tmp, d_out_nhwc, d_out_u8, oh, ow, t, swap_rb, s);
    BP_HIP(hipGetLastError());
    BP_HIP(hipStreamSynchronize(s));   // tables live in a local arena
    return 0;
    BP_CATCH
}

int bp_heatmap_argmax(const float* d_hm, int batch, int C, int H, int W, float* d_kp, void* stream) {
    BP_TRY
    BP_CHECK(d_hm && d_kp && batch > 0 && C > 0 && H > 0 && W > 0, "bad argument");
    bp::launch_heatmap_argmax(d_hm, batch, C, H, W, d_kp, (hipStream_t)stream);
    BP_HIP(hipGetLastError());
    return 0;
    BP_CATCH
}

int bp_conv2d(const float* d_in, int N, int H, int W, int Cin, const float* h_w, const float* h_bias, int Cout, int k,
              int stride, int pad, int act, int store_mode, const float* d_res, int res_after_act, int tile, int splits,
              float* d_out, int iters, float* ms_per_iter, void* stream) {
    return bp_conv2d_planes(d_in, N, H, W, Cin, h_w, h_bias, Cout, k, stride, pad, act, store_mode, d_res, res_after_act, tile,
                            splits, d_out, nullptr, iters, ms_per_iter, stream);
}

int bp_conv2d_planes(const float* d_in, int N, int H, int W, int Cin, const float* h_w, const float* h_bias, int Cout, int k,
                     int stride, int pad, int act, int store_mode, const float* d_res, int res_after_act, int tile, int splits,
                     float* d_out, unsigned short* d_out_planes, int iters, float* ms_per_iter, void* stream) {
    BP_TRY
    BP_CHECK(d_in && h_w && d_out, "null argument");
    hipStream_t s = (hipStream_t)stream;
    struct OneConv : bp::Net {
        OneConv() : Net(1) {}
        using Net::add_conv;
        using Net::finalize;
        using Net::ops_;
        using Net::partial_;
        using Net::partial_floats_;
        using Net::arena_;
    } net;
    const int OH = (H + 2 * pad - k) / stride + 1, OW = (W + 2 * pad - k) / stride + 1;
    bp::Tensor in; in.p = const_cast<float*>(d_in); in.H = H; in.W = W; in.C = Cin; in.ld = Cin;
    bp::Tensor out; out.p = d_out; out.H = OH; out.W = OW; out.C = Cout;
    out.ld = store_mode == bp::ST_PIXSHUF ? Cout / 4 : Cout;
    bp::Tensor res; res.p = const_cast<float*>(d_res); res.ld = Cout; res.C = Cout; res.H = OH; res.W = OW;
    bp::ConvWeights cw; cw.w = h_w; cw.bias = h_bias;
    net.add_conv("conv", in, out, cw, Cout, k, stride, pad, act, store_mode, d_res ? &res : nullptr, nullptr,
                 res_after_act, 1e-5f, OH, OW);
    int t = tile;
    int prec = bp::PREC_F32;
    if (t >= 256) {   // + 256: fp16-MFMA operands, + 512: bf16x3 split operands
        prec = t >= 512 ? bp::PREC_BF16X3 : bp::PREC_F16;
        t -= t >= 512 ? 512 : 256;
        BP_CHECK(Cin % 32 == 0, "layer is not eligible for the 16-bit MFMA paths (needs Cin % 32 == 0)");
        net.set_precision(prec);
        net.ops_[0].conv.mfma_mode = prec;
#ifndef BP_EXPERIMENTAL
        BP_CHECK(bp::conv_tile_is_pl(t) || ((t == bp::TILE_64x64_BD || t == bp::TILE_BD_K2 || bp::conv_tile_is_halo(t)) && prec == bp::PREC_BF16X3),
                 "this kernel id exists only in the experimental library (python -m betapose_amd.build --experimental, BP_LIB)");
#endif
    } else {
        if (t < 0 && bp::conv_stem3_eligible(net.ops_[0].conv)) t = bp::TILE_STEM3;      // as the engine plans it
        BP_CHECK(t <= bp::TILE_128x64 || (t == bp::TILE_STEM3 && bp::conv_stem3_eligible(net.ops_[0].conv)) || (t == bp::TILE_STEM7 && bp::conv_stem7_eligible(net.ops_[0].conv)),
                 "this tile needs a 16-bit precision mode (tile + 256 / + 512), or the layer is not a 3x3 / stride-1 / 4-channel-packed stem");
    }
    bp::ConvParams p = net.ops_[0].conv;
    p.N = N; p.M = N * OH * OW;
    if (t < 0) t = bp::TILE_64x64;
    const int np = prec == bp::PREC_F16 ? 1 : 3;
    if (bp::conv_tile_is_pl(t)) {
        // the caller's activations are fp32: their operand planes are made here, as the layer's producer would have
        const long long in_elems = (long long)N * H * W * Cin;
        BP_CHECK(Cin % 32 == 0 && k * k <= 32, "layer is not eligible for the operand-plane kernels (needs Cin % 32 == 0, k*k <= 32)");
        unsigned short* d_in16 = (unsigned short*)net.arena_.alloc_bytes((size_t)np * in_elems * 2);
        bp::launch_f32_to_planes(d_in, Cin, (long long)N * H * W, Cin, d_in16, in_elems, np, s);
        p.in16 = d_in16; p.in16_plane = in_elems;
        auto& packed = np == 1 ? net.weight_store()->wpl1 : net.weight_store()->wpl3;
        auto it = packed.find(p.w);
        if (it == packed.end()) {
            unsigned short* d = (unsigned short*)net.weight_store()->arena.alloc_bytes((size_t)np * p.CoutPad * p.Kpad * 2);
            bp::launch_pack_wpl(p.w, d, p.CoutPad, p.Kpad, p.Cin, p.ksize, np, s);
            it = packed.emplace(p.w, d).first;
        }
        p.wpl = it->second;
        if (t == bp::TILE_PL64BD) {   // the filters-direct plane tile reads stage-packed fragments in the plane kernels' K order
            auto& staged = net.weight_store()->wbd3;
            auto ib = staged.find(p.w);
            if (ib == staged.end()) {
                unsigned short* d = (unsigned short*)net.weight_store()->arena.alloc_bytes((size_t)3 * p.CoutPad * p.Kpad * 2);
                bp::launch_f32_to_bf16x3_staged(p.w, d, p.CoutPad, p.Kpad, s, p.Cin);
                ib = staged.emplace(p.w, d).first;
            }
            p.wbd = ib->second;
        }
    }
    if (const char* e = std::getenv("BP_PL_ABL")) p.abl = std::atoi(e);   // (read by experimental builds only)
    if (d_out_planes) {   // the epilogue's operand planes of the output (any kernel): [np][the output tensor's elements]
        BP_CHECK(prec != bp::PREC_F32, "output planes need a 16-bit precision mode");
        long long out_elems = (long long)N * OH * OW * Cout;
        if (store_mode == bp::ST_UP2) out_elems *= 4;
        p.out16 = d_out_planes; p.out16_plane = out_elems; p.out_np = np;
    }
    // BP_CONV_F16R=1 (bench tools): the launch as the engine's 'f16r' plan makes it -- the skip connection read from an fp16 plane of the
    // residual tensor (ConvParams::res16) and, when output planes are asked for, the fp32 store dropped (ConvParams::skip_f32)
    if (prec == bp::PREC_F16 && std::getenv("BP_CONV_F16R")) {
        if (p.res) {
            const long long n_res = (long long)N * OH * OW;
            unsigned short* r16 = (unsigned short*)net.arena_.alloc_bytes((size_t)n_res * p.res_ld * 2);
            bp::launch_f32_to_planes(p.res, p.res_ld, n_res, Cout, r16, n_res * p.res_ld, 1, s);
            p.res16 = r16;
        }
        if (p.out16) p.skip_f32 = 1;
    }
    int sp = splits;
    if (sp <= 0) {
        const long long blocks = bp::conv_tiles(p, t);
        sp = 1;
        while (blocks * sp < 512 && p.nchunks / (sp + 1) >= 4 && sp < 64) ++sp;
    }
    if (t == bp::TILE_S1 || t == bp::TILE_P3) sp = 1;     // (a persistent grid: no K slices)
    int per = 0;
    bp::conv_split_plan(p, t, sp, &sp, &per);
    p.splits = sp; p.chunks_per_split = per;
    if (sp > 1) {
        const int tiles = bp::conv_tiles(p, t);
        p.partial = net.arena_.alloc((size_t)sp * tiles * bp::conv_tile_bm(t) * bp::conv_tile_bn(t));
        p.tickets = (int*)net.arena_.alloc_bytes((size_t)(2 + 64) * tiles * sizeof(int));
        BP_HIP(hipMemset(p.tickets, 0, (size_t)(2 + 64) * tiles * sizeof(int)));
    }
    if (sp == 1 && !std::getenv("BP_CONV_SELF_PREFETCH")) {   // as the engine launches it: hybrid grid when the last round is badly filled
        const size_t cap = (size_t)8 * 256 * bp::conv_tile_bm(t) * bp::conv_tile_bn(t);     // <= 8 slices of <= 256 tail tiles
        int full = 0, hs = 0, hcps = 0;
        if (bp::conv_hybrid_plan(p, t, cap, &full, &hs, &hcps)) {
            const int tail = bp::conv_tiles(p, t) - full;
            p.partial = net.arena_.alloc((size_t)hs * tail * bp::conv_tile_bm(t) * bp::conv_tile_bn(t));
            p.tickets = (int*)net.arena_.alloc_bytes((size_t)tail * sizeof(int));
            BP_HIP(hipMemset(p.tickets, 0, (size_t)tail * sizeof(int)));
            p.hy_full = full; p.hy_splits = hs; p.hy_cps = hcps;
        }
    }
    if (bp::conv_home_layout(t, sp)) {   // as the engine launches it: all K slices of a tile on one XCD, hand-off through that XCD's L2
        const int tiles = bp::conv_tiles(p, t);
        p.xcd_home = 1;
        p.tickets_local = p.tickets + tiles;
        p.xcc_of = p.tickets + 2 * tiles;
    }
    if (std::getenv("BP_CONV_SELF_PREFETCH")) bp::conv_prefetch_of(p, p, t, sp, per);   // (tests: the launch carries prefetch blocks, for its own filters)
    bp::launch_conv(p, t, s);
    BP_HIP(hipStreamSynchronize(s));
    if (const char* e = std::getenv("BP_CONV_STAMPS")) {   // debug: per-block s_memtime marks of one extra launch
        const int nb = bp::conv_tiles(p, t) * p.splits;
        unsigned long long* d = (unsigned long long*)net.arena_.alloc_bytes((size_t)nb * 8 * 8);
        BP_HIP(hipMemset(d, 0, (size_t)nb * 64));
        bp::ConvParams q = p; q.stamps = d;
        bp::launch_conv(q, t, s);
        BP_HIP(hipStreamSynchronize(s));
        std::vector<unsigned long long> h((size_t)nb * 8);
        BP_HIP(hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost));
        unsigned long long t0 = ~0ull;
        for (int b = 0; b < nb; ++b)
            if (h[(size_t)b * 8]) t0 = std::min(t0, h[(size_t)b * 8]);
        double acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, last = 0;
        int cnt[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int b = 0; b < nb; ++b)
            for (int k = 0; k < 8; ++k) {
                const unsigned long long v = h[(size_t)b * 8 + k];
                if (!v) continue;
                acc[k] += (double)(v - h[(size_t)b * 8]); ++cnt[k];      // relative to the block's own entry
                last = std::max(last, (double)(v - t0));
            }
        auto mean = [&](int k) { return cnt[k] ? acc[k] / cnt[k] : 0.0; };
        if (std::getenv("BP_W64_ABLATE") && (std::atoi(std::getenv("BP_W64_ABLATE")) & 64)) {
            double sum[4] = {0, 0, 0, 0};
            for (int b = 0; b < nb; ++b) for (int k = 0; k < 4; ++k) sum[k] += (double)h[(size_t)b * 8 + 4 + k];
            const double stages = 2.0 * p.chunks_per_split * nb;
            std::fprintf(stderr, "[stage timing] cycles per stage: issue slots 0..UPW %.0f | slots ..SYNC %.0f | wait+barrier %.0f | SYNC..end %.0f\n",
                         sum[0] / stages, sum[1] / stages, sum[2] / stages, sum[3] / stages);
        }
        std::fprintf(stderr, "[stamps] blocks=%d  mean 10-ns ticks (s_memrealtime, 100 MHz) since the block's entry: index math done %.0f | chunk 0 in LDS %.0f | "
                     "K loop done %.0f | in-block sums (conv_kg) or cycles parked at the stage waits (conv_pl) %.0f | slab parked + ticket %.0f (%d blocks) | slices combined %.0f (%d) | stores done %.0f (%d) | "
                     "last mark of the grid %.0f after the first entry\n", nb, mean(1), mean(2), mean(3), mean(7), mean(5), cnt[5], mean(6),
                     cnt[6], mean(4), cnt[4], last);
    }
    if (iters > 0 && ms_per_iter) {
        hipEvent_t e0, e1;
        BP_HIP(hipEventCreate(&e0));
        BP_HIP(hipEventCreate(&e1));
        BP_HIP(hipEventRecord(e0, s));
        for (int i = 0; i < iters; ++i) bp::launch_conv(p, t, s);
        BP_HIP(hipEventRecord(e1, s));
        BP_HIP(hipEventSynchronize(e1));
        float ms = 0;
        BP_HIP(hipEventElapsedTime(&ms, e0, e1));
        *ms_per_iter = ms / iters;
        (void)hipEventDestroy(e0);
        (void)hipEventDestroy(e1);
    }
    return 0;
    BP_CATCH
}

// ------------------------------------------------------------------ fused per-frame pipeline
static void pipeline_enqueue(bp_pipeline* p, hipStream_t s) {
    bp::YoloNet& yn = *p->y->net;
    bp::KpdNet& kn = *p->k->net;
    const int reso = yn.reso();
    bp::ResizeTables t{p->hb, p->hk, p->ksh, p->vb, p->vk, p->ksv};
    // a1: Pillow-exact bicubic stretch to reso x reso, BGR -> RGB, /255, straight into the detector's NHWC input
#ifdef BP_EXPERIMENTAL   // timing experiment only (WRONG results): BP_ABLATE_RESIZE=1 leaves both resize launches out of
    // the frame (the detector sees whatever its input buffer holds) -- the upper bound of what fusing them into the stem could buy (round-4 verdict item 6; tools/ab_resize.sh)
    static const bool no_resize = std::getenv("BP_ABLATE_RESIZE") != nullptr;
    if (!no_resize) bp::launch_resize_bicubic(p->frames, p->batch, p->H, p->W, p->tmp, yn.input_nhwc(), nullptr, reso, reso, t, 1, s);
#else
    bp::launch_resize_bicubic(p->frames, p->batch, p->H, p->W, p->tmp, yn.input_nhwc(), nullptr, reso, reso, t, 1, s);
#endif
    // a3-a5: detector + decode + arg-max objectness
    // every stage writes its part of the frame's result row directly (sel[8] | pts[8] | kp[50][6]): no gather launch
    const int R = BP_RESULT_FLOATS;
    yn.forward(yn.input_nhwc(), true, p->batch, nullptr, p->conf, p->num_classes, p->results, s, R);
    // a6-a7: box rescale + crop window + bilinear crop into the KPD's NHWC input
    bp::launch_crop(p->frames, p->batch, p->H, p->W, p->use_fixed ? nullptr : p->results, reso,
                    p->use_fixed ? p->fixed_box : nullptr, kn.input_nhwc(), nullptr, p->results + 8, kn.in_h(), kn.in_w(), s, R, R);
    // a8-a9: KPD + heat-map arg-max
    kn.forward(kn.input_nhwc(), true, p->batch, p->hm, p->results + 16, s, R);
    BP_HIP(hipGetLastError());
}

int bp_pipeline_create(bp_yolo* y, bp_kpd* k, int frame_h, int frame_w, int batch, float conf, int num_classes,
                       uint8_t* d_frames, float* d_results, float* d_hm, bp_pipeline** out) {
    BP_TRY
    BP_CHECK(y && k && out, "null argument");
    BP_CHECK(batch >= 1 && batch <= y->net->max_batch() && batch <= k->net->max_batch(), "pipeline batch > engine max_batch");
    BP_CHECK(k->net->out_c() == 50, "pipeline expects 50 key points");
    BP_HIP(hipSetDevice(y->device));
    std::unique_ptr<bp_pipeline> p(new bp_pipeline);
    p->y = y; p->k = k; p->H = frame_h; p->W = frame_w; p->batch = batch; p->conf = conf; p->num_classes = num_classes;
    const int reso = y->net->reso();
    p->frames = d_frames ? d_frames : (uint8_t*)p->arena.alloc_bytes((size_t)batch * frame_h * frame_w * 3);
    p->tmp = (uint8_t*)p->arena.alloc_bytes((size_t)batch * frame_h * reso * 3);
    p->results = d_results ? d_results : p->arena.alloc((size_t)batch * BP_RESULT_FLOATS);
    p->hm = d_hm ? d_hm : p->arena.alloc((size_t)batch * 50 * k->net->out_h() * k->net->out_w());
    p->fixed_box = p->arena.alloc((size_t)batch * 4);
    const bp::ResizePlan ph = bp::make_bicubic_plan(frame_w, reso), pv = bp::make_bicubic_plan(frame_h, reso);
    p->ksh = ph.ksize; p->ksv = pv.ksize;
    p->hb = (int*)p->arena.alloc_bytes(ph.bounds.size() * 4);
    p->hk = (int*)p->arena.alloc_bytes(ph.coeffs.size() * 4);
    p->vb = (int*)p->arena.alloc_bytes(pv.bounds.size() * 4);
    p->vk = (int*)p->arena.alloc_bytes(pv.coeffs.size() * 4);
    BP_HIP(hipMemcpy(p->hb, ph.bounds.data(), ph.bounds.size() * 4, hipMemcpyHostToDevice));
    BP_HIP(hipMemcpy(p->hk, ph.coeffs.data(), ph.coeffs.size() * 4, hipMemcpyHostToDevice));
    BP_HIP(hipMemcpy(p->vb, pv.bounds.data(), pv.bounds.size() * 4, hipMemcpyHostToDevice));
    BP_HIP(hipMemcpy(p->vk, pv.coeffs.data(), pv.coeffs.size() * 4, hipMemcpyHostToDevice));
    BP_HIP(hipMemset(p->results, 0, (size_t)batch * BP_RESULT_FLOATS * sizeof(float)));
    *out = p.release();
    return 0;
    BP_CATCH
}
void bp_pipeline_destroy(bp_pipeline* p) { delete p; }
uint8_t* bp_pipeline_frames(bp_pipeline* p) { return p ? p->frames : nullptr; }
float* bp_pipeline_results(bp_pipeline* p) { return p ? p->results : nullptr; }
float* bp_pipeline_heatmaps(bp_pipeline* p) { return p ? p->hm : nullptr; }
int bp_pipeline_kernel_count(bp_pipeline* p) {
    if (!p || !p->graph) return -1;
    size_t n = 0;
    if (hipGraphGetNodes(p->graph, nullptr, &n) != hipSuccess) return -1;
    return (int)n;
}

static void drop_graph(bp_pipeline* p) {
    if (p->exec) { (void)hipGraphExecDestroy(p->exec); p->exec = nullptr; }
    if (p->graph) { (void)hipGraphDestroy(p->graph); p->graph = nullptr; }
}

int bp_pipeline_set_fixed_box(bp_pipeline* p, const float* box) {
    BP_TRY
    drop_graph(p);
    p->use_fixed = box != nullptr;
    if (box) {
        std::vector<float> h((size_t)p->batch * 4);
        for (int b = 0; b < p->batch; ++b) std::memcpy(&h[b * 4], box, 4 * sizeof(float));
        BP_HIP(hipMemcpy(p->fixed_box, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice));
    }
    return 0;
    BP_CATCH
}

// (re)build the frame's hipGraph when the launch plan changed since the capture -- records the launches, executes nothing
static void pipeline_capture(bp_pipeline* p) {
    if (p->exec && (p->ver_y != p->y->net->plan_version() || p->ver_k != p->k->net->plan_version())) {
        // launch policy / precision changed since the capture: the recorded kernels are stale
        (void)hipGraphExecDestroy(p->exec);
        (void)hipGraphDestroy(p->graph);
        p->exec = nullptr;
        p->graph = nullptr;
    }
    if (!p->exec) {
        p->ver_y = p->y->net->plan_version();
        p->ver_k = p->k->net->plan_version();
        if (!p->cap_stream) BP_HIP(hipStreamCreateWithFlags(&p->cap_stream, hipStreamNonBlocking));
        BP_HIP(hipStreamBeginCapture(p->cap_stream, hipStreamCaptureModeThreadLocal));
        try {
            pipeline_enqueue(p, p->cap_stream);
        } catch (...) {
            hipGraph_t g = nullptr;
            (void)hipStreamEndCapture(p->cap_stream, &g);
            if (g) (void)hipGraphDestroy(g);
            throw;
        }
        BP_HIP(hipStreamEndCapture(p->cap_stream, &p->graph));
        BP_HIP(hipGraphInstantiate(&p->exec, p->graph, nullptr, nullptr, 0));
    }
}

int bp_pipeline_prepare(bp_pipeline* p) {
    BP_TRY
    BP_CHECK(p, "null argument");
    pipeline_capture(p);
    return 0;
    BP_CATCH
}

static void pipeline_launch(bp_pipeline* p, int use_graph, hipStream_t s) {
    if (!use_graph) {
        pipeline_enqueue(p, s);
        return;
    }
    pipeline_capture(p);
    BP_HIP(hipGraphLaunch(p->exec, s));
}

int bp_pipeline_run(bp_pipeline* p, int use_graph, void* stream) {
    BP_TRY
    BP_CHECK(p, "null argument");
    hipStream_t s = (hipStream_t)stream;
    pipeline_launch(p, use_graph, s);
    // Lone-frame latency mode (bp_*_set_prefetch): a launch that found a K slice on the wrong XCD raised the engine's error word and
    // left its tile unstored, so the frame's record is void.  The mode is one-frame-at-a-time by definition: wait for the frame here,
    // read the words, and on a fault clear them, switch the mode off for both engines and run the SAME frame again on the ordinary
    // hand-off -- whoever drives the pipeline (FramePipeline.run, StreamedRunner, a C caller) gets a valid record or an error.
    if (p->y->net->prefetch() || p->k->net->prefetch()) {
        BP_HIP(hipSetDevice(p->y->device));
        const int bad = p->y->net->take_xcd_errors(s) + p->k->net->take_xcd_errors(s);
        if (bad) {
            ++p->latency_faults;
            p->y->net->set_prefetch(false);
            p->k->net->set_prefetch(false);
            pipeline_launch(p, use_graph, s);
        }
    }
    return 0;
    BP_CATCH
}
int bp_pipeline_latency_faults(const bp_pipeline* p) { return p ? p->latency_faults : -1; }

// ------------------------------------------------------------------ xcd mode (mega.inc): prototype entry point, experimental library only
#ifdef BP_EXPERIMENTAL
// The convolution launch lists of up to eight detector engines (clones: own activations, shared filters), one per XCD, in ONE
// persistent launch; `iters` launches timed with events.  The engines' inputs must already be in place (a forward pass of the
// ordinary path leaves them there); afterwards every engine's activations hold what its own launches would have produced.
int bp_mega_yolo_convs_stamped(bp_yolo** ys, int n, int blocks_per_xcd, int iters, float* ms_per_launch, unsigned* err_word,
                               float* op_us, int* op_info, int cap, void* stream);
int bp_mega_yolo_convs(bp_yolo** ys, int n, int blocks_per_xcd, int iters, float* ms_per_launch, unsigned* err_word, void* stream) {
    return bp_mega_yolo_convs_stamped(ys, n, blocks_per_xcd, iters, ms_per_launch, err_word, nullptr, nullptr, 0, stream);
}
// ... with per-op marks of XCD 0's launch list: op_us[i] = duration of op i incl. its barrier, op_info[3 i] = {type, items, K slices}
int bp_mega_yolo_convs_stamped(bp_yolo** ys, int n, int blocks_per_xcd, int iters, float* ms_per_launch, unsigned* err_word,
                               float* op_us, int* op_info, int cap, void* stream) {
    BP_TRY
    BP_CHECK(ys && n >= 1 && n <= 8 && blocks_per_xcd >= 1 && blocks_per_xcd <= 256 && iters >= 1, "bad argument");
    hipStream_t s = (hipStream_t)stream;
    BP_HIP(hipSetDevice(ys[0]->device));
    bp::MegaArgs a{};
    std::vector<void*> dev;
    size_t lds = 0;
    for (int k = 0; k < n; ++k) {
        std::vector<bp::MegaOp> ops;
        ys[k]->net->emit_conv_ops(1, ops);
        if (const char* e = std::getenv("BP_MEGA_EXP")) {          // prototype experiments: 1 = barriers only (no work items), 2 = without the RGB stem
            const int m = std::atoi(e);
            for (bp::MegaOp& o : ops) {
                if (m == 1) o.items = 0;
                if (m == 2 && o.type == bp::MO_STEM3) o.items = 0;
            }
        }
        for (const bp::MegaOp& o : ops) lds = std::max(lds, bp::mega_lds_bytes(o));
        void* d = nullptr;
        BP_HIP(hipMalloc(&d, ops.size() * sizeof(bp::MegaOp)));
        dev.push_back(d);
        BP_HIP(hipMemcpy(d, ops.data(), ops.size() * sizeof(bp::MegaOp), hipMemcpyHostToDevice));
        a.prog[k] = (const bp::MegaOp*)d;
        a.n_ops[k] = (int)ops.size();
    }
    unsigned* sync = nullptr;
    BP_HIP(hipMalloc((void**)&sync, 129 * sizeof(unsigned)));
    dev.push_back(sync);
    a.sync = sync;
    a.nb = blocks_per_xcd;
    unsigned long long* d_st = nullptr;
    if (op_us) {
        BP_HIP(hipMalloc((void**)&d_st, 8 * 512 * sizeof(unsigned long long)));
        BP_HIP(hipMemset(d_st, 0, 8 * 512 * sizeof(unsigned long long)));
        dev.push_back(d_st);
        a.stamps = d_st;
    }
    hipEvent_t e0, e1;
    BP_HIP(hipEventCreate(&e0));
    BP_HIP(hipEventCreate(&e1));
    bp::launch_mega(a, lds, s);                 // warm
    BP_HIP(hipStreamSynchronize(s));
    BP_HIP(hipEventRecord(e0, s));
    for (int i = 0; i < iters; ++i) bp::launch_mega(a, lds, s);
    BP_HIP(hipEventRecord(e1, s));
    BP_HIP(hipEventSynchronize(e1));
    float ms = 0;
    BP_HIP(hipEventElapsedTime(&ms, e0, e1));
    if (ms_per_launch) *ms_per_launch = ms / iters;
    unsigned err = 0;
    BP_HIP(hipMemcpy(&err, sync + 128, sizeof(unsigned), hipMemcpyDeviceToHost));
    if (err_word) *err_word = err;
    if (op_us) {
        std::vector<unsigned long long> h(512);
        BP_HIP(hipMemcpy(h.data(), d_st, 512 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        std::vector<bp::MegaOp> ops;
        ys[0]->net->emit_conv_ops(1, ops);
        for (int i = 0; i < (int)ops.size() && i < cap; ++i) {
            op_us[i] = (float)((double)(h[i + 1] - h[i]) / 100.0);         // 100 MHz reference clock
            if (op_info) { op_info[3 * i] = ops[i].type; op_info[3 * i + 1] = ops[i].items; op_info[3 * i + 2] = ops[i].conv.splits; }
        }
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    for (void* d : dev) (void)hipFree(d);
    return 0;
    BP_CATCH
}
#endif

// ------------------------------------------------------------------ host post-processing
int bp_solve_pnp(const double* pts3d, const double* pts2d, int n, const double* K, double* R, double* t) {
    BP_TRY
    BP_CHECK(pts3d && pts2d && K && R && t, "null argument");
    const int rc = bp::solve_pnp(pts3d, pts2d, n, K, R, t);
    if (rc != 0) throw bp::Error("solve_pnp failed (need >= 6 non-degenerate points, or >= 4 coplanar ones)");
    return 0;
    BP_CATCH
}

int bp_solve_pnp_refined(const double* pts3d, const double* pts2d, int n, const double* K, double* R, double* t) {
    BP_TRY
    BP_CHECK(pts3d && pts2d && K && R && t, "null argument");
    const int rc = bp::solve_pnp_refined(pts3d, pts2d, n, K, R, t);
    if (rc != 0) throw bp::Error("solve_pnp_refined failed (need >= 6 non-degenerate points)");
    return 0;
    BP_CATCH
}

int bp_solve_pnp_ransac(const double* pts3d, const double* pts2d, int n, const double* K, double reproj_err,
                        int max_trials, double confidence, double* R, double* t, unsigned char* inliers) {
    BP_TRY
    BP_CHECK(pts3d && pts2d && K && R && t, "null argument");
    BP_CHECK(reproj_err > 0 && max_trials >= 1 && confidence > 0 && confidence < 1, "RANSAC parameters out of range");
    const int rc = bp::solve_pnp_ransac(pts3d, pts2d, n, K, reproj_err, max_trials, confidence, R, t, inliers);
    if (rc != 0) throw bp::Error("solve_pnp_ransac failed (need >= 6 points and a 6-point consensus)");
    return 0;
    BP_CATCH
}

int bp_pose_nms(const float* bboxes, const float* bbox_scores, const float* preds, const float* scores, int n, int K,
                int* out_pick, float* out_pose, float* out_score, float* out_prop) {
    BP_TRY
    BP_CHECK(n >= 0 && K >= 1, "bad sizes");
    BP_CHECK(n == 0 || (bboxes && bbox_scores && preds && scores && out_pick && out_pose && out_score && out_prop), "null argument");
    return bp::pose_nms(bboxes, bbox_scores, preds, scores, n, K, out_pick, out_pose, out_score, out_prop);
    BP_CATCH
}

// ------------------------------------------------------------------ frame input (host)
struct bp_loader {
    std::unique_ptr<bp::FrameLoader> l;
};

static void* pinned_alloc(size_t bytes) {
    void* p = nullptr;
    if (hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess) {
        (void)hipGetLastError();
        return nullptr;
    }
    return p;
}
static void pinned_free(void* p) { (void)hipHostFree(p); }

int bp_png_info(const unsigned char* data, size_t n, int* h, int* w, int* channels) {
    BP_TRY
    BP_CHECK(data, "null argument");
    bp::png_info(data, n, h, w, channels);
    return 0;
    BP_CATCH
}

int bp_png_decode_bgr(const unsigned char* data, size_t n, unsigned char* out_bgr, size_t cap, int* h, int* w) {
    BP_TRY
    BP_CHECK(data && out_bgr, "null argument");
    static thread_local std::vector<uint8_t> scratch;
    bp::png_decode_bgr(data, n, out_bgr, cap, h, w, scratch);
    return 0;
    BP_CATCH
}

int bp_loader_create(const char* const* paths, int n, int H, int W, int threads, int depth, int pinned, bp_loader** out) {
    BP_TRY
    BP_CHECK(paths && out && n >= 0, "null argument");
    std::vector<std::string> v;
    for (int i = 0; i < n; ++i) {
        BP_CHECK(paths[i], "null path");
        v.emplace_back(paths[i]);
    }
    int ndev = 0;
    const bool pin = pinned && hipGetDeviceCount(&ndev) == hipSuccess && ndev > 0;
    if (!pin) (void)hipGetLastError();
    auto* L = new bp_loader;
    try {
        L->l.reset(new bp::FrameLoader(std::move(v), H, W, threads, depth, pin ? pinned_alloc : nullptr,
                                       pin ? pinned_free : nullptr));
    } catch (...) {
        delete L;
        throw;
    }
    *out = L;
    return 0;
    BP_CATCH
}

void bp_loader_destroy(bp_loader* l) { delete l; }

int bp_loader_next(bp_loader* l, long long* index, const unsigned char** bgr) {
    BP_TRY
    BP_CHECK(l, "null argument");
    std::string err;
    const int rc = l->l->next(index, bgr, &err);
    if (rc < 0) g_err = err;
    return rc;
    BP_CATCH
}

int bp_loader_release(bp_loader* l, long long index) {
    BP_TRY
    BP_CHECK(l, "null argument");
    l->l->release(index);
    return 0;
    BP_CATCH
}

int bp_upload(void* d_dst, const void* h_src, size_t bytes, void* stream) {
    BP_TRY
    BP_CHECK(d_dst && h_src, "null argument");
    BP_HIP(hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, (hipStream_t)stream));
    return 0;
    BP_CATCH
}

}  // extern "C"
