// Network plans for the two models of the hot path, built on the fused kernels.
#pragma once
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "bp_common.h"

namespace bp {

struct Tensor {       // NHWC view
    float* p = nullptr;
    int H = 0, W = 0, C = 0;
    int ld = 0;       // floats between consecutive pixels (>= C; > C inside a concat buffer)
};

class Arena {
public:
    ~Arena();
    float* alloc(size_t n_floats);
    void* alloc_bytes(size_t bytes);
    size_t total_bytes() const { return total_; }
private:
    std::vector<void*> ptrs_;
    size_t total_ = 0;
};

enum OpType : int {
    OP_CONV, OP_MAXPOOL, OP_ADD, OP_UPSAMPLE, OP_COPYCH, OP_PIXSHUF, OP_AVGPOOL, OP_FC
};

struct Op {
    OpType type;
    ConvParams conv{};   // OP_CONV (per-image geometry; N/M/partial patched at run time)
    int tile = TILE_64x64;
    // generic operands
    const float* a = nullptr; int a_ld = 0;
    const float* b = nullptr; int b_ld = 0;
    float* out = nullptr; int out_ld = 0;
    int H = 0, W = 0, C = 0, OH = 0, OW = 0;
    // fc
    const float* w = nullptr; const float* bias = nullptr; int Cin = 0, Cout = 0, act = 0;
    int in_parts = 1; float in_scale = 1.f;
    float* pool_out = nullptr;   // OP_CONV: the buffer of the OP_AVGPOOL that follows (its slice sums can ride in this conv's epilogue)
    unsigned short* out16 = nullptr; long long out16_plane = 0;   // operand planes of `out` (non-conv producers; set_precision)
    std::string name;
    double flops = 0;    // per image
    double bytes = 0;    // algorithmic bytes per image (weights counted once per launch elsewhere)
};

struct ConvWeights {   // one conv as it arrives in the stream (un-folded)
    const float* w = nullptr;          // OIHW
    const float* bn_bias = nullptr, *bn_scale = nullptr, *bn_mean = nullptr, *bn_var = nullptr;
    const float* bias = nullptr;
};

// packed (BN-folded) filters on the device, shared by every engine clone that serves another stream
struct WeightStore {
    Arena arena;
    std::vector<float*> ptrs;
    std::map<const float*, unsigned short*> f16;      // fp16 copies of packed filters, made on first use
    std::map<const float*, unsigned short*> bf16x3;   // three bf16 planes per filter (PREC_BF16X3)
    std::map<const float*, unsigned short*> bf16x3s;  // ... and their stage-packed copy (filters-direct kernels, conv_kg/rd.hip)
    std::map<const float*, unsigned short*> f16s;     // stage-packed fp16 copy
    std::map<const float*, unsigned short*> wbd3;         // stage-packed bf16x3 fragments in conv_pl.hip's K order (TILE_PL64BD)
    std::map<const float*, unsigned short*> wpl1, wpl3;   // conv_pl.hip's LDS image of the filters: fp16 / three bf16 planes
    std::mutex f16_mutex;
};

class Net {
public:
    explicit Net(int max_batch, std::shared_ptr<WeightStore> store = nullptr)
        : max_batch_(max_batch), store_(store ? store : std::make_shared<WeightStore>()), reuse_(store != nullptr) {}
    std::shared_ptr<WeightStore> weight_store() const { return store_; }
    virtual ~Net() = default;
    int max_batch() const { return max_batch_; }
    void run_ops(int batch, hipStream_t s);
    void run_op(const Op& op, int batch, hipStream_t s);
    void prepare_conv(const Op& op, int batch, ConvParams& p, int& tile);
#ifdef BP_EXPERIMENTAL
    void emit_conv_ops(int batch, std::vector<MegaOp>& out);   // xcd-mode prototype (mega.inc): the pass's convolution launches as descriptors
#endif
    // eager profiling pass: mean device ms per op; info[i] = {is_conv, tile, kernel (0 scalar-gather fp32, 1 vector
    // fp32, 2 fp16-MFMA, 3 bf16x3), splits}
    int profile(int batch, int iters, float* ms, int* info, int cap, hipStream_t s);
    size_t device_bytes() const { return arena_.total_bytes(); }
    const std::vector<Op>& ops() const { return ops_; }
    // test hook: copy a recorded intermediate (NHWC view) to a dense NCHW device buffer
    int tap_count() const { return (int)taps_.size(); }
    const char* tap_name(int i) const { return tap_names_[i].c_str(); }
    void tap_shape(int i, int* C, int* H, int* W) const { *C = taps_[i].C; *H = taps_[i].H; *W = taps_[i].W; }
    void tap_copy(int i, int batch, float* d_out_nchw, hipStream_t s);
    void set_splitk_policy(int target_blocks, int min_chunks) { sk_target_ = target_blocks; sk_min_chunks_ = min_chunks; ++plan_version_; }
    void set_max_splits(int m) { sk_max_splits_ = m; ++plan_version_; }
    void set_force_tile(int t) { force_tile_ = t; ++plan_version_; }
    // PREC_F16 / PREC_BF16X3: eligible convs (Cin % 32 == 0) run on the 16-bit MFMA kernels with converted filter copies
    // prec 3 (PREC_F16_RES): the fp16 mode with fp16 SKIP CONNECTIONS -- a residual is read from the fp16 plane its producer wrote
    // for the next convolution, and a tensor that only convolutions and residual adds read loses its fp32 store (ConvParams::res16)
    void set_precision(int prec);
    int precision() const { return precision_; }
    int precision_id() const { return (precision_ == PREC_F16 && f16_res_) ? PREC_F16_RES : precision_; }
    unsigned plan_version() const { return plan_version_; }
    // in-situ timing: every convolution launch whose grid has at most `slots` blocks writes 8 u64 s_memrealtime (100 MHz) marks per block
    // (entry, index math done, -, K loop done, stores done, slab parked, slices combined, -) at d_buf + (conv ordinal * slots +
    // block) * 8; null switches it off.  Graphs must be re-captured (plan_version).
    void set_stamps(unsigned long long* d_buf, int slots) { stamps_ = d_buf; stamp_slots_ = slots; ++plan_version_; }
    // lone-frame latency mode: split-K hand-off inside one XCD's L2 + blocks that pull the next layer's filters into the L2
    // that will read them (bp_common.h ConvParams::xcd_home / pf_*).  Off by default: +4.7 % one frame at a time, -1 .. -2 %
    // with two to four in flight
    void set_prefetch(bool on) { prefetch_ = on; ++plan_version_; }
    // conv -> conv fusion of whole residual / bottleneck blocks (conv_fused.hip, round 5; on by default, BP_NO_FUSION=1 / set_fusion(false)
    // for the unfused plan: A/B runs, the fused-vs-unfused tests)
    void set_fusion(bool on) { fusion_ = on; ++plan_version_; }
    bool fusion() const { return fusion_; }
    int fused_launches(int batch);        // groups the current plan runs as ONE launch at this batch size
    int take_xcd_errors(hipStream_t s);   // non-zero: some launch of the latency mode found a K slice on the wrong XCD and skipped its tile -- run the frame again without the mode
    bool pool_in_epilogue(const Op& conv, int batch, int tile) const;
    bool pooled_by_conv(const Op& pool, int batch) const;
    bool prefetch() const { return prefetch_; }
    const char* op_name(int i) const { return ops_[i].name.c_str(); }

protected:
    // emit a fused conv; returns index into ops_
    int add_conv(const std::string& name, const Tensor& in, const Tensor& out_view, const ConvWeights& cw, int Cout,
                 int k, int stride, int pad, int act, int store_mode, const Tensor* res, const float* res_scale,
                 int res_after_act, float bn_eps, int OH, int OW);
    void add_tap(const std::string& name, const Tensor& t) { taps_.push_back(t); tap_names_.push_back(name); }
    Tensor new_tensor(int H, int W, int C);
    void finalize();   // allocate split-K workspace
    size_t workspace_need() const;

    float* upload_weights(const float* host, size_t count);   // through the shared store (reused by clones)
    // activation allocations (new_tensor) and their operand planes: planes mirror the fp32 allocation element for element
    // (bp_common.h ConvParams::in16), so every view (pointer, ld) into an allocation has its planes view for free
    struct ActAlloc { float* base = nullptr; size_t elems = 0; unsigned short* planes = nullptr; bool wanted = false;
                      bool f32_read = true; };   // f32_read: something reads the fp32 tensor (a residual, a pooling kernel, a head)
    std::vector<ActAlloc> acts_;
    ActAlloc* find_act(const float* p);
    void plan_planes(int prec, int mix_hw = 0);   // allocate the planes 16-bit consumers need, point every producer / consumer at them
    int max_batch_;
    std::shared_ptr<WeightStore> store_;
    bool reuse_;
    size_t store_cursor_ = 0;
    Arena arena_;
    std::vector<Op> ops_;
    std::vector<Tensor> taps_;
    std::vector<std::string> tap_names_;
    float* partial_ = nullptr;
    size_t partial_floats_ = 0;
    int* tickets_ = nullptr;
    size_t tickets_count_ = 0;
    int sk_target_ = 512, sk_min_chunks_ = 4, sk_max_splits_ = 8;
    int force_tile_ = -1;
    bool darknet_bn_ = false;
    int precision_ = PREC_F32;
    unsigned long long* stamps_ = nullptr;
    int stamp_slots_ = 0;
    bool prefetch_ = false;
    // fusion groups: consecutive convolutions [1x1] -> [3x3 / stride 1] (-> [1x1]) whose intermediate tensors nobody else reads (found once, in
    // finalize()); which of them run fused is a property of the plan (precision, batch size): roles_ caches it per (batch, plan version)
    struct FuseGroup { int pre, c3, post; };
    enum FuseRole : int { FR_NONE = 0, FR_SKIP = 1, FR_HEAD2 = 2, FR_HEAD3 = 3 };
    std::vector<FuseGroup> fuse_groups_;
    std::vector<int> roles_, role_group_;
    int roles_batch_ = -1;
    unsigned roles_version_ = ~0u;
    bool fusion_ = true;
    void find_fuse_groups();
    void plan_roles(int batch);
    void run_op_unfused(const Op& op, int batch, hipStream_t s);
    bool f16_res_ = false;        // fp16 skip connections (set_precision(PREC_F16_RES))
    unsigned plan_version_ = 0;   // bumped whenever launches would change (captured graphs must be rebuilt)
};

class YoloNet : public Net {
public:
    // darknet_bn: fold BatchNorm the Darknet-C way, scale / (sqrt(var) + 1e-6) (blas.c:136, network.c:827-834), instead
    // of PyTorch's scale / sqrt(var + 1e-5) (yolo/darknet.py:256) -- for the Darknet-API-compatible detector
    YoloNet(const std::string& cfg_text, const float* stream, size_t n_floats, int reso, int max_batch,
            std::shared_ptr<WeightStore> store = nullptr, bool darknet_bn = false);
    YoloNet* clone() const { return new YoloNet(cfg_text_, nullptr, n_floats_, reso_, max_batch_, store_, darknet_bn_); }
    int rows() const { return rows_; }
    int attrs() const { return attrs_; }
    int reso() const { return reso_; }
    // img: NCHW f32 [B,3,reso,reso] RGB 0..1 (or NHWC when nhwc_input); pred [B,rows,attrs] (may be null when sel given)
    void forward(const float* d_img, bool nhwc_input, int batch, float* d_pred, float conf, int num_classes,
                 float* d_sel, hipStream_t s, int sel_ld = 8);
    float* input_nhwc() { return in_nhwc_; }
    float* pred_buffer() { return pred_; }
private:
    std::string cfg_text_;
    size_t n_floats_ = 0;
    int reso_, rows_ = 0, attrs_ = 0;
    float* in_nhwc_ = nullptr;
    float* pred_ = nullptr;
    std::vector<YoloHead> heads_;
};

class KpdNet : public Net {
public:
    KpdNet(const float* stream, size_t n_floats, int n_classes, int max_batch, int inH = 320, int inW = 256,
           std::shared_ptr<WeightStore> store = nullptr);
    KpdNet* clone() const { return new KpdNet(nullptr, n_floats_, n_classes_, max_batch_, inH_, inW_, store_); }
    int out_c() const { return outC_; }
    int out_h() const { return inH_ / 4; }
    int out_w() const { return inW_ / 4; }
    int in_h() const { return inH_; }
    int in_w() const { return inW_; }
    // inps: NCHW f32 [B,3,320,256] (or NHWC); hm NCHW [B,50,80,64] (null -> internal); kp [B,50,6] (nullable)
    void forward(const float* d_inps, bool nhwc_input, int batch, float* d_hm, float* d_kp, hipStream_t s, int kp_ld = 0);
    float* input_nhwc() { return in_nhwc_; }
private:
    size_t n_floats_ = 0;
    int n_classes_ = 0;
    int inH_, inW_, outC_;
    float* in_nhwc_ = nullptr;
    float* hm_ = nullptr;
    int hm_op_ = -1;
};

// host-side Pillow coefficient tables (a1)
struct ResizePlan {
    int in_size = 0, out_size = 0, ksize = 0;
    std::vector<int> bounds, coeffs;
};
ResizePlan make_bicubic_plan(int in_size, int out_size);

}  // namespace bp
