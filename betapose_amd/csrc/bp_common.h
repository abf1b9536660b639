// Shared declarations for the betapose HIP engine (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <mutex>
#include <set>
#include <stdexcept>
#include <string>
#include <utility>

namespace bp {

struct Error : std::runtime_error {
    using std::runtime_error::runtime_error;
};

#define BP_HIP(expr)                                                                      \
    do {                                                                                  \
        hipError_t _e = (expr);                                                           \
        if (_e != hipSuccess)                                                             \
            throw ::bp::Error(std::string(#expr) + ": " + hipGetErrorString(_e) + " @" +  \
                              __FILE__ + ":" + std::to_string(__LINE__));                 \
    } while (0)

#define BP_CHECK(cond, msg)                                                               \
    do {                                                                                  \
        if (!(cond)) throw ::bp::Error(std::string(msg) + " (" #cond ") @" + __FILE__ +   \
                                       ":" + std::to_string(__LINE__));                   \
    } while (0)

// More than 64 KB of dynamic LDS needs an opt-in that belongs to the DEVICE's loaded function, not to the process: engines on a second
// GPU of the same process (the C API takes a device per engine) must set it again, and two host threads may make their first launch
// together (round-4 advisor finding: a function-local `static bool` covered neither).  One entry per (device, kernel).
inline void allow_big_lds(const void* kernel) {
    static std::mutex mu;
    static std::set<std::pair<int, const void*>> done;
    int dev = 0;
    BP_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lk(mu);
    if (done.count({dev, kernel})) return;
    BP_HIP(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64));
    done.insert({dev, kernel});
}

// The clock of the kernels' debug / in-situ stamps (ConvParams::stamps): s_memrealtime, the constant 100 MHz reference clock
// that is ONE counter for the whole device.  (s_memtime, the shader-cycle counter, is per XCD: marks of blocks on different
// XCDs differ by seconds, so a layer's first-entry -> last-store span cannot be taken from it.)
__device__ __forceinline__ unsigned long long bp_clock() { return __builtin_amdgcn_s_memrealtime(); }

enum Act : int { ACT_LINEAR = 0, ACT_LEAKY = 1, ACT_RELU = 2 };

// where the epilogue puts element (m = (b,oy,ox), n = out channel)
enum StoreMode : int {
    ST_NHWC = 0,     // out[m*ld + n]
    ST_UP2 = 1,      // nearest x2: the 4 pixels (2oy+dy, 2ox+dx) of a [2OH x 2OW] NHWC map
    ST_PIXSHUF = 2,  // PixelShuffle(2) with filters pre-permuted to n' = (i*2+j)*(Cout/4)+c
    ST_NCHW = 3      // out[(b*Cout + n)*OH*OW + oy*OW + ox]
};

// One fused convolution: implicit GEMM  D[m,n] = sum_k A[m,k] W[n,k],  k = (ky,kx,ci),
// NHWC activations, filters packed [CoutPad][Kpad] (BN folded), epilogue
//   v = acc + bias[n];  res_after_act ? (act(v) + r) : act(v + r),  r = res[m,n]*scale[b,n]
struct ConvParams {
    const float* in;
    int in_ld;
    int N, H, W, Cin;
    const float* w;
    int Kpad;   // multiple of 32
    int Ktrue;  // ksize*ksize*Cin
    const float* bias;
    float* out;
    int out_ld;
    int OH, OW, Cout;
    int ksize, stride, pad;
    int act;
    const float* res;
    int res_ld;
    const float* res_scale;  // [N][Cout] or null
    int res_after_act;
    int store_mode;
    int M;                 // N*OH*OW
    int nchunks;           // Kpad/32
    int splits;            // split-K factor (>=1)
    int chunks_per_split;
    float* partial;        // [splits][tiles][BM*64] fragment-order slabs when splits > 1
    int* tickets;          // [tiles] arrival counters (zero between launches) when splits > 1
    int CoutPad;
    int cin_pack;          // channels per filter tap in the packed K order (= Cin, or 4 for Cin <= 4 stems)
    const unsigned short* w16;   // 16-bit copy of w for mfma_mode: fp16 [CoutPad][Kpad], or three bf16 planes [3][CoutPad][Kpad]
    const unsigned short* w16s;  // the same operands stage-packed (aux_kernels.hip): filters-direct kernels, conv_kg.hip, conv_rd.hip
    int mfma_mode;         // Precision the launch uses (PREC_F16 / PREC_BF16X3 need w16 and an eligible layer)
    unsigned long long* stamps;   // debug (tools/bench_conv.py --stamps): per-block s_memtime marks, null in production
    // ---- operand planes (conv_pl.hip).  A tensor's planes MIRROR its fp32 view: element (pixel, c) of plane pl lives at
    // planes + pl * plane_elems + pixel * ld + c (same ld as the fp32 view, 2 bytes per element): three bf16 planes with
    // x == p0 + p1 + p2 exactly (PREC_BF16X3) or one fp16 plane (PREC_F16).  They are written by the PRODUCER's epilogue
    // (every conv kernel through conv_tail.inc, the pooling / shuffle kernels) and fetched by LDS-DMA in the consumer.
    const unsigned short* in16;   // planes of `in` (null: the layer runs on a kernel that reads fp32 activations)
    long long in16_plane;         // elements between planes
    unsigned short* out16;        // planes of `out` (null: not emitted)
    long long out16_plane;
    int out_np;                   // planes the epilogue emits: 0 none, 1 fp16, 3 bf16x3
    int skip_f32;                 // 1: every reader of `out` takes the planes -- the fp32 store is dropped (engine.cpp plan_planes)
    const unsigned short* res16;  // fp16 plane of the residual view `res` (same ld): the epilogue reads the residual from it instead of the
                                  // fp32 tensor (fp16 mode with fp16 skip connections, Net::set_f16_residuals; `res` stays set: it selects the epilogue)
    int net_prec;                 // the ENGINE's precision mode (Net::set_precision) -- for the layers the 16-bit modes do not cover by themselves (the
                                  // RGB stems keep mfma_mode == PREC_F32): which of their kernels fits the mode the rest of the network runs in
    int abl;                      // timing ablations, experimental builds only (-DBP_EXPERIMENTAL; tools/abl_pl.sh): 0 in the product
    const unsigned short* wbd;    // filters as stage-packed fragments in conv_pl.hip's K order (TILE_PL64BD: launch_f32_to_bf16x3_staged with Cin)
    const unsigned short* wpl;    // filters as conv_pl.hip's LDS image: [CoutPad/64][nchunks][plane][64 rows][64 B swizzled]
    // ---- launch layout by XCD (the hardware puts block b of a 1-D grid on XCD (b + c) % 8, every XCD with its own 4 MB L2;
    // tools/micro/xcc_map.hip: one c per queue, also with four streams in flight)
    //  * xcd_home (split-K launches): ALL K slices of an output tile run on ONE XCD (tile t on the XCD of block residue
    //    t % 8), so the slab hand-off never leaves that XCD's L2: workgroup-scope (sc0) slab stores and loads, an L2-local
    //    ticket -- instead of write-through to the memory side and back (conv_tail.inc).  Correctness then rests on the
    //    round-robin dispatch.  It is CHECKED in every launch: each slice publishes its XCC_ID (agent scope, xcc_of), the
    //    reducing block compares them with its own before it stores anything and on a mismatch raises err_word and skips the tile
    //    (the host re-runs the frame without the layout; round 4 trapped, which cost the process its whole HIP context).
    //  * filter prefetch for the NEXT convolution on the stream (lone-frame latency mode, Net::set_prefetch): the launch
    //    carries extra blocks past its work grid that pull the head of the next layer's filters into the L2 of the XCD whose
    //    blocks will read them, while this layer computes.  At batch 1 every layer's filters come cold from HBM (730 MB per
    //    frame do not stay in the 256 MB MALL), pulled by the few blocks that need them at 17-22 B/clk/CU where an L2 hit
    //    delivers 46-60 (tools/micro/cold_fetch.hip).  Under both layouts used here (xcd_home, and the plain one-slice grid
    //    of the 64x64 filters-direct kernel) the blocks of residue x read the N-tiles n == x (mod g), g = min(N-tiles, 8).
    // SE blocks: the global average pool of this convolution's output rides in its epilogue -- every 64-row tile writes
    // the column sums of its rows (fixed order; the linear, residual-free output = accumulator + bias) to
    // pool_out[tile_m][Cout], which the first fc of the block reads as `in_parts` slice sums (aux_kernels.hip fc_kernel)
    float* pool_out;              // nullptr: none
    // Hybrid grid (conv_pl.hip, launches with splits == 1 whose tile count leaves the chip badly filled at the end -- 296 tiles
    // of 256x128 on 256 CUs, 1 184 of 128x128): the first hy_full tiles run whole, the rest are cut hy_splits ways along K
    // (hy_cps chunks each) and combined by the usual last-arriver reduction, so that the tail of the launch spreads over all
    // CUs instead of running one more round on a few.  Block b < hy_full: tile b; else tile hy_full + (b - hy_full) / hy_splits.
    int hy_full, hy_splits, hy_cps;   // hy_splits == 0: off
    int xcd_home;                 // 1: block b -> x = b % 8, i = b / 8: tile (i / splits) * 8 + x, K slice i % splits
    int* xcc_of;                  // [tiles][64] XCC_ID of every K slice of the running launch
    int* err_word;                // raised (bit 0) by a reducing block whose slices did not all run on its XCD: the tile is NOT stored (conv_dev.h xcd_home_verify)
    int* tickets_local;           // [tiles] arrival counters of the xcd_home launches (touched by L2-local atomics only)
    int mtiles, n_tiles;          // M-tiles, output tiles of the launch (set by the launchers)
    int work_blocks;              // blocks of the work grid incl. padding (set by the launchers; 0 = the whole grid)
    const void* pf_ptr;           // next layer's filters (nullptr: no prefetch blocks)
    int pf_first;                 // first prefetch block (work_blocks rounded up to 8, set by the launchers)
    int pf_ntn, pf_splits, pf_cps, pf_nchunks;     // the next launch: N-tiles (a power of two), K slices, chunks per slice, chunks
    int pf_tile_stride, pf_chunk_bytes;            // its filter layout: bytes per N-tile and per 32-k chunk (contiguous per pair)
    int pf_cap;                   // bytes pulled per (N-tile, K-slice) pair at most
};

// tile configuration ids for launch_conv
// arithmetic of the matrix-core operands (accumulation and activations are always fp32)
// PREC_F16_RES is an engine-level mode (Net::set_precision): the kernels see PREC_F16 plus ConvParams::res16
enum Precision : int { PREC_F32 = 0, PREC_F16 = 1, PREC_BF16X3 = 2, PREC_F16_RES = 3 };

// TILE_W64_<WM>x<WN>: conv_w64.hip, WM x WN waves of 64x64 each (16-bit precision modes only)
// TILE_KG<G>: conv_kg.hip, 64x64 block tile, G groups of 4 waves each on its own K range (bf16x3 mode only)
enum ConvTile : int { TILE_64x64 = 0, TILE_128x64 = 1, TILE_W64_1x1 = 2, TILE_W64_1x2 = 3, TILE_W64_2x1 = 5, TILE_W64_2x2 = 6,
                      TILE_KG1 = 7, TILE_KG2 = 8, TILE_KG4 = 9,
                      // TILE_RD<W>: conv_rd.hip, one wave per 64x64 tile and K range, W K ranges per block, no LDS stage (bf16x3)
                      TILE_RD4 = 10, TILE_RD8 = 11,
                      // conv_igemm.hip's 64x64-block bf16x3 kernel with the filter fragments fetched straight into registers
                      TILE_64x64_BD = 12,
                      // conv_pl.hip: both operands by LDS-DMA from 16-bit planes, <block tile>, waves x wave tile
                      TILE_PL64 = 13,       // 64x64, 2x2 waves of 32x32
                      TILE_PL128 = 14,      // 128x128, 2x2 waves of 64x64
                      TILE_PL128x64 = 15,   // 128x64, 2x2 waves of 64x32
                      TILE_PL256x128 = 16,  // 256x128, 4x2 waves of 64x64
                      TILE_STEM3 = 20,      // the 3x3 / stride-1 / 3-channel stem as a direct convolution on the vector pipe (conv_igemm.hip stem3x3_kernel)
                      TILE_PL64BD = 19,     // 64x64 on operand planes, filter fragments direct from global memory (conv_pl.hip BDIR; bf16x3)
                      TILE_PL64K2 = 18,     // 64x64, two K groups of 2x2 waves (8 waves, a ring per group): for launches of at most one block per CU
                      TILE_PL128S = 17,     // 128x128, 2x2 compute waves of 64x64 + 4 loader waves (wave specialisation; no K slices)
                      // conv_halo.hip (round 4): 3x3 / stride 1, the nine taps read from ONE LDS-resident activation halo per 32-channel group;
                      // 64 rows x 32 columns per wave, filter fragments direct from global memory (bf16x3)
                      TILE_HALO64 = 21,     // 64x64 block, 2 waves
                      TILE_HALO128 = 22,    // 64x128 block, 4 waves
                      TILE_HALO64K2 = 23,   // 64x64 block, 4 waves = 2 K groups x 2 column halves: the K split inside the block (no slabs)
                      TILE_BD_K2 = 24,      // the filters-direct 64x64 tile with 8 waves = 2 K groups of 2x2 waves (conv_igemm.hip, any kernel size / stride)
                      TILE_PLH128 = 25,     // conv_pl.hip 128x128 with the activations of a 3x3 / stride-1 layer from an LDS-resident halo (fp16; round 4)
                      TILE_S1 = 26,         // conv_s1.hip: the 1x1 layers of the batched fp16 runs as a persistent streaming kernel (32-row M-tiles resident in LDS, 128-column passes; round 5)
                      TILE_P3 = 27,         // conv_p3.hip: the 3x3 / stride-1 layers of the batched fp16 runs as a persistent kernel (128 pixels x 32 columns per wave, filter fragments global -> registers, zero-padded halo in LDS with the taps as instruction immediates, register-only epilogue; round 6)
                      TILE_STEM7 = 28,      // the key-point detector's 7x7 / stride-2 RGB stem on the fp16 matrix pipe (conv_igemm.hip stem7x7_f16_kernel; fp16 modes only; round 6)
                      TILE_LAST = 27,       // (the last id a policy may force)
                      TILE_FUSED = 40 };    // reporting only (Net::profile): the op is the last member of a block fused into one launch (conv_fused.hip)

// when non-null, launch_conv brackets the implicit-GEMM kernel itself (not the split-K reduce) with these events
struct ConvProfHook { hipEvent_t e0, e1; };
extern thread_local ConvProfHook* g_conv_prof;
bool conv_h16_eligible(const ConvParams& p);   // layer can run on the 16-bit-operand kernels (w16 present, Cin % 32 == 0)
int conv_vec_mode(const ConvParams& p);   // 0 scalar gather, 1 Cin % 32 == 0, 2 four-channel-packed stem

void launch_conv(const ConvParams& p, int tile, hipStream_t s);
void launch_conv_w64(const ConvParams& p, int tile, hipStream_t s);   // conv_w64.hip
bool conv_tile_is_w64(int tile);
void launch_conv_kg(const ConvParams& p, int tile, hipStream_t s);    // conv_kg.hip
bool conv_tile_is_kg(int tile);
void launch_conv_rd(const ConvParams& p, int tile, hipStream_t s);    // conv_rd.hip
bool conv_tile_is_rd(int tile);
void launch_conv_pl(const ConvParams& p, int tile, hipStream_t s);    // conv_pl.hip
bool conv_tile_is_pl(int tile);
void launch_conv_halo(const ConvParams& p, int tile, hipStream_t s);  // conv_halo.hip
bool conv_tile_is_halo(int tile);
bool conv_halo_eligible(const ConvParams& p, int tile);   // 3x3 / stride 1 / pad 1, Cin % 32 == 0, stage-packed filters, W within the LDS budget
// K slices of a launch: `want` slices asked for -> slices and chunks per slice the kernel `tile` runs (the halo tiles cut K
// by whole 32-channel groups = 9 chunks)
void conv_split_plan(const ConvParams& p, int tile, int want, int* splits, int* cps);
inline bool conv_tile_is_plh(int tile) { return tile == TILE_PLH128; }
bool conv_plh_eligible(const ConvParams& p);  // TILE_PLH128 can run the layer (fp16, 3x3 / stride 1 / pad 1, W <= 126, whole channel groups per K slice)
bool conv_pl_eligible(const ConvParams& p);   // planes + wpl present, Cin % 32 == 0, taps fit the 32-bit mask
bool conv_s1_eligible(const ConvParams& p, long long M);   // TILE_S1 can run the layer at M output pixels (fp16, 1x1 / stride 1 or 2, NHWC store, 64 <= K <= 1024, N >= 128, M >= 2048)
void launch_conv_s1(const ConvParams& p, hipStream_t s);   // conv_s1.hip
bool conv_p3_eligible(const ConvParams& p, long long M);   // TILE_P3 can run the layer at M output pixels (fp16, 3x3 / stride 1 / pad 1, NHWC store, W one of 13 / 16 / 26 / 32 / 52 / 104, M >= 4096)
void launch_conv_p3(const ConvParams& p, hipStream_t s);   // conv_p3.hip
// filters [CoutPad][Kpad] fp32 -> conv_pl.hip's LDS image, np = 1 (fp16) or 3 (exact bf16 split)
void launch_pack_wpl(const float* in, unsigned short* out, int CoutPad, int Kpad, int Cin, int ksize, int np, hipStream_t s);
// fp32 NHWC view -> its operand planes (producers that are not convolutions; tests)
void launch_f32_to_planes(const float* in, int ld, long long pixels, int C, unsigned short* planes, long long plane_elems,
                          int np, hipStream_t s);
// ... and back (test taps of tensors whose fp32 store was dropped): exact for np == 3
void launch_planes_to_f32(const unsigned short* planes, long long plane_elems, int np, float* out, int ld, long long pixels, int C,
                          hipStream_t s);
// conv_fused.hip (round 5): [1x1] -> [3x3 / stride 1] (-> [1x1]) + skip connection of one residual / bottleneck block in ONE launch on
// 8 x 8 output patches (bf16x3, fp32 activations); `post` may be null
bool conv_fused_eligible(const ConvParams& pre, const ConvParams& c3, const ConvParams* post);
int conv_fused_blocks(const ConvParams& pre, const ConvParams& c3, const ConvParams* post);
void launch_conv_fused(const ConvParams& pre, const ConvParams& c3, const ConvParams* post, hipStream_t s);
int conv_tiles(const ConvParams& p, int tile);   // blocks per K-slice
bool conv_stem3_eligible(const ConvParams& p);   // conv_igemm.hip: the layer can run on TILE_STEM3
bool conv_stem7_eligible(const ConvParams& p);   // ... on TILE_STEM7 (7x7 / stride 2 / pad 3, 4-channel-packed RGB in, 64 channels out)
void conv_grid_setup(ConvParams& q, int bm, int bn);   // fills mtiles / n_tiles / work_blocks / pf_first from M, CoutPad, splits, xcd_home
int conv_grid_blocks(const ConvParams& q);
int xcc_base();                                                // engine.cpp: XCC_ID of block 0 (round-robin dispatch), -1: unusable
bool conv_hybrid_plan(const ConvParams& p, int tile, size_t partial_floats, int* full, int* hs, int* hcps);   // engine.cpp
bool conv_home_layout(int tile, int splits);                   // engine.cpp: the launch keeps all K slices of a tile on one XCD
void conv_prefetch_of(ConvParams& p, const ConvParams& next, int next_tile, int next_splits, int next_cps);   // engine.cpp              // work blocks + padding + prefetch blocks
int conv_tile_bm(int tile);
int conv_tile_bn(int tile);

// ---- the persistent per-XCD launch (mega.inc, compiled in kernels_unity.hip): one frame's launch list per XCD
enum MegaOpType : int { MO_CONV_BD = 0, MO_CONV_K2 = 1, MO_CONV_F32V2 = 2, MO_STEM3 = 3 };

struct MegaOp {
    int type, items, npass, pad;
    ConvParams conv;
};
struct MegaArgs {
    const MegaOp* prog[8];     // launch list of the frame on XCD x
    int n_ops[8];
    unsigned* sync;            // [8][16] per XCD {joined, arrived, ...} then [128] = error word; zeroed before every launch
    int nb;                    // blocks per XCD
    unsigned long long* stamps; // debug: [8][512] s_memrealtime after every op's barrier (block of rank 0), null in production
};

size_t mega_lds_bytes(const MegaOp& o);                                  // dynamic LDS the op's body needs
void mega_make_conv_op(const ConvParams& p, int tile, MegaOp* out);        // p as launch_conv would get it (tile, slices, workspaces set)
void launch_mega(const MegaArgs& a, size_t lds_bytes, hipStream_t s);      // zeroes a.sync, launches 8 * a.nb blocks

// ---- auxiliary kernels (aux_kernels.hip) ----
void launch_nchw_to_nhwc(const float* in, float* out, int N, int C, int H, int W, hipStream_t s);
void launch_nhwc_to_nchw(const float* in, int in_ld, float* out, int N, int C, int H, int W, hipStream_t s);
// planes / plane_elems / np: operand planes of `out` written alongside (ConvParams::out16; null = none)
void launch_maxpool3s2p1(const float* in, float* out, int N, int H, int W, int C, int OH, int OW, hipStream_t s,
                         unsigned short* planes = nullptr, long long plane_elems = 0, int np = 0);
void launch_add(const float* a, int a_ld, const float* b, int b_ld, float* out, int out_ld,
                long long pixels, int C, hipStream_t s);
void launch_upsample2(const float* in, int in_ld, float* out, int out_ld, int N, int H, int W, int C, hipStream_t s);
void launch_copy_channels(const float* in, int in_ld, float* out, int out_ld, long long pixels, int C, hipStream_t s);
void launch_pixel_shuffle2(const float* in, float* out, int N, int H, int W, int C, hipStream_t s,
                           unsigned short* planes = nullptr, long long plane_elems = 0, int np = 0);
// out [N][P][C] per-slice sums, P = avgpool_parts(HW); the consumer (launch_fc, in_parts=P) finishes the mean
int avgpool_parts(int HW);
void launch_avgpool(const float* in, int in_ld, float* out, int N, int HW, int C, hipStream_t s);
// out[b][o] = act(bias[o] + sum_i w[o][i]*x[b][i]);  act: 2 relu, 3 sigmoid;
// x = in (in_parts == 1) or in_scale * sum_p in[b][p][i]
void launch_fc(const float* in, const float* w, const float* bias, float* out, int N, int Cin, int Cout,
               int act, int in_parts, float in_scale, hipStream_t s);

struct YoloHead {
    const float* t;  // NHWC [N][g][g][nA*attrs], ld = nA*attrs
    int g;
    float aw[3], ah[3];  // anchors in pixels
    int row_off;         // first row of this head in the [rows][attrs] output
};
// pred: [N][rows][attrs] (DetectionLayer row order); sel: [N][8] floats
// (idx as int bits, x1,y1,x2,y2,obj,cls_conf,cls_idx); idx = -1 when nothing > conf
void launch_yolo_decode(const YoloHead* heads, int nheads, int N, int reso, int attrs, int rows,
                        float* pred, hipStream_t s);
// sel_ld: floats between consecutive images' records (8 dense; the fused pipeline writes straight into its result rows)
void launch_yolo_select(const float* pred, int N, int rows, int attrs, float conf, int num_classes,
                        float* sel, hipStream_t s, int sel_ld = 8);
// both in one launch, for callers that read the select record only (no [rows][attrs] tensor is written): same records
void launch_yolo_decode_select(const YoloHead* heads, int nheads, int N, int reso, int attrs, int rows, float conf, int num_classes,
                               float* sel, hipStream_t s, int sel_ld = 8);
// hm NCHW [N][C][H*W] -> out [N][C][6] = (idx as int bits, max, left, right, up, down)
// out_ld: floats between consecutive images' [C][6] blocks (0 = dense C*6)
void launch_heatmap_argmax(const float* hm, int N, int C, int H, int W, float* out, hipStream_t s, int out_ld = 0);

void launch_f32_to_f16(const float* in, unsigned short* out, long long n, hipStream_t s);
void launch_f32_to_bf16x3(const float* in, unsigned short* out_planes, long long n, hipStream_t s);
void launch_f32_to_f16_staged(const float* in, unsigned short* out, int CoutPad, int Kpad, hipStream_t s);      // ... fp16 mode
void launch_f32_to_bf16x3_staged(const float* in, unsigned short* out, int CoutPad, int Kpad, hipStream_t s, int Cin_pl = 0);   // Cin_pl > 0: K in conv_pl.hip's order (32-channel group, tap, channel)   // conv_kg.hip's layout
void launch_probe_placement(int* d_out, int blocks, hipStream_t s);
void launch_spin_ticks(long long ticks, hipStream_t s);   // one thread spinning until bp_clock() has advanced by `ticks`

// crop stage (dataloader.py:794-835 + img.py:242-262) on device.
//  frames: BGR u8 [batch][H][W][3]; sel: [batch][8] select records (box in YOLO-input pixels) or boxes [batch][4];
//  out_nhwc [oh][ow][3] (engine input) and/or out_nchw [3][oh][ow]; pts: (ul.x, ul.y, br.x, br.y)
void launch_crop(const uint8_t* frames, int batch, int H, int W, const float* sel, int reso, const float* boxes,
                 float* out_nhwc, float* out_nchw, float* pts, int oh, int ow, hipStream_t s, int sel_ld = 8, int pts_ld = 8);

// Pillow-exact antialiased bicubic resize of a u8 HWC frame (two passes, 22-bit fixed point);
// coefficient tables are built on the host (engine.cpp).
struct ResizeTables {
    const int* hb;      // [ow][2] (xmin, xsize)
    const int* hk;      // [ow][ksize_h] fixed-point coeffs
    int ksize_h;
    const int* vb;      // [oh][2]
    const int* vk;      // [oh][ksize_v]
    int ksize_v;
};
// in: u8 [H][W][3] (BGR when swap_rb) -> tmp u8 [H][ow][3] -> out f32 NHWC [oh][ow][3] RGB /255
void launch_resize_bicubic(const uint8_t* in, int batch, int H, int W, uint8_t* tmp, float* out_nhwc, uint8_t* out_u8,
                           int oh, int ow, const ResizeTables& t, int swap_rb, hipStream_t s);

}  // namespace bp
