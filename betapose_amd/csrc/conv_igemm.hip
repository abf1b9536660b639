// Fused convolution for gfx950: implicit-GEMM (no im2col buffer) on the fp32 MFMA
// pipe (v_mfma_f32_32x32x2_f32: exact fp32 FMA chain, 256 FLOP/clk/CU), NHWC
// activations, BN-folded filters packed [Cout][ky][kx][ci], epilogue fusing bias,
// LeakyReLU/ReLU, residual add (YOLO shortcut / ResNet bottleneck), SE channel
// scale, nearest-x2 upsample, PixelShuffle(2) and NCHW stores.
//
// Replaces, for the hot path, cudnnConvolutionForward + normalize/scale_bias/
// add_bias/activate/shortcut/upsample kernels of the reference's Darknet CUDA
// backend (train_YOLO/src/convolutional_kernels.cu:121-383, blas_kernels.cu) and
// the torch.nn Conv2d/BatchNorm2d/LeakyReLU/ReLU/PixelShuffle modules of
// yolo/darknet.py:240-259 and KPD/src/models/layers/{SE_Resnet,DUC}.py.
//
// Block = 256 threads = 4 waves (one per SIMD), 2x2 over a (64*TM)x(64*TN) output
// tile; K is walked in chunks of 32 (one filter-tap slice: ci0..ci0+31 are
// contiguous in NHWC).  global -> registers -> LDS, register prefetch TWO chunks
// ahead (two register sets, loop unrolled by two), LDS double buffer, one barrier
// per chunk.  The chunk phase is an explicit software pipeline (ablation on MI355X: with
// MFMAs, loads and stores all disabled the first version still took 60 % of its time --
// per chunk ~550 cycles of instruction issue and ~480 cycles parked at waits/barrier ran
// back-to-back with the 1024 MFMA cycles): the prefetch loads, their address math and the
// LDS stores of the NEXT chunk are issued between the MFMAs of the current one (an MFMA
// occupies the matrix pipe for 64 cycles after a ~8-cycle issue), and the one barrier sits
// BEFORE the last MFMA group, so the next chunk's first fragments are read from LDS under
// those MFMAs.  Main-loop instruction diet (rocprofv3: the first version spent 29 %
// of wave cycles issuing non-MFMA instructions): the chunk -> (tap, ci0) walk is
// wave-uniform scalar state advanced by increments (no divisions); every lane
// keeps one 32-bit byte offset per tile row plus a bit-mask of which filter taps
// fall inside the image; loads go through raw buffer descriptors, so an
// out-of-image tap is just an out-of-range offset that the hardware zero-fills.
// LDS rows are padded to 36 floats: the ds_read_b128 fragment reads are bank-
// conflict free (row stride 144 B covers all 64 banks over 16 rows).
// Fragment trick: lane l reads A[row=l&31][4h..4h+3] (h=l>>5) as one b128 and feeds
// component j to MFMA j, so MFMA j contracts k = {j, 4+j} of the 8-wide sub-chunk;
// B uses the same k mapping, and the sum over k is order-free.
#include <cstdlib>

#include "conv_dev.h"

namespace bp {

// VEC: how a thread fetches its 4 consecutive K elements of an A row --
//   0  scalar gather (any Cin): 4 address computations + 4 dword loads per element group;
//   1  Cin % 32 == 0: a 32-wide chunk lies inside one filter tap, one 16-B load, wave-uniform K walk;
//   2  Cin <= 4 with filters packed 4 channels per tap (the two RGB stems): one 16-B load per TAP -- the thread's
//      group is tap chunk*8 + (tid & 7); with Cin = 3 the 4th float it reads is the neighbouring pixel's first channel
//      (or 0 past the tensor), which meets a zero filter entry.
// (kernel bodies are __device__ functions of (launch descriptor, block index, LDS base, LDS flag word): the __global__ kernels
// below them are thin wrappers, and the persistent per-XCD kernel of mega.inc calls the same bodies with block indices of its own)
template <int TM, int TN, int VEC, class P>
__device__ __forceinline__ void conv_igemm_body(const P& p, const int bp_bid, float* const smem, int* const s_last_p) {
    constexpr int BM = 64 * TM, BN = 64 * TN;
    constexpr int RA = BM / 32, RB = BN / 32;
    float (*As)[BM * LDS_LD] = reinterpret_cast<float (*)[BM * LDS_LD]>(smem);
    float (*Bs)[BN * LDS_LD] = reinterpret_cast<float (*)[BN * LDS_LD]>(smem + 2 * BM * LDS_LD);
    constexpr int LDT = BN + 4;                 // row stride of the output tile staged through LDS in the epilogue
    static_assert(BM * LDT <= 2 * (BM + BN) * LDS_LD, "LDS too small for the staged epilogue");

    const unsigned long long t_entry = p.stamps ? bp_clock() : 0ull;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int n_tiles_n = p.CoutPad / BN;
    // 1-D grid, K-slice fastest: consecutive block ids (= consecutive XCDs) take different K-slices of the same
    // output tile, so with 8 slices every XCD streams its own 1/8 of the filters through its private L2
    const int split = bp_bid % p.splits;
    const int tile_id = bp_bid / p.splits;
    const int tile_n = tile_id % n_tiles_n;
    const int tile_m = tile_id / n_tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int c_begin = split * p.chunks_per_split;
    const int c_end = min(p.nchunks, c_begin + p.chunks_per_split);

    const int lr = tid >> 3, c4 = tid & 7;
    const int hw = p.OH * p.OW;

    const __amdgpu_buffer_rsrc_t rsrcA = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.in), 0, (int)min((long long)p.N * p.H * p.W * p.in_ld * 4, (long long)OOB), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrcB =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.w), 0, p.CoutPad * p.Kpad * 4, 0x00020000);

    // per tile row: byte offset of the (ky=0,kx=0,ci=c4*4) element (may be "negative" = before the image: wraps,
    // only ever used added to a tap delta that brings it back in range) and the in-image tap mask
    unsigned a_base[RA];
    unsigned long long a_mask[RA];
    int a_iy0[RA], a_ix0[RA], a_bh[RA];   // scalar-gather path only
    const float rcp_hw = 1.0f / (float)hw, rcp_ow = 1.0f / (float)p.OW;
#pragma unroll
    for (int i = 0; i < RA; ++i) {
        const int m = m0 + lr + 32 * i;
        const bool ok = m < p.M;
        const int mm = ok ? m : 0;
        const int b = fast_div(mm, hw, rcp_hw);
        const int rem = mm - b * hw;
        const int oy = fast_div(rem, p.OW, rcp_ow);
        const int ox = rem - oy * p.OW;
        const int iy0 = oy * p.stride - p.pad, ix0 = ox * p.stride - p.pad;
        a_base[i] = (unsigned)((((b * p.H + iy0) * p.W + ix0) * p.in_ld + (VEC == 2 ? 0 : c4 * 4)) * 4);
        // taps inside the image: kx in [kx_lo, kx_hi), ky in [ky_lo, ky_hi)
        const int kx_lo = max(0, -ix0), kx_hi = min(p.ksize, p.W - ix0);
        const int ky_lo = max(0, -iy0), ky_hi = min(p.ksize, p.H - iy0);
        unsigned long long mask = 0;
        if (ok && kx_hi > kx_lo) {
            const unsigned long long rowbits = ((1ull << kx_hi) - 1ull) & ~((1ull << kx_lo) - 1ull);
            for (int ky = ky_lo; ky < ky_hi; ++ky) mask |= rowbits << (ky * p.ksize);
        }
        a_mask[i] = mask;
        a_bh[i] = b * p.H;
        a_iy0[i] = ok ? iy0 : -(1 << 20);
        a_ix0[i] = ix0;
    }
    unsigned b_base[RB];
#pragma unroll
    for (int i = 0; i < RB; ++i) b_base[i] = (unsigned)(((n0 + lr + 32 * i) * p.Kpad + c4 * 4) * 4);

    // wave-uniform walk over K: chunk -> (tap, ky, kx, ci0); one division here, increments afterwards
    const int cpt = VEC == 1 ? (p.Cin >> 5) : 1;  // chunks per filter tap
    int w_c = c_begin;                        // next chunk to load
    int w_tap = VEC == 1 ? c_begin / cpt : 0;
    int w_ci = VEC == 1 ? (c_begin - w_tap * cpt) << 5 : 0;
    int w_ky = VEC == 1 ? w_tap / p.ksize : 0;
    int w_kx = VEC == 1 ? w_tap - w_ky * p.ksize : 0;
    const float rcp_ks = 1.0f / (float)p.ksize;

    f32x4 ra0[RA], rb0[RB], ra1[RA], rb1[RB];

    // address registers for the NEXT prefetch: computed one phase ahead (under the last MFMAs of a phase), so
    // issuing the loads is 4 instructions.  BP_ADDR also advances the walk (clamped at the last chunk: the
    // speculative loads past the end re-read it and are never used).
    unsigned va[RA];
    int sb = 0, ld_c = 0;
#define BP_ADDR()                                                                                      \
    {                                                                                                  \
        if constexpr (VEC == 1) {                                                                      \
            const unsigned delta = (unsigned)(((w_ky * p.W + w_kx) * p.in_ld + w_ci) * 4);            \
            _Pragma("unroll") for (int i = 0; i < RA; ++i) {                                           \
                const bool ok = (a_mask[i] >> w_tap) & 1ull;                                           \
                va[i] = ok ? a_base[i] + delta : OOB;                                                  \
            }                                                                                          \
        } else if constexpr (VEC == 2) {                                                               \
            const int tap = w_c * 8 + c4;                   /* this thread's tap of the chunk */        \
            const int ky = fast_div(tap, p.ksize, rcp_ks);                                             \
            const int kx = tap - ky * p.ksize;                                                         \
            const unsigned delta = (unsigned)(((ky * p.W + kx) * p.in_ld) * 4);                        \
            _Pragma("unroll") for (int i = 0; i < RA; ++i) {                                           \
                const bool ok = tap < 64 && ((a_mask[i] >> tap) & 1ull);                               \
                va[i] = ok ? a_base[i] + delta : OOB;                                                  \
            }                                                                                          \
        }                                                                                              \
        ld_c = w_c;                                                                                    \
        sb = w_c * (BK * 4);                                                                           \
        if (w_c + 1 < c_end) {                                                                         \
            ++w_c;                                                                                     \
            if constexpr (VEC == 1) {                                                                  \
                w_ci += 32;                                                                            \
                if (w_ci == p.Cin) {                                                                   \
                    w_ci = 0;                                                                          \
                    ++w_tap;                                                                           \
                    if (++w_kx == p.ksize) { w_kx = 0; ++w_ky; }                                       \
                }                                                                                      \
            }                                                                                          \
        }                                                                                              \
    }
#define BP_LOAD_A(ra_)                                                                                 \
    {                                                                                                  \
        if constexpr (VEC != 0) {                                                                      \
            _Pragma("unroll") for (int i = 0; i < RA; ++i) ra_[i] = buf_load4(rsrcA, va[i], 0);        \
        } else {                                                                                       \
            _Pragma("unroll") for (int i = 0; i < RA; ++i) {                                           \
                float v_[4];                                                                           \
                _Pragma("unroll") for (int e = 0; e < 4; ++e) {                                        \
                    const int k = ld_c * BK + c4 * 4 + e;                                              \
                    float x = 0.f;                                                                     \
                    if (k < p.Ktrue) {                                                                 \
                        const int tap = k / p.cin_pack;                                                \
                        const int ci = k - tap * p.cin_pack;                                           \
                        const int ky = tap / p.ksize;                                                  \
                        const int kx = tap - ky * p.ksize;                                             \
                        const int iy = a_iy0[i] + ky, ix = a_ix0[i] + kx;                              \
                        if ((unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W && ci < p.Cin) \
                            x = p.in[((long long)(a_bh[i] + iy) * p.W + ix) * p.in_ld + ci];           \
                    }                                                                                  \
                    v_[e] = x;                                                                         \
                }                                                                                      \
                ra_[i] = f32x4{v_[0], v_[1], v_[2], v_[3]};                                            \
            }                                                                                          \
        }                                                                                              \
    }
#define BP_LOAD_B(rb_)                                                                                 \
    _Pragma("unroll") for (int i = 0; i < RB; ++i) rb_[i] = buf_load4(rsrcB, b_base[i], sb);
#define BP_STORE_A(buf_, ra_)                                                                          \
    _Pragma("unroll") for (int i = 0; i < RA; ++i)                                                     \
        *reinterpret_cast<f32x4*>(&As[buf_][(lr + 32 * i) * LDS_LD + c4 * 4]) = ra_[i];
#define BP_STORE_B(buf_, rb_)                                                                          \
    _Pragma("unroll") for (int i = 0; i < RB; ++i)                                                     \
        *reinterpret_cast<f32x4*>(&Bs[buf_][(lr + 32 * i) * LDS_LD + c4 * 4]) = rb_[i];
#define BP_STORE_LDS(buf_, ra_, rb_)                                                                   \
    {                                                                                                  \
        _Pragma("unroll") for (int i = 0; i < RA; ++i)                                                 \
            *reinterpret_cast<f32x4*>(&As[buf_][(lr + 32 * i) * LDS_LD + c4 * 4]) = ra_[i];            \
        _Pragma("unroll") for (int i = 0; i < RB; ++i)                                                 \
            *reinterpret_cast<f32x4*>(&Bs[buf_][(lr + 32 * i) * LDS_LD + c4 * 4]) = rb_[i];            \
    }

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int frag_off = (lane & 31) * LDS_LD + (lane >> 5) * 4;
    const int a_off = wm * (BM / 2) * LDS_LD + frag_off;
    const int b_off = wn * (BN / 2) * LDS_LD + frag_off;

    // fragment registers for two consecutive 8-wide sub-chunks
    f32x4 fa0[TM], fb0[TN], fa1[TM], fb1[TN];
#define BP_RD(buf_, ks_, fa_, fb_)                                                                     \
    {                                                                                                  \
        _Pragma("unroll") for (int i = 0; i < TM; ++i) fa_[i] =                                        \
            *reinterpret_cast<const f32x4*>(&As[buf_][a_off + i * 32 * LDS_LD + (ks_) * 8]);           \
        _Pragma("unroll") for (int j = 0; j < TN; ++j) fb_[j] =                                        \
            *reinterpret_cast<const f32x4*>(&Bs[buf_][b_off + j * 32 * LDS_LD + (ks_) * 8]);           \
    }
#define BP_MF(fa_, fb_, comp_)                                                                         \
    _Pragma("unroll") for (int i = 0; i < TM; ++i) _Pragma("unroll") for (int j = 0; j < TN; ++j)      \
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa_[i].comp_, fb_[j].comp_, acc[i][j], 0, 0, 0);
#define BP_SB() __builtin_amdgcn_sched_barrier(0)
    // One chunk phase.  On entry: LDS[cur_] holds chunk c and (fa0, fb0) its sub-chunk 0; (rna_, rnb_) hold chunk
    // c+1 (in flight since the previous phase); (rfa_, rfb_) receive chunk c+2, whose addresses (va, sb) were
    // computed under the previous phase's last MFMAs.  The barrier guarantees both "everybody's chunk c+1 is in
    // LDS[cur_^1]" and "everybody is done reading LDS[cur_]".  Every non-MFMA group is small enough for the
    // 64-cycle shadow of the MFMA issued just before it.
#define BP_PHASE(cur_, rna_, rnb_, rfa_, rfb_)                                                         \
    {                                                                                                  \
        BP_RD(cur_, 1, fa1, fb1);                    BP_SB();                                          \
        BP_MF(fa0, fb0, x);  BP_LOAD_A(rfa_);        BP_SB();                                          \
        BP_MF(fa0, fb0, y);  BP_LOAD_B(rfb_);        BP_SB();                                          \
        BP_MF(fa0, fb0, z);                          BP_SB();                                          \
        BP_MF(fa0, fb0, w);  BP_RD(cur_, 2, fa0, fb0);  BP_SB();                                       \
        BP_MF(fa1, fb1, x);  BP_STORE_A((cur_) ^ 1, rna_);  BP_SB();                                   \
        BP_MF(fa1, fb1, y);  BP_STORE_B((cur_) ^ 1, rnb_);  BP_SB();                                   \
        BP_MF(fa1, fb1, z);                          BP_SB();                                          \
        BP_MF(fa1, fb1, w);  BP_RD(cur_, 3, fa1, fb1);  BP_SB();                                       \
        BP_MF(fa0, fb0, x);                                                                            \
        BP_MF(fa0, fb0, y);                                                                            \
        BP_MF(fa0, fb0, z);                                                                            \
        BP_MF(fa0, fb0, w);                          BP_SB();                                          \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                             \
        __syncthreads();                             BP_SB();                                          \
        BP_RD((cur_) ^ 1, 0, fa0, fb0);              BP_SB();                                          \
        BP_MF(fa1, fb1, x);  BP_ADDR();              BP_SB();                                          \
        BP_MF(fa1, fb1, y);                                                                            \
        BP_MF(fa1, fb1, z);                                                                            \
        BP_MF(fa1, fb1, w);                          BP_SB();                                          \
    }

#define BP_STAMP(k_) if (p.stamps && tid == 0) p.stamps[(long long)bp_bid * 8 + (k_)] = bp_clock();
    if (p.stamps && tid == 0) p.stamps[(long long)bp_bid * 8 + 0] = t_entry;
    BP_STAMP(1);   // index math done
    if (c_begin < c_end) {
        // prologue: chunks c, c+1 in flight; park c in LDS[0]; first fragments; addresses of chunk c+2
        BP_ADDR(); BP_LOAD_A(ra0); BP_LOAD_B(rb0);
        BP_ADDR(); BP_LOAD_A(ra1); BP_LOAD_B(rb1);
        BP_ADDR();
        BP_STORE_A(0, ra0);
        BP_STORE_B(0, rb0);
        __syncthreads();
        BP_STAMP(2);   // first chunk landed in LDS
        BP_RD(0, 0, fa0, fb0);
        // two chunks per trip (the register sets swap roles; after a pair everything is back in place), then an
        // odd tail.  Past the last chunk the pipeline keeps loading / parking / reading the clamped last chunk
        // into buffers nobody consumes -- no conditionals inside a phase, one loop exit (a mid-body exit made
        // hipcc ping-pong the 16 accumulator registers through v_accvgpr_mov every phase).
        const int nch = c_end - c_begin;
        for (int it = 0; it < (nch >> 1); ++it) {
            BP_PHASE(0, ra1, rb1, ra0, rb0);
            BP_PHASE(1, ra0, rb0, ra1, rb1);
        }
        if (nch & 1) BP_PHASE(0, ra1, rb1, ra0, rb0);
    }
    __syncthreads();
    BP_STAMP(3);   // K loop done

    const int w_row0 = wm * (BM / 2), w_col0 = wn * (BN / 2);
#define BP_NT 256
#define BP_SLAST (*s_last_p)
#define BP_TAIL_STAMP(k_) BP_STAMP(k_)
#define BP_EP_RES_SCALE true
#include "conv_tail.inc"
#undef BP_EP_RES_SCALE
#undef BP_TAIL_STAMP
#undef BP_NT
#undef BP_SLAST
    if (p.stamps) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); BP_STAMP(4); }
}

template <int TM, int TN, int VEC>
__global__ __launch_bounds__(256) void conv_igemm_kernel(const ConvParams p) {
    __shared__ __attribute__((aligned(16))) float smem[2 * (64 * TM + 64 * TN) * LDS_LD];
    __shared__ int s_last;
    conv_igemm_body<TM, TN, VEC>(p, (int)blockIdx.x, smem, &s_last);
}

// The product launches ONE instantiation of the template below, <1, 1, 3, true> ("filters direct", the bf16x3 mode's planned
// kernel: profiles/r03_ab_pipeline.txt); the LDS-staged variants, the fp16 variants and the 128x64 tile are instantiated in
// the experimental library only (build.py --experimental).
// =====================================================================================================================
// 16-bit-operand MFMA variants.  Same tiling, K walk, split-K hand-off and epilogue (conv_tail.inc) as the fp32 kernel;
// what changes is what the matrix cores multiply.  Activations stay fp32 in HBM and are converted when a chunk is
// parked in LDS; filters are converted once on the device from the packed fp32 ones; accumulation and outputs are fp32.
//   NP = 1  fp16 operands (BASELINE.json configs[2], "fp16 MFMA conv path"): round-to-nearest fp16 of both operands,
//           one v_mfma_f32_32x32x16_f16 per 16 k (16x the fp32 MFMA rate).  Results carry fp16 rounding (~1e-3).
//   NP = 3  fp32-accurate on the bf16 pipe: every fp32 operand x is split EXACTLY into three bf16 terms
//           x = x1 + x2 + x3 (8 + 8 + 8 significand bits; bf16 has fp32's exponent range, so no overflow/underflow
//           cases), and a*b is formed from the six partial products whose weight is >= 2^-16 of the leading one:
//           a1b1 + a1b2 + a2b1 + a1b3 + a2b2 + a3b1 (each exact in the fp32 accumulator path).  The three dropped terms
//           are <= 2^-23 relative -- below the rounding of the fp32 accumulation itself (measured: 3e-8 vs 3e-6
//           relative to an fp64 conv) -- at 6/16 of the fp32-MFMA cycles.  LDS carries three planes per operand.
// Fragment layout of the 32x32x16 forms: lane l supplies row (l & 31), k = 8*(l>>5) .. 8*(l>>5)+7 (one 16-B LDS read).
// =====================================================================================================================
// timing experiments only (tools/ablate_pipeline.sh, wrong results): -DBP_ABLATE_SPLIT parks raw bits instead of the
// three-way split (same LDS traffic, no conversions), -DBP_ABLATE_MFMA reads the fragments but multiplies nothing
#ifdef BP_ABLATE_SPLIT
static constexpr bool BP_ABL_SPLIT = true;
#else
static constexpr bool BP_ABL_SPLIT = false;
#endif
#ifdef BP_ABLATE_MFMA
static constexpr bool BP_ABL_MFMA = true;
#else
static constexpr bool BP_ABL_MFMA = false;
#endif
// BD ("filters direct", NP = 3, 64x64 tile): the filter fragments skip LDS -- every wave fetches the six 1 KB fragments
// of its 32 columns straight from the stage-packed copy ConvParams::w16s (fully coalesced, one chunk ahead) -- so LDS
// carries the activations only: 36 KB per chunk instead of 72 KB.  In the pipeline LDS bandwidth is what the K loops
// run into (profiles/r02_ablate_pipeline.txt: with the MFMAs AND the operand split compiled out the frame gets 10 %
// faster, with the K loops cut to one chunk 84 %); the price is that a fragment is fetched by the two waves that share
// its columns (32 KB instead of 20 KB per chunk through the vector-memory path, which has the room).
template <int TM, int TN, int NP, bool BD>
struct IgemmHLds {
    static constexpr int BM = 64 * TM, BN = 64 * TN, LDT = BN + 4;
    static constexpr int STAGE_HALFS = NP * (BM + (BD ? 0 : BN)) * LDH;
    static constexpr int FLOATS = (2 * STAGE_HALFS / 2 > BM * LDT) ? 2 * STAGE_HALFS / 2 : BM * LDT;
};
// KG = 2 (filters-direct variant only, round 4): the block is EIGHT waves = two K groups of the 2x2 waves described above; group g
// walks the g-th half of the block's chunk range with its own LDS stages and its own filter prefetch, the two partial sums meet in
// LDS behind the K loops (fixed order: deterministic), one tail per tile.  A launch is a latency chain (K loop ~ chunks x memory
// latency / prefetch depth): two groups halve it without a slab hand-off, or halve the K slices between blocks where slices remain.
template <int TM, int TN, int NP, bool BD, class P, int KG = 1>
__device__ __forceinline__ void conv_igemm_h_body(const P& p, const int bp_bid, float* const smem, int* const s_last_p) {
    static_assert(KG == 1 || (KG == 2 && BD), "K groups: filters-direct variant");
    static_assert(!BD || (TM == 1 && TN == 1), "filters-direct variant: 64x64 tile");
    constexpr int BM = 64 * TM, BN = 64 * TN;
    constexpr int RA = BM / 64;          // fp32 A rows per thread (8 consecutive floats each: two 16-B loads, one 16-B LDS store per plane)
    constexpr int RBH = BN / 64;         // 16-bit B rows per thread and plane (8 elements each)
    constexpr int LDT = BN + 4;
    constexpr int STAGE_HALFS = IgemmHLds<TM, TN, NP, BD>::STAGE_HALFS;
    unsigned short* const sh = reinterpret_cast<unsigned short*>(smem) + (KG > 1 ? ((int)threadIdx.x >> 8) * (2 * STAGE_HALFS) : 0);   // the K group's stages
    typedef typename HalfOps<NP>::frag frag_t;
    // stage s: NP planes of A rows, then NP planes of B rows
#define BH_AS(s_, pl_) (sh + (s_) * STAGE_HALFS + (pl_) * (BM * LDH))
#define BH_BS(s_, pl_) (sh + (s_) * STAGE_HALFS + NP * (BM * LDH) + (pl_) * (BN * LDH))

    if (p.work_blocks && bp_bid >= p.work_blocks) { prefetch_block<256 * KG>(p, reinterpret_cast<char*>(smem), bp_bid); return; }
    const unsigned long long t_entry = p.stamps ? bp_clock() : 0ull;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int kgrp = KG > 1 ? __builtin_amdgcn_readfirstlane(wave >> 2) : 0;     // K group of the wave
    const int tq = KG > 1 ? (tid & 255) : tid;                                   // thread inside its K group
    const int wm = KG > 1 ? ((wave >> 1) & 1) : (wave >> 1), wn = wave & 1;
    const int n_tiles_n = p.CoutPad / BN;
    int split, tile_n, tile_m;
    if (p.xcd_home) {  // all K slices of a tile on one XCD (ConvParams::xcd_home)
        const int i = bp_bid >> 3, tl = i / p.splits;
        const int t = tl * 8 + (bp_bid & 7);
        if (t >= p.n_tiles) return;                      // padding of the last round of tiles
        split = i - tl * p.splits;
        tile_n = t % n_tiles_n;
        tile_m = t / n_tiles_n;
        xcd_home_mark(p, t, split);
    } else {
        split = bp_bid % p.splits;
        const int t = bp_bid / p.splits;
        tile_n = t % n_tiles_n;
        tile_m = t / n_tiles_n;
    }
    const int tile_id = tile_m * n_tiles_n + tile_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int cb_blk = split * p.chunks_per_split;
#ifdef BP_ABLATE_KLOOP   // timing experiment only (wrong results): one chunk per block, i.e. the fixed cost of the launch chain
    const int ce_blk = min(p.nchunks, cb_blk + 1);
#else
    const int ce_blk = min(p.nchunks, cb_blk + p.chunks_per_split);
#endif
    // K groups: group g takes chunks [cb_blk + g per, ...) of the block's range; every group makes the same number of trips (`nch`
    // below): past its own range a group multiplies zero filter fragments (out-of-range loads)
    const int kg_per = (ce_blk - cb_blk + KG - 1) / KG;
    const int c_begin = KG > 1 ? min(cb_blk + kgrp * kg_per, ce_blk) : cb_blk;
    const int c_end = KG > 1 ? min(c_begin + kg_per, ce_blk) : ce_blk;

    const int lr = tq >> 2, a8 = (tq & 3) * 8; // A: row lr (+64i), floats a8..a8+7
    const int br = tq >> 2, b8 = (tq & 3) * 8; // B: row br (+64i), elements b8..+7
    const int hw = p.OH * p.OW;
    const int plane_bytes = p.CoutPad * p.Kpad * 2;

    const __amdgpu_buffer_rsrc_t rsrcA = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.in), 0, (int)min((long long)p.N * p.H * p.W * p.in_ld * 4, (long long)OOB), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrcB = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<unsigned short*>(BD ? p.w16s : p.w16), 0, NP * plane_bytes, 0x00020000);
    // BD: fragment (plane, k-step) of chunk c for this wave's 32 columns = 1 KB at bd_tile + (2 c + k-step) * 6 KB +
    // plane * 2 KB; lane -> row 32 wn + (lane & 31), granule (lane >> 5) at slot granule ^ ((row >> 3) & 1)
    const int bd_tile = tile_n * (p.Kpad >> 4) * (NP * 64 * 32);
    const unsigned bd_voff = (unsigned)((32 * wn + (lane & 31)) * 32 + (((lane >> 5) ^ ((lane >> 3) & 1)) << 4));
    // BD: the six filter fragments (k-step ks in rb_[pl][ks]) of the next chunk of the block's range; past the range the
    // offset is out of range (zeros, no memory traffic)
    int bd_c = c_begin;
#define BH_LOAD_BD(rb_)                                                                                \
    {                                                                                                  \
        const int so_ = bd_c < c_end ? bd_tile + bd_c * (2 * NP * 64 * 32) : (int)OOB;                 \
        ++bd_c;                                                                                        \
        _Pragma("unroll") for (int pl = 0; pl < NP; ++pl)                                              \
            _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                           \
                rb_[pl][ks] = __builtin_amdgcn_raw_buffer_load_b128(rsrcB, (int)bd_voff, so_ + ks * (NP * 64 * 32) + pl * 2048, 0); \
    }
    constexpr int RBN = BD ? 2 : RBH;
    u32x4 rb0[NP][RBN], rb1[NP][RBN];
    if constexpr (BD) {   // the first chunk's filters need no index math: requested before it (a cold kernel waits > 1 us for its first operands)
        BH_LOAD_BD(rb1);  // the first phase multiplies with rb1
        __builtin_amdgcn_sched_barrier(0);
    }

    unsigned a_base[RA];
    unsigned long long a_mask[RA];
    const float rcp_hw = 1.0f / (float)hw, rcp_ow = 1.0f / (float)p.OW;
#pragma unroll
    for (int i = 0; i < RA; ++i) {
        const int m = m0 + lr + 64 * i;
        const bool ok = m < p.M;
        const int mm = ok ? m : 0;
        const int b = fast_div(mm, hw, rcp_hw);
        const int rem = mm - b * hw;
        const int oy = fast_div(rem, p.OW, rcp_ow);
        const int ox = rem - oy * p.OW;
        const int iy0 = oy * p.stride - p.pad, ix0 = ox * p.stride - p.pad;
        a_base[i] = (unsigned)((((b * p.H + iy0) * p.W + ix0) * p.in_ld + a8) * 4);
        const int kx_lo = max(0, -ix0), kx_hi = min(p.ksize, p.W - ix0);
        const int ky_lo = max(0, -iy0), ky_hi = min(p.ksize, p.H - iy0);
        unsigned long long mask = 0;
        if (ok && kx_hi > kx_lo) {
            const unsigned long long rowbits = ((1ull << kx_hi) - 1ull) & ~((1ull << kx_lo) - 1ull);
            for (int ky = ky_lo; ky < ky_hi; ++ky) mask |= rowbits << (ky * p.ksize);
        }
        a_mask[i] = mask;
    }
    unsigned b_base[RBH];
#pragma unroll
    for (int i = 0; i < RBH; ++i) b_base[i] = (unsigned)(((n0 + br + 64 * i) * p.Kpad + b8) * 2);

    // wave-uniform walk over K, as in the fp32 kernel
    const int cpt = p.Cin >> 5;
    int w_c = c_begin;
    int w_tap = c_begin / cpt;
    int w_ci = (c_begin - w_tap * cpt) << 5;
    int w_ky = w_tap / p.ksize;
    int w_kx = w_tap - w_ky * p.ksize;
    unsigned va[RA];
    int sb = 0;
#define BH_ADDR()                                                                                      \
    {                                                                                                  \
        const unsigned delta = (unsigned)(((w_ky * p.W + w_kx) * p.in_ld + w_ci) * 4);                \
        _Pragma("unroll") for (int i = 0; i < RA; ++i) {                                               \
            const bool ok = (a_mask[i] >> w_tap) & 1ull;                                               \
            va[i] = ok ? a_base[i] + delta : OOB;                                                      \
        }                                                                                              \
        sb = w_c * (BK * 2);                                                                           \
        if (w_c + 1 < c_end) {                                                                         \
            ++w_c;                                                                                     \
            w_ci += 32;                                                                                \
            if (w_ci == p.Cin) {                                                                       \
                w_ci = 0;                                                                              \
                ++w_tap;                                                                               \
                if (++w_kx == p.ksize) { w_kx = 0; ++w_ky; }                                           \
            }                                                                                          \
        }                                                                                              \
    }
#define BH_LOAD(ra_, rb_)                                                                              \
    {                                                                                                  \
        _Pragma("unroll") for (int i = 0; i < RA; ++i) {                                               \
            ra_[2 * i] = buf_load4(rsrcA, va[i], 0);                                                   \
            ra_[2 * i + 1] = buf_load4(rsrcA, va[i], 16);                                              \
        }                                                                                              \
        if constexpr (!BD) {                                                                           \
            _Pragma("unroll") for (int pl = 0; pl < NP; ++pl)                                          \
                _Pragma("unroll") for (int i = 0; i < RBH; ++i)                                        \
                    rb_[pl][i] = __builtin_amdgcn_raw_buffer_load_b128(rsrcB, (int)b_base[i] + pl * plane_bytes, sb, 0); \
        }                                                                                              \
    }

#define BH_STORE(s_, ra_, rb_)                                                                         \
    {                                                                                                  \
        _Pragma("unroll") for (int i = 0; i < RA; ++i) {                                               \
            unsigned short* dst = BH_AS(s_, 0) + (lr + 64 * i) * LDH + a_st_off;                       \
            const f32x4 lo = ra_[2 * i], hi = ra_[2 * i + 1];                                          \
            if constexpr (NP == 1) {                                                                   \
                const f16x4 l = __builtin_convertvector(lo, f16x4), h = __builtin_convertvector(hi, f16x4); \
                *reinterpret_cast<f16x8*>(dst) = __builtin_shufflevector(l, h, 0, 1, 2, 3, 4, 5, 6, 7); \
            } else if constexpr (BP_ABL_SPLIT) {   /* timing experiment: same LDS traffic, no conversions */ \
                *reinterpret_cast<f32x4*>(dst) = lo;                                                   \
                *reinterpret_cast<f32x4*>(dst + BM * LDH) = hi;                                        \
                *reinterpret_cast<f32x4*>(dst + 2 * BM * LDH) = lo;                                    \
            } else {                                                                                   \
                const bf16x4 l1 = __builtin_convertvector(lo, bf16x4), h1 = __builtin_convertvector(hi, bf16x4); \
                const f32x4 rl1 = lo - __builtin_convertvector(l1, f32x4), rh1 = hi - __builtin_convertvector(h1, f32x4); \
                const bf16x4 l2 = __builtin_convertvector(rl1, bf16x4), h2 = __builtin_convertvector(rh1, bf16x4); \
                const f32x4 rl2 = rl1 - __builtin_convertvector(l2, f32x4), rh2 = rh1 - __builtin_convertvector(h2, f32x4); \
                const bf16x4 l3 = __builtin_convertvector(rl2, bf16x4), h3 = __builtin_convertvector(rh2, bf16x4); \
                *reinterpret_cast<bf16x8*>(dst) = __builtin_shufflevector(l1, h1, 0, 1, 2, 3, 4, 5, 6, 7); \
                *reinterpret_cast<bf16x8*>(dst + BM * LDH) = __builtin_shufflevector(l2, h2, 0, 1, 2, 3, 4, 5, 6, 7); \
                *reinterpret_cast<bf16x8*>(dst + 2 * BM * LDH) = __builtin_shufflevector(l3, h3, 0, 1, 2, 3, 4, 5, 6, 7); \
            }                                                                                          \
        }                                                                                              \
        if constexpr (!BD) {                                                                           \
            _Pragma("unroll") for (int pl = 0; pl < NP; ++pl)                                          \
                _Pragma("unroll") for (int i = 0; i < RBH; ++i)                                        \
                    *reinterpret_cast<u32x4*>(BH_BS(s_, pl) + (br + 64 * i) * LDH + b_st_off) = rb_[pl][i]; \
        }                                                                                              \
    }

    // the epilogue's bias (4 consecutive channels per thread, conv_tail.inc) is requested before the K loop: at the
    // block's end it would be a cold load of > 1 us on the critical path (the array is padded to CoutPad)
    const f32x4 bias_early = *reinterpret_cast<const f32x4*>(p.bias + min(n0 + (tid % (BN / 4)) * 4, p.CoutPad - 4));
    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // swizzled element offsets inside a row (all row bases used below are multiples of 32 rows, so (row >> 1) & 3
    // depends on the thread's own row index only)
    const int a_st_off = ((tq & 3) ^ ((lr >> 1) & 3)) << 3;
    const int b_st_off = ((tq & 3) ^ ((br >> 1) & 3)) << 3;
    const int frow = lane & 31, fsw = (frow >> 1) & 3;
    const int frag_ks0 = frow * LDH + (((lane >> 5)) ^ fsw) * 8;          // logical granule (lane>>5)     (k-step 0)
    const int frag_ks1 = frow * LDH + ((2 + (lane >> 5)) ^ fsw) * 8;      // logical granule 2 + (lane>>5) (k-step 1)
    const int a_row0 = wm * (BM / 2) * LDH;
    const int b_row0 = wn * (BN / 2) * LDH;

    f32x4 ra0[2 * RA], ra1[2 * RA];
    // partial products (A plane, B plane), smallest first
    constexpr int NPROD = NP == 1 ? 1 : 6;
    constexpr int PA[6] = {2, 1, 0, 1, 0, 0};
    constexpr int PB[6] = {0, 1, 2, 0, 1, 0};
    // One chunk: LDS[cur_] holds chunk c; (rna_, rnb_) hold chunk c+1 (in flight since the previous phase);
    // (rfa_, rfb_) receive chunk c+2, whose addresses were computed in the previous phase.
    // BD: rnb_ holds the filter fragments of chunk c (requested in the previous phase), rfb_ receives those of c+1.
#define BH_PHASE(cur_, rna_, rnb_, rfa_, rfb_)                                                         \
    {                                                                                                  \
        BH_LOAD(rfa_, rfb_);                                                                           \
        if constexpr (BD) BH_LOAD_BD(rfb_);                                                            \
        _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) {                                             \
            frag_t fa[NP][TM], fb[NP][TN];                                                             \
            _Pragma("unroll") for (int pl = 0; pl < NP; ++pl) {                                        \
                _Pragma("unroll") for (int i = 0; i < TM; ++i) fa[pl][i] =                             \
                    *reinterpret_cast<const frag_t*>(BH_AS(cur_, pl) + a_row0 + i * 32 * LDH + (ks ? frag_ks1 : frag_ks0)); \
                _Pragma("unroll") for (int j = 0; j < TN; ++j) {                                       \
                    if constexpr (BD) fb[pl][j] = __builtin_bit_cast(frag_t, rnb_[pl][ks]);            \
                    else fb[pl][j] = *reinterpret_cast<const frag_t*>(BH_BS(cur_, pl) + b_row0 + j * 32 * LDH + (ks ? frag_ks1 : frag_ks0)); \
                }                                                                                      \
            }                                                                                          \
            _Pragma("unroll") for (int q = 0; q < NPROD; ++q)                                          \
                _Pragma("unroll") for (int i = 0; i < TM; ++i) _Pragma("unroll") for (int j = 0; j < TN; ++j) \
                    if constexpr (BP_ABL_MFMA) asm volatile("" :: "v"(fa[NP == 1 ? 0 : PA[q]][i]), "v"(fb[NP == 1 ? 0 : PB[q]][j])); \
                    else acc[i][j] = HalfOps<NP>::mfma(fa[NP == 1 ? 0 : PA[q]][i], fb[NP == 1 ? 0 : PB[q]][j], acc[i][j]); \
        }                                                                                              \
        BH_STORE((cur_) ^ 1, BH_PARKED(rna_, rfa_), rnb_);                                             \
        BH_ADDR();                                                                                     \
        __syncthreads();                                                                               \
    }
#ifdef BP_ABLATE_PREFETCH   // timing experiment (BD variant only, results stay right): activations ONE chunk ahead instead of two
#define BH_PARKED(rn_, rf_) rf_
#else
#define BH_PARKED(rn_, rf_) rn_
#endif

    if (KG > 1 ? cb_blk < ce_blk : c_begin < c_end) {
        BH_ADDR(); BH_LOAD(ra0, rb0);
#ifndef BP_ABLATE_PREFETCH
        BH_ADDR(); BH_LOAD(ra1, rb1);
#endif
        BH_ADDR();
        BH_STORE(0, ra0, rb0);
        __syncthreads();
        const int nch = KG > 1 ? kg_per : c_end - c_begin;
        for (int it = 0; it < (nch >> 1); ++it) {
            BH_PHASE(0, ra1, rb1, ra0, rb0);
            BH_PHASE(1, ra0, rb0, ra1, rb1);
        }
        if (nch & 1) BH_PHASE(0, ra1, rb1, ra0, rb0);
    }
    __syncthreads();
    if constexpr (KG > 1) {
        // the groups' partial sums meet in LDS: group 1 parks its accumulators in fragment order (16 B per lane and store), group 0 adds
        float* const xch = smem + (wave & 3) * 1024 + lane * 4;
        if (kgrp == 1) {
#pragma unroll
            for (int q = 0; q < 4; ++q)
                *reinterpret_cast<f32x4*>(xch + q * 256) = f32x4{acc[0][0][4 * q], acc[0][0][4 * q + 1], acc[0][0][4 * q + 2], acc[0][0][4 * q + 3]};
        }
        __syncthreads();
        if (kgrp == 0) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(xch + q * 256);
                acc[0][0][4 * q] += v.x; acc[0][0][4 * q + 1] += v.y; acc[0][0][4 * q + 2] += v.z; acc[0][0][4 * q + 3] += v.w;
            }
        }
        __syncthreads();
    }
#define BH_STAMP(k_) if (p.stamps && tid == 0) p.stamps[(long long)(tile_id * p.splits + split) * 8 + (k_)] = bp_clock();
    if (p.stamps && tid == 0) p.stamps[(long long)(tile_id * p.splits + split) * 8 + 0] = t_entry;
    BH_STAMP(3);   // K loop done

    const int w_row0 = wm * (BM / 2), w_col0 = wn * (BN / 2);
    const bool kg_has_acc = KG == 1 || kgrp == 0;        // (the slab offsets of conv_tail.inc use `wave`: group 0's waves are 0..3)
#define BP_NT (256 * KG)
#define BP_SLAST (*s_last_p)
#define BP_EARLY_BIAS bias_early
#define BP_HAS_ACC kg_has_acc
#define BP_TAIL_STAMP(k_) BH_STAMP(k_)
#define BP_EP_RES_SCALE true            /* the SE blocks' downsample layers (both the four-wave tile and the K2 form run some of them) */
#define BP_EP_UP2                       /* the detector's two upsampling 1x1 layers */
#include "conv_tail.inc"
#undef BP_EP_UP2
#undef BP_EP_RES_SCALE
#undef BP_HAS_ACC
#undef BP_EARLY_BIAS
#undef BP_TAIL_STAMP
#undef BP_NT
#undef BP_SLAST
    if (p.stamps) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); BH_STAMP(4); }
#undef BH_STAMP
#undef BH_AS
#undef BH_BS
#undef BH_ADDR
#undef BH_LOAD
#undef BH_PARKED
#undef BH_LOAD_BD
#undef BH_STORE
#undef BH_PHASE
}

template <int TM, int TN, int NP, bool BD = false>
__global__ __launch_bounds__(256) void conv_igemm_h_kernel(const ConvParams p) {
    __shared__ __attribute__((aligned(16))) float smem[IgemmHLds<TM, TN, NP, BD>::FLOATS];
    __shared__ int s_last;
    conv_igemm_h_body<TM, TN, NP, BD>(p, (int)blockIdx.x, smem, &s_last);
}
// TILE_BD_K2: the filters-direct 64x64 tile with two K groups inside the block (eight waves)
__global__ __launch_bounds__(512) void conv_igemm_bdk2_kernel(const ConvParams p) {
    __shared__ __attribute__((aligned(16))) float smem[2 * IgemmHLds<1, 1, 3, true>::FLOATS];
    __shared__ int s_last;
    conv_igemm_h_body<1, 1, 3, true, ConvParams, 2>(p, (int)blockIdx.x, smem, &s_last);
}

thread_local ConvProfHook* g_conv_prof = nullptr;

// =====================================================================================================================
// The RGB stem of YOLOv3 (conv0: 416x416x3 -> 32, 3x3, stride 1, pad 1; yolo/darknet.py:240-259 with cfg/yolov3 layer 0) as
// a DIRECT convolution on the vector pipe.  As an implicit GEMM the layer is K = 27, N = 32 on tiles of K = 64 x N = 64 --
// 4.7x the multiplies, on the fp32 MFMA pipe (157 TFLOP/s): 24.6 us at batch 1 and 546 us at batch 28, the slowest launch
// of the batched runs at 12-15 TFLOP/s.  Here a lane owns one output pixel and 8 of the output channels (a wave = 64
// consecutive pixels, the waves of a block = the channel groups): nine 16-B loads of the packed RGBx input per lane
// (coalesced, zero outside the image), the filters broadcast from LDS, 216 FMAs per lane, then bias / activation / store
// (fp32 and / or the operand planes of the next layer).  Same sums as the GEMM in a fixed tap-major order.
// =====================================================================================================================
template <int CG, class P>   // channel groups of 8 = waves per block
__device__ __forceinline__ void stem3x3_body(const P& p, const int bp_bid, f32x4* const tile) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int cg = __builtin_amdgcn_readfirstlane(tid >> 6);
    // the filters of the wave's 8 channels are wave-uniform: scalar loads (p.w [CoutPad][Kpad], k = tap * 4 + ci), operands of
    // the FMAs straight from scalar registers
    const float* __restrict__ wrow = p.w + (long long)(cg * 8) * p.Kpad;
    // tile: 64 * (2 CG + 1) x 16 B of LDS -- the block's 64 pixels x 8 CG channels (+ 16 B per row against bank conflicts)
    const int m0 = bp_bid * 64;
    const int m = min(m0 + lane, p.M - 1);      // (rows past M compute a duplicate of the last pixel and are not stored)
    const int hw = p.OH * p.OW;
    const int b = m / hw, rem = m - b * hw;
    const int oy = rem / p.OW, ox = rem - oy * p.OW;
    float acc[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) acc[c] = 0.f;
    // a tap = one 16-B buffer load at the pixel (RGB frames are 12 B per pixel: the fourth lane is the neighbour's red and
    // meets no multiply); outside the image the offset is out of range: zeros
    const __amdgpu_buffer_rsrc_t rsrcA = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.in), 0, (int)min((long long)p.N * p.H * p.W * p.in_ld * 4, (long long)OOB), 0x00020000);
    // the CG waves of a block need the same nine taps of the same 64 pixels: each tap is fetched by ONE wave (wave cg takes taps cg,
    // cg + CG, ...) and shared through LDS (xs [tap][lane], behind the output tile) -- 9 instead of 9 CG 16-B requests per pixel
    f32x4* const xs = tile + 64 * (2 * CG + 1);
#pragma unroll 1
    for (int t = cg; t < 9; t += CG) {
        const int ky = (t * 11) >> 5, kx = t - 3 * ky;          // t / 3 for t < 9
        const int iy = oy + ky - 1, ix = ox + kx - 1;
        const bool in_img = (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
        xs[t * 64 + lane] = buf_load4(rsrcA, in_img ? (unsigned)((((b * p.H + iy) * p.W + ix) * p.in_ld) * 4) : OOB, 0);
    }
    __syncthreads();
    // a REAL loop over the taps (unrolled, hipcc loads all 72 filter quads up front whatever the source order: 288 registers --
    // scalar ones spilled lane by lane into vector registers, or 256 vector registers at one wave per SIMD; both measured
    // slower than the MFMA kernel).  Eight waves per SIMD cover the per-tap latency instead.
#pragma unroll 1
    for (int t = 0; t < 9; ++t) {
        const f32x4 x = xs[t * 64 + lane];
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const f32x4 w = *reinterpret_cast<const f32x4*>(wrow + c * p.Kpad + t * 4);
            acc[c] = fmaf(x.x, w.x, acc[c]);
            acc[c] = fmaf(x.y, w.y, acc[c]);
            acc[c] = fmaf(x.z, w.z, acc[c]);
            if (p.Cin == 4) acc[c] = fmaf(x.w, w.w, acc[c]);      // (the packed RGB frames carry three channels)
        }
    }
    // through LDS, so that every lane stores 16 contiguous bytes of a pixel's channel row (a wave covers whole 128-B lines)
    tile[lane * (2 * CG + 1) + 2 * cg] = f32x4{acc[0], acc[1], acc[2], acc[3]};
    tile[lane * (2 * CG + 1) + 2 * cg + 1] = f32x4{acc[4], acc[5], acc[6], acc[7]};
    __syncthreads();
    const PlaneDesc pd = make_plane_desc(p);
    const __amdgpu_buffer_rsrc_t rsrcO =
        __builtin_amdgcn_make_buffer_rsrc(p.out, 0, (int)min((long long)p.M * p.out_ld * 4, (long long)OOB), 0x00020000);
    const int q = tid % (2 * CG), n = 4 * q;     // this thread's four channels
    if (n < p.Cout) {
        const f32x4 bias = *reinterpret_cast<const f32x4*>(p.bias + n);
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
            const int r = pass * 32 + tid / (2 * CG);
            if (m0 + r >= p.M) break;
            f32x4 v = tile[r * (2 * CG + 1) + q];
            v += bias;
            if (p.act == ACT_LEAKY) {
                v.x = v.x > 0.f ? v.x : 0.1f * v.x; v.y = v.y > 0.f ? v.y : 0.1f * v.y;
                v.z = v.z > 0.f ? v.z : 0.1f * v.z; v.w = v.w > 0.f ? v.w : 0.1f * v.w;
            } else if (p.act == ACT_RELU) {
                v.x = v.x > 0.f ? v.x : 0.f; v.y = v.y > 0.f ? v.y : 0.f;
                v.z = v.z > 0.f ? v.z : 0.f; v.w = v.w > 0.f ? v.w : 0.f;
            }
            const unsigned off = (unsigned)(((m0 + r) * p.out_ld + n) * 4);
            const u32x4 o = {__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w)};
            if (pd.f32) __builtin_amdgcn_raw_buffer_store_b128(o, rsrcO, (int)off, 0, 0);
            emit_planes4(pd, v, off >> 1);
        }
    }
}

template <int CG>
__global__ __launch_bounds__(64 * CG) void stem3x3_kernel(const ConvParams p) {
    __shared__ f32x4 tile[64 * (2 * CG + 1) + 9 * 64];       // the output tile + the nine taps of the block's pixels
    stem3x3_body<CG>(p, (int)blockIdx.x, tile);
}

// =====================================================================================================================
// The same stem in the fp16 modes (round 5): on the matrix cores.  At 28 frames per launch the direct convolution above takes 296 us --
// 4 % of configs[2] -- with its waves issuing 22 % of their cycles (profiles/r04_pmc_wave_stalls_batch28_f16r.json), while the layer is
// 8.4 GFLOP and 310 MB of fp16 output: a 100 us pass.  Here a block owns 128 consecutive pixels: every tap is ONE 16-B load per pixel as
// before (the packed RGB frames: red, green, blue and the neighbour's red, which meets a zero filter entry), converted to four fp16 and
// parked as the pixel's im2col row in LDS -- 9 taps x 4 = 36 of 48 k, rows of 112 B (an odd multiple of 16 B: conflict-free fragment
// reads); each wave multiplies its 32 pixels by the 32 filters (fp16, from the packed fp32 filters, k = tap 4 + channel as they are stored)
// with three v_mfma_f32_32x32x16_f16; bias / activation, then 16-B stores of the fp16 plane (+ the fp32 tensor where it is still read)
// through an LDS tile.  fp16 operands, fp32 accumulation: the arithmetic of the mode (train_YOLO/src/convolutional_kernels.cu:268-280
// converts the first layer's activations too); the fp32-accurate modes keep the direct convolution.
// =====================================================================================================================
// A block walks G consecutive groups of 128 pixels (round 5; G = groups / grid, the launcher's choice): the taps of group g + 1 are requested before group g goes through LDS, the
// matrix cores and its stores.  One group per block was a chain of latencies -- kernel arguments, index math, the taps' round trip, three
// barriers, the stores' acknowledgement -- per 18 KB of traffic: 174 us for a 232 MB layer at 28 frames per launch.
__global__ __launch_bounds__(256) void stem3x3_f16_kernel(const ConvParams p) {
    constexpr int RB = 112, LDT = 36;                    // im2col row bytes; staging row floats
    __shared__ __attribute__((aligned(16))) char lds[128 * LDT * 4 > 128 * RB ? 128 * LDT * 4 : 128 * RB];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // the filters of this lane's channel (column lane & 31), k half lane >> 5 of the three k-steps: fp32 -> fp16 once per block
    f16x8 fb[3];
    {
        const float* w = p.w + (long long)(lane & 31) * p.Kpad + 8 * (lane >> 5);
#pragma unroll
        for (int ks = 0; ks < 3; ++ks) {
            const f32x4 lo = *reinterpret_cast<const f32x4*>(w + 16 * ks), hi = *reinterpret_cast<const f32x4*>(w + 16 * ks + 4);
            const f16x4 l = __builtin_convertvector(lo, f16x4), h = __builtin_convertvector(hi, f16x4);
            fb[ks] = __builtin_shufflevector(l, h, 0, 1, 2, 3, 4, 5, 6, 7);
        }
    }
    const f32x4 bias4 = *reinterpret_cast<const f32x4*>(p.bias + (tid & 1) * 16), bias4b = *reinterpret_cast<const f32x4*>(p.bias + (tid & 1) * 16 + 4),
                bias4c = *reinterpret_cast<const f32x4*>(p.bias + (tid & 1) * 16 + 8), bias4d = *reinterpret_cast<const f32x4*>(p.bias + (tid & 1) * 16 + 12);
    const __amdgpu_buffer_rsrc_t rsrcA = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.in), 0, (int)min((long long)p.N * p.H * p.W * p.in_ld * 4, (long long)OOB), 0x00020000);
    const int pix = tid >> 1;                            // thread -> pixel tid >> 1 of the group, taps of parity tid & 1
    const int hw = p.OH * p.OW;
    // the five taps of this thread for the group that starts at pixel m0 (rows past M: a duplicate of the last pixel, not stored)
    auto load_taps = [&](int m0, f32x4* x) __attribute__((always_inline)) {
        const int m = min(m0 + pix, p.M - 1);
        const int b = m / hw, rem = m - b * hw;
        const int oy = rem / p.OW, ox = rem - oy * p.OW;
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            const int t = 2 * i + (tid & 1);
            const int ky = (t * 11) >> 5, kx = t - 3 * ky;   // t / 3 for t < 9 (t = 9: never in the image test below)
            const int iy = oy + ky - 1, ix = ox + kx - 1;
            const bool ok = t < 9 && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
            x[i] = buf_load4(rsrcA, ok ? (unsigned)((((b * p.H + iy) * p.W + ix) * p.in_ld) * 4) : OOB, 0);
        }
    };
    const int groups = (p.M + 127) / 128, G = (groups + (int)gridDim.x - 1) / (int)gridDim.x;
    const int g0 = (int)blockIdx.x * G;
    f32x4 xn[5];
    load_taps(g0 * 128, xn);
    for (int g = 0; g < G; ++g) {
        const int m0 = (g0 + g) * 128;
        if (m0 >= p.M) break;                            // (block-uniform)
        f32x4 x[5];
#pragma unroll
        for (int i = 0; i < 5; ++i) x[i] = xn[i];
        if (g + 1 < G && m0 + 128 < p.M) load_taps(m0 + 128, xn);     // the next group's taps travel while this one is processed
        // ---- im2col rows
        {
            char* const row = lds + pix * RB;
#pragma unroll
            for (int i = 0; i < 5; ++i) {
                const int t = 2 * i + (tid & 1);
                f32x4 v = x[i];
                if (p.Cin < 4) v.w = 0.f;                    // (the neighbour's red: its filter entry is zero, but the last pixel of the tensor reads past it)
                if (t < 9) *reinterpret_cast<f16x4*>(row + 8 * t) = __builtin_convertvector(v, f16x4);
            }
            // k 36 .. 47 of the row: zeros (12 fp16 = 24 B; the odd-tap thread writes them)
            if (tid & 1) {
                *reinterpret_cast<u32x2*>(row + 72) = u32x2{0u, 0u};
                *reinterpret_cast<u32x4*>(row + 80) = u32x4{0u, 0u, 0u, 0u};
            }
        }
        __syncthreads();
        // ---- 32 pixels x 32 channels per wave
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        {
            const char* const arow = lds + (32 * wave + (lane & 31)) * RB + (lane >> 5) * 16;
#pragma unroll
            for (int ks = 0; ks < 3; ++ks)
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(*reinterpret_cast<const f16x8*>(arow + 32 * ks), fb[ks], acc, 0, 0, 0);
        }
        __syncthreads();                                     // the im2col rows are dead: the tile is staged over them
        float* const S = reinterpret_cast<float*>(lds);
#pragma unroll
        for (int r = 0; r < 16; ++r)
            S[(32 * wave + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * LDT + (lane & 31)] = acc[r];
        __syncthreads();
        // ---- stores: thread -> pixel tid >> 1, channels 16 (tid & 1) .. + 15
        const int m = m0 + pix, c0 = (tid & 1) * 16;
        if (m < p.M && c0 < p.Cout) {
            const float* srow = S + pix * LDT + c0;
            f32x4 v[4] = {*reinterpret_cast<const f32x4*>(srow) + bias4, *reinterpret_cast<const f32x4*>(srow + 4) + bias4b,
                          *reinterpret_cast<const f32x4*>(srow + 8) + bias4c, *reinterpret_cast<const f32x4*>(srow + 12) + bias4d};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (p.act == ACT_LEAKY) {
                    v[q].x = v[q].x > 0.f ? v[q].x : 0.1f * v[q].x; v[q].y = v[q].y > 0.f ? v[q].y : 0.1f * v[q].y;
                    v[q].z = v[q].z > 0.f ? v[q].z : 0.1f * v[q].z; v[q].w = v[q].w > 0.f ? v[q].w : 0.1f * v[q].w;
                } else if (p.act == ACT_RELU) {
                    v[q].x = v[q].x > 0.f ? v[q].x : 0.f; v[q].y = v[q].y > 0.f ? v[q].y : 0.f;
                    v[q].z = v[q].z > 0.f ? v[q].z : 0.f; v[q].w = v[q].w > 0.f ? v[q].w : 0.f;
                }
            }
            const long long e = (long long)m * p.out_ld + c0;
            if (!(p.out16 && p.skip_f32)) {
#pragma unroll
                for (int q = 0; q < 4; ++q) *reinterpret_cast<f32x4*>(p.out + e + 4 * q) = v[q];
            }
            if (p.out16) {                                   // (one fp16 plane: RNE of the fp32 value, as every producer writes it)
                const f16x4 h0 = __builtin_convertvector(v[0], f16x4), h1 = __builtin_convertvector(v[1], f16x4),
                            h2 = __builtin_convertvector(v[2], f16x4), h3 = __builtin_convertvector(v[3], f16x4);
                *reinterpret_cast<f16x8*>(p.out16 + e) = __builtin_shufflevector(h0, h1, 0, 1, 2, 3, 4, 5, 6, 7);
                *reinterpret_cast<f16x8*>(p.out16 + e + 8) = __builtin_shufflevector(h2, h3, 0, 1, 2, 3, 4, 5, 6, 7);
            }
        }
        __syncthreads();                                     // the staging tile is read: the next group's rows go over it
    }
}

// =====================================================================================================================
// The key-point detector's 7x7 / stride-2 / pad-3 RGB stem in the fp16 modes (round 6; KPD/src/models/layers/SE_Resnet.py:58-60 conv1 + bn1 +
// relu).  On the fp32 MFMA kernel the layer is COMPUTE-bound at 28 frames per launch -- 10.8 GFLOP at 68 TFLOP/s, 45 % of that pipe's roof:
// 159 us, 2.7 % of configs[2] -- while its bytes (28 MB of crops in, 147 MB of fp32 out) are a 30 us pass.  Same recipe as the detector's stem above:
// a block owns 64 consecutive output pixels per group, every tap is ONE 16-B load per pixel (the packed RGB crop: red, green, blue, a zero),
// converted to four fp16 and parked as the pixel's im2col row in LDS -- 49 taps x 4 = 196 of 224 k, rows of 464 B (an odd multiple of 16 B) --
// four threads per pixel, 12-13 taps each; each wave multiplies 32 pixels by 32 of the 64 filters (fp16 from the packed fp32 filters,
// k = tap 4 + channel as they are stored; 56 registers for the whole kernel) with fourteen v_mfma_f32_32x32x16_f16; bias / activation and 16-B
// stores through a staging tile.  A block walks G groups with the next group's taps in flight.  fp16 operands, fp32 accumulation: the
// arithmetic of the mode; the fp32-accurate modes keep the fp32 MFMA kernel.
// =====================================================================================================================
__global__ __launch_bounds__(256, 2) void stem7x7_f16_kernel(const ConvParams p) {
    constexpr int RB = 464, LDT = 68, GP = 64, NT = 13;     // im2col row bytes; staging row floats; pixels per group; taps per thread
    __shared__ __attribute__((aligned(16))) char rows[GP * RB];
    __shared__ __attribute__((aligned(16))) float S[GP * LDT];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wp = wave >> 1, wc = wave & 1;               // the wave's 32 pixels / 32 channels of the group
    // the filters of this lane's channel, k half lane >> 5 of the fourteen k-steps: fp32 -> fp16 once per block
    f16x8 fb[14];
    {
        const float* w = p.w + (long long)(32 * wc + (lane & 31)) * p.Kpad + 8 * (lane >> 5);
#pragma unroll
        for (int ks = 0; ks < 14; ++ks) {
            const f32x4 lo = *reinterpret_cast<const f32x4*>(w + 16 * ks), hi = *reinterpret_cast<const f32x4*>(w + 16 * ks + 4);
            const f16x4 l = __builtin_convertvector(lo, f16x4), h = __builtin_convertvector(hi, f16x4);
            fb[ks] = __builtin_shufflevector(l, h, 0, 1, 2, 3, 4, 5, 6, 7);
        }
    }
    const int pix = tid >> 2, tq = tid & 3;                // thread -> pixel tid >> 2 of the group, taps tq, tq + 4, ...
    const int c0 = tq * 16;                                // ... and, in the stores, channels 16 tq .. + 15
    const f32x4 b4a = *reinterpret_cast<const f32x4*>(p.bias + c0), b4b = *reinterpret_cast<const f32x4*>(p.bias + c0 + 4),
                b4c = *reinterpret_cast<const f32x4*>(p.bias + c0 + 8), b4d = *reinterpret_cast<const f32x4*>(p.bias + c0 + 12);
    const __amdgpu_buffer_rsrc_t rsrcA = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.in), 0, (int)min((long long)p.N * p.H * p.W * p.in_ld * 4, (long long)OOB), 0x00020000);
    const int hw = p.OH * p.OW;
    const float rcp_hw = 1.0f / (float)hw, rcp_ow = 1.0f / (float)p.OW;
    // k 196 .. 223 of every row: zeros, once (the taps never write there; the filters are zero there, but 0 x garbage may be NaN)
    if (tq == 0) {
        *reinterpret_cast<u32x2*>(rows + pix * RB + 392) = u32x2{0u, 0u};
        *reinterpret_cast<u32x4*>(rows + pix * RB + 400) = u32x4{0u, 0u, 0u, 0u};
        *reinterpret_cast<u32x4*>(rows + pix * RB + 416) = u32x4{0u, 0u, 0u, 0u};
        *reinterpret_cast<u32x4*>(rows + pix * RB + 432) = u32x4{0u, 0u, 0u, 0u};
    }
    auto load_taps = [&](int m0, f32x4* x) __attribute__((always_inline)) {
        const int m = min(m0 + pix, p.M - 1);
        const int b = fast_div(m, hw, rcp_hw), rem = m - b * hw;
        const int oy = fast_div(rem, p.OW, rcp_ow), ox = rem - oy * p.OW;
        const int base = (b * p.H + 2 * oy - 3) * p.W + 2 * ox - 3;
#pragma unroll
        for (int i = 0; i < NT; ++i) {
            const int t = 4 * i + tq;
            const int ky = (t * 37) >> 8, kx = t - 7 * ky;      // t / 7 for t < 49
            const int iy = 2 * oy - 3 + ky, ix = 2 * ox - 3 + kx;
            const bool ok = (t < 49) & ((unsigned)iy < (unsigned)p.H) & ((unsigned)ix < (unsigned)p.W);
            const unsigned off = (unsigned)((base + ky * p.W + kx) * p.in_ld * 4);
            // one 16-B load per tap also where a pixel is three floats (the engine's crop tensor: 12-B strides, dword-aligned -- 88 us against
            // 144 us with 12-B loads at 28 frames); the fourth dword is the neighbour's red and is dropped below; behind the LAST pixel of the
            // tensor it is out of the descriptor's range, which a raw buffer load checks per dword: it reads as zero, nothing is touched
            x[i] = buf_load4(rsrcA, ok ? off : OOB, 0);
        }
    };
    const int groups = (p.M + GP - 1) / GP, G = (groups + (int)gridDim.x - 1) / (int)gridDim.x;
    const int g0 = (int)blockIdx.x * G;
    f32x4 xn[NT];
    load_taps(g0 * GP, xn);
    for (int g = 0; g < G; ++g) {
        const int m0 = (g0 + g) * GP;
        if (m0 >= p.M) break;                            // (block-uniform)
        f32x4 x[NT];
#pragma unroll
        for (int i = 0; i < NT; ++i) x[i] = xn[i];
        if (g + 1 < G && m0 + GP < p.M) load_taps(m0 + GP, xn);     // the next group's taps travel while this one is processed
        {
            char* const row = rows + pix * RB;
#pragma unroll
            for (int i = 0; i < NT; ++i) {
                const int t = 4 * i + tq;
                f32x4 v = x[i];
                if (p.Cin < 4) v.w = 0.f;                    // (the neighbour pixel's red: its filter entry is zero, but keep the product finite)
                if (t < 49) *reinterpret_cast<f16x4*>(row + 8 * t) = __builtin_convertvector(v, f16x4);
            }
        }
        __syncthreads();
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        {
            const char* const arow = rows + (32 * wp + (lane & 31)) * RB + (lane >> 5) * 16;
#pragma unroll
            for (int ks = 0; ks < 14; ++ks)
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(*reinterpret_cast<const f16x8*>(arow + 32 * ks), fb[ks], acc, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r)
            S[(32 * wp + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * LDT + 32 * wc + (lane & 31)] = acc[r];
        __syncthreads();                                     // the tile is staged; the im2col rows are free for the next group
        const int m = m0 + pix;
        if (m < p.M && c0 < p.Cout) {
            const float* srow = S + pix * LDT + c0;
            f32x4 v[4] = {*reinterpret_cast<const f32x4*>(srow) + b4a, *reinterpret_cast<const f32x4*>(srow + 4) + b4b,
                          *reinterpret_cast<const f32x4*>(srow + 8) + b4c, *reinterpret_cast<const f32x4*>(srow + 12) + b4d};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (p.act == ACT_LEAKY) {
                    v[q].x = v[q].x > 0.f ? v[q].x : 0.1f * v[q].x; v[q].y = v[q].y > 0.f ? v[q].y : 0.1f * v[q].y;
                    v[q].z = v[q].z > 0.f ? v[q].z : 0.1f * v[q].z; v[q].w = v[q].w > 0.f ? v[q].w : 0.1f * v[q].w;
                } else if (p.act == ACT_RELU) {
                    v[q].x = v[q].x > 0.f ? v[q].x : 0.f; v[q].y = v[q].y > 0.f ? v[q].y : 0.f;
                    v[q].z = v[q].z > 0.f ? v[q].z : 0.f; v[q].w = v[q].w > 0.f ? v[q].w : 0.f;
                }
            }
            const long long e = (long long)m * p.out_ld + c0;
            if (!(p.out16 && p.skip_f32)) {
#pragma unroll
                for (int q = 0; q < 4; ++q) *reinterpret_cast<f32x4*>(p.out + e + 4 * q) = v[q];
            }
            if (p.out16 && p.out_np == 1) {
                const f16x4 h0 = __builtin_convertvector(v[0], f16x4), h1 = __builtin_convertvector(v[1], f16x4),
                            h2 = __builtin_convertvector(v[2], f16x4), h3 = __builtin_convertvector(v[3], f16x4);
                *reinterpret_cast<f16x8*>(p.out16 + e) = __builtin_shufflevector(h0, h1, 0, 1, 2, 3, 4, 5, 6, 7);
                *reinterpret_cast<f16x8*>(p.out16 + e + 8) = __builtin_shufflevector(h2, h3, 0, 1, 2, 3, 4, 5, 6, 7);
            }
        }
        // (the next group's staging writes sit behind its own first barrier, after every thread's reads of S above)
    }
}

bool conv_stem7_eligible(const ConvParams& p) {
    return p.cin_pack == 4 && p.Cin <= 4 && (p.in_ld == 4 || (p.in_ld == 3 && p.Cin == 3)) && p.ksize == 7 && p.stride == 2 && p.pad == 3 && p.Cout == 64 && p.CoutPad == 64 &&
           p.Kpad == 224 && p.store_mode == ST_NHWC && p.res == nullptr && p.res_scale == nullptr && p.pool_out == nullptr && (p.out_ld & 7) == 0 &&
           (p.out16 == nullptr || p.out_np == 1) && (reinterpret_cast<uintptr_t>(p.out) & 15) == 0 && (reinterpret_cast<uintptr_t>(p.out16) & 15) == 0 &&
           (long long)p.M * p.out_ld * 4 < (long long)OOB && (long long)p.N * p.H * p.W * 16 < (long long)OOB && p.M < (1 << 24);
}

static void launch_stem7(const ConvParams& p, hipStream_t s) {
    BP_CHECK(conv_stem7_eligible(p) && p.splits == 1, "not the 7x7 / stride-2 / packed-RGB stem");
    // groups of 64 pixels per block: 8 (BP_STEM7_G for the sweep), fewer where that would leave less than two blocks per CU
    static const int g_env = std::getenv("BP_STEM7_G") ? std::atoi(std::getenv("BP_STEM7_G")) : 0;
    const int groups = (p.M + 63) / 64;
    int per = g_env > 0 ? g_env : 8;
    while (per > 1 && groups / per < 512) --per;
    const dim3 g((groups + per - 1) / per);
    if (g_conv_prof) hipExtLaunchKernelGGL(stem7x7_f16_kernel, g, dim3(256), 0, s, g_conv_prof->e0, g_conv_prof->e1, 0, p);
    else hipLaunchKernelGGL(stem7x7_f16_kernel, g, dim3(256), 0, s, p);
}

// the fp16 modes' stem: 32 output channels, an fp16 plane wanted by the next layer (BP_NO_STEM_F16=1: the direct convolution, A/B runs)
static bool stem3_f16_wanted(const ConvParams& p) {
    static const bool off = std::getenv("BP_NO_STEM_F16") != nullptr;
    return !off && p.out16 != nullptr && p.out_np == 1 && p.Cout == 32 && p.Kpad >= 48 && p.out_ld % 8 == 0 &&
           (reinterpret_cast<uintptr_t>(p.out16) & 15) == 0 && (reinterpret_cast<uintptr_t>(p.out) & 15) == 0;
}

bool conv_stem3_eligible(const ConvParams& p) {
    return p.cin_pack == 4 && p.Cin <= 4 && p.in_ld >= p.Cin && p.in_ld <= 4 && p.ksize == 3 && p.stride == 1 && p.pad == 1 && p.OH == p.H && p.OW == p.W &&
           p.Cout % 4 == 0 && p.Cout <= 64 && p.CoutPad >= ((p.Cout + 7) / 8) * 8 && p.Kpad >= 36 && p.store_mode == ST_NHWC &&
           p.res == nullptr && p.res_scale == nullptr && (p.out_ld & 3) == 0 && p.pool_out == nullptr &&
           (long long)p.M * p.out_ld * 4 < (long long)OOB;
}

static void launch_stem3(const ConvParams& p, hipStream_t s) {
    BP_CHECK(conv_stem3_eligible(p) && p.splits == 1, "not a 3x3 / stride-1 / packed-RGB stem");
    if (stem3_f16_wanted(p)) {
        // groups per block: 8 (one launch at a time, 28 frames per launch: 190 us with one group per block, 142 / 126 / 116 / 121 with 2 / 4 / 8 / 16;
        // BP_STEM_G for the sweep), fewer where that would leave less than two blocks per CU (one frame: 2 groups, 9.4 us)
        static const int g_env = std::getenv("BP_STEM_G") ? std::atoi(std::getenv("BP_STEM_G")) : 0;
        const int groups = (p.M + 127) / 128;
        int per = g_env > 0 ? g_env : 8;
        while (per > 1 && groups / per < 512) --per;
        const dim3 g((groups + per - 1) / per);
        if (g_conv_prof) hipExtLaunchKernelGGL(stem3x3_f16_kernel, g, dim3(256), 0, s, g_conv_prof->e0, g_conv_prof->e1, 0, p);
        else hipLaunchKernelGGL(stem3x3_f16_kernel, g, dim3(256), 0, s, p);
        return;
    }
    const int cg = (p.Cout + 7) / 8;
    const dim3 grid((p.M + 63) / 64);
#define BP_STEM(CG_)                                                                                                       \
    if (g_conv_prof) hipExtLaunchKernelGGL((stem3x3_kernel<CG_>), grid, dim3(64 * CG_), 0, s, g_conv_prof->e0, g_conv_prof->e1, 0, p); \
    else hipLaunchKernelGGL((stem3x3_kernel<CG_>), grid, dim3(64 * CG_), 0, s, p)
    switch (cg) {
        case 1: BP_STEM(1); break;
        case 2: BP_STEM(2); break;
        case 4: BP_STEM(4); break;
        case 8: BP_STEM(8); break;
        default: throw Error("stem kernel: 8, 16, 32 or 64 output channels");
    }
#undef BP_STEM
}

// the 1-D launch grid of the kernels that take ConvParams::xcd_home / pf_*: [work blocks | padding to 8 | prefetch blocks]
void conv_grid_setup(ConvParams& q, int bm, int bn) {
    q.mtiles = (q.M + bm - 1) / bm;
    const int ntn = (q.CoutPad + bn - 1) / bn;
    q.n_tiles = q.mtiles * ntn;
    q.work_blocks = q.hy_splits > 0 ? q.hy_full + (q.n_tiles - q.hy_full) * q.hy_splits
                   : q.xcd_home ? ((q.n_tiles + 7) / 8) * 8 * q.splits : q.n_tiles * q.splits;
    q.pf_first = (q.work_blocks + 7) & ~7;
}
int conv_grid_blocks(const ConvParams& q) {
    if (!q.pf_ptr) return q.work_blocks;
    const int g = q.pf_ntn < 8 ? q.pf_ntn : 8;
    return q.pf_first + 8 * (q.pf_ntn / g) * q.pf_splits;   // one prefetch block per pair and XCD that reads it
}

int conv_tile_bm(int tile) {
    switch (tile) {
        case TILE_128x64: case TILE_W64_2x1: case TILE_W64_2x2: case TILE_PL128: case TILE_PL128x64: case TILE_PL128S: case TILE_PLH128: case TILE_P3: return 128;
        case TILE_PL256x128: return 256;
        case TILE_S1: return 32;
        default: return 64;
    }
}
int conv_tile_bn(int tile) {
    switch (tile) {
        case TILE_W64_1x2: case TILE_W64_2x2: case TILE_PL128: case TILE_PL256x128: case TILE_PL128S: case TILE_HALO128: case TILE_PLH128: case TILE_S1: case TILE_P3: return 128;
        default: return 64;
    }
}

template <int TM, int TN>
static void launch_t(const ConvParams& p, hipStream_t s) {
    constexpr int BM = 64 * TM, BN = 64 * TN;
    dim3 grid(((p.M + BM - 1) / BM) * (p.CoutPad / BN) * p.splits);
    // vector paths: whole 32-channel chunks inside one filter tap (taps addressable by a 64-bit mask), or filters
    // packed 4 channels per tap for the RGB stems; anything else gathers scalars
    const int vec = conv_vec_mode(p);
    if (g_conv_prof) {
        // hipExtLaunchKernelGGL stamps the events with the kernel's own begin/end (no host-side event gap)
        if (vec == 1)
            hipExtLaunchKernelGGL((conv_igemm_kernel<TM, TN, 1>), grid, dim3(256), 0, s, g_conv_prof->e0, g_conv_prof->e1, 0, p);
        else if (vec == 2)
            hipExtLaunchKernelGGL((conv_igemm_kernel<TM, TN, 2>), grid, dim3(256), 0, s, g_conv_prof->e0, g_conv_prof->e1, 0, p);
        else
            hipExtLaunchKernelGGL((conv_igemm_kernel<TM, TN, 0>), grid, dim3(256), 0, s, g_conv_prof->e0, g_conv_prof->e1, 0, p);
    } else if (vec == 1) {
        hipLaunchKernelGGL((conv_igemm_kernel<TM, TN, 1>), grid, dim3(256), 0, s, p);
    } else if (vec == 2) {
        hipLaunchKernelGGL((conv_igemm_kernel<TM, TN, 2>), grid, dim3(256), 0, s, p);
    } else {
        hipLaunchKernelGGL((conv_igemm_kernel<TM, TN, 0>), grid, dim3(256), 0, s, p);
    }
}

template <int TM, int TN, int NP, bool BD = false>
static void launch_h_t(const ConvParams& p, hipStream_t s) {
    constexpr int BM = 64 * TM, BN = 64 * TN;
    ConvParams q = p;
    conv_grid_setup(q, BM, BN);
    dim3 grid(conv_grid_blocks(q));
    if (g_conv_prof)
        hipExtLaunchKernelGGL((conv_igemm_h_kernel<TM, TN, NP, BD>), grid, dim3(256), 0, s, g_conv_prof->e0, g_conv_prof->e1, 0, q);
    else
        hipLaunchKernelGGL((conv_igemm_h_kernel<TM, TN, NP, BD>), grid, dim3(256), 0, s, q);
}

int conv_vec_mode(const ConvParams& p) {
    if ((p.Cin % 32 == 0) && (p.in_ld % 4 == 0) && p.ksize <= 8) return 1;
    if (p.cin_pack == 4 && p.Cin <= 4 && p.ksize <= 8) return 2;
    return 0;
}

bool conv_h16_eligible(const ConvParams& p) {   // the fp32-activation 16-bit kernels: a converted filter copy + whole 32-channel chunks
    return (p.w16 != nullptr || p.w16s != nullptr) && (p.Cin % 32 == 0) && (p.in_ld % 4 == 0) && p.ksize <= 8;
}

void conv_split_plan(const ConvParams& p, int tile, int want, int* splits, int* cps) {
    const int unit = ((conv_tile_is_halo(tile) || conv_tile_is_plh(tile)) && p.nchunks % 9 == 0 && p.nchunks >= 9) ? 9 : 1;          // chunks that stay together (a layer the halo tiles cannot run is refused by the launcher)
    const int units = p.nchunks / unit;
    int s = want < 1 ? 1 : (want > units ? units : want);
    const int per = (units + s - 1) / s;
    s = (units + per - 1) / per;
    *splits = s; *cps = per * unit;
}

int conv_tiles(const ConvParams& p, int tile) {
    const int bm = conv_tile_bm(tile), bn = conv_tile_bn(tile);
    return ((p.M + bm - 1) / bm) * ((p.CoutPad + bn - 1) / bn);
}

void launch_conv(const ConvParams& p, int tile, hipStream_t s) {
    BP_CHECK(p.CoutPad % 64 == 0, "CoutPad must be a multiple of 64");
    BP_CHECK(p.Kpad % BK == 0 && p.nchunks == p.Kpad / BK, "Kpad");
    BP_CHECK(p.splits >= 1 && (p.splits == 1 || (p.partial != nullptr && p.tickets != nullptr)), "split-K workspace");
    BP_CHECK((long long)p.N * p.H * p.W * p.in_ld * 4 < (long long)OOB, "activation tensor too large for 32-bit offsets");
    BP_CHECK((long long)p.M * p.out_ld * 4 < (long long)OOB && (p.res == nullptr || (long long)p.M * p.res_ld * 4 < (long long)OOB),
             "output / residual tensor too large for 32-bit offsets");
    if (tile == TILE_STEM3) {
        launch_stem3(p, s);
    } else if (tile == TILE_STEM7) {
        launch_stem7(p, s);
    } else if (conv_tile_is_pl(tile)) {
        launch_conv_pl(p, tile, s);
    } else if (conv_tile_is_halo(tile)) {
        launch_conv_halo(p, tile, s);
    } else if (tile == TILE_64x64_BD && p.mfma_mode == PREC_BF16X3) {
        BP_CHECK(conv_h16_eligible(p) && p.w16s != nullptr, "filters-direct tile needs the stage-packed filter copy and Cin % 32 == 0");
        BP_CHECK((long long)3 * p.CoutPad * p.Kpad * 2 < (long long)OOB, "filter planes too large for 32-bit offsets");
        launch_h_t<1, 1, 3, true>(p, s);
    } else if (tile == TILE_BD_K2) {
        BP_CHECK(p.mfma_mode == PREC_BF16X3 && conv_h16_eligible(p) && p.w16s != nullptr, "filters-direct K2 tile: bf16x3 mode, stage-packed filter copy, Cin % 32 == 0");
        BP_CHECK((long long)3 * p.CoutPad * p.Kpad * 2 < (long long)OOB, "filter planes too large for 32-bit offsets");
        BP_CHECK(!p.xcd_home && !p.pf_ptr, "filters-direct K2 tile: no latency-mode layouts");
        ConvParams q = p;
        conv_grid_setup(q, 64, 64);
        if (g_conv_prof) hipExtLaunchKernelGGL(conv_igemm_bdk2_kernel, dim3(conv_grid_blocks(q)), dim3(512), 0, s, g_conv_prof->e0, g_conv_prof->e1, 0, q);
        else hipLaunchKernelGGL(conv_igemm_bdk2_kernel, dim3(conv_grid_blocks(q)), dim3(512), 0, s, q);
#ifdef BP_EXPERIMENTAL
    } else if (conv_tile_is_w64(tile)) {
        launch_conv_w64(p, tile, s);
    } else if (conv_tile_is_kg(tile)) {
        launch_conv_kg(p, tile, s);
    } else if (conv_tile_is_rd(tile)) {
        launch_conv_rd(p, tile, s);
    } else if (p.mfma_mode == PREC_F16 && conv_h16_eligible(p)) {
        switch (tile) {
            case TILE_64x64_BD:
                BP_CHECK(p.w16s != nullptr, "filters-direct tile needs the stage-packed filter copy");
                launch_h_t<1, 1, 1, true>(p, s);
                break;
            case TILE_128x64: launch_h_t<2, 1, 1>(p, s); break;
            default: launch_h_t<1, 1, 1>(p, s); break;
        }
    } else if (p.mfma_mode == PREC_BF16X3 && conv_h16_eligible(p)) {
        BP_CHECK(tile == TILE_64x64 && p.w16 != nullptr, "the LDS-staged bf16x3 kernel is built for the 64x64 tile");
        launch_h_t<1, 1, 3>(p, s);
#endif
    } else {
        BP_CHECK(tile == TILE_64x64 || tile == TILE_128x64,
                 "this kernel id exists only in the experimental library (python -m betapose_amd.build --experimental, BP_LIB)");
        switch (tile) {
            case TILE_128x64: launch_t<2, 1>(p, s); break;
            default: launch_t<1, 1>(p, s); break;
        }
    }
    BP_HIP(hipGetLastError());
}

}  // namespace bp
