// Implicit-GEMM convolution on the bf16 matrix pipe with NO shared-memory stage: operands go global -> registers -> MFMA
// (gfx950).  Same contract as the other conv kernels (NHWC fp32 activations, BN-folded filters as pre-split bf16 planes,
// fp32-accurate six-product arithmetic, fused epilogue; replaces cudnnConvolutionForward + the bias / activation /
// shortcut kernels of the reference's Darknet CUDA backend, train_YOLO/src/convolutional_kernels.cu:121-383, and the
// torch Conv2d/BatchNorm2d/ReLU modules of yolo/darknet.py:240-259 and KPD/src/models/layers/SE_Resnet.py:25-42).
//
// Why: the LDS-staged kernels are LDS-bound (conv_igemm.hip / conv_kg.hip: 72 KB through LDS per 64x64x32 chunk, the
// matrix pipe idle two thirds of the K loop), and what LDS buys them is only the redistribution of a 64x64 block tile
// over four waves.  Here ONE wave owns the whole 64x64 tile (2 x 2 MFMA tiles, each operand fragment feeds two
// instructions), so nothing needs redistributing:
//  * activations: lane l fetches the 8 consecutive k of row (l & 31) it supplies to the MFMA (two 16-B loads per
//    32-row tile), splits them exactly into three bf16 terms in registers -- that IS the A fragment;
//  * filters: from the stage-packed copy ConvParams::w16s (one contiguous 6 KB block per 64-row tile and 16-k stage);
//    a fragment is one fully coalesced 1 KB wave load;
//  * no LDS, no barrier, no DMA in the K loop; L1/L2 traffic per MFMA is the same as in the staged kernels (both move
//    one 64x64 tile's operands per stage);
//  * a block is W waves on W consecutive K ranges of the same tile (the fill a batch-1 layer needs comes from cutting
//    K); their partial sums meet in LDS once, at the end, in wave order: no slab in HBM, no ticket, one epilogue per tile.
//    Cross-block slices (p.splits) remain for the layers with too few tiles even so (shared tail, conv_tail.inc).
#include "conv_dev.h"

namespace bp {

// The K loop's loads are issued from inline asm and waited for by hand: left to hipcc, every wait at the loop header
// degenerates to "everything but the loads of this stage" (its waitcnt pass merges the loop-carried state
// conservatively), which puts the full L2 latency of the filter fragments requested a few slots earlier on the critical
// path of every other stage.  A load's destination registers are tied ("+v") into the s_waitcnt that covers it, so no
// use can be scheduled above the wait.  (asm with register constraints lives in __device__ functions: directly in a
// __global__ template it silently drops the kernel's host stub.)
typedef int i32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ i32x4 rd_rsrc(const void* base, unsigned bytes) {
    const unsigned long long a = (unsigned long long)base;
    return i32x4{(int)(unsigned)a, (int)((a >> 32) & 0xffffu), (int)bytes, 0x00020000};
}
template <int OFF, class V>
__device__ __forceinline__ void rd_load16(V& out, i32x4 rs, unsigned voff, int soff) {
    asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen offset:%4" : "=&v"(out) : "v"(voff), "s"(rs), "s"(soff), "n"(OFF));
}
template <int N>
__device__ __forceinline__ void rd_wait4(f32x4& a, f32x4& b, f32x4& c, f32x4& d) {
    asm volatile("s_waitcnt vmcnt(%4)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "n"(N));
}
template <int N>
__device__ __forceinline__ void rd_wait6(u32x4& a, u32x4& b, u32x4& c, u32x4& d, u32x4& e, u32x4& f) {
    asm volatile("s_waitcnt vmcnt(%6)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f) : "n"(N));
}

template <int W>
__global__ __launch_bounds__(64 * W) __attribute__((amdgpu_waves_per_eu(2, 2))) void conv_rd_kernel(const ConvParams p) {
    constexpr int NP = 3;
    constexpr int BM = 64, BN = 64, TM = 2, TN = 2;
    constexpr int LDT = BN + 4;
    constexpr int KS = 16;
    constexpr int B_SLOT = NP * BN * KS * 2;              // bytes of one stage of one 64-row filter tile in w16s
    constexpr int EPI_BYTES = BM * LDT * 4;
    constexpr int SLAB = BM * BN * 4;                     // one wave's accumulators
    constexpr int SMEM_BYTES = ((W - 1) * SLAB > EPI_BYTES ? (W - 1) * SLAB : EPI_BYTES) + 16;
    __shared__ __attribute__((aligned(16))) float smem[SMEM_BYTES / 4];
    typedef bf16x8 frag_t;

    const unsigned long long t_entry = p.stamps ? __builtin_readcyclecounter() : 0ull;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave_k = __builtin_amdgcn_readfirstlane(tid >> 6);      // K range of this wave
#define RD_STAMP(k_) if (p.stamps && tid == 0) p.stamps[(long long)blockIdx.x * 8 + (k_)] = __builtin_readcyclecounter();
    if (p.stamps && tid == 0) p.stamps[(long long)blockIdx.x * 8 + 0] = t_entry;
    const int n_tiles_n = p.CoutPad / BN;
    const int split = (int)blockIdx.x % p.splits;
    const int tile_id = (int)blockIdx.x / p.splits;
    const int tile_n = tile_id % n_tiles_n;
    const int tile_m = tile_id / n_tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int c_begin = split * p.chunks_per_split;                    // the block's K range, in 32-k chunks
    const int c_end = min(p.nchunks, c_begin + p.chunks_per_split);
    const int ns = 2 * (c_end - c_begin);                              // ... in 16-k stages
    const int nsw = (ns + W - 1) / W;
    const int s_begin = 2 * c_begin + wave_k * nsw;
    const int s_end = min(2 * c_end, s_begin + nsw);

    const i32x4 rsrcA = rd_rsrc(p.in, (unsigned)min((long long)p.N * p.H * p.W * p.in_ld * 4, (long long)OOB));
    const i32x4 rsrcB = rd_rsrc(p.w16s, (unsigned)(NP * p.CoutPad * p.Kpad * 2));

    // ---- activations: lane -> rows m0 + 32 i + (lane & 31), k half (lane >> 5) of every stage
    unsigned a_base[TM];
    unsigned long long a_mask[TM];
    {
        const int hw = p.OH * p.OW;
        const float rcp_hw = 1.0f / (float)hw, rcp_ow = 1.0f / (float)p.OW;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int m = m0 + 32 * i + (lane & 31);
            const bool ok = m < p.M;
            const int mm = ok ? m : 0;
            const int b = fast_div(mm, hw, rcp_hw);
            const int rem = mm - b * hw;
            const int oy = fast_div(rem, p.OW, rcp_ow);
            const int ox = rem - oy * p.OW;
            const int iy0 = oy * p.stride - p.pad, ix0 = ox * p.stride - p.pad;
            a_base[i] = (unsigned)((((b * p.H + iy0) * p.W + ix0) * p.in_ld + (lane >> 5) * 8) * 4);
            const int kx_lo = max(0, -ix0), kx_hi = min(p.ksize, p.W - ix0);
            const int ky_lo = max(0, -iy0), ky_hi = min(p.ksize, p.H - iy0);
            unsigned long long mask = 0;
            if (ok && kx_hi > kx_lo) {
                const unsigned long long rowbits = ((1ull << kx_hi) - 1ull) & ~((1ull << kx_lo) - 1ull);
                for (int ky = ky_lo; ky < ky_hi; ++ky) mask |= rowbits << (ky * p.ksize);
            }
            a_mask[i] = mask;
        }
    }
    // wave-uniform walk over K in 16-k stages: stage -> (tap, ky, kx, ci0)
    const int spt = p.Cin >> 4;
    int w_s = s_begin;
    int w_tap = w_s / spt;
    int w_ci = (w_s - w_tap * spt) << 4;
    int w_ky = w_tap / p.ksize;
    int w_kx = w_tap - w_ky * p.ksize;
    // raw activations of the stage the walk points at -> ra_[2 i], ra_[2 i + 1]; past the wave's range the offsets are
    // out of range (zeros, no memory traffic), so the loop body needs no conditionals
#define RD_FETCH_A(ra_)                                                                                \
    {                                                                                                  \
        const unsigned delta = (unsigned)(((w_ky * p.W + w_kx) * p.in_ld + w_ci) * 4);                \
        const bool in_ = w_s < s_end;                                                                  \
        _Pragma("unroll") for (int i = 0; i < TM; ++i) {                                               \
            const bool ok = ((a_mask[i] >> (w_tap & 63)) & 1ull) && in_;                               \
            const unsigned va = ok ? a_base[i] + delta : OOB;                                          \
            rd_load16<0>(ra_[2 * i], rsrcA, va, 0);                                                    \
            rd_load16<16>(ra_[2 * i + 1], rsrcA, va, 0);                                               \
        }                                                                                              \
        ++w_s;                                                                                         \
        const int wrap = (w_ci + KS == p.Cin) ? 1 : 0;                                                 \
        w_ci = wrap ? 0 : w_ci + KS;                                                                   \
        w_tap += wrap;                                                                                 \
        const int kx1 = w_kx + wrap;                                                                   \
        const int wrap2 = (kx1 == p.ksize) ? 1 : 0;                                                    \
        w_kx = wrap2 ? 0 : kx1;                                                                        \
        w_ky += wrap2;                                                                                 \
    }
    // ---- filters: fragment (plane, 32-column tile j) of stage s = 1 KB at b_tile + s * B_SLOT + plane * 2 KB, lane ->
    // row 32 j + (lane & 31), granule (lane >> 5) at slot granule ^ ((row >> 3) & 1) (w16s is the LDS image of conv_kg.hip)
    const int b_tile = tile_n * (p.Kpad >> 4) * B_SLOT;
    unsigned b_voff[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j)
        b_voff[j] = (unsigned)((32 * j + (lane & 31)) * 32 + (((lane >> 5) ^ ((lane >> 3) & 1)) << 4));
    int b_s = s_begin;
    // one filter fragment of the stage whose source offset is so_ (computed once per stage by RD_B_OFF)
#define RD_B_OFF() const int so_ = b_s < s_end ? b_tile + b_s * B_SLOT : (int)OOB; ++b_s;
#define RD_FETCH_B1(rb_, pl_, j_) rd_load16<0>(rb_[pl_][j_], rsrcB, b_voff[j_], so_ + (pl_) * 2048);
#define RD_FETCH_B(rb_)                                                                                \
    {                                                                                                  \
        RD_B_OFF();                                                                                    \
        _Pragma("unroll") for (int pl = 0; pl < NP; ++pl)                                              \
            _Pragma("unroll") for (int j = 0; j < TN; ++j) RD_FETCH_B1(rb_, pl, j);                    \
    }
    // ---- exact three-way split of the 16 floats a lane holds per stage (8 per row tile): x = h1 + h2 + h3.  Two pairs
    // of floats (a "group") are split in lockstep, and every sub-stage (plane term + residual) is cut into three micro-ops
    // -- 2 conversions / 4 bit operations / 4 subtractions -- that go into DIFFERENT MFMA slots: an in-order wave sits
    // out the latency of every dependent VALU pair, so a dependent chain issued in one piece delays the next MFMA
    // (conv_w64.hip measured it).  7 micro-ops per group (3 + 3 + the plane-3 conversion); pair pr is 32-bit element
    // pr & 3 of the fragment of row tile pr >> 2.
    constexpr int NPAIR = 8, NGRP = 4, MPG = 7, NMU = NGRP * MPG;
    float xr[NPAIR][2];                           // running residuals
    unsigned tu[2], tb[2][2];                     // a group's conversions and their two halves as fp32 bit patterns
#define RD_MICRO(fn_, ra, k_)                                                                          \
    {                                                                                                  \
        const int gi = (k_) / MPG, l = (k_) % MPG;          /* constants once the caller's loop is unrolled */ \
        const int sg = l == 6 ? 2 : l / 3, ph = l == 6 ? 0 : l % 3;                                    \
        _Pragma("unroll") for (int e = 0; e < 2; ++e) {                                                \
            const int pr = 2 * gi + e;                                                                 \
            if (ph == 0) {                                                                             \
                if (sg == 0) { xr[pr][0] = ra[pr / 2][(pr % 2) * 2]; xr[pr][1] = ra[pr / 2][(pr % 2) * 2 + 1]; } \
                const bf16x2 h = __builtin_convertvector(f32x2{xr[pr][0], xr[pr][1]}, bf16x2);         \
                tu[e] = __builtin_bit_cast(unsigned, h);                                               \
                fn_[sg][pr >> 2][pr & 3] = tu[e];                                                      \
            } else if (ph == 1) {                                                                      \
                tb[e][0] = tu[e] << 16;                                                                \
                tb[e][1] = tu[e] & 0xffff0000u;                                                        \
            } else {   /* two plain v_sub_f32: hipcc would pack them into a v_pk_add_f32, slow beside MFMAs */ \
                xr[pr][0] = sub_f32(xr[pr][0], __uint_as_float(tb[e][0]));                             \
                xr[pr][1] = sub_f32(xr[pr][1], __uint_as_float(tb[e][1]));                             \
            }                                                                                          \
        }                                                                                              \
    }
#define RD_STEPS(fn_, ra_, lo_, hi_) _Pragma("unroll") for (int t = (lo_); t < (hi_); ++t) RD_MICRO(fn_, ra_, t)

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    constexpr int PA[6] = {2, 1, 0, 1, 0, 0};      // partial products (A plane, B plane), smallest first
    constexpr int PB[6] = {0, 1, 2, 0, 1, 0};

    // Outstanding loads at the start of stage s, oldest first: B(s) x6 | A(s+1) x4, B(s+1) x6 -- the MFMAs need B(s): at
    // most 10 may stay in flight; slot 0 adds A(s+2) x4, and the split in slot 1 needs A(s+1): again 10.
    // One stage s = 24 MFMA slots; everything else is issued in the shadows of the MFMAs, a few instructions per slot
    // (sched_barrier pins them: left alone hipcc sinks the loads to just before their first use).  Fragments (fa_, rb_)
    // are multiplied; the raw activations of s+1 (`rn_`, requested a stage ago) are split into `fn_`, a micro-op or two
    // per slot; the raw activations of s+2 are requested in slot 0 (into the set whose split finished a stage ago); a
    // filter fragment of s+2 is requested as soon as the last MFMA reading its registers has been issued (plane 2 after
    // product 2, plane 1 after product 4, plane 0 at the end).
#define RD_MF(fa_, rb_, g_)                                                                            \
    acc[((g_) >> 1) & 1][(g_) & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(                          \
        __builtin_bit_cast(frag_t, fa_[PA[(g_) >> 2]][((g_) >> 1) & 1]), __builtin_bit_cast(frag_t, rb_[PB[(g_) >> 2]][(g_) & 1]), \
        acc[((g_) >> 1) & 1][(g_) & 1], 0, 0, 0);
    constexpr int NMF = 24, SPLIT_SLOTS = 22;
#define RD_STAGE(fa_, rb_, rf_, rn_, fn_)                                                              \
    {                                                                                                  \
        RD_B_OFF();                                                                                    \
        rd_wait6<10>(rb_[0][0], rb_[0][1], rb_[1][0], rb_[1][1], rb_[2][0], rb_[2][1]);               \
        static_for<NMF>([&](auto gc) __attribute__((always_inline)) {                                  \
            constexpr int g = decltype(gc)::value;                                                     \
            RD_MF(fa_, rb_, g);                                                                        \
            if constexpr (g == 0) { RD_FETCH_A(rf_); rd_wait4<10>(rn_[0], rn_[1], rn_[2], rn_[3]); }   \
            if constexpr (g >= 1 && g <= SPLIT_SLOTS)                                                  \
                RD_STEPS(fn_, rn_, ((g - 1) * NMU + SPLIT_SLOTS - 1) / SPLIT_SLOTS, (g * NMU + SPLIT_SLOTS - 1) / SPLIT_SLOTS); \
            if constexpr (g == 12) RD_FETCH_B1(rb_, 2, 0);                                             \
            if constexpr (g == 13) RD_FETCH_B1(rb_, 2, 1);                                             \
            if constexpr (g == 20) RD_FETCH_B1(rb_, 1, 0);                                             \
            if constexpr (g == 21) RD_FETCH_B1(rb_, 1, 1);                                             \
            if constexpr (g == 23) { RD_FETCH_B1(rb_, 0, 0); RD_FETCH_B1(rb_, 0, 1); }                 \
            __builtin_amdgcn_sched_barrier(0);                                                         \
        });                                                                                            \
    }

    f32x4 raX[2 * TM], raY[2 * TM];
    u32x4 rbX[NP][TN], rbY[NP][TN];
    u32x4 faX[NP][TM], faY[NP][TM];
    RD_STAMP(1);   // index math done
    if (s_begin < s_end) {
        RD_FETCH_A(raX);
        RD_FETCH_B(rbX);
        RD_FETCH_A(raY);
        RD_FETCH_B(rbY);
        rd_wait4<16>(raX[0], raX[1], raX[2], raX[3]);
        RD_STEPS(faX, raX, 0, NMU);
        RD_STAMP(2);   // first stage's fragments in registers
        const int n_my = s_end - s_begin;
        for (int it = 0; it < ((n_my + 1) >> 1); ++it) {
            RD_STAGE(faX, rbX, raX, raY, faY);
            RD_STAGE(faY, rbY, raY, raX, faX);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the prefetches past the range (out of range: zeros)
    }
    RD_STAMP(3);   // K loop done (wave 0)

    // ---- the waves' partial sums meet in LDS: waves 1.. park their accumulators in fragment order (16 B per lane and
    // instruction), wave 0 adds them in wave order (deterministic) and carries on alone
    const int wave = 0, wm = 0, wn = 0;
    (void)wm; (void)wn;
    if constexpr (W > 1) {
        char* const sb0 = reinterpret_cast<char*>(smem);
        if (wave_k > 0) {
            char* slab = sb0 + (wave_k - 1) * SLAB + lane * 16;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        *reinterpret_cast<f32x4*>(slab + ((i * TN + j) * 4 + q) * 1024) =
                            f32x4{acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]};
        }
        __syncthreads();
        if (wave_k > 0) return;
#pragma unroll 1
        for (int g = 1; g < W; ++g)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const f32x4 v = *reinterpret_cast<const f32x4*>(sb0 + (g - 1) * SLAB + lane * 16 + ((i * TN + j) * 4 + q) * 1024);
                        acc[i][j][4 * q] += v.x; acc[i][j][4 * q + 1] += v.y; acc[i][j][4 * q + 2] += v.z; acc[i][j][4 * q + 3] += v.w;
                    }
    }

    RD_STAMP(7);   // the block's partial sums combined
    const int w_row0 = 0, w_col0 = 0;
#define BP_NT 64
#define BP_SLAST (reinterpret_cast<int*>(smem)[SMEM_BYTES / 4 - 1])
#define BP_TAIL_STAMP(k_) RD_STAMP(k_)
#include "conv_tail.inc"
    if (p.stamps) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); RD_STAMP(4); }
#undef RD_STAMP
#undef BP_NT
#undef BP_SLAST
#undef BP_TAIL_STAMP
#undef RD_FETCH_A
#undef RD_FETCH_B
#undef RD_MICRO
#undef RD_STEPS
#undef RD_MF
#undef RD_B_OFF
#undef RD_FETCH_B1
#undef RD_STAGE
}

bool conv_tile_is_rd(int tile) { return tile == TILE_RD4 || tile == TILE_RD8; }

template <int W>
static void launch_rd_t(const ConvParams& p, hipStream_t s) {
    dim3 grid(((p.M + 63) / 64) * (p.CoutPad / 64) * p.splits);
    if (g_conv_prof)
        hipExtLaunchKernelGGL((conv_rd_kernel<W>), grid, dim3(64 * W), 0, s, g_conv_prof->e0, g_conv_prof->e1, 0, p);
    else
        hipLaunchKernelGGL((conv_rd_kernel<W>), grid, dim3(64 * W), 0, s, p);
}

void launch_conv_rd(const ConvParams& p, int tile, hipStream_t s) {
    BP_CHECK(conv_h16_eligible(p), "layer is not eligible for the 16-bit MFMA paths (needs Cin % 32 == 0)");
    BP_CHECK((long long)3 * p.CoutPad * p.Kpad * 2 < (long long)OOB, "filter planes too large for 32-bit offsets");
    BP_CHECK(p.mfma_mode == PREC_BF16X3 && p.w16s != nullptr, "register-direct tiles are built for the bf16x3 mode");
    switch (tile) {
        case TILE_RD4: launch_rd_t<4>(p, s); break;
        case TILE_RD8: launch_rd_t<8>(p, s); break;
        default: throw Error("not a register-direct tile");
    }
}

}  // namespace bp
