// Implicit-GEMM convolution on the 16-bit matrix pipe with the K dimension split INSIDE the block (gfx950).
//
// Same contract as conv_igemm.hip / conv_w64.hip (NHWC fp32 activations, BN-folded filters as pre-split 16-bit planes,
// fused epilogue; replaces cudnnConvolutionForward + the bias/activation/shortcut kernels of the reference's Darknet
// CUDA backend, train_YOLO/src/convolutional_kernels.cu:121-383, and the torch Conv2d/BatchNorm2d/ReLU modules of
// yolo/darknet.py:240-259 and KPD/src/models/layers/SE_Resnet.py:25-42).  What it is for: at batch 1 a layer has 24-172
// output tiles of 64x64 for 256 CUs, so K has to be cut to fill the chip -- and cutting it ACROSS blocks costs more
// than the K loop saves (measured, profiles/r02_w64_stage_ablation.txt: slab write-through + ticket + read-back + the
// epilogue are 7.5 us against 12 us of K loop; profiles/r02_ablate_pipeline.txt: 16 % of the frame; and the slabs are a
// third of the frame's HBM traffic).  Here a block is
// G groups of 4 waves; every group runs the 64x64 tile over its own K range with its own LDS stages, and the partial
// sums meet in LDS: no slab, no ticket, one epilogue per tile.  Cross-block slices (p.splits) stay available for the
// layers with too few tiles even so and go through the shared tail (conv_tail.inc) from group 0.
//
//  * LDS stage = 16 k (one MFMA k-step) per group: rows of 32 B, granule g of row r at slot g ^ ((r >> 3) & 1) (conflict
//    free for the 16-lane groups a ds_read_b128 is served in); per group two activation slots and three filter slots
//    = 31 KB in the bf16x3 mode, so G = 4 is 124 KB and 16 waves on a CU, and the kernel is held to 128 registers: four
//    waves per SIMD, whose loads, conversions and LDS traffic hide behind each other's MFMAs without hand-placed slots;
//  * filters by LDS-DMA from a STAGE-PACKED copy of the pre-split planes (ConvParams::w16s: per 64-row tile and 16-k
//    stage one contiguous 6 KB block that already is the LDS image, swizzle included), two stages ahead; with the
//    K-contiguous layout of the other kernels a stage touched 32 B of each of 192 cache lines;
//  * activations fp32 -> registers (two stages in flight) -> exact three-way bf16 split -> 8-B LDS stores per plane;
//  * K ranges are contiguous per group; all groups run the same number of stages (the block-wide barrier), a group past
//    its range multiplies zeros (out-of-range offsets: no memory traffic).
//
//   NP = 3  fp32-accurate: x = x1 + x2 + x3 exactly (three bf16 terms), six partial products per k (see conv_igemm.hip)
//   NP = 1  fp16 operands
#include "conv_dev.h"

namespace bp {

typedef __attribute__((address_space(3))) void lds_void_kg_t;
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

template <int G, int NP>
__global__ __launch_bounds__(256 * G) __attribute__((amdgpu_waves_per_eu(4, 4))) void conv_kg_kernel(const ConvParams p) {
    constexpr int BM = 64, BN = 64, TM = 1, TN = 1;
    constexpr int LDT = BN + 4;
    constexpr int KS = 16, ROWB = KS * 2;
    constexpr int A_PLANE = BM * ROWB, B_PLANE = BN * ROWB;
    constexpr int A_SLOT = NP * A_PLANE, B_SLOT = NP * B_PLANE;   // one stage of activations / filters (all planes)
    constexpr int B_BASE = 2 * A_SLOT;            // two activation slots, then three filter slots
    constexpr int DMA_SCRATCH = B_BASE + 3 * B_SLOT;   // 1 KB behind them
    constexpr int EPI_BYTES = BM * LDT * 4;
    constexpr int GROUP_BYTES = DMA_SCRATCH + 1024 > EPI_BYTES ? DMA_SCRATCH + 1024 : EPI_BYTES;   // >= one 16 KB accumulator slab
    constexpr int SMEM_BYTES = G * GROUP_BYTES + 16;
    __shared__ __attribute__((aligned(16))) float smem[SMEM_BYTES / 4];
    char* const sb0 = reinterpret_cast<char*>(smem);
    typedef typename HalfOps<NP>::frag frag_t;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave_all = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave_all >> 2;                 // K group
    const int wave = wave_all & 3;                 // wave inside the group: 2 x 2 MFMA tiles of 32 x 32
    const int t = tid & 255;
    const int wm = wave >> 1, wn = wave & 1;
    char* const sb = sb0 + grp * GROUP_BYTES;
    const int n_tiles_n = p.CoutPad / BN;
    const int split = (int)blockIdx.x % p.splits;
    const int tile_id = (int)blockIdx.x / p.splits;
    const int tile_n = tile_id % n_tiles_n;
    const int tile_m = tile_id / n_tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int c_begin = split * p.chunks_per_split;                    // the block's K range, in 32-k chunks
    const int c_end = min(p.nchunks, c_begin + p.chunks_per_split);
    // the group's K range, in 16-k stages; every group runs `nsg` stages
    const int ns = 2 * (c_end - c_begin);
    const int nsg = (ns + G - 1) / G;
    const int s_begin = 2 * c_begin + grp * nsg;
    const int s_end = min(2 * c_end, s_begin + nsg);

    const __amdgpu_buffer_rsrc_t rsrcA = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.in), 0, (int)min((long long)p.N * p.H * p.W * p.in_ld * 4, (long long)OOB), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrcB = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<unsigned short*>(p.w16s), 0, NP * p.CoutPad * p.Kpad * 2, 0x00020000);

    // ---- filters: one DMA instruction = 1 KB (32 filter rows x 32 B of one plane) of the stage's contiguous block,
    // lane-linear on both sides; NP * 2 such units per stage and group, unit u on wave u & 3, padded so every wave issues
    // exactly two (the waits below count outstanding loads): a wave whose second unit does not exist fetches from out of
    // range (no memory traffic) into a 1 KB scratch area
    constexpr int NU = NP * 2;
    const int u0 = wave, u1 = wave + 4;
    const bool has_u1 = u1 < NU;
    static_assert(NU >= 4, "every wave's first unit must exist");
    const int b_tile = tile_n * (p.Kpad >> 4) * B_SLOT;          // byte offset of the tile's stage 0
    int b_s = s_begin;
    int b_cur = B_BASE, b_nxt = B_BASE + B_SLOT, b_dma = B_BASE + 2 * B_SLOT;   // filter slots: being read / next / DMA target
#define KG_DMA(slot_)                                                                                  \
    {                                                                                                  \
        const bool in_ = b_s < s_end;                                                                  \
        const int so0_ = in_ ? b_tile + b_s * B_SLOT + u0 * 1024 : (int)OOB;                           \
        const int so1_ = (in_ && has_u1) ? b_tile + b_s * B_SLOT + u1 * 1024 : (int)OOB;               \
        const int lo0_ = (slot_) + u0 * 1024, lo1_ = has_u1 ? (slot_) + u1 * 1024 : DMA_SCRATCH;       \
        ++b_s;                                                                                         \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcB, (lds_void_kg_t*)(sb + lo0_), 16, lane * 16, so0_, 0, 0); \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcB, (lds_void_kg_t*)(sb + lo1_), 16, lane * 16, so1_, 0, 0); \
    }

    // ---- activations: thread -> im2col row t >> 2, floats 4 * (t & 3) .. + 3 of every stage
    const int row_a = t >> 2, part = t & 3;
    unsigned a_base;
    unsigned long long a_mask = 0;
    {
        const int hw = p.OH * p.OW;
        const int m = m0 + row_a;
        const bool ok = m < p.M;
        const int mm = ok ? m : 0;
        const int b = fast_div(mm, hw, 1.0f / (float)hw);
        const int rem = mm - b * hw;
        const int oy = fast_div(rem, p.OW, 1.0f / (float)p.OW);
        const int ox = rem - oy * p.OW;
        const int iy0 = oy * p.stride - p.pad, ix0 = ox * p.stride - p.pad;
        a_base = (unsigned)((((b * p.H + iy0) * p.W + ix0) * p.in_ld + part * 4) * 4);
        const int kx_lo = max(0, -ix0), kx_hi = min(p.ksize, p.W - ix0);
        const int ky_lo = max(0, -iy0), ky_hi = min(p.ksize, p.H - iy0);
        if (ok && kx_hi > kx_lo) {
            const unsigned long long rowbits = ((1ull << kx_hi) - 1ull) & ~((1ull << kx_lo) - 1ull);
            for (int ky = ky_lo; ky < ky_hi; ++ky) a_mask |= rowbits << (ky * p.ksize);
        }
    }
    // 8 B per plane: granule part >> 1 of the row (swizzled), half part & 1 of the granule
    const int a_wr = row_a * ROWB + ((((part >> 1) ^ ((row_a >> 3) & 1))) << 4) + ((part & 1) << 3);

    // wave-uniform walk over K in 16-k stages: stage -> (tap, ky, kx, ci0)
    const int spt = p.Cin >> 4;
    int w_s = s_begin;
    int w_tap = w_s / spt;
    int w_ci = (w_s - w_tap * spt) << 4;
    int w_ky = w_tap / p.ksize;
    int w_kx = w_tap - w_ky * p.ksize;
#define KG_FETCH_A(ra_)                                                                                \
    {                                                                                                  \
        const unsigned delta = (unsigned)(((w_ky * p.W + w_kx) * p.in_ld + w_ci) * 4);                \
        const bool ok = ((a_mask >> (w_tap & 63)) & 1ull) && w_s < s_end;                              \
        ra_ = buf_load4(rsrcA, ok ? a_base + delta : OOB, 0);                                          \
        ++w_s;                                                                                         \
        const int wrap = (w_ci + KS == p.Cin) ? 1 : 0;                                                 \
        w_ci = wrap ? 0 : w_ci + KS;                                                                   \
        w_tap += wrap;                                                                                 \
        const int kx1 = w_kx + wrap;                                                                   \
        const int wrap2 = (kx1 == p.ksize) ? 1 : 0;                                                    \
        w_kx = wrap2 ? 0 : kx1;                                                                        \
        w_ky += wrap2;                                                                                 \
    }
#define KG_PARK_A(st_, ra_)                                                                            \
    {                                                                                                  \
        char* dst = sb + (st_) + a_wr;                                                                 \
        if constexpr (NP == 1) {                                                                       \
            const f16x4 h = __builtin_convertvector(ra_, f16x4);                                       \
            *reinterpret_cast<u32x2*>(dst) = __builtin_bit_cast(u32x2, h);                             \
        } else {                                                                                       \
            const bf16x4 h1 = __builtin_convertvector(ra_, bf16x4);                                    \
            const f32x4 r1 = ra_ - __builtin_convertvector(h1, f32x4);                                 \
            const bf16x4 h2 = __builtin_convertvector(r1, bf16x4);                                     \
            const f32x4 r2 = r1 - __builtin_convertvector(h2, f32x4);                                  \
            const bf16x4 h3 = __builtin_convertvector(r2, bf16x4);                                     \
            *reinterpret_cast<u32x2*>(dst) = __builtin_bit_cast(u32x2, h1);                            \
            *reinterpret_cast<u32x2*>(dst + A_PLANE) = __builtin_bit_cast(u32x2, h2);                  \
            *reinterpret_cast<u32x2*>(dst + 2 * A_PLANE) = __builtin_bit_cast(u32x2, h3);              \
        }                                                                                              \
    }

    // ---- fragment reads: lane -> row (lane & 31), granule (lane >> 5)
    const int fr = (lane & 31) * ROWB + ((((lane >> 5)) ^ ((lane >> 3) & 1)) << 4);
    const int a_rd = wm * 32 * ROWB + fr;         // + activation slot + plane * A_PLANE
    const int b_rd = wn * 32 * ROWB + fr;         // + filter slot + plane * B_PLANE

    f32x16 acc[TM][TN];
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[0][0][r] = 0.f;
    constexpr int NPROD = NP == 1 ? 1 : 6;
    constexpr int PA[6] = {2, 1, 0, 1, 0, 0};      // partial products (A plane, B plane), smallest first
    constexpr int PB[6] = {0, 1, 2, 0, 1, 0};

    // One stage s.  On entry: activation slot a_cur and filter slot b_cur hold stage s (published by the previous
    // barrier); filter slot b_nxt is receiving stage s+1 (DMA issued a stage ago); `rp_` holds the activations of stage
    // s+1 (requested a stage ago).  Outstanding loads, oldest first: A(s+1), DMA(s+1) x2 | and what this stage adds:
    // A(s+2) -> `rf_`, DMA(s+2) x2.  The barrier needs A(s+1) parked and DMA(s+1) landed: the three youngest loads stay
    // in flight.  Two stages per trip (the register sets swap roles), one loop exit: with an exit between the stages
    // hipcc copies the accumulator registers at the merge, every trip.
    int a_cur = 0;
#define KG_STAGE(rp_, rf_)                                                                             \
    {                                                                                                  \
        KG_FETCH_A(rf_);                                                                               \
        KG_DMA(b_dma);                                                                                 \
        frag_t fa[NP], fb[NP];                                                                         \
        _Pragma("unroll") for (int pl = 0; pl < NP; ++pl) {                                            \
            fa[pl] = *reinterpret_cast<const frag_t*>(sb + a_cur + a_rd + pl * A_PLANE);               \
            fb[pl] = *reinterpret_cast<const frag_t*>(sb + b_cur + b_rd + pl * B_PLANE);               \
        }                                                                                              \
        _Pragma("unroll") for (int q = 0; q < NPROD; ++q)                                              \
            acc[0][0] = HalfOps<NP>::mfma(fa[NP == 1 ? 0 : PA[q]], fb[NP == 1 ? 0 : PB[q]], acc[0][0]); \
        KG_PARK_A(a_cur ^ A_SLOT, rp_);                                                                \
        asm volatile("s_waitcnt vmcnt(3) lgkmcnt(0)\n\ts_barrier" ::: "memory");                       \
        a_cur ^= A_SLOT;                                                                               \
        { const int t_ = b_cur; b_cur = b_nxt; b_nxt = b_dma; b_dma = t_; }                            \
    }

    f32x4 ra0, ra1;
    if (c_begin < c_end) {
        KG_FETCH_A(ra0);
        KG_DMA(b_cur);
        KG_FETCH_A(ra1);
        KG_DMA(b_nxt);
        KG_PARK_A(0, ra0);
        asm volatile("s_waitcnt vmcnt(3) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        for (int it = 0; it < ((nsg + 1) >> 1); ++it) {
            KG_STAGE(ra1, ra0);
            KG_STAGE(ra0, ra1);
        }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();

    // ---- the groups' partial sums meet in LDS: groups 1.. park their accumulators in fragment order (16 B per lane and
    // instruction) in their own stage area, group 0 adds them in group order (deterministic) and carries on alone
    if constexpr (G > 1) {
        if (grp > 0) {
#pragma unroll
            for (int q = 0; q < 4; ++q)
                *reinterpret_cast<f32x4*>(sb + ((wave * 4 + q) * 64 + lane) * 16) =
                    f32x4{acc[0][0][4 * q], acc[0][0][4 * q + 1], acc[0][0][4 * q + 2], acc[0][0][4 * q + 3]};
        }
        __syncthreads();
        if (grp > 0) return;
#pragma unroll
        for (int g = 1; g < G; ++g)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(sb0 + g * GROUP_BYTES + ((wave * 4 + q) * 64 + lane) * 16);
                acc[0][0][4 * q] += v.x; acc[0][0][4 * q + 1] += v.y; acc[0][0][4 * q + 2] += v.z; acc[0][0][4 * q + 3] += v.w;
            }
    }

    const int w_row0 = wm * 32, w_col0 = wn * 32;
#define BP_NT 256
#define BP_SLAST (reinterpret_cast<int*>(smem)[SMEM_BYTES / 4 - 1])
#define BP_TAIL_STAMP(k_)
#include "conv_tail.inc"
#undef BP_NT
#undef BP_SLAST
#undef BP_TAIL_STAMP
#undef KG_DMA
#undef KG_FETCH_A
#undef KG_PARK_A
#undef KG_STAGE
}

bool conv_tile_is_kg(int tile) { return tile == TILE_KG1 || tile == TILE_KG2 || tile == TILE_KG4; }

template <int G, int NP>
static void launch_kg_t(const ConvParams& p, hipStream_t s) {
    dim3 grid(((p.M + 63) / 64) * (p.CoutPad / 64) * p.splits);
    if (g_conv_prof)
        hipExtLaunchKernelGGL((conv_kg_kernel<G, NP>), grid, dim3(256 * G), 0, s, g_conv_prof->e0, g_conv_prof->e1, 0, p);
    else
        hipLaunchKernelGGL((conv_kg_kernel<G, NP>), grid, dim3(256 * G), 0, s, p);
}

void launch_conv_kg(const ConvParams& p, int tile, hipStream_t s) {
    BP_CHECK(conv_h16_eligible(p), "layer is not eligible for the 16-bit MFMA paths (needs Cin % 32 == 0)");
    BP_CHECK((long long)3 * p.CoutPad * p.Kpad * 2 < (long long)OOB, "filter planes too large for 32-bit offsets");
    BP_CHECK(p.mfma_mode == PREC_BF16X3 && p.w16s != nullptr, "K-group tiles are built for the bf16x3 mode");
    switch (tile) {
        case TILE_KG1: launch_kg_t<1, 3>(p, s); break;
        case TILE_KG2: launch_kg_t<2, 3>(p, s); break;
        case TILE_KG4: launch_kg_t<4, 3>(p, s); break;
        default: throw Error("not a K-group tile");
    }
}

}  // namespace bp
