// Implicit-GEMM convolution on the 16-bit matrix pipe with BOTH operands delivered by LDS-DMA (gfx950).
//
// Same contract as conv_igemm.hip (fused bias / LeakyReLU / ReLU / residual / SE scale / upsample / PixelShuffle / NCHW
// epilogue, in-kernel split-K with a last-arriver reduction: conv_tail.inc; replaces cudnnConvolutionForward + the
// bias / activation / shortcut / upsample kernels of the reference's Darknet CUDA backend,
// train_YOLO/src/convolutional_kernels.cu:121-383, and the Conv2d / BatchNorm2d / ReLU modules of
// yolo/darknet.py:240-259 and KPD/src/models/layers/SE_Resnet.py:25-42).  What changed against the round-1/2 kernels
// is WHERE the operand format is made:
//
//  * the PRODUCER of an activation tensor writes what the consumer's matrix cores multiply -- three bf16 planes with
//    x == p0 + p1 + p2 exactly (fp32-accurate mode) or one fp16 plane (the reference's own half path converts the
//    activations too, convolutional_kernels.cu:268-280) -- next to the fp32 tensor (ConvParams::out16, conv_dev.h
//    emit_planes4).  A value is split once, where it is computed, instead of once per tap and per N-tile that reads it
//    (~36x for a 128 -> 256 3x3 layer);
//  * the consumer's K loop therefore has no conversion, no operand VGPRs and no ds_write: activations AND filters go
//    HBM/L2 -> LDS by `buffer_load_dwordx4 ... lds`, the im2col gather being the per-lane source offset (out-of-image
//    taps are out-of-range offsets: the hardware writes zeros), and the only VALU work per stage is that offset;
//  * with no registers tied up by prefetch, the LDS ring is NST stages deep with counted `s_waitcnt vmcnt(N)` and ONE
//    raw `s_barrier` per stage: NST - 1 stages stay in flight across every barrier.
//
// LDS image of one 32-k chunk: per plane [rows][64 B], 16-B granule g (k = 8g .. 8g+7) of row r at slot g ^ ((r >> 2) & 3):
// the four 16-lane groups a ds_read_b128 is served in ({0-3,12-15,20-27}, {4-11,16-19,28-31}, +32: MI355X_MICROARCH.md,
// LDS) each touch all 64 banks once.  An LDS-DMA instruction writes lane-linear (base + 16 * lane = 16 rows x 64 B), so
// the swizzle goes on the SOURCE: an activation lane (row = lane >> 2, slot = lane & 3) fetches granule
// slot ^ ((row >> 2) & 3) of its pixel's channel run; the filters are stored in HBM as that image already
// (ConvParams::wpl, launch_pack_wpl: [CoutPad/64][chunk][plane][64 rows][64 B]), so their DMA is one contiguous 1 KB
// per instruction.
//
//   NP = 3  fp32-accurate: six partial products per k on v_mfma_f32_32x32x16_bf16 (see conv_igemm.hip)
//   NP = 1  fp16 operands on v_mfma_f32_32x32x16_f16 (BASELINE configs[2])
#include <algorithm>
#include <cstdlib>

#include "conv_dev.h"

namespace bp {

typedef __attribute__((address_space(3))) void lds_void_t;

// one LDS-DMA instruction: 64 lanes x 16 B, LDS destination lane-linear from `lds`.  Operands come in as plain locals:
// array elements written straight into the builtin's argument list made hipcc (ROCm 7.2) drop the kernel's host stub.
__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t rsrc, char* lds, unsigned voff, int soff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void_t*)lds, 16, (int)voff, soff, 0, 0);
}

// WM x WN waves, each a (32 TM) x (32 TN) accumulator tile; NST LDS stages of CPS 32-k chunks each.
template <int NP, int WM, int WN, int TM, int TN, int NST, int CPS>
__global__ __launch_bounds__(64 * WM * WN) void conv_pl_kernel(const ConvParams p) {
    constexpr int NW = WM * WN, NT = 64 * NW;
    constexpr int BM = 32 * TM * WM, BN = 32 * TN * WN;
    constexpr int LDT = BN + 4;
    constexpr int A_PLANE = BM * 64, B_PLANE = BN * 64;        // bytes per plane and chunk
    constexpr int CHUNK = NP * (A_PLANE + B_PLANE);
    constexpr int STAGE = CPS * CHUNK;
    constexpr int EPI_BYTES = BM * LDT * 4;
    constexpr int SMEM_BYTES = (NST * STAGE > EPI_BYTES ? NST * STAGE : EPI_BYTES) + 16;
    // the ONE LDS object of the kernel (a second one makes hipcc drain vmcnt before every fragment read)
    __shared__ __attribute__((aligned(16))) float smem[SMEM_BYTES / 4];
    char* const sb = reinterpret_cast<char*>(smem);
    typedef typename HalfOps<NP>::frag frag_t;

    const unsigned long long t_entry = p.stamps ? __builtin_readcyclecounter() : 0ull;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int n_tiles_n = (p.CoutPad + BN - 1) / BN;
    int split, tile_id;
    if (p.splits > 1) {
        // K-slice fastest: consecutive block ids (= consecutive XCDs) take different K ranges of one output tile, so every
        // XCD streams its own share of the filters through its private L2
        split = (int)blockIdx.x % p.splits;
        tile_id = (int)blockIdx.x / p.splits;
    } else {
        // block b runs on XCD b % 8: give every XCD a CONTIGUOUS range of tiles (N-tiles of one M-tile next to each other,
        // neighbouring M-tiles share their halo rows), so the re-reads of an activation tile hit that XCD's L2
        const int nblk = (int)gridDim.x, q = nblk >> 3, r = nblk & 7;
        const int xcd = (int)blockIdx.x & 7, i = (int)blockIdx.x >> 3;
        split = 0;
        tile_id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + i;
    }
    const int tile_n = tile_id % n_tiles_n;
    const int tile_m = tile_id / n_tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int c_begin = split * p.chunks_per_split;
    const int c_end = min(p.nchunks, c_begin + p.chunks_per_split);
#ifdef BP_EXPERIMENTAL   // timing ablations (wrong results): 1 no activation DMA, 2 no filter DMA, 4 no MFMA, 8 no fragment reads
    const int abl = p.abl;
#else
    constexpr int abl = 0;
#endif

    // the epilogue's bias (conv_tail.inc) is requested before anything else: the oldest load, retired by the first counted wait
    const f32x4 bias_early = *reinterpret_cast<const f32x4*>(p.bias + min(n0 + (tid % (BN / 4)) * 4, p.CoutPad - 4));

    // ---- descriptors.  Activations: ONE descriptor over the planes (base = plane 0 of the view, plane pl at scalar offset
    // pl * plane bytes); a tap outside the image, a row past M and a chunk past the K range are offsets >= OOB: zeros.
    const unsigned pl_bytes = NP == 3 ? (unsigned)(p.in16_plane * 2) : 0u;
    const __amdgpu_buffer_rsrc_t rsrcA = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<unsigned short*>(p.in16), 0, (int)min(2ll * pl_bytes + (long long)p.N * p.H * p.W * p.in_ld * 2, (long long)OOB), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrcB = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<unsigned short*>(p.wpl), 0, (int)((long long)NP * p.CoutPad * p.Kpad * 2), 0x00020000);

    // ---- A side: a DMA instruction covers 16 tile rows of one plane; wave w owns row groups w, w + NW, ... (all planes)
    constexpr int RGA = BM / 16;
    static_assert(RGA % NW == 0, "A row groups must divide over the waves");
    constexpr int GA = RGA / NW;
    unsigned a_base[GA];
    unsigned a_mask[GA];      // bit t: filter tap t of this row lies inside the image (ksize * ksize <= 32)
    {
        const int hw = p.OH * p.OW;
        const float rcp_hw = 1.0f / (float)hw, rcp_ow = 1.0f / (float)p.OW;
        const int gsw = (lane & 3) ^ ((lane >> 4) & 3);       // the granule this lane fetches (swizzle on the source)
#pragma unroll
        for (int gi = 0; gi < GA; ++gi) {
            const int m = m0 + 16 * (wave + NW * gi) + (lane >> 2);
            const bool ok = m < p.M;
            const int mm = ok ? m : 0;
            const int b = fast_div(mm, hw, rcp_hw);
            const int rem = mm - b * hw;
            const int oy = fast_div(rem, p.OW, rcp_ow);
            const int ox = rem - oy * p.OW;
            const int iy0 = oy * p.stride - p.pad, ix0 = ox * p.stride - p.pad;
            a_base[gi] = (unsigned)((((b * p.H + iy0) * p.W + ix0) * p.in_ld + gsw * 8) * 2);
            const int kx_lo = max(0, -ix0), kx_hi = min(p.ksize, p.W - ix0);
            const int ky_lo = max(0, -iy0), ky_hi = min(p.ksize, p.H - iy0);
            unsigned mask = 0;
            if (ok && kx_hi > kx_lo) {
                const unsigned rowbits = ((1u << kx_hi) - 1u) & ~((1u << kx_lo) - 1u);
                for (int ky = ky_lo; ky < ky_hi; ++ky) mask |= rowbits << (ky * p.ksize);
            }
            a_mask[gi] = mask;
        }
    }
    // ---- B side: a DMA instruction covers 16 filter rows of one plane = one contiguous 1 KB of the packed image; wave w
    // owns row groups w, w + NW, ... of the block's BN rows (all planes), so every offset is (wave base) + constant
    constexpr int RGB = BN / 16;
    static_assert(RGB % NW == 0 && (NW == 4 || NW == 8), "filter row groups must divide over the waves");
    constexpr int KB = RGB / NW;
    // row group wave + NW k: filter row n0 + 16 wave + 16 NW k -> 64-row tile (n0 >> 6) + (NW / 4) k [+ wave >> 2], row 16 (wave & 3)
    const int b_src0 = (((n0 >> 6) + (wave >> 2)) * p.nchunks * NP) * 4096 + (wave & 3) * 1024;
    const int b_srck = (NW / 4) * p.nchunks * NP * 4096;                 // per k
    const int b_lds0 = NP * A_PLANE + wave * 1024;                      // + pl * B_PLANE + k * NW * 1024
    const unsigned b_voff = (unsigned)(lane * 16);
    constexpr int IPW = CPS * NP * (GA + KB);     // DMA instructions per wave and stage

    // ---- wave-uniform walk over K in 32-k chunks: chunk -> (tap bit, byte offset of the tap's channel run)
    int w_left = c_end - c_begin;                 // chunks of this block's K range not yet requested
    int w_ci, w_kx;
    unsigned w_tapbit, w_delta;
    int w_bsrc = c_begin * (NP * 4096);
    {
        const int cpt = p.Cin >> 5;
        const int tap = c_begin / cpt;
        const int ky = tap / p.ksize;
        w_kx = tap - ky * p.ksize;
        w_ci = (c_begin - tap * cpt) << 5;
        w_tapbit = 1u << tap;
        w_delta = (unsigned)(((ky * p.W + w_kx) * p.in_ld + w_ci) * 2);
    }
    const int k_cin = p.Cin, k_ks = p.ksize;
    const unsigned k_step_tap = (unsigned)((p.in_ld - p.Cin + 32) * 2);       // last chunk of a tap -> first chunk of the next
    const unsigned k_step_row = (unsigned)(((p.W - p.ksize) * p.in_ld) * 2);  // ... and on to the next filter row

    // all DMAs of the stage the walk points at -> LDS ring slot at byte offset `so`; advances the walk.  Past the K range
    // every offset is out of range (no memory traffic; zeros land in a slot nobody reads), so the loop body is branch-free
    // and every stage issues the same number of instructions -- which is what the counted waits count.
    auto issue_stage = [&](int so) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < CPS; ++j) {
            const unsigned tapbit = w_left > 0 ? w_tapbit : 0u;
            const unsigned dead = w_left > 0 ? 0u : OOB;
            char* const dst = sb + so + j * CHUNK;
#pragma unroll
            for (int gi = 0; gi < GA; ++gi) {
                const unsigned va = (a_mask[gi] & tapbit) ? a_base[gi] + w_delta : OOB;
                char* const d = dst + (wave + NW * gi) * 1024;
                if (abl & 1) continue;
                dma16(rsrcA, d, va, 0);
                if constexpr (NP == 3) {
                    dma16(rsrcA, d + A_PLANE, va, (int)pl_bytes);
                    dma16(rsrcA, d + 2 * A_PLANE, va, (int)(2 * pl_bytes));
                }
            }
            const unsigned vb = b_voff | dead;
#pragma unroll
            for (int pl = 0; pl < NP; ++pl)
#pragma unroll
                for (int k = 0; k < KB; ++k)
                    if (!(abl & 2)) dma16(rsrcB, dst + b_lds0 + pl * B_PLANE + k * (NW * 1024), vb, b_src0 + w_bsrc + k * b_srck + pl * 4096);
            --w_left;
            w_bsrc += NP * 4096;
            const bool wrap = w_ci + 32 == k_cin;
            w_ci = wrap ? 0 : w_ci + 32;
            w_delta += wrap ? k_step_tap : 64u;
            w_tapbit = wrap ? w_tapbit << 1 : w_tapbit;
            const int kx1 = w_kx + (wrap ? 1 : 0);
            const bool wrap2 = kx1 == k_ks;
            w_kx = wrap2 ? 0 : kx1;
            w_delta += wrap2 ? k_step_row : 0u;
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // ---- fragment reads: lane -> row (lane & 31), logical granule 2 ks + (lane >> 5)
    const int fsw = ((lane & 31) >> 2) & 3;
    const int fr0 = (lane & 31) * 64 + ((((lane >> 5)) ^ fsw) << 4);          // k-step 0; k-step 1 = fr0 ^ 32
    const int a_rd = wm * (32 * TM) * 64 + fr0;
    const int b_rd = NP * A_PLANE + wn * (32 * TN) * 64 + fr0;
    constexpr int NPROD = NP == 1 ? 1 : 6;
    constexpr int PA[6] = {2, 1, 0, 1, 0, 0};     // partial products (A plane, B plane), smallest first
    constexpr int PB[6] = {0, 1, 2, 0, 1, 0};
    constexpr int NSTEP = 2 * CPS;                 // 16-k MFMA steps per stage
    frag_t fa[2][NP][TM], fb[2][NP][TN];

    auto read_step = [&](int so, auto sc, auto setc) __attribute__((always_inline)) {
        constexpr int s = decltype(sc)::value, fs = decltype(setc)::value;
        constexpr int j = s >> 1, ks = s & 1;
        const char* const base = sb + so + j * CHUNK;
        if (abl & 8) return;
#pragma unroll
        for (int pl = 0; pl < NP; ++pl) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
                fa[fs][pl][i] = *reinterpret_cast<const frag_t*>(base + pl * A_PLANE + i * (32 * 64) + (ks ? (a_rd ^ 32) : a_rd));
#pragma unroll
            for (int jn = 0; jn < TN; ++jn)
                fb[fs][pl][jn] = *reinterpret_cast<const frag_t*>(base + pl * B_PLANE + jn * (32 * 64) + (ks ? (b_rd ^ 32) : b_rd));
        }
    };
    auto mfma_step = [&](auto setc) __attribute__((always_inline)) {
        constexpr int fs = decltype(setc)::value;
        if (abl & 4) return;
#pragma unroll
        for (int q = 0; q < NPROD; ++q)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int jn = 0; jn < TN; ++jn)
                    acc[i][jn] = HalfOps<NP>::mfma(fa[fs][NP == 1 ? 0 : PA[q]][i], fb[fs][NP == 1 ? 0 : PB[q]][jn], acc[i][jn]);
    };

#define PL_STAMP(k_) if (p.stamps && tid == 0) p.stamps[(long long)blockIdx.x * 8 + (k_)] = __builtin_readcyclecounter();
    if (p.stamps && tid == 0) p.stamps[(long long)blockIdx.x * 8 + 0] = t_entry;
    PL_STAMP(1);   // index math done
    unsigned long long t_wait = 0;     // debug (p.stamps): cycles parked at the stage waits
    {   // (launches give every K slice at least one chunk)
        // prologue: NST - 1 stages in flight
#pragma unroll
        for (int s = 0; s < NST - 1; ++s) issue_stage(s * STAGE);
        int rd_off = 0, wr_off = (NST - 1) * STAGE;
        int c = c_begin;
        do {
            // this wave's DMAs of the oldest stage have landed (NST - 2 younger stages stay in flight); behind the barrier
            // everybody's have, and everybody is done reading the slot the next issue overwrites
            const unsigned long long tw0 = p.stamps ? __builtin_readcyclecounter() : 0ull;
            asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" :: "n"((NST - 2) * IPW) : "memory");
            if (p.stamps) {
                t_wait += __builtin_readcyclecounter() - tw0;
                if (c == c_begin) PL_STAMP(2);   // first stage in LDS
            }
            issue_stage(wr_off);
            wr_off = (wr_off + STAGE == NST * STAGE) ? 0 : wr_off + STAGE;
            read_step(rd_off, std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
            static_for<NSTEP>([&](auto sc) __attribute__((always_inline)) {
                constexpr int s = decltype(sc)::value;
                if constexpr (s + 1 < NSTEP)
                    read_step(rd_off, std::integral_constant<int, s + 1>{}, std::integral_constant<int, (s + 1) & 1>{});
                mfma_step(std::integral_constant<int, s & 1>{});
            });
            rd_off = (rd_off + STAGE == NST * STAGE) ? 0 : rd_off + STAGE;
            c += CPS;
        } while (c < c_end);
    }
    // the ring (still receiving the out-of-range tail issues) becomes the epilogue's staging tile
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    PL_STAMP(3);   // K loop done
    if (p.stamps && tid == 0) p.stamps[(long long)blockIdx.x * 8 + 7] = t_entry + t_wait;   // (read as a duration)

    const int w_row0 = wm * (32 * TM), w_col0 = wn * (32 * TN);
#define BP_NT NT
#define BP_SLAST (reinterpret_cast<int*>(smem)[SMEM_BYTES / 4 - 1])
#define BP_EARLY_BIAS bias_early
#define BP_TAIL_STAMP(k_) PL_STAMP(k_)
#include "conv_tail.inc"
#undef BP_NT
#undef BP_SLAST
#undef BP_EARLY_BIAS
#undef BP_TAIL_STAMP
    if (p.stamps) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); PL_STAMP(4); }
#undef PL_STAMP
}

bool conv_tile_is_pl(int tile) { return tile == TILE_PL64 || tile == TILE_PL128 || tile == TILE_PL128x64 || tile == TILE_PL256x128; }

bool conv_pl_eligible(const ConvParams& p) {
    return p.in16 != nullptr && p.wpl != nullptr && (p.Cin % 32 == 0) && (p.in_ld % 8 == 0) && p.ksize * p.ksize <= 32 &&
           ((reinterpret_cast<uintptr_t>(p.in16) & 15) == 0) && (p.in16_plane % 8 == 0);
}

template <int NP, int WM, int WN, int TM, int TN, int NST, int CPS>
static void launch_pl_t(const ConvParams& p, hipStream_t s) {
    constexpr int BM = 32 * TM * WM, BN = 32 * TN * WN;
    dim3 grid(((p.M + BM - 1) / BM) * ((p.CoutPad + BN - 1) / BN) * p.splits);
    if (g_conv_prof)
        hipExtLaunchKernelGGL((conv_pl_kernel<NP, WM, WN, TM, TN, NST, CPS>), grid, dim3(64 * WM * WN), 0, s, g_conv_prof->e0, g_conv_prof->e1, 0, p);
    else
        hipLaunchKernelGGL((conv_pl_kernel<NP, WM, WN, TM, TN, NST, CPS>), grid, dim3(64 * WM * WN), 0, s, p);
}

template <int NP>
static void launch_pl_np(const ConvParams& p, int tile, hipStream_t s) {
    switch (tile) {
        case TILE_PL64: launch_pl_t<NP, 2, 2, 1, 1, 3, NP == 1 ? 2 : 1>(p, s); break;
        case TILE_PL128: launch_pl_t<NP, 2, 2, 2, 2, NP == 1 ? 4 : 3, 1>(p, s); break;
        case TILE_PL128x64: launch_pl_t<NP, 2, 2, 2, 1, 3, NP == 1 ? 2 : 1>(p, s); break;
        case TILE_PL256x128: launch_pl_t<NP, 4, 2, 2, 2, NP == 1 ? 3 : 2, NP == 1 ? 2 : 1>(p, s); break;
        default: throw Error("not a conv_pl tile");
    }
}

void launch_conv_pl(const ConvParams& p, int tile, hipStream_t s) {
    BP_CHECK(conv_pl_eligible(p), "layer is not eligible for the operand-plane kernels (needs input planes, packed filters, Cin % 32 == 0, k*k <= 32)");
    BP_CHECK((long long)3 * p.CoutPad * p.Kpad * 2 < (long long)OOB, "filter planes too large for 32-bit offsets");
    BP_CHECK((long long)p.N * p.H * p.W * p.in_ld * 2 < (long long)OOB, "activation planes too large for 32-bit offsets");
    if (p.mfma_mode == PREC_BF16X3) launch_pl_np<3>(p, tile, s);
    else if (p.mfma_mode == PREC_F16) launch_pl_np<1>(p, tile, s);
    else throw Error("conv_pl tiles need a 16-bit precision mode");
}

// ---- filters fp32 [CoutPad][Kpad] -> the LDS image above
template <int NP>
__global__ void pack_wpl_kernel(const float* __restrict__ in, unsigned short* __restrict__ out, int CoutPad, int Kpad) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)CoutPad * Kpad) return;
    const int n = (int)(i / Kpad), k = (int)(i - (long long)n * Kpad);
    const int t64 = n >> 6, r = n & 63, c = k >> 5, kk = k & 31;
    const int slot = (kk >> 3) ^ ((r >> 2) & 3);
    const long long base = (((long long)t64 * (Kpad >> 5) + c) * NP) * 2048 + r * 32 + slot * 8 + (kk & 7);
    const float x = in[i];
    if constexpr (NP == 1) {
        const _Float16 h = (_Float16)x;
        out[base] = __builtin_bit_cast(unsigned short, h);
    } else {
        const __bf16 h1 = (__bf16)x;
        const float r1 = x - (float)h1;
        const __bf16 h2 = (__bf16)r1;
        const float r2 = r1 - (float)h2;
        const __bf16 h3 = (__bf16)r2;
        out[base] = __builtin_bit_cast(unsigned short, h1);
        out[base + 2048] = __builtin_bit_cast(unsigned short, h2);
        out[base + 4096] = __builtin_bit_cast(unsigned short, h3);
    }
}

void launch_pack_wpl(const float* in, unsigned short* out, int CoutPad, int Kpad, int np, hipStream_t s) {
    const long long n = (long long)CoutPad * Kpad;
    BP_CHECK(CoutPad % 64 == 0 && Kpad % 32 == 0, "pack_wpl: padded filter shape");
    if (np == 1) hipLaunchKernelGGL(pack_wpl_kernel<1>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, in, out, CoutPad, Kpad);
    else hipLaunchKernelGGL(pack_wpl_kernel<3>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, in, out, CoutPad, Kpad);
}

// ---- fp32 NHWC view -> operand planes, for producers that are not convolutions
__global__ void f32_to_planes_kernel(const float* __restrict__ in, int ld, int pixels, int C4, unsigned short* __restrict__ planes,
                                     long long plane_elems, int np) {
    const int total = pixels * C4;
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
        const int c4 = e % C4, px = e / C4;
        const long long idx = (long long)px * ld + c4 * 4;
        const f32x4 v = *reinterpret_cast<const f32x4*>(in + idx);
        if (np == 1) {
            *reinterpret_cast<f16x4*>(planes + idx) = __builtin_convertvector(v, f16x4);
        } else {
            const bf16x4 h1 = __builtin_convertvector(v, bf16x4);
            const f32x4 r1 = v - __builtin_convertvector(h1, f32x4);
            const bf16x4 h2 = __builtin_convertvector(r1, bf16x4);
            const f32x4 r2 = r1 - __builtin_convertvector(h2, f32x4);
            *reinterpret_cast<bf16x4*>(planes + idx) = h1;
            *reinterpret_cast<bf16x4*>(planes + plane_elems + idx) = h2;
            *reinterpret_cast<bf16x4*>(planes + 2 * plane_elems + idx) = __builtin_convertvector(r2, bf16x4);
        }
    }
}

void launch_f32_to_planes(const float* in, int ld, long long pixels, int C, unsigned short* planes, long long plane_elems,
                          int np, hipStream_t s) {
    BP_CHECK(C % 4 == 0 && ld % 4 == 0, "f32_to_planes: C and ld must be multiples of 4");
    BP_CHECK(pixels * (C / 4) < (1ll << 31), "f32_to_planes: tensor too large");
    BP_CHECK(np == 1 || np == 3, "f32_to_planes: 1 or 3 planes");
    const long long total = pixels * (C / 4);
    const int grid = (int)std::min<long long>((total + 255) / 256, 4096);
    hipLaunchKernelGGL(f32_to_planes_kernel, dim3(grid < 1 ? 1 : grid), dim3(256), 0, s, in, ld, (int)pixels, C / 4, planes, plane_elems, np);
}

}  // namespace bp
