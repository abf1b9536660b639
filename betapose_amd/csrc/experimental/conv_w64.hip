// Implicit-GEMM convolution on the 16-bit matrix pipe with a 64x64 accumulator tile PER WAVE (gfx950).
//
// Same contract as conv_igemm.hip (NHWC fp32 activations in HBM, BN-folded filters, fused epilogue, in-kernel split-K
// reduction; replaces cudnnConvolutionForward + the bias/activation/shortcut/upsample kernels of the reference's
// Darknet CUDA backend, train_YOLO/src/convolutional_kernels.cu:121-383, and the torch Conv2d/BatchNorm2d/ReLU modules
// of yolo/darknet.py:240-259 and KPD/src/models/layers/SE_Resnet.py:25-42), but built around what bounded the first
// 16-bit kernels (profiles/r01_lds_bandwidth.txt: they were LDS-bound, 72 KB through LDS per 64x64x32 chunk):
//
//  * every wave owns a 64x64 output tile = 2x2 MFMA tiles of 32x32, so each 16-B operand fragment read from LDS feeds
//    two matrix instructions (0.5 KB of LDS reads per MFMA instead of 1 KB);
//  * a block is WM x WN waves (block tile 64*WM x 64*WN); the shapes instantiated below trade tile count against
//    operand reuse per layer shape (engine.cpp picks per layer);
//  * filters arrive by LDS-DMA (buffer_load_dwordx4 ... lds) straight from the pre-split 16-bit planes in HBM: no
//    VGPR round trip, no ds_write; the LDS image of a DMA is lane-linear, so the bank swizzle is applied to the
//    per-lane SOURCE address and again on the fragment read (cdna_hip_programming.md rule 21);
//  * activations stay fp32 in HBM (every other consumer reads them as such); a thread fetches 8..32 consecutive floats
//    of ONE im2col row, splits them in registers and parks 16-B granules per plane in LDS;
//  * an LDS stage holds 16 k (one MFMA k-step): rows of 32 B, unpadded, granule g of row r at slot g ^ ((r >> 3) & 1),
//    so the 16-lane groups the hardware serves a ds_read_b128 in ({0-3,12-15,20-27}, {4-11,16-19,28-31}, +32) touch
//    all 64 banks once; two such stages (double buffer) keep a 64x128 block at 36 KB of LDS = 4 blocks = 2 waves per
//    SIMD on a CU: with 32-k stages (72 KB, one wave per SIMD) every barrier, LDS latency and the whole split-K tail
//    sat exposed (measured: 2200-2400 cycles per 48 MFMAs, 1536 being the matrix pipe's own time).
//
//   NP = 3  fp32-accurate: x = x1 + x2 + x3 exactly (three bf16 terms), six partial products per k (see conv_igemm.hip)
//   NP = 1  fp16 operands (BASELINE configs[2])
#include <cstdlib>

#include "conv_dev.h"

namespace bp {

typedef __attribute__((address_space(3))) void lds_void_t;

// ABL (debug builds only, -DBP_W64_DEBUG): bit 0 no MFMA, 1 no activation split, 2 no filter DMA, 3 no activation
// loads, 4 no barrier, 5 no fragment reads, 6 per-stage s_memtime accounting -- the ablations behind the numbers in
// DESIGN.md section 3.1d
template <int WM, int WN, int NP, int ABL = 0>
__global__ __launch_bounds__(64 * WM * WN) void conv_w64_kernel(const ConvParams p) {
    constexpr int NW = WM * WN, NT = 64 * NW;
    constexpr int BM = 64 * WM, BN = 64 * WN;
    constexpr int TM = 2, TN = 2;                 // 32x32 MFMA tiles per wave
    constexpr int LDT = BN + 4;
    constexpr int KS = 16;                        // k per LDS stage = one MFMA k-step
    constexpr int ROWB = KS * 2;                  // bytes per operand row, plane and stage
    constexpr int A_PLANE = BM * ROWB, B_PLANE = BN * ROWB;
    constexpr int STAGE = NP * (A_PLANE + B_PLANE);
    constexpr int EPI_BYTES = BM * LDT * 4;
    constexpr int SMEM_BYTES = (2 * STAGE > EPI_BYTES ? 2 * STAGE : EPI_BYTES) + 16;
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
    const int split = (int)blockIdx.x % p.splits;
    const int tile_id = (int)blockIdx.x / p.splits;
    const int tile_n = tile_id % n_tiles_n;
    const int tile_m = tile_id / n_tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int c_begin = split * p.chunks_per_split;                    // in 32-k chunks (the launch's unit)
    const int c_end = min(p.nchunks, c_begin + p.chunks_per_split);
    const int s_end = 2 * c_end;                                       // in 16-k stages
    const int plane_bytes = p.CoutPad * p.Kpad * 2;

    const __amdgpu_buffer_rsrc_t rsrcA = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.in), 0, (int)min((long long)p.N * p.H * p.W * p.in_ld * 4, (long long)OOB), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrcB = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<unsigned short*>(p.w16), 0, NP * plane_bytes, 0x00020000);

    // ---- A side: thread -> one im2col row and FPT consecutive floats of every stage
    constexpr int TPR = NT / BM;                  // threads per row (= WN)
    constexpr int FPT = KS / TPR;                 // floats per thread and stage
    constexpr int NLA = FPT / 4;                  // 16-B loads
    constexpr int GA = FPT / 8;                   // 16-B granules per plane
    static_assert(TPR == 1 || TPR == 2, "unsupported wave grid");
    const int row_a = tid / TPR, part = tid % TPR;
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
        a_base = (unsigned)((((b * p.H + iy0) * p.W + ix0) * p.in_ld + part * FPT) * 4);
        const int kx_lo = max(0, -ix0), kx_hi = min(p.ksize, p.W - ix0);
        const int ky_lo = max(0, -iy0), ky_hi = min(p.ksize, p.H - iy0);
        if (ok && kx_hi > kx_lo) {
            const unsigned long long rowbits = ((1ull << kx_hi) - 1ull) & ~((1ull << kx_lo) - 1ull);
            for (int ky = ky_lo; ky < ky_hi; ++ky) a_mask |= rowbits << (ky * p.ksize);
        }
    }
    // LDS rows are ROWB = 32 B (two 16-B granules); granule g of row r sits at slot g ^ ((r >> 3) & 1)
    const int a_wr = row_a * ROWB;                // + ((part * GA + q) ^ a_sw) * 16
    const int a_sw = (row_a >> 3) & 1;

    // ---- B side: one DMA instruction = 32 filter rows x 32 B of one plane; lane -> (row = lane >> 1, slot = lane & 1),
    // fetching granule slot ^ ((row >> 3) & 1) -- the LDS image of a DMA is lane-linear, so the swizzle goes on the source
    constexpr int RBU = BN / 32;                  // 32-row units per plane
    constexpr int NU = NP * RBU;                  // DMA units per stage
    constexpr int UPW = NU / NW;                  // per wave
    static_assert(NU % NW == 0, "DMA units must divide over the waves");
    const unsigned b_lane = (unsigned)((lane >> 1) * p.Kpad * 2 + (((lane & 1) ^ ((lane >> 4) & 1)) << 4));
    int dma_soff[UPW], dma_lds[UPW];             // wave-uniform: source offset of the unit's first row / LDS offset in a stage
#pragma unroll
    for (int i = 0; i < UPW; ++i) {
        const int u = wave + NW * i;
        const int pl = u / RBU, rb = u % RBU;
        dma_soff[i] = pl * plane_bytes + (n0 + rb * 32) * p.Kpad * 2;
        dma_lds[i] = NP * A_PLANE + pl * B_PLANE + rb * 1024;
    }

    // ---- fragment reads: lane -> row (lane & 31), granule (lane >> 5)
    const int fr = (lane & 31) * ROWB + ((((lane >> 5)) ^ ((lane >> 3) & 1)) << 4);
    const int a_rd = wm * 64 * ROWB + fr;                     // + stage + plane*A_PLANE + i*32*ROWB
    const int b_rd = NP * A_PLANE + wn * 64 * ROWB + fr;      // + stage + plane*B_PLANE + j*32*ROWB

    // ---- wave-uniform walk over K in 16-k stages: stage -> (tap, ky, kx, ci0)
    const int spt = p.Cin >> 4;                   // stages per filter tap
    int w_s = 2 * c_begin;                        // next stage whose activations get fetched
    int w_tap = w_s / spt;
    int w_ci = (w_s - w_tap * spt) << 4;
    int w_ky = w_tap / p.ksize;
    int w_kx = w_tap - w_ky * p.ksize;
    int b_s = 2 * c_begin;                        // next stage whose filters get fetched

    f32x4 ra0[NLA], ra1[NLA];                     // two activation register sets: stages s+1 (being parked) and s+2 (in flight)
    // activations of the stage the walk points at -> `ra_`, advance the walk.  Past the last stage the offset is out of
    // range: the load returns zeros without touching memory, so the loop body needs no conditionals.  The walk is
    // branch-free (scalar selects): a branch would split the stage body into basic blocks and let hipcc sink the
    // split instructions out of their MFMA slots.
#define BW_FETCH_A(ra_)                                                                                \
    {                                                                                                  \
        const unsigned delta = (unsigned)(((w_ky * p.W + w_kx) * p.in_ld + w_ci) * 4);                \
        const bool ok = ((a_mask >> w_tap) & 1ull) && w_s < s_end;                                     \
        const unsigned va = ok ? a_base + delta : OOB;                                                 \
        _Pragma("unroll") for (int j = 0; j < NLA; ++j) ra_[j] = buf_load4(rsrcA, va + 16 * j, 0);     \
        ++w_s;                                                                                         \
        const int wrap = (w_ci + KS == p.Cin) ? 1 : 0;                                                 \
        w_ci = wrap ? 0 : w_ci + KS;                                                                   \
        w_tap += wrap;                                                                                 \
        const int kx1 = w_kx + wrap;                                                                   \
        const int wrap2 = (kx1 == p.ksize) ? 1 : 0;                                                    \
        w_kx = wrap2 ? 0 : kx1;                                                                        \
        w_ky += wrap2;                                                                                 \
    }
    // filters -> LDS stage `st_` by DMA: unit i of this wave's share of the 32-row units (sbk_: the stage's byte offset
    // inside a filter row, out of range past the last stage)
#define BW_DMA_UNIT(st_, i_, sbk_)                                                                     \
    {   /* operands through locals: with the array elements written straight into the builtin's argument list   */ \
        /* hipcc (ROCm 7.2) silently drops the kernel's host stub                                                */ \
        const int lo_ = dma_lds[i_], so_ = dma_soff[i_] + (sbk_);                                      \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcB, (lds_void_t*)(sb + (st_) + lo_), 16, (int)b_lane, so_, 0, 0); \
    }
    // ---- splitting the fetched activations.  Two pairs of floats (a "group") are split in lockstep, and every
    // sub-stage (plane term + residual) is cut into three micro-ops -- 2 conversions / 4 bit operations / 4 subtractions --
    // that go into DIFFERENT MFMA slots: an in-order wave sits out the 4-8 cycle latency of every dependent VALU pair,
    // so a dependent chain issued in one piece delays the next MFMA (measured: the split cost 330 of 1190 cycles per
    // stage that way).  NP = 3: 7 micro-ops per group (3 + 3 + the plane-3 conversion); NP = 1: one fp16 conversion.
    constexpr int NPAIR = FPT / 2;
    constexpr int NGRP = NPAIR / 2;
    constexpr int MPG = NP == 3 ? 7 : 1;          // micro-ops per group
    constexpr int NMU = NGRP * MPG;
    float xr[NPAIR][2];                           // running residuals
    unsigned pk[NP][NPAIR];                       // packed 16-bit pairs per plane
    unsigned tu[2], tb[2][2];                     // a group's conversions and their two halves as fp32 bit patterns
#define BW_MICRO(ra, k_)                                                                               \
    {                                                                                                  \
        const int gi = (k_) / MPG, l = (k_) % MPG;          /* constants once the caller's loop is unrolled */ \
        const int sg = l == 6 ? 2 : l / 3, ph = l == 6 ? 0 : l % 3;                                    \
        _Pragma("unroll") for (int e = 0; e < 2; ++e) {                                                \
            const int pr = 2 * gi + e;                                                                 \
            if constexpr (NP == 1) {                                                                   \
                typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));                           \
                const f16x2 h = __builtin_convertvector(f32x2{ra[pr / 2][(pr % 2) * 2], ra[pr / 2][(pr % 2) * 2 + 1]}, f16x2); \
                pk[0][pr] = __builtin_bit_cast(unsigned, h);                                           \
            } else if (ph == 0) {                                                                      \
                if (sg == 0) { xr[pr][0] = ra[pr / 2][(pr % 2) * 2]; xr[pr][1] = ra[pr / 2][(pr % 2) * 2 + 1]; } \
                const bf16x2 h = __builtin_convertvector(f32x2{xr[pr][0], xr[pr][1]}, bf16x2);         \
                tu[e] = __builtin_bit_cast(unsigned, h);                                               \
                pk[sg][pr] = tu[e];                                                                    \
            } else if (ph == 1) {                                                                      \
                tb[e][0] = tu[e] << 16;                                                                \
                tb[e][1] = tu[e] & 0xffff0000u;                                                        \
            } else {   /* two plain v_sub_f32: hipcc would pack them into a v_pk_add_f32, slow beside MFMAs */ \
                xr[pr][0] = sub_f32(xr[pr][0], __uint_as_float(tb[e][0]));                             \
                xr[pr][1] = sub_f32(xr[pr][1], __uint_as_float(tb[e][1]));                             \
            }                                                                                          \
        }                                                                                              \
    }
    // after the last micro-op of granule q (2 groups = 4 pairs = 8 k): one 16-B LDS store per plane
#define BW_PARK_GRANULE(st_, q_)                                                                       \
    {                                                                                                  \
        char* dst = sb + (st_) + a_wr + (((part * GA + (q_)) ^ a_sw) << 4);                            \
        _Pragma("unroll") for (int pl = 0; pl < NP; ++pl)                                              \
            *reinterpret_cast<u32x4*>(dst + pl * A_PLANE) =                                            \
                u32x4{pk[pl][4 * (q_)], pk[pl][4 * (q_) + 1], pk[pl][4 * (q_) + 2], pk[pl][4 * (q_) + 3]}; \
    }
    // micro-ops [lo_, hi_) of the split, parking every granule that completes
#define BW_STEPS(st_, ra_, lo_, hi_)                                                                   \
    _Pragma("unroll") for (int t = (lo_); t < (hi_); ++t) {                                            \
        BW_MICRO(ra_, t);                                                                              \
        if ((t + 1) % (2 * MPG) == 0) BW_PARK_GRANULE(st_, (t + 1) / (2 * MPG) - 1);                   \
    }

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // partial products (A plane, B plane), smallest first
    constexpr int NPROD = NP == 1 ? 1 : 6;
    constexpr int NMF = 4 * NPROD;                // MFMAs per stage
    constexpr int PA[6] = {2, 1, 0, 1, 0, 0};
    constexpr int PB[6] = {0, 1, 2, 0, 1, 0};
    frag_t fa[2][NP][TM], fb[2][NP][TN];          // operand fragments: the stage being multiplied and the next one
#define BW_RD_PLANE(st_, fs_, pl)                                                                      \
    {                                                                                                  \
        _Pragma("unroll") for (int i = 0; i < TM; ++i) fa[fs_][pl][i] =                                \
            *reinterpret_cast<const frag_t*>(sb + (st_) + a_rd + (pl) * A_PLANE + i * 32 * ROWB);      \
        _Pragma("unroll") for (int j = 0; j < TN; ++j) fb[fs_][pl][j] =                                \
            *reinterpret_cast<const frag_t*>(sb + (st_) + b_rd + (pl) * B_PLANE + j * 32 * ROWB);      \
    }
#define BW_RD(st_, fs_) _Pragma("unroll") for (int pl = 0; pl < NP; ++pl) BW_RD_PLANE(st_, fs_, pl)
    // MFMA g of a stage: product g / 4, tile ((g >> 1) & 1, g & 1) -- consecutive MFMAs hit different accumulators
#define BW_MF(fs_, g_)                                                                                 \
    if constexpr (!(ABL & 1)) acc[((g_) >> 1) & 1][(g_) & 1] = HalfOps<NP>::mfma(fa[fs_][NP == 1 ? 0 : PA[(g_) >> 2]][((g_) >> 1) & 1], \
                                                       fb[fs_][NP == 1 ? 0 : PB[(g_) >> 2]][(g_) & 1], \
                                                       acc[((g_) >> 1) & 1][(g_) & 1]);
#define BW_SB() __builtin_amdgcn_sched_barrier(0)
#define BW_SYNC() asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory")

#define BW_STAMP(k_) if (p.stamps && tid == 0) p.stamps[(long long)blockIdx.x * 8 + (k_)] = __builtin_readcyclecounter();
    if (p.stamps && tid == 0) p.stamps[(long long)blockIdx.x * 8 + 0] = t_entry;
    BW_STAMP(1);   // index math done
    // One stage = NMF MFMA slots; everything else is issued in the 32-cycle shadows of the MFMAs, a few instructions per
    // slot (left to itself hipcc emits the split as one clump of VALU between the MFMA groups).
    // On entry: LDS[st_] holds stage s and the barrier that published it has been passed; fragment set `fs_` holds its
    // operands (read under the previous stage's last MFMAs); `rp_` holds the activations of stage s+1.
    //   slots 0..UPW-1      one filter DMA each: stage s+1 -> the other LDS stage
    //   slot  UPW           activations of stage s+2 -> `rf_` (after the DMAs, so a counted vmcnt can tell them apart)
    //   slots 0..SYNC-1     split of stage s+1's activations, one micro-op per slot, a granule parked whenever 8 k are done
    //   slot  SYNC          DMAs landed (vmcnt leaves the NLA activation loads in flight) + barrier: stage s+1 is
    //                       published; its fragments are read under the remaining MFMAs of stage s
    // Past the last stage the prefetches are out of range (no memory traffic) and the body parks zeros nobody reads.
    constexpr int SYNC = NMF - (NMF + 5) / 6;     // 20 of 24
    // debug (ABL & 64): cycles per stage spent in slots [0, UPW] / (UPW, SYNC) / wait + barrier / [SYNC, end)
    unsigned long long tacc[4] = {0, 0, 0, 0}, tprev = 0;
#define BW_STAGE(st_, fs_, rp_, rf_)                                                                   \
    {                                                                                                  \
        const int sbk = b_s < s_end ? b_s * ROWB : (int)OOB;                                           \
        ++b_s;                                                                                         \
        static_for<NMF>([&](auto gc) __attribute__((always_inline)) {                                  \
            constexpr int g = decltype(gc)::value;                                                     \
            if constexpr ((ABL & 64) && (g == 0 || g == UPW + 1 || g == SYNC)) {                       \
                const unsigned long long t_ = __builtin_readcyclecounter();                            \
                tacc[g == 0 ? 3 : (g == SYNC ? 1 : 0)] += t_ - tprev; tprev = t_;                      \
            }                                                                                          \
            BW_MF(fs_, g);                                                                             \
            if constexpr (g < UPW && !(ABL & 4)) BW_DMA_UNIT((st_) ^ STAGE, g, sbk);                   \
            if constexpr (g == UPW && !(ABL & 8)) BW_FETCH_A(rf_);                                     \
            if constexpr (g < SYNC && !(ABL & 2))                                                      \
                BW_STEPS((st_) ^ STAGE, rp_, (g * NMU + SYNC - 1) / SYNC, ((g + 1) * NMU + SYNC - 1) / SYNC);  \
            if constexpr (g == SYNC) {                                                                 \
                if constexpr (!(ABL & 16)) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" :: "n"(NLA) : "memory"); \
                if constexpr (ABL & 64) { const unsigned long long t_ = __builtin_readcyclecounter(); tacc[2] += t_ - tprev; tprev = t_; } \
            }                                                                                          \
            /* the next stage's fragments, one plane per slot behind the barrier (12 reads in one slot cost ~130 */ \
            /* cycles of issue; plane order = the order the MFMAs of a stage first need them: 2, 0, 1)            */ \
            if constexpr (g >= SYNC && g - SYNC < NP && !(ABL & 32))                                   \
                BW_RD_PLANE((st_) ^ STAGE, (fs_) ^ 1, NP == 1 ? 0 : (g - SYNC == 0 ? 2 : g - SYNC - 1)); \
            BW_SB();                                                                                   \
        });                                                                                            \
    }
    if (c_begin < c_end) {
        {
            const int sbk = b_s * ROWB;
            ++b_s;
            static_for<UPW>([&](auto ic) __attribute__((always_inline)) { BW_DMA_UNIT(0, decltype(ic)::value, sbk); });
        }
        BW_FETCH_A(ra0);
        BW_FETCH_A(ra1);
        BW_STEPS(0, ra0, 0, NMU);
        BW_SYNC();
        BW_STAMP(2);   // first stage in LDS
        BW_RD(0, 0);
        tprev = __builtin_readcyclecounter();
        // two stages per trip (LDS stages, fragment and activation register sets swap roles): the stage count of a
        // K-slice is even (2 per 32-k chunk).  Straight-line bodies, one loop exit -- with an if/else inside the loop
        // hipcc copies all 64 accumulator registers at the merge, every trip.
        for (int c = c_begin; c < c_end; ++c) {
            BW_STAGE(0, 0, ra1, ra0);
            BW_STAGE(STAGE, 1, ra0, ra1);
        }
    }
    __syncthreads();
    BW_STAMP(3);   // K loop done
    if constexpr (ABL & 64) {
        if (p.stamps && tid == 0) {
            p.stamps[(long long)blockIdx.x * 8 + 4] = tacc[0]; p.stamps[(long long)blockIdx.x * 8 + 5] = tacc[1];
            p.stamps[(long long)blockIdx.x * 8 + 6] = tacc[2]; p.stamps[(long long)blockIdx.x * 8 + 7] = tacc[3];
            return;
        }
    }

    const int w_row0 = wm * 64, w_col0 = wn * 64;
#define BP_NT NT
#define BP_SLAST (reinterpret_cast<int*>(smem)[SMEM_BYTES / 4 - 1])
#define BP_TAIL_STAMP(k_) BW_STAMP(k_)
#include "conv_tail.inc"
#undef BP_NT
#undef BP_SLAST
#undef BP_TAIL_STAMP
    if (p.stamps) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); BW_STAMP(4); }
#undef BW_STAMP
#undef BW_FETCH_A
#undef BW_DMA_UNIT
#undef BW_STAGE
#undef BW_MICRO
#undef BW_STEPS
#undef BW_PARK_GRANULE
#undef BW_RD
#undef BW_RD_PLANE
#undef BW_MF
#undef BW_SB
#undef BW_SYNC
}

bool conv_tile_is_w64(int tile) { return tile == TILE_W64_1x1 || tile == TILE_W64_1x2 || tile == TILE_W64_2x1 || tile == TILE_W64_2x2; }

template <int WM, int WN, int NP>
static void launch_w64_t(const ConvParams& p, hipStream_t s) {
    constexpr int BM = 64 * WM, BN = 64 * WN;
    dim3 grid(((p.M + BM - 1) / BM) * ((p.CoutPad + BN - 1) / BN) * p.splits);
    if (g_conv_prof)
        hipExtLaunchKernelGGL((conv_w64_kernel<WM, WN, NP>), grid, dim3(64 * WM * WN), 0, s, g_conv_prof->e0, g_conv_prof->e1, 0, p);
    else
        hipLaunchKernelGGL((conv_w64_kernel<WM, WN, NP>), grid, dim3(64 * WM * WN), 0, s, p);
}

template <int NP>
static void launch_w64_np(const ConvParams& p, int tile, hipStream_t s) {
    switch (tile) {
        case TILE_W64_1x1: launch_w64_t<1, 1, NP>(p, s); break;
        case TILE_W64_1x2: {
#ifdef BP_W64_DEBUG   // ablation / stage-timing builds (tools/abl.sh): -DBP_W64_DEBUG, BP_W64_ABLATE=<bits> at run time
            static const int abl = std::getenv("BP_W64_ABLATE") ? std::atoi(std::getenv("BP_W64_ABLATE")) : 0;
            if constexpr (NP == 3) {
                dim3 grid(((p.M + 63) / 64) * ((p.CoutPad + 127) / 128) * p.splits);
#define BW_ABL(a_) case a_: hipLaunchKernelGGL((conv_w64_kernel<1, 2, 3, a_>), grid, dim3(128), 0, s, p); return;
                switch (abl) { BW_ABL(1) BW_ABL(2) BW_ABL(4) BW_ABL(8) BW_ABL(16) BW_ABL(3) BW_ABL(12) BW_ABL(14) BW_ABL(32) BW_ABL(46) BW_ABL(62) BW_ABL(64) BW_ABL(66) default: break; }
#undef BW_ABL
            }
#endif
            launch_w64_t<1, 2, NP>(p, s);
        } break;
        case TILE_W64_2x1: launch_w64_t<2, 1, NP>(p, s); break;
        case TILE_W64_2x2: launch_w64_t<2, 2, NP>(p, s); break;
        default: throw Error("not a w64 tile");
    }
}

void launch_conv_w64(const ConvParams& p, int tile, hipStream_t s) {
    BP_CHECK(conv_h16_eligible(p), "layer is not eligible for the 16-bit MFMA paths (needs Cin % 32 == 0)");
    BP_CHECK((long long)3 * p.CoutPad * p.Kpad * 2 < (long long)OOB, "filter planes too large for 32-bit offsets");
    if (p.mfma_mode == PREC_BF16X3) launch_w64_np<3>(p, tile, s);
    else if (p.mfma_mode == PREC_F16) launch_w64_np<1>(p, tile, s);
    else throw Error("w64 tiles need a 16-bit precision mode");
}

}  // namespace bp
