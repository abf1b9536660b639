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

// (dma16, conv_dev.h: operands come in as plain locals -- array elements written straight into the builtin's argument list
// made hipcc (ROCm 7.2) drop the kernel's host stub)

// WM x WN compute waves, each a (32 TM) x (32 TN) accumulator tile; NST LDS stages of CPS 32-k chunks each.
// LW > 0: LW further waves do nothing but issue the DMAs (wave specialisation).  An LDS-DMA instruction parks the issuing
// wave for 60-180 cycles, and a fp16 128x128 stage is 16 of them against 256 MFMA cycles per SIMD: with every wave doing both,
// the matrix pipe idles while its wave is stuck in the memory pipeline's queue (22 % MFMA-busy, profiles/
// r03_pmc_mfma_busy_f16_batch28.json).  A loader wave per SIMD takes that wait; the compute waves only read fragments and
// multiply.  (K slices are not supported by the specialised form: its tiles run where the grid fills the chip anyway.)
//
// KG > 1: KG wave groups of WM x WN waves per block, each with its OWN ring, walking its own contiguous part of the block's K
// range into its own accumulators; group 0 collects the partial sums through LDS behind the K loop.  For launches that
// cannot put more than one 4-wave block on a CU (the M <= 1 280 layers of the batch-1 frame): a wave issues in order and an
// LDS-DMA instruction holds it for 60+ cycles, so with one wave per SIMD the matrix pipe idles through every DMA issue
// (measured 860 cycles per 384-cycle bf16x3 chunk); a second wave on the SIMD multiplies meanwhile -- K parallelism without
// the slab hand-off of the K slices between blocks (3-5 us per launch, tools/bench_pl.py with BP_CONV_STAMPS).
//
// BDIR: the FILTER fragments skip LDS -- every wave fetches the 16-B fragments of its 32 columns straight from the
// stage-packed image ConvParams::wbd (fully coalesced 1 KB per instruction, one chunk ahead, into registers), as the
// round-2 filters-direct kernel does, while the activations still arrive as operand planes by LDS-DMA.  LDS then carries
// the activation planes only (12 KB per bf16x3 stage instead of 24: four blocks per CU instead of two), the K loop has
// neither the in-kernel fp32 -> 3 x bf16 conversion of the round-2 kernel (44 vector instructions per thread and chunk
// beside 12 MFMAs per wave: it is what the shader clock gives way to) nor the filter DMAs of the plain
// plane kernel.  64x64 tile, one chunk per stage.
//
// HRT > 0 (round 4; 3x3 / stride 1 / pad 1, fp16): the ACTIVATIONS come from an LDS-resident halo instead of one DMA'd tile per tap.
// BM consecutive output pixels (flattened over the batch) read input pixels m - W - 1 .. m + W + 1: BM + 2 W + 2 consecutive
// pixels, i.e. HRT >= BM + 2 W + 2 rows of 64 B per 32-channel group, fetched ONCE per group (HRT / 16 DMA instructions instead of
// 9 BM / 16) into one of two buffers behind the filter ring, spread over the stages of the previous group; the nine taps read
// their fragments from it at per-lane row addresses (row + ky W + kx; the source-side swizzle keyed on (row >> 2) & 3 is
// conflict-free for any 16 consecutive rows, so for any tap shift), a tap outside the image reads a zero row.  The ring
// carries the filters only.  L2 -> LDS bytes per group at 128x128: 88 KB instead of 144 KB -- the fp16 K loop is bound by exactly
// that.  K slices are cut by whole groups (nine chunks).
// B3 (round 5, halo form): THREE blocks per CU -- the epilogue staged in two slabs of 64 rows (the 128-row staging tile alone was 67.6 KB, more
// than the ring and the halos together), and the register budget of three waves per SIMD asked of the compiler (launch bounds)
template <int NP, int WM, int WN, int TM, int TN, int NST, int CPS, int LW = 0, int KG = 1, bool BDIR = false, int HRT = 0, bool B3 = false>
__global__ __launch_bounds__(64 * (WM * WN * KG + LW), B3 ? 3 : 1) void conv_pl_kernel(const ConvParams p) {
    constexpr bool HALO = HRT > 0;
    static_assert(!HALO || (NP == 1 && CPS == 1 && LW == 0 && KG == 1 && !BDIR && HRT % (16 * WM * WN) == 0), "halo form: fp16, one chunk per stage");
    static_assert(LW == 0 || KG == 1, "loader waves and K groups are alternatives");
    static_assert(!BDIR || (TM == 1 && TN == 1 && CPS == 1 && LW == 0 && KG == 1 && WM == 2 && WN == 2), "filters-direct form: 64x64 tile, 2x2 waves");
    constexpr int NWC = WM * WN;                  // compute waves (per K group)
    constexpr int NW = LW ? LW : NWC;             // waves (of a group) that issue DMAs
    constexpr int NT = 64 * (NWC * KG + LW);
    constexpr int BM = 32 * TM * WM, BN = 32 * TN * WN;
    constexpr int LDT = BN + 4;
    constexpr int A_PLANE = HALO ? 0 : BM * 64, B_PLANE = BDIR ? 0 : BN * 64;        // LDS bytes per plane and chunk
    constexpr int HALO_BYTES = HALO ? (HRT + 1) * 64 : 0;                 // one halo buffer: HRT rows + the zero row
    constexpr int CHUNK = NP * (A_PLANE + B_PLANE);
    constexpr int STAGE = CPS * CHUNK;
    constexpr int EP_SLABS_ = BM > 128 ? BM / (32 * TM > 64 ? 32 * TM : 64) : (B3 ? 2 : 1);       // epilogue staging in slabs of 64 rows (or one wave's rows) on the big tiles (conv_tail.inc)
    constexpr int EPI_BYTES = (BM / EP_SLABS_) * LDT * 4;
    constexpr int RING = NST * STAGE;
    constexpr int KGSUM_BYTES = (KG - 1) * NWC * TM * TN * 4096;      // the other groups' accumulators, fragment order
    static_assert(KGSUM_BYTES <= KG * RING, "");
    constexpr int SMEM_BYTES = (KG * RING + 2 * HALO_BYTES > EPI_BYTES ? KG * RING + 2 * HALO_BYTES : EPI_BYTES) + 16;
    // the ONE LDS object of the kernel (a second one makes hipcc drain vmcnt before every fragment read)
    __shared__ __attribute__((aligned(16))) float smem[SMEM_BYTES / 4];
    typedef typename HalfOps<NP>::frag frag_t;

    if (p.work_blocks && (int)blockIdx.x >= p.work_blocks) { prefetch_block<NT>(p, reinterpret_cast<char*>(smem), (int)blockIdx.x); return; }
    const unsigned long long t_entry = p.stamps ? bp_clock() : 0ull;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave_id = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kgrp = KG > 1 ? wave_id / NWC : 0;             // K group of the wave
    const int wv = wave_id - kgrp * NWC;                     // wave inside its group
    char* const sb = reinterpret_cast<char*>(smem) + kgrp * RING;     // the group's ring
    const bool is_loader = LW ? wave_id >= NWC : true;      // issues DMAs
    const bool has_acc = LW ? wave_id < NWC : kgrp == 0;     // owns the block's accumulators in the tail (conv_tail.inc)
    const int wave = LW ? (is_loader ? wave_id - NWC : 0) : wv;        // index among the DMA-issuing waves of the group
    const int wm = (LW ? (wave_id < NWC ? wave_id : 0) : wv) / WN, wn = (LW ? (wave_id < NWC ? wave_id : 0) : wv) % WN;
    const int n_tiles_n = (p.CoutPad + BN - 1) / BN;
    int split, tile_id;
    int my_splits = p.splits, my_cps = p.chunks_per_split;      // K slices of this block's tile (hybrid grids: 1 or hy_splits)
    if (p.hy_splits > 0 && (int)blockIdx.x >= p.hy_full) {      // hybrid grid: one of the last tiles, cut along K
        const int e = (int)blockIdx.x - p.hy_full, tl = e / p.hy_splits;
        tile_id = p.hy_full + tl;
        split = e - tl * p.hy_splits;
        my_splits = p.hy_splits;
        my_cps = p.hy_cps;
    } else if (p.xcd_home) {  // all K slices of a tile on one XCD (ConvParams::xcd_home)
        const int i = (int)blockIdx.x >> 3, tl = i / p.splits;
        tile_id = tl * 8 + ((int)blockIdx.x & 7);
        if (tile_id >= p.n_tiles) return;                // padding of the last round of tiles
        split = i - tl * p.splits;
        xcd_home_mark(p, tile_id, split);
    } else if (p.splits > 1) {
        // K-slice fastest: consecutive block ids (= consecutive XCDs) take different K ranges of one output tile, so every
        // XCD streams its own share of the filters through its private L2
        split = (int)blockIdx.x % p.splits;
        tile_id = (int)blockIdx.x / p.splits;
    } else {
        // block b runs on XCD b % 8: give every XCD a CONTIGUOUS range of tiles (N-tiles of one M-tile next to each other,
        // neighbouring M-tiles share their halo rows), so the re-reads of an activation tile hit that XCD's L2
        const int nblk = p.hy_splits > 0 ? p.hy_full : (p.n_tiles ? p.n_tiles : (int)gridDim.x), q = nblk >> 3, r = nblk & 7;
        const int xcd = (int)blockIdx.x & 7, i = (int)blockIdx.x >> 3;
        split = 0;
        tile_id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + i;
    }
    const int tile_n = tile_id % n_tiles_n;
    const int tile_m = tile_id / n_tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int cb_blk = split * my_cps, ce_blk = min(p.nchunks, cb_blk + my_cps);
    const int c_per = (ce_blk - cb_blk + KG - 1) / KG;                // chunks per K group (the last may get fewer, or none)
    const int c_begin = KG > 1 ? min(cb_blk + kgrp * c_per, ce_blk) : cb_blk;
    const int c_end = KG > 1 ? min(c_begin + c_per, ce_blk) : ce_blk;
    const int n_it = (c_per + CPS - 1) / CPS;                         // stages: the same for every group (they share the barriers)
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
        const_cast<unsigned short*>(BDIR ? p.wbd : p.wpl), 0, (int)((long long)NP * p.CoutPad * p.Kpad * 2), 0x00020000);

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
        unsigned row_pattern = 0;                             // bit ksize * ky for every filter row (wave-uniform)
        for (int ky = 0; ky < p.ksize; ++ky) row_pattern |= 1u << (ky * p.ksize);
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
            // taps (ky, kx) inside the image = [ky_lo, ky_hi) x [kx_lo, kx_hi): the kx run, repeated at bit ksize * ky for
            // every ky of the range -- one multiply by the 'one bit per filter row' pattern (no overlap: the run is < 2^ksize)
            unsigned mask = 0;
            if (ok && kx_hi > kx_lo && ky_hi > ky_lo) {
                const unsigned rowbits = ((1u << kx_hi) - 1u) & ~((1u << kx_lo) - 1u);
                const unsigned kyrange = (ky_hi * p.ksize >= 32 ? 0xffffffffu : (1u << (ky_hi * p.ksize)) - 1u) & ~((1u << (ky_lo * p.ksize)) - 1u);
                mask = rowbits * (row_pattern & kyrange);
            }
            a_mask[gi] = mask;
        }
    }
    // ---- B side: a DMA instruction covers 16 filter rows of one plane = one contiguous 1 KB of the packed image; wave w
    // owns row groups w, w + NW, ... of the block's BN rows (all planes), so every offset is (wave base) + constant
    constexpr int RGB = BN / 16;
    static_assert(RGB % NW == 0 && (NW == 4 || NW == 8), "filter row groups must divide over the waves");
    constexpr int KB = BDIR ? 0 : RGB / NW;
    // row group wave + NW k: filter row n0 + 16 wave + 16 NW k -> 64-row tile (n0 >> 6) + (NW / 4) k [+ wave >> 2], row 16 (wave & 3)
    const int b_src0 = (((n0 >> 6) + (wave >> 2)) * p.nchunks * NP) * 4096 + (wave & 3) * 1024;
    const int b_srck = (NW / 4) * p.nchunks * NP * 4096;                 // per k
    const int b_lds0 = NP * A_PLANE + wave * 1024;                      // + pl * B_PLANE + k * NW * 1024
    const unsigned b_voff = (unsigned)(lane * 16);
    constexpr int IPW = CPS * NP * (GA + KB);     // DMA instructions per wave and stage
    // ---- halo form: wave w fetches the 16-row pieces w, w + NW, ... of the halo (HPW per wave and group); lane -> row (lane >> 2)
    // of the piece, granule (lane & 3) ^ ((row >> 2) & 3) of input pixel m0 - W - 1 + row (outside the tensor: out of range, zeros)
    constexpr int HPW = HALO ? HRT / (16 * NW) : 1;
    static_assert(!HALO || 9 - HPW >= NST - 1, "the prologue's stages carry no halo pieces");
    char* const hal = reinterpret_cast<char*>(smem) + KG * RING;
    unsigned h_va[HPW];
    unsigned fa_off[9][TM];
    if constexpr (HALO) {
        const int gsw = (lane & 3) ^ ((lane >> 4) & 3);
        const long long npix = (long long)p.N * p.H * p.W;
#pragma unroll
        for (int k = 0; k < HPW; ++k) {
            const long long pix = (long long)m0 - p.W - 1 + 16 * (wave + NW * k) + (lane >> 2);
            h_va[k] = (pix >= 0 && pix < npix) ? (unsigned)((pix * p.in_ld + gsw * 8) * 2) : OOB;
        }
        const int hw = p.OH * p.OW;
#pragma unroll
        for (int e = 0; e < TM; ++e) {
            const int lm = wm * (32 * TM) + e * 32 + (lane & 31);
            const int m = m0 + lm;
            const bool ok = m < p.M;
            const int mm = ok ? m : 0;
            const int rem = mm % hw;
            const int oy = rem / p.OW, ox = rem - oy * p.OW;
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int ky = t / 3, kx = t - 3 * ky;
                const int iy = oy + ky - 1, ix = ox + kx - 1;
                const bool in_img = ok && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
                const int r = lm + ky * p.W + kx;
                fa_off[t][e] = in_img ? (unsigned)(r * 64 + (((lane >> 5) ^ ((r >> 2) & 3)) << 4)) : (unsigned)(HRT * 64 + ((lane >> 5) << 4));
            }
        }
        if (tid < 8) *reinterpret_cast<u32x4*>(hal + (tid >> 2) * HALO_BYTES + HRT * 64 + (tid & 3) * 16) = u32x4{0u, 0u, 0u, 0u};   // the zero rows
        __syncthreads();
    }

    // ---- wave-uniform walk over K in 32-k chunks.  K ORDER: (32-channel group, ky, kx, channel) -- the filter taps are
    // the INNER loop.  A 3x3 layer re-reads every activation row once per tap; tap-major order (the order of the round-1/2
    // kernels) puts those nine reads a whole channel sweep apart, and at batch 28 the rows an XCD's blocks hold between two
    // reads of the same line exceed its 4 MB L2 (tools/micro/dma_bw.hip: LDS-DMA moves 55 B/clk/CU from L2, 14 from the
    // memory side).  With the taps innermost the nine reads of a 64-B channel run are nine CONSECUTIVE stages.
    // chunk c -> (group c / taps, tap c % taps); the packed filters (launch_pack_wpl) use the same order.
    int w_left = c_end - c_begin;                 // chunks of this block's K range not yet requested
    int w_kx, w_tap;
    unsigned w_delta;
    int w_bsrc = c_begin * (NP * 4096);
    const int k_ks = p.ksize, k_taps = p.ksize * p.ksize;
    {
        const int grp = c_begin / k_taps;
        w_tap = c_begin - grp * k_taps;
        const int ky = w_tap / p.ksize;
        w_kx = w_tap - ky * p.ksize;
        w_delta = (unsigned)(((ky * p.W + w_kx) * p.in_ld + grp * 32) * 2);
    }
    const unsigned k_step_kx = (unsigned)(p.in_ld * 2);                                      // next tap in the filter row
    const unsigned k_step_row = (unsigned)(((p.W - (p.ksize - 1)) * p.in_ld) * 2);           // ... first tap of the next row
    const unsigned k_d_row = k_step_row - k_step_kx;
    const unsigned k_step_grp = (unsigned)(64 - (((p.ksize - 1) * p.W + (p.ksize - 1)) * p.in_ld) * 2);   // ... tap 0 of the next group

    const unsigned k_d_grp = k_step_grp - k_step_row;
    // ---- one stage's DMA list, addressed by a compile-time index so that every instruction can be pinned into its own
    // MFMA slot.  Per chunk j of the stage: GA x NP activation pieces (row group, plane), then NP x KB filter pieces.
    // The per-lane / per-wave offsets of the stage are computed once (stage_addr) from the walk, BEFORE the stage wait.
    // Past the K range every offset is out of range (no memory traffic; zeros land in a slot nobody reads), so the loop
    // body is branch-free and every stage issues the same number of instructions -- what the counted waits count.
    constexpr int DPC = NP * (GA + KB);           // pieces per chunk
    static_assert(IPW == CPS * DPC, "");
    unsigned st_va[CPS][GA], st_vb[CPS];
    int st_bs[CPS];
    auto stage_addr = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < CPS; ++j) {
            const unsigned tapbit = w_left > 0 ? (1u << w_tap) : 0u;
            const unsigned dead = w_left > 0 ? 0u : OOB;
#pragma unroll
            for (int gi = 0; gi < GA; ++gi) st_va[j][gi] = (a_mask[gi] & tapbit) ? a_base[gi] + w_delta : OOB;
            st_vb[j] = b_voff | dead;
            st_bs[j] = b_src0 + w_bsrc;
            --w_left;
            w_bsrc += NP * 4096;
            // (plain sums of selected increments: with the nested select `a ? x : (b ? y : z)` on this captured state hipcc
            // kept the whole walk in scratch and wrapped every DMA in a readfirstlane waterfall loop -- 2-3x slower)
            const int last_tap = (w_tap + 1 == k_taps) ? 1 : 0;
            const int last_kx = (w_kx + 1 == k_ks) ? 1 : 0;
            w_delta += k_step_kx + (last_kx ? k_d_row : 0u) + (last_tap ? k_d_grp : 0u);
            w_tap = last_tap ? 0 : w_tap + 1;
            w_kx = last_kx ? 0 : w_kx + 1;
        }
    };
    auto dma_piece = [&](int so, auto dc) __attribute__((always_inline)) {
        constexpr int d = decltype(dc)::value;
        constexpr int j = d / DPC, e = d % DPC;
        char* const dst = sb + so + j * CHUNK;
        if constexpr (e < GA * NP) {
            constexpr int gi = e / NP, pl = e % NP;
            if (abl & 1) return;
            const unsigned va = st_va[j][gi];
            dma16(rsrcA, dst + pl * A_PLANE + (wave + NW * gi) * 1024, va, pl == 0 ? 0 : (pl == 1 ? (int)pl_bytes : (int)(2 * pl_bytes)));
        } else {
            constexpr int pl = (e - GA * NP) / KB, k = (e - GA * NP) % KB;
            if (abl & 2) return;
            const unsigned vb = st_vb[j];
            const int so_ = st_bs[j] + k * b_srck + pl * 4096;
            dma16(rsrcB, dst + b_lds0 + pl * B_PLANE + k * (NW * 1024), vb, so_);
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // ---- BDIR: filter fragments of chunk c for this wave's 32 columns = NP x 2 pieces of 1 KB at bd_tile + c * (2 NP) KB x 2:
    // [k-step][plane][64 rows x 32 B]; lane -> row 32 wn + (lane & 31), granule (lane >> 5) at slot granule ^ ((row >> 3) & 1)
    // (the image of launch_f32_to_bf16x3_staged, in this kernel's K order).  Past the K range: out of range, zeros.
    const int bd_tile = tile_n * p.nchunks * (2 * NP * 2048);
    const unsigned bd_voff = (unsigned)((32 * wn + (lane & 31)) * 32 + (((lane >> 5) ^ ((lane >> 3) & 1)) << 4));
    int bd_c = c_begin;
    frag_t fbq[2][NP][2];          // [chunk parity][plane][k-step]
    auto load_b = [&](auto bufc) __attribute__((always_inline)) {
        constexpr int buf = decltype(bufc)::value;
        const int so = bd_c < c_end ? bd_tile + bd_c * (2 * NP * 2048) : (int)OOB;
        ++bd_c;
        static_for<NP * 2>([&](auto ec) __attribute__((always_inline)) {
            constexpr int pl = decltype(ec)::value / 2, ks = decltype(ec)::value % 2;
            fbq[buf][pl][ks] = __builtin_bit_cast(frag_t, __builtin_amdgcn_raw_buffer_load_b128(rsrcB, (int)bd_voff, so + ks * (NP * 2048) + pl * 2048, 0));
        });
    };

    // ---- fragment reads: lane -> row (lane & 31), logical granule 2 ks + (lane >> 5)
    const int fsw = ((lane & 31) >> 2) & 3;
    const int fr0 = (lane & 31) * 64 + ((((lane >> 5)) ^ fsw) << 4);          // k-step 0; k-step 1 = fr0 ^ 32
    const int a_rd = wm * (32 * TM) * 64 + fr0;
    const int b_rd = NP * A_PLANE + wn * (32 * TN) * 64 + fr0;
    constexpr int NPROD = NP == 1 ? 1 : 6;
    constexpr int PA[6] = {2, 1, 0, 1, 0, 0};     // partial products (A plane, B plane), smallest first
    constexpr int PB[6] = {0, 1, 2, 0, 1, 0};
    constexpr int NSTEP = 2 * CPS;                 // 16-k MFMA steps per stage
    constexpr int NMF = NPROD * TM * TN;           // MFMAs per step
    constexpr int TNR = BDIR ? 0 : TN;             // filter fragments read from LDS per plane and step
    constexpr int NRD = NP * (TM + TNR);           // fragments per step
    frag_t fa[2][NP][TM], fb[2][NP][TN];
    // fragment r of a step, in the order the step's MFMAs first need them: (A plane 2, B plane 0), (A 1, B 1), (A 0, B 2)
    constexpr int RPA[3] = {2, 1, 0}, RPB[3] = {0, 1, 2};
    auto read_frag = [&](int so, auto sc, auto rc) __attribute__((always_inline)) {
        constexpr int s = decltype(sc)::value, r = decltype(rc)::value, fs = s & 1;
        constexpr int j = s >> 1, ks = s & 1;
        constexpr int grp = r / (TM + TNR), e = r % (TM + TNR);
        if (abl & 8) return;
        const char* const base = sb + so + j * CHUNK;
        if constexpr (e < TM) {
            constexpr int pl = NP == 1 ? 0 : RPA[grp];
            fa[fs][pl][e] = *reinterpret_cast<const frag_t*>(base + pl * A_PLANE + e * (32 * 64) + (ks ? (a_rd ^ 32) : a_rd));
        } else {
            constexpr int pl = NP == 1 ? 0 : RPB[grp];
            fb[fs][pl][e - TM] = *reinterpret_cast<const frag_t*>(base + pl * B_PLANE + (e - TM) * (32 * 64) + (ks ? (b_rd ^ 32) : b_rd));
        }
    };
    auto mfma_one = [&](auto sc, auto mc, auto parc) __attribute__((always_inline)) {
        constexpr int fs = decltype(sc)::value & 1, m = decltype(mc)::value, par = decltype(parc)::value;
        constexpr int q = m / (TM * TN), i = (m / TN) % TM, jn = m % TN;
        if (abl & 4) return;
        if constexpr (BDIR) acc[i][jn] = HalfOps<NP>::mfma(fa[fs][NP == 1 ? 0 : PA[q]][i], fbq[par][NP == 1 ? 0 : PB[q]][fs], acc[i][jn]);
        else acc[i][jn] = HalfOps<NP>::mfma(fa[fs][NP == 1 ? 0 : PA[q]][i], fb[fs][NP == 1 ? 0 : PB[q]][jn], acc[i][jn]);
    };

#define PL_STAMP(k_) if (p.stamps && tid == 0) p.stamps[(long long)(p.hy_splits > 0 ? (int)blockIdx.x : tile_id * p.splits + split) * 8 + (k_)] = bp_clock();
#define PL_SB() __builtin_amdgcn_sched_barrier(0)
    if (p.stamps && tid == 0) p.stamps[(long long)(p.hy_splits > 0 ? (int)blockIdx.x : tile_id * p.splits + split) * 8 + 0] = t_entry;
    PL_STAMP(1);   // index math done
    unsigned long long t_wait = 0;     // debug (p.stamps): cycles parked at the stage waits
    // One stage = NSTEP x NMF MFMA slots.  An LDS-DMA instruction costs the issuing wave 60-180 cycles and a wave issues in
    // order, so a stage's DMAs in one clump leave the matrix pipe idle for their whole issue time (measured: 1 130 cycles
    // per 384-cycle stage that way, of which 120 parked at the wait).  So every non-MFMA instruction is pinned into the
    // shadow of an MFMA (sched_barrier): behind the barrier the first step's fragments are requested, N0 DMAs are issued
    // while they arrive, and every MFMA slot then carries its share of the remaining DMAs and of the NEXT step's
    // fragment reads (front-loaded into the first half of the step, so the last ones are back before they are needed).
    constexpr int SLOTS = NSTEP * NMF;
    constexpr int N0 = IPW < 2 ? IPW : 2;
    constexpr int DREM = IPW - N0;
    constexpr int HALF = NMF / 2 > 0 ? NMF / 2 : 1;
    if constexpr (HALO) {
        // bundle of the stage with tap t = its KB filter pieces + (t >= 9 - HPW) one piece of the NEXT group's halo; issued NST - 1
        // stages ahead like every stage's DMAs, so the counted wait that covers a stage covers the halo pieces issued with it
        auto hp_of = [](int t) constexpr { return t >= 9 - HPW ? 1 : 0; };
        unsigned h_ndelta = (unsigned)((c_begin / 9) * 64);      // byte offset of the 32-channel group whose halo is fetched next
        int h_nbuf = 0, h_cur = 0;                                // buffer that receives it / buffer the current group reads
        int h_left = (c_end - c_begin) / 9;                       // groups of this block's K range not yet fetched
        unsigned hst_vb; int hst_bs;
        auto stage_addr_b = [&]() __attribute__((always_inline)) {
            hst_vb = b_voff | (w_left > 0 ? 0u : OOB);
            hst_bs = b_src0 + w_bsrc;
            --w_left;
            w_bsrc += NP * 4096;
        };
        auto dma_piece_h = [&](int so, auto tisc, auto dc) __attribute__((always_inline)) {
            constexpr int tis = decltype(tisc)::value, d = decltype(dc)::value;
            if constexpr (d < KB) {
                dma16(rsrcB, sb + so + b_lds0 + d * (NW * 1024), hst_vb, hst_bs + d * b_srck);
            } else {
                constexpr int kk = tis - (9 - HPW);
                dma16(rsrcA, hal + h_nbuf + (wave + NW * kk) * 1024, h_left > 0 ? h_va[kk] + h_ndelta : OOB, 0);
            }
        };
        auto read_frag_h = [&](int so, auto tapc, auto sc, auto rc) __attribute__((always_inline)) {
            constexpr int tap = decltype(tapc)::value, s_ = decltype(sc)::value, r = decltype(rc)::value, fs = s_ & 1, ks = s_ & 1;
            if constexpr (r < TM) {
                const unsigned o = ks ? (fa_off[tap][r] ^ 32u) : fa_off[tap][r];
                fa[fs][0][r] = *reinterpret_cast<const frag_t*>(hal + h_cur + o);
            } else {
                fb[fs][0][r - TM] = *reinterpret_cast<const frag_t*>(sb + so + (r - TM) * (32 * 64) + (ks ? (b_rd ^ 32) : b_rd));
            }
        };
        // prologue: the first group's halo, then NST - 1 stages of filters
        static_for<HPW>([&](auto kc) __attribute__((always_inline)) {
            dma16(rsrcA, hal + (wave + NW * decltype(kc)::value) * 1024, h_va[decltype(kc)::value] + h_ndelta, 0);
        });
        h_ndelta += 64; h_nbuf = HALO_BYTES; --h_left;
        static_for<NST - 1>([&](auto sc) __attribute__((always_inline)) {
            stage_addr_b();
            static_for<KB>([&](auto dc) __attribute__((always_inline)) { dma_piece_h(decltype(sc)::value * STAGE, sc, dc); });
        });
        int rd_off = 0, wr_off = (NST - 1) * STAGE;
        auto stage_body_h = [&](auto tapc) __attribute__((always_inline)) {
            constexpr int tap = decltype(tapc)::value;
            constexpr int tis = (tap + NST - 1) % 9;                    // tap of the stage whose bundle is issued here
            constexpr int IPWT = KB + hp_of(tis);
            constexpr int N0T = IPWT < 2 ? IPWT : 2, DREMT = IPWT - N0T;
            constexpr int WAITN = [&]() constexpr { int n = 0; for (int k = 1; k <= NST - 2; ++k) n += KB + hp_of((tap + k) % 9); return n; }();
            stage_addr_b();
            asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" :: "n"(WAITN) : "memory");
            static_for<NRD>([&](auto rc) __attribute__((always_inline)) { read_frag_h(rd_off, tapc, std::integral_constant<int, 0>{}, rc); });
            PL_SB();
            static_for<N0T>([&](auto dc) __attribute__((always_inline)) { dma_piece_h(wr_off, std::integral_constant<int, tis>{}, dc); });
            PL_SB();
            static_for<SLOTS>([&](auto gc) __attribute__((always_inline)) {
                constexpr int g = decltype(gc)::value, s_ = g / NMF, m = g % NMF;
                mfma_one(std::integral_constant<int, s_>{}, std::integral_constant<int, m>{}, std::integral_constant<int, 0>{});
                constexpr int d_lo = N0T + (g * DREMT + SLOTS - 1) / SLOTS, d_hi = N0T + ((g + 1) * DREMT + SLOTS - 1) / SLOTS;
                static_for<d_hi - d_lo>([&](auto k) __attribute__((always_inline)) {
                    dma_piece_h(wr_off, std::integral_constant<int, tis>{}, std::integral_constant<int, d_lo + decltype(k)::value>{});
                });
                if constexpr (s_ + 1 < NSTEP && m < HALF) {
                    constexpr int r_lo = (m * NRD + HALF - 1) / HALF, r_hi = ((m + 1) * NRD + HALF - 1) / HALF;
                    static_for<r_hi - r_lo>([&](auto k) __attribute__((always_inline)) {
                        read_frag_h(rd_off, tapc, std::integral_constant<int, s_ + 1>{}, std::integral_constant<int, r_lo + decltype(k)::value>{});
                    });
                }
                PL_SB();
            });
            wr_off = (wr_off + STAGE == NST * STAGE) ? 0 : wr_off + STAGE;
            rd_off = (rd_off + STAGE == NST * STAGE) ? 0 : rd_off + STAGE;
            if constexpr (tis == 8) { h_ndelta += 64; h_nbuf ^= HALO_BYTES; --h_left; }     // the next group's halo is on its way
            if constexpr (tap == 8) h_cur ^= HALO_BYTES;                                      // this group is done
        };
        const int n_grp = (n_it + 8) / 9;
        for (int gi = 0; gi < n_grp; ++gi)
            static_for<9>([&](auto tapc) __attribute__((always_inline)) { stage_body_h(tapc); });
    } else
    {   // (launches give every K slice at least one chunk)
        // prologue: NST - 1 stages in flight
        static_for<NST - 1>([&](auto sc) __attribute__((always_inline)) {
            stage_addr();
            if (LW == 0 || is_loader)
                static_for<IPW>([&](auto dc) __attribute__((always_inline)) { dma_piece(decltype(sc)::value * STAGE, dc); });
            // BDIR: filter fragments run TWO chunks ahead (a cold chunk comes from HBM: ~2 us against ~0.3 us of work per
            // chunk); issue order [stage 0, filters 0, stage 1, filters 1] so that the first counted wait covers stage 0 and
            // filters 0
            if constexpr (BDIR) {
                static_assert(NST == 3, "the filters-direct form counts on a 3-deep ring");
                load_b(std::integral_constant<int, decltype(sc)::value>{});
            }
        });
        int rd_off = 0, wr_off = (NST - 1) * STAGE;
        int it = 0;
        auto stage_body = [&](auto parc) __attribute__((always_inline)) {
            stage_addr();
            // this wave's DMAs of the oldest stage have landed (NST - 2 younger stages stay in flight); behind the barrier
            // everybody's have, and everybody is done reading the slot the next issue overwrites
#ifdef BP_EXPERIMENTAL
            const unsigned long long tw0 = p.stamps ? bp_clock() : 0ull;
#endif
            // (BDIR: per stage the wave issues [the IPW DMAs of stage i + 2, then the 2 NP filter loads of chunk i + 2]; in
            // flight behind the wait = what the previous stage issued -- stage i and chunk i, issued before that, are in)
            asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" :: "n"(BDIR ? IPW + 2 * NP : (NST - 2) * IPW) : "memory");
#ifdef BP_EXPERIMENTAL
            if (p.stamps) t_wait += bp_clock() - tw0;
#endif
            if constexpr (LW == 0) {
                static_for<NRD>([&](auto rc) __attribute__((always_inline)) { read_frag(rd_off, std::integral_constant<int, 0>{}, rc); });
                PL_SB();
                static_for<N0>([&](auto dc) __attribute__((always_inline)) { dma_piece(wr_off, dc); });
                PL_SB();
                static_for<SLOTS>([&](auto gc) __attribute__((always_inline)) {
                    constexpr int g = decltype(gc)::value, s = g / NMF, m = g % NMF;
                    mfma_one(std::integral_constant<int, s>{}, std::integral_constant<int, m>{}, parc);
                    constexpr int d_lo = N0 + (g * DREM + SLOTS - 1) / SLOTS, d_hi = N0 + ((g + 1) * DREM + SLOTS - 1) / SLOTS;
                    static_for<d_hi - d_lo>([&](auto k) __attribute__((always_inline)) {
                        dma_piece(wr_off, std::integral_constant<int, d_lo + decltype(k)::value>{});
                    });
                    if constexpr (s + 1 < NSTEP && m < HALF) {
                        constexpr int r_lo = (m * NRD + HALF - 1) / HALF, r_hi = ((m + 1) * NRD + HALF - 1) / HALF;
                        static_for<r_hi - r_lo>([&](auto k) __attribute__((always_inline)) {
                            read_frag(rd_off, std::integral_constant<int, s + 1>{}, std::integral_constant<int, r_lo + decltype(k)::value>{});
                        });
                    }
                    PL_SB();
                });
            } else if (is_loader) {      // (wave-uniform) loader waves: the whole stage's DMAs, nothing else
                static_for<IPW>([&](auto dc) __attribute__((always_inline)) { dma_piece(wr_off, dc); });
            } else {                     // compute waves: fragment reads pinned into the MFMA slots, no memory instruction
                static_for<NRD>([&](auto rc) __attribute__((always_inline)) { read_frag(rd_off, std::integral_constant<int, 0>{}, rc); });
                PL_SB();
                static_for<SLOTS>([&](auto gc) __attribute__((always_inline)) {
                    constexpr int g = decltype(gc)::value, s = g / NMF, m = g % NMF;
                    mfma_one(std::integral_constant<int, s>{}, std::integral_constant<int, m>{}, parc);
                    if constexpr (s + 1 < NSTEP && m < HALF) {
                        constexpr int r_lo = (m * NRD + HALF - 1) / HALF, r_hi = ((m + 1) * NRD + HALF - 1) / HALF;
                        static_for<r_hi - r_lo>([&](auto k) __attribute__((always_inline)) {
                            read_frag(rd_off, std::integral_constant<int, s + 1>{}, std::integral_constant<int, r_lo + decltype(k)::value>{});
                        });
                    }
                    PL_SB();
                });
            }
            // BDIR: the registers this stage multiplied from take the filter fragments of the chunk two stages on (requested
            // behind the stage's last MFMA: the loads return long after it has read its operands)
            if constexpr (BDIR) { load_b(parc); PL_SB(); }
            wr_off = (wr_off + STAGE == NST * STAGE) ? 0 : wr_off + STAGE;
            rd_off = (rd_off + STAGE == NST * STAGE) ? 0 : rd_off + STAGE;
            ++it;
        };
        if constexpr (BDIR) {     // two stages per trip: the filter fragment registers swap roles at compile time
            for (;;) {
                stage_body(std::integral_constant<int, 0>{});
                if (it >= n_it) break;
                stage_body(std::integral_constant<int, 1>{});
                if (it >= n_it) break;
            }
        } else {
            do { stage_body(std::integral_constant<int, 0>{}); } while (it < n_it);
        }
    }
    // the ring (still receiving the out-of-range tail issues) becomes the epilogue's staging tile
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    PL_STAMP(3);   // K loop done
    if constexpr (KG > 1) {
        // groups 1 .. KG-1 park their accumulators in LDS (fragment order: 16 B per lane and instruction), group 0 adds them
        // in group order -- the same sum chain whichever group finished first
        char* const kb = reinterpret_cast<char*>(smem) + (wv * (TM * TN * 4)) * 1024 + lane * 16;
        if (kgrp > 0)
            static_for<TM * TN * 4>([&](auto ec) __attribute__((always_inline)) {
                constexpr int t = decltype(ec)::value / 4, q = decltype(ec)::value % 4;
                f32x4 v;
                v.x = acc[t / TN][t % TN][4 * q]; v.y = acc[t / TN][t % TN][4 * q + 1]; v.z = acc[t / TN][t % TN][4 * q + 2]; v.w = acc[t / TN][t % TN][4 * q + 3];
                *reinterpret_cast<f32x4*>(kb + ((kgrp - 1) * NWC * TM * TN * 4 + t * 4 + q) * 1024) = v;
            });
        __syncthreads();
        if (kgrp == 0)
            static_for<(KG - 1) * TM * TN * 4>([&](auto ec) __attribute__((always_inline)) {
                constexpr int g = decltype(ec)::value / (TM * TN * 4), t = (decltype(ec)::value / 4) % (TM * TN), q = decltype(ec)::value % 4;
                const f32x4 v = *reinterpret_cast<const f32x4*>(kb + (g * NWC * TM * TN * 4 + t * 4 + q) * 1024);
                acc[t / TN][t % TN][4 * q] += v.x; acc[t / TN][t % TN][4 * q + 1] += v.y; acc[t / TN][t % TN][4 * q + 2] += v.z; acc[t / TN][t % TN][4 * q + 3] += v.w;
            });
        __syncthreads();
    }
    if (p.stamps && tid == 0) p.stamps[(long long)(p.hy_splits > 0 ? (int)blockIdx.x : tile_id * p.splits + split) * 8 + 7] = t_entry + t_wait;   // (read as a duration)

    const int w_row0 = wm * (32 * TM), w_col0 = wn * (32 * TN);
#define BP_NT NT
#define BP_SLAST (reinterpret_cast<int*>(smem)[SMEM_BYTES / 4 - 1])
#define BP_EARLY_BIAS bias_early
#define BP_TAIL_STAMP(k_) PL_STAMP(k_)
#define BP_EP_SLABS EP_SLABS_
#define BP_HAS_ACC has_acc
#define BP_SPLITS my_splits
#define BP_SLAB_TILE (p.hy_splits > 0 ? tile_id - p.hy_full : tile_id)
#define BP_SLAB_TILES (p.hy_splits > 0 ? p.n_tiles - p.hy_full : (p.n_tiles ? p.n_tiles : (int)gridDim.x / p.splits))
#define BP_EP_RES_SCALE true           /* the SE blocks' downsample layers in the fp16 modes */
#define BP_EP_UP2                      /* the detector's two upsampling 1x1 layers */
#define BP_EP_PIXSHUF                  /* the DUC layers' PixelShuffle stores through the staged epilogue (round 5: they ran 229 us each at batch 28 on the element-wise path) */
#include "conv_tail.inc"
#undef BP_EP_PIXSHUF
#undef BP_EP_UP2
#undef BP_EP_RES_SCALE
#undef BP_SPLITS
#undef BP_SLAB_TILE
#undef BP_SLAB_TILES
#undef BP_HAS_ACC
#undef BP_EP_SLABS
#undef BP_NT
#undef BP_SLAST
#undef BP_EARLY_BIAS
#undef BP_TAIL_STAMP
    if (p.stamps) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); PL_STAMP(4); }
#undef PL_STAMP
#undef PL_SB
}

bool conv_tile_is_pl(int tile) {
#ifdef BP_EXPERIMENTAL
    if (tile == TILE_PL128S || tile == TILE_PL64K2 || tile == TILE_PL64BD) return true;
#endif
    return tile == TILE_PL64 || tile == TILE_PL128 || tile == TILE_PL128x64 || tile == TILE_PL256x128 || conv_tile_is_plh(tile) || tile == TILE_S1 || tile == TILE_P3;
}

bool conv_plh_eligible(const ConvParams& p) {
    return conv_pl_eligible(p) && p.mfma_mode == PREC_F16 && p.ksize == 3 && p.stride == 1 && p.pad == 1 && p.OH == p.H && p.OW == p.W &&
           p.W <= 126 && p.nchunks % 9 == 0 && p.Kpad == 9 * p.Cin;
}

bool conv_pl_eligible(const ConvParams& p) {
    return p.in16 != nullptr && (p.wpl != nullptr || p.wbd != nullptr) && (p.Cin % 32 == 0) && (p.in_ld % 8 == 0) && p.ksize * p.ksize <= 32 &&
           ((reinterpret_cast<uintptr_t>(p.in16) & 15) == 0) && (p.in16_plane % 8 == 0);
}

template <int NP, int WM, int WN, int TM, int TN, int NST, int CPS, int LW = 0, int KG = 1, bool BDIR = false, int HRT = 0, bool B3 = false>
static void launch_pl_t(const ConvParams& p, hipStream_t s) {
    BP_CHECK(BDIR ? p.wbd != nullptr : p.wpl != nullptr, "conv_pl: the filter image of this tile is missing");
    constexpr int BM = 32 * TM * WM, BN = 32 * TN * WN;
    BP_CHECK(LW == 0 || p.splits == 1, "the wave-specialised operand-plane tiles do not take K slices");
    BP_CHECK(p.hy_splits == 0 || (p.splits == 1 && !p.xcd_home && LW == 0 && KG == 1 && p.partial != nullptr && p.tickets != nullptr && p.hy_splits >= 2 && p.hy_cps >= 1),
             "hybrid grid: a one-slice launch with a split-K workspace");
    ConvParams q = p;
    conv_grid_setup(q, BM, BN);
    BP_CHECK(q.hy_splits == 0 || (q.hy_full >= 0 && q.hy_full < q.n_tiles), "hybrid grid: whole tiles out of range");
    dim3 grid(conv_grid_blocks(q));
    if (g_conv_prof)
        hipExtLaunchKernelGGL((conv_pl_kernel<NP, WM, WN, TM, TN, NST, CPS, LW, KG, BDIR, HRT, B3>), grid, dim3(64 * (WM * WN * KG + LW)), 0, s, g_conv_prof->e0, g_conv_prof->e1, 0, q);
    else
        hipLaunchKernelGGL((conv_pl_kernel<NP, WM, WN, TM, TN, NST, CPS, LW, KG, BDIR, HRT, B3>), grid, dim3(64 * (WM * WN * KG + LW)), 0, s, q);
}

template <int NP>
static void launch_pl_np(const ConvParams& p, int tile, hipStream_t s) {
    switch (tile) {
        case TILE_PL64: launch_pl_t<NP, 2, 2, 1, 1, 3, NP == 1 ? 2 : 1>(p, s); break;
        case TILE_PL128: launch_pl_t<NP, 2, 2, 2, 2, NP == 1 ? 4 : 3, 1>(p, s); break;
        case TILE_PL128x64: launch_pl_t<NP, 2, 2, 2, 1, 3, NP == 1 ? 2 : 1>(p, s); break;
        case TILE_PL256x128: launch_pl_t<NP, 4, 2, 2, 2, NP == 1 ? 3 : 2, 1>(p, s); break;
        case TILE_PLH128:     // 128x128 with the activations from an LDS-resident halo (fp16, 3x3 / stride 1 / pad 1, W <= 63)
            if constexpr (NP == 1) {
                BP_CHECK(conv_plh_eligible(p) && (p.splits == 1 || p.chunks_per_split % 9 == 0) && !p.xcd_home && p.hy_splits == 0,
                         "halo plane tile: fp16 mode, 3x3 / stride 1 / pad 1, W <= 126, K slices of whole channel groups, plain grid");
                // maps up to 31 wide whose grid is MORE than two blocks per CU: three blocks per CU (template parameter B3: 3-deep ring, two epilogue
                // slabs, 154 registers).  Batch 28, one launch at a time: 26x26 256 -> 512 (592 blocks) 61-67 -> 52-55 us, DUC1 (560) 111 -> 100;
                // grids that fit two per CU anyway lose the fourth ring stage for nothing (13x13 512 -> 1024, 296 blocks: 63 -> 63; 20x16 256 -> 256,
                // 140 blocks: 26.6 -> 29.8) and keep the form below.  BP_NO_PLH_B3=1: off (A/B runs; read per call)
                const long long tiles128 = (((long long)p.M + 127) / 128) * ((p.CoutPad + 127) / 128);
                if (p.W <= 31 && tiles128 > 512 && !std::getenv("BP_NO_PLH_B3")) launch_pl_t<1, 2, 2, 2, 2, 3, 1, 0, 1, false, 192, true>(p, s);
                else if (p.W <= 31) launch_pl_t<1, 2, 2, 2, 2, 4, 1, 0, 1, false, 192>(p, s);
                // ... and maps up to 63 wide (halos of 2 x 16.4 KB): three blocks per CU only with a TWO-deep ring.  One launch at a time it is a tie
                // (52x52 128 -> 256, 1 184 blocks: 60-73 -> 56-73 us by layer), with three streams in flight +0.7 % (5 335 -> 5 375 frames/s,
                // profiles/r05_ab_plh_b3.txt): taken
                else if (p.W <= 63 && tiles128 > 512 && !std::getenv("BP_NO_PLH_B3")) launch_pl_t<1, 2, 2, 2, 2, 2, 1, 0, 1, false, 256, true>(p, s);
                else if (p.W <= 63) launch_pl_t<1, 2, 2, 2, 2, 4, 1, 0, 1, false, 256>(p, s);
                // (round 5) maps up to 126 wide -- the detector's two 104x104 3x3 layers: 384 halo rows (a 128-pixel strip reads 338 input pixels,
                // 2.6x instead of 9x), and a 3-deep filter ring so that two blocks still fit a CU (2 x 24.6 KB of halo + 24 KB)
                else launch_pl_t<1, 2, 2, 2, 2, 3, 1, 0, 1, false, 384>(p, s);
            } else throw Error("the halo plane tile is an fp16 tile");
            break;
        // (round 5: the halo form with 128 x 64 outputs per wave -- 2x2 waves on a 256x128 block, <1, 2, 2, 4, 2, 4 | 3, 1, 0, 1, false, 320 | 384>,
        // an A / B fragment pair feeding 8 MFMAs instead of 4 -- was built on this template, parity-green, and is SLOWER: 204 VGPRs + 128
        // accumulator registers leave one wave per SIMD, 97.7 / 93.8 / 77.1 us against 66.6 / 63.4 / 66.6 us for the 52x52 / 26x26 / 13x13
        // layers at batch 28, profiles/r05_plh_kernels.txt; removed)
#ifdef BP_EXPERIMENTAL   // two round-3 forms that are parity-green and bring nothing:
        // planes for the activations + filter fragments direct from global memory (BDIR): 4 blocks per CU and no in-kernel
        // operand split, but both waves of a column pair pull the same fragments through the vector-memory path -- alone
        // within 10 % of the all-DMA tile, in the pipeline 778 against 900 frames/s (K loops 25-35 % longer, in-situ stamps)
        case TILE_PL64BD:
            if constexpr (NP == 3) launch_pl_t<3, 2, 2, 1, 1, 3, 1, 0, 1, true>(p, s);
            else throw Error("the filters-direct plane tile is a bf16x3 tile");
            break;
        // K groups inside the block (8 waves, two rings): the M <= 1 280 layers at 7.8-24.2 us against 7.7-25.0 us
        case TILE_PL64K2: launch_pl_t<NP, 2, 2, 1, 1, 3, NP == 1 ? 2 : 1, 0, 2>(p, s); break;
        // wave specialisation (4 loader waves): fp16 batch-28 3x3 115 us against 76 us
        case TILE_PL128S: launch_pl_t<NP, 2, 2, 2, 2, NP == 1 ? 4 : 3, NP == 1 ? 2 : 1, 4>(p, s); break;
#endif
        default: throw Error("not a conv_pl tile");
    }
}

void launch_conv_pl(const ConvParams& p, int tile, hipStream_t s) {
    BP_CHECK(conv_pl_eligible(p), "layer is not eligible for the operand-plane kernels (needs input planes, packed filters, Cin % 32 == 0, k*k <= 32)");
    BP_CHECK((long long)3 * p.CoutPad * p.Kpad * 2 < (long long)OOB, "filter planes too large for 32-bit offsets");
    BP_CHECK((long long)p.N * p.H * p.W * p.in_ld * 2 < (long long)OOB, "activation planes too large for 32-bit offsets");
    if (tile == TILE_S1) { launch_conv_s1(p, s); return; }     // conv_s1.hip
    if (tile == TILE_P3) { launch_conv_p3(p, s); return; }     // conv_p3.hip
    if (p.mfma_mode == PREC_BF16X3) launch_pl_np<3>(p, tile, s);
    else if (p.mfma_mode == PREC_F16) launch_pl_np<1>(p, tile, s);
    else throw Error("conv_pl tiles need a 16-bit precision mode");
}

// ---- filters fp32 [CoutPad][Kpad] (K order ky, kx, ci) -> the LDS image above in the kernel's K order (channel group, ky, kx, ci % 32)
template <int NP>
__global__ void pack_wpl_kernel(const float* __restrict__ in, unsigned short* __restrict__ out, int CoutPad, int Kpad, int Cin, int taps) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)CoutPad * Kpad) return;
    const int n = (int)(i / Kpad), k = (int)(i - (long long)n * Kpad);
    const int tap = k / Cin, ci = k - tap * Cin;
    const int t64 = n >> 6, r = n & 63, c = (ci >> 5) * taps + tap, kk = ci & 31;
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

void launch_pack_wpl(const float* in, unsigned short* out, int CoutPad, int Kpad, int Cin, int ksize, int np, hipStream_t s) {
    const long long n = (long long)CoutPad * Kpad;
    BP_CHECK(CoutPad % 64 == 0 && Cin % 32 == 0 && Kpad == ksize * ksize * Cin, "pack_wpl: filter shape");
    if (np == 1) hipLaunchKernelGGL(pack_wpl_kernel<1>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, in, out, CoutPad, Kpad, Cin, ksize * ksize);
    else hipLaunchKernelGGL(pack_wpl_kernel<3>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, in, out, CoutPad, Kpad, Cin, ksize * ksize);
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

__global__ void planes_to_f32_kernel(const unsigned short* __restrict__ planes, long long plane_elems, int np, float* __restrict__ out,
                                     int ld, int pixels, int C) {
    const int total = pixels * C;
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
        const int c = e % C, px = e / C;
        const long long idx = (long long)px * ld + c;
        float v;
        if (np == 1) {
            v = (float)__builtin_bit_cast(_Float16, planes[idx]);
        } else {
            const float a = __uint_as_float((unsigned)planes[idx] << 16), b = __uint_as_float((unsigned)planes[plane_elems + idx] << 16),
                        c3 = __uint_as_float((unsigned)planes[2 * plane_elems + idx] << 16);
            v = (a + b) + c3;      // exact: 8 + 8 + 8 significand bits
        }
        out[idx] = v;
    }
}

void launch_planes_to_f32(const unsigned short* planes, long long plane_elems, int np, float* out, int ld, long long pixels, int C,
                          hipStream_t s) {
    BP_CHECK(pixels * C < (1ll << 31) && (np == 1 || np == 3), "planes_to_f32: arguments");
    const int grid = (int)std::min<long long>((pixels * C + 255) / 256, 4096);
    hipLaunchKernelGGL(planes_to_f32_kernel, dim3(grid < 1 ? 1 : grid), dim3(256), 0, s, planes, plane_elems, np, out, ld, (int)pixels, C);
}

}  // namespace bp
