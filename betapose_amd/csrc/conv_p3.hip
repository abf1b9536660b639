// The 3x3 / stride-1 convolutions of the BATCHED fp16 runs (BASELINE configs[2]: 28 frames per launch) as a PERSISTENT kernel (round 6).
//
// Same contract as conv_pl.hip's halo tile (TILE_PLH128) for the layers it takes (3x3, stride 1, pad 1, NHWC store; bias / LeakyReLU / ReLU /
// skip connection before or after the activation; fp16 plane and / or fp32 tensor out; yolo/darknet.py:240-259 + the shortcut of :338-340,
// SE_Resnet.py:12-14 conv2), same operands (the producer's fp16 plane, conv_pl.hip's packed filter image), the same MFMA sequence per output
// element ((32-channel group, tap, k-step) in order, bias added to the finished fp32 sum) -- so its results are bit-identical to TILE_PLH128's.
//
// Why (round-5 verdict item 1; profiles/r05_per_op_b28_f16r.txt, r05_pmc_mfma_busy_batch28_f16r.json): at batch 28 these layers are the
// compute-bound class (AI 740-2300) and ran at 25-33 % of the fp16 matrix roof.  A block of the halo plane tile lives ~25 us of which its 36-144
// stages are 10-20: index math, first operands cold, an LDS-staged epilogue -- and inside the K loop every 8 MFMAs of a wave sit behind a
// block-wide barrier, two filter LDS-DMAs (60-180 cycles of issue each, in order) and eight fragment reads (measured per stage: ~840 cycles
// with one wave per SIMD, ~1 280 with three, for 256 cycles of matrix work).  The fp32-accurate halo kernel (conv_halo.hip) drives TWICE the matrix
// work per second on the same layers with a different skeleton -- filter fragments global -> registers, one barrier per channel group -- so
// this kernel is that skeleton for fp16 planes, made persistent:
//   * block = 128 consecutive output pixels x 128 columns, four waves, each ALL 128 pixels x 32 columns (four 32x32 MFMA tiles): a filter
//     fragment (global -> registers, three-deep ring two taps ahead, conv_pl.hip's packed image read in place) feeds four MFMAs and no two
//     waves fetch the same fragment; the activations of a 32-channel group -- the 128 pixels' halo -- are fetched ONCE (global -> registers ->
//     LDS under the previous group's MFMAs), the nine taps read them at compile-time immediate offsets;
//   * the halo in LDS is ZERO-PADDED: position q = (image b, padded row, padded column) of a [N (H + 1) + 1][W + 2] grid (one shared zero row
//     between images, a zero column either side), rows of 80 B (64 B of channels + 16: any 16 rows a ds_read_b128 lane group touches cover the
//     64 banks once).  A tap is then the SAME shift of (ky (W + 2) + kx) positions for every pixel -- an instruction immediate, W being a
//     template parameter: four address registers per lane for the whole K loop, no per-tap address math, no border masks;
//   * ONE block-wide barrier per channel group (72 MFMAs per wave), nothing in the K loop waits for LDS-DMA (there is none);
//   * the MFMA operands are SWAPPED (filters as the row operand): the accumulator then holds, per lane, four consecutive CHANNELS of one
//     pixel -- the epilogue is wave-private and register-only (bias from LDS, activation, skip connection, 8-B fp16 / 16-B fp32 stores straight
//     from the accumulators): no staging tile, no barrier, no LDS;
//   * PERSISTENT over tiles (block -> XCD -> a contiguous range of tiles, N-tiles of an M-tile next to each other): the next tile's first
//     filter fragments and first halo are requested during the current tile's last channel group and land while its epilogue runs; the
//     skip-connection rows (fp16 plane) are requested four taps before the epilogue needs them.
#include <algorithm>
#include <cstdlib>

#include "conv_dev.h"

// timing ablations (tools/p3_variants.sh builds conv_p3.o with -DP3_ABL=n into its own library; WRONG results): 1 no filter loads in the K
// loop, 2 no fragment reads, 4 no MFMAs, 8 no halo loads / parks, 16 no epilogue stores
#ifndef P3_ABL
#define P3_ABL 0
#endif

namespace bp {

struct P3Args {
    int NTN;      // N tiles of 128 columns
    int T;        // tiles = M tiles x N tiles (tile t: M tile t / NTN, N tile t % NTN)
    int G;        // 32-channel groups of the layer's K
    int PX;       // blocks per XCD (grid = 8 PX): block (xcd, j) takes the tiles lo(xcd) + j, + PX, ... of the XCD's range [lo, hi)
    int skew;     // start skew: block j of its XCD sleeps (j % 8) x skew x 64 cycles before its first request (0: none)
};

#ifndef P3_ROWX_B
#define P3_ROWX_B 96      // (-DP3_ROWX_B=0: the 3x3 form's halo without the per-row skew, A/B builds of tools/p3_variants.sh)
#endif
static constexpr int P3_BM = 128, P3_BN = 128, P3_PITCH = 80, P3_ROWX = P3_ROWX_B;
static constexpr int P3_SOOB = 0x40000000;      // a scalar offset beyond every descriptor (adding a few KB to it does not wrap)

// KSZ: 3 (3x3 / stride 1 / pad 1) or 1 (1x1 / stride 1: the "halo" of a 128-channel group is the tile's own 128 pixels, the "taps" are its four
// 32-channel chunks).  W: map width of the 3x3 form (tap shifts are instruction immediates).  HRT: halo rows (positions) per stage.
// RES: 0 no skip connection, 1 from its fp16 plane (ConvParams::res16), 2 from the fp32 tensor.  BPC: blocks per CU the register budget is asked
// for (2: <= 256 registers, the fp16 skip connection requested four taps ahead of the epilogue; 3: <= 168, requested inside the epilogue).
template <int KSZ, int W, int HRT, int RES, int BPC>
__global__ __launch_bounds__(256, BPC) void conv_p3_kernel(const ConvParams p, const P3Args a) {
    constexpr bool K3 = KSZ == 3;
    constexpr int WP = W + 2;
    constexpr int TAPS = K3 ? 9 : 4;                      // 32-k chunks per group
    constexpr int GR = K3 ? 4 : 16;                       // 16-B granules of a halo row (32 / 128 channels)
    constexpr int PITCH = K3 ? P3_PITCH : 272;            // row bytes + 16: conflict-free for any 16 rows of a ds_read_b128 lane group
    // 3x3 form: a halo position's LDS address is position x 80 B + grid row x 96 B.  Consecutive positions are 5 bank groups (of 16 B) apart, so
    // any 16 positions that differ mod 16 -- a ds_read_b128 lane group's, whatever the lanes -- cover the 64 banks once; the two pad columns a
    // row change skips would break that (+ 10 groups), the 96 B per row restore it (+ 16: round 6, PMC: SQ_LDS_BANK_CONFLICT was half of the
    // LDS-array cycles at W = 13 / 16 / 26, profiles/r06_pmc_p3.txt)
    constexpr int ROWX = K3 ? P3_ROWX : 0;
    constexpr int STAGE = HRT * PITCH + (K3 ? (HRT / WP + 2) * ROWX : 0);
    constexpr int RPP = 256 / GR;                         // halo rows per loader pass (256 threads x 16 B)
    constexpr int NPASS = (HRT + RPP - 1) / RPP;
    constexpr int LASTSTEP = 2 * TAPS - 1;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    float* const ldsBias = reinterpret_cast<float*>(lds + 2 * STAGE);
    const int tid = threadIdx.x, lane = tid & 63;
    const unsigned long long t_entry = p.stamps ? bp_clock() : 0ull;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int xcd = (int)blockIdx.x & 7, bj = (int)blockIdx.x >> 3;
    const int t_lo = (int)((long long)xcd * a.T >> 3), t_hi = (int)((long long)(xcd + 1) * a.T >> 3);
    int tile = t_lo + bj;
    if (tile >= t_hi) return;
    for (int i = (bj & 7) * a.skew; i > 0; --i) __builtin_amdgcn_s_sleep(1);     // (BP_P3_SKEW = 1 .. 8, measured: no difference -- these launches sit at the power cap)

    const int H = p.H, hw = p.H * p.W, H1 = H + 1;
    const float rcp_hw = 1.0f / (float)hw, rcp_h1 = 1.0f / (float)H1;
    const int G = a.G;

    const __amdgpu_buffer_rsrc_t rsrcA = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<unsigned short*>(p.in16), 0, (int)min((long long)p.N * hw * p.in_ld * 2, (long long)OOB), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrcB = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<unsigned short*>(p.wpl), 0, (int)((long long)p.CoutPad * p.Kpad * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrcO =
        __builtin_amdgcn_make_buffer_rsrc(p.out, 0, (int)min((long long)p.M * p.out_ld * 4, (long long)OOB), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrcR = __builtin_amdgcn_make_buffer_rsrc(
        RES == 1 ? (void*)const_cast<unsigned short*>(p.res16) : (void*)const_cast<float*>(RES == 2 ? p.res : p.out), 0,
        RES ? (int)min((long long)p.M * p.res_ld * (RES == 1 ? 2 : 4), (long long)OOB) : 0, 0x00020000);
    const PlaneDesc pd = make_plane_desc(p);

    // the layer's bias (padded to CoutPad by the engine) -> LDS, once per block
    for (int i = tid; i < a.NTN * P3_BN; i += 256) ldsBias[i] = i < p.CoutPad ? p.bias[i] : 0.f;

    // centre position of output pixel m = (b, oy, ox) in the padded grid (3x3 form)
    auto cpos = [&](int m) __attribute__((always_inline)) {
        const int b = fast_div(m, hw, rcp_hw);
        const int rem = m - b * hw;
        const int oy = rem / (K3 ? W : 1), ox = rem - oy * (K3 ? W : 1);
        return (b * H1 + oy + 1) * WP + ox + 1;
    };
    // ---- per tile: the loader's source offsets (3x3: halo row j <-> position q0 + j, q0 = the centre of the tile's first pixel - (W + 2) - 1;
    // 1x1: halo row j = pixel m0 + j), the lane's four fragment rows, the wave's filter base
    const int lrow = tid / GR, lgr = tid % GR;
    auto tile_setup = [&](int t, bool live, int& m0, int& n0, unsigned (&avo)[NPASS], unsigned (&awo)[NPASS], unsigned (&bse)[4], int& bsrc) __attribute__((always_inline)) {
        const int tm = t / a.NTN, tn = t - tm * a.NTN;
        m0 = tm * P3_BM;
        n0 = tn * P3_BN;
        if constexpr (K3) {
            const int q0 = cpos(min(m0, p.M - 1)) - WP - 1;
            const int R0 = q0 / WP;
#pragma unroll
            for (int k = 0; k < NPASS; ++k) {
                const int j = lrow + RPP * k, q = q0 + j;
                const int qq = max(q, 0);
                const int R = qq / WP, col = qq - R * WP;
                awo[k] = (unsigned)(j * PITCH + (R - R0) * ROWX + lgr * 16);
                const int b = fast_div(R, H1, rcp_h1), rr = R - b * H1;
                const bool ok = live & (q >= 0) & (j < HRT) & (col >= 1) & (col <= W) & (rr >= 1) & (b < p.N);
                const unsigned so = (unsigned)((((b * H + rr - 1) * W + col - 1) * p.in_ld + lgr * 8) * 2);
                avo[k] = ok ? so : OOB;
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int m = min(m0 + 32 * e + (lane & 31), p.M - 1);
                // tap (ky, kx) of pixel m reads position cpos(m) + (ky - 1)(W + 2) + (kx - 1) = halo row (cpos(m) - (W + 2) - 1 - q0) + ky (W + 2) + kx
                const int P = cpos(m) - WP - 1;
                bse[e] = (unsigned)((P - q0) * PITCH + (P / WP - R0) * ROWX + (lane >> 5) * 16);
            }
        } else {
#pragma unroll
            for (int k = 0; k < NPASS; ++k) {
                const int m = m0 + lrow + RPP * k;
                const bool ok = live & (m < p.M);
                const unsigned so = (unsigned)((m * p.in_ld + lgr * 8) * 2);
                avo[k] = ok ? so : OOB;
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) bse[e] = (unsigned)((32 * e + (lane & 31)) * PITCH + (lane >> 5) * 16);
        }
        const int col0 = n0 + 32 * wave;
        bsrc = live ? ((col0 >> 6) * p.nchunks) * 4096 + ((col0 >> 5) & 1) * 2048 : P3_SOOB;
    };

    // filter fragment geometry (conv_pl.hip's packed image, [CoutPad / 64][chunk][64 rows][64 B], granule g of row r at slot g ^ ((r >> 2) & 3)):
    // lane -> row (lane & 31) of the wave's 32, logical granule 2 ks + (lane >> 5)
    const int fsw = ((lane & 31) >> 2) & 3;
    const int bfr0 = (lane & 31) * 64 + (((lane >> 5) ^ fsw) << 4), bfr1 = bfr0 ^ 32;

    constexpr int NSLOT = K3 ? 3 : 4;         // filter ring, two taps ahead: tap t of a group in slot t % NSLOT (9 % 3 == 0, 4 % 4 == 0: every group starts at slot 0)
    u32x4 rb[NSLOT][2];              // [slot][k-step]
    auto load_b = [&](auto slotc, int so) __attribute__((always_inline)) {
        constexpr int slot = decltype(slotc)::value;
        rb[slot][0] = __builtin_amdgcn_raw_buffer_load_b128(rsrcB, bfr0, so, 0);
        rb[slot][1] = __builtin_amdgcn_raw_buffer_load_b128(rsrcB, bfr1, so, 0);
    };
    auto load_b_loop = [&](auto slotc, int so) __attribute__((always_inline)) {
        if constexpr (!(P3_ABL & 1)) load_b(slotc, so);
    };
    const unsigned a_woff = (unsigned)(lrow * PITCH + lgr * 16);
    unsigned awo[NPASS];             // 3x3 form: LDS offsets of the thread's halo rows (tile_setup)
    u32x4 ra[NPASS];                 // a group's halo on its way to LDS
    auto load_a = [&](const unsigned (&avo)[NPASS], int so) __attribute__((always_inline)) {
#pragma unroll
        for (int k = 0; k < NPASS; ++k) ra[k] = __builtin_amdgcn_raw_buffer_load_b128(rsrcA, (int)avo[k], so, 0);
    };
    auto park_a = [&](auto kc, unsigned so) __attribute__((always_inline)) {
        constexpr int k = decltype(kc)::value;
        // (3x3 form: the row's LDS offset carries its grid row's 96 B, rebuilt per tile; 1x1 form: rows are the tile's pixels, a fixed stride)
        const unsigned wo = K3 ? awo[k] : a_woff + k * (RPP * PITCH);
        if (k < NPASS - 1 || lrow + RPP * k < HRT) *reinterpret_cast<u32x4*>(lds + so + wo) = ra[k];
    };

    int m0, n0, bsrc, m0n = 0, n0n = 0, bsrcn = P3_SOOB;
    unsigned avo[NPASS], bse[4], bsen[4];
    tile_setup(tile, true, m0, n0, avo, awo, bse, bsrc);
    // the first two taps' filters, the first group's halo
    load_b(std::integral_constant<int, 0>{}, bsrc);
    load_b(std::integral_constant<int, 1>{}, bsrc + 4096);
    load_a(avo, 0);
    __builtin_amdgcn_sched_barrier(0);

    f16x8 fr[2][4];                  // activation fragments: [step parity][pixel sub-tile]
    auto read_frags = [&](auto parc, auto tapc, auto ksc, unsigned so, const unsigned (&b4)[4]) __attribute__((always_inline)) {
        constexpr int par = decltype(parc)::value, tap = decltype(tapc)::value, ks = decltype(ksc)::value;
        constexpr int imm = K3 ? ((tap / 3) * WP + (tap % 3)) * PITCH + (tap / 3) * ROWX + ks * 32 : tap * 64 + ks * 32;
#pragma unroll
        for (int e = 0; e < 4; ++e) fr[par][e] = *reinterpret_cast<const f16x8*>(lds + so + b4[e] + imm);
    };

    static_for<NPASS>([&](auto kc) __attribute__((always_inline)) { park_a(kc, 0u); });
    // 1x1 form: a group is four taps (~0.5-1.5 us), shorter than a cold fetch -- the halo of group g + 2 is requested at the start of group g,
    // waits in registers for a whole group and is parked at the start of group g + 1 (two LDS stages, as in the 3x3 form)
    if constexpr (!K3) load_a(avo, G > 1 ? 256 : (int)OOB);
    __syncthreads();
    read_frags(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, 0u, bse);

    // debug marks (ConvParams::stamps, tools/bench_pl.py-style BP_CONV_STAMPS=1 through bp_conv2d): per BLOCK, 10-ns ticks -- 0 entry | 1 first
    // fragments read | 2 first group done | 3 first tile's K loop done | 7 its epilogue issued | 5 / 6 the same for the block's second tile | 4 end
    unsigned long long* const stm = (p.stamps && tid == 0) ? p.stamps + (long long)blockIdx.x * 8 : nullptr;
    if (stm) { stm[0] = t_entry; stm[1] = bp_clock(); }
    int tiles_done = 0;
    f32x16 acc[4];
    unsigned so_cur = 0u, so_nxt = (unsigned)STAGE;
    // skip-connection rows of the tile (fp16 plane): [pixel sub-tile][pair of channel quads], 16 B = the lane's pixel x 8 consecutive channels
    // (the same lane-half exchange as the stores, see the epilogue).  BPC == 2: requested inside the tile's last channel group
    constexpr bool RPRE = RES == 1 && BPC == 2;
    u32x4 rr16[RPRE ? 4 : 1][2];
    auto load_res16 = [&](int e, int j) __attribute__((always_inline)) {
        const int m = m0 + 32 * e + (lane & 31), ch = n0 + 32 * wave + 8 * (2 * j + (lane >> 5));
        const bool ok = (m < p.M) & (ch < p.Cout);          // (bitwise: a short-circuit && became control flow around every load)
        const int ro = (m * p.res_ld + ch) * 2;
        return __builtin_amdgcn_raw_buffer_load_b128(rsrcR, ok ? ro : (int)OOB, 0, 0);
    };
    // fp32 skip connection (RES == 2): a pixel sub-tile's four quads, requested one sub-tile ahead (the first inside the last channel group)
    f32x4 rq[RES == 2 ? 2 : 1][4];
    auto load_res32 = [&](int e, int q) __attribute__((always_inline)) {
        const int m = m0 + 32 * e + (lane & 31), ch = n0 + 32 * wave + 8 * q + 4 * (lane >> 5);
        const bool ok = (m < p.M) & (ch < p.Cout);
        const unsigned ro = (unsigned)((m * p.res_ld + ch) * 4);
        return buf_load4(rsrcR, ok ? ro : OOB, 0);
    };
    int next = 0;
    bool has_next = false;

    // one channel group: TAPS taps.  POS = 2: the tile's last group (the next item is the NEXT tile's group 0); POS = 1: the one before it
    // (1x1 form: the next tile's first halo is requested here); POS = 0: any other.
    auto group = [&](auto posc, int g) __attribute__((always_inline)) {
        constexpr int POS = decltype(posc)::value;
        constexpr bool LAST = POS == 2;
        if constexpr (K3 ? LAST : (POS == 1)) {
            // the next tile's geometry: this tile's halos have all been requested, so the loader offsets are rebuilt in place
            tile_setup(has_next ? next : tile, has_next, m0n, n0n, avo, awo, bsen, bsrcn);
        }
        const int bs_cur = bsrc + g * (TAPS * 4096);
        const int bs_nxt = LAST ? bsrcn : bs_cur + TAPS * 4096;
        if constexpr (!(P3_ABL & 8)) {
            if constexpr (K3) {
                if constexpr (LAST) load_a(avo, 0);
                else load_a(avo, (g + 1) * 64);
            } else {
                // park group g + 1 (requested a group ago), then request group g + 2
                static_for<NPASS>([&](auto kc) __attribute__((always_inline)) { park_a(kc, so_nxt); });
                if constexpr (POS == 0) load_a(avo, (g + 2) * 256);
                else if constexpr (POS == 1) load_a(avo, 0);                            // the next tile's group 0 (avo rebuilt above)
                else load_a(avo, a.G > 1 ? 256 : (int)OOB);                             // ... its group 1
            }
        }
        static_for<TAPS>([&](auto tapc) __attribute__((always_inline)) {
            constexpr int tap = decltype(tapc)::value;
            constexpr int slot = tap % NSLOT, slot2 = (tap + 2) % NSLOT;
            if constexpr (tap < TAPS - 2) load_b_loop(std::integral_constant<int, slot2>{}, bs_cur + (tap + 2) * 4096);
            else load_b_loop(std::integral_constant<int, slot2>{}, bs_nxt + (tap - (TAPS - 2)) * 4096);
            if constexpr (LAST && RPRE && tap == (TAPS > 4 ? 4 : 1)) {
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int j = 0; j < 2; ++j) rr16[e][j] = load_res16(e, j);
            }
            if constexpr (LAST && RES == 2 && tap == (TAPS > 5 ? 5 : 2)) {
#pragma unroll
                for (int q = 0; q < 4; ++q) rq[0][q] = load_res32(0, q);
            }
            static_for<2>([&](auto ksc) __attribute__((always_inline)) {
                constexpr int ks = decltype(ksc)::value;
                constexpr int step = tap * 2 + ks, par = step & 1;
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (step < LASTSTEP) {
                    if constexpr (!(P3_ABL & 2))
                        read_frags(std::integral_constant<int, par ^ 1>{}, std::integral_constant<int, (step + 1) / 2>{},
                                   std::integral_constant<int, (step + 1) & 1>{}, so_cur, bse);
                } else {
                    // everybody's next halo is parked and nobody reads this stage any more (the last step's fragments are in registers)
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __syncthreads();
                    if constexpr (LAST) read_frags(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, so_nxt, bsen);
                    else read_frags(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, so_nxt, bse);
                }
                __builtin_amdgcn_sched_barrier(0);
                const f16x8 bf = __builtin_bit_cast(f16x8, rb[slot][ks]);
                if constexpr (!(P3_ABL & 4)) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[e] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bf, fr[par][e], acc[e], 0, 0, 0);
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[e][0] += (float)bf[0] + (float)fr[par][e][0];      // (keeps the operands live)
                }
            });
            if constexpr (K3 && tap <= 7 && tap >= 8 - NPASS && !(P3_ABL & 8)) park_a(std::integral_constant<int, tap - (8 - NPASS)>{}, so_nxt);
        });
        const unsigned t_ = so_cur; so_cur = so_nxt; so_nxt = t_;
    };
    for (;;) {
        next = tile + a.PX;
        has_next = next < t_hi;
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[e][r] = 0.f;
        if constexpr (K3) {
            for (int g = 0; g < G - 1; ++g) {
                group(std::integral_constant<int, 0>{}, g);
                if (stm && g == 0 && tiles_done == 0) stm[2] = bp_clock();
            }
            group(std::integral_constant<int, 2>{}, G - 1);
        } else {
            for (int g = 0; g < G - 2; ++g) {
                group(std::integral_constant<int, 0>{}, g);
                if (stm && g == 0 && tiles_done == 0) stm[2] = bp_clock();
            }
            group(std::integral_constant<int, 1>{}, G - 2);
            group(std::integral_constant<int, 2>{}, G - 1);
        }
        if (stm && tiles_done < 2) stm[tiles_done == 0 ? 3 : 5] = bp_clock();

        // ---- epilogue, wave-private, straight from the accumulators.  C/D layout of the 32x32 MFMA with the filters as the row operand:
        // lane -> pixel (lane & 31) of the sub-tile, register r -> channel (r & 3) + 8 (r >> 2) + 4 (lane >> 5) of the wave's 32.
        // (activation and fp32-store as compile-time cases of one switch: as run-time tests inside the 16 (sub-tile, quad) bodies they were
        // ~100 scalar branches per tile)
        auto epilogue = [&](auto actc, auto f32c) __attribute__((always_inline)) {
            constexpr int ACT = decltype(actc)::value;
            constexpr bool F32 = decltype(f32c)::value;
            const int cbase = n0 + 32 * wave + 4 * (lane >> 5);
            const bool after = p.res_after_act != 0, plane = pd.np == 1;
            f32x4 bias4[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) bias4[q] = *reinterpret_cast<const f32x4*>(ldsBias + cbase + 8 * q);
            // (RES == 1, BPC == 3: the skip-connection rows are requested here, one pixel sub-tile ahead of their use)
            u32x4 rl[2][2];
            if constexpr (RES == 1 && !RPRE) { rl[0][0] = load_res16(0, 0); rl[0][1] = load_res16(0, 1); }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int m = m0 + 32 * e + (lane & 31);
                f32x4 r4[4];
                if constexpr (RES == 2) {
                    if (e < 3) {
#pragma unroll
                        for (int q = 0; q < 4; ++q) rq[RES == 2 ? (e + 1) & 1 : 0][q] = load_res32(e + 1, q);
                    }
#pragma unroll
                    for (int q = 0; q < 4; ++q) r4[q] = rq[RES == 2 ? e & 1 : 0][q];
                }
                if constexpr (RES == 1) {
                    if constexpr (!RPRE) { if (e < 3) { rl[(e + 1) & 1][0] = load_res16(e + 1, 0); rl[(e + 1) & 1][1] = load_res16(e + 1, 1); } }
                    // 16 B of the plane = 8 consecutive channels of the lane's pixel: lanes 0-31 hold the quads (2 j, low half | high half), lanes
                    // 32-63 the quads (2 j + 1, low | high); one v_permlane32_swap per dword hands every lane its own two quads' halves
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const u32x4 t = RPRE ? rr16[RPRE ? e : 0][j] : rl[e & 1][j];
                        const auto s0 = __builtin_amdgcn_permlane32_swap(t.x, t.z, false, false);
                        const auto s1 = __builtin_amdgcn_permlane32_swap(t.y, t.w, false, false);
                        r4[2 * j] = __builtin_convertvector(__builtin_bit_cast(f16x4, u32x2{s0[0], s1[0]}), f32x4);
                        r4[2 * j + 1] = __builtin_convertvector(__builtin_bit_cast(f16x4, u32x2{s0[1], s1[1]}), f32x4);
                    }
                }
                u32x2 hq[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int ch = cbase + 8 * q;
                    f32x4 v = {acc[e][4 * q], acc[e][4 * q + 1], acc[e][4 * q + 2], acc[e][4 * q + 3]};
                    v += bias4[q];
                    if constexpr (RES != 0) { if (!after) v += r4[q]; }
                    if constexpr (ACT == ACT_LEAKY) {
                        v.x = v.x > 0.f ? v.x : 0.1f * v.x; v.y = v.y > 0.f ? v.y : 0.1f * v.y;
                        v.z = v.z > 0.f ? v.z : 0.1f * v.z; v.w = v.w > 0.f ? v.w : 0.1f * v.w;
                    } else if constexpr (ACT == ACT_RELU) {
                        v.x = v.x > 0.f ? v.x : 0.f; v.y = v.y > 0.f ? v.y : 0.f;
                        v.z = v.z > 0.f ? v.z : 0.f; v.w = v.w > 0.f ? v.w : 0.f;
                    }
                    if constexpr (RES != 0) { if (after) v += r4[q]; }
                    hq[q] = __builtin_bit_cast(u32x2, __builtin_convertvector(v, f16x4));
                    if constexpr (P3_ABL & 16) { if (v.x == 123.456f) __builtin_amdgcn_raw_buffer_store_b64(hq[q], pd.r0, 0, 0, 0); continue; }
                    if constexpr (F32) {
                        // (rows past M and columns past Cout: out-of-range offsets, dropped by the hardware)
                        const bool ok = (m < p.M) & (ch < p.Cout);
                        const unsigned oo = (unsigned)(m * p.out_ld + ch);
                        const u32x4 o = {__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w)};
                        __builtin_amdgcn_raw_buffer_store_b128(o, rsrcO, (int)((ok ? oo : (OOB >> 2)) * 4), 0, 0);
                    }
                }
                if constexpr (!(P3_ABL & 16)) {
                    if (plane) {
                        // the fp16 plane as 16-B stores (the same exchange: 8 stores per tile and wave instead of 16 -- the store tail of such an
                        // epilogue is issue-bound, MI355X_MICROARCH.md price list)
#pragma unroll
                        for (int j = 0; j < 2; ++j) {
                            const auto s0 = __builtin_amdgcn_permlane32_swap(hq[2 * j].x, hq[2 * j + 1].x, false, false);
                            const auto s1 = __builtin_amdgcn_permlane32_swap(hq[2 * j].y, hq[2 * j + 1].y, false, false);
                            const int ch = n0 + 32 * wave + 8 * (2 * j + (lane >> 5));
                            const bool ok = (m < p.M) & (ch < p.Cout);
                            const unsigned oo = (unsigned)(m * p.out_ld + ch);
                            __builtin_amdgcn_raw_buffer_store_b128(u32x4{s0[0], s1[0], s0[1], s1[1]}, pd.r0, (int)((ok ? oo : (OOB >> 2)) * 2), 0, 0);
                        }
                    }
                }
            }
        };
        {
            const int sel = (p.act == ACT_LEAKY ? 0 : (p.act == ACT_RELU ? 1 : 2)) * 2 + (pd.f32 ? 1 : 0);
            switch (sel) {
                case 0: epilogue(std::integral_constant<int, ACT_LEAKY>{}, std::false_type{}); break;
                case 1: epilogue(std::integral_constant<int, ACT_LEAKY>{}, std::true_type{}); break;
                case 2: epilogue(std::integral_constant<int, ACT_RELU>{}, std::false_type{}); break;
                case 3: epilogue(std::integral_constant<int, ACT_RELU>{}, std::true_type{}); break;
                case 4: epilogue(std::integral_constant<int, ACT_LINEAR>{}, std::false_type{}); break;
                default: epilogue(std::integral_constant<int, ACT_LINEAR>{}, std::true_type{}); break;
            }
        }
        if (stm && tiles_done < 2) stm[tiles_done == 0 ? 7 : 6] = bp_clock();
        ++tiles_done;
        if (!has_next) break;
        tile = next;
        m0 = m0n; n0 = n0n; bsrc = bsrcn;
#pragma unroll
        for (int e = 0; e < 4; ++e) bse[e] = bsen[e];
    }
    if (stm) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); stm[4] = bp_clock(); }
}

// ---- host side
static int p3_width_class(int W) { return (W == 13 || W == 16 || W == 26 || W == 32 || W == 52 || W == 104) ? W : 0; }
static int p3_hrt(int W) { return (W == 13 || W == 16) ? 208 : (W == 52 ? 304 : (W == 104 ? 464 : 240)); }

// halo rows the layer's worst tile needs: centre span of its 128 pixels + a padded row and a pixel either side
static int p3_rows_needed(const ConvParams& p, long long M) {
    const int W = p.W, H = p.H, hw = H * W, WP = W + 2;
    auto cpos = [&](long long m) { const long long b = m / hw, rem = m % hw; return (b * (H + 1) + rem / W + 1) * WP + rem % W + 1; };
    long long need = 0;
    for (long long m0 = 0; m0 < M; m0 += P3_BM) need = std::max(need, cpos(std::min(m0 + P3_BM - 1, M - 1)) - cpos(m0) + 2 * WP + 3);
    return (int)need;
}

bool conv_p3_eligible(const ConvParams& p, long long M) {
    if (!(conv_pl_eligible(p) && p.wpl != nullptr && p.mfma_mode == PREC_F16)) return false;
    if (p.store_mode != ST_NHWC || p.res_scale != nullptr || p.pool_out != nullptr) return false;
    if ((p.Cout & 7) || (p.out_ld & 7) || p.CoutPad < P3_BN) return false;       // (16-B stores of 8 fp16 channels)
    if (p.res && (p.res_ld & 7)) return false;
    if (p.out16 != nullptr && p.out_np != 1) return false;
    if (M * p.out_ld * 4 >= (long long)OOB || (p.res && M * p.res_ld * 4 >= (long long)OOB)) return false;
    if (M >= (1 << 24) || (long long)(p.N * (p.H + 1) + 1) * (p.W + 2) >= (1 << 24)) return false;       // (fast_div's range)
    // a persistent grid pays its start-up over the work it walks: layers of the batched runs only
    if (M < 4096) return false;
    if (p.ksize == 1)      // the 1x1 form: stride 1, whole 128-channel groups, at least two of them
        return p.stride == 1 && p.pad == 0 && p.OH == p.H && p.OW == p.W && p.Kpad == p.Cin && p.Cin % 128 == 0 && p.Cin >= 256;
    return conv_plh_eligible(p) && p3_width_class(p.W) && p3_rows_needed(p, M) <= p3_hrt(p.W);
}

template <int KSZ, int W, int HRT, int BPC>
static void launch_p3_w(const ConvParams& p, const P3Args& a, int grid, int lds_bytes, hipStream_t s) {
    const int res = p.res ? (p.res16 ? 1 : 2) : 0;
#define P3_GO(RES_)                                                                                                                         \
    do {                                                                                                                                    \
        if (lds_bytes > 64 * 1024) allow_big_lds(reinterpret_cast<const void*>(conv_p3_kernel<KSZ, W, HRT, RES_, BPC>));                    \
        if (g_conv_prof)                                                                                                                    \
            hipExtLaunchKernelGGL((conv_p3_kernel<KSZ, W, HRT, RES_, BPC>), dim3(grid), dim3(256), lds_bytes, s, g_conv_prof->e0, g_conv_prof->e1, 0, p, a); \
        else                                                                                                                                \
            hipLaunchKernelGGL((conv_p3_kernel<KSZ, W, HRT, RES_, BPC>), dim3(grid), dim3(256), lds_bytes, s, p, a);                        \
    } while (0)
    if (res == 0) P3_GO(0);
    else if (res == 1) P3_GO(1);
    else P3_GO(2);
#undef P3_GO
}

void launch_conv_p3(const ConvParams& p, hipStream_t s) {
    BP_CHECK(conv_p3_eligible(p, p.M), "conv_p3: fp16 planes, NHWC store, M >= 4096; 3x3 / stride 1 / pad 1 at a width the kernel is built for, or 1x1 / stride 1 with Cin % 128 == 0");
    const bool k3 = p.ksize == 3;
    P3Args a{};
    a.NTN = (p.CoutPad + P3_BN - 1) / P3_BN;
    a.T = ((p.M + P3_BM - 1) / P3_BM) * a.NTN;
    a.G = k3 ? p.Cin / 32 : p.Cin / 128;
    const int lds_bytes = 2 * (k3 ? p3_hrt(p.W) * P3_PITCH + (p3_hrt(p.W) / (p.W + 2) + 2) * P3_ROWX : 128 * 272) + a.NTN * P3_BN * 4;     // two halo stages + the bias (a whole number of N tiles)
    // (Also built and removed in round 6: the 3x3 / STRIDE-2 layers on the same position grid over the input map (centres at (2 oy, 2 ox), taps still
    // immediates, two-way bank conflicts, 736 / 960 halo rows = one block per CU, 12-15 loader passes): 104x104 -> 52x52 63.0 against 67.1 us on the
    // 128x128 plane tile, 52x52 -> 26x26 66.2 against 70.3, 26x26 -> 13x13 83.1 against 70.1 -- one lone-wave block per CU is no better than the tile
    // it would replace; those layers want a 2-D patch / row-segment halo)
    // (Built, parity-green at the accumulation-order bar, and removed in round 6: two K teams inside an eight-wave block for launches of at most
    // one tile per CU -- 20x16 256 -> 256: 140 tiles, 19.2 -> 18.7 us; 1 024 -> 256: 12.2 -> 12.2.  A lone wave already drives 74 % of its SIMD's
    // matrix pipe inside the K loop, so a second wave on the same SIMD can add a quarter at most, and the LDS hand-over takes it back; what these
    // launches lack is the other 116 CUs, i.e. a cross-CU reduction, which costs more than it saves: profiles/r06_plh_splits.txt.)
    // (Built, bit-identical, and removed in round 6: 64-PIXEL tiles (two 32-pixel sub-tiles per wave) for launches whose 128-pixel tiles leave CUs without
    // a block -- the key-point detector's 20x16 maps at 28 frames are 140 tiles for 256 CUs.  280 tiles of half the work each were SLOWER one
    // launch at a time: 3x3 256 -> 256 19.3 -> 26.3 us, 1x1 1 024 -> 256 12.4 -> 14.5, 13x13 1 024 -> 512 12.5 -> 14.6 -- a filter fragment then
    // feeds two MFMAs instead of four and the ring's two taps of prefetch are half as long in time, so the waves wait on L2 where they did not.)
    // blocks per CU: two.  Three (<= 168 registers: the skip connection requested inside the epilogue, a few spills there) tie one launch at a
    // time and LOSE 3 % with three streams in flight (5 550 against 5 380 frames/s, configs[2] f16r on one box): these layers run AT the socket
    // power cap (profiles/r06_p3_clock_probe.txt), a third resident block adds register-file and LDS traffic, not matrix work.
    // BP_P3_BPC = 2 | 3 (A/B runs; read per call; 3x3 form only)
    int bpc = 2;
    if (const char* e = std::getenv("BP_P3_BPC")) bpc = (std::atoi(e) == 3 && k3) ? 3 : 2;
    a.PX = std::max(1, std::min((a.T + 7) / 8, 32 * bpc));
    if (const char* e = std::getenv("BP_P3_SKEW")) a.skew = std::atoi(e);
    const int grid = 8 * a.PX;
    if (!k3) { launch_p3_w<1, 0, 128, 2>(p, a, grid, lds_bytes, s); return; }
#define P3_W(W_, HRT_) do { if (bpc == 3) launch_p3_w<3, W_, HRT_, 3>(p, a, grid, lds_bytes, s); else launch_p3_w<3, W_, HRT_, 2>(p, a, grid, lds_bytes, s); } while (0)
    switch (p.W) {
        case 13: P3_W(13, 208); break;
        case 16: P3_W(16, 208); break;
        case 26: P3_W(26, 240); break;
        case 32: P3_W(32, 240); break;
        case 52: P3_W(52, 304); break;
        case 104: P3_W(104, 464); break;
        default: throw Error("conv_p3: width not instantiated");
    }
#undef P3_W
}

}  // namespace bp
