// 3x3 / stride-1 convolution with a TAP-RESIDENT activation halo (round 4): the fp32-accurate bf16x3 mode's kernel for the
// 3x3 layers of both networks (yolo/darknet.py:240-259 conv + BN + LeakyReLU blocks, cfg/yolov3 3x3 layers;
// KPD/src/models/layers/SE_Resnet.py:13-15,25-42 conv2 of every bottleneck, DUC.py 3x3 + PixelShuffle).
//
// What the round-3 measurements said about the filters-direct kernel (conv_igemm.hip <1,1,3,true>): per 64x64x32 chunk a
// block pulls 24 KB of filter fragments + 8 KB of activations through the vector-memory path (64 B/clk/CU at best) for
// 384 MFMA cycles per wave, re-fetches and re-SPLITS (fp32 -> 3 x bf16, 44 VALU per thread and chunk) every activation
// once per tap and per 64-wide N tile that reads it (36x for a 128 -> 256 3x3), and issued more VALU than MFMA cycles.
//
// Here a block owns 64 consecutive output pixels x (32 NW) output channels and walks K as (32-channel group, tap):
//   * the activations of a channel group are fetched ONCE per block: the 64 + 2W + 2 consecutive input pixels that the nine
//     taps of the block's 64 output pixels touch (a contiguous pixel range: fully coalesced 128-B runs), split to three
//     bf16 planes once, parked in LDS (rows of 208 B = 3 planes x 64 B + 16 B pad: an odd multiple of 16 B, so the 16-B
//     fragment reads of any 16 consecutive rows cover the 64 banks once WHATEVER the tap's row shift);
//   * a tap is a per-lane LDS row address: halo row r + ky W + kx for output row r, or the block's zero row when the tap
//     falls outside the image (left / right / top / bottom borders, rows past M) -- 18 addresses per lane computed once
//     per block, no select or mask in the K loop;
//   * every wave owns all 64 rows x 32 columns (two 32x32 MFMA tiles): a filter fragment fetched from the stage-packed
//     planes (ConvParams::w16s, the filters-direct format) feeds TWO MFMAs and no two waves fetch the same fragment --
//     per 768 MFMA cycles a block moves 24 KB of filters + 1.4-2.4 KB of activations through the vector-memory path
//     (33 B/clk/CU against 85) and issues a third of the split VALU;
//   * filter fragments two taps ahead in a three-deep register ring, the next channel group's halo in flight from the
//     start of the current group and split / parked in the other LDS stage under the MFMAs of taps .. 7; one block-wide
//     barrier per channel group (216 MFMAs per wave).
// K slices are ranges of channel groups (all nine taps of a group stay in one block).  Split-K hand-off and the fused
// epilogue are the shared conv_tail.inc (same 64-row M tiles as the other kernels: slabs, pooled epilogue, residuals,
// operand planes and store modes unchanged).  Sums per output element: fixed order (group, tap, k), deterministic.
#include "conv_dev.h"

namespace bp {

static constexpr int HALO_ROW_B = 208;

template <int NW, int NPASS>
__global__ __launch_bounds__(64 * NW) void conv_halo_kernel(const ConvParams p) {
    constexpr int NT = 64 * NW, BM = 64, BN = 32 * NW, TM = 2, TN = 1, LDT = BN + 4;
    constexpr int RPP = NT / 4;                  // halo rows per loader pass (4 threads x 8 channels per row)
    extern __shared__ __attribute__((aligned(16))) float smem[];
    char* const lds = reinterpret_cast<char*>(smem);

    const unsigned long long t_entry = p.stamps ? bp_clock() : 0ull;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n_tiles_n = p.CoutPad / BN;
    const int split = (int)blockIdx.x % p.splits;
    const int tile_id = (int)blockIdx.x / p.splits;
    const int tile_n = tile_id % n_tiles_n;
    const int tile_m = tile_id / n_tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int cpt = p.Cin >> 5;                                     // channel groups = chunks per tap
    const int gps = p.chunks_per_split / 9;                         // channel groups per K slice
    const int g_begin = split * gps;
    const int g_end = min(cpt, g_begin + gps);
    const int W = p.W, H = p.H, hw = H * W;
    const int HR = 64 + 2 * W + 2;                                  // halo rows; row HR is the zero row
    const int stage_b = (HR + 1) * HALO_ROW_B;

    // ---- filters: this wave's 32 columns, stage-packed fragments (aux_kernels.hip f32_to_bf16x3_staged_kernel): per 64-row
    // tile and 16-k stage 6 KB = [plane][row][32 B], granule g of row r at slot g ^ ((r >> 3) & 1)
    const int nc0 = n0 + 32 * wave;
    const __amdgpu_buffer_rsrc_t rsrcB = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<unsigned short*>(p.w16s), 0, 3 * p.CoutPad * p.Kpad * 2, 0x00020000);
    const int bd_tile = (nc0 >> 6) * (p.Kpad >> 4) * 6144;
    const unsigned bd_voff = (unsigned)(((nc0 & 32) + (lane & 31)) * 32 + (((lane >> 5) ^ ((lane >> 3) & 1)) << 4));
    u32x4 rb[3][3][2];                                              // [ring slot][plane][k-step]
    auto load_b = [&](auto slotc, int g, int tap) __attribute__((always_inline)) {
        constexpr int slot = decltype(slotc)::value;
        const int so = g < g_end ? bd_tile + (tap * cpt + g) * 12288 : (int)OOB;
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
                rb[slot][pl][ks] = __builtin_amdgcn_raw_buffer_load_b128(rsrcB, (int)bd_voff, so + ks * 6144 + pl * 2048, 0);
    };
    // the first two taps' filters need no index math: requested first (a cold kernel waits > 1 us for its first operands)
    load_b(std::integral_constant<int, 0>{}, g_begin, 0);
    load_b(std::integral_constant<int, 1>{}, g_begin, 1);
    __builtin_amdgcn_sched_barrier(0);

    // ---- activations: halo row j <-> input pixel q0 + j of the [N*H*W] pixel list (stride 1, same padding: OH = H, OW = W)
    const __amdgpu_buffer_rsrc_t rsrcA = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.in), 0, (int)min((long long)p.N * hw * p.in_ld * 4, (long long)OOB), 0x00020000);
    const int jl = tid >> 2, gr = tid & 3;
    const int q0 = m0 - W - 1;
    const int npix = p.N * hw;
    unsigned a_voff[NPASS];
#pragma unroll
    for (int j = 0; j < NPASS; ++j) {
        const int row = jl + j * RPP, q = q0 + row;
        a_voff[j] = (row < HR && q >= 0 && q < npix) ? (unsigned)((q * p.in_ld + gr * 8) * 4) : OOB;
    }
    const unsigned a_woff = (unsigned)(jl * HALO_ROW_B + gr * 16);
    f32x4 ra[NPASS][2];
    auto load_a = [&](int g) __attribute__((always_inline)) {
        const int so = g < g_end ? g * 128 : (int)OOB;
#pragma unroll
        for (int j = 0; j < NPASS; ++j) {
            ra[j][0] = buf_load4(rsrcA, a_voff[j], so);
            ra[j][1] = buf_load4(rsrcA, a_voff[j], so + 16);
        }
    };
    // pass j of the parked halo: exact three-way bf16 split of 8 channels, one 16-B LDS store per plane
    auto park_a = [&](auto jc, unsigned so) __attribute__((always_inline)) {
        constexpr int j = decltype(jc)::value;
        if (j < NPASS - 1 || jl + j * RPP < HR) {
            const f32x4 lo = ra[j][0], hi = ra[j][1];
            const bf16x4 l1 = __builtin_convertvector(lo, bf16x4), h1 = __builtin_convertvector(hi, bf16x4);
            const f32x4 rl1 = lo - __builtin_convertvector(l1, f32x4), rh1 = hi - __builtin_convertvector(h1, f32x4);
            const bf16x4 l2 = __builtin_convertvector(rl1, bf16x4), h2 = __builtin_convertvector(rh1, bf16x4);
            const f32x4 rl2 = rl1 - __builtin_convertvector(l2, f32x4), rh2 = rh1 - __builtin_convertvector(h2, f32x4);
            const bf16x4 l3 = __builtin_convertvector(rl2, bf16x4), h3 = __builtin_convertvector(rh2, bf16x4);
            char* dst = lds + so + a_woff + j * (RPP * HALO_ROW_B);
            *reinterpret_cast<bf16x8*>(dst) = __builtin_shufflevector(l1, h1, 0, 1, 2, 3, 4, 5, 6, 7);
            *reinterpret_cast<bf16x8*>(dst + 64) = __builtin_shufflevector(l2, h2, 0, 1, 2, 3, 4, 5, 6, 7);
            *reinterpret_cast<bf16x8*>(dst + 128) = __builtin_shufflevector(l3, h3, 0, 1, 2, 3, 4, 5, 6, 7);
        }
    };
    load_a(g_begin);

    // the zero row of both stages
    if (tid < 26) {
        const int st = tid / 13, u = tid - st * 13;
        *reinterpret_cast<u32x4*>(lds + st * stage_b + HR * HALO_ROW_B + u * 16) = u32x4{0u, 0u, 0u, 0u};
    }

    // ---- per lane: the LDS row of every (tap, output row) pair.  Output rows r = (lane & 31) + 32 i; lane >> 5 picks the
    // k half of a 16-k step (v_mfma_f32_32x32x16_bf16: lane l supplies row l & 31, k = 8 (l >> 5) .. + 7)
    unsigned fa_addr[9][2];
    {
        const float rcp_hw = 1.0f / (float)hw, rcp_w = 1.0f / (float)W;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int r = (lane & 31) + 32 * i;
            const int m = m0 + r;
            const bool ok = m < p.M;
            const int mm = ok ? m : 0;
            const int b = fast_div(mm, hw, rcp_hw);
            const int rem = mm - b * hw;
            const int oy = fast_div(rem, W, rcp_w);
            const int ox = rem - oy * W;
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int ky = t / 3, kx = t % 3;
                const bool valid = ok && (unsigned)(oy + ky - 1) < (unsigned)H && (unsigned)(ox + kx - 1) < (unsigned)W;
                const int row = valid ? r + ky * W + kx : HR;
                fa_addr[t][i] = (unsigned)(row * HALO_ROW_B + (lane >> 5) * 16);
            }
        }
    }
    // the epilogue's bias, requested before the K loop (conv_tail.inc BP_EARLY_BIAS)
    const f32x4 bias_early = *reinterpret_cast<const f32x4*>(p.bias + min(n0 + (tid % (BN / 4)) * 4, p.CoutPad - 4));

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][0][r] = 0.f;

    typedef bf16x8 frag_t;
    frag_t fr[2][3][2];                                             // [step parity][plane][row half]
    auto read_frags = [&](auto parc, auto tapc, auto ksc, unsigned so) __attribute__((always_inline)) {
        constexpr int par = decltype(parc)::value, tap = decltype(tapc)::value, ks = decltype(ksc)::value;
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
#pragma unroll
            for (int i = 0; i < 2; ++i)
                fr[par][pl][i] = *reinterpret_cast<const frag_t*>(lds + so + fa_addr[tap][i] + (pl * 64 + ks * 32));
    };
    // partial products (A plane, B plane), smallest first (conv_igemm.hip)
    constexpr int PA[6] = {2, 1, 0, 1, 0, 0};
    constexpr int PB[6] = {0, 1, 2, 0, 1, 0};

    // prologue: the first group's halo into stage 0
    static_for<NPASS>([&](auto jc) __attribute__((always_inline)) { park_a(jc, 0u); });
    __syncthreads();
    read_frags(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, 0u);

    unsigned so_cur = 0u, so_nxt = (unsigned)stage_b;
    for (int g = g_begin; g < g_end; ++g) {
        load_a(g + 1);                                              // (past the slice: out-of-range offsets, zeros, no traffic)
        static_for<9>([&](auto tapc) __attribute__((always_inline)) {
            constexpr int tap = decltype(tapc)::value;
            // filters two taps ahead
            if constexpr (tap < 7) load_b(std::integral_constant<int, (tap + 2) % 3>{}, g, tap + 2);
            else load_b(std::integral_constant<int, (tap + 2) % 3>{}, g + 1, tap - 7);
            static_for<2>([&](auto ksc) __attribute__((always_inline)) {
                constexpr int ks = decltype(ksc)::value;
                constexpr int step = tap * 2 + ks, par = step & 1;
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (step < 17) {
                    read_frags(std::integral_constant<int, par ^ 1>{}, std::integral_constant<int, (step + 1) / 2>{},
                               std::integral_constant<int, (step + 1) & 1>{}, so_cur);
                } else {
                    // everybody's next halo is parked and nobody reads this stage any more (the last step's fragments
                    // are in registers): swap stages
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __syncthreads();
                    read_frags(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, so_nxt);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int q = 0; q < 6; ++q)
#pragma unroll
                    for (int i = 0; i < 2; ++i)
                        acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                            fr[par][PA[q]][i], __builtin_bit_cast(frag_t, rb[tap % 3][PB[q]][ks]), acc[i][0], 0, 0, 0);
            });
            // the next group's halo: one loader pass per tap, the last one behind tap 7 (all of them parked before the barrier
            // inside tap 8; their loads are a few thousand cycles old by then)
            if constexpr (tap <= 7 && tap >= 8 - NPASS) park_a(std::integral_constant<int, tap - (8 - NPASS)>{}, so_nxt);
        });
        const unsigned t_ = so_cur; so_cur = so_nxt; so_nxt = t_;
    }
    __syncthreads();
#define BHL_STAMP(k_) if (p.stamps && tid == 0) p.stamps[(long long)(tile_id * p.splits + split) * 8 + (k_)] = bp_clock();
    if (p.stamps && tid == 0) p.stamps[(long long)(tile_id * p.splits + split) * 8 + 0] = t_entry;
    BHL_STAMP(3);   // K loop done

    const int w_row0 = 0, w_col0 = 32 * wave;
    __shared__ int s_last;
#define BP_NT NT
#define BP_SLAST s_last
#define BP_EARLY_BIAS bias_early
#define BP_EP_PF_MAX 8
#define BP_EP_PIXSHUF
#define BP_TAIL_STAMP(k_) BHL_STAMP(k_)
#include "conv_tail.inc"
#undef BP_EP_PF_MAX
#undef BP_EP_PIXSHUF
#undef BP_EARLY_BIAS
#undef BP_TAIL_STAMP
#undef BP_NT
#undef BP_SLAST
    if (p.stamps) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); BHL_STAMP(4); }
#undef BHL_STAMP
}

// =====================================================================================================================
// TILE_HALO64K2 (round 4): the same halo scheme with the K split INSIDE the block -- 64x64 output tile, four waves as
// 2 K groups x 2 column halves.  K group kg walks the channel groups g = g_begin + kg, + 2, ... of the block's slice with its own
// LDS halo stage and its own filter ring; the two partial sums meet in LDS once, then one epilogue per tile.  For the layers whose
// 64x128 tiles needed 2 K slices to fill the chip (52x52 128 -> 256: 86 tiles x 2 slices) this is the same number of blocks
// (172) with NO slab, ticket or cross-block reduction -- the reducing block's tail was ~10 us of a ~22 us layer
// (bench.py --insitu, BP_INSITU_DUMP) -- and where cross-block slices remain their count halves.
// The halo stage of a K group is single-buffered (two stages = the LDS footprint of the double-buffered 64x128 tile): the next
// channel group's halo waits in registers during the nine taps and is split / parked between two barriers.
// =====================================================================================================================
template <int NPASS, class P>
__device__ __forceinline__ void conv_halo_k2_body(const P& p, const int bp_bid, float* const smem, int* const s_last_p) {
    constexpr int NT = 256, BM = 64, BN = 64, TM = 2, TN = 1, LDT = BN + 4;
    constexpr int RPP = 32;                      // halo rows per loader pass of a K group (128 threads, 4 per row)
    char* const lds = reinterpret_cast<char*>(smem);

    const unsigned long long t_entry = p.stamps ? bp_clock() : 0ull;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kg = wave >> 1, wn = wave & 1;
    const int n_tiles_n = p.CoutPad / BN;
    const int split = bp_bid % p.splits;
    const int tile_id = bp_bid / p.splits;
    const int tile_n = tile_id % n_tiles_n;
    const int tile_m = tile_id / n_tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int cpt = p.Cin >> 5;
    const int gps = p.chunks_per_split / 9;
    const int g_begin = split * gps;
    const int g_end = min(cpt, g_begin + gps);
    const int W = p.W, H = p.H, hw = H * W;
    const int HR = 64 + 2 * W + 2;
    const int stage_b = (HR + 1) * HALO_ROW_B;
    const unsigned so_mine = (unsigned)(kg * stage_b);              // this K group's halo stage

    const int nc0 = n0 + 32 * wn;
    const __amdgpu_buffer_rsrc_t rsrcB = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<unsigned short*>(p.w16s), 0, 3 * p.CoutPad * p.Kpad * 2, 0x00020000);
    const int bd_tile = (nc0 >> 6) * (p.Kpad >> 4) * 6144;
    const unsigned bd_voff = (unsigned)(((nc0 & 32) + (lane & 31)) * 32 + (((lane >> 5) ^ ((lane >> 3) & 1)) << 4));
    u32x4 rb[3][3][2];
    auto load_b = [&](auto slotc, int g, int tap) __attribute__((always_inline)) {
        constexpr int slot = decltype(slotc)::value;
        const int so = g < g_end ? bd_tile + (tap * cpt + g) * 12288 : (int)OOB;
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
                rb[slot][pl][ks] = __builtin_amdgcn_raw_buffer_load_b128(rsrcB, (int)bd_voff, so + ks * 6144 + pl * 2048, 0);
    };
    const int g0 = g_begin + kg;                                    // this K group's first channel group
    load_b(std::integral_constant<int, 0>{}, g0, 0);
    load_b(std::integral_constant<int, 1>{}, g0, 1);
    __builtin_amdgcn_sched_barrier(0);

    const __amdgpu_buffer_rsrc_t rsrcA = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.in), 0, (int)min((long long)p.N * hw * p.in_ld * 4, (long long)OOB), 0x00020000);
    const int t2 = tid & 127;                                       // thread inside the K group
    const int jl = t2 >> 2, gr = t2 & 3;
    const int q0 = m0 - W - 1;
    const int npix = p.N * hw;
    unsigned a_voff[NPASS];
#pragma unroll
    for (int j = 0; j < NPASS; ++j) {
        const int row = jl + j * RPP, q = q0 + row;
        a_voff[j] = (row < HR && q >= 0 && q < npix) ? (unsigned)((q * p.in_ld + gr * 8) * 4) : OOB;
    }
    const unsigned a_woff = so_mine + (unsigned)(jl * HALO_ROW_B + gr * 16);
    f32x4 ra[NPASS][2];
    auto load_a = [&](int g) __attribute__((always_inline)) {
        const int so = g < g_end ? g * 128 : (int)OOB;
#pragma unroll
        for (int j = 0; j < NPASS; ++j) {
            ra[j][0] = buf_load4(rsrcA, a_voff[j], so);
            ra[j][1] = buf_load4(rsrcA, a_voff[j], so + 16);
        }
    };
    auto park_a = [&](auto jc) __attribute__((always_inline)) {
        constexpr int j = decltype(jc)::value;
        if (j < NPASS - 1 || jl + j * RPP < HR) {
            const f32x4 lo = ra[j][0], hi = ra[j][1];
            const bf16x4 l1 = __builtin_convertvector(lo, bf16x4), h1 = __builtin_convertvector(hi, bf16x4);
            const f32x4 rl1 = lo - __builtin_convertvector(l1, f32x4), rh1 = hi - __builtin_convertvector(h1, f32x4);
            const bf16x4 l2 = __builtin_convertvector(rl1, bf16x4), h2 = __builtin_convertvector(rh1, bf16x4);
            const f32x4 rl2 = rl1 - __builtin_convertvector(l2, f32x4), rh2 = rh1 - __builtin_convertvector(h2, f32x4);
            const bf16x4 l3 = __builtin_convertvector(rl2, bf16x4), h3 = __builtin_convertvector(rh2, bf16x4);
            char* dst = lds + a_woff + j * (RPP * HALO_ROW_B);
            *reinterpret_cast<bf16x8*>(dst) = __builtin_shufflevector(l1, h1, 0, 1, 2, 3, 4, 5, 6, 7);
            *reinterpret_cast<bf16x8*>(dst + 64) = __builtin_shufflevector(l2, h2, 0, 1, 2, 3, 4, 5, 6, 7);
            *reinterpret_cast<bf16x8*>(dst + 128) = __builtin_shufflevector(l3, h3, 0, 1, 2, 3, 4, 5, 6, 7);
        }
    };
    load_a(g0);

    if (tid < 26) {   // the zero row of both stages
        const int st = tid / 13, u = tid - st * 13;
        *reinterpret_cast<u32x4*>(lds + st * stage_b + HR * HALO_ROW_B + u * 16) = u32x4{0u, 0u, 0u, 0u};
    }
    unsigned fa_addr[9][2];
    {
        const float rcp_hw = 1.0f / (float)hw, rcp_w = 1.0f / (float)W;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int r = (lane & 31) + 32 * i;
            const int m = m0 + r;
            const bool ok = m < p.M;
            const int mm = ok ? m : 0;
            const int b = fast_div(mm, hw, rcp_hw);
            const int rem = mm - b * hw;
            const int oy = fast_div(rem, W, rcp_w);
            const int ox = rem - oy * W;
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int ky = t / 3, kx = t % 3;
                const bool valid = ok && (unsigned)(oy + ky - 1) < (unsigned)H && (unsigned)(ox + kx - 1) < (unsigned)W;
                const int row = valid ? r + ky * W + kx : HR;
                fa_addr[t][i] = so_mine + (unsigned)(row * HALO_ROW_B + (lane >> 5) * 16);
            }
        }
    }
    const f32x4 bias_early = *reinterpret_cast<const f32x4*>(p.bias + min(n0 + (tid % (BN / 4)) * 4, p.CoutPad - 4));

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][0][r] = 0.f;

    typedef bf16x8 frag_t;
    frag_t fr[2][3][2];
    auto read_frags = [&](auto parc, auto tapc, auto ksc) __attribute__((always_inline)) {
        constexpr int par = decltype(parc)::value, tap = decltype(tapc)::value, ks = decltype(ksc)::value;
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
#pragma unroll
            for (int i = 0; i < 2; ++i)
                fr[par][pl][i] = *reinterpret_cast<const frag_t*>(lds + fa_addr[tap][i] + (pl * 64 + ks * 32));
    };
    constexpr int PA[6] = {2, 1, 0, 1, 0, 0};
    constexpr int PB[6] = {0, 1, 2, 0, 1, 0};

    static_for<NPASS>([&](auto jc) __attribute__((always_inline)) { park_a(jc); });
    __syncthreads();
    read_frags(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});

    // both K groups make the same number of trips (a group past the slice multiplies zeros: out-of-range loads)
    const int trips = (g_end - g_begin + 1) >> 1;
    for (int it = 0, g = g0; it < trips; ++it, g += 2) {
        load_a(g + 2);
        static_for<9>([&](auto tapc) __attribute__((always_inline)) {
            constexpr int tap = decltype(tapc)::value;
            if constexpr (tap < 7) load_b(std::integral_constant<int, (tap + 2) % 3>{}, g, tap + 2);
            else load_b(std::integral_constant<int, (tap + 2) % 3>{}, g + 2, tap - 7);
            static_for<2>([&](auto ksc) __attribute__((always_inline)) {
                constexpr int ks = decltype(ksc)::value;
                constexpr int step = tap * 2 + ks, par = step & 1;
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (step < 17)
                    read_frags(std::integral_constant<int, par ^ 1>{}, std::integral_constant<int, (step + 1) / 2>{},
                               std::integral_constant<int, (step + 1) & 1>{});
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int q = 0; q < 6; ++q)
#pragma unroll
                    for (int i = 0; i < 2; ++i)
                        acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                            fr[par][PA[q]][i], __builtin_bit_cast(frag_t, rb[tap % 3][PB[q]][ks]), acc[i][0], 0, 0, 0);
            });
        });
        // every fragment of this group is in registers or consumed: the stage may be overwritten
        __syncthreads();
        static_for<NPASS>([&](auto jc) __attribute__((always_inline)) { park_a(jc); });
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __syncthreads();
        read_frags(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
    }
    __syncthreads();
#define BHK_STAMP(k_) if (p.stamps && tid == 0) p.stamps[(long long)(tile_id * p.splits + split) * 8 + (k_)] = bp_clock();
    if (p.stamps && tid == 0) p.stamps[(long long)(tile_id * p.splits + split) * 8 + 0] = t_entry;
    BHK_STAMP(3);   // K loops done

    // the two K groups' partial sums meet in LDS: group 1 parks its 64x32 per wave in fragment order (16 B per lane and store,
    // conflict-free), group 0 adds them -- the fixed order (group 0) + (group 1) keeps the result deterministic
    {
        float* const xch = smem + wn * 2048 + lane * 4;
        if (kg == 1) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    *reinterpret_cast<f32x4*>(xch + (i * 4 + q) * 256) = f32x4{acc[i][0][4 * q], acc[i][0][4 * q + 1], acc[i][0][4 * q + 2], acc[i][0][4 * q + 3]};
        }
        __syncthreads();
        if (kg == 0) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4 v = *reinterpret_cast<const f32x4*>(xch + (i * 4 + q) * 256);
                    acc[i][0][4 * q] += v.x; acc[i][0][4 * q + 1] += v.y; acc[i][0][4 * q + 2] += v.z; acc[i][0][4 * q + 3] += v.w;
                }
        }
        __syncthreads();
    }

    const int w_row0 = 0, w_col0 = 32 * wn;
    const bool has_acc = kg == 0;
    // (conv_tail.inc indexes the slab by `wave`: the accumulator-owning waves are 0 and 1 = the two column halves)
#define BP_NT NT
#define BP_SLAST (*s_last_p)
#define BP_EARLY_BIAS bias_early
#define BP_HAS_ACC has_acc
#define BP_EP_PIXSHUF
#define BP_TAIL_STAMP(k_) BHK_STAMP(k_)
#include "conv_tail.inc"
#undef BP_EP_PIXSHUF
#undef BP_HAS_ACC
#undef BP_EARLY_BIAS
#undef BP_TAIL_STAMP
#undef BP_NT
#undef BP_SLAST
    if (p.stamps) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); BHK_STAMP(4); }
#undef BHK_STAMP
}

template <int NPASS>
__global__ __launch_bounds__(256) void conv_halo_k2_kernel(const ConvParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    __shared__ int s_last;
    conv_halo_k2_body<NPASS>(p, (int)blockIdx.x, smem, &s_last);
}

bool conv_tile_is_halo(int tile) { return tile == TILE_HALO64 || tile == TILE_HALO128 || tile == TILE_HALO64K2; }

static int halo_nw(int tile) { return tile == TILE_HALO128 ? 4 : 2; }
static int halo_passes(const ConvParams& p, int tile) { return (64 + 2 * p.W + 2 + 16 * halo_nw(tile) - 1) / (16 * halo_nw(tile)); }   // (the K2 tile loads with 128 threads per K group, like the 64x64 tile)

bool conv_halo_eligible(const ConvParams& p, int tile) {
    if (!conv_tile_is_halo(tile)) return false;
    if (!(p.ksize == 3 && p.stride == 1 && p.pad == 1 && p.OH == p.H && p.OW == p.W)) return false;
    if (!(p.w16s != nullptr && p.Cin % 32 == 0 && p.in_ld % 4 == 0 && p.Kpad == 9 * p.Cin)) return false;
    if (p.CoutPad % (tile == TILE_HALO64K2 ? 64 : 32 * halo_nw(tile)) != 0) return false;
    const int np = halo_passes(p, tile);
    return tile == TILE_HALO128 ? (np >= 2 && np <= 4) : (np >= 3 && np <= 7);
}

size_t conv_halo_lds_bytes(const ConvParams& p, int tile) {
    const size_t stages = (size_t)2 * (64 + 2 * p.W + 2 + 1) * HALO_ROW_B;
    const size_t staging = (size_t)64 * ((tile == TILE_HALO64K2 ? 64 : 32 * halo_nw(tile)) + 4) * 4;
    return stages > staging ? stages : staging;
}

template <int NW, int NPASS>
static void launch_halo_t(const ConvParams& p, int tile, hipStream_t s) {
    ConvParams q = p;
    conv_grid_setup(q, 64, 32 * NW);
    const size_t lds = conv_halo_lds_bytes(p, tile);
    allow_big_lds(reinterpret_cast<const void*>(&conv_halo_kernel<NW, NPASS>));   // (> 64 KB of dynamic LDS: once per device and kernel)
    dim3 grid(q.n_tiles * q.splits);
    if (g_conv_prof)
        hipExtLaunchKernelGGL((conv_halo_kernel<NW, NPASS>), grid, dim3(64 * NW), lds, s, g_conv_prof->e0, g_conv_prof->e1, 0, q);
    else
        hipLaunchKernelGGL((conv_halo_kernel<NW, NPASS>), grid, dim3(64 * NW), lds, s, q);
}

template <int NPASS>
static void launch_halo_k2_t(const ConvParams& p, int tile, hipStream_t s) {
    ConvParams q = p;
    conv_grid_setup(q, 64, 64);
    const size_t lds = conv_halo_lds_bytes(p, tile);
    allow_big_lds(reinterpret_cast<const void*>(&conv_halo_k2_kernel<NPASS>));
    dim3 grid(q.n_tiles * q.splits);
    if (g_conv_prof)
        hipExtLaunchKernelGGL((conv_halo_k2_kernel<NPASS>), grid, dim3(256), lds, s, g_conv_prof->e0, g_conv_prof->e1, 0, q);
    else
        hipLaunchKernelGGL((conv_halo_k2_kernel<NPASS>), grid, dim3(256), lds, s, q);
}

void launch_conv_halo(const ConvParams& p, int tile, hipStream_t s) {
    BP_CHECK(p.mfma_mode == PREC_BF16X3 && conv_halo_eligible(p, tile),
             "halo tile: bf16x3 mode, 3x3 / stride 1 / pad 1, Cin % 32 == 0, stage-packed filters, W <= 95 (64x128) or 79 (64x64)");
    BP_CHECK((long long)3 * p.CoutPad * p.Kpad * 2 < (long long)OOB, "filter planes too large for 32-bit offsets");
    BP_CHECK(p.chunks_per_split % 9 == 0, "halo tile: K slices are whole channel groups (chunks_per_split % 9 == 0)");
    BP_CHECK(!p.xcd_home && !p.pf_ptr && p.hy_splits == 0, "halo tile: no latency-mode layouts");
    const int np = halo_passes(p, tile);
    if (tile == TILE_HALO64K2) {
        switch (np) {
            case 3: launch_halo_k2_t<3>(p, tile, s); break;
            case 4: launch_halo_k2_t<4>(p, tile, s); break;
            case 5: launch_halo_k2_t<5>(p, tile, s); break;
            case 6: launch_halo_k2_t<6>(p, tile, s); break;
            default: launch_halo_k2_t<7>(p, tile, s); break;
        }
    } else if (tile == TILE_HALO128) {
        switch (np) {
            case 2: launch_halo_t<4, 2>(p, tile, s); break;
            case 3: launch_halo_t<4, 3>(p, tile, s); break;
            default: launch_halo_t<4, 4>(p, tile, s); break;
        }
    } else {
        switch (np) {
            case 3: launch_halo_t<2, 3>(p, tile, s); break;
            case 4: launch_halo_t<2, 4>(p, tile, s); break;
            case 5: launch_halo_t<2, 5>(p, tile, s); break;
            case 6: launch_halo_t<2, 6>(p, tile, s); break;
            default: launch_halo_t<2, 7>(p, tile, s); break;
        }
    }
}

}  // namespace bp
