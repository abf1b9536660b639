// Conv -> conv fusion inside a residual / bottleneck block (round 5): ONE launch computes, for an 8 x 8 patch of output pixels,
//     [1x1 conv + BN + activation]  ->  [3x3 / stride-1 conv + BN + activation]  ( ->  [1x1 conv + BN] )  + skip connection
// i.e. the body of yolo/darknet.py:319-363's forward loop for a Darknet-53 residual block (cfg: 1x1 C -> C/2, 3x3 C/2 -> C,
// shortcut from = -3, darknet.py:338-340) and KPD/src/models/layers/SE_Resnet.py:25-42 Bottleneck.forward (conv1 / bn1 / relu,
// conv2 / bn2 / relu, conv3 / bn3, += residual, relu).  The intermediate tensors never leave LDS / registers.
//
// Which blocks: a block that computes the 1x1 in front of a 3x3 has to own every mid channel of its pixels AND of their 3x3 halo,
// and a block that feeds the trailing 1x1 every output channel of the 3x3 -- so this form pays where M is large and the channel
// counts are small: the 208x208 / 104x104 residual blocks of the detector and the 80x64 bottlenecks of the key-point detector
// (676 / 169 / 80 patches).  At 52x52 and below the same block would recompute the 1x1 once per 64-wide N tile of the 3x3 on a
// 2.7x (1-D strip) or 1.56x (this 2-D patch) halo while the layers are already cut 3-10 ways along K to fill 256 CUs (DESIGN.md
// section 4): measured, not planned there (profiles/r05_fused_blocks.txt).
//
// The patch: 8 x 8 output pixels = the 64 rows of the GEMM tile; their 3x3 taps touch a 10 x 10 input patch (1.56x instead of the
// 2.66x of a 64-pixel strip of a 104-wide map -- 64 + 2 W + 2 pixels -- and 7.5x at W = 208, which is why these layers had stayed
// on the round-2 kernel).  Per block, bf16x3 arithmetic throughout (exact 3-way split, six products, fp32 accumulate):
//   1. X patch (100 pixels x Cin of the first 1x1, fp32, zeros outside the image) streamed by 32-channel groups: fetched once,
//      split once, parked in LDS as three bf16 planes (rows of 208 B as in conv_halo.hip);
//      mid[128 x MID] += X_g W1_g on the matrix cores (each wave 32 patch rows x all MID channels, filter fragments straight from
//      the stage-packed planes into registers);
//   2. mid + bias, activation, ZERO where the patch pixel lies outside the image (the 3x3 pads its INPUT with zeros), transposed
//      through a wave-private LDS tile, split, parked as the 3x3's tap-resident halo Y[MID / 32][100 rows];
//   3. the 3x3 as nine per-lane LDS row offsets (compile-time immediates: row = py 10 + px, tap offset ky 10 + kx), 2 x 2 waves of
//      32 x 32, filter fragments two chunks ahead in a three-deep register ring;
//   4. (bottleneck form) + bias, activation, split, parked as Z[64 rows x 64]; the expanding 1x1: every wave 64 rows x 64 of the up
//      to 256 output channels;
//   5. epilogue through an LDS staging tile: + bias, skip connection (before or after the activation), 16-B stores at the patch
//      rows' pixel offsets.
// Sums per output element are in a fixed order (deterministic); the 1x1's are the unfused kernel's own order, so its result is
// bit-identical to the unfused launch's.
#include "conv_dev.h"

namespace bp {

static constexpr int FB_ROW_B = 208;              // LDS row: 3 planes x 64 B + 16 B (an odd multiple of 16 B)
static constexpr int FB_IW = 10, FB_NR = 100;     // input patch 10 x 10
static constexpr int FB_YG_B = 104 * FB_ROW_B;    // one 32-channel group of the 3x3's halo

__host__ __device__ constexpr int fb_max(int a, int b) { return a > b ? a : b; }
template <int MIDG, bool POST>
struct FusedLds {
    static constexpr int MID = 32 * MIDG, LDTP = MID + 4;
    static constexpr int A_B = fb_max(fb_max(128 * FB_ROW_B, 128 * LDTP * 4), 64 * 68 * 4);   // X stage | pre transposition tiles | staging
    static constexpr int B_B = fb_max(MIDG * FB_YG_B, POST ? 2 * 64 * FB_ROW_B : 0);            // Y halo | Z
    static constexpr int BYTES = A_B + B_B;
};

// one stage's filter fragment: 32 columns starting at nc0, k16 stage `st` of the stage-packed planes, plane pl
__device__ __forceinline__ u32x4 fb_load_b(__amdgpu_buffer_rsrc_t r, unsigned voff, int so) {
    return __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, so, 0);
}

// exact 3-way split of 8 fp32 values -> three 16-B LDS stores (planes 64 B apart)
__device__ __forceinline__ void fb_park8(char* dst, f32x4 lo, f32x4 hi) {
    const bf16x4 l1 = __builtin_convertvector(lo, bf16x4), h1 = __builtin_convertvector(hi, bf16x4);
    const f32x4 rl1 = lo - __builtin_convertvector(l1, f32x4), rh1 = hi - __builtin_convertvector(h1, f32x4);
    const bf16x4 l2 = __builtin_convertvector(rl1, bf16x4), h2 = __builtin_convertvector(rh1, bf16x4);
    const f32x4 rl2 = rl1 - __builtin_convertvector(l2, f32x4), rh2 = rh1 - __builtin_convertvector(h2, f32x4);
    const bf16x4 l3 = __builtin_convertvector(rl2, bf16x4), h3 = __builtin_convertvector(rh2, bf16x4);
    *reinterpret_cast<bf16x8*>(dst) = __builtin_shufflevector(l1, h1, 0, 1, 2, 3, 4, 5, 6, 7);
    *reinterpret_cast<bf16x8*>(dst + 64) = __builtin_shufflevector(l2, h2, 0, 1, 2, 3, 4, 5, 6, 7);
    *reinterpret_cast<bf16x8*>(dst + 128) = __builtin_shufflevector(l3, h3, 0, 1, 2, 3, 4, 5, 6, 7);
}

__device__ __forceinline__ f32x4 fb_act4(f32x4 v, int act) {
    if (act == ACT_LEAKY) {
        v.x = v.x > 0.f ? v.x : 0.1f * v.x; v.y = v.y > 0.f ? v.y : 0.1f * v.y;
        v.z = v.z > 0.f ? v.z : 0.1f * v.z; v.w = v.w > 0.f ? v.w : 0.1f * v.w;
    } else if (act == ACT_RELU) {
        v.x = v.x > 0.f ? v.x : 0.f; v.y = v.y > 0.f ? v.y : 0.f;
        v.z = v.z > 0.f ? v.z : 0.f; v.w = v.w > 0.f ? v.w : 0.f;
    }
    return v;
}

// G1 = 32-channel groups of the first 1x1's input (2, 4 or 8): its loop is unrolled, the X rows of up to four groups are in flight
// (registers) -- one group ahead left ~2 us of cold-fetch latency exposed per group (first GPU run of round 5: 18 / 22 / 30 us for the
// 208x208 / 104x104 / 80x64 blocks against 24 / 25 / 25.5 us unfused)
template <int MIDG, bool POST, int G1>
__global__ __launch_bounds__(256) void conv_fused_kernel(const ConvParams pre, const ConvParams c3, const ConvParams post) {
    typedef FusedLds<MIDG, POST> L;
    constexpr int MID = L::MID, LDTP = L::LDTP;
    constexpr int LDT = 68;                                // staging rows (floats) of the 64-wide tiles
    extern __shared__ __attribute__((aligned(16))) float smem[];
    char* const lds = reinterpret_cast<char*>(smem);
    char* const ldsA = lds;
    char* const ldsB = lds + L::A_B;
    typedef bf16x8 frag_t;
    constexpr int PA[6] = {2, 1, 0, 1, 0, 0};              // partial products (A plane, B plane), smallest first (conv_igemm.hip)
    constexpr int PB[6] = {0, 1, 2, 0, 1, 0};

    // in-situ marks (Net::set_stamps, of the block's LAST member): 0 entry | 1 first 1x1 done | 2 halo parked | 3 3x3 done ("K loop done") |
    // 5 trailing 1x1 done | 4 stores done
    unsigned long long* const stamps = POST ? post.stamps : c3.stamps;
    const unsigned long long t_entry = stamps ? bp_clock() : 0ull;
#define FB_STAMP(k_) if (stamps && threadIdx.x == 0) stamps[(long long)blockIdx.x * 8 + (k_)] = bp_clock();
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int H = pre.H, W = pre.W;
    const int ntn = POST ? 1 : (c3.CoutPad >> 6);
    const int bid = (int)blockIdx.x;
    const int patch = bid / ntn, tile_n = bid - patch * ntn;
    const int TX = (W + 7) >> 3, TY = (H + 7) >> 3;
    const int b = patch / (TX * TY);
    const int pr = patch - b * (TX * TY);
    const int ty = pr / TX, tx = pr - ty * TX;
    const int iy0 = ty * 8 - 1, ix0 = tx * 8 - 1;

    // ---- stage 1 operands.  Filters of the first 1x1: the wave needs all MID columns
    const __amdgpu_buffer_rsrc_t rsrcW1 = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<unsigned short*>(pre.w16s), 0, 3 * pre.CoutPad * pre.Kpad * 2, 0x00020000);
    const unsigned bd_lane = (unsigned)((lane & 31) * 32 + (((lane >> 5) ^ ((lane >> 3) & 1)) << 4));   // row (lane & 31) of a 32-column half, k half (lane >> 5)
    u32x4 rb1[2][MIDG][3][2];                                // [group parity][column tile][plane][k-step]
    constexpr int groups1 = G1;
    auto load_w1 = [&](auto parc, auto gc) __attribute__((always_inline)) {
        constexpr int par = decltype(parc)::value, g = decltype(gc)::value;
        if constexpr (g < G1) {
#pragma unroll
            for (int j = 0; j < MIDG; ++j)
#pragma unroll
                for (int pl = 0; pl < 3; ++pl)
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks)
                        rb1[par][j][pl][ks] = fb_load_b(rsrcW1, bd_lane + (unsigned)((j & 1) * 1024), (j >> 1) * (pre.Kpad >> 4) * 6144 + g * 12288 + ks * 6144 + pl * 2048);
        }
    };
    load_w1(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
    __builtin_amdgcn_sched_barrier(0);

    // the X patch: thread -> patch row (tid >> 2) + 64 j, 8 channels (tid & 3) of the group
    const __amdgpu_buffer_rsrc_t rsrcX = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(pre.in), 0, (int)min((long long)pre.N * H * W * pre.in_ld * 4, (long long)OOB), 0x00020000);
    unsigned x_voff[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int row = (tid >> 2) + 64 * j;
        const int ly = (row * 205) >> 11, lx = row - 10 * ly;
        const int iy = iy0 + ly, ix = ix0 + lx;
        const bool ok = row < FB_NR && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
        x_voff[j] = ok ? (unsigned)((((b * H + iy) * W + ix) * pre.in_ld + (tid & 3) * 8) * 4) : OOB;
    }
    const bool x_row1 = (tid >> 2) + 64 < FB_NR;              // the second pass covers patch rows 64 .. 99 only
    constexpr int XD = G1 < 4 ? G1 : 4;                       // groups of X rows in flight
    f32x4 rx[XD][2][2];
    auto load_x = [&](auto gc) __attribute__((always_inline)) {
        constexpr int g = decltype(gc)::value;
        if constexpr (g < G1) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                rx[g % XD][j][0] = buf_load4(rsrcX, x_voff[j], g * 128);
                rx[g % XD][j][1] = buf_load4(rsrcX, x_voff[j], g * 128 + 16);
            }
        }
    };
    auto park_x = [&](auto gc) __attribute__((always_inline)) {
        constexpr int sl = decltype(gc)::value % XD;
        char* dst = ldsA + (tid >> 2) * FB_ROW_B + (tid & 3) * 16;
        fb_park8(dst, rx[sl][0][0], rx[sl][0][1]);
        if (x_row1) fb_park8(dst + 64 * FB_ROW_B, rx[sl][1][0], rx[sl][1][1]);
    };
    static_for<XD>([&](auto gc) __attribute__((always_inline)) { load_x(gc); });
    // patch rows 100 .. 127 of the X stage are never written: zero them once (the first GEMM reads 128 rows; rows >= 100 are dropped later)
    for (int u = tid; u < 28 * 13; u += 256) *reinterpret_cast<u32x4*>(ldsA + FB_NR * FB_ROW_B + u * 16) = u32x4{0u, 0u, 0u, 0u};

    f32x16 acc1[MIDG];
#pragma unroll
    for (int j = 0; j < MIDG; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc1[j][r] = 0.f;
    const unsigned fa1 = (unsigned)((32 * wave + (lane & 31)) * FB_ROW_B + (lane >> 5) * 16);   // this wave's 32 patch rows

    park_x(std::integral_constant<int, 0>{});
    load_x(std::integral_constant<int, XD>{});                // (slot 0 is free again)
    __syncthreads();
    // one 32-channel group: the next group's filters are requested first, then 2 k-steps x 6 MIDG MFMAs on the parked group
    static_for<G1>([&](auto gc) __attribute__((always_inline)) {
        constexpr int g = decltype(gc)::value, par = g & 1;
        load_w1(std::integral_constant<int, par ^ 1>{}, std::integral_constant<int, g + 1>{});
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            frag_t fa[3];
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) fa[pl] = *reinterpret_cast<const frag_t*>(ldsA + fa1 + pl * 64 + ks * 32);
#pragma unroll
            for (int q = 0; q < 6; ++q)
#pragma unroll
                for (int j = 0; j < MIDG; ++j)
                    acc1[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[PA[q]], __builtin_bit_cast(frag_t, rb1[par][j][PB[q]][ks]), acc1[j], 0, 0, 0);
        }
        __syncthreads();                                      // every wave has read the stage
        if constexpr (g + 1 < G1) {
            park_x(std::integral_constant<int, g + 1>{});
            load_x(std::integral_constant<int, g + 1 + XD>{});
        }
        __syncthreads();
    });

    if (stamps && tid == 0) stamps[(long long)blockIdx.x * 8] = t_entry;
    FB_STAMP(1);
    // ---- stage 3 filters: requested now, they arrive under the transposition.  Wave = (row half wm, column half wn) of the 64 x 64 tile
    const int wm = wave >> 1, wn = wave & 1;
    const int n0 = tile_n * 64;
    const __amdgpu_buffer_rsrc_t rsrcW3 = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<unsigned short*>(c3.w16s), 0, 3 * c3.CoutPad * c3.Kpad * 2, 0x00020000);
    const int w3_tile = tile_n * (c3.Kpad >> 4) * 6144;
    const unsigned w3_voff = bd_lane + (unsigned)(wn * 1024);
    u32x4 rb3[3][3][2];                                       // [ring slot][plane][k-step]
    auto load_w3 = [&](auto slotc, auto chunkc) __attribute__((always_inline)) {
        constexpr int slot = decltype(slotc)::value, chunk = decltype(chunkc)::value;
        if constexpr (chunk < 9 * MIDG) {
#pragma unroll
            for (int pl = 0; pl < 3; ++pl)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
                    rb3[slot][pl][ks] = fb_load_b(rsrcW3, w3_voff, w3_tile + chunk * 12288 + ks * 6144 + pl * 2048);
        }
    };
    load_w3(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
    load_w3(std::integral_constant<int, 1>{}, std::integral_constant<int, 1>{});

    // ---- stage 2: mid = act(acc1 + bias1), zero outside the image, -> Y.  Wave-private transposition tile T[32][LDTP] in region A
    // (the barrier at the end of the last group1 means nobody reads the X stage any more)
    {
        float* const T = reinterpret_cast<float*>(ldsA) + wave * (32 * LDTP);
#pragma unroll
        for (int j = 0; j < MIDG; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                T[((r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * LDTP + 32 * j + (lane & 31)] = acc1[j][r];
        constexpr int UPR = MID / 8;                          // 8-channel units per row
        const int c8 = lane % UPR;
        const f32x4 b_lo = *reinterpret_cast<const f32x4*>(pre.bias + 8 * c8), b_hi = *reinterpret_cast<const f32x4*>(pre.bias + 8 * c8 + 4);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");    // (same wave wrote the tile; LDS serves a wave's requests in order)
#pragma unroll
        for (int i = 0; i < UPR / 2; ++i) {
            const int row = (lane + 64 * i) / UPR;
            const int prow = 32 * wave + row;
            const int ly = (prow * 205) >> 11, lx = prow - 10 * ly;
            const bool ok = prow < FB_NR && (unsigned)(iy0 + ly) < (unsigned)H && (unsigned)(ix0 + lx) < (unsigned)W;
            f32x4 lo = *reinterpret_cast<const f32x4*>(T + row * LDTP + 8 * c8), hi = *reinterpret_cast<const f32x4*>(T + row * LDTP + 8 * c8 + 4);
            lo = fb_act4(lo + b_lo, pre.act);
            hi = fb_act4(hi + b_hi, pre.act);
            if (!ok) { lo = f32x4{0.f, 0.f, 0.f, 0.f}; hi = lo; }
            if (prow < 104) fb_park8(ldsB + (c8 >> 2) * FB_YG_B + prow * FB_ROW_B + (c8 & 3) * 16, lo, hi);
        }
    }
    __syncthreads();

    FB_STAMP(2);
    float* const S = reinterpret_cast<float*>(ldsA);
    // ---- epilogue of a 64-column slab staged in region A: + bias, skip connection, activation, 16-B stores at the patch rows' pixels.
    // The skip connection's rows are requested EARLY (res_fetch, before the matrix work that precedes the slab): at the end of the block
    // they would be a load of > 1 us on the critical path
    unsigned ep_m[4];                                         // pixel index of this thread's four rows (OOB: outside the image)
#pragma unroll
    for (int pass = 0; pass < 4; ++pass) {
        const int r = (tid >> 4) + 16 * pass;
        const int oy = ty * 8 + (r >> 3), ox = tx * 8 + (r & 7);
        ep_m[pass] = (oy < H && ox < W) ? (unsigned)((b * H + oy) * W + ox) : OOB;
    }
    const int ep_n4 = (tid & 15) * 4;
    f32x4 rres[POST ? 2 : 1][4];
    auto res_fetch = [&](auto bufc, const ConvParams& e, int nbase) __attribute__((always_inline)) {
        constexpr int buf = decltype(bufc)::value;
        if (e.res == nullptr || nbase + ep_n4 >= e.Cout) return;
        const __amdgpu_buffer_rsrc_t rsrcR = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(e.res), 0, (int)min((long long)e.N * H * W * e.res_ld * 4, (long long)OOB), 0x00020000);
#pragma unroll
        for (int pass = 0; pass < 4; ++pass)
            rres[buf][pass] = buf_load4(rsrcR, ep_m[pass] != OOB ? (ep_m[pass] * (unsigned)e.res_ld + (unsigned)(nbase + ep_n4)) * 4u : OOB, 0);
    };
    auto store_slab = [&](auto bufc, const ConvParams& e, int nbase) __attribute__((always_inline)) {
        constexpr int buf = decltype(bufc)::value;
        const int n = nbase + ep_n4;
        if (n >= e.Cout) return;
        const f32x4 bias4 = *reinterpret_cast<const f32x4*>(e.bias + n);
        const __amdgpu_buffer_rsrc_t rsrcO = __builtin_amdgcn_make_buffer_rsrc(
            e.out, 0, (int)min((long long)e.N * H * W * e.out_ld * 4, (long long)OOB), 0x00020000);
        const int resmode = e.res ? (e.res_after_act ? 2 : 1) : 0;
#pragma unroll
        for (int pass = 0; pass < 4; ++pass) {
            const int r = (tid >> 4) + 16 * pass;
            f32x4 v = *reinterpret_cast<const f32x4*>(S + r * LDT + ep_n4) + bias4;
            if (resmode == 1) v += rres[buf][pass];
            v = fb_act4(v, e.act);
            if (resmode == 2) v += rres[buf][pass];
            const u32x4 o = {__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w)};
            __builtin_amdgcn_raw_buffer_store_b128(o, rsrcO, ep_m[pass] != OOB ? (int)((ep_m[pass] * (unsigned)e.out_ld + (unsigned)n) * 4u) : (int)OOB, 0, 0);
        }
    };

    if constexpr (!POST) res_fetch(std::integral_constant<int, 0>{}, c3, tile_n * 64);

    // ---- stage 3: the 3x3 from the resident halo.  Lane -> output row 32 wm + (lane & 31) = pixel (py, px); a tap is an immediate offset
    f32x16 acc3;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc3[r] = 0.f;
    {
        const int r3 = 32 * wm + (lane & 31);
        const unsigned fa3 = (unsigned)(((r3 >> 3) * FB_IW + (r3 & 7)) * FB_ROW_B + (lane >> 5) * 16);
        static_for<9 * MIDG>([&](auto cc) __attribute__((always_inline)) {
            constexpr int chunk = decltype(cc)::value;        // (tap, group): the stage-packed filters' K order is (tap, channel)
            constexpr int tap = chunk / MIDG, g = chunk % MIDG;
            constexpr int toff = ((tap / 3) * FB_IW + (tap % 3)) * FB_ROW_B + g * FB_YG_B;
            load_w3(std::integral_constant<int, (chunk + 2) % 3>{}, std::integral_constant<int, chunk + 2>{});
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                frag_t fa[3];
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) fa[pl] = *reinterpret_cast<const frag_t*>(ldsB + fa3 + toff + pl * 64 + ks * 32);
#pragma unroll
                for (int q = 0; q < 6; ++q)
                    acc3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[PA[q]], __builtin_bit_cast(frag_t, rb3[chunk % 3][PB[q]][ks]), acc3, 0, 0, 0);
            }
        });
    }

    FB_STAMP(3);
    if constexpr (!POST) {
        // the 3x3 is the block's last convolution
#pragma unroll
        for (int r = 0; r < 16; ++r)
            S[(32 * wm + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * LDT + 32 * wn + (lane & 31)] = acc3[r];
        __syncthreads();
        store_slab(std::integral_constant<int, 0>{}, c3, n0);
    } else {
        // ---- stage 4: z = act(acc3 + bias3) -> Z planes (region B: the halo is dead once every wave has left the tap loop), then the
        // expanding 1x1.  Its filters first: 2 channel groups x 2 k-steps x 2 column tiles x 3 planes
        const __amdgpu_buffer_rsrc_t rsrcW4 = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<unsigned short*>(post.w16s), 0, 3 * post.CoutPad * post.Kpad * 2, 0x00020000);
        res_fetch(std::integral_constant<int, 0>{}, post, 0);
        u32x4 rb4[2][2][2][3];                                // [group][k-step][column tile][plane]
        const bool have_cols = wave * 64 < post.CoutPad;
#pragma unroll
        for (int g = 0; g < 2; ++g)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl)
                        rb4[g][ks][j][pl] = fb_load_b(rsrcW4, bd_lane + (unsigned)(j * 1024),
                                                      have_cols ? wave * (post.Kpad >> 4) * 6144 + g * 12288 + ks * 6144 + pl * 2048 : (int)OOB);
#pragma unroll
        for (int r = 0; r < 16; ++r)
            S[(32 * wm + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * LDT + 32 * wn + (lane & 31)] = acc3[r];
        __syncthreads();
        {
            const int c8 = tid & 7;                           // 64 rows x 8 units: two units per thread
            const f32x4 b_lo = *reinterpret_cast<const f32x4*>(c3.bias + 8 * c8), b_hi = *reinterpret_cast<const f32x4*>(c3.bias + 8 * c8 + 4);
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int row = (tid >> 3) + 32 * i;
                f32x4 lo = *reinterpret_cast<const f32x4*>(S + row * LDT + 8 * c8), hi = *reinterpret_cast<const f32x4*>(S + row * LDT + 8 * c8 + 4);
                lo = fb_act4(lo + b_lo, c3.act);
                hi = fb_act4(hi + b_hi, c3.act);
                fb_park8(ldsB + (c8 >> 2) * (64 * FB_ROW_B) + row * FB_ROW_B + (c8 & 3) * 16, lo, hi);
            }
        }
        __syncthreads();
        f32x16 acc4[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc4[i][j][r] = 0.f;
        const unsigned fa4 = (unsigned)((lane & 31) * FB_ROW_B + (lane >> 5) * 16);
#pragma unroll
        for (int g = 0; g < 2; ++g)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                frag_t fa[3][2];
#pragma unroll
                for (int pl = 0; pl < 3; ++pl)
#pragma unroll
                    for (int i = 0; i < 2; ++i)
                        fa[pl][i] = *reinterpret_cast<const frag_t*>(ldsB + g * (64 * FB_ROW_B) + fa4 + i * (32 * FB_ROW_B) + pl * 64 + ks * 32);
#pragma unroll
                for (int q = 0; q < 6; ++q)
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j)
                            acc4[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[PA[q]][i], __builtin_bit_cast(frag_t, rb4[g][ks][j][PB[q]]), acc4[i][j], 0, 0, 0);
            }
        FB_STAMP(5);
        // ---- stage 5: four 64-column slabs through the staging tile, slab c = wave c's accumulators; the next slab's skip-connection
        // rows are requested before this slab is staged
        static_for<4>([&](auto cc) __attribute__((always_inline)) {
            constexpr int c = decltype(cc)::value;
            if (c * 64 < post.CoutPad) {                      // (block-uniform)
                if constexpr (c + 1 < 4) res_fetch(std::integral_constant<int, (c + 1) & 1>{}, post, (c + 1) * 64);
                __syncthreads();                              // the staging tile is free (slab c - 1 stored / Z written from it)
                if (wave == c) {
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j)
#pragma unroll
                            for (int r = 0; r < 16; ++r)
                                S[(32 * i + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * LDT + 32 * j + (lane & 31)] = acc4[i][j][r];
                }
                __syncthreads();
                store_slab(std::integral_constant<int, c & 1>{}, post, c * 64);
            }
        });
    }
    if (stamps) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); FB_STAMP(4); }
#undef FB_STAMP
}

// ---- host side ----------------------------------------------------------------------------------------------------------
static bool fused_stage_ok(const ConvParams& c) {
    return c.w16s != nullptr && c.mfma_mode == PREC_BF16X3 && c.Cin % 32 == 0 && c.in16 == nullptr && c.out16 == nullptr &&
           c.store_mode == ST_NHWC && c.res_scale == nullptr && c.pool_out == nullptr && (long long)3 * c.CoutPad * c.Kpad * 2 < (long long)OOB;
}

static bool fused_form(const ConvParams& pre, const ConvParams& c3, const ConvParams* post);
// pre: 1x1 / stride 1, no skip connection; c3: 3x3 / stride 1 / pad 1 reading pre's output; post (optional): 1x1 reading c3's output
bool conv_fused_eligible(const ConvParams& pre, const ConvParams& c3, const ConvParams* post) {
    if (!(fused_stage_ok(pre) && fused_stage_ok(c3)) || !fused_form(pre, c3, post)) return false;
    if (!(pre.ksize == 1 && pre.stride == 1 && pre.pad == 0 && pre.res == nullptr && pre.Kpad == pre.Cin && pre.Cin % 64 == 0 && pre.in_ld % 4 == 0)) return false;
    if (!(c3.ksize == 3 && c3.stride == 1 && c3.pad == 1 && c3.Kpad == 9 * c3.Cin && c3.Cin == pre.Cout && (c3.Cin == 32 || c3.Cin == 64))) return false;
    if (!(c3.H == pre.H && c3.W == pre.W && c3.OH == c3.H && c3.OW == c3.W && pre.OH == pre.H && pre.OW == pre.W)) return false;
    if ((long long)pre.N * pre.H * pre.W * pre.in_ld * 4 >= (long long)OOB) return false;
    if (post) {
        if (!fused_stage_ok(*post)) return false;
        if (!(post->ksize == 1 && post->stride == 1 && post->pad == 0 && post->Cin == 64 && c3.Cout == 64 && post->Kpad == 64 && c3.res == nullptr)) return false;
        if (!(post->CoutPad <= 256 && post->Cout % 4 == 0 && post->out_ld % 4 == 0 && (post->res == nullptr || post->res_ld % 4 == 0))) return false;
        if ((long long)post->N * pre.H * pre.W * post->out_ld * 4 >= (long long)OOB) return false;
    } else {
        if (!(c3.Cout % 4 == 0 && c3.out_ld % 4 == 0 && (c3.res == nullptr || c3.res_ld % 4 == 0))) return false;
        if ((long long)c3.N * pre.H * pre.W * c3.out_ld * 4 >= (long long)OOB) return false;
    }
    return true;
}

int conv_fused_blocks(const ConvParams& pre, const ConvParams& c3, const ConvParams* post) {
    const int patches = pre.N * ((pre.H + 7) / 8) * ((pre.W + 7) / 8);
    return patches * (post ? 1 : c3.CoutPad / 64);
}

template <int MIDG, bool POST, int G1>
static void launch_fused_t(const ConvParams& pre, const ConvParams& c3, const ConvParams& post, int blocks, hipStream_t s) {
    constexpr size_t lds = FusedLds<MIDG, POST>::BYTES;
    allow_big_lds(reinterpret_cast<const void*>(&conv_fused_kernel<MIDG, POST, G1>));
    if (g_conv_prof)
        hipExtLaunchKernelGGL((conv_fused_kernel<MIDG, POST, G1>), dim3(blocks), dim3(256), lds, s, g_conv_prof->e0, g_conv_prof->e1, 0, pre, c3, post);
    else
        hipLaunchKernelGGL((conv_fused_kernel<MIDG, POST, G1>), dim3(blocks), dim3(256), lds, s, pre, c3, post);
}

// the instantiated forms: {mid channels / 32, trailing 1x1, Cin of the first 1x1 / 32} -- the residual blocks of Darknet-53 (C -> C/2 -> C:
// 64 -> 32, 128 -> 64) and the bottlenecks of SE-ResNet's first stage (64 -> 64, 256 -> 64 -> 64 [-> 256])
static bool fused_form(const ConvParams& pre, const ConvParams& c3, const ConvParams* post) {
    const int g1 = pre.Cin / 32, mg = c3.Cin / 32;
    if (post) return mg == 2 && (g1 == 2 || g1 == 8);
    return (mg == 1 && g1 == 2) || (mg == 2 && (g1 == 2 || g1 == 4 || g1 == 8));
}

void launch_conv_fused(const ConvParams& pre, const ConvParams& c3, const ConvParams* post, hipStream_t s) {
    BP_CHECK(conv_fused_eligible(pre, c3, post), "fused block: bf16x3, 1x1 (Cin % 64) -> 3x3 / stride 1 (32 or 64 mid channels) [-> 1x1 from 64 channels, <= 256 out]");
    const int blocks = conv_fused_blocks(pre, c3, post);
    const ConvParams& p4 = post ? *post : c3;
    const int g1 = pre.Cin / 32;
    if (c3.Cin == 32) launch_fused_t<1, false, 2>(pre, c3, p4, blocks, s);
    else if (post) { if (g1 == 2) launch_fused_t<2, true, 2>(pre, c3, p4, blocks, s); else launch_fused_t<2, true, 8>(pre, c3, p4, blocks, s); }
    else if (g1 == 2) launch_fused_t<2, false, 2>(pre, c3, p4, blocks, s);
    else if (g1 == 4) launch_fused_t<2, false, 4>(pre, c3, p4, blocks, s);
    else launch_fused_t<2, false, 8>(pre, c3, p4, blocks, s);
}

}  // namespace bp
