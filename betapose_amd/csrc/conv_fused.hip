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
//
// NP = 3: the fp32-accurate mode as described (fp32 activations in and out).  NP = 1 (the fp16 modes, BASELINE configs[2]): the same
// block on the operand-plane data path of conv_pl.hip -- X arrives as the fp16 plane its producer wrote (16 B per 8 channels, no
// conversion), the filters are conv_pl's packed fp16 image (ConvParams::wpl) read as fragments, one v_mfma_f32_32x32x16_f16 per k-step,
// the intermediates are rounded to fp16 exactly where the unfused launches round them (the plane a producer emits = RNE of its fp32
// output), and the epilogue writes the fp16 plane (+ the fp32 tensor where somebody still reads it) and takes the skip connection from
// the fp32 tensor or from its fp16 plane (fp16 skip connections, ConvParams::res16).  At 28 frames per launch the big-map blocks are
// HBM-bound passes (profiles/r05_per_op_b28_f16r.txt: 117 + 210 us for the 208x208 block's two layers): here the block's input is read
// once and only its output written.
#include <cstdlib>

#include "conv_dev.h"

namespace bp {

template <int NP> struct FusedFmt {
    static constexpr int ROW_B = NP == 3 ? 208 : 80;      // LDS row of one 32-channel group: NP planes x 64 B + 16 B (an odd multiple of 16 B)
    static constexpr int NPL = NP == 3 ? 3 : 1;
    static constexpr int YG_B = 104 * ROW_B;              // one 32-channel group of the 3x3's halo
};
static constexpr int FB_IW = 10, FB_NR = 100;             // input patch 10 x 10

__host__ __device__ constexpr int fb_max(int a, int b) { return a > b ? a : b; }
template <int NP, int MIDG, bool POST>
struct FusedLds {
    static constexpr int MID = 32 * MIDG, LDTP = MID + 4, RB = FusedFmt<NP>::ROW_B;
    static constexpr int A_B = fb_max(fb_max(128 * RB, 128 * LDTP * 4), 64 * 68 * 4);   // X stage | pre transposition tiles | staging
    static constexpr int B_B = fb_max(MIDG * FusedFmt<NP>::YG_B, POST ? 2 * 64 * RB : 0);   // Y halo | Z
    static constexpr int BYTES = A_B + B_B;
};

__device__ __forceinline__ u32x4 fb_load_b(__amdgpu_buffer_rsrc_t r, unsigned voff, int so) {
    return __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, so, 0);
}

// 8 consecutive channels of one LDS row: exact 3-way bf16 split (three 16-B stores, planes 64 B apart) or one RNE fp16 store
template <int NP>
__device__ __forceinline__ void fb_park8(char* dst, f32x4 lo, f32x4 hi) {
    if constexpr (NP == 3) {
        const bf16x4 l1 = __builtin_convertvector(lo, bf16x4), h1 = __builtin_convertvector(hi, bf16x4);
        const f32x4 rl1 = lo - __builtin_convertvector(l1, f32x4), rh1 = hi - __builtin_convertvector(h1, f32x4);
        const bf16x4 l2 = __builtin_convertvector(rl1, bf16x4), h2 = __builtin_convertvector(rh1, bf16x4);
        const f32x4 rl2 = rl1 - __builtin_convertvector(l2, f32x4), rh2 = rh1 - __builtin_convertvector(h2, f32x4);
        const bf16x4 l3 = __builtin_convertvector(rl2, bf16x4), h3 = __builtin_convertvector(rh2, bf16x4);
        *reinterpret_cast<bf16x8*>(dst) = __builtin_shufflevector(l1, h1, 0, 1, 2, 3, 4, 5, 6, 7);
        *reinterpret_cast<bf16x8*>(dst + 64) = __builtin_shufflevector(l2, h2, 0, 1, 2, 3, 4, 5, 6, 7);
        *reinterpret_cast<bf16x8*>(dst + 128) = __builtin_shufflevector(l3, h3, 0, 1, 2, 3, 4, 5, 6, 7);
    } else {
        const f16x4 l = __builtin_convertvector(lo, f16x4), h = __builtin_convertvector(hi, f16x4);
        *reinterpret_cast<f16x8*>(dst) = __builtin_shufflevector(l, h, 0, 1, 2, 3, 4, 5, 6, 7);
    }
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

// one 16-k step of a 32x32 tile: fa / fb = the NPL planes' fragments
template <int NP>
__device__ __forceinline__ f32x16 fb_mma(const u32x4* fa, const u32x4* fb, f32x16 acc) {
    if constexpr (NP == 3) {
        constexpr int PA[6] = {2, 1, 0, 1, 0, 0};          // partial products (A plane, B plane), smallest first (conv_igemm.hip)
        constexpr int PB[6] = {0, 1, 2, 0, 1, 0};
#pragma unroll
        for (int q = 0; q < 6; ++q)
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa[PA[q]]), __builtin_bit_cast(bf16x8, fb[PB[q]]), acc, 0, 0, 0);
    } else {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, fa[0]), __builtin_bit_cast(f16x8, fb[0]), acc, 0, 0, 0);
    }
    return acc;
}

// Filter fragments of one stage (`wp` = that convolution's descriptor): 32 columns nc0 .. nc0 + 31 (nc0 a multiple of 32), 32-k chunk `chunk`
// of the packed K order, both k-steps, every plane.  NP = 3: the stage-packed planes (ConvParams::w16s, aux_kernels.hip: per 64-row tile
// and 16-k stage [plane][row][32 B], granule g of row r at slot g ^ ((r >> 3) & 1)).  NP = 1: conv_pl.hip's LDS image (ConvParams::wpl:
// per 64-row tile and chunk [row][64 B], granule g at slot g ^ ((r >> 2) & 3)).
template <int NP>
struct FusedW {
    __amdgpu_buffer_rsrc_t rsrc;
    unsigned v0;          // per-lane offset of k-step 0 inside a 64-row tile's chunk (rows 0 .. 31; + 32 rows: row32_b)
    int tile_b, chunk_b;  // bytes per 64-row tile / per chunk
    static constexpr int row32_b = NP == 3 ? 1024 : 2048;
    __device__ __forceinline__ void init(const ConvParams& c, int lane) {
        if constexpr (NP == 3) {
            rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(c.w16s), 0, 3 * c.CoutPad * c.Kpad * 2, 0x00020000);
            v0 = (unsigned)((lane & 31) * 32 + (((lane >> 5) ^ ((lane >> 3) & 1)) << 4));
            tile_b = (c.Kpad >> 4) * 6144; chunk_b = 12288;
        } else {
            rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(c.wpl), 0, c.CoutPad * c.Kpad * 2, 0x00020000);
            v0 = (unsigned)((lane & 31) * 64 + (((lane >> 5) ^ ((lane >> 2) & 3)) << 4));      // (rows 32 .. 63: the same (r >> 2) & 3)
            tile_b = (c.Kpad >> 5) * 4096; chunk_b = 4096;
        }
    }
    // out[ks][pl]
    __device__ __forceinline__ void load(u32x4 (&out)[2][FusedFmt<NP>::NPL], int nc0, int chunk, bool live = true) const {
        const int so = live ? (nc0 >> 6) * tile_b + chunk * chunk_b : (int)OOB;
        const unsigned v = v0 + (unsigned)(((nc0 >> 5) & 1) * row32_b);
        if constexpr (NP == 3) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) out[ks][pl] = fb_load_b(rsrc, v, so + ks * 6144 + pl * 2048);
        } else {
            out[0][0] = fb_load_b(rsrc, v, so);
            out[1][0] = fb_load_b(rsrc, v ^ 32u, so);          // granule 2 + h at slot (2 + h) ^ s = (h ^ s) ^ 2
        }
    }
};

// G1 = 32-channel groups of the first 1x1's input (2, 4 or 8): its loop is unrolled, the X rows of up to four groups are in flight
// (registers) -- one group ahead left ~2 us of cold-fetch latency exposed per group (first GPU run of round 5)
template <int NP, int MIDG, bool POST, int G1>
__global__ __launch_bounds__(256) void conv_fused_kernel(const ConvParams pre, const ConvParams c3, const ConvParams post) {
    typedef FusedLds<NP, MIDG, POST> L;
    typedef FusedFmt<NP> F;
    constexpr int MID = L::MID, LDTP = L::LDTP, RB = F::ROW_B, NPL = F::NPL, YG_B = F::YG_B;
    constexpr int LDT = 68;                                // staging rows (floats) of the 64-wide tiles
    extern __shared__ __attribute__((aligned(16))) float smem[];
    char* const lds = reinterpret_cast<char*>(smem);
    char* const ldsA = lds;
    char* const ldsB = lds + L::A_B;

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
    FusedW<NP> W1;
    W1.init(pre, lane);
    u32x4 rb1[2][MIDG][2][NPL];                              // [group parity][column tile][k-step][plane]
    auto load_w1 = [&](auto parc, auto gc) __attribute__((always_inline)) {
        constexpr int par = decltype(parc)::value, g = decltype(gc)::value;
        if constexpr (g < G1) {
#pragma unroll
            for (int j = 0; j < MIDG; ++j) W1.load(rb1[par][j], 32 * j, g);
        }
    };
    load_w1(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
    // every bias the block will need is requested NOW: read where it is used, each was a cold load of 1-2 us on the critical path (stage marks
    // of the first version, tools/fused_stamps.py: 2.5-3.3 us for the transposition, 7 us for four slabs of stores)
    constexpr int UPR = MID / 8;                              // 8-channel units per row of the intermediate
    const f32x4 b1_lo = *reinterpret_cast<const f32x4*>(pre.bias + 8 * (lane % UPR)), b1_hi = *reinterpret_cast<const f32x4*>(pre.bias + 8 * (lane % UPR) + 4);
    const f32x4 b3_lo = *reinterpret_cast<const f32x4*>(c3.bias + (POST ? 8 * (tid & 7) : min(tile_n * 64 + (tid & 15) * 4, c3.CoutPad - 4)));
    const f32x4 b3_hi = *reinterpret_cast<const f32x4*>(c3.bias + (POST ? 8 * (tid & 7) + 4 : 0));       // (bottleneck form: the 3x3's bias by 8-channel unit; else its four epilogue channels in b3_lo)
    __builtin_amdgcn_sched_barrier(0);

    // the X patch: thread -> patch row (tid >> 2) + 64 j, 8 channels (tid & 3) of the group (fp32 tensor, or its fp16 plane)
    constexpr int XE = NP == 3 ? 4 : 2;                       // bytes per element of what is read
    const __amdgpu_buffer_rsrc_t rsrcX = __builtin_amdgcn_make_buffer_rsrc(
        NP == 3 ? (void*)const_cast<float*>(pre.in) : (void*)const_cast<unsigned short*>(pre.in16), 0,
        (int)min((long long)pre.N * H * W * pre.in_ld * XE, (long long)OOB), 0x00020000);
    unsigned x_voff[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int row = (tid >> 2) + 64 * j;
        const int ly = (row * 205) >> 11, lx = row - 10 * ly;
        const int iy = iy0 + ly, ix = ix0 + lx;
        const bool ok = row < FB_NR && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
        x_voff[j] = ok ? (unsigned)((((b * H + iy) * W + ix) * pre.in_ld + (tid & 3) * 8) * XE) : OOB;
    }
    const bool x_row1 = (tid >> 2) + 64 < FB_NR;              // the second pass covers patch rows 64 .. 99 only
    constexpr int XD = G1 < 4 ? G1 : 4;                       // groups of X rows in flight
    constexpr int XV = NP == 3 ? 2 : 1;                       // 16-B loads per row and group
    u32x4 rx[XD][2][XV];
    auto load_x = [&](auto gc) __attribute__((always_inline)) {
        constexpr int g = decltype(gc)::value;
        if constexpr (g < G1) {
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int v = 0; v < XV; ++v)
                    rx[g % XD][j][v] = __builtin_amdgcn_raw_buffer_load_b128(rsrcX, (int)x_voff[j], g * (32 * XE) + 16 * v, 0);
        }
    };
    auto park_x = [&](auto gc) __attribute__((always_inline)) {
        constexpr int sl = decltype(gc)::value % XD;
        char* dst = ldsA + (tid >> 2) * RB + (tid & 3) * 16;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            if (j == 1 && !x_row1) break;
            if constexpr (NP == 3) fb_park8<3>(dst + j * 64 * RB, __builtin_bit_cast(f32x4, rx[sl][j][0]), __builtin_bit_cast(f32x4, rx[sl][j][XV - 1]));
            else *reinterpret_cast<u32x4*>(dst + j * 64 * RB) = rx[sl][j][0];
        }
    };
    static_for<XD>([&](auto gc) __attribute__((always_inline)) { load_x(gc); });
    // patch rows 100 .. 127 of the X stage are never written: zero them once (the first GEMM reads 128 rows; rows >= 100 are dropped later)
    for (int u = tid; u < 28 * (RB / 16); u += 256) *reinterpret_cast<u32x4*>(ldsA + FB_NR * RB + u * 16) = u32x4{0u, 0u, 0u, 0u};

    f32x16 acc1[MIDG];
#pragma unroll
    for (int j = 0; j < MIDG; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc1[j][r] = 0.f;
    const unsigned fa1 = (unsigned)((32 * wave + (lane & 31)) * RB + (lane >> 5) * 16);   // this wave's 32 patch rows

    park_x(std::integral_constant<int, 0>{});
    load_x(std::integral_constant<int, XD>{});                // (slot 0 is free again)
    __syncthreads();
    // one 32-channel group: the next group's filters are requested first, then 2 k-steps of MFMAs on the parked group
    static_for<G1>([&](auto gc) __attribute__((always_inline)) {
        constexpr int g = decltype(gc)::value, par = g & 1;
        load_w1(std::integral_constant<int, par ^ 1>{}, std::integral_constant<int, g + 1>{});
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            u32x4 fa[NPL];
#pragma unroll
            for (int pl = 0; pl < NPL; ++pl) fa[pl] = *reinterpret_cast<const u32x4*>(ldsA + fa1 + pl * 64 + ks * 32);
#pragma unroll
            for (int j = 0; j < MIDG; ++j) acc1[j] = fb_mma<NP>(fa, rb1[par][j][ks], acc1[j]);
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
    // ---- stage 3 filters: requested now, they arrive under the transposition.  Wave = (row half wm, column half wn) of the 64 x 64 tile.
    // (A second form -- every wave all 64 rows x 32 columns of one of two K groups, so that a filter fragment feeds two MFMAs and no two waves
    // fetch the same one -- shortened the stage by 15 % per block and LOST: 168 instead of 118 registers put the 676-block launch of the
    // 208x208 block into two rounds, 23.0 against 17.6 us; profiles/r05_fused_blocks.txt.)
    const int wm = wave >> 1, wn = wave & 1;
    const int n0 = tile_n * 64;
    FusedW<NP> W3;
    W3.init(c3, lane);
    u32x4 rb3[3][2][NPL];                                     // [ring slot][k-step][plane]
    auto load_w3 = [&](auto slotc, auto chunkc) __attribute__((always_inline)) {
        constexpr int slot = decltype(slotc)::value, chunk = decltype(chunkc)::value;     // chunk = tap * MIDG + group (this loop's order)
        if constexpr (chunk < 9 * MIDG) {
            constexpr int tap = chunk / MIDG, g = chunk % MIDG;
            W3.load(rb3[slot], n0 + 32 * wn, NP == 3 ? tap * MIDG + g : g * 9 + tap);      // (the two filter images' K orders)
        }
    };
    load_w3(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
    load_w3(std::integral_constant<int, 1>{}, std::integral_constant<int, 1>{});

    // ---- stage 2: mid = act(acc1 + bias1), zero outside the image, -> Y.  Wave-private transposition tile T[32][LDTP] in region A
    // (the barrier at the end of the last group means nobody reads the X stage any more)
    {
        float* const T = reinterpret_cast<float*>(ldsA) + wave * (32 * LDTP);
#pragma unroll
        for (int j = 0; j < MIDG; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                T[((r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * LDTP + 32 * j + (lane & 31)] = acc1[j][r];
        const int c8 = lane % UPR;
        const f32x4 b_lo = b1_lo, b_hi = b1_hi;
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
            if (prow < 104) fb_park8<NP>(ldsB + (c8 >> 2) * YG_B + prow * RB + (c8 & 3) * 16, lo, hi);
        }
    }
    __syncthreads();
    FB_STAMP(2);

    // ---- epilogue of a 64-column slab staged in region A: + bias, skip connection, activation, 16-B stores at the patch rows' pixels
    // (fp32 tensor and / or operand planes, conv_dev.h emit_planes4).  The skip connection's rows are requested EARLY (res_fetch, before the
    // matrix work that precedes the slab): at the end of the block they would be a load of > 1 us on the critical path
    float* const S = reinterpret_cast<float*>(ldsA);
    unsigned ep_m[4];                                         // pixel index of this thread's four rows (OOB: outside the image)
#pragma unroll
    for (int pass = 0; pass < 4; ++pass) {
        const int r = (tid >> 4) + 16 * pass;
        const int oy = ty * 8 + (r >> 3), ox = tx * 8 + (r & 7);
        ep_m[pass] = (oy < H && ox < W) ? (unsigned)((b * H + oy) * W + ox) : OOB;
    }
    const int ep_n4 = (tid & 15) * 4;
    f32x4 rres[POST ? 4 : 1][4], rbias[POST ? 4 : 1];      // (bottleneck form: all four slabs' rows are requested before the 3x3 -- one slab ahead left 1.9 us per slab exposed)
    auto res_fetch = [&](auto bufc, const ConvParams& e, int nbase) __attribute__((always_inline)) {
        constexpr int buf = decltype(bufc)::value;
        if constexpr (POST) rbias[buf] = *reinterpret_cast<const f32x4*>(e.bias + min(nbase + ep_n4, e.CoutPad - 4));
        else rbias[buf] = b3_lo;
        if (e.res == nullptr || nbase + ep_n4 >= e.Cout) return;
        const bool r16 = e.res16 != nullptr;
        const unsigned rb_ = r16 ? 2u : 4u;
        const __amdgpu_buffer_rsrc_t rsrcR = __builtin_amdgcn_make_buffer_rsrc(
            r16 ? (void*)const_cast<unsigned short*>(e.res16) : (void*)const_cast<float*>(e.res), 0,
            (int)min((long long)e.N * H * W * e.res_ld * rb_, (long long)OOB), 0x00020000);
#pragma unroll
        for (int pass = 0; pass < 4; ++pass)
            rres[buf][pass] = load_res4(rsrcR, ep_m[pass] != OOB ? (ep_m[pass] * (unsigned)e.res_ld + (unsigned)(nbase + ep_n4)) * rb_ : OOB, r16);
    };
    auto store_slab = [&](auto bufc, const ConvParams& e, int nbase) __attribute__((always_inline)) {
        constexpr int buf = decltype(bufc)::value;
        const int n = nbase + ep_n4;
        if (n >= e.Cout) return;
        const f32x4 bias4 = rbias[buf];
        const __amdgpu_buffer_rsrc_t rsrcO = __builtin_amdgcn_make_buffer_rsrc(
            e.out, 0, (int)min((long long)e.N * H * W * e.out_ld * 4, (long long)OOB), 0x00020000);
        const PlaneDesc pd = make_plane_desc(e);
        const int resmode = e.res ? (e.res_after_act ? 2 : 1) : 0;
#pragma unroll
        for (int pass = 0; pass < 4; ++pass) {
            const int r = (tid >> 4) + 16 * pass;
            f32x4 v = *reinterpret_cast<const f32x4*>(S + r * LDT + ep_n4) + bias4;
            if (resmode == 1) v += rres[buf][pass];
            v = fb_act4(v, e.act);
            if (resmode == 2) v += rres[buf][pass];
            const u32x4 o = {__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w)};
            const unsigned off = ep_m[pass] != OOB ? (ep_m[pass] * (unsigned)e.out_ld + (unsigned)n) * 4u : OOB;
            if (pd.f32) __builtin_amdgcn_raw_buffer_store_b128(o, rsrcO, (int)off, 0, 0);
            emit_planes4(pd, v, off >> 1);      // (an out-of-range fp32 offset halves to an out-of-range plane offset: OOB >> 1 > any plane)
        }
    };
    if constexpr (!POST) res_fetch(std::integral_constant<int, 0>{}, c3, tile_n * 64);
    else static_for<4>([&](auto cc) __attribute__((always_inline)) { if (decltype(cc)::value * 64 < post.CoutPad) res_fetch(cc, post, decltype(cc)::value * 64); });

    // ---- stage 3: the 3x3 from the resident halo.  Lane -> output row 32 wm + (lane & 31) = pixel (py, px); a tap is an immediate offset
    f32x16 acc3;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc3[r] = 0.f;
    {
        const int r3 = 32 * wm + (lane & 31);
        const unsigned fa3 = (unsigned)(((r3 >> 3) * FB_IW + (r3 & 7)) * RB + (lane >> 5) * 16);
        static_for<9 * MIDG>([&](auto cc) __attribute__((always_inline)) {
            constexpr int chunk = decltype(cc)::value;
            constexpr int tap = chunk / MIDG, g = chunk % MIDG;
            constexpr int toff = ((tap / 3) * FB_IW + (tap % 3)) * RB + g * YG_B;
            load_w3(std::integral_constant<int, (chunk + 2) % 3>{}, std::integral_constant<int, chunk + 2>{});
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                u32x4 fa[NPL];
#pragma unroll
                for (int pl = 0; pl < NPL; ++pl) fa[pl] = *reinterpret_cast<const u32x4*>(ldsB + fa3 + toff + pl * 64 + ks * 32);
                acc3 = fb_mma<NP>(fa, rb3[chunk % 3][ks], acc3);
            }
        });
    }
    // the 64 x 64 tile is staged in region A (free since the barrier behind stage 2)
#pragma unroll
    for (int r = 0; r < 16; ++r)
        S[(32 * wm + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * LDT + 32 * wn + (lane & 31)] = acc3[r];
    FB_STAMP(3);

    if constexpr (!POST) {
        // the 3x3 is the block's last convolution
        __syncthreads();
        store_slab(std::integral_constant<int, 0>{}, c3, n0);
    } else {
        // ---- stage 4: z = act(acc3 + bias3) -> Z (region B: the halo is dead once every wave has left the tap loop), then the
        // expanding 1x1.  Its filters first: 2 channel groups x 2 column tiles
        FusedW<NP> W4;
        W4.init(post, lane);
        u32x4 rb4[2][2][2][NPL];                              // [group][column tile][k-step][plane]
        const bool have_cols = wave * 64 < post.CoutPad;
#pragma unroll
        for (int g = 0; g < 2; ++g)
#pragma unroll
            for (int j = 0; j < 2; ++j) W4.load(rb4[g][j], 64 * wave + 32 * j, g, have_cols);
        __syncthreads();
        {
            const int c8 = tid & 7;                           // 64 rows x 8 units: two units per thread
            const f32x4 b_lo = b3_lo, b_hi = b3_hi;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int row = (tid >> 3) + 32 * i;
                f32x4 lo = *reinterpret_cast<const f32x4*>(S + row * LDT + 8 * c8), hi = *reinterpret_cast<const f32x4*>(S + row * LDT + 8 * c8 + 4);
                lo = fb_act4(lo + b_lo, c3.act);
                hi = fb_act4(hi + b_hi, c3.act);
                fb_park8<NP>(ldsB + (c8 >> 2) * (64 * RB) + row * RB + (c8 & 3) * 16, lo, hi);
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
        const unsigned fa4 = (unsigned)((lane & 31) * RB + (lane >> 5) * 16);
#pragma unroll
        for (int g = 0; g < 2; ++g)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                u32x4 fa[2][NPL];
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int pl = 0; pl < NPL; ++pl)
                        fa[i][pl] = *reinterpret_cast<const u32x4*>(ldsB + g * (64 * RB) + fa4 + i * (32 * RB) + pl * 64 + ks * 32);
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc4[i][j] = fb_mma<NP>(fa[i], rb4[g][j][ks], acc4[i][j]);
            }
        FB_STAMP(5);
        // ---- stage 5: four 64-column slabs through the staging tile, slab c = wave c's accumulators
        static_for<4>([&](auto cc) __attribute__((always_inline)) {
            constexpr int c = decltype(cc)::value;
            if (c * 64 < post.CoutPad) {                      // (block-uniform)
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
                store_slab(cc, post, c * 64);
            }
        });
    }
    if (stamps) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); FB_STAMP(4); }
#undef FB_STAMP
}

// ---- host side ----------------------------------------------------------------------------------------------------------
// the arithmetic a group runs in: 3 = bf16x3 on fp32 activations, 1 = fp16 on the operand planes, 0 = not fusable
static int fused_np(const ConvParams& c) {
    if (c.mfma_mode == PREC_BF16X3 && c.w16s != nullptr && c.in16 == nullptr && c.out16 == nullptr) return 3;
    if (c.mfma_mode == PREC_F16 && c.wpl != nullptr && c.in16 != nullptr) return 1;
    return 0;
}
static bool fused_stage_ok(const ConvParams& c, int np) {
    return fused_np(c) == np && c.Cin % 32 == 0 && c.store_mode == ST_NHWC && c.res_scale == nullptr && c.pool_out == nullptr &&
           (long long)3 * c.CoutPad * c.Kpad * 2 < (long long)OOB;
}
// the last member's outputs: fp32 tensor and / or fp16 plane, 16-B stores; the skip connection from the fp32 tensor or its fp16 plane
static bool fused_out_ok(const ConvParams& e, int np, int H, int W) {
    if (!(e.Cout % 4 == 0 && e.out_ld % 4 == 0 && (e.res == nullptr || e.res_ld % 4 == 0))) return false;
    if (np == 1 && !(e.out16 == nullptr || e.out_np == 1)) return false;
    if (np == 1 && e.out16 == nullptr && e.skip_f32) return false;
    if (np == 3 && e.res16 != nullptr) return false;
    return (long long)e.N * H * W * e.out_ld * 4 < (long long)OOB && (e.res == nullptr || (long long)e.N * H * W * e.res_ld * 4 < (long long)OOB);
}

static bool fused_form(const ConvParams& pre, const ConvParams& c3, const ConvParams* post);
// pre: 1x1 / stride 1, no skip connection; c3: 3x3 / stride 1 / pad 1 reading pre's output; post (optional): 1x1 reading c3's output
bool conv_fused_eligible(const ConvParams& pre, const ConvParams& c3, const ConvParams* post) {
    const int np = fused_np(pre);
    if (np == 0 || !(fused_stage_ok(pre, np) && fused_stage_ok(c3, np)) || !fused_form(pre, c3, post)) return false;
    if (!(pre.ksize == 1 && pre.stride == 1 && pre.pad == 0 && pre.res == nullptr && pre.Kpad == pre.Cin && pre.Cin % 64 == 0 && pre.in_ld % (np == 1 ? 8 : 4) == 0)) return false;
    if (np == 1 && (reinterpret_cast<uintptr_t>(pre.in16) & 15) != 0) return false;
    if (!(c3.ksize == 3 && c3.stride == 1 && c3.pad == 1 && c3.Kpad == 9 * c3.Cin && c3.Cin == pre.Cout && (c3.Cin == 32 || c3.Cin == 64))) return false;
    if (!(c3.H == pre.H && c3.W == pre.W && c3.OH == c3.H && c3.OW == c3.W && pre.OH == pre.H && pre.OW == pre.W)) return false;
    if ((long long)pre.N * pre.H * pre.W * pre.in_ld * 4 >= (long long)OOB) return false;
    if (post) {
        if (!fused_stage_ok(*post, np)) return false;
        if (!(post->ksize == 1 && post->stride == 1 && post->pad == 0 && post->Cin == 64 && c3.Cout == 64 && post->Kpad == 64 && c3.res == nullptr)) return false;
        if (!(post->CoutPad <= 256 && fused_out_ok(*post, np, pre.H, pre.W))) return false;
    } else if (!fused_out_ok(c3, np, pre.H, pre.W)) return false;
    return true;
}

int conv_fused_blocks(const ConvParams& pre, const ConvParams& c3, const ConvParams* post) {
    const int patches = pre.N * ((pre.H + 7) / 8) * ((pre.W + 7) / 8);
    return patches * (post ? 1 : c3.CoutPad / 64);
}

template <int NP, int MIDG, bool POST, int G1>
static void launch_fused_t(const ConvParams& pre, const ConvParams& c3, const ConvParams& post, int blocks, hipStream_t s) {
    constexpr size_t lds = FusedLds<NP, MIDG, POST>::BYTES;
    allow_big_lds(reinterpret_cast<const void*>(&conv_fused_kernel<NP, MIDG, POST, G1>));
    if (g_conv_prof)
        hipExtLaunchKernelGGL((conv_fused_kernel<NP, MIDG, POST, G1>), dim3(blocks), dim3(256), lds, s, g_conv_prof->e0, g_conv_prof->e1, 0, pre, c3, post);
    else
        hipLaunchKernelGGL((conv_fused_kernel<NP, MIDG, POST, G1>), dim3(blocks), dim3(256), lds, s, pre, c3, post);
}

// the instantiated forms: {mid channels / 32, trailing 1x1, Cin of the first 1x1 / 32} -- the residual blocks of Darknet-53 (C -> C/2 -> C:
// 64 -> 32, 128 -> 64) and the bottlenecks of SE-ResNet's first stage (64 -> 64, 256 -> 64 -> 64 [-> 256])
static bool fused_form(const ConvParams& pre, const ConvParams& c3, const ConvParams* post) {
    const int g1 = pre.Cin / 32, mg = c3.Cin / 32;
    if (post) return mg == 2 && (g1 == 2 || g1 == 8);
    // (fp16: the 128 -> 64 -> 128 block is 157 us fused against 35 + 110 us unfused at 28 frames per launch -- its two 64-wide N tiles recompute the
    // 1x1 and the 128x128 plane tile of the 3x3 is the better kernel there; the 64 -> 32 -> 64 block is 155 against 117 + 210 us.  bf16x3 at
    // batch 1: 21.8 against 6.3 + 18.6 us)
    if (fused_np(pre) == 1 && g1 == 4 && !std::getenv("BP_FUSE_F16_ALL")) return false;
    return (mg == 1 && g1 == 2) || (mg == 2 && (g1 == 2 || g1 == 4 || g1 == 8));
}

template <int NP>
static void launch_fused_np(const ConvParams& pre, const ConvParams& c3, const ConvParams* post, hipStream_t s) {
    const int blocks = conv_fused_blocks(pre, c3, post);
    const ConvParams& p4 = post ? *post : c3;
    const int g1 = pre.Cin / 32;
    if (c3.Cin == 32) launch_fused_t<NP, 1, false, 2>(pre, c3, p4, blocks, s);
    else if (post) { if (g1 == 2) launch_fused_t<NP, 2, true, 2>(pre, c3, p4, blocks, s); else launch_fused_t<NP, 2, true, 8>(pre, c3, p4, blocks, s); }
    else if (g1 == 2) launch_fused_t<NP, 2, false, 2>(pre, c3, p4, blocks, s);
    else if (g1 == 4) launch_fused_t<NP, 2, false, 4>(pre, c3, p4, blocks, s);
    else launch_fused_t<NP, 2, false, 8>(pre, c3, p4, blocks, s);
}

void launch_conv_fused(const ConvParams& pre, const ConvParams& c3, const ConvParams* post, hipStream_t s) {
    BP_CHECK(conv_fused_eligible(pre, c3, post), "fused block: bf16x3 or fp16, 1x1 (Cin % 64) -> 3x3 / stride 1 (32 or 64 mid channels) [-> 1x1 from 64 channels, <= 256 out]");
    if (fused_np(pre) == 3) launch_fused_np<3>(pre, c3, post, s);
    else launch_fused_np<1>(pre, c3, post, s);
}

}  // namespace bp
