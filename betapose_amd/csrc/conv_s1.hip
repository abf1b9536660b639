// The 1x1 convolutions of the BATCHED fp16 runs (BASELINE configs[2]: 28 frames per launch) as a PERSISTENT streaming kernel (round 5).
//
// Same contract as conv_pl.hip for the layers it takes (1x1, stride 1 or 2, NHWC store; bias / LeakyReLU / ReLU / skip connection before or
// after the activation / SE scale of the skip connection; fp16 plane and / or fp32 tensor out; yolo/darknet.py:240-259, SE_Resnet.py:25-42),
// same operands (the producer's fp16 plane, conv_pl.hip's packed filter image), same MFMA sequence per output element (chunks in order, two
// v_mfma_f32_32x32x16_f16 per 32-k chunk, bias added to the finished sum) -- so its results are bit-identical to TILE_PL64's (the K = 1 024
// form, which adds two half-K chains, at the accumulation-order bar: see KH below).
//
// Why: at batch 28 these layers are memory streams (AI 60-250 FLOP/B), and on the 64x64 plane tile they ran at 1.5-2.3 TB/s of HBM traffic.
// In-kernel stamps: a block's life is 7 us for one 64x64x256 tile (0.9 us of index math, 3.5 us for four stages whose first operands come
// cold, 2.8 us of epilogue), three such blocks per CU; and every 64-row tile pulls its own copy of the filters L2 -> LDS: 151 MB of LDS-DMA for
// the 52x52 256 -> 128 layer whose HBM traffic is 58 MB.  Two forms of this kernel that kept streaming the filters through an LDS ring (per
// 32-row tile) were bit-exact and SLOWER than the tile they replace (37 us against 28 us) for exactly that reason.  What runs now:
//   * a wave keeps the FILTER FRAGMENTS of its 32 output columns IN REGISTERS for the whole kernel (K / 4 VGPRs: 64 at K = 256, 128 at
//     K = 512): read once per block, never again -- no filter traffic, no filter waits, no per-stage barrier;
//   * a block (4 compute waves = 128 columns, + 1 loader wave) is PERSISTENT over the 32-row M-tiles of its column group; the tiles'
//     activations (2 KB per 32-k chunk) arrive in LDS by LDS-DMA one tile ahead (two buffers), the skip-connection tile (+ SE scales) too;
//   * everything that comes from HBM is issued by the LOADER wave, which does nothing else: s_waitcnt vmcnt counts in order, so a wave that
//     mixed these fetches with anything short would wait for HBM every time.  The loader keeps a SCOREBOARD -- the number of vector-memory
//     instructions it has issued and, per buffer, that number at the time of the buffer's fetch (one VGPR, buffer = lane) -- and waits with
//     s_waitcnt vmcnt(issued - that); the compute waves never wait for memory at all (their stores are fire-and-forget);
//   * ONE s_barrier per M-tile: "tile i has landed" for the compute waves, "tile i - 1's buffers are free" for the loader;
//   * the epilogue is wave-private (each wave owns 32 columns and its own fp32 staging half-tile).
#include <algorithm>
#include <cstdlib>

#include "conv_dev.h"

namespace bp {

struct S1Args {
    int MT;               // M-tiles of 32 rows
    int NG;               // column groups of 128 (one per block)
    int MS;               // M-tile slots: block (xcd, ng, ms) takes the M-tiles (ms + k MS) 8 + xcd, k = 0, 1, ...
    int NL;               // activation tiles in LDS (look-ahead)
    int res_bytes;        // skip connection: 0 none, 2 from its fp16 plane, 4 from the fp32 tensor
    int rq;               // skip-connection buffers (tiles of look-ahead): 1 or 2
    int rbuf;             // bytes of one such buffer (the block's 32 x 128 tile + 1 KB of SE scales when the layer has them)
    int off_stg, off_res, off_bias;    // LDS byte offsets (the activation tiles start at 0)
    int hw;               // OH * OW
};

static constexpr int S1_BM = 32, S1_BN = 128;
static constexpr int S1_STG_FLOATS = 16 * 36;       // a wave's staging half-tile: 16 rows x (32 + 4) floats

// wait until at most `allowed` of this wave's vector-memory instructions are outstanding (they return in order)
#define S1_W(n) case n: asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory"); break;
__device__ __forceinline__ void s1_vm_wait(int allowed) {
    const int a = __builtin_amdgcn_readfirstlane(allowed);
    switch (a < 0 ? 0 : (a > 60 ? 60 : a)) {
        S1_W(0) S1_W(1) S1_W(2) S1_W(3) S1_W(4) S1_W(5) S1_W(6) S1_W(7) S1_W(8) S1_W(9) S1_W(10) S1_W(11) S1_W(12) S1_W(13) S1_W(14) S1_W(15)
        S1_W(16) S1_W(17) S1_W(18) S1_W(19) S1_W(20) S1_W(21) S1_W(22) S1_W(23) S1_W(24) S1_W(25) S1_W(26) S1_W(27) S1_W(28) S1_W(29) S1_W(30)
        S1_W(31) S1_W(32) S1_W(33) S1_W(34) S1_W(35) S1_W(36) S1_W(37) S1_W(38) S1_W(39) S1_W(40) S1_W(41) S1_W(42) S1_W(43) S1_W(44) S1_W(45)
        S1_W(46) S1_W(47) S1_W(48) S1_W(49) S1_W(50) S1_W(51) S1_W(52) S1_W(53) S1_W(54) S1_W(55) S1_W(56) S1_W(57) S1_W(58) S1_W(59)
        default: asm volatile("s_waitcnt vmcnt(60)" ::: "memory"); break;
    }
}
#undef S1_W

// NCH: 32-k chunks of the layer's K (compile-time: the filter fragments are a register array).
// KH = 2 (K = 1 024): a wave's 32 columns x K would be 256 registers, so the block takes 64 columns and its four compute waves are 2 column
// halves x 2 K HALVES (128 filter registers each); behind the tile's MFMAs the two K halves of a column half swap partial sums through their
// staging tiles -- the wave of the low K half finishes rows 0-15, the other one rows 16-31 (the sum of the two halves is the same number either
// way; it is NOT the 64x64 plane tile's one-chain sum: equal within the accumulation-order bar, not bit for bit).
template <int NCH, int KH>
__global__ __launch_bounds__(384) void conv_s1_kernel(const ConvParams p, const S1Args a) {
    constexpr int GW = S1_BN / KH;          // columns of the block's group
    constexpr int NCW = NCH / KH;           // chunks per compute wave
    // the ONE LDS object of the kernel (conv_pl.hip: a second one makes hipcc drain vmcnt before every fragment read)
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);     // 0-3: compute waves (32 columns each), 4 (and 5): the loader wave(s)
    const int nlw = (int)(blockDim.x >> 6) - 4;                 // loader waves: 1, or 2 (K >= 384: loader l fetches the chunks c with c % 2 == l)
    const int oR = a.off_res;
    float* const ldsBias = reinterpret_cast<float*>(lds + a.off_bias);
    const int rrow = GW * a.res_bytes;                      // bytes of a row of the skip-connection tile
    const int rtile = 32 * rrow;                            // ... of the tile inside a buffer; the scales sit behind it
    // block -> (XCD, column group, M-tile slot): the column groups of an M-tile run on ONE XCD (its activations are fetched from HBM once)
    const int xcd = (int)blockIdx.x & 7, bq = (int)blockIdx.x >> 3;
    const int ng = bq % a.NG, ms = bq / a.NG;
    const int MT8 = (a.MT + 7) >> 3;
    // M-tiles of this block: (ms + k MS) 8 + xcd for k = 0 .. nit - 1 (the last group of eight may end before xcd)
    int nit = ms < MT8 ? (MT8 - 1 - ms) / a.MS + 1 : 0;
    if (nit > 0 && (ms + (nit - 1) * a.MS) * 8 + xcd >= a.MT) --nit;
    const int n0 = ng * GW;

    // the column group's bias (padded to CoutPad by the engine) -> LDS, once
    if (tid < GW) ldsBias[tid] = n0 + tid < p.CoutPad ? p.bias[n0 + tid] : 0.f;
    const bool has_res = a.res_bytes != 0, r16 = a.res_bytes == 2, has_scale = p.res_scale != nullptr;
    const float rcp_hw = 1.0f / (float)a.hw, rcp_ow = 1.0f / (float)p.OW;

    if (w >= 4) {
        // ================= the LOADER wave(s): everything that comes from HBM (one wave issues a 1 KB piece every ~100 cycles, i.e. ~20 GB/s:
        // a 32 KB tile of a K = 512 layer per 1.5 us -- two loaders there)
        const int li = w - 4;
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __syncthreads();
        if (nit == 0) return;
        const __amdgpu_buffer_rsrc_t rsrcA = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<unsigned short*>(p.in16), 0, (int)min((long long)p.N * p.H * p.W * p.in_ld * 2, (long long)OOB), 0x00020000);
        const __amdgpu_buffer_rsrc_t rsrcR = __builtin_amdgcn_make_buffer_rsrc(
            r16 ? (void*)const_cast<unsigned short*>(p.res16) : (void*)const_cast<float*>(p.res ? p.res : p.out), 0,
            (int)min((long long)p.M * (p.res ? p.res_ld : p.out_ld) * (r16 ? 2 : 4), (long long)OOB), 0x00020000);
        const __amdgpu_buffer_rsrc_t rsrcS = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(has_scale ? p.res_scale : p.bias), 0, has_scale ? p.N * p.Cout * 4 : 0, 0x00020000);
        int vm = 0;                 // vector-memory instructions this wave has issued
        int seqa = 0, seqr = 0;     // lane j: `vm` right behind the fetch into activation buffer j / skip-connection buffer j
        // the activations of the block's tile number k -> buffer k % NL: NCH chunks of [32 rows][64 B] (conv_pl.hip's swizzled image)
        auto a_issue = [&](int k) __attribute__((always_inline)) {
            if (k >= nit) return;
            const int mt = (ms + k * a.MS) * 8 + xcd, j = k % a.NL;
            const int gsw = (lane & 3) ^ ((lane >> 4) & 3);        // swizzle on the source, as in conv_pl.hip
            unsigned vo[2];
#pragma unroll
            for (int rg = 0; rg < 2; ++rg) {
                const int m = mt * S1_BM + 16 * rg + (lane >> 2);
                long long pix = m;
                if (p.stride != 1) {
                    const int mm = m < p.M ? m : 0;
                    const int b = fast_div(mm, a.hw, rcp_hw);
                    const int rem = mm - b * a.hw;
                    const int oy = fast_div(rem, p.OW, rcp_ow), ox = rem - oy * p.OW;
                    pix = ((long long)b * p.H + oy * p.stride) * p.W + ox * p.stride;
                }
                vo[rg] = m < p.M ? (unsigned)((pix * p.in_ld + gsw * 8) * 2) : OOB;
            }
            char* const dst = lds + j * (NCH * 2048);
            int cnt = 0;
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                if (nlw == 2 && (c & 1) != li) continue;
                dma16(rsrcA, dst + c * 2048, vo[0], c * 64);
                dma16(rsrcA, dst + c * 2048 + 1024, vo[1], c * 64);
                cnt += 2;
            }
            vm += cnt;
            seqa = lane == j ? vm : seqa;
        };
        // the skip-connection tile (32 rows x the group's 128 columns) of tile number k -> buffer k % rq
        auto r_issue = [&](int k) __attribute__((always_inline)) {
            if (!has_res || li != 0 || k >= nit) return;
            const int mt = (ms + k * a.MS) * 8 + xcd, j = k % a.rq;
            const int m0 = mt * S1_BM;
            char* const dst = lds + oR + j * a.rbuf;
            if (r16) {        // [32 rows][GW x 2 B]: a piece = 1 KB = 512 / GW rows, lane -> (row, 16 B = 8 columns)
                constexpr int LPR = GW / 8, RPP = 64 / LPR, NP_ = 32 / RPP;      // lanes per row, rows per piece, pieces
#pragma unroll
                for (int i = 0; i < NP_; ++i) {
                    const int m = m0 + RPP * i + lane / LPR, n = n0 + 8 * (lane % LPR);
                    dma16(rsrcR, dst + i * 1024, (m < p.M && n < p.Cout) ? (unsigned)((m * p.res_ld + n) * 2) : OOB, 0);
                }
                vm += NP_;
            } else {          // [32 rows][GW x 4 B]: lane -> (row, 16 B = 4 columns)
                constexpr int LPR = GW / 4, RPP = 64 / LPR, NP_ = 32 / RPP;
#pragma unroll
                for (int i = 0; i < NP_; ++i) {
                    const int m = m0 + RPP * i + lane / LPR, n = n0 + 4 * (lane % LPR);
                    dma16(rsrcR, dst + i * 1024, (m < p.M && n < p.Cout) ? (unsigned)((m * p.res_ld + n) * 4) : OOB, 0);
                }
                vm += NP_;
            }
            if (has_scale) {  // [2 images][GW columns] floats: GW / 4 lanes per image -- the image of the tile's first row, then the next one
                constexpr int LPI = GW / 4;
                const int b0 = fast_div(min(m0, p.M - 1), a.hw, rcp_hw);
                const int b = min(b0 + lane / LPI, p.N - 1), n = n0 + 4 * (lane % LPI);
                dma16(rsrcS, dst + rtile, (lane < 2 * LPI && n < p.Cout) ? (unsigned)((b * p.Cout + n) * 4) : OOB, 0);
                ++vm;
            }
            seqr = lane == j ? vm : seqr;
        };
        // (activations and skip connections interleaved in tile order: the in-order queue then returns tile k's pieces before tile k + 1's)
        for (int k = 0; k < a.NL; ++k) { a_issue(k); if (k < a.rq) r_issue(k); }
        for (int k = 0; k < nit; ++k) {
            int need = __builtin_amdgcn_readlane(seqa, k % a.NL);
            if (has_res && li == 0) need = max(need, __builtin_amdgcn_readlane(seqr, k % a.rq));
            s1_vm_wait(vm - need);
            asm volatile("s_barrier" ::: "memory");       // tile k has landed; every compute wave is done with tile k - 1
            if (k > 0) { a_issue(k - 1 + a.NL); r_issue(k - 1 + a.rq); }
            if constexpr (KH == 2) asm volatile("s_barrier" ::: "memory");      // (the compute waves' exchange of partial sums)
        }
        return;
    }

    // ================= the four COMPUTE waves
    const __amdgpu_buffer_rsrc_t rsrcB = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<unsigned short*>(p.wpl), 0, (int)((long long)p.CoutPad * p.Kpad * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrcO =
        __builtin_amdgcn_make_buffer_rsrc(p.out, 0, (int)min((long long)p.M * p.out_ld * 4, (long long)OOB), 0x00020000);
    const PlaneDesc pd = make_plane_desc(p);
    float* const stg = reinterpret_cast<float*>(lds + a.off_stg) + w * S1_STG_FLOATS;
    const int kh = KH == 2 ? (w >> 1) : 0;            // K half of the wave
    const int wq = KH == 2 ? (w & 1) : w;             // its 32 columns inside the group
    // fragment geometry: lane -> row (lane & 31), logical granule 2 ks + (lane >> 5) at slot granule ^ ((row >> 2) & 3)
    const int fsw = ((lane & 31) >> 2) & 3;
    const int fr0 = (lane & 31) * 64 + ((((lane >> 5)) ^ fsw) << 4), fr1 = fr0 ^ 32;
    // ---- the wave's filter fragments: its 32 columns x K, read ONCE from the packed image ([CoutPad / 64][chunk][64 rows][64 B])
    f16x8 fb[NCW][2];
    {
        const int col0 = n0 + 32 * wq;
        const int t64 = col0 >> 6;
        const int row_off = ((col0 >> 5) & 1) * 2048;  // the wave's 32 rows inside the 64-row tile (the swizzle key (row >> 2) & 3 is unchanged)
        // every block of a column group reads the SAME fragments, at the same moment (kernel start): the chunks are requested in one of
        // four rotated orders by block, so that the blocks of an XCD do not all queue on one L2 line at a time
        auto load_from = [&](auto startc) __attribute__((always_inline)) {
            constexpr int start = decltype(startc)::value;
            static_for<NCW>([&](auto ic) __attribute__((always_inline)) {
                constexpr int c = (decltype(ic)::value + start) % NCW;
                const int so = (t64 * NCH + kh * NCW + c) * 4096 + row_off;
                fb[c][0] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(rsrcB, fr0, so, 0));
                fb[c][1] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(rsrcB, fr1, so, 0));
            });
        };
        switch (NCW >= 4 ? (ms & 3) : 0) {
            case 1: load_from(std::integral_constant<int, NCW / 4>{}); break;
            case 2: load_from(std::integral_constant<int, NCW / 2>{}); break;
            case 3: load_from(std::integral_constant<int, (3 * NCW) / 4>{}); break;
            default: load_from(std::integral_constant<int, 0>{}); break;
        }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();
    // epilogue geometry of the lane: staging row (lane >> 3) + 8 pass, columns 4 (lane & 7) ..+3 of the wave's 32
    const int e_row = lane >> 3, e_q = lane & 7;
    const int nl = 32 * wq + 4 * e_q, n = n0 + nl;     // the lane's first column inside the group / the layer
    const f32x4 bias4 = *reinterpret_cast<const f32x4*>(ldsBias + nl);
    const bool n_ok = n < p.Cout;

    for (int k = 0; k < nit; ++k) {
        const int mt = (ms + k * a.MS) * 8 + xcd;
        const int m0 = mt * S1_BM;
        asm volatile("s_barrier" ::: "memory");       // tile k (activations, skip connection) is in LDS
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        {
            const char* const ab = lds + (k % a.NL) * (NCH * 2048) + kh * (NCW * 2048);
            static_for<NCW>([&](auto cc) __attribute__((always_inline)) {
                constexpr int c = decltype(cc)::value;
                const f16x8 a0 = *reinterpret_cast<const f16x8*>(ab + c * 2048 + fr0);
                const f16x8 a1 = *reinterpret_cast<const f16x8*>(ab + c * 2048 + fr1);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, fb[c][0], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, fb[c][1], acc, 0, 0, 0);
            });
        }
        // epilogue, wave-private: the finished sums through the wave's staging half-tile so that a lane owns 4 consecutive channels of a row
        // (C/D layout of the 32x32 MFMA: column = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5): registers 0-7 hold rows 0-15,
        // registers 8-15 rows 16-31)
        {
            const char* const rb = lds + oR + (k % a.rq) * a.rbuf;
            const int b0 = has_scale ? fast_div(min(m0, p.M - 1), a.hw, rcp_hw) : 0;
            float own[8];     // KH == 2: the sums of the 16 rows this wave finishes
            if constexpr (KH == 2) {
                // the wave of the low K half finishes rows 0-15 (registers 0-7), the other one rows 16-31: each parks the rows it gives away
                // in the PARTNER's staging tile, and behind a barrier adds what the partner parked in its own
                float* const pstg = reinterpret_cast<float*>(lds + a.off_stg) + (w ^ 2) * S1_STG_FLOATS;
#pragma unroll
                for (int r = 0; r < 8; ++r) {
                    const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                    own[r] = kh ? acc[8 + r] : acc[r];
                    pstg[row * 36 + (lane & 31)] = kh ? acc[r] : acc[8 + r];
                }
                asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#pragma unroll
                for (int r = 0; r < 8; ++r) {
                    const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                    own[r] += stg[row * 36 + (lane & 31)];
                }
            }
#pragma unroll
            for (int half = 0; half < (KH == 2 ? 1 : 2); ++half) {
#pragma unroll
                for (int r = 0; r < 8; ++r) {
                    const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);     // row inside the half
                    if constexpr (KH == 2) stg[row * 36 + (lane & 31)] = own[r];
                    else stg[row * 36 + (lane & 31)] = acc[8 * half + r];
                }
#pragma unroll
                for (int ps = 0; ps < 2; ++ps) {
                    const int hrow = 8 * ps + e_row, row = 16 * (KH == 2 ? kh : half) + hrow;
                    const int m = m0 + row;
                    f32x4 v = *reinterpret_cast<const f32x4*>(stg + hrow * 36 + 4 * e_q);
                    v += bias4;
                    f32x4 r4 = {0.f, 0.f, 0.f, 0.f};
                    if (has_res) {
                        if (r16) r4 = __builtin_convertvector(*reinterpret_cast<const f16x4*>(rb + row * rrow + nl * 2), f32x4);
                        else r4 = *reinterpret_cast<const f32x4*>(rb + row * rrow + nl * 4);
                        if (has_scale) {
                            const int b = fast_div(min(m, p.M - 1), a.hw, rcp_hw);
                            r4 *= *reinterpret_cast<const f32x4*>(rb + rtile + (b - b0) * (GW * 4) + nl * 4);
                        }
                        if (!p.res_after_act) v += r4;
                    }
                    if (p.act == ACT_LEAKY) {
                        v.x = v.x > 0.f ? v.x : 0.1f * v.x; v.y = v.y > 0.f ? v.y : 0.1f * v.y;
                        v.z = v.z > 0.f ? v.z : 0.1f * v.z; v.w = v.w > 0.f ? v.w : 0.1f * v.w;
                    } else if (p.act == ACT_RELU) {
                        v.x = v.x > 0.f ? v.x : 0.f; v.y = v.y > 0.f ? v.y : 0.f;
                        v.z = v.z > 0.f ? v.z : 0.f; v.w = v.w > 0.f ? v.w : 0.f;
                    }
                    if (has_res && p.res_after_act) v += r4;
                    // (rows past M and columns past Cout: out-of-range offsets, dropped by the hardware)
                    const unsigned off = (m < p.M && n_ok) ? (unsigned)(m * p.out_ld + n) : (OOB >> 2);
                    if (pd.f32) {
                        const u32x4 o = {__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w)};
                        __builtin_amdgcn_raw_buffer_store_b128(o, rsrcO, (int)(off * 4), 0, 0);
                    }
                    if (pd.np == 1) {
                        const f16x4 h = __builtin_convertvector(v, f16x4);
                        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, h), pd.r0, (int)(off * 2), 0, 0);
                    }
                }
            }
        }
    }
}

// ---- host side
// the launch shape for this layer: false when not even one activation tile fits the LDS that is left
static bool s1_plan(const ConvParams& p, long long M, int res_bytes, S1Args* out, int* grid, int* lds_bytes) {
    S1Args a{};
    const int nch = p.nchunks;
    a.MT = (int)((M + S1_BM - 1) / S1_BM);
    const int gw = nch == 32 ? S1_BN / 2 : S1_BN;      // K = 1 024: 64 columns per block (two K halves per column half)
    a.NG = (p.CoutPad + gw - 1) / gw;
    a.res_bytes = res_bytes;
    a.hw = p.OH * p.OW;
    a.rbuf = res_bytes ? 32 * gw * res_bytes + (p.res_scale ? 1024 : 0) : 0;
    const int bias_b = S1_BN * 4, stg_b = 4 * S1_STG_FLOATS * 4, tile_b = nch * 2048;
    // blocks per CU: the filter fragments take K / 4 registers per lane -- two blocks (ten waves) per CU up to K = 256, one beyond
    static const int force_bpc = std::getenv("BP_S1_BPC") ? std::atoi(std::getenv("BP_S1_BPC")) : 0;     // (A/B runs)
    static const int force_nl = std::getenv("BP_S1_NL") ? std::atoi(std::getenv("BP_S1_NL")) : 0;
    const int bpc = force_bpc ? force_bpc : (nch <= 8 ? 2 : 1);
    const int budget = (160 * 1024 - 192) / bpc;
    a.rq = res_bytes ? 2 : 0;
    const int nl = (budget - stg_b - bias_b - a.rq * a.rbuf) / tile_b;
    // the hand-over needs two buffers of each kind: tile k + 1 (and its skip-connection rows) land while tile k is read.  A single
    // skip-connection buffer would be refilled behind barrier B_k while the compute waves still read it in tile k's epilogue (the
    // round-5 advisor's finding on the old rq == 1 fall-back, reachable through BP_S1_BPC): such a layer stays on the plane tile
    if (nl < 2) return false;
    // look-ahead: TWO tiles.  More is slower (measured at batch 28, fp16 skip connections: 52x52 256 -> 128 17.2 us with two tiles, 18.0 with
    // three, 19.6 with four; 256 -> 1024 15.5 / 16.4 / 16.1): a block has 4-9 tiles in all, so deep look-ahead is every block asking for most of
    // its input at once -- the first tiles queue behind everybody's later ones
    a.NL = std::min(nl, force_nl >= 2 ? force_nl : 2);
    const int MT8 = (a.MT + 7) / 8;
    a.MS = std::max(1, std::min(MT8, (bpc * 256) / (8 * a.NG)));
    a.off_stg = a.NL * tile_b;
    a.off_res = a.off_stg + stg_b;
    a.off_bias = a.off_res + a.rq * a.rbuf;
    *out = a;
    *grid = 8 * a.NG * a.MS;
    *lds_bytes = a.off_bias + bias_b;
    return true;
}

bool conv_s1_eligible(const ConvParams& p, long long M) {
    if (!(conv_pl_eligible(p) && p.wpl != nullptr && p.mfma_mode == PREC_F16)) return false;
    if (!(p.ksize == 1 && p.pad == 0 && (p.stride == 1 || p.stride == 2) && p.Kpad == p.Cin && p.store_mode == ST_NHWC)) return false;
    const int nch = p.nchunks;
    if (!(nch == 2 || nch == 4 || nch == 8 || nch == 12 || nch == 16 || nch == 32)) return false;      // K = 64, 128, 256, 384, 512, 1 024 (the instantiated forms)
    if (M < 2048 || p.pool_out != nullptr) return false;       // (the cheap rejections first: choose_launch asks for every fp16 convolution of every eager pass)
    // K = 1 024 (64 columns per block, two K halves per column half), measured at batch 28 against the 64x64 plane tile on one box: 1 024 -> 256
    // ties (16.0 against 16.2 us), 13x13 1 024 -> 512 loses (17.4 against 16.3), 1 024 -> 2 048 / stride 2 wins (30.2 against 36.5): the wide
    // layers only.  (K = 512: 512 -> 128 16.7 against 20.8 us, 26x26 512 -> 256 17.2 against 18.8 -- all of them.)  BP_S1_K512=1 takes every
    // K = 1 024 layer (tests, A/B runs; read per call: the tests switch it inside one process)
    const bool k1024_all = std::getenv("BP_S1_K512") != nullptr;
    if (nch == 32 && p.CoutPad < 1024 && !k1024_all) return false;
    if (p.CoutPad < S1_BN || (p.Cout & 3) || (p.out_ld & 3) || p.OH * p.OW < S1_BM) return false;
    if (p.res && ((p.res_ld & 7) != 0)) return false;
    if (p.res_scale && !p.res) return false;
    if (p.out16 != nullptr && p.out_np != 1) return false;
    if (M * p.out_ld * 4 >= (long long)OOB || (p.res && M * p.res_ld * 4 >= (long long)OOB)) return false;
    S1Args a; int g, l;
    if (!s1_plan(p, M, p.res ? 4 : 0, &a, &g, &l)) return false;     // (sized for the larger skip-connection format: the plan holds whichever the launch gets)
    // a persistent block pays its start-up (filter fragments, first tile) over the tiles it walks: with fewer than three tiles per block the
    // 64x64 plane tile wins -- configs-style runs of 2 / 4 / 8 frames per launch x 3 streams lost 0.7 / 1.3 / 2.0 % with this kernel on every
    // M >= 2 048 layer (28 frames per launch: 4.4-17 tiles per block, +6.5 %).  BP_S1_K512=1 (tests) takes such layers too.
    return k1024_all || (a.MT + 7) / 8 >= 3 * a.MS;
}

template <int NCH, int KH = 1>
static void launch_s1_t(const ConvParams& p, const S1Args& a, int grid, int lds_bytes, hipStream_t s) {
    if (lds_bytes > 64 * 1024) allow_big_lds(reinterpret_cast<const void*>(conv_s1_kernel<NCH, KH>));
    if (g_conv_prof)
        hipExtLaunchKernelGGL((conv_s1_kernel<NCH, KH>), dim3(grid), dim3(NCH >= 12 ? 384 : 320), lds_bytes, s, g_conv_prof->e0, g_conv_prof->e1, 0, p, a);
    else
        hipLaunchKernelGGL((conv_s1_kernel<NCH, KH>), dim3(grid), dim3(NCH >= 12 ? 384 : 320), lds_bytes, s, p, a);
}

void launch_conv_s1(const ConvParams& p, hipStream_t s) {
    BP_CHECK(conv_s1_eligible(p, p.M) && p.splits == 1 && p.hy_splits == 0 && !p.xcd_home,
             "streaming 1x1 tile: fp16 mode, 1x1 / stride 1 or 2, NHWC store, K in {64, 128, 256, 384, 512, 1024}, N >= 128, M >= 2048, one K slice");
    BP_CHECK((long long)p.N * p.H * p.W * p.in_ld * 2 < (long long)OOB, "activation planes too large for 32-bit offsets");
    S1Args a; int grid = 0, lds_bytes = 0;
    BP_CHECK(s1_plan(p, p.M, p.res ? (p.res16 ? 2 : 4) : 0, &a, &grid, &lds_bytes), "streaming 1x1 tile: no LDS plan");
    switch (p.nchunks) {
        case 2: launch_s1_t<2>(p, a, grid, lds_bytes, s); break;
        case 4: launch_s1_t<4>(p, a, grid, lds_bytes, s); break;
        case 8: launch_s1_t<8>(p, a, grid, lds_bytes, s); break;
        case 12: launch_s1_t<12>(p, a, grid, lds_bytes, s); break;
        case 32: launch_s1_t<32, 2>(p, a, grid, lds_bytes, s); break;
        default: launch_s1_t<16>(p, a, grid, lds_bytes, s); break;
    }
}

}  // namespace bp
