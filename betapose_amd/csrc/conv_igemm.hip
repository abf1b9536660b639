// Fused convolution for gfx950: implicit-GEMM (no im2col buffer) on the fp32 MFMA
// pipe (v_mfma_f32_32x32x2_f32: exact fp32 FMA chain, 256 FLOP/clk/CU), NHWC
// activations, BN-folded filters packed [Cout][ky][kx][ci], epilogue fusing bias,
// LeakyReLU/ReLU, residual add (YOLO shortcut / ResNet bottleneck), SE channel
// scale, nearest-x2 upsample, PixelShuffle(2) and NCHW stores.
//
// Replaces, for the hot path, cudnnConvolutionForward + normalize/scale_bias/
// add_bias/activate/shortcut/upsample kernels of the reference's Darknet CUDA
// backend (train_YOLO/src/convolutional_kernels.cu:121-383, blas_kernels.cu) and
// the torch.nn Conv2d/BatchNorm2d/LeakyReLU/ReLU/PixelShuffle modules of
// yolo/darknet.py:240-259 and KPD/src/models/layers/{SE_Resnet,DUC}.py.
//
// Block = 256 threads = 4 waves (one per SIMD), 2x2 over a (64*TM)x(64*TN) output
// tile; K is walked in chunks of 32 (one filter tap slice: ci0..ci0+31 are
// contiguous in NHWC), global -> registers -> LDS double buffer, one barrier per
// chunk.  LDS rows are padded to 36 floats so the ds_read_b128 fragment reads are
// bank-conflict free (row stride 144 B covers all 64 banks over 16 rows).
// Fragment trick: lane l reads A[row=l&31][4h..4h+3] (h=l>>5) as one b128 and feeds
// component j to MFMA j, so MFMA j contracts k = {j, 4+j} of the 8-wide sub-chunk;
// B uses the same k mapping, and the sum over k is order-free.
#include <hip/hip_ext.h>

#include "bp_common.h"

namespace bp {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));   // native vector: stays in VGPRs (HIP's float4 struct
                                                           // made hipcc park the prefetch registers in scratch)

static constexpr int BK = 32;
static constexpr int LDS_LD = 36;

__device__ __forceinline__ float apply_act(float v, int act) {
    if (act == ACT_LEAKY) return v > 0.f ? v : 0.1f * v;
    if (act == ACT_RELU) return v > 0.f ? v : 0.f;
    return v;
}

// v = raw accumulator for output element (m, n)
__device__ __forceinline__ void epilogue_store(const ConvParams& p, int m, int n, float v) {
    v += p.bias[n];
    int b = 0, pix = m;
    const int hw = p.OH * p.OW;
    const bool need_pix = p.store_mode != ST_NHWC || p.res_scale != nullptr;
    if (need_pix) {
        b = m / hw;
        pix = m - b * hw;
    }
    float r = 0.f;
    if (p.res) {
        r = p.res[(long long)m * p.res_ld + n];
        if (p.res_scale) r *= p.res_scale[b * p.Cout + n];
    }
    if (!p.res_after_act) v += r;
    v = apply_act(v, p.act);
    if (p.res_after_act) v += r;
    switch (p.store_mode) {
        case ST_NHWC:
            p.out[(long long)m * p.out_ld + n] = v;
            break;
        case ST_UP2: {
            const int oy = pix / p.OW, ox = pix - oy * p.OW;
            const int W2 = 2 * p.OW;
            float* o = p.out + ((long long)(b * 2 * p.OH + 2 * oy) * W2 + 2 * ox) * p.out_ld + n;
            o[0] = v;
            o[p.out_ld] = v;
            o[(long long)W2 * p.out_ld] = v;
            o[(long long)(W2 + 1) * p.out_ld] = v;
        } break;
        case ST_PIXSHUF: {
            const int oy = pix / p.OW, ox = pix - oy * p.OW;
            const int cq = p.Cout >> 2;
            const int ij = n / cq, c = n - ij * cq;
            const int y = 2 * oy + (ij >> 1), x = 2 * ox + (ij & 1);
            p.out[((long long)(b * 2 * p.OH + y) * (2 * p.OW) + x) * p.out_ld + c] = v;
        } break;
        case ST_NCHW:
            p.out[((long long)b * p.Cout + n) * hw + pix] = v;
            break;
    }
}

// one float4 (4 consecutive k) of im2col row for chunk c; zero outside the image / past Ktrue
template <bool VEC>
__device__ __forceinline__ f32x4 load_a_row(const ConvParams& p, int c, int cpt, int c4, int bh, int iy0, int ix0,
                                             bool& ok_out) {
    f32x4 out = {0.f, 0.f, 0.f, 0.f};
    ok_out = true;
    if constexpr (VEC) {
        const int tap = c / cpt;
        const int ci0 = (c - tap * cpt) << 5;
        const int ky = tap / p.ksize;
        const int kx = tap - ky * p.ksize;
        const int iy = iy0 + ky, ix = ix0 + kx;
        // branch-free: out-of-image taps read a valid address and are zeroed by a select at LDS-store
        // time, so the prefetch block is one basic block and nothing consumes the loads before the MFMAs
        const bool ok = (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
        const long long off = ok ? ((long long)(bh + iy) * p.W + ix) * p.in_ld + ci0 + c4 * 4 : 0;
        out = *reinterpret_cast<const f32x4*>(p.in + off);   // zeroed (if !ok) when parked in LDS, after the MFMAs
        ok_out = ok;
    } else {
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int k = c * BK + c4 * 4 + e;
            float x = 0.f;
            if (k < p.Ktrue) {
                const int tap = k / p.Cin;
                const int ci = k - tap * p.Cin;
                const int ky = tap / p.ksize;
                const int kx = tap - ky * p.ksize;
                const int iy = iy0 + ky, ix = ix0 + kx;
                if ((unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W)
                    x = p.in[((long long)(bh + iy) * p.W + ix) * p.in_ld + ci];
            }
            v[e] = x;
        }
        out = f32x4{v[0], v[1], v[2], v[3]};
    }
    return out;
}

template <int TM, int TN, bool VEC>
__global__ __launch_bounds__(256) void conv_igemm_kernel(const ConvParams p) {
    constexpr int BM = 64 * TM, BN = 64 * TN;
    constexpr int RA = BM / 32, RB = BN / 32;
    __shared__ __attribute__((aligned(16))) float As[2][BM * LDS_LD];
    __shared__ __attribute__((aligned(16))) float Bs[2][BN * LDS_LD];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int n_tiles_n = p.CoutPad / BN;
    // 1-D grid, K-slice fastest: consecutive block ids (= consecutive XCDs) take different K-slices of the same
    // output tile, so with 8 slices every XCD streams its own 1/8 of the filters through its private L2
    const int split = (int)blockIdx.x % p.splits;
    const int tile_id = (int)blockIdx.x / p.splits;
    const int tile_n = tile_id % n_tiles_n;
    const int tile_m = tile_id / n_tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int c_begin = split * p.chunks_per_split;
    const int c_end = min(p.nchunks, c_begin + p.chunks_per_split);

    const int lr = tid >> 3, c4 = tid & 7;
    const int hw = p.OH * p.OW;

    int a_bh[RA], a_iy0[RA], a_ix0[RA];
#pragma unroll
    for (int i = 0; i < RA; ++i) {
        const int m = m0 + lr + 32 * i;
        const bool ok = m < p.M;
        const int mm = ok ? m : 0;
        const int b = mm / hw;
        const int rem = mm - b * hw;
        const int oy = rem / p.OW;
        const int ox = rem - oy * p.OW;
        a_bh[i] = b * p.H;
        a_iy0[i] = ok ? oy * p.stride - p.pad : -(1 << 20);
        a_ix0[i] = ox * p.stride - p.pad;
    }

    f32x4 ra[RA], rb[RB];
    bool ra_ok[RA];
    const int cpt = VEC ? (p.Cin >> 5) : 1;  // chunks per filter tap

#define BP_LOAD_CHUNK(c_)                                                                          \
    {                                                                                              \
        const int cc = (c_);                                                                       \
        _Pragma("unroll") for (int i = 0; i < RA; ++i)                                             \
            ra[i] = load_a_row<VEC>(p, cc, cpt, c4, a_bh[i], a_iy0[i], a_ix0[i], ra_ok[i]);        \
        _Pragma("unroll") for (int i = 0; i < RB; ++i)                                             \
            rb[i] = *reinterpret_cast<const f32x4*>(p.w + (long long)(n0 + lr + 32 * i) * p.Kpad + \
                                                     cc * BK + c4 * 4);                            \
    }
#define BP_STORE_LDS(buf_)                                                                         \
    {                                                                                              \
        _Pragma("unroll") for (int i = 0; i < RA; ++i) {                                           \
            f32x4 v_ = ra[i];                                                                     \
            if (!ra_ok[i]) v_ = f32x4{0.f, 0.f, 0.f, 0.f};                                       \
            *reinterpret_cast<f32x4*>(&As[buf_][(lr + 32 * i) * LDS_LD + c4 * 4]) = v_;            \
        }                                                                                          \
        _Pragma("unroll") for (int i = 0; i < RB; ++i)                                             \
            *reinterpret_cast<f32x4*>(&Bs[buf_][(lr + 32 * i) * LDS_LD + c4 * 4]) = rb[i];         \
    }

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int frag_off = (lane & 31) * LDS_LD + (lane >> 5) * 4;
    const int a_off = wm * (BM / 2) * LDS_LD + frag_off;
    const int b_off = wn * (BN / 2) * LDS_LD + frag_off;

    if (c_begin < c_end) {
        BP_LOAD_CHUNK(c_begin);
        BP_STORE_LDS(0);
    }
    __syncthreads();
#define BP_COMPUTE(buf_)                                                                              \
    {                                                                                                 \
        const float* Ab = &As[buf_][a_off];                                                           \
        const float* Bb = &Bs[buf_][b_off];                                                           \
        _Pragma("unroll") for (int ks = 0; ks < 4; ++ks) {                                            \
            f32x4 a[TM], b[TN];                                                                       \
            _Pragma("unroll") for (int i = 0; i < TM; ++i) a[i] =                                     \
                *reinterpret_cast<const f32x4*>(Ab + i * 32 * LDS_LD + ks * 8);                       \
            _Pragma("unroll") for (int j = 0; j < TN; ++j) b[j] =                                     \
                *reinterpret_cast<const f32x4*>(Bb + j * 32 * LDS_LD + ks * 8);                       \
            _Pragma("unroll") for (int i = 0; i < TM; ++i) _Pragma("unroll") for (int j = 0; j < TN; ++j) { \
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].x, b[j].x, acc[i][j], 0, 0, 0); \
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].y, b[j].y, acc[i][j], 0, 0, 0); \
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].z, b[j].z, acc[i][j], 0, 0, 0); \
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].w, b[j].w, acc[i][j], 0, 0, 0); \
            }                                                                                         \
        }                                                                                             \
    }
    // steady state: prefetch chunk c+1 into registers, contract chunk c from LDS, park c+1 in the
    // other LDS buffer, one barrier.  The last chunk is peeled so the loop body has no conditionals
    // (a conditional prefetch makes hipcc park the prefetch registers in scratch behind a vmcnt(0)).
    int buf = 0;
    for (int c = c_begin; c + 1 < c_end; ++c) {
        BP_LOAD_CHUNK(c + 1);
        __builtin_amdgcn_sched_barrier(0);   // all global loads of chunk c+1 are issued before the MFMAs of chunk c
        BP_COMPUTE(buf);
        __builtin_amdgcn_sched_barrier(0);
        BP_STORE_LDS(buf ^ 1);
        __syncthreads();
        buf ^= 1;
    }
    if (c_begin < c_end) BP_COMPUTE(buf);

    // epilogue: C/D layout of 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
    if (p.splits == 1) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int n = n0 + wn * (BN / 2) + j * 32 + (lane & 31);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = m0 + wm * (BM / 2) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                    if (m < p.M && n < p.Cout) epilogue_store(p, m, n, acc[i][j][r]);
                }
            }
        return;
    }
    // ---- split-K: every slice parks its fp32 accumulators in a slab laid out in FRAGMENT order
    // (slice, tile, wave, quad, lane) so each lane moves 16 B per instruction, fully coalesced; the LAST slice
    // to arrive at the tile's ticket counter sums all slabs in slice order (deterministic, independent of arrival
    // order) and runs the epilogue.  Hand-off without fences (cdna_hip_programming.md G16 "R1"): write-through
    // (sc1) slab stores, every storing wave drains vmcnt, one relaxed agent-scope ticket, sc1 loads by the reducer.
    const int tiles = (int)gridDim.x / p.splits;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
        p.partial, 0, (int)((long long)p.splits * tiles * (TM * TN * 4096) * 4), 0x00020000);
    constexpr int TILE_FLOATS = TM * TN * 4096;
    const int my_off = ((split * tiles + tile_id) * TILE_FLOATS + wave * (TM * TN * 1024) + lane * 4) * 4;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                u32x4 v;
                v.x = __float_as_uint(acc[i][j][4 * q]);
                v.y = __float_as_uint(acc[i][j][4 * q + 1]);
                v.z = __float_as_uint(acc[i][j][4 * q + 2]);
                v.w = __float_as_uint(acc[i][j][4 * q + 3]);
                __builtin_amdgcn_raw_buffer_store_b128(v, rsrc, my_off + (((i * TN + j) * 4 + q) * 256) * 4, 0, 16 /*sc1*/);
            }
    __shared__ int s_last;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
        const int ticket = __hip_atomic_fetch_add(&p.tickets[tile_id], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_last = ticket == p.splits - 1;
        if (s_last) __hip_atomic_store(&p.tickets[tile_id], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // re-arm
    }
    __syncthreads();
    if (!s_last) return;
    const int base_off = ((tile_id) * TILE_FLOATS + wave * (TM * TN * 1024) + lane * 4) * 4;
    const int slice_stride = tiles * TILE_FLOATS * 4;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = n0 + wn * (BN / 2) + j * 32 + (lane & 31);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float v0 = 0.f, v1 = 0.f, v2 = 0.f, v3 = 0.f;
                const int off = base_off + (((i * TN + j) * 4 + q) * 256) * 4;
                for (int sidx = 0; sidx < p.splits; ++sidx) {
                    const u32x4 t = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off + sidx * slice_stride, 0, 16 /*sc1*/);
                    v0 += __uint_as_float(t.x); v1 += __uint_as_float(t.y);
                    v2 += __uint_as_float(t.z); v3 += __uint_as_float(t.w);
                }
                const float vv[4] = {v0, v1, v2, v3};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int m = m0 + wm * (BM / 2) + i * 32 + e + 8 * q + 4 * (lane >> 5);
                    if (m < p.M && n < p.Cout) epilogue_store(p, m, n, vv[e]);
                }
            }
        }
}

thread_local ConvProfHook* g_conv_prof = nullptr;

int conv_tile_bm(int tile) { return tile == TILE_128x64 ? 128 : 64; }
int conv_tile_bn(int tile) { return 64; }

template <int TM, int TN>
static void launch_t(const ConvParams& p, hipStream_t s) {
    constexpr int BM = 64 * TM, BN = 64 * TN;
    dim3 grid(((p.M + BM - 1) / BM) * (p.CoutPad / BN) * p.splits);
    const bool vec = (p.Cin % 32 == 0) && (p.in_ld % 4 == 0);
    hipEvent_t e0 = g_conv_prof ? g_conv_prof->e0 : nullptr, e1 = g_conv_prof ? g_conv_prof->e1 : nullptr;
    // hipExtLaunchKernelGGL stamps e0/e1 with the kernel's own begin/end (no host-side event gap)
    if (vec)
        hipExtLaunchKernelGGL((conv_igemm_kernel<TM, TN, true>), grid, dim3(256), 0, s, e0, e1, 0, p);
    else
        hipExtLaunchKernelGGL((conv_igemm_kernel<TM, TN, false>), grid, dim3(256), 0, s, e0, e1, 0, p);
}

int conv_tiles(const ConvParams& p, int tile) {
    const int bm = conv_tile_bm(tile);
    return ((p.M + bm - 1) / bm) * (p.CoutPad / 64);
}

void launch_conv(const ConvParams& p, int tile, hipStream_t s) {
    BP_CHECK(p.CoutPad % 64 == 0, "CoutPad must be a multiple of 64");
    BP_CHECK(p.Kpad % BK == 0 && p.nchunks == p.Kpad / BK, "Kpad");
    BP_CHECK(p.splits >= 1 && (p.splits == 1 || (p.partial != nullptr && p.tickets != nullptr)), "split-K workspace");
    switch (tile) {
        case TILE_128x64: launch_t<2, 1>(p, s); break;
        default: launch_t<1, 1>(p, s); break;
    }
    BP_HIP(hipGetLastError());
}

}  // namespace bp
