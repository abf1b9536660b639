// Small HBM-bound kernels around the convolutions (gfx950): layout changes,
// max-pool, residual add, nearest upsample, PixelShuffle, SE squeeze/excite,
// YOLO head decode + objectness arg-max, heat-map arg-max, device-side crop and
// Pillow-exact bicubic resize.  64-wide wavefront reductions via __shfl_down.
#include "bp_common.h"

namespace bp {

// ---- operand planes (bp_common.h ConvParams::out16) from the small producers: one fp16 plane or three bf16 planes that
// sum to the fp32 value exactly, at the same element index as the fp32 store
__device__ __forceinline__ void store_plane(unsigned short* planes, long long idx, long long plane_elems, int np, float v) {
    if (np == 1) {
        const _Float16 h = (_Float16)v;
        planes[idx] = __builtin_bit_cast(unsigned short, h);
    } else if (np == 3) {
        const __bf16 h1 = (__bf16)v;
        const float r1 = v - (float)h1;
        const __bf16 h2 = (__bf16)r1;
        const __bf16 h3 = (__bf16)(r1 - (float)h2);
        planes[idx] = __builtin_bit_cast(unsigned short, h1);
        planes[idx + plane_elems] = __builtin_bit_cast(unsigned short, h2);
        planes[idx + 2 * plane_elems] = __builtin_bit_cast(unsigned short, h3);
    }
}

static inline int grid_for(long long n, int block = 256, int cap = 4096) {
    BP_CHECK(n < (1ll << 31), "tensor too large for 32-bit indexing");
    long long g = (n + block - 1) / block;
    return (int)(g < 1 ? 1 : (g > cap ? cap : g));
}

// ---------------------------------------------------------------- layout
__global__ void nchw_to_nhwc_kernel(const float* __restrict__ in, float* __restrict__ out, int N, int C, int HW) {
    const long long total = (long long)N * C * HW;
    // 32-bit index math: every tensor on this path has < 2^31 elements (checked by the launchers), and 64-bit
    // div/mod costs ~100 instructions on gfx950
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < (int)total; e += gridDim.x * blockDim.x) {
        const int c = (int)(e % C);
        const int t = e / C;
        const int p = t % HW;
        const int n = t / HW;
        out[e] = in[((long long)n * C + c) * HW + p];
    }
}
void launch_nchw_to_nhwc(const float* in, float* out, int N, int C, int H, int W, hipStream_t s) {
    const long long total = (long long)N * C * H * W;
    hipLaunchKernelGGL(nchw_to_nhwc_kernel, dim3(grid_for(total)), dim3(256), 0, s, in, out, N, C, H * W);
}

__global__ void nhwc_to_nchw_kernel(const float* __restrict__ in, int in_ld, float* __restrict__ out, int N, int C, int HW) {
    const long long total = (long long)N * C * HW;
    // 32-bit index math: every tensor on this path has < 2^31 elements (checked by the launchers), and 64-bit
    // div/mod costs ~100 instructions on gfx950
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < (int)total; e += gridDim.x * blockDim.x) {
        const int p = (int)(e % HW);
        const int t = e / HW;
        const int c = t % C;
        const int n = t / C;
        out[e] = in[((long long)n * HW + p) * in_ld + c];
    }
}
void launch_nhwc_to_nchw(const float* in, int in_ld, float* out, int N, int C, int H, int W, hipStream_t s) {
    const long long total = (long long)N * C * H * W;
    hipLaunchKernelGGL(nhwc_to_nchw_kernel, dim3(grid_for(total)), dim3(256), 0, s, in, in_ld, out, N, C, H * W);
}

// ---------------------------------------------------------------- max-pool 3x3 / stride 2 / pad 1 (NHWC)
__global__ void maxpool3s2p1_kernel(const float* __restrict__ in, float* __restrict__ out, int N, int H, int W, int C,
                                    int OH, int OW, unsigned short* __restrict__ planes, long long plane_elems, int np) {
    const int C4 = C >> 2;
    const long long total = (long long)N * OH * OW * C4;
    // 32-bit index math: every tensor on this path has < 2^31 elements (checked by the launchers), and 64-bit
    // div/mod costs ~100 instructions on gfx950
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < (int)total; e += gridDim.x * blockDim.x) {
        const int c4 = (int)(e % C4);
        int t = e / C4;
        const int ox = (int)(t % OW);
        t /= OW;
        const int oy = (int)(t % OH);
        const int n = (int)(t / OH);
        float4 m = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
        for (int dy = 0; dy < 3; ++dy) {
            const int iy = oy * 2 - 1 + dy;
            if ((unsigned)iy >= (unsigned)H) continue;
            for (int dx = 0; dx < 3; ++dx) {
                const int ix = ox * 2 - 1 + dx;
                if ((unsigned)ix >= (unsigned)W) continue;
                const float4 v = *reinterpret_cast<const float4*>(in + (((long long)n * H + iy) * W + ix) * C + c4 * 4);
                m.x = fmaxf(m.x, v.x); m.y = fmaxf(m.y, v.y); m.z = fmaxf(m.z, v.z); m.w = fmaxf(m.w, v.w);
            }
        }
        const long long o = (((long long)n * OH + oy) * OW + ox) * C + c4 * 4;
        *reinterpret_cast<float4*>(out + o) = m;
        if (np) {
            store_plane(planes, o, plane_elems, np, m.x); store_plane(planes, o + 1, plane_elems, np, m.y);
            store_plane(planes, o + 2, plane_elems, np, m.z); store_plane(planes, o + 3, plane_elems, np, m.w);
        }
    }
}
void launch_maxpool3s2p1(const float* in, float* out, int N, int H, int W, int C, int OH, int OW, hipStream_t s,
                         unsigned short* planes, long long plane_elems, int np) {
    BP_CHECK(C % 4 == 0, "maxpool: C % 4");
    const long long total = (long long)N * OH * OW * (C / 4);
    hipLaunchKernelGGL(maxpool3s2p1_kernel, dim3(grid_for(total)), dim3(256), 0, s, in, out, N, H, W, C, OH, OW,
                       planes, plane_elems, planes ? np : 0);
}

// ---------------------------------------------------------------- residual add / channel copy / upsample (fallbacks)
__global__ void add_kernel(const float* __restrict__ a, int a_ld, const float* __restrict__ b, int b_ld,
                           float* __restrict__ out, int out_ld, long long pixels, int C) {
    const long long total = pixels * C;
    // 32-bit index math: every tensor on this path has < 2^31 elements (checked by the launchers), and 64-bit
    // div/mod costs ~100 instructions on gfx950
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < (int)total; e += gridDim.x * blockDim.x) {
        const int c = (int)(e % C);
        const int p = e / C;
        out[p * out_ld + c] = a[p * a_ld + c] + b[p * b_ld + c];
    }
}
void launch_add(const float* a, int a_ld, const float* b, int b_ld, float* out, int out_ld, long long pixels, int C,
                hipStream_t s) {
    hipLaunchKernelGGL(add_kernel, dim3(grid_for(pixels * C)), dim3(256), 0, s, a, a_ld, b, b_ld, out, out_ld, pixels, C);
}

__global__ void copy_channels_kernel(const float* __restrict__ in, int in_ld, float* __restrict__ out, int out_ld,
                                     long long pixels, int C) {
    const long long total = pixels * C;
    // 32-bit index math: every tensor on this path has < 2^31 elements (checked by the launchers), and 64-bit
    // div/mod costs ~100 instructions on gfx950
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < (int)total; e += gridDim.x * blockDim.x) {
        const int c = (int)(e % C);
        const int p = e / C;
        out[p * out_ld + c] = in[p * in_ld + c];
    }
}
void launch_copy_channels(const float* in, int in_ld, float* out, int out_ld, long long pixels, int C, hipStream_t s) {
    hipLaunchKernelGGL(copy_channels_kernel, dim3(grid_for(pixels * C)), dim3(256), 0, s, in, in_ld, out, out_ld, pixels, C);
}

__global__ void upsample2_kernel(const float* __restrict__ in, int in_ld, float* __restrict__ out, int out_ld, int N,
                                 int H, int W, int C) {
    const long long total = (long long)N * 2 * H * 2 * W * C;
    // 32-bit index math: every tensor on this path has < 2^31 elements (checked by the launchers), and 64-bit
    // div/mod costs ~100 instructions on gfx950
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < (int)total; e += gridDim.x * blockDim.x) {
        const int c = (int)(e % C);
        int t = e / C;
        const int x = (int)(t % (2 * W));
        t /= 2 * W;
        const int y = (int)(t % (2 * H));
        const int n = (int)(t / (2 * H));
        out[(((long long)n * 2 * H + y) * 2 * W + x) * out_ld + c] = in[(((long long)n * H + (y >> 1)) * W + (x >> 1)) * in_ld + c];
    }
}
void launch_upsample2(const float* in, int in_ld, float* out, int out_ld, int N, int H, int W, int C, hipStream_t s) {
    hipLaunchKernelGGL(upsample2_kernel, dim3(grid_for((long long)N * 4 * H * W * C)), dim3(256), 0, s, in, in_ld, out,
                       out_ld, N, H, W, C);
}

// PixelShuffle(2), NHWC in [N][H][W][C] -> out [N][2H][2W][C/4]; NCHW semantics: out[c, 2h+i, 2w+j] = in[c*4+i*2+j, h, w]
__global__ void pixel_shuffle2_kernel(const float* __restrict__ in, float* __restrict__ out, int N, int H, int W, int C,
                                      unsigned short* __restrict__ planes, long long plane_elems, int np) {
    const int Co = C >> 2;
    const long long total = (long long)N * H * W * C;
    // 32-bit index math: every tensor on this path has < 2^31 elements (checked by the launchers), and 64-bit
    // div/mod costs ~100 instructions on gfx950
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < (int)total; e += gridDim.x * blockDim.x) {
        const int co = (int)(e % Co);
        int t = e / Co;
        const int x = (int)(t % (2 * W));
        t /= 2 * W;
        const int y = (int)(t % (2 * H));
        const int n = (int)(t / (2 * H));
        const int ci = co * 4 + (y & 1) * 2 + (x & 1);
        const float v = in[(((long long)n * H + (y >> 1)) * W + (x >> 1)) * C + ci];
        out[e] = v;
        if (np) store_plane(planes, e, plane_elems, np, v);
    }
}
void launch_pixel_shuffle2(const float* in, float* out, int N, int H, int W, int C, hipStream_t s,
                           unsigned short* planes, long long plane_elems, int np) {
    hipLaunchKernelGGL(pixel_shuffle2_kernel, dim3(grid_for((long long)N * H * W * C)), dim3(256), 0, s, in, out, N, H, W, C,
                       planes, plane_elems, planes ? np : 0);
}

// ---------------------------------------------------------------- SE: global average pool + FC
// grid (ceil(C/64), P, N), block 256 = 4 pixel-groups x 64 channels; lanes run over channels (coalesced
// NHWC).  Block (.,p,.) sums pixel slice p; out[(n*P + p)*C + c] holds the slice SUM (not mean): the first SE
// linear layer adds the P partials and applies 1/HW, so the reduction order is fixed (deterministic).
__global__ void avgpool_partial_kernel(const float* __restrict__ in, int in_ld, float* __restrict__ out, int HW, int C, int P) {
    __shared__ float red[4][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63);
    const int g = threadIdx.x >> 6;
    const int p = blockIdx.y, n = blockIdx.z;
    const int per = (HW + P - 1) / P;
    const int lo = p * per, hi = min(HW, lo + per);
    float s = 0.f;
    if (c < C)
        for (int q = lo + g; q < hi; q += 4) s += in[((long long)n * HW + q) * in_ld + c];
    red[g][threadIdx.x & 63] = s;
    __syncthreads();
    if (g == 0 && c < C) {
        const float t = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
        out[((long long)n * P + p) * C + c] = t;
    }
}
int avgpool_parts(int HW) { return HW >= 2048 ? 64 : (HW >= 512 ? 32 : (HW >= 128 ? 8 : 2)); }
void launch_avgpool(const float* in, int in_ld, float* out, int N, int HW, int C, hipStream_t s) {
    const int P = avgpool_parts(HW);
    hipLaunchKernelGGL(avgpool_partial_kernel, dim3((C + 63) / 64, P, N), dim3(256), 0, s, in, in_ld, out, HW, C, P);
}

// GEMV of the SE blocks: pure HBM-bound weight stream (C = 2048: 16.8 MB per layer).  A block of 4 waves produces
// FC_PER_BLOCK neurons: the input vector (the slice sums of avgpool_partial_kernel, reduced and scaled here) is staged
// once in LDS, then every wave streams its rows with 16-B loads, several in flight per lane.  The per-lane summation
// order (i = lane, lane+64, ...) and the shuffle tree are fixed, so results do not depend on the launch shape.
// in_parts > 1: in is [N][in_parts][Cin] partial sums, scaled by in_scale
static constexpr int FC_PER_BLOCK = 8;
__global__ __launch_bounds__(256) void fc_kernel(const float* __restrict__ in, const float* __restrict__ w,
                                                  const float* __restrict__ bias, float* __restrict__ out, int Cin,
                                                  int Cout, int act, int in_parts, float in_scale) {
    extern __shared__ float4 xs[];   // Cin / 4 (+ 3 more copies while the slice sums are being combined)
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int n = blockIdx.y;
    const int c4 = Cin >> 2;
    const float4* xr = reinterpret_cast<const float4*>(in + (long long)n * in_parts * Cin);
    // stage the input vector.  With slice sums (in_parts > 1) and a short vector the 256 threads split the slices
    // G ways (G = 256 / c4, at most 4), so no thread walks all of them serially; the G partial vectors are then
    // added in a fixed order
    const int G = (in_parts > 1 && c4 <= 128) ? (c4 <= 64 ? 4 : 2) : 1;
    for (int i0 = 0; i0 < c4; i0 += 256 / G) {
        const int i = i0 + (int)threadIdx.x % (256 / G), g = (int)threadIdx.x / (256 / G);
        if (i < c4) {
            float4 b = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
            for (int q = g; q < in_parts; q += G) {
                const float4 t = xr[(long long)q * c4 + i];
                b.x += t.x; b.y += t.y; b.z += t.z; b.w += t.w;
            }
            xs[g * c4 + i] = b;
        }
    }
    __syncthreads();
    if (G > 1 || in_parts > 1) {
        for (int i = threadIdx.x; i < c4; i += 256) {
            float4 b = xs[i];
            for (int g = 1; g < G; ++g) {
                const float4 t = xs[g * c4 + i];
                b.x += t.x; b.y += t.y; b.z += t.z; b.w += t.w;
            }
            if (in_parts > 1) { b.x *= in_scale; b.y *= in_scale; b.z *= in_scale; b.w *= in_scale; }
            xs[i] = b;
        }
        __syncthreads();
    }
#pragma unroll
    for (int r = 0; r < FC_PER_BLOCK / 4; ++r) {
        const int o = blockIdx.x * FC_PER_BLOCK + r * 4 + wave;
        if (o >= Cout) continue;
        const float4* wr = reinterpret_cast<const float4*>(w + (long long)o * Cin);
        float s = 0.f;
#pragma unroll 8
        for (int i = lane; i < c4; i += 64) {
            const float4 a = wr[i];
            const float4 b = xs[i];
            s += a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w;
        }
        for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
        if (lane == 0) {
            s += bias[o];
            if (act == 2) s = s > 0.f ? s : 0.f;
            else if (act == 3) s = 1.f / (1.f + __expf(-s));
            out[(long long)n * Cout + o] = s;
        }
    }
}
void launch_fc(const float* in, const float* w, const float* bias, float* out, int N, int Cin, int Cout, int act,
               int in_parts, float in_scale, hipStream_t s) {
    BP_CHECK(Cin % 4 == 0 && Cin * 4 <= 64 * 1024, "fc: Cin % 4, Cin <= 16384");
    hipLaunchKernelGGL(fc_kernel, dim3((Cout + FC_PER_BLOCK - 1) / FC_PER_BLOCK, N), dim3(256), (size_t)Cin * 4 * (Cin <= 512 ? 4 : 1), s, in, w,
                       bias, out, Cin, Cout, act, in_parts, in_scale);
}

// ---------------------------------------------------------------- YOLO head decode (yolo/darknet.py:129-169)
struct YoloHeads {
    YoloHead h[4];
    int n;
};
__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }

__global__ void yolo_decode_kernel(YoloHeads hs, int N, int reso, int attrs, int rows, float* __restrict__ pred) {
    const long long total = (long long)N * rows;
    // 32-bit index math: every tensor on this path has < 2^31 elements (checked by the launchers), and 64-bit
    // div/mod costs ~100 instructions on gfx950
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < (int)total; e += gridDim.x * blockDim.x) {
        const int row = (int)(e % rows);
        const int n = (int)(e / rows);
        int hi = 0;
        for (int k = 1; k < hs.n; ++k)
            if (row >= hs.h[k].row_off) hi = k;
        const YoloHead& H = hs.h[hi];
        const int g = H.g;
        const int r = row - H.row_off;
        const int a = r / (g * g);
        const int cell = r - a * g * g;
        const int gy = cell / g, gx = cell - gy * g;
        const float stride = (float)(reso / g);
        const float* t = H.t + ((long long)n * g * g + cell) * (3 * attrs) + a * attrs;
        float* o = pred + e * attrs;
        // anchors are divided by the stride then the box is multiplied back (darknet.py:151,165)
        o[0] = (sigmoidf_(t[0]) + (float)gx) * stride;
        o[1] = (sigmoidf_(t[1]) + (float)gy) * stride;
        o[2] = (expf(t[2]) * (H.aw[a] / stride)) * stride;
        o[3] = (expf(t[3]) * (H.ah[a] / stride)) * stride;
        for (int k = 4; k < attrs; ++k) o[k] = sigmoidf_(t[k]);
    }
}
void launch_yolo_decode(const YoloHead* heads, int nheads, int N, int reso, int attrs, int rows, float* pred,
                        hipStream_t s) {
    BP_CHECK(nheads <= 4, "at most 4 yolo heads");
    YoloHeads hs;
    hs.n = nheads;
    for (int i = 0; i < nheads; ++i) hs.h[i] = heads[i];
    hipLaunchKernelGGL(yolo_decode_kernel, dim3(grid_for((long long)N * rows)), dim3(256), 0, s, hs, N, reso, attrs, rows, pred);
}

// write_results with nms=False (yolo/util.py:118-223): per image the arg-max objectness row (first on
// ties) among rows with obj > conf whose arg-max class is 0.  One block per image.
__global__ __launch_bounds__(1024) void yolo_select_kernel(const float* __restrict__ pred, int rows, int attrs,
                                                            float conf, int num_classes, float* __restrict__ sel, int sel_ld) {
    __shared__ float sv[16];
    __shared__ int si[16];
    const int n = blockIdx.x;
    const float* P = pred + (long long)n * rows * attrs;
    const int ncls = min(num_classes, attrs - 5);
    float best = -1.f;
    int bi = 0x7fffffff;
    for (int r = threadIdx.x; r < rows; r += blockDim.x) {
        const float* q = P + (long long)r * attrs;
        const float obj = q[4];
        if (!(obj > conf)) continue;
        // class arg-max over the masked row (first max wins); only class 0 rows are kept
        int cls = 0;
        float cm = q[5];
        for (int k = 1; k < ncls; ++k)
            if (q[5 + k] > cm) { cm = q[5 + k]; cls = k; }
        if (cls != 0) continue;
        if (obj > best || (obj == best && r < bi)) { best = obj; bi = r; }
    }
    for (int off = 32; off > 0; off >>= 1) {
        const float ov = __shfl_down(best, off, 64);
        const int oi = __shfl_down(bi, off, 64);
        if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
    }
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { sv[w] = best; si[w] = bi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int k = 1; k < (int)(blockDim.x >> 6); ++k)
            if (sv[k] > best || (sv[k] == best && si[k] < bi)) { best = sv[k]; bi = si[k]; }
        float* o = sel + (long long)n * sel_ld;
        if (best < 0.f) {
            o[0] = __int_as_float(-1);
            for (int k = 1; k < 8; ++k) o[k] = 0.f;
        } else {
            const float* q = P + (long long)bi * attrs;
            o[0] = __int_as_float(bi);
            o[1] = q[0] - q[2] / 2;
            o[2] = q[1] - q[3] / 2;
            o[3] = q[0] + q[2] / 2;
            o[4] = q[1] + q[3] / 2;
            o[5] = q[4];
            o[6] = q[5];
            o[7] = 0.f;
        }
    }
}
void launch_yolo_select(const float* pred, int N, int rows, int attrs, float conf, int num_classes, float* sel,
                        hipStream_t s, int sel_ld) {
    hipLaunchKernelGGL(yolo_select_kernel, dim3(N), dim3(1024), 0, s, pred, rows, attrs, conf, num_classes, sel, sel_ld);
}

// DetectionLayer.forward + write_results in ONE launch (round 4, node diet): when nobody asks for the [rows][attrs] prediction
// tensor -- the fused per-frame pipeline reads the select record only -- every thread decodes the objectness and class scores of its
// rows from the head tensors with the SAME float operations as yolo_decode_kernel, the block reduces with the same first-max rule
// as yolo_select_kernel, and thread 0 decodes the winner's box.  Identical records by construction (asserted in
// tests/test_gpu_stages.py against the two-kernel path).
__global__ __launch_bounds__(1024) void yolo_decode_select_kernel(YoloHeads hs, int reso, int attrs, int rows, float conf, int num_classes,
                                                                   float* __restrict__ sel, int sel_ld) {
    __shared__ float sv[16];
    __shared__ int si[16];
    const int n = blockIdx.x;
    const int ncls = min(num_classes, attrs - 5);
    auto locate = [&](int row, int* a) -> const float* {
        int hi = 0;
        for (int k = 1; k < hs.n; ++k)
            if (row >= hs.h[k].row_off) hi = k;
        const YoloHead& H = hs.h[hi];
        const int g = H.g, r = row - H.row_off;
        *a = r / (g * g);
        const int cell = r - *a * g * g;
        return H.t + ((long long)n * g * g + cell) * (3 * attrs) + *a * attrs;
    };
    float best = -1.f;
    int bi = 0x7fffffff;
    for (int r = threadIdx.x; r < rows; r += blockDim.x) {
        int a;
        const float* t = locate(r, &a);
        const float obj = sigmoidf_(t[4]);
        if (!(obj > conf)) continue;
        int cls = 0;
        float cm = sigmoidf_(t[5]);
        for (int k = 1; k < ncls; ++k) {
            const float v = sigmoidf_(t[5 + k]);
            if (v > cm) { cm = v; cls = k; }
        }
        if (cls != 0) continue;
        if (obj > best || (obj == best && r < bi)) { best = obj; bi = r; }
    }
    for (int off = 32; off > 0; off >>= 1) {
        const float ov = __shfl_down(best, off, 64);
        const int oi = __shfl_down(bi, off, 64);
        if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
    }
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { sv[w] = best; si[w] = bi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int k = 1; k < (int)(blockDim.x >> 6); ++k)
            if (sv[k] > best || (sv[k] == best && si[k] < bi)) { best = sv[k]; bi = si[k]; }
        float* o = sel + (long long)n * sel_ld;
        if (best < 0.f) {
            o[0] = __int_as_float(-1);
            for (int k = 1; k < 8; ++k) o[k] = 0.f;
        } else {
            int a, hi = 0;
            const float* t = locate(bi, &a);
            for (int k = 1; k < hs.n; ++k)
                if (bi >= hs.h[k].row_off) hi = k;
            const YoloHead& H = hs.h[hi];
            const int g = H.g, r = bi - H.row_off, cell = r - a * g * g;
            const int gy = cell / g, gx = cell - gy * g;
            const float stride = (float)(reso / g);
            const float q0 = (sigmoidf_(t[0]) + (float)gx) * stride;
            const float q1 = (sigmoidf_(t[1]) + (float)gy) * stride;
            const float q2 = (expf(t[2]) * (H.aw[a] / stride)) * stride;
            const float q3 = (expf(t[3]) * (H.ah[a] / stride)) * stride;
            o[0] = __int_as_float(bi);
            o[1] = q0 - q2 / 2;
            o[2] = q1 - q3 / 2;
            o[3] = q0 + q2 / 2;
            o[4] = q1 + q3 / 2;
            o[5] = sigmoidf_(t[4]);
            o[6] = sigmoidf_(t[5]);
            o[7] = 0.f;
        }
    }
}
void launch_yolo_decode_select(const YoloHead* heads, int nheads, int N, int reso, int attrs, int rows, float conf, int num_classes,
                               float* sel, hipStream_t s, int sel_ld) {
    BP_CHECK(nheads <= 4, "at most 4 yolo heads");
    YoloHeads hs;
    hs.n = nheads;
    for (int i = 0; i < nheads; ++i) hs.h[i] = heads[i];
    hipLaunchKernelGGL(yolo_decode_select_kernel, dim3(N), dim3(1024), 0, s, hs, reso, attrs, rows, conf, num_classes, sel, sel_ld);
}

// ---------------------------------------------------------------- heat-map arg-max (+4 neighbours), eval.py:113-147
__global__ __launch_bounds__(256) void heatmap_argmax_kernel(const float* __restrict__ hm, int H, int W,
                                                              float* __restrict__ out, int C, int out_ld) {
    __shared__ float sv[4];
    __shared__ int si[4];
    const int HW = H * W;
    const float* P = hm + (long long)blockIdx.x * HW;
    float best = -INFINITY;
    int bi = 0x7fffffff;
    for (int i = threadIdx.x; i < HW; i += 256) {
        const float v = P[i];
        if (v > best) { best = v; bi = i; }   // ascending i per thread: first max kept
    }
    for (int off = 32; off > 0; off >>= 1) {
        const float ov = __shfl_down(best, off, 64);
        const int oi = __shfl_down(bi, off, 64);
        if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
    }
    if ((threadIdx.x & 63) == 0) { sv[threadIdx.x >> 6] = best; si[threadIdx.x >> 6] = bi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int k = 1; k < 4; ++k)
            if (sv[k] > best || (sv[k] == best && si[k] < bi)) { best = sv[k]; bi = si[k]; }
        if (bi < 0 || bi >= HW) { bi = 0; best = P[0]; }   // no element compared greater (all -inf / NaN): stay in bounds
        float* o = out + (long long)(blockIdx.x / C) * out_ld + (blockIdx.x % C) * 6;
        const int x = bi % W, y = bi / W;
        o[0] = __int_as_float(bi);
        o[1] = best;
        const bool inner = x > 0 && x < W - 1 && y > 0 && y < H - 1;
        o[2] = inner ? P[bi - 1] : 0.f;
        o[3] = inner ? P[bi + 1] : 0.f;
        o[4] = inner ? P[bi - W] : 0.f;
        o[5] = inner ? P[bi + W] : 0.f;
    }
}
void launch_heatmap_argmax(const float* hm, int N, int C, int H, int W, float* out, hipStream_t s, int out_ld) {
    hipLaunchKernelGGL(heatmap_argmax_kernel, dim3(N * C), dim3(256), 0, s, hm, H, W, out, C, out_ld > 0 ? out_ld : C * 6);
}

// ---------------------------------------------------------------- crop (dataloader.py:794-835, img.py:242-262)
// One launch: thread 0 of each block recomputes the (cheap) window parameters; every thread then
// produces output pixels.  The frame is BGR u8; the crop is RGB/255 - (0.406, 0.457, 0.480).
struct CropWin {
    float ulx, uly, brx, bry;  // pt1 / pt2 (float, pre-truncation)
    int x0, y0, cw, ch;        // integer crop origin and size
    int padl, padt, PW, PH;    // centred zero pad into a PH x PW canvas
};
__device__ __forceinline__ CropWin crop_window(float x1, float y1, float x2, float y2, int H, int W, int oh, int ow) {
    CropWin c;
    const float ht = y2 - y1, width = x2 - x1;
    const float rate = width > 100.f ? 0.2f : 0.3f;
    c.ulx = fmaxf(0.f, x1 - width * rate / 2);
    c.uly = fmaxf(0.f, y1 - ht * rate / 2);
    c.brx = fmaxf(fminf((float)(W - 1), x2 + width * rate / 2), c.ulx + 5);
    c.bry = fmaxf(fminf((float)(H - 1), y2 + ht * rate / 2), c.uly + 5);
    const int ulx = (int)c.ulx, uly = (int)c.uly, brx = (int)c.brx, bry = (int)c.bry;
    // cropBox under torch>=1.x division semantics: int*int/int is a true division (fp32)
    const int bh = bry - uly, bw = brx - ulx;
    const float wscaled = (float)(bw * oh) / (float)ow;
    float lenH = (float)bh >= wscaled ? (float)bh : wscaled;   // python max(a, b): first wins ties
    float lenW = lenH * (float)ow / (float)oh;
    c.PH = (int)lenH;
    c.PW = (int)lenW;
    c.x0 = ulx;
    c.y0 = uly;
    c.ch = min(bh, H - uly);
    c.cw = min(bw, W - ulx);
    c.ch = max(c.ch, 0);
    c.cw = max(c.cw, 0);
    const int dh = max(c.PH - c.ch, 0), dw = max(c.PW - c.cw, 0);
    c.padt = (dh + 1) >> 1;  // ceil(d/2) before, floor(d/2) after
    c.padl = (dw + 1) >> 1;
    // Pad never crops: canvas is at least the crop
    c.PH = max(c.PH, c.ch);
    c.PW = max(c.PW, c.cw);
    return c;
}
__device__ __forceinline__ float crop_tap(const uint8_t* __restrict__ frame, int W, const CropWin& c, int y, int x, int ch) {
    const int cy = y - c.padt, cx = x - c.padl;
    if ((unsigned)cy >= (unsigned)c.ch || (unsigned)cx >= (unsigned)c.cw) return 0.f;
    const uint8_t* px = frame + ((long long)(c.y0 + cy) * W + (c.x0 + cx)) * 3;
    // ch: 0=R,1=G,2=B of the RGB image; frame is BGR
    const float mean = ch == 0 ? 0.406f : (ch == 1 ? 0.457f : 0.480f);
    return (float)px[2 - ch] / 255.f - mean;
}
__global__ __launch_bounds__(256) void crop_kernel(const uint8_t* __restrict__ frame, int H, int W,
                                                    const float* __restrict__ sel, int reso,
                                                    const float* __restrict__ box_override, float* __restrict__ out_nhwc,
                                                    float* __restrict__ out_nchw, float* __restrict__ pts, int oh, int ow,
                                                    int sel_ld, int pts_ld) {
    const int img = blockIdx.y;
    frame += (long long)img * H * W * 3;
    if (sel) sel += (long long)img * sel_ld;
    if (box_override) box_override += img * 4;
    if (out_nhwc) out_nhwc += (long long)img * oh * ow * 3;
    if (out_nchw) out_nchw += (long long)img * oh * ow * 3;
    if (pts) pts += (long long)img * pts_ld;
    float x1, y1, x2, y2;
    if (box_override) {
        x1 = box_override[0]; y1 = box_override[1]; x2 = box_override[2]; y2 = box_override[3];
    } else {
        // box rescale of DetectionLoader.update (dataloader.py:354-364): stretch back to frame pixels
        const float wr = (float)W / (float)reso, hr = (float)H / (float)reso;
        x1 = sel[1] * wr; y1 = sel[2] * hr; x2 = sel[3] * wr; y2 = sel[4] * hr;
    }
    const CropWin c = crop_window(x1, y1, x2, y2, H, W, oh, ow);
    if (blockIdx.x == 0 && threadIdx.x == 0 && pts) {
        pts[0] = c.ulx; pts[1] = c.uly; pts[2] = c.brx; pts[3] = c.bry;
        pts[4] = x1; pts[5] = y1; pts[6] = x2; pts[7] = y2;
    }
    // bilinear, align_corners=True (torch upsample_bilinear2d): src = dst*(in-1)/(out-1)
    const float sy = oh > 1 ? (float)(c.PH - 1) / (float)(oh - 1) : 0.f;
    const float sx = ow > 1 ? (float)(c.PW - 1) / (float)(ow - 1) : 0.f;
    const int total = oh * ow;
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
        const int oy = e / ow, ox = e - oy * ow;
        const float fy = sy * oy, fx = sx * ox;
        const int y0 = (int)fy, x0i = (int)fx;
        const int y1i = y0 + (y0 < c.PH - 1 ? 1 : 0), x1i = x0i + (x0i < c.PW - 1 ? 1 : 0);
        const float ly = fy - y0, lx = fx - x0i;
        const float hy = 1.f - ly, hx = 1.f - lx;
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) {
            const float v = hy * (hx * crop_tap(frame, W, c, y0, x0i, ch) + lx * crop_tap(frame, W, c, y0, x1i, ch)) +
                            ly * (hx * crop_tap(frame, W, c, y1i, x0i, ch) + lx * crop_tap(frame, W, c, y1i, x1i, ch));
            if (out_nhwc) out_nhwc[(long long)e * 3 + ch] = v;
            if (out_nchw) out_nchw[(long long)ch * total + e] = v;
        }
    }
}
void launch_crop(const uint8_t* frames, int batch, int H, int W, const float* sel, int reso, const float* boxes,
                 float* out_nhwc, float* out_nchw, float* pts, int oh, int ow, hipStream_t s, int sel_ld, int pts_ld) {
    hipLaunchKernelGGL(crop_kernel, dim3(grid_for((long long)oh * ow, 256, 1024), batch), dim3(256), 0, s, frames, H, W,
                       sel, reso, boxes, out_nhwc, out_nchw, pts, oh, ow, sel_ld, pts_ld);
}

// ---------------------------------------------------------------- Pillow-exact bicubic resize (u8, two passes)
// Pillow's ImagingResample for 8-bit: per output pixel a window [xmin, xmin+xsize) with integer
// coefficients (PRECISION_BITS = 22), accumulate ss = 1<<21 + sum(pix*k), result = clip8(ss >> 22);
// horizontal pass first, then vertical, each rounding to u8.
// (one thread per output PIXEL: the window and its coefficients are fetched once for the three channels, three independent sums)
__global__ void resize_h_kernel(const uint8_t* __restrict__ in, int H, int W, uint8_t* __restrict__ tmp, int ow,
                                const int* __restrict__ hb, const int* __restrict__ hk, int ks) {
    in += (long long)blockIdx.y * H * W * 3;
    tmp += (long long)blockIdx.y * H * ow * 3;
    const int total = H * ow;
    for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < total; t += gridDim.x * blockDim.x) {
        const int ox = t % ow, y = t / ow;
        const int xmin = hb[2 * ox], n = hb[2 * ox + 1];
        const int* k = hk + ox * ks;
        int s0 = 1 << 21, s1 = 1 << 21, s2 = 1 << 21;
        const uint8_t* row = in + ((long long)y * W + xmin) * 3;
        for (int i = 0; i < n; ++i) {
            const int c = k[i];
            s0 += (int)row[i * 3] * c; s1 += (int)row[i * 3 + 1] * c; s2 += (int)row[i * 3 + 2] * c;
        }
        s0 >>= 22; s1 >>= 22; s2 >>= 22;
        uint8_t* o = tmp + (long long)t * 3;
        o[0] = (uint8_t)(s0 < 0 ? 0 : (s0 > 255 ? 255 : s0));
        o[1] = (uint8_t)(s1 < 0 ? 0 : (s1 > 255 ? 255 : s1));
        o[2] = (uint8_t)(s2 < 0 ? 0 : (s2 > 255 ? 255 : s2));
    }
}
__global__ void resize_v_kernel(const uint8_t* __restrict__ tmp, int H, int ow, float* __restrict__ out_nhwc,
                                uint8_t* __restrict__ out_u8, int oh, const int* __restrict__ vb,
                                const int* __restrict__ vk, int ks, int swap_rb) {
    tmp += (long long)blockIdx.y * H * ow * 3;
    if (out_nhwc) out_nhwc += (long long)blockIdx.y * oh * ow * 3;
    if (out_u8) out_u8 += (long long)blockIdx.y * oh * ow * 3;
    const int total = oh * ow;
    for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < total; t += gridDim.x * blockDim.x) {
        const int ox = t % ow, oy = t / ow;
        const int ymin = vb[2 * oy], n = vb[2 * oy + 1];
        const int* k = vk + oy * ks;
        int ss[3] = {1 << 21, 1 << 21, 1 << 21};
        const uint8_t* col = tmp + ((long long)ymin * ow + ox) * 3;
        for (int i = 0; i < n; ++i) {
            const int c = k[i];
            const uint8_t* q = col + (long long)i * ow * 3;
            ss[0] += (int)q[0] * c; ss[1] += (int)q[1] * c; ss[2] += (int)q[2] * c;
        }
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) {
            const int r = ss[swap_rb ? 2 - ch : ch] >> 22;
            const int v = r < 0 ? 0 : (r > 255 ? 255 : r);
            if (out_u8) out_u8[(long long)t * 3 + ch] = (uint8_t)v;
            if (out_nhwc) out_nhwc[(long long)t * 3 + ch] = (float)v / 255.f;
        }
    }
}
void launch_resize_bicubic(const uint8_t* in, int batch, int H, int W, uint8_t* tmp, float* out_nhwc, uint8_t* out_u8,
                           int oh, int ow, const ResizeTables& t, int swap_rb, hipStream_t s) {
    hipLaunchKernelGGL(resize_h_kernel, dim3(grid_for((long long)H * ow), batch), dim3(256), 0, s, in, H, W, tmp, ow,
                       t.hb, t.hk, t.ksize_h);
    hipLaunchKernelGGL(resize_v_kernel, dim3(grid_for((long long)oh * ow), batch), dim3(256), 0, s, tmp, H, ow,
                       out_nhwc, out_u8, oh, t.vb, t.vk, t.ksize_v, swap_rb);
}

// ---- the same three planes, STAGE-PACKED for the filters-direct kernel (conv_igemm.hip; also conv_kg.hip / conv_rd.hip): per 64-row filter tile and 16-k stage one contiguous block
// [plane][row 0..63][32 B] that already is the kernel's LDS image (granule g of row r at slot g ^ ((r >> 3) & 1)), so a
// stage arrives by six 1 KB lane-linear DMA instructions from 6 KB of consecutive addresses
__global__ void f32_to_bf16x3_staged_kernel(const float* __restrict__ in, __bf16* __restrict__ out, int CoutPad, int Kpad, int Cin_pl) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)CoutPad * Kpad) return;
    const int n = (int)(i / Kpad);
    int k = (int)(i - (long long)n * Kpad);
    if (Cin_pl > 0) {   // source K order (tap, channel) -> conv_pl.hip's (32-channel group, tap, channel % 32)
        const int tap = k / Cin_pl, ci = k - tap * Cin_pl;
        k = ((ci >> 5) * (Kpad / Cin_pl) + tap) * 32 + (ci & 31);
    }
    const float x = in[i];
    const __bf16 h1 = (__bf16)x;
    const float r1 = x - (float)h1;
    const __bf16 h2 = (__bf16)r1;
    const float r2 = r1 - (float)h2;
    const int tile = n >> 6, r = n & 63, stage = k >> 4, kk = k & 15;
    const int slot = (kk >> 3) ^ ((r >> 3) & 1);
    const long long base = ((long long)tile * (Kpad >> 4) + stage) * (3 * 64 * 16) + r * 16 + slot * 8 + (kk & 7);
    out[base] = h1;
    out[base + 64 * 16] = h2;
    out[base + 2 * 64 * 16] = (__bf16)r2;
}

void launch_f32_to_bf16x3_staged(const float* in, unsigned short* out, int CoutPad, int Kpad, hipStream_t s, int Cin_pl) {
    const long long n = (long long)CoutPad * Kpad;
    BP_CHECK(Cin_pl == 0 || (Cin_pl % 32 == 0 && Kpad % Cin_pl == 0), "stage-packed filters in conv_pl order: Cin % 32, Kpad = taps * Cin");
    hipLaunchKernelGGL(f32_to_bf16x3_staged_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, in,
                       reinterpret_cast<__bf16*>(out), CoutPad, Kpad, Cin_pl);
}

#ifdef BP_EXPERIMENTAL   // filter formats of the other round-1/2 16-bit kernels (conv_igemm_h staged / conv_w64)
// ---- packed filters fp32 -> fp16 (RNE), once per weight store when the fp16-MFMA path is switched on
__global__ void f32_to_f16_kernel(const float* __restrict__ in, _Float16* __restrict__ out, long long n) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (_Float16)in[i];
}

void launch_f32_to_f16(const float* in, unsigned short* out, long long n, hipStream_t s) {
    const int threads = 256;
    hipLaunchKernelGGL(f32_to_f16_kernel, dim3((unsigned)((n + threads - 1) / threads)), dim3(threads), 0, s, in,
                       reinterpret_cast<_Float16*>(out), n);
}

// ---- packed filters fp32 -> three bf16 planes with x == p0 + p1 + p2 exactly (PREC_BF16X3)
__global__ void f32_to_bf16x3_kernel(const float* __restrict__ in, __bf16* __restrict__ out, long long n) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float x = in[i];
    const __bf16 h1 = (__bf16)x;
    const float r1 = x - (float)h1;
    const __bf16 h2 = (__bf16)r1;
    const float r2 = r1 - (float)h2;
    out[i] = h1;
    out[n + i] = h2;
    out[2 * n + i] = (__bf16)r2;
}

void launch_f32_to_bf16x3(const float* in, unsigned short* out, long long n, hipStream_t s) {
    const int threads = 256;
    hipLaunchKernelGGL(f32_to_bf16x3_kernel, dim3((unsigned)((n + threads - 1) / threads)), dim3(threads), 0, s, in,
                       reinterpret_cast<__bf16*>(out), n);
}

// fp16 operands, stage-packed the same way (one plane: 2 KB per 64-row tile and 16-k stage)
__global__ void f32_to_f16_staged_kernel(const float* __restrict__ in, _Float16* __restrict__ out, int CoutPad, int Kpad) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)CoutPad * Kpad) return;
    const int n = (int)(i / Kpad), k = (int)(i - (long long)n * Kpad);
    const int tile = n >> 6, r = n & 63, stage = k >> 4, kk = k & 15;
    const int slot = (kk >> 3) ^ ((r >> 3) & 1);
    out[((long long)tile * (Kpad >> 4) + stage) * (64 * 16) + r * 16 + slot * 8 + (kk & 7)] = (_Float16)in[i];
}

void launch_f32_to_f16_staged(const float* in, unsigned short* out, int CoutPad, int Kpad, hipStream_t s) {
    const long long n = (long long)CoutPad * Kpad;
    hipLaunchKernelGGL(f32_to_f16_staged_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, in,
                       reinterpret_cast<_Float16*>(out), CoutPad, Kpad);
}

#endif   // BP_EXPERIMENTAL

// ---- placement probe: which XCD / CU every workgroup of a grid landed on (CU-mask experiments, tests)
__global__ void probe_placement_kernel(int* out) {
    if (threadIdx.x == 0) {
        const unsigned xcc = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | ((4 - 1) << 11));    // HW_REG_XCC_ID[3:0]
        const unsigned hw = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | ((32 - 1) << 11));     // HW_REG_HW_ID
        out[2 * blockIdx.x] = (int)xcc;
        out[2 * blockIdx.x + 1] = (int)hw;
    }
}

// ---- clock calibration for the s_memtime stamps of the convolution kernels (c_api.cpp bp_calibrate_ticks)
__global__ void spin_ticks_kernel(long long ticks) {
    const unsigned long long t0 = bp_clock();
    while ((long long)(bp_clock() - t0) < ticks) __builtin_amdgcn_s_sleep(1);
}
void launch_spin_ticks(long long ticks, hipStream_t s) { hipLaunchKernelGGL(spin_ticks_kernel, dim3(1), dim3(1), 0, s, ticks); }

void launch_probe_placement(int* d_out, int blocks, hipStream_t s) {
    hipLaunchKernelGGL(probe_placement_kernel, dim3(blocks), dim3(64), 0, s, d_out);
}

}  // namespace bp
