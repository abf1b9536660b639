// Where does a kernel's FIRST read of a buffer come from, and how fast can one CU pull it?  (gfx950 / MI355X)
// B blocks (one per CU while B <= 256) each read a private F-byte region ONCE through an LDS-DMA ring (as conv_pl.hip reads
// its filters), timed per block with s_memrealtime (100 MHz).  Scenarios:
//   cold      : 1 GB of other data streamed through the chip just before (region in HBM only)
//   prev-same : the previous kernel on the stream read the same regions with the same block -> region mapping
//               (same XCD: does the XCD's L2 keep clean lines across a kernel boundary?)
//   prev-shift: the previous kernel read them with the mapping shifted by one block (another XCD's L2; MALL at best)
//   2nd-pass  : the same block reads its region twice inside one launch, second pass timed (true L2 / L1-miss hits)
// Build + run on the GPU box:  tools/micro/run_cold_fetch.sh
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

typedef __attribute__((address_space(3))) void lds_void_t;

template <int D>
__global__ __launch_bounds__(256) void fetch_kernel(const char* src, unsigned region, int shift, int passes, unsigned long long* ticks) {
    __shared__ __attribute__((aligned(16))) char smem[64 * 1024];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int blk = ((int)blockIdx.x + shift) % (int)gridDim.x;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(src) + (size_t)blk * region, 0, (int)region, 0x00020000);
    unsigned long long t0 = 0;
    for (int ps = 0; ps < passes; ++ps) {
        __syncthreads();
        t0 = __builtin_amdgcn_s_memrealtime();
        unsigned off = wave * 1024 + lane * 16;
#pragma unroll 1
        for (unsigned it = 0; it < region / (4096 * D); ++it) {
#pragma unroll
            for (int k = 0; k < D; ++k) {
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void_t*)(smem + ((wave * D + k) * 1024) % (64 * 1024)), 16, (int)off, 0, 0, 0);
                off += 4096;
            }
            asm volatile("s_waitcnt vmcnt(%0)" :: "n"(D / 2) : "memory");
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
    if (tid == 0) ticks[blockIdx.x] = t1 - t0;
}

__global__ void flush_kernel(const float4* p, size_t n, float* sink) {
    float a = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) a += p[i].x;
    if (a == 12345.f) *sink = a;
}

int main() {
    const size_t total = 1ull << 30;
    char* d_src; hipMalloc(&d_src, total); hipMemset(d_src, 1, total);
    char* d_flush; hipMalloc(&d_flush, total); hipMemset(d_flush, 2, total);
    float* sink; hipMalloc(&sink, 4);
    unsigned long long* d_t; hipMalloc(&d_t, 4096 * 8);
    auto flush = [&]() { hipLaunchKernelGGL(flush_kernel, dim3(2048), dim3(256), 0, 0, (const float4*)d_flush, total / 16, sink); };
    auto report = [&](const char* name, int blocks, unsigned region) {
        hipDeviceSynchronize();
        std::vector<unsigned long long> h(blocks);
        hipMemcpy(h.data(), d_t, blocks * 8, hipMemcpyDeviceToHost);
        std::sort(h.begin(), h.end());
        const double med = (double)h[blocks / 2] * 10.0, mx = (double)h[blocks - 1] * 10.0;   // ns
        printf("  %-11s median %7.2f us (%5.1f GB/s per CU = %5.1f B/clk at 2.4 GHz)   slowest %7.2f us\n", name, med * 1e-3, region / med, region / med / 2.4, mx * 1e-3);
    };
    constexpr int D = 8;
    for (unsigned region : {64u << 10, 256u << 10}) for (int blocks : {16, 64, 128, 256, 512}) {
        printf("%d blocks x %u KB (%.1f MB), ring of %d KB per block\n", blocks, region >> 10, blocks * (region / 1048576.0), D * 4);
        flush(); hipLaunchKernelGGL(fetch_kernel<D>, dim3(blocks), dim3(256), 0, 0, d_src, region, 0, 1, d_t); report("cold", blocks, region);
        flush(); hipLaunchKernelGGL(fetch_kernel<D>, dim3(blocks), dim3(256), 0, 0, d_src, region, 0, 1, d_t);
        hipLaunchKernelGGL(fetch_kernel<D>, dim3(blocks), dim3(256), 0, 0, d_src, region, 0, 1, d_t); report("prev-same", blocks, region);
        flush(); hipLaunchKernelGGL(fetch_kernel<D>, dim3(blocks), dim3(256), 0, 0, d_src, region, 1, 1, d_t);
        hipLaunchKernelGGL(fetch_kernel<D>, dim3(blocks), dim3(256), 0, 0, d_src, region, 0, 1, d_t); report("prev-shift", blocks, region);
        flush(); hipLaunchKernelGGL(fetch_kernel<D>, dim3(blocks), dim3(256), 0, 0, d_src, region, 0, 2, d_t); report("2nd-pass", blocks, region);
    }
    return 0;
}
