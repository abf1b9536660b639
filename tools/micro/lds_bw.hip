// LDS bandwidth micro-benchmark for gfx950 (MI355X): bytes per clock per CU of 16-B LDS reads / writes in the access
// patterns the conv kernels use (row-strided fragment reads of 64 lanes; 8-B and 16-B row stores), with 1..4 blocks
// of 256 threads per CU.  Build + run on the GPU box:  tools/micro/run_lds_bw.sh
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

template <int MODE>   // 0: ds_read_b128 conflict-free pattern, 1: ds_write_b128, 2: ds_write_b64
__global__ __launch_bounds__(256) void lds_kernel(int iters, unsigned long long* cycles, unsigned* sink) {
    __shared__ __attribute__((aligned(16))) unsigned smem[8192];   // 32 KB
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 8192; i += 256) smem[i] = i;
    __syncthreads();
    // fragment-style address: row (lane&31) of a 36-dword padded row, 4-dword granule (lane>>5), per-wave base
    const int base = wave * 2048 + (lane & 31) * 36 + (lane >> 5) * 4;
    u32x4 acc = {0, 0, 0, 0};
    const unsigned long long t0 = __builtin_readcyclecounter();
    if (MODE == 0) {
#pragma unroll 1
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                u32x4 v;
                const unsigned addr = (unsigned)(((base + k * 8 + (it & 7) * 288) & 8188) * 4);
                asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr) : "memory");
                asm volatile("s_waitcnt lgkmcnt(7)" ::: "memory");
                acc += v;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
    } else if (MODE == 1) {
#pragma unroll 1
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const unsigned addr = (unsigned)(((base + k * 8 + (it & 7) * 288) & 8188) * 4);
                asm volatile("ds_write_b128 %0, %1" :: "v"(addr), "v"(acc) : "memory");
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            acc.x += it;
        }
    } else {
#pragma unroll 1
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const unsigned addr = (unsigned)(((base + k * 8 + (it & 7) * 288) & 8190) * 4);
                const u32x2 v2 = {acc.x, acc.y};
                asm volatile("ds_write_b64 %0, %1" :: "v"(addr), "v"(v2) : "memory");
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            acc.x += it;
        }
    }
    __syncthreads();
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (tid == 0) cycles[blockIdx.x] = t1 - t0;
    if (acc.x == 0xdeadbeef) sink[0] = acc.x + acc.y + acc.z + acc.w + smem[tid];
}

template <int MODE>
static void run(const char* name, int bytes_per_lane, int blocks_per_cu) {
    const int cus = 256, iters = 2000;
    const int blocks = cus * blocks_per_cu;
    unsigned long long* d_c; unsigned* d_s;
    hipMalloc(&d_c, blocks * sizeof(unsigned long long)); hipMalloc(&d_s, 4);
    hipLaunchKernelGGL(lds_kernel<MODE>, dim3(blocks), dim3(256), 0, 0, 10, d_c, d_s);
    hipLaunchKernelGGL(lds_kernel<MODE>, dim3(blocks), dim3(256), 0, 0, iters, d_c, d_s);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(blocks);
    hipMemcpy(h.data(), d_c, blocks * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    double mean = 0; for (auto v : h) mean += (double)v; mean /= blocks;
    const double bytes_per_block = (double)iters * 8 * 256 * bytes_per_lane;
    // all blocks of a CU run concurrently: CU throughput = blocks_per_cu * bytes / cycles
    printf("%-34s %d block(s)/CU: %7.1f cycles/iter/block -> %6.1f B/clk/CU\n", name, blocks_per_cu, mean / iters,
           blocks_per_cu * bytes_per_block / mean);
    hipFree(d_c); hipFree(d_s);
}

int main() {
    for (int b = 1; b <= 4; ++b) run<0>("ds_read_b128 (fragment pattern)", 16, b);
    for (int b = 1; b <= 4; ++b) run<1>("ds_write_b128", 16, b);
    for (int b = 1; b <= 4; ++b) run<2>("ds_write_b64", 8, b);
    return 0;
}
