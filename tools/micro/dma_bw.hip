// LDS-DMA (buffer_load_dwordx4 ... lds) throughput micro-benchmark for gfx950 (MI355X): bytes per clock per CU that a
// conv_pl.hip-style ring can move, by where the data comes from and how many bytes are in flight.
//   footprint per block F: 16 KB (lives in the CU's L1), 512 KB (the XCD's L2), 64 MB shared by all blocks (L2 / MALL
//   with re-use across blocks), 8 GB-ish private streams (HBM)
//   in-flight depth D: DMA instructions per wave outstanding (counted vmcnt), 4 waves per block, 1 or 2 blocks per CU
// Build + run on the GPU box:  tools/micro/run_dma_bw.sh
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef __attribute__((address_space(3))) void lds_void_t;

template <int D, int LDSKB>
__global__ __launch_bounds__(256) void dma_kernel(const char* src, unsigned long long bytes_total, unsigned foot_mask, unsigned block_stride,
                                                  int iters, unsigned long long* cycles) {
    __shared__ __attribute__((aligned(16))) char smem[LDSKB * 1024];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(src), 0, (int)(bytes_total > 0x7fffff00ull ? 0x7fffff00ull : bytes_total), 0x00020000);
    const unsigned base = (unsigned)(((unsigned long long)blockIdx.x * block_stride) % (bytes_total - foot_mask - 1));
    unsigned off = wave * 1024 + lane * 16;
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < D; ++k) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void_t*)(smem + ((wave * D + k) * 1024) % (LDSKB * 1024)), 16,
                                                     (int)(base + (off & foot_mask)), 0, 0, 0);
            off += 4096;     // the block's four waves cover 4 KB per round
        }
        asm volatile("s_waitcnt vmcnt(%0)" :: "n"(D / 2) : "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (tid == 0) cycles[blockIdx.x] = t1 - t0;
}

template <int D, int LDSKB>
static void run(const char* name, const char* d_src, unsigned long long total, unsigned foot, unsigned stride, int blocks, unsigned long long* d_cyc) {
    const int iters = 2000 / D * 4;
    hipLaunchKernelGGL((dma_kernel<D, LDSKB>), dim3(blocks), dim3(256), 0, 0, d_src, total, foot - 1, stride, iters, d_cyc);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL((dma_kernel<D, LDSKB>), dim3(blocks), dim3(256), 0, 0, d_src, total, foot - 1, stride, iters, d_cyc);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(blocks);
    hipMemcpy(h.data(), d_cyc, blocks * 8, hipMemcpyDeviceToHost);
    double mean = 0; for (auto v : h) mean += (double)v; mean /= blocks;
    const double bytes_block = (double)iters * D * 4 * 1024;
    const int per_cu = blocks / 256;
    printf("%-34s D=%2d (in flight per CU %4d KB) blocks/CU %d : %6.1f B/clk/CU by block cycles, %6.2f TB/s chip by wall clock\n", name, D,
           D * 4 * per_cu, per_cu, bytes_block * per_cu / mean, bytes_block * blocks / (ms * 1e-3) / 1e12);
}

int main() {
    const unsigned long long total = 1ull << 31;   // 2 GB source
    char* d_src; hipMalloc(&d_src, total); hipMemset(d_src, 1, total);
    unsigned long long* d_cyc; hipMalloc(&d_cyc, 4096 * 8);
    for (int per_cu = 1; per_cu <= 2; ++per_cu) {
        const int blocks = 256 * per_cu;
        run<8, 64>("L1-resident (16 KB per block)", d_src, total, 16 << 10, 1 << 20, blocks, d_cyc);
        run<16, 64>("L1-resident (16 KB per block)", d_src, total, 16 << 10, 1 << 20, blocks, d_cyc);
        run<8, 64>("L2-resident (256 KB per block)", d_src, total, 256 << 10, 1 << 20, blocks, d_cyc);
        run<16, 64>("L2-resident (256 KB per block)", d_src, total, 256 << 10, 1 << 20, blocks, d_cyc);
        run<8, 64>("shared 1 MB (all blocks, L2 hits)", d_src, total, 1 << 20, 0, blocks, d_cyc);
        run<16, 64>("shared 1 MB (all blocks, L2 hits)", d_src, total, 1 << 20, 0, blocks, d_cyc);
        run<4, 64>("HBM stream (4 MB per block)", d_src, total, 4 << 20, 4 << 20, blocks, d_cyc);
        run<8, 64>("HBM stream (4 MB per block)", d_src, total, 4 << 20, 4 << 20, blocks, d_cyc);
        run<16, 64>("HBM stream (4 MB per block)", d_src, total, 4 << 20, 4 << 20, blocks, d_cyc);
    }
    return 0;
}
