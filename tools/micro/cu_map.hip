// Which CU does block b of a 1-D grid run on?  (gfx950 / MI355X)  Every block records XCC_ID and HW_ID (SE / SH / CU); the host counts, per
// launch, the distinct CUs used and the most blocks any CU received while all blocks of the grid are alive (they spin ~30 us).
// Question: do the 86-240-block launches of the batch-1 frame spread one block per CU, or pack several on some CUs while others idle?
// Build + run on the GPU box:  tools/micro/run_cu_map.sh
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <vector>

template <int NT, int LDSKB>
__global__ __launch_bounds__(NT) void probe(unsigned* out, int spin) {
    __shared__ char smem[LDSKB * 1024];
    unsigned xcc, hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    if (threadIdx.x == 0) out[blockIdx.x] = ((xcc & 15) << 16) | ((hw >> 8) & 0xff);      // cu_id[3:0], sh_id, se_id[2:0]
    unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)spin) { smem[threadIdx.x] = (char)spin; }
}

template <int NT, int LDSKB>
static void run(int grid, unsigned* d, hipStream_t s) {
    hipLaunchKernelGGL((probe<NT, LDSKB>), dim3(grid), dim3(NT), 0, s, d, 3000);     // 3000 ticks of the 100 MHz clock = 30 us
    hipStreamSynchronize(s);
    std::vector<unsigned> h(grid);
    hipMemcpy(h.data(), d, grid * 4, hipMemcpyDeviceToHost);
    std::map<unsigned, int> per;
    for (unsigned v : h) ++per[v];
    int mx = 0; std::map<int, int> hist;
    for (auto& kv : per) { mx = kv.second > mx ? kv.second : mx; ++hist[kv.second]; }
    printf("threads %4d  LDS %3d KB  grid %4d: %3zu distinct CUs, max %d blocks on one CU; CUs by block count:", NT, LDSKB, grid, per.size(), mx);
    for (auto& kv : hist) printf(" %dx%d", kv.second, kv.first);
    printf("\n");
}

// four launches alive at once on four streams (the batch-1 pipeline's situation): blocks per CU over all four
template <int NT, int LDSKB>
static void run4(int grid, unsigned** d, hipStream_t* st) {
    for (int k = 0; k < 4; ++k) hipLaunchKernelGGL((probe<NT, LDSKB>), dim3(grid), dim3(NT), 0, st[k], d[k], 6000);
    hipDeviceSynchronize();
    std::map<unsigned, int> per;
    for (int k = 0; k < 4; ++k) {
        std::vector<unsigned> h(grid);
        hipMemcpy(h.data(), d[k], grid * 4, hipMemcpyDeviceToHost);
        for (unsigned v : h) ++per[v];
    }
    int mx = 0; std::map<int, int> hist;
    for (auto& kv : per) { mx = kv.second > mx ? kv.second : mx; ++hist[kv.second]; }
    printf("4 streams x grid %4d (threads %d, LDS %d KB): %3zu distinct CUs, max %d blocks on one CU; CUs by block count:", grid, NT, LDSKB, per.size(), mx);
    for (auto& kv : hist) printf(" %dx%d", kv.second, kv.first);
    printf("\n");
}

int main() {
    hipStream_t s; hipStreamCreate(&s);
    unsigned* d; hipMalloc(&d, 8192 * 4);
    {
        hipStream_t st[4]; unsigned* dd[4];
        for (int k = 0; k < 4; ++k) { hipStreamCreate(&st[k]); hipMalloc(&dd[k], 8192 * 4); }
        for (int rep = 0; rep < 2; ++rep)
            for (int grid : {64, 86, 144, 192, 240}) { run4<512, 50>(grid, dd, st); run4<256, 24>(grid, dd, st); }
    }
    for (int grid : {86, 144, 192, 240, 256, 344, 512}) {
        run<256, 24>(grid, d, s);
        run<512, 50>(grid, d, s);
        run<512, 90>(grid, d, s);
        run<256, 68>(grid, d, s);
    }
    return 0;
}
