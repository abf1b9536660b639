// Which XCD does block b of a 1-D grid run on?  (gfx950 / MI355X)  Every block records s_getreg(HW_REG_XCC_ID); the host
// checks xcc == (b + c) % 8 with one c per launch (c may differ by stream / hardware queue) for grids of many sizes, launched alone and from four streams at once (other kernels in flight),
// with small and large LDS footprints.  Build + run on the GPU box:  tools/micro/run_xcc_map.sh
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int LDSKB>
__global__ __launch_bounds__(256) void probe(int* xcc, int spin) {
    __shared__ char smem[LDSKB * 1024];
    unsigned id;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
    if (threadIdx.x == 0) xcc[blockIdx.x] = (int)(id & 15);
    // keep the block alive for a while so that grids overlap
    unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)spin) { smem[threadIdx.x] = (char)spin; }
}

int main() {
    const int sizes[] = {1, 7, 8, 9, 20, 86, 100, 160, 255, 256, 257, 344, 440, 676, 1000, 1352, 2704, 5000};
    hipStream_t st[4];
    for (auto& s : st) hipStreamCreate(&s);
    int* d[4];
    for (auto& p : d) hipMalloc(&p, 8192 * 4);
    long long checked = 0, bad = 0;
    int hist[16] = {0};
    long long offs[4][8] = {{0}};
    for (int round = 0; round < 20; ++round)
        for (int n : sizes) {
            for (int k = 0; k < 4; ++k) {
                const int nn = (k == 0) ? n : sizes[(round + 3 * k + n) % 18];
                if ((round + k) & 1) hipLaunchKernelGGL(probe<72>, dim3(nn), dim3(256), 0, st[k], d[k], 300 + 100 * k);
                else hipLaunchKernelGGL(probe<8>, dim3(nn), dim3(256), 0, st[k], d[k], 200 + 150 * k);
            }
            hipDeviceSynchronize();
            for (int k = 0; k < 4; ++k) {        // every stream's launch is checked: the rotation may differ per hardware queue
                const int nn = (k == 0) ? n : sizes[(round + 3 * k + n) % 18];
                std::vector<int> h(nn);
                hipMemcpy(h.data(), d[k], nn * 4, hipMemcpyDeviceToHost);
                const int off0 = ((h[0] % 8) + 8) % 8;
                ++offs[k][off0];
                for (int b = 0; b < nn; ++b) {
                    ++checked; ++hist[h[b] & 15];
                    if ((((h[b] - b) % 8 + 8) % 8) != off0) { if (bad < 10) printf("stream %d grid %d: block 0 on XCC %d but block %d on XCC %d\n", k, nn, h[0], b, h[b]); ++bad; }
                }
            }
        }
    for (int k = 0; k < 4; ++k) {
        printf("stream %d: launches by XCC of block 0:", k);
        for (int i = 0; i < 8; ++i) printf(" %lld", offs[k][i]);
        printf("\n");
    }
    printf("checked %lld blocks (4 streams in flight): %lld break the round robin inside their launch; histogram", checked, bad);
    for (int i = 0; i < 8; ++i) printf(" %d", hist[i]);
    printf("\n");
    return bad != 0;
}
