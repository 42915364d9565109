// What limits resident workgroups per CU on this chip?  hipOccupancyMaxActiveBlocksPerMultiprocessor for
// synthetic kernels (threads, static LDS bytes, live VGPRs), plus a measured residency: every workgroup
// records its start time and spins ~20 us, so the number of workgroups started in the first 5 us is
// what was resident at once.   hipcc --offload-arch=gfx950 -O3 tools/ubench_occupancy.hip -o /tmp/occ
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
template <int THREADS, int LDS, int REGS>
__global__ __launch_bounds__(THREADS) void probe(unsigned long long *t, float *sink) {
    __shared__ char lds[LDS];
    float r[REGS];
    for (int i = 0; i < REGS; i++) r[i] = threadIdx.x * 0.5f + i;
    const unsigned long long t0 = wall_clock64();
    if (threadIdx.x == 0) t[blockIdx.x] = t0;
    lds[threadIdx.x % LDS] = (char)threadIdx.x;
    __syncthreads();
    while (wall_clock64() - t0 < 2000) {                 // 100 MHz clock: 20 us
        for (int i = 0; i < REGS; i++) r[i] = r[i] * 1.0001f + lds[(threadIdx.x + i) % LDS];
    }
    float s = 0;
    for (int i = 0; i < REGS; i++) s += r[i];
    if (s == 12345.f) sink[0] = s;
}
template <int THREADS, int LDS, int REGS>
void run(const char *name, int cus) {
    int per = 0;
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&per, probe<THREADS, LDS, REGS>, THREADS, 0);
    hipFuncAttributes fa;
    hipFuncGetAttributes(&fa, reinterpret_cast<const void *>(probe<THREADS, LDS, REGS>));
    const int grid = cus * 12;
    unsigned long long *t; float *sink;
    hipMalloc(&t, grid * 8); hipMalloc(&sink, 4);
    hipLaunchKernelGGL((probe<THREADS, LDS, REGS>), dim3(grid), dim3(THREADS), 0, 0, t, sink);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(grid);
    hipMemcpy(h.data(), t, grid * 8, hipMemcpyDeviceToHost);
    const unsigned long long t0 = *std::min_element(h.begin(), h.end());
    int early = 0;
    for (auto v : h) early += (v - t0) < 500;
    printf("%-28s threads %4d lds %6d regs %3d (numRegs %3d)  api %2d/CU   measured %.2f WGs/CU = %.1f waves/SIMD\n", name, THREADS, LDS,
           REGS, fa.numRegs, per, early / (double)cus, early / (double)cus * THREADS / 256.0);
    hipFree(t); hipFree(sink);
}
int main() {
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    printf("%s CUs %d  sharedMemPerMultiprocessor %zu  maxThreadsPerMultiProcessor %d regsPerMultiprocessor %d\n", p.gcnArchName,
           p.multiProcessorCount, p.sharedMemPerMultiprocessor, p.maxThreadsPerMultiProcessor, p.regsPerMultiprocessor);
    const int cus = p.multiProcessorCount;
    run<512, 1024, 8>("512t small", cus);
    run<512, 36000, 8>("512t 36KB", cus);
    run<512, 36000, 40>("512t 36KB 40regs", cus);
    run<512, 20000, 40>("512t 20KB 40regs", cus);
    run<512, 16000, 40>("512t 16KB 40regs", cus);
    run<256, 36000, 40>("256t 36KB 40regs", cus);
    run<256, 18000, 40>("256t 18KB 40regs", cus);
    run<256, 1024, 40>("256t 1KB 40regs", cus);
    run<256, 1024, 100>("256t 1KB 100regs", cus);
    run<64, 1024, 40>("64t 1KB 40regs", cus);
    run<1024, 1024, 40>("1024t 1KB 40regs", cus);
    return 0;
}
