// Micro-benchmark (development tool): what does one more dependent kernel in a stream cost on MI355X, and what does a memory round trip inside it cost?
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/mb_launch tools/microbench_launch.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
__global__ void k_empty() {}
__global__ void k_chain(const double* in, double* out, int trips) {      // `trips` dependent global round trips per workgroup, then one store
    double v = in[threadIdx.x & 63];
    for (int t = 1; t < trips; ++t) v = in[((int)v + threadIdx.x + t) & 1023];
    if (threadIdx.x == 0) out[blockIdx.x] = v;
}
int main() {
    double *a, *b; CK(hipMalloc(&a, 8192 * 8)); CK(hipMalloc(&b, 8192 * 8)); CK(hipMemset(a, 0, 8192 * 8));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto time = [&](auto launch, int n) { for (int i = 0; i < 20; ++i) launch(i); CK(hipDeviceSynchronize()); CK(hipEventRecord(e0)); for (int i = 0; i < n; ++i) launch(i); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); return ms / n * 1e3; };
    printf("empty kernel, 1 workgroup x 64          : %.2f us per launch\n", time([&](int) { k_empty<<<1, 64>>>(); }, 2000));
    printf("empty kernel, 252 workgroups x 768      : %.2f us per launch\n", time([&](int) { k_empty<<<252, 768>>>(); }, 2000));
    for (int trips : {1, 2, 4, 8})
        printf("252 x 768, %d dependent global round trips: %.2f us per launch\n", trips, time([&](int i) { k_chain<<<252, 768>>>((i & 1) ? a : b, (i & 1) ? b : a, trips); }, 2000));
    return 0;
}
