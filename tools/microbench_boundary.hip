// Micro-benchmark (development tool): what does a dependent kernel boundary cost after a kernel that streams
// hundreds of MB of stores, by store flavour (plain / nt / sc0 sc1 write-through)?
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/mb_boundary tools/microbench_boundary.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));

template <int ST> __device__ __forceinline__ void st(f4* p, f4 v) {
    if (ST == 0) *p = v;
    else if (ST == 1) __builtin_nontemporal_store(v, p);
    else if (ST == 2) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
    else asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
}
// 5 read streams, 3 write streams (PCGStep2 shape)
template <int ST>
__global__ __launch_bounds__(256) void k(f4* __restrict__ a, const f4* __restrict__ b, f4* __restrict__ c, const f4* __restrict__ d, const f4* __restrict__ e, f4* __restrict__ z, long n, float s) {
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += gridDim.x * 256L) {
        f4 A = a[i], B = b[i], C = c[i], D = d[i], E = e[i];
        st<ST>(a + i, A + s * B); f4 r = C - s * D; st<ST>(c + i, r); st<ST>(z + i, E * r);
    }
}
int main() {
    long n = 4096L * 4096L * 3 / 4;
    f4* v[6]; for (int i = 0; i < 6; ++i) { CK(hipMalloc(&v[i], n * 16)); CK(hipMemset(v[i], 1, n * 16)); }
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto run = [&](int st, int reps) {
        for (int i = 0; i < reps; ++i) {
            if (st == 0) k<0><<<2048, 256>>>(v[0], v[1], v[2], v[3], v[4], v[5], n, 1e-3f);
            if (st == 1) k<1><<<2048, 256>>>(v[0], v[1], v[2], v[3], v[4], v[5], n, 1e-3f);
            if (st == 2) k<2><<<2048, 256>>>(v[0], v[1], v[2], v[3], v[4], v[5], n, 1e-3f);
            if (st == 3) k<3><<<2048, 256>>>(v[0], v[1], v[2], v[3], v[4], v[5], n, 1e-3f);
        }
    };
    const char* names[4] = {"plain", "nt", "sc0 sc1", "sc1"};
    for (int pass = 0; pass < 2; ++pass)
        for (int st = 0; st < 4; ++st) {
            run(st, 3); CK(hipDeviceSynchronize());
            // (a) back-to-back dependent launches: per-launch wall = kernel + boundary
            CK(hipEventRecord(e0)); run(st, 40); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float chain; CK(hipEventElapsedTime(&chain, e0, e1)); chain /= 40;
            // (b) isolated launches: event pair around each
            float iso = 0;
            for (int i = 0; i < 10; ++i) { CK(hipDeviceSynchronize()); CK(hipEventRecord(e0)); run(st, 1); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float t; CK(hipEventElapsedTime(&t, e0, e1)); iso += t; }
            iso /= 10;
            printf("%-8s chained %.1f us/launch  isolated %.1f us  (%.0f GB/s chained)\n", names[st], chain * 1e3, iso * 1e3, 8.0 * n * 16 / 1e9 / (chain * 1e-3));
        }
    return 0;
}
