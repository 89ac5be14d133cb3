// Micro-benchmark (development tool): (a) what a cooperative launch costs next to a plain one for a persistent kernel whose workgroups meet
// once (an arrival counter), per launch and back to back; (b) the float4 copy ceiling of this box on 2 GiB (default / nontemporal accesses).
// Build: hipcc --offload-arch=gfx950 -O3 -o mb_coop tools/microbench_coop.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

typedef float float4_ __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(512) void k_meet(unsigned* cnt, unsigned target, long long* out, long long bound) {
    __shared__ int ok;
    if (threadIdx.x == 0) {
        const long long t0 = wall_clock64();
        __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int good = 1;
        while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            if (wall_clock64() - t0 > bound) { good = 0; break; }
            __builtin_amdgcn_s_sleep(1);
        }
        ok = good;
        if (blockIdx.x == 0) out[0] = wall_clock64() - t0;
        if (!good) out[1] = 1;
    }
    __syncthreads();
}

template <bool NT> __global__ __launch_bounds__(256) void k_copy(const float4_* __restrict__ a, float4_* __restrict__ b, long n) {
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += gridDim.x * 256L) {
        float4_ v = NT ? __builtin_nontemporal_load(a + i) : a[i];
        if (NT) __builtin_nontemporal_store(v, b + i); else b[i] = v;
    }
}

int main() {
    int cus = 0; CK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
    int coop = 0; CK(hipDeviceGetAttribute(&coop, hipDeviceAttributeCooperativeLaunch, 0));
    printf("CUs %d cooperativeLaunch %d\n", cus, coop);
    unsigned* cnt; long long* out; CK(hipMalloc(&cnt, 4)); CK(hipMalloc(&out, 16)); CK(hipMemset(out, 0, 16));
    hipStream_t s; CK(hipStreamCreate(&s));
    const long long bound = 100000000LL;   // 1 s in 100 MHz ticks
    for (int grid : {64, 128, 256}) {
        for (int mode = 0; mode < 2; ++mode) {
            unsigned target = 0;
            auto launch = [&] {
                target += grid;
                if (mode == 0) k_meet<<<grid, 512, 0, s>>>(cnt, target, out, bound);
                else {
                    void* args[] = {(void*)&cnt, (void*)&target, (void*)&out, (void*)&bound};
                    CK(hipLaunchCooperativeKernel((const void*)k_meet, dim3(grid), dim3(512), args, 0, s));
                }
            };
            CK(hipMemsetAsync(cnt, 0, 4, s));
            for (int i = 0; i < 5; ++i) launch();
            CK(hipStreamSynchronize(s));
            // (i) one launch + synchronise, host clock
            double best = 1e9, sum = 0; const int reps = 50;
            for (int i = 0; i < reps; ++i) {
                auto t0 = std::chrono::steady_clock::now();
                launch(); CK(hipStreamSynchronize(s));
                double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
                best = us < best ? us : best; sum += us;
            }
            // (ii) 50 launches back to back
            auto t0 = std::chrono::steady_clock::now();
            for (int i = 0; i < reps; ++i) launch();
            CK(hipStreamSynchronize(s));
            double train = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / reps;
            long long h[2]; CK(hipMemcpy(h, out, 16, hipMemcpyDeviceToHost));
            printf("grid %3d %-11s launch+sync best %7.1f us avg %7.1f us | back-to-back %6.1f us/launch | wg0 meet wait %5.2f us | timeouts %lld\n", grid, mode ? "cooperative" : "plain",
                   best, sum / reps, train, h[0] / 100.0, h[1]);
        }
    }
    // (b) copy ceiling
    const long n = (1L << 30) / 16;   // 1 GiB in, 1 GiB out = 2 GiB moved
    float4_ *a, *b; CK(hipMalloc(&a, n * 16)); CK(hipMalloc(&b, n * 16)); CK(hipMemset(a, 1, n * 16)); CK(hipMemset(b, 0, n * 16));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int nt = 0; nt < 2; ++nt) for (int g : {2048, 4096, 8192, 16384}) {
        for (int w = 0; w < 3; ++w) { if (nt) k_copy<true><<<g, 256, 0, s>>>(a, b, n); else k_copy<false><<<g, 256, 0, s>>>(a, b, n); }
        CK(hipEventRecord(e0, s));
        const int reps = 20;
        for (int i = 0; i < reps; ++i) { if (nt) k_copy<true><<<g, 256, 0, s>>>(a, b, n); else k_copy<false><<<g, 256, 0, s>>>(a, b, n); }
        CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
        printf("copy float4 %s grid %5d: %7.1f us  %6.0f GB/s (2 GiB moved)\n", nt ? "nontemporal" : "default    ", g, ms * 1e3, 2.0 * n * 16 / 1e9 / ms * 1e3);
    }
    return 0;
}
