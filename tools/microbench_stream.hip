// Micro-benchmark (development tool, not part of the library): how fast can a PCGStep2-shaped streaming
// kernel (5 read streams, 3 write streams, 16-byte accesses) run on this GPU, and which launch shape /
// cache hints get closest to the float4-copy ceiling?  Build: hipcc --offload-arch=gfx950 -O3 -o mb tools/microbench_stream.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

typedef float float4_ __attribute__((ext_vector_type(4)));

template <bool NT> __device__ __forceinline__ float4_ ld(const float4_* p) { return NT ? __builtin_nontemporal_load(p) : *p; }
template <bool NT> __device__ __forceinline__ void st(float4_* p, float4_ v) { if (NT) __builtin_nontemporal_store(v, p); else *p = v; }

__global__ __launch_bounds__(256) void k_copy(const float4_* __restrict__ a, float4_* __restrict__ b, long n) {
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += gridDim.x * 256L) b[i] = a[i];
}

// MODE 0: grid-stride; MODE 1: block-contiguous chunks; UNROLL packs per thread per trip
template <bool NTL, bool NTS, int MODE, int UNROLL>
__global__ __launch_bounds__(256) void k_step2(float4_* __restrict__ delta, const float4_* __restrict__ p, float4_* __restrict__ r, const float4_* __restrict__ Ap,
                                               const float4_* __restrict__ pre, float4_* __restrict__ z, long n, float alpha, double* __restrict__ partials) {
    double acc = 0;
    long begin, end, stride;
    if (MODE == 0) { begin = (blockIdx.x * 256L + threadIdx.x); end = n; stride = gridDim.x * 256L; }
    else { long per = (n + gridDim.x - 1) / gridDim.x; per = (per + 255) / 256 * 256; begin = blockIdx.x * per + threadIdx.x; end = min(n, (blockIdx.x + 1) * per); stride = 256; }
    for (long i = begin; i < end; i += stride * UNROLL) {
        float4_ D[UNROLL], P[UNROLL], R[UNROLL], A[UNROLL], M[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) { long j = i + u * stride; if (j < end) { D[u] = ld<NTL>(delta + j); P[u] = ld<NTL>(p + j); R[u] = ld<NTL>(r + j); A[u] = ld<NTL>(Ap + j); M[u] = ld<NTL>(pre + j); } }
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            long j = i + u * stride;
            if (j < end) {
                float4_ d = D[u] + alpha * P[u], rr = R[u] - alpha * A[u], zz = M[u] * rr;
                st<NTS>(delta + j, d); st<NTS>(r + j, rr); st<NTS>(z + j, zz);
                acc += (double)(zz.x * rr.x) + (double)(zz.y * rr.y) + (double)(zz.z * rr.z) + (double)(zz.w * rr.w);
            }
        }
    }
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o, 64);
    __shared__ double s[4];
    if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partials[blockIdx.x] = s[0] + s[1] + s[2] + s[3];
}

template <class F> float timeit(F f, int reps = 20) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    f(); f(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(a)); for (int i = 0; i < reps; ++i) f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); return ms / reps;
}

int main(int argc, char** argv) {
    long npx = 4096L * 4096L; long n = npx * 3 / 4;   // float4 packs in one 3-channel vector
    float4_* v[8];
    for (int i = 0; i < 8; ++i) { CK(hipMalloc(&v[i], n * 16)); CK(hipMemset(v[i], 0, n * 16)); }
    double* part; CK(hipMalloc(&part, 65536 * 8));
    double gb_copy = 2.0 * n * 16 / 1e9, gb_s2 = 8.0 * n * 16 / 1e9;
    for (int g : {1024, 2048, 4096, 8192}) {
        float ms = timeit([&] { k_copy<<<g, 256>>>(v[0], v[1], n); });
        printf("copy            grid %5d : %7.1f us  %6.0f GB/s\n", g, ms * 1e3, gb_copy / ms * 1e3);
    }
#define RUN(NTL, NTS, MODE, UNR, G) { float ms = timeit([&] { k_step2<NTL, NTS, MODE, UNR><<<G, 256>>>(v[0], v[1], v[2], v[3], v[4], v[5], n, 0.5f, part); }); \
        printf("step2 ntl%d nts%d mode%d unr%d grid %5d : %7.1f us  %6.0f GB/s\n", NTL, NTS, MODE, UNR, G, ms * 1e3, gb_s2 / ms * 1e3); }
    RUN(false, false, 0, 1, 1024) RUN(false, false, 0, 1, 2048) RUN(false, false, 0, 1, 4096) RUN(false, false, 0, 1, 8192) RUN(false, false, 0, 1, 49152)
    RUN(false, true, 0, 1, 2048) RUN(true, true, 0, 1, 2048) RUN(true, false, 0, 1, 2048)
    RUN(false, false, 0, 2, 2048) RUN(false, false, 0, 2, 1024) RUN(false, false, 0, 4, 1024) RUN(true, true, 0, 2, 2048)
    RUN(false, false, 1, 1, 2048) RUN(false, false, 1, 1, 1024) RUN(false, false, 1, 2, 2048) RUN(true, true, 1, 2, 2048) RUN(false, false, 1, 1, 4096)
    return 0;
}
