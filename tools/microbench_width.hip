// Micro-benchmark (development tool): does access width per lane limit a many-stream streaming kernel on MI355X?
// 12 read streams + 8 write streams (the shape of iw_pcgIter), each stream accessed with 4, 8 or 16 bytes per lane,
// 256-thread workgroups, grid-stride.  Build: hipcc --offload-arch=gfx950 -O3 -o tools/mb_width tools/microbench_width.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
template <int W> struct Vec;
template <> struct Vec<4> { typedef float type; };
template <> struct Vec<8> { typedef float type __attribute__((ext_vector_type(2))); };
template <> struct Vec<16> { typedef float type __attribute__((ext_vector_type(4))); };
struct Ptrs { const void* in[12]; void* out[8]; };

template <int W, int NIN, int NOUT>
__global__ __launch_bounds__(256) void k(Ptrs P, long n /* elements of width W per stream */) {
    typedef typename Vec<W>::type V;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += gridDim.x * 256L) {
        V acc = ((const V*)P.in[0])[i];
#pragma unroll
        for (int s = 1; s < NIN; ++s) acc += ((const V*)P.in[s])[i];
#pragma unroll
        for (int s = 0; s < NOUT; ++s) ((V*)P.out[s])[i] = acc * (float)(s + 1);
    }
}
int main() {
    const long bytesPerStream = 4096L * 4096L * 8;   // 134 MB per stream
    Ptrs P;
    for (int i = 0; i < 12; ++i) { void* p; CK(hipMalloc(&p, bytesPerStream)); CK(hipMemset(p, 1, bytesPerStream)); P.in[i] = p; }
    for (int i = 0; i < 8; ++i) { void* p; CK(hipMalloc(&p, bytesPerStream)); P.out[i] = p; }
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto time = [&](auto launch) { launch(); launch(); CK(hipDeviceSynchronize()); CK(hipEventRecord(e0)); for (int i = 0; i < 10; ++i) launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); return ms / 10; };
    for (int g : {1024, 2048}) {
        float t4 = time([&] { k<4, 12, 8><<<g, 256>>>(P, bytesPerStream / 4); });
        float t8 = time([&] { k<8, 12, 8><<<g, 256>>>(P, bytesPerStream / 8); });
        float t16 = time([&] { k<16, 12, 8><<<g, 256>>>(P, bytesPerStream / 16); });
        double gb = 20.0 * bytesPerStream / 1e9;
        printf("12R+8W streams grid %d:  4B/lane %.0f GB/s   8B/lane %.0f GB/s   16B/lane %.0f GB/s\n", g, gb / t4 * 1e3, gb / t8 * 1e3, gb / t16 * 1e3);
        float u4 = time([&] { k<4, 5, 3><<<g, 256>>>(P, bytesPerStream / 4); });
        float u8 = time([&] { k<8, 5, 3><<<g, 256>>>(P, bytesPerStream / 8); });
        float u16 = time([&] { k<16, 5, 3><<<g, 256>>>(P, bytesPerStream / 16); });
        gb = 8.0 * bytesPerStream / 1e9;
        printf(" 5R+3W streams grid %d:  4B/lane %.0f GB/s   8B/lane %.0f GB/s   16B/lane %.0f GB/s\n", g, gb / u4 * 1e3, gb / u8 * 1e3, gb / u16 * 1e3);
    }
    return 0;
}
