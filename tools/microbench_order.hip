// Micro-benchmark (development tool): does the ORDER in which a many-stream kernel walks memory matter on MI355X?
// 7 read + 6 write float2 streams over a 4096x4096 image (the shape of iw_pcgIter), same bytes in every mode:
//   MODE 0  flat grid-stride: the whole chip moves through memory as one dense front
//   MODE 1  strip marching: co-resident grid, each 512-thread workgroup walks down the rows of its own 512-pixel-wide strip
//           (what iw_pcgIter does: 252 workgroups x 13 streams = thousands of separate 4 KB-at-a-time walks)
//   MODE 2  row fronts: workgroup b takes rows b, b+G, b+2G, ... whole rows (32 KB contiguous per stream), so the chip works on G adjacent rows
//   MODE 3  strip marching, but consecutive workgroups take adjacent strips of the SAME rows and all march in step (short row groups, in order)
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/mb_order tools/microbench_order.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef float V __attribute__((ext_vector_type(2)));
constexpr int NIN = 7, NOUT = 6, W = 4096, H = 4096;
struct Ptrs { const V* in[NIN]; V* out[NOUT]; };

__device__ __forceinline__ void px(const Ptrs& P, long i) {
    V acc = __builtin_nontemporal_load(P.in[0] + i);
#pragma unroll
    for (int s = 1; s < NIN; ++s) acc += __builtin_nontemporal_load(P.in[s] + i);
#pragma unroll
    for (int s = 0; s < NOUT; ++s) P.out[s][i] = acc * (float)(s + 1);
}
template <int MODE>
__global__ __launch_bounds__(512) void k(Ptrs P, int rowsPerGroup) {
    if (MODE == 0) {
        for (long i = blockIdx.x * 512L + threadIdx.x; i < (long)W * H; i += gridDim.x * 512L) px(P, i);
    } else if (MODE == 1 || MODE == 3) {
        const int gx = W / 512, bx = blockIdx.x % gx, by = blockIdx.x / gx;
        const int x = bx * 512 + threadIdx.x, yb = by * rowsPerGroup, ye = min(yb + rowsPerGroup, H);
        for (int y = yb; y < ye; ++y) px(P, (long)y * W + x);
    } else {
        for (int y = blockIdx.x; y < H; y += gridDim.x)
            for (int x = threadIdx.x; x < W; x += 512) px(P, (long)y * W + x);
    }
}
int main() {
    const long bytes = (long)W * H * sizeof(V);
    Ptrs P;
    for (int i = 0; i < NIN; ++i) { void* p; CK(hipMalloc(&p, bytes)); CK(hipMemset(p, 1, bytes)); P.in[i] = (const V*)p; }
    for (int i = 0; i < NOUT; ++i) { void* p; CK(hipMalloc(&p, bytes)); P.out[i] = (V*)p; }
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto time = [&](auto launch) { launch(); launch(); CK(hipDeviceSynchronize()); CK(hipEventRecord(e0)); for (int i = 0; i < 10; ++i) launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); return ms / 10; };
    const double gb = (double)(NIN + NOUT) * bytes / 1e9;
    for (int rep = 0; rep < 2; ++rep) {
        printf("flat grid-stride   (2048 wg): %.0f GB/s\n", gb / time([&] { k<0><<<2048, 512>>>(P, 0); }) * 1e3);
        printf("flat grid-stride   ( 256 wg): %.0f GB/s\n", gb / time([&] { k<0><<<256, 512>>>(P, 0); }) * 1e3);
        printf("strip marching     ( 256 wg x 128 rows): %.0f GB/s\n", gb / time([&] { k<1><<<8 * 32, 512>>>(P, 128); }) * 1e3);
        printf("strip marching     ( 512 wg x  64 rows): %.0f GB/s\n", gb / time([&] { k<1><<<8 * 64, 512>>>(P, 64); }) * 1e3);
        printf("strip, short groups(4096 wg x   8 rows): %.0f GB/s\n", gb / time([&] { k<3><<<8 * 512, 512>>>(P, 8); }) * 1e3);
        printf("strip, short groups(16384 wg x  2 rows): %.0f GB/s\n", gb / time([&] { k<3><<<8 * 2048, 512>>>(P, 2); }) * 1e3);
        printf("row fronts         ( 256 wg): %.0f GB/s\n", gb / time([&] { k<2><<<256, 512>>>(P, 0); }) * 1e3);
        printf("row fronts         ( 512 wg): %.0f GB/s\n", gb / time([&] { k<2><<<512, 512>>>(P, 0); }) * 1e3);
    }
    return 0;
}
