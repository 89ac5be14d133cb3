// Box probes of libOpt.so (include/OptAmd.h): nothing here is on the solver's path.
//  * OptAmd_MeasureCopyBandwidth: the float4 copy ceiling of THIS box (the roofline's second denominator: MI355X_MICROARCH.md measures 6.29 TB/s this way, boxes of
//    the pool differ) -- a grid-stride 16-byte-per-lane copy kernel, default or nontemporal accesses, timed with hipEvents on its own stream.
//  * OptAmd_DebugOccupy: a test hook that holds CUs the way a foreign tenant would (tests/test_coresidency_gpu.py): `workgroups` single-wave workgroups spin on the
//    device's wall clock for `milliseconds`, each holding 150 KB of its CU's LDS; while they run, no workgroup of a persistent solver kernel fits on those CUs.
#include "common.h"
#include "../../include/OptAmd.h"

namespace optamd {
namespace {
typedef float probe_f4 __attribute__((ext_vector_type(4)));

template <bool NT> __global__ __launch_bounds__(256) void k_probeCopy(const probe_f4* __restrict__ a, probe_f4* __restrict__ b, long n) {
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += gridDim.x * 256L) {
        const probe_f4 v = NT ? __builtin_nontemporal_load(a + i) : a[i];
        if (NT) __builtin_nontemporal_store(v, b + i); else b[i] = v;
    }
}
__global__ void k_probeOccupy(long long ticks, int* sink) {
    const long long t0 = wall_clock64();
    int spins = 0;
    while (wall_clock64() - t0 < ticks) { __builtin_amdgcn_s_sleep(8); ++spins; }
    if (sink && spins < 0) sink[0] = spins;
}
}  // namespace
}  // namespace optamd

using namespace optamd;

extern "C" double OptAmd_MeasureCopyBandwidth(long bytes, int nontemporal, int reps) {
    if (bytes < (1 << 20) || reps < 1) return 0.0;
    const long n = bytes / 2 / 16;      // half in, half out
    probe_f4 *a = nullptr, *b = nullptr;
    if (hipMalloc((void**)&a, n * 16) != hipSuccess) return 0.0;
    if (hipMalloc((void**)&b, n * 16) != hipSuccess) { (void)hipFree(a); return 0.0; }
    hipStream_t s; hipEvent_t e0, e1;
    HIP_CHECK(hipStreamCreate(&s)); HIP_CHECK(hipEventCreate(&e0)); HIP_CHECK(hipEventCreate(&e1));
    HIP_CHECK(hipMemsetAsync(a, 1, n * 16, s)); HIP_CHECK(hipMemsetAsync(b, 0, n * 16, s));
    int dev = 0, cus = 256; HIP_CHECK(hipGetDevice(&dev)); HIP_CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    const int grid = cus * 8;
    auto run = [&] { if (nontemporal) k_probeCopy<true><<<grid, 256, 0, s>>>(a, b, n); else k_probeCopy<false><<<grid, 256, 0, s>>>(a, b, n); };
    for (int i = 0; i < 3; ++i) run();
    HIP_CHECK(hipEventRecord(e0, s));
    for (int i = 0; i < reps; ++i) run();
    HIP_CHECK(hipEventRecord(e1, s)); HIP_CHECK(hipEventSynchronize(e1));
    float ms = 0; HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
    (void)hipFree(a); (void)hipFree(b); (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); (void)hipStreamDestroy(s);
    return ms > 0 ? 2.0 * n * 16 * reps / (ms * 1e-3) / 1e9 : 0.0;
}

extern "C" int OptAmd_DebugOccupy(int workgroups, double milliseconds, void* stream) {
    if (workgroups < 1 || milliseconds <= 0 || milliseconds > 5000.0) return 0;
    // each workgroup is one wave that holds 150 of its CU's 160 KB of LDS: at most one per CU, and no workgroup of a persistent solver kernel (tens of KB of LDS each)
    // fits beside it -- a small-register tenant alone would share the CU with the 105-VGPR variants
    constexpr int kLds = 150 * 1024;
    static bool once = false;
    if (!once) { if (hipFuncSetAttribute((const void*)k_probeOccupy, hipFuncAttributeMaxDynamicSharedMemorySize, kLds) != hipSuccess) { (void)hipGetLastError(); return 0; } once = true; }
    k_probeOccupy<<<workgroups, kWave, kLds, (hipStream_t)stream>>>((long long)(milliseconds * 1e5), nullptr);
    return hipGetLastError() == hipSuccess ? 1 : 0;
}
