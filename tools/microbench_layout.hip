// Micro-benchmark (development tool): the memory side of iw_pcgIter2 with different vector LAYOUTS and ADDRESSING, no arithmetic.
// Same grid as the real kernel at 4096^2: 768-thread workgroups, 720-pixel strips (12 waves x 60 output pixels, 4 overlap lanes per
// wave), 6 x 42 workgroups marching 98 rows each, three row buffers in flight, one workgroup barrier per three rows.
//   LAYOUT 0  planar, what the solver vectors look like now: r = [O.x O.y] x N + [a] x N (8 B + 4 B streams), same for p, delta
//   LAYOUT 1  one 24 B record per pixel {r.ox r.oy r.a p.ox p.oy p.a} (dwordx4 + dwordx2), delta 12 B records
//   LAYOUT 2  12 B records per vector: r, p, delta each one stream of dwordx3
//   EVEN      the launch also carries the paired delta update: + delta in/out, + p_{k-2} in
//   BUF       buffer loads/stores with the row base in an SGPR offset (no per-row 64-bit VALU address arithmetic)
// Reports the time per launch and GB/s of IDEAL bytes (53 B/px odd, 89 B/px even: what bench.py's byte model charges).
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/mb_layout tools/microbench_layout.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef float F2 __attribute__((ext_vector_type(2)));
typedef float F3 __attribute__((ext_vector_type(3)));
typedef float F4 __attribute__((ext_vector_type(4)));
typedef unsigned U2 __attribute__((ext_vector_type(2)));
typedef unsigned U3 __attribute__((ext_vector_type(3)));
typedef unsigned U4 __attribute__((ext_vector_type(4)));
constexpr int W = 4096, H = 4096, BLOCK = 768, SPAN = 60, STRIP = (BLOCK / 64) * SPAN;
constexpr long N = (long)W * H;

struct Bufs {
    const float* rIn; const float* pIn; float* rOut; float* pOut; float* delta;     // planar: 3N floats each ([2N | N]); records: 3N floats AoS
    const float* rpIn; float* rpOut;                                                // 24 B records
    const float* angle; const uint8_t* flags;
};
struct Row { float v[6]; float ang; int f; float d[3]; float q[3]; };

template <int LAYOUT, bool EVEN, bool NT, bool BUF>
__device__ __forceinline__ Row loadRow(const Bufs& B, int x, int y, __amdgpu_buffer_rsrc_t rR, __amdgpu_buffer_rsrc_t rP, __amdgpu_buffer_rsrc_t rRP,
                                       __amdgpu_buffer_rsrc_t rA, __amdgpu_buffer_rsrc_t rF) {
    Row r;
    const int yc = min(max(y, 0), H - 1), xc = min(max(x, 0), W - 1);
    const long i = (long)yc * W + xc;
    constexpr int aux = NT ? 2 : 0;
    if (BUF) {
        const int rowPx = yc * W;       // uniform -> SGPR
        if (LAYOUT == 0) {
            U2 a = __builtin_bit_cast(U2, __builtin_amdgcn_raw_buffer_load_b64(rR, xc * 8, rowPx * 8, aux));
            unsigned b = __builtin_amdgcn_raw_buffer_load_b32(rR, xc * 4, (int)(2 * N * 4 + (long)rowPx * 4), aux);
            U2 c = __builtin_bit_cast(U2, __builtin_amdgcn_raw_buffer_load_b64(rP, xc * 8, rowPx * 8, aux));
            unsigned d = __builtin_amdgcn_raw_buffer_load_b32(rP, xc * 4, (int)(2 * N * 4 + (long)rowPx * 4), aux);
            r.v[0] = __uint_as_float(a.x); r.v[1] = __uint_as_float(a.y); r.v[2] = __uint_as_float(b);
            r.v[3] = __uint_as_float(c.x); r.v[4] = __uint_as_float(c.y); r.v[5] = __uint_as_float(d);
        } else if (LAYOUT == 1) {
            U4 a = __builtin_bit_cast(U4, __builtin_amdgcn_raw_buffer_load_b128(rRP, xc * 24, rowPx * 24, aux));
            U2 b = __builtin_bit_cast(U2, __builtin_amdgcn_raw_buffer_load_b64(rRP, xc * 24 + 16, rowPx * 24, aux));
            r.v[0] = __uint_as_float(a.x); r.v[1] = __uint_as_float(a.y); r.v[2] = __uint_as_float(a.z); r.v[3] = __uint_as_float(a.w);
            r.v[4] = __uint_as_float(b.x); r.v[5] = __uint_as_float(b.y);
        } else {
            U3 a = __builtin_bit_cast(U3, __builtin_amdgcn_raw_buffer_load_b96(rR, xc * 12, rowPx * 12, aux));
            U3 b = __builtin_bit_cast(U3, __builtin_amdgcn_raw_buffer_load_b96(rP, xc * 12, rowPx * 12, aux));
            r.v[0] = __uint_as_float(a.x); r.v[1] = __uint_as_float(a.y); r.v[2] = __uint_as_float(a.z);
            r.v[3] = __uint_as_float(b.x); r.v[4] = __uint_as_float(b.y); r.v[5] = __uint_as_float(b.z);
        }
        r.ang = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rA, xc * 4, rowPx * 4, aux));
        r.f = __builtin_amdgcn_raw_buffer_load_b8(rF, xc, rowPx, aux);
    } else {
        if (LAYOUT == 0) {
            const F2 a = NT ? __builtin_nontemporal_load((const F2*)B.rIn + i) : ((const F2*)B.rIn)[i];
            const float b = NT ? __builtin_nontemporal_load(B.rIn + 2 * N + i) : B.rIn[2 * N + i];
            const F2 c = NT ? __builtin_nontemporal_load((const F2*)B.pIn + i) : ((const F2*)B.pIn)[i];
            const float d = NT ? __builtin_nontemporal_load(B.pIn + 2 * N + i) : B.pIn[2 * N + i];
            r.v[0] = a.x; r.v[1] = a.y; r.v[2] = b; r.v[3] = c.x; r.v[4] = c.y; r.v[5] = d;
        } else if (LAYOUT == 1) {
            const F2* p = (const F2*)(B.rpIn + 6 * i);
            const F2 a = NT ? __builtin_nontemporal_load(p) : p[0], b = NT ? __builtin_nontemporal_load(p + 1) : p[1], c = NT ? __builtin_nontemporal_load(p + 2) : p[2];
            r.v[0] = a.x; r.v[1] = a.y; r.v[2] = b.x; r.v[3] = b.y; r.v[4] = c.x; r.v[5] = c.y;
        } else {
            const float* pr = B.rIn + 3 * i; const float* pp = B.pIn + 3 * i;
            for (int k = 0; k < 3; ++k) { r.v[k] = NT ? __builtin_nontemporal_load(pr + k) : pr[k]; r.v[3 + k] = NT ? __builtin_nontemporal_load(pp + k) : pp[k]; }
        }
        r.ang = NT ? __builtin_nontemporal_load(B.angle + i) : B.angle[i];
        r.f = B.flags[i];
    }
    r.d[0] = r.d[1] = r.d[2] = r.q[0] = r.q[1] = r.q[2] = 0;
    return r;
}

template <int LAYOUT, bool EVEN, bool NT, bool BUF, bool OVERLAP>
__global__ __launch_bounds__(BLOCK) void k(Bufs B, int rowsPerGroup, int gx, float* sink) {
    const int bx = blockIdx.x % gx, by = blockIdx.x / gx;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int span = OVERLAP ? SPAN : 64, strip = (BLOCK / 64) * span;
    const int x = bx * strip + wave * span + lane - (OVERLAP ? 2 : 0);
    const bool writer = x >= 0 && x < W && (!OVERLAP || (lane >= 2 && lane < 62));
    const int yb = by * rowsPerGroup, ye = min(yb + rowsPerGroup, H);
    __amdgpu_buffer_rsrc_t rR = __builtin_amdgcn_make_buffer_rsrc((void*)B.rIn, 0, (int)(N * 12), 0x00020000), rP = __builtin_amdgcn_make_buffer_rsrc((void*)B.pIn, 0, (int)(N * 12), 0x00020000),
                           rRP = __builtin_amdgcn_make_buffer_rsrc((void*)B.rpIn, 0, (int)(N * 24), 0x00020000), rA = __builtin_amdgcn_make_buffer_rsrc((void*)B.angle, 0, (int)(N * 4), 0x00020000),
                           rF = __builtin_amdgcn_make_buffer_rsrc((void*)B.flags, 0, (int)N, 0x00020000);
    __amdgpu_buffer_rsrc_t wR = __builtin_amdgcn_make_buffer_rsrc((void*)B.rOut, 0, (int)(N * 12), 0x00020000), wP = __builtin_amdgcn_make_buffer_rsrc((void*)B.pOut, 0, (int)(N * 12), 0x00020000),
                           wRP = __builtin_amdgcn_make_buffer_rsrc((void*)B.rpOut, 0, (int)(N * 24), 0x00020000), wD = __builtin_amdgcn_make_buffer_rsrc((void*)B.delta, 0, (int)(N * 12), 0x00020000);
    float acc = 0;
    auto consume = [&](int y, const Row& r, bool live) {
        float s = r.ang + (float)r.f;
        for (int k2 = 0; k2 < 6; ++k2) s += r.v[k2];
        acc += s;
        if (!(writer && live)) return;
        const long i = (long)y * W + x;
        const int rowPx = y * W;
        float o[6];
        for (int k2 = 0; k2 < 6; ++k2) o[k2] = r.v[k2] * 0.5f + s;
        if (EVEN) {     // delta += a2 p_{k-2} + a1 p_{k-1}: reads delta and p_{k-2} (the p buffer about to be overwritten), writes delta
            float d[3], q[3];
            if (LAYOUT == 0) {
                const F2 dd = ((const F2*)B.delta)[i]; d[0] = dd.x; d[1] = dd.y; d[2] = B.delta[2 * N + i];
                const F2 qq = ((const F2*)B.pOut)[i]; q[0] = qq.x; q[1] = qq.y; q[2] = B.pOut[2 * N + i];
            } else if (LAYOUT == 1) {
                if (BUF) {
                    U3 dd = __builtin_bit_cast(U3, __builtin_amdgcn_raw_buffer_load_b96(wD, x * 12, rowPx * 12, 0));
                    U3 qq = __builtin_bit_cast(U3, __builtin_amdgcn_raw_buffer_load_b96(wRP, x * 24 + 12, rowPx * 24, 0));
                    d[0] = __uint_as_float(dd.x); d[1] = __uint_as_float(dd.y); d[2] = __uint_as_float(dd.z); q[0] = __uint_as_float(qq.x); q[1] = __uint_as_float(qq.y); q[2] = __uint_as_float(qq.z);
                } else for (int k2 = 0; k2 < 3; ++k2) { d[k2] = B.delta[3 * i + k2]; q[k2] = B.rpOut[6 * i + 3 + k2]; }
            } else {
                if (BUF) {
                    U3 dd = __builtin_bit_cast(U3, __builtin_amdgcn_raw_buffer_load_b96(wD, x * 12, rowPx * 12, 0));
                    U3 qq = __builtin_bit_cast(U3, __builtin_amdgcn_raw_buffer_load_b96(wP, x * 12, rowPx * 12, 0));
                    d[0] = __uint_as_float(dd.x); d[1] = __uint_as_float(dd.y); d[2] = __uint_as_float(dd.z); q[0] = __uint_as_float(qq.x); q[1] = __uint_as_float(qq.y); q[2] = __uint_as_float(qq.z);
                } else for (int k2 = 0; k2 < 3; ++k2) { d[k2] = B.delta[3 * i + k2]; q[k2] = B.pOut[3 * i + k2]; }
            }
            for (int k2 = 0; k2 < 3; ++k2) d[k2] += 0.25f * q[k2] + 0.125f * r.v[3 + k2];
            if (LAYOUT == 0) { ((F2*)B.delta)[i] = F2{d[0], d[1]}; B.delta[2 * N + i] = d[2]; }
            else if (BUF) { U3 dd = {__float_as_uint(d[0]), __float_as_uint(d[1]), __float_as_uint(d[2])}; __builtin_amdgcn_raw_buffer_store_b96(dd, wD, x * 12, rowPx * 12, 0); }
            else for (int k2 = 0; k2 < 3; ++k2) B.delta[3 * i + k2] = d[k2];
        }
        if (LAYOUT == 0) {
            ((F2*)B.rOut)[i] = F2{o[0], o[1]}; B.rOut[2 * N + i] = o[2]; ((F2*)B.pOut)[i] = F2{o[3], o[4]}; B.pOut[2 * N + i] = o[5];
        } else if (LAYOUT == 1) {
            if (BUF) {
                U4 a = {__float_as_uint(o[0]), __float_as_uint(o[1]), __float_as_uint(o[2]), __float_as_uint(o[3])}; U2 b = {__float_as_uint(o[4]), __float_as_uint(o[5])};
                __builtin_amdgcn_raw_buffer_store_b128(a, wRP, x * 24, rowPx * 24, 0); __builtin_amdgcn_raw_buffer_store_b64(b, wRP, x * 24 + 16, rowPx * 24, 0);
            } else { F2* p = (F2*)(B.rpOut + 6 * i); p[0] = F2{o[0], o[1]}; p[1] = F2{o[2], o[3]}; p[2] = F2{o[4], o[5]}; }
        } else {
            if (BUF) {
                U3 a = {__float_as_uint(o[0]), __float_as_uint(o[1]), __float_as_uint(o[2])}, b = {__float_as_uint(o[3]), __float_as_uint(o[4]), __float_as_uint(o[5])};
                __builtin_amdgcn_raw_buffer_store_b96(a, wR, x * 12, rowPx * 12, 0); __builtin_amdgcn_raw_buffer_store_b96(b, wP, x * 12, rowPx * 12, 0);
            } else for (int k2 = 0; k2 < 3; ++k2) { B.rOut[3 * i + k2] = o[k2]; B.pOut[3 * i + k2] = o[3 + k2]; }
        }
    };
#define LD(yy) loadRow<LAYOUT, EVEN, NT, BUF>(B, x, yy, rR, rP, rRP, rA, rF)
    Row a = LD(yb - 2), b = LD(yb - 1), c = LD(yb);
    for (int y = yb - 2; y < ye; y += 3) {           // like the kernel: 2 halo rows above, 2 below come with the prefetch
        __syncthreads();
        { const Row w = a; a = LD(y + 3); consume(y, w, y >= yb); }
        { const Row w = b; b = LD(y + 4); consume(y + 1, w, y + 1 >= yb && y + 1 < ye); }
        { const Row w = c; c = LD(y + 5); consume(y + 2, w, y + 2 >= yb && y + 2 < ye); }
    }
#undef LD
    if (acc == 12345.678f) sink[0] = acc;
}

int main() {
    Bufs B;
    void* p[10];
    const size_t sz[10] = {(size_t)N * 12, (size_t)N * 12, (size_t)N * 12, (size_t)N * 12, (size_t)N * 12, (size_t)N * 24, (size_t)N * 24, (size_t)N * 4, (size_t)N, 64};
    for (int i = 0; i < 10; ++i) { CK(hipMalloc(&p[i], sz[i])); CK(hipMemset(p[i], 0, sz[i])); }
    B.rIn = (const float*)p[0]; B.pIn = (const float*)p[1]; B.rOut = (float*)p[2]; B.pOut = (float*)p[3]; B.delta = (float*)p[4];
    B.rpIn = (const float*)p[5]; B.rpOut = (float*)p[6]; B.angle = (const float*)p[7]; B.flags = (const uint8_t*)p[8];
    float* sink = (float*)p[9];
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto time = [&](auto launch) { launch(); launch(); CK(hipDeviceSynchronize()); CK(hipEventRecord(e0)); for (int i = 0; i < 20; ++i) launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); CK(hipGetLastError()); return ms / 20; };
    const int gxO = (W + STRIP - 1) / STRIP, gxN = (W + 767) / 768;
    const int gy = 42, rpg = (H + gy - 1) / gy;
    auto report = [&](const char* name, bool even, float ms) { const double gb = (even ? 89.0 : 53.0) * N / 1e9; printf("%-58s %s  %7.1f us  %6.0f GB/s (ideal bytes)\n", name, even ? "even" : "odd ", ms * 1e3, gb / ms * 1e3); };
#define RUN(L, E, NTV, BF, OV, name) report(name, E, time([&] { k<L, E, NTV, BF, OV><<<(OV ? gxO : gxN) * gy, BLOCK>>>(B, rpg, OV ? gxO : gxN, sink); }))
    for (int rep = 0; rep < 2; ++rep) {
        RUN(0, false, true, false, true, "planar      nt  global-ptr  overlap (= current kernel)");
        RUN(0, true, true, false, true, "planar      nt  global-ptr  overlap (= current kernel)");
        RUN(0, false, false, false, true, "planar      --  global-ptr  overlap");
        RUN(0, false, true, true, true, "planar      nt  buffer      overlap");
        RUN(0, true, true, true, true, "planar      nt  buffer      overlap");
        RUN(0, false, true, false, false, "planar      nt  global-ptr  no overlap lanes");
        RUN(1, false, true, false, true, "rec24       nt  global-ptr  overlap");
        RUN(1, true, true, false, true, "rec24       nt  global-ptr  overlap");
        RUN(1, false, true, true, true, "rec24       nt  buffer      overlap");
        RUN(1, true, true, true, true, "rec24       nt  buffer      overlap");
        RUN(1, false, false, true, true, "rec24       --  buffer      overlap");
        RUN(1, true, false, true, true, "rec24       --  buffer      overlap");
        RUN(1, false, true, true, false, "rec24       nt  buffer      no overlap lanes");
        RUN(2, false, true, true, true, "rec12 x 2   nt  buffer      overlap");
        RUN(2, true, true, true, true, "rec12 x 2   nt  buffer      overlap");
        RUN(2, false, false, true, true, "rec12 x 2   --  buffer      overlap");
        RUN(2, true, false, true, true, "rec12 x 2   --  buffer      overlap");
        printf("\n");
    }
    return 0;
}
