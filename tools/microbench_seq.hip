// Micro-benchmark (development tool): the memory side of the PCG iteration kernel as the SEQUENCE the solver really runs -- launch k reads
// r, p set k % 2 and writes set (k + 1) % 2, every second launch also carries the paired delta update (+ delta in / out, + p_{k-2} in), and
// successive launches sweep the image in opposite directions -- with the per-lane access width, overlap lanes, cache hints, barrier and
// workgroup shape as template parameters.  No arithmetic beyond a few adds: what is measured is what the memory system gives this
// access pattern at 4096^2.  Output: average time per launch over 40 launches and the GB/s of ideal bytes (71 B/px per launch on average).
//   PX       pixels per lane: 1 (8 B + 4 B + 1 B accesses per vector, what iw_pcgIter2 does) or 2 (16 B + 8 B + 2 B)
//   OVERLAP  waves overlap by 4 pixels (the kernel's DPP halo) or tile the row exactly
//   NTL      non-temporal loads;  SYNC  workgroup barrier per three rows
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/mb_seq tools/microbench_seq.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef float F2 __attribute__((ext_vector_type(2)));
typedef float F4 __attribute__((ext_vector_type(4)));
#ifndef H_ROWS
#define H_ROWS 4096
#endif
constexpr int W = 4096, H = H_ROWS;      // -DH_ROWS=528: a 1/8 slab of 4096^2 (+ ghost rows)
// -DSTRIPMAJOR: every vector stored strip by strip (720-pixel column strips, the workgroup's own: [strip][row][x in strip]) instead of row by row, so that a workgroup's
// streams are contiguous in memory from its first row to its last (the halo lanes of the edge waves read the neighbouring strip's storage)
#ifdef STRIPMAJOR
constexpr int kStripPx = 720, kStrips = (W + kStripPx - 1) / kStripPx;
constexpr long N = (long)kStrips * kStripPx * H;
__device__ __forceinline__ long pxIndex(int x, int yphys) { const int s = x / kStripPx; return ((long)s * H + yphys) * kStripPx + (x - s * kStripPx); }
#else
constexpr long N = (long)W * H;
__device__ __forceinline__ long pxIndex(int x, int yphys) { return (long)yphys * W + x; }
#endif

struct Bufs { const float* rIn; const float* pIn; float* rOut; float* pOut; float* delta; const float* angle; const uint8_t* flags; };
template <int PX> struct Row { float v[6 * PX]; float ang[PX]; int f; };

template <int PX, bool NTL> __device__ __forceinline__ Row<PX> loadRow(const Bufs& B, int x, int y, bool flip) {
    Row<PX> r;
    const int yc = min(max(y, 0), H - 1), xc = min(max(x, 0), W - PX);
    const long i = pxIndex(xc, flip ? H - 1 - yc : yc);
    if constexpr (PX == 1) {
        const F2 a = NTL ? __builtin_nontemporal_load((const F2*)B.rIn + i) : ((const F2*)B.rIn)[i];
        const float b = NTL ? __builtin_nontemporal_load(B.rIn + 2 * N + i) : B.rIn[2 * N + i];
        const F2 c = NTL ? __builtin_nontemporal_load((const F2*)B.pIn + i) : ((const F2*)B.pIn)[i];
        const float d = NTL ? __builtin_nontemporal_load(B.pIn + 2 * N + i) : B.pIn[2 * N + i];
        r.v[0] = a.x; r.v[1] = a.y; r.v[2] = b; r.v[3] = c.x; r.v[4] = c.y; r.v[5] = d;
        r.ang[0] = NTL ? __builtin_nontemporal_load(B.angle + i) : B.angle[i];
        r.f = B.flags[i];
    } else {
        const F4 a = NTL ? __builtin_nontemporal_load((const F4*)(B.rIn + 2 * i)) : *(const F4*)(B.rIn + 2 * i);
        const F2 b = NTL ? __builtin_nontemporal_load((const F2*)(B.rIn + 2 * N + i)) : *(const F2*)(B.rIn + 2 * N + i);
        const F4 c = NTL ? __builtin_nontemporal_load((const F4*)(B.pIn + 2 * i)) : *(const F4*)(B.pIn + 2 * i);
        const F2 d = NTL ? __builtin_nontemporal_load((const F2*)(B.pIn + 2 * N + i)) : *(const F2*)(B.pIn + 2 * N + i);
        r.v[0] = a.x; r.v[1] = a.y; r.v[2] = a.z; r.v[3] = a.w; r.v[4] = b.x; r.v[5] = b.y;
        r.v[6] = c.x; r.v[7] = c.y; r.v[8] = c.z; r.v[9] = c.w; r.v[10] = d.x; r.v[11] = d.y;
        const F2 g = NTL ? __builtin_nontemporal_load((const F2*)(B.angle + i)) : *(const F2*)(B.angle + i);
        r.ang[0] = g.x; r.ang[PX - 1] = g.y;
        r.f = *(const uint16_t*)(B.flags + i);
    }
    return r;
}

template <int PX, bool EVEN, bool NTL, bool OVERLAP, bool SYNC, int BLOCK, int GEO = 0, int DEPTH = 3>
__global__ __launch_bounds__(BLOCK) void k(Bufs B, int rowsPerGroup, int gx, int flipI, float* sink) {
    const bool flip = flipI != 0;
    const int bx = blockIdx.x % gx, by = blockIdx.x / gx;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int halo = OVERLAP ? (PX == 1 ? 2 : 1) : 0;                  // halo lanes per side (2 pixels)
    constexpr int span = (64 - 2 * halo) * PX, strip = (BLOCK / 64) * span;
    int x = bx * strip + wave * span + (lane - halo) * PX;
    bool writer = x >= 0 && x + PX <= W && lane >= halo && lane < 64 - halo;
    int xs = x;                                   // where this lane stores
    if (GEO == 2) { xs = bx * (BLOCK / 64) * 60 + wave * 60 + lane; x = xs - 2; writer = xs < W; }                       // loads misaligned (overlap pitch), stores all 64 lanes
    if (GEO == 3) { x = bx * BLOCK + wave * 64 + lane; xs = x - 2; writer = xs >= 0 && xs < W && lane >= 2 && lane < 62; } // loads aligned, stores misaligned 60-lane
    if (GEO == 4 || GEO == 5) {                   // exact 64-pixel waves inside the workgroup, strips overlap by 2 x margin, margins not stored
        const int margin = GEO == 4 ? 32 : 16, pitch = BLOCK - 2 * margin;
        x = bx * pitch - margin + wave * 64 + lane; xs = x;
        const int sx = wave * 64 + lane;          // position inside the strip
        writer = x >= 0 && x < W && sx >= margin && sx < BLOCK - margin;
    }
    const int yb = by * rowsPerGroup, ye = min(yb + rowsPerGroup, H);
    float acc = 0;
    auto consume = [&](int y, const Row<PX>& r, bool live) {
        float s = (float)r.f;
        for (int k2 = 0; k2 < 6 * PX; ++k2) s += r.v[k2];
        for (int k2 = 0; k2 < PX; ++k2) s += r.ang[k2];
        acc += s;
        if (!(writer && live)) return;
        const long i = pxIndex(xs, flip ? H - 1 - y : y);
        float o[6 * PX];
        for (int k2 = 0; k2 < 6 * PX; ++k2) o[k2] = r.v[k2] * 0.5f + s;
        if constexpr (PX == 1) {
            if (EVEN) {
                F2 dd = ((const F2*)B.delta)[i]; float da = B.delta[2 * N + i];
#ifdef RECON_P     // p_{k-2} rebuilt from p_{k-1}, r_{k-1} (what the kernel does since round 2) instead of read
                const F2 qq = F2{r.v[3] - r.v[0], r.v[4] - r.v[1]}; const float qa = r.v[5] - r.v[2];
#else
                const F2 qq = ((const F2*)B.pOut)[i]; const float qa = B.pOut[2 * N + i];
#endif
                dd += 0.25f * qq + 0.125f * F2{r.v[3], r.v[4]}; da += 0.25f * qa + 0.125f * r.v[5];
                ((F2*)B.delta)[i] = dd; B.delta[2 * N + i] = da;
            }
#ifndef RFREE      // r-free loop (round 2): the two input vectors are p_{k-1}, p_{k-2}, only p_k is written
            ((F2*)B.rOut)[i] = F2{o[0], o[1]}; B.rOut[2 * N + i] = o[2];
#endif
            ((F2*)B.pOut)[i] = F2{o[3], o[4]}; B.pOut[2 * N + i] = o[5];
        } else {
            if (EVEN) {
                F4 dd = *(const F4*)(B.delta + 2 * i); F2 da = *(const F2*)(B.delta + 2 * N + i);
                const F4 qq = *(const F4*)(B.pOut + 2 * i); const F2 qa = *(const F2*)(B.pOut + 2 * N + i);
                dd += 0.25f * qq + 0.125f * F4{r.v[6], r.v[7], r.v[8], r.v[9]}; da += 0.25f * qa + 0.125f * F2{r.v[10], r.v[11]};
                *(F4*)(B.delta + 2 * i) = dd; *(F2*)(B.delta + 2 * N + i) = da;
            }
            *(F4*)(B.rOut + 2 * i) = F4{o[0], o[1], o[2], o[3]}; *(F2*)(B.rOut + 2 * N + i) = F2{o[4], o[5]};
            *(F4*)(B.pOut + 2 * i) = F4{o[6], o[7], o[8], o[9]}; *(F2*)(B.pOut + 2 * N + i) = F2{o[10], o[11]};
        }
    };
#define LD(yy) loadRow<PX, NTL>(B, x, yy, flip)
    if constexpr (DEPTH == 6) {      // six row buffers in flight, one barrier per six rows
        Row<PX> a = LD(yb - 2), b = LD(yb - 1), c = LD(yb), d = LD(yb + 1), e = LD(yb + 2), f = LD(yb + 3);
        for (int y = yb - 2; y < ye; y += 6) {
            if (SYNC) __syncthreads();
            { const Row<PX> w = a; a = LD(y + 6); consume(y, w, y >= yb); }
            { const Row<PX> w = b; b = LD(y + 7); consume(y + 1, w, y + 1 >= yb && y + 1 < ye); }
            { const Row<PX> w = c; c = LD(y + 8); consume(y + 2, w, y + 2 >= yb && y + 2 < ye); }
            { const Row<PX> w = d; d = LD(y + 9); consume(y + 3, w, y + 3 >= yb && y + 3 < ye); }
            { const Row<PX> w = e; e = LD(y + 10); consume(y + 4, w, y + 4 >= yb && y + 4 < ye); }
            { const Row<PX> w = f; f = LD(y + 11); consume(y + 5, w, y + 5 >= yb && y + 5 < ye); }
        }
        if (acc == 12345.678f) sink[0] = acc;
        return;
    }
    Row<PX> a = LD(yb - 2), b = LD(yb - 1), c = LD(yb);
    for (int y = yb - 2; y < ye; y += 3) {
        if (SYNC) __syncthreads();
        if (GEO == 6 && (lane == 0 || lane == 63)) { const Row<PX> h = loadRow<PX, NTL>(B, lane == 0 ? x - 1 : x + 1, y + 3, flip); acc += h.v[0] + h.v[2] + h.v[3] + h.v[5] + h.ang[0] + (float)h.f; }
        { const Row<PX> w = a; a = LD(y + 3); consume(y, w, y >= yb); }
        if (GEO == 6 && (lane == 0 || lane == 63)) { const Row<PX> h = loadRow<PX, NTL>(B, lane == 0 ? x - 1 : x + 1, y + 4, flip); acc += h.v[0] + h.v[2] + h.v[3] + h.v[5] + h.ang[0] + (float)h.f; }
        { const Row<PX> w = b; b = LD(y + 4); consume(y + 1, w, y + 1 >= yb && y + 1 < ye); }
        if (GEO == 6 && (lane == 0 || lane == 63)) { const Row<PX> h = loadRow<PX, NTL>(B, lane == 0 ? x - 1 : x + 1, y + 5, flip); acc += h.v[0] + h.v[2] + h.v[3] + h.v[5] + h.ang[0] + (float)h.f; }
        { const Row<PX> w = c; c = LD(y + 5); consume(y + 2, w, y + 2 >= yb && y + 2 < ye); }
    }
#undef LD
    if (acc == 12345.678f) sink[0] = acc;
}

int main(int argc, char** argv) {
    void* p[8];
    const size_t sz[8] = {(size_t)N * 12, (size_t)N * 12, (size_t)N * 12, (size_t)N * 12, (size_t)N * 12, (size_t)N * 4, (size_t)N + 64, 64};
    for (int i = 0; i < 8; ++i) { CK(hipMalloc(&p[i], sz[i])); CK(hipMemset(p[i], 0, sz[i])); }
    float* sink = (float*)p[7];
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    int cus = 256;
    CK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
    // one "sequence step": launch k (even launches carry delta), ping-pong sets, alternating direction when `alt`
    auto seq = [&](auto launchOdd, auto launchEven, bool alt, const char* name) {
        auto run = [&](int k2) {
            Bufs B;
            B.rIn = (const float*)p[(k2 & 1) ? 2 : 0]; B.pIn = (const float*)p[(k2 & 1) ? 3 : 1]; B.rOut = (float*)p[(k2 & 1) ? 0 : 2]; B.pOut = (float*)p[(k2 & 1) ? 1 : 3];
            B.delta = (float*)p[4]; B.angle = (const float*)p[5]; B.flags = (const uint8_t*)p[6];
            const int flip = alt ? (k2 & 1) : 0;
            if (k2 & 1) launchOdd(B, flip); else launchEven(B, flip);
        };
        for (int k2 = 0; k2 < 4; ++k2) run(k2);
        CK(hipDeviceSynchronize()); CK(hipEventRecord(e0));
        for (int k2 = 0; k2 < 40; ++k2) run(k2);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipGetLastError());
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 40;
        printf("%-78s %7.1f us/launch  %6.0f GB/s ideal\n", name, ms * 1e3, 71.0 * N / 1e9 / ms * 1e3);
    };
#define CASE(PX, NTL, OV, SY, BLOCK, OCC, ALT, name)                                                                                     \
    {                                                                                                                                    \
        constexpr int halo = OV ? (PX == 1 ? 2 : 1) : 0, span = (64 - 2 * halo) * PX, strip = (BLOCK / 64) * span;                         \
        const int gx = (W + strip - 1) / strip, gy = cus * OCC / gx, rpg = (H + gy - 1) / gy, gy2 = (H + rpg - 1) / rpg;                   \
        seq([&](const Bufs& B, int flip) { k<PX, false, NTL, OV, SY, BLOCK><<<gx * gy2, BLOCK>>>(B, rpg, gx, flip, sink); },                \
            [&](const Bufs& B, int flip) { k<PX, true, NTL, OV, SY, BLOCK><<<gx * gy2, BLOCK>>>(B, rpg, gx, flip, sink); }, ALT, name);   \
    }
#define CASEG(NTL, BLOCK, GEO, PITCH, name)                                                                                               \
    {                                                                                                                                    \
        const int gx = (W + PITCH - 1) / PITCH, gy = cus / gx, rpg = (H + gy - 1) / gy, gy2 = (H + rpg - 1) / rpg;                         \
        seq([&](const Bufs& B, int flip) { k<1, false, NTL, GEO == 0, true, BLOCK, GEO><<<gx * gy2, BLOCK>>>(B, rpg, gx, flip, sink); },    \
            [&](const Bufs& B, int flip) { k<1, true, NTL, GEO == 0, true, BLOCK, GEO><<<gx * gy2, BLOCK>>>(B, rpg, gx, flip, sink); }, true, name); \
    }
#define CASED(NTL, BLOCK, DEPTH, SY, name)                                                                                                 \
    {                                                                                                                                    \
        const int gx = (W + 720 * BLOCK / 768 - 1) / (720 * BLOCK / 768), gy = cus / gx, rpg = (H + gy - 1) / gy, gy2 = (H + rpg - 1) / rpg;   \
        seq([&](const Bufs& B, int flip) { k<1, false, NTL, true, SY, BLOCK, 0, DEPTH><<<gx * gy2, BLOCK>>>(B, rpg, gx, flip, sink); },      \
            [&](const Bufs& B, int flip) { k<1, true, NTL, true, SY, BLOCK, 0, DEPTH><<<gx * gy2, BLOCK>>>(B, rpg, gx, flip, sink); }, true, name); \
    }
    if (argc > 1 && argv[1][0] == 's') {      // the current kernel's shape only (plain loads): row-major against -DSTRIPMAJOR builds
        for (int rep = 0; rep < 3; ++rep) {
            CASE(1, false, true, true, 768, 1, true, "1 px/lane  --  overlap  sync  768 thr  alt sweep   (= current kernel)");
            CASE(1, false, true, true, 768, 1, false, "1 px/lane  --  overlap  sync  768 thr  same sweep");
            CASE(1, false, true, false, 768, 1, true, "1 px/lane  --  overlap  free  768 thr  alt sweep");
        }
        return 0;
    }
    if (argc > 1 && argv[1][0] == 'd') {
        for (int rep = 0; rep < 2; ++rep) {
            CASED(false, 768, 3, true, "depth 3 (current)   768 thr sync");
            CASED(false, 768, 6, true, "depth 6             768 thr sync/6 rows");
            CASED(false, 768, 6, false, "depth 6             768 thr free");
            CASED(false, 512, 6, true, "depth 6             512 thr sync/6 rows");
            CASED(false, 1024, 3, true, "depth 3            1024 thr sync");
            printf("\n");
        }
        return 0;
    }
    if (argc > 1) {
        for (int rep = 0; rep < 2; ++rep) {
            CASEG(true, 768, 0, 720, "GEO0 overlap loads+stores (current)                nt");
            CASEG(false, 768, 0, 720, "GEO0 overlap loads+stores (current)                --");
            CASEG(true, 768, 1, 768, "GEO1 exact                                         nt");
            CASEG(false, 768, 1, 768, "GEO1 exact                                         --");
            CASEG(true, 768, 2, 720, "GEO2 loads misaligned/overlapped, stores aligned   nt");
            CASEG(false, 768, 2, 720, "GEO2 loads misaligned/overlapped, stores aligned   --");
            CASEG(true, 768, 3, 768, "GEO3 loads aligned, stores misaligned 60 lanes     nt");
            CASEG(false, 768, 3, 768, "GEO3 loads aligned, stores misaligned 60 lanes     --");
            CASEG(true, 768, 4, 704, "GEO4 exact waves, strip pitch 704 (32 px margins)  nt");
            CASEG(false, 768, 4, 704, "GEO4 exact waves, strip pitch 704 (32 px margins)  --");
            CASEG(true, 768, 5, 736, "GEO5 exact waves, strip pitch 736 (16 px margins)  nt");
            CASEG(false, 768, 5, 736, "GEO5 exact waves, strip pitch 736 (16 px margins)  --");
            CASEG(true, 768, 6, 768, "GEO6 exact + 2-lane halo loads per row              nt");
            CASEG(false, 768, 6, 768, "GEO6 exact + 2-lane halo loads per row              --");
            CASEG(true, 1024, 4, 960, "GEO4 1024 thr, pitch 960                            nt");
            CASEG(true, 512, 4, 448, "GEO4 512 thr, pitch 448                             nt");
            printf("\n");
        }
        return 0;
    }
    for (int rep = 0; rep < 2; ++rep) {
        CASE(1, true, true, true, 768, 1, true, "1 px/lane  nt  overlap  sync  768 thr  alt sweep   (= current kernel)");
        CASE(1, true, true, true, 768, 1, false, "1 px/lane  nt  overlap  sync  768 thr  same sweep");
        CASE(1, false, true, true, 768, 1, true, "1 px/lane  --  overlap  sync  768 thr  alt sweep");
        CASE(1, false, true, true, 768, 1, false, "1 px/lane  --  overlap  sync  768 thr  same sweep");
        CASE(1, true, false, true, 768, 1, true, "1 px/lane  nt  exact    sync  768 thr  alt sweep");
        CASE(1, false, false, true, 768, 1, true, "1 px/lane  --  exact    sync  768 thr  alt sweep");
        CASE(1, true, true, false, 768, 1, true, "1 px/lane  nt  overlap  free  768 thr  alt sweep");
        CASE(1, false, true, false, 768, 1, true, "1 px/lane  --  overlap  free  768 thr  alt sweep");
        CASE(1, false, true, true, 1024, 1, true, "1 px/lane  --  overlap  sync 1024 thr  alt sweep");
        CASE(1, false, true, true, 512, 1, true, "1 px/lane  --  overlap  sync  512 thr  alt sweep");
        CASE(1, false, true, true, 512, 2, true, "1 px/lane  --  overlap  sync  512 thr x2/CU  alt sweep");
        CASE(1, false, true, true, 256, 4, true, "1 px/lane  --  overlap  sync  256 thr x4/CU  alt sweep");
        CASE(2, true, true, true, 768, 1, true, "2 px/lane  nt  overlap  sync  768 thr  alt sweep");
        CASE(2, false, true, true, 768, 1, true, "2 px/lane  --  overlap  sync  768 thr  alt sweep");
        CASE(2, false, true, true, 384, 1, true, "2 px/lane  --  overlap  sync  384 thr  alt sweep");
        CASE(2, false, true, true, 384, 2, true, "2 px/lane  --  overlap  sync  384 thr x2/CU  alt sweep");
        CASE(2, false, true, true, 512, 1, true, "2 px/lane  --  overlap  sync  512 thr  alt sweep");
        CASE(2, false, false, true, 512, 1, true, "2 px/lane  --  exact    sync  512 thr  alt sweep");
        CASE(2, false, true, false, 512, 1, true, "2 px/lane  --  overlap  free  512 thr  alt sweep");
        CASE(2, true, true, true, 512, 1, true, "2 px/lane  nt  overlap  sync  512 thr  alt sweep");
        CASE(2, false, true, true, 256, 2, true, "2 px/lane  --  overlap  sync  256 thr x2/CU  alt sweep");
        printf("\n");
    }
    return 0;
}
