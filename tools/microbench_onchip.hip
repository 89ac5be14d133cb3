// What the ARITHMETIC of a persistent (whole-PCG-solve) image_warping slab kernel would cost per iteration: a 4096 x 512 slab held on
// chip -- one workgroup of 8 waves per CU owns a 256 x 32 tile, a lane holds 16 rows of one column: p, r and A p of the pixel in registers (144 of 256 VGPRs at two waves per SIMD),
// cos / sin of its angle in registers too (constant over a linear solve), delta in LDS (96 KB) -- and runs, per iteration, PCGStep2 + PCGStep3 on its pixels followed by ONE
// stencil evaluation with the sums of the expanded beta numerator (exact double products), with the real expressions of iw_pcgIter2's unit-lattice path.  What is NOT here is the
// communication: vertical neighbours outside the lane's 16 rows and horizontal neighbours outside the wave are taken as "absent" (on = 0), alpha / beta are constants; the two
// synchronisations a real kernel needs per iteration are measured separately (tools/microbench_gridsync.hip: 3.0 us sum + 3.15 us halo hand-over).
//   hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize -o /tmp/onchip tools/microbench_onchip.hip && /tmp/onchip [iterations]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// Where the state lives is the question (p, r: always registers):
//   CS_REGS   1: cos / sin of the lane's 16 angles in 32 registers;  0: recomputed from the angle (kept packed in LDS? no: in 16 registers) every iteration
//   AP_LDS    0: A p in 48 registers;  1: in LDS (96 KB)
//   DELTA_LDS 1: delta in LDS (96 KB);  0: not modelled (it would be streamed to HBM every second iteration: 12 B per pixel and iteration)
#ifndef CS_REGS
#define CS_REGS 1
#endif
#ifndef AP_LDS
#define AP_LDS 0
#endif
#ifndef DELTA_LDS
#define DELTA_LDS 1
#endif
//   DELTA_ATOMIC 1 (with DELTA_LDS 0): delta += alpha p as a no-return hardware float atomic per scalar (one lane per address: deterministic; no registers, no exposed latency;
//                the 25 MB of delta mostly stay in the L2s)
#ifndef DELTA_ATOMIC
#define DELTA_ATOMIC 0
#endif
constexpr int W = 4096, H = 512, ROWS = 16, BLOCK = 512, TILE_W = 256, TILE_H = 32;
constexpr long N = (long)W * H;

template <bool RIGHT> __device__ __forceinline__ float dppShift(float v) {      // value of lane + 1 (RIGHT) / lane - 1, 0 at the wave's edge
    int r;
    if (RIGHT) r = __builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x130 /* wave_shl:1 */, 0xf, 0xf, false);
    else r = __builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
    return __int_as_float(r);
}
struct Q { float ox, oy, a, c, s, on; };
// the two residuals shared by centre c and its lattice neighbour n in direction (DX, DY)  (energy_image_warping.hip iw_pairQ, unit lattice)
template <int DX, int DY>
__device__ __forceinline__ void pairQ(const Q& c, const Q& n, float& ax, float& ay, float& aa) {
    const float Dcx = DX ? float(DX) * c.s : float(DY) * c.c, Dcy = DX ? float(-DX) * c.c : float(DY) * c.s;
    const float Dnx = DX ? float(-DX) * n.s : float(-DY) * n.c, Dny = DX ? float(DX) * n.c : float(-DY) * n.s;
    const float dx = c.ox - n.ox, dy = c.oy - n.oy;
    const float jcx = dx - Dcx * c.a, jcy = dy - Dcy * c.a;
    const float jnx = -dx - Dnx * n.a, jny = -dy - Dny * n.a;
    ax += n.on * (jcx - jnx); ay += n.on * (jcy - jny);
    aa -= n.on * (Dcx * jcx + Dcy * jcy);
}

__global__ __launch_bounds__(BLOCK) void k_onchip(const float* __restrict__ pIn, const float* __restrict__ rIn, const float* __restrict__ angle, const uint8_t* __restrict__ flags,
                                                  float* __restrict__ pOut, float* __restrict__ deltaG, double* __restrict__ sums, int iters, float alpha, float beta, float w2, float wf2) {
    extern __shared__ float lds[];                    // delta: [row][component][thread]  (conflict-free), then the two 16-entry preconditioner tables
    float* dl = lds;                                  // [ROWS * 3][BLOCK] if DELTA_LDS
    float* apl = lds + (DELTA_LDS ? ROWS * 3 * BLOCK : 0);      // [ROWS * 3][BLOCK] if AP_LDS
    float* mTab = apl + (AP_LDS ? ROWS * 3 * BLOCK : 0);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int tx = blockIdx.x % (W / TILE_W), ty = blockIdx.x / (W / TILE_W);
    const int x = tx * TILE_W + (wave & 3) * 64 + lane, y0 = ty * TILE_H + (wave >> 2) * ROWS;
    if (threadIdx.x < 32) mTab[threadIdx.x] = 1.0f / ((1.0f + sqrtf(2.0f * w2 * (threadIdx.x & 7) + ((threadIdx.x & 8) ? wf2 : 0.0f))) * (1.0f + sqrtf(2.0f * w2 * (threadIdx.x & 7))));
    float p[ROWS][3], r[ROWS][3], ap[AP_LDS ? 1 : ROWS][3], cs[CS_REGS ? ROWS : 1][2], ang[CS_REGS ? 1 : ROWS];
    unsigned fl[ROWS / 4];
#pragma unroll
    for (int y = 0; y < ROWS; ++y) {
        const long i = (long)(y0 + y) * W + x;
        p[y][0] = pIn[2 * i]; p[y][1] = pIn[2 * i + 1]; p[y][2] = pIn[2 * N + i];
        r[y][0] = rIn[2 * i]; r[y][1] = rIn[2 * i + 1]; r[y][2] = rIn[2 * N + i];
        if (AP_LDS) { for (int c = 0; c < 3; ++c) apl[(y * 3 + c) * BLOCK + threadIdx.x] = 0; } else ap[AP_LDS ? 0 : y][0] = ap[AP_LDS ? 0 : y][1] = ap[AP_LDS ? 0 : y][2] = 0;
        if (CS_REGS) { float sn, cn; sincosf(angle[i], &sn, &cn); cs[CS_REGS ? y : 0][0] = cn; cs[CS_REGS ? y : 0][1] = sn; } else ang[CS_REGS ? 0 : y] = angle[i];
        if ((y & 3) == 0) fl[y / 4] = 0;
        fl[y / 4] |= (unsigned)flags[i] << (8 * (y & 3));
        if (DELTA_LDS) for (int c = 0; c < 3; ++c) dl[(y * 3 + c) * BLOCK + threadIdx.x] = 0;
    }
    __syncthreads();
    double accDen = 0, accNum = 0, acc2 = 0, acc3 = 0;
    for (int it = 0; it < iters; ++it) {
        accDen = accNum = acc2 = acc3 = 0;
        // PCGStep2 + PCGStep3 of the previous iteration on this lane's 16 pixels
#pragma unroll
        for (int y = 0; y < ROWS; ++y) {
            const unsigned f = (fl[y / 4] >> (8 * (y & 3))) & 0xff;
            const float mO = mTab[(f >> 2) & 15], mA = mTab[16 + ((f >> 2) & 7)];
            const float m[3] = {mO, mO, mA};
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                if (DELTA_LDS) { float* d = dl + (y * 3 + c) * BLOCK + threadIdx.x; *d = *d + alpha * p[y][c]; }
                if (DELTA_ATOMIC) { const long i = (long)(y0 + y) * W + x; unsafeAtomicAdd(c < 2 ? deltaG + 2 * i + c : deltaG + 2 * N + i, alpha * p[y][c]); }
                r[y][c] = r[y][c] - alpha * (AP_LDS ? apl[(y * 3 + c) * BLOCK + threadIdx.x] : ap[AP_LDS ? 0 : y][c]);
                p[y][c] = m[c] * r[y][c] + beta * p[y][c];
            }
        }
        // PCGStep1 on the new p, with the sums of the expansion
        float csPrev[2] = {1, 0}, csCur[2] = {1, 0}, csNext[2] = {1, 0};      // CS_REGS == 0: a three-row window of recomputed cos / sin
        if (!CS_REGS) { sincosf(ang[0], &csCur[1], &csCur[0]); }
#pragma unroll
        for (int y = 0; y < ROWS; ++y) {
            const unsigned f = (fl[y / 4] >> (8 * (y & 3))) & 0xff;
            if (!CS_REGS && y + 1 < ROWS) sincosf(ang[CS_REGS ? 0 : y + 1], &csNext[1], &csNext[0]);
            Q c{p[y][0], p[y][1], p[y][2], CS_REGS ? cs[CS_REGS ? y : 0][0] : csCur[0], CS_REGS ? cs[CS_REGS ? y : 0][1] : csCur[1], (f & 1) ? 1.0f : 0.0f};
            const float fw = (f & 2) ? wf2 : 0.0f;
            Q lf{dppShift<false>(c.ox), dppShift<false>(c.oy), dppShift<false>(c.a), dppShift<false>(c.c), dppShift<false>(c.s), dppShift<false>(c.on)};
            Q rt{dppShift<true>(c.ox), dppShift<true>(c.oy), dppShift<true>(c.a), dppShift<true>(c.c), dppShift<true>(c.s), dppShift<true>(c.on)};
            Q up{0, 0, 0, 1, 0, 0}, dn{0, 0, 0, 1, 0, 0};
            if (y > 0) { const unsigned g = (fl[(y - 1) / 4] >> (8 * ((y - 1) & 3))) & 0xff; up = Q{p[y - 1][0], p[y - 1][1], p[y - 1][2], CS_REGS ? cs[CS_REGS ? y - 1 : 0][0] : csPrev[0], CS_REGS ? cs[CS_REGS ? y - 1 : 0][1] : csPrev[1], (g & 1) ? 1.0f : 0.0f}; }
            if (y + 1 < ROWS) { const unsigned g = (fl[(y + 1) / 4] >> (8 * ((y + 1) & 3))) & 0xff; dn = Q{p[y + 1][0], p[y + 1][1], p[y + 1][2], CS_REGS ? cs[CS_REGS ? y + 1 : 0][0] : csNext[0], CS_REGS ? cs[CS_REGS ? y + 1 : 0][1] : csNext[1], (g & 1) ? 1.0f : 0.0f}; }
            float ax = 0, ay = 0, aa = 0;
            pairQ<1, 0>(c, rt, ax, ay, aa); pairQ<-1, 0>(c, lf, ax, ay, aa); pairQ<0, 1>(c, dn, ax, ay, aa); pairQ<0, -1>(c, up, ax, ay, aa);
            const float o[3] = {c.on * (w2 * ax + fw * c.ox), c.on * (w2 * ay + fw * c.oy), c.on * (w2 * aa)};
            const float mO = mTab[(f >> 2) & 15], mA = mTab[16 + ((f >> 2) & 7)];
            const double m[3] = {(double)mO, (double)mO, (double)mA};
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                if (AP_LDS) apl[(y * 3 + k) * BLOCK + threadIdx.x] = o[k]; else ap[AP_LDS ? 0 : y][k] = o[k];
                const double rr = (double)r[y][k], aA = (double)o[k], mr = m[k] * rr;
                accDen += (double)(p[y][k] * o[k]);
                accNum += mr * rr; acc2 += mr * aA; acc3 += (m[k] * aA) * aA;
            }
            if (!CS_REGS) { csPrev[0] = csCur[0]; csPrev[1] = csCur[1]; csCur[0] = csNext[0]; csCur[1] = csNext[1]; }
        }
    }
    // (a real kernel reduces the four sums over the workgroup and the grid every iteration: tools/microbench_gridsync.hip)
    if (accDen + accNum + acc2 + acc3 == 12345.678) sums[0] = accDen;
    const long i = (long)y0 * W + x;
    pOut[i] = p[0][0] + p[ROWS - 1][2] + (AP_LDS ? apl[threadIdx.x] : ap[AP_LDS ? 0 : 3][1]) + r[7][0] + (DELTA_LDS ? dl[threadIdx.x] : 0.0f);
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 200;
    float *pIn, *rIn, *angle, *pOut, *deltaG; uint8_t* flags; double* sums;
    CHECK(hipMalloc(&deltaG, 3 * N * 4)); CHECK(hipMemset(deltaG, 0, 3 * N * 4));
    CHECK(hipMalloc(&pIn, 3 * N * 4)); CHECK(hipMalloc(&rIn, 3 * N * 4)); CHECK(hipMalloc(&angle, N * 4)); CHECK(hipMalloc(&pOut, N * 4)); CHECK(hipMalloc(&flags, N)); CHECK(hipMalloc(&sums, 64));
    std::vector<float> h(3 * N); std::vector<uint8_t> hf(N);
    for (long i = 0; i < 3 * N; ++i) h[i] = 1e-3f * (float)((i * 2654435761u) % 1000) - 0.5f;
    CHECK(hipMemcpy(pIn, h.data(), 3 * N * 4, hipMemcpyHostToDevice)); CHECK(hipMemcpy(rIn, h.data(), 3 * N * 4, hipMemcpyHostToDevice)); CHECK(hipMemcpy(angle, h.data(), N * 4, hipMemcpyHostToDevice));
    for (long i = 0; i < N; ++i) hf[i] = (uint8_t)(1 | ((i % 97 == 0) ? 2 : 0) | (4 << 2));
    CHECK(hipMemcpy(flags, hf.data(), N, hipMemcpyHostToDevice));
    const size_t ldsBytes = (size_t)((DELTA_LDS ? ROWS * 3 * BLOCK : 0) + (AP_LDS ? ROWS * 3 * BLOCK : 0) + 32) * sizeof(float);
    printf("CS_REGS=%d AP_LDS=%d DELTA_LDS=%d DELTA_ATOMIC=%d\n", CS_REGS, AP_LDS, DELTA_LDS, DELTA_ATOMIC);
    CHECK(hipFuncSetAttribute((const void*)k_onchip, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsBytes));
    const int grid = (W / TILE_W) * (H / TILE_H);
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    printf("%d x %d pixels on %d workgroups of %d threads, %zu KB of LDS each; %d iterations per launch\n", W, H, grid, BLOCK, ldsBytes / 1024, iters);
    for (int rep = 0; rep < 3; ++rep) {
        float ms[2];
        for (int k = 0; k < 2; ++k) {      // two launch lengths: the difference is free of the load / store ends of the kernel
            const int n = k ? iters : iters / 2;
            CHECK(hipEventRecord(e0));
            k_onchip<<<grid, BLOCK, ldsBytes>>>(pIn, rIn, angle, flags, pOut, deltaG, sums, n, 1e-3f, 0.5f, 1.0f, 0.25f);
            CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1)); CHECK(hipGetLastError());
            CHECK(hipEventElapsedTime(&ms[k], e0, e1));
        }
        printf("rep %d: %d iterations %.1f us, %d iterations %.1f us  ->  %.2f us per iteration (arithmetic of PCGStep2 + PCGStep3 + PCGStep1 + sums, no communication)\n",
               rep, iters / 2, ms[0] * 1e3, iters, ms[1] * 1e3, (ms[1] - ms[0]) * 1e3 / (iters - iters / 2));
    }
    return 0;
}
