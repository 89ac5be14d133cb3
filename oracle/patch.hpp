// TEST INFRASTRUCTURE ONLY (CPU oracle) -- never linked into, imported by, or called from the product path.
//
// CPU restatement of the block-local "patch" solver for poisson_image_editing.  It follows the reference's
// hand-written comparator examples/poisson_image_editing/src/PatchSolverWarping.cu: the outer loops :211-241 (nNonLinearIterations x
// nLinearIterations sweeps, tiling shifted by the Halton points of :208-209), and per patch the kernel :67-199 (b = -J^T F at the current X,
// Jacobi-preconditioned CG over the patch's own pixels with p = 0 outside the patch and on masked pixels, X += delta at the end), with the
// operator and right-hand side of PatchSolverWarpingEquations.h:52-112 written in Opt's scaling of the same energy
// (poisson_image_editing.t: A = J^T J = 2 L, b = 2 sum_n [(t_c - t_n) - (x_c - x_n)]; the comparator's are 1/2 of both, which leaves every
// iterate unchanged).  Differences from the comparator, shared with the HIP kernel and stated in DESIGN.md: a sweep reads the X of the
// previous sweep everywhere (the CUDA kernel updates X in place while other blocks still read it), alpha / beta are guarded by "> 0" like
// Opt's PCG (solver.t:456-459, 544-547) instead of FLOAT_EPSILON, and the patch size is a parameter.  Parity is "unpinned" in the sense of
// the task statement: the comparator cannot be built here (CUDA); this file pins the HIP kernel to an independent scalar implementation.
#pragma once
#include <cmath>
#include <vector>

namespace oracle {

inline double radicalInverse(int i, int base) { double f = 1, r = 0; while (i > 0) { f /= base; r += f * (i % base); i /= base; } return r; }

template <class T>
struct PoissonPatch {
    int W, H;
    const T* Tg; const T* M;
    static constexpr int dx[4] = {1, -1, 0, 0}, dy[4] = {0, 0, 1, -1};
    bool inside(int x, int y) const { return x >= 0 && x < W && y >= 0 && y < H; }

    // 1/2 sum r^2 over non-excluded pixels (solver.t:580-592)
    double cost(const std::vector<T>& X) const {
        double acc = 0;
        for (int y = 0; y < H; ++y) for (int x = 0; x < W; ++x) {
            const long c = (long)y * W + x;
            if (M[c] != T(0)) continue;
            T e = 0;
            for (int n = 0; n < 4; ++n) {
                if (!inside(x + dx[n], y + dy[n])) continue;
                const long ni = (long)(y + dy[n]) * W + x + dx[n];
                for (int k = 0; k < 4; ++k) { const T r = (X[4 * c + k] - X[4 * ni + k]) - (Tg[4 * c + k] - Tg[4 * ni + k]); e += r * r; }
            }
            acc += (double)(T(0.5) * e);
        }
        return acc;
    }

    // one patch with corner (x0, y0): PatchSolverWarping.cu:67-199
    void solvePatch(const std::vector<T>& Xin, std::vector<T>& Xout, int x0, int y0, int PS, int nPatchIters) const {
        const int n = PS * PS;
        std::vector<T> R(4 * n, 0), P(4 * n, 0), AP(4 * n, 0), Z(4 * n, 0), D(4 * n, 0), pre(n, 0);
        std::vector<char> act(n, 0);
        auto local = [&](int gx, int gy) { return (gy - y0) * PS + (gx - x0); };
        auto inPatch = [&](int gx, int gy) { return gx >= x0 && gx < x0 + PS && gy >= y0 && gy < y0 + PS; };
        T rz = 0;
        for (int ty = 0; ty < PS; ++ty) for (int tx = 0; tx < PS; ++tx) {
            const int gx = x0 + tx, gy = y0 + ty, l = ty * PS + tx;
            if (!inside(gx, gy)) continue;
            const long c = (long)gy * W + gx;
            if (M[c] != T(0)) continue;
            act[l] = 1;
            T cnt = 0;
            for (int q = 0; q < 4; ++q) {
                if (!inside(gx + dx[q], gy + dy[q])) continue;
                const long ni = (long)(gy + dy[q]) * W + gx + dx[q];
                for (int k = 0; k < 4; ++k) { const T e = (Xin[4 * c + k] - Xin[4 * ni + k]) - (Tg[4 * c + k] - Tg[4 * ni + k]); R[4 * l + k] -= e + e; }
                cnt += T(2);
            }
            pre[l] = cnt > T(0) ? T(1) / cnt : T(1);
            for (int k = 0; k < 4; ++k) { P[4 * l + k] = pre[l] * R[4 * l + k]; rz += R[4 * l + k] * P[4 * l + k]; }
        }
        if (rz > T(0)) {
            for (int it = 0; it < nPatchIters; ++it) {
                T den = 0;
                for (int ty = 0; ty < PS; ++ty) for (int tx = 0; tx < PS; ++tx) {
                    const int gx = x0 + tx, gy = y0 + ty, l = ty * PS + tx;
                    if (!act[l]) continue;
                    for (int k = 0; k < 4; ++k) AP[4 * l + k] = 0;
                    for (int q = 0; q < 4; ++q) {
                        const int nx = gx + dx[q], ny = gy + dy[q];
                        if (!inside(nx, ny)) continue;
                        const bool live = inPatch(nx, ny) && act[local(nx, ny)];
                        for (int k = 0; k < 4; ++k) { const T d = P[4 * l + k] - (live ? P[4 * local(nx, ny) + k] : T(0)); AP[4 * l + k] += d + d; }
                    }
                    for (int k = 0; k < 4; ++k) den += P[4 * l + k] * AP[4 * l + k];
                }
                const T alpha = den > T(0) ? rz / den : T(0);
                T rzNew = 0;
                for (int l = 0; l < n; ++l) {
                    if (!act[l]) continue;
                    for (int k = 0; k < 4; ++k) {
                        D[4 * l + k] += alpha * P[4 * l + k];
                        R[4 * l + k] -= alpha * AP[4 * l + k];
                        Z[4 * l + k] = pre[l] * R[4 * l + k];
                        rzNew += Z[4 * l + k] * R[4 * l + k];
                    }
                }
                const T beta = rz > T(0) ? rzNew / rz : T(0);
                for (int l = 0; l < n; ++l) if (act[l]) for (int k = 0; k < 4; ++k) P[4 * l + k] = Z[4 * l + k] + beta * P[4 * l + k];
                rz = rzNew;
            }
        }
        for (int ty = 0; ty < PS; ++ty) for (int tx = 0; tx < PS; ++tx) {
            const int gx = x0 + tx, gy = y0 + ty, l = ty * PS + tx;
            if (!inside(gx, gy)) continue;
            const long c = (long)gy * W + gx;
            for (int k = 0; k < 4; ++k) Xout[4 * c + k] = Xin[4 * c + k] + D[4 * l + k];
        }
    }

    // one sweep: every patch of the tiling shifted by (ox, oy) (PatchSolverWarping.cu:194-199: one extra block per axis)
    void sweep(const std::vector<T>& Xin, std::vector<T>& Xout, int ox, int oy, int PS, int nPatchIters) const {
        const int bx = (W + PS - 1) / PS + 1, by = (H + PS - 1) / PS + 1;
        for (int j = 0; j < by; ++j) for (int i = 0; i < bx; ++i) solvePatch(Xin, Xout, i * PS - ox, j * PS - oy, PS, nPatchIters);
    }

    // costs[0] = initial cost, costs[s + 1] = cost after outer step s (PatchSolverWarping.cu:211-241)
    void solve(T* Xio, int nIterations, int lIterations, int nPatchIters, int PS, double* costs) const {
        std::vector<T> X(Xio, Xio + 4L * W * H), Y(X);
        costs[0] = cost(X);
        int o = 0;
        for (int it = 0; it < nIterations; ++it) {
            for (int l = 0; l < lIterations; ++l) {
                sweep(X, Y, (int)((float)radicalInverse(o % 8, 2) * PS), (int)((float)radicalInverse(o % 8, 3) * PS), PS, nPatchIters);
                X.swap(Y); ++o;
            }
            costs[it + 1] = cost(X);
        }
        for (long i = 0; i < 4L * W * H; ++i) Xio[i] = X[i];
    }
};

}  // namespace oracle
