// TEST INFRASTRUCTURE ONLY (CPU oracle) -- never linked into, imported by, or called from the product path.
//
// examples/shape_from_shading/shape_from_shading.t:1-90 restated for the oracle.
// ComputedArrays: Opt stores the array value plus one "gradient image" per unknown it depends on, all
// re-evaluated by the `precompute` kernel after every update / revert (o.t:1007-1040, 2387-2409,
// solver.t:607-614, 1005, 1116, 1155); a residual that reads the array gets its partials through those
// stored gradient images (o.t:913-925).  Here: B_I + 3 gradient arrays; `valid` is boolean (zero gradients).
#pragma once
#include "dual.hpp"
#include "solver.hpp"
#include <cstdint>

namespace oracle {

template <class T>
struct ShapeFromShading : Energy<T> {
    long W, H;
    T w_p = 0, w_s = 0, w_g = 0, f_x = 0, f_y = 0, u_x = 0, u_y = 0, L[9];
    T* X = nullptr; const T *D_i = nullptr, *Im = nullptr; const uint8_t *edgeMaskR = nullptr, *edgeMaskC = nullptr;
    std::vector<T> B_I, B_I_d0, B_I_d1, B_I_d2, validArr;   // d0: X(-1,0)  d1: X(0,0)  d2: X(0,-1)

    ShapeFromShading(const unsigned* dims) : W(dims[0]), H(dims[1]) {
        this->usePreconditioner = false;   // no UsePreconditioner call in the .t -> default (o.t:214)
        this->addUnknown(W * H, 1);
        for (auto* v : {&B_I, &B_I_d0, &B_I_d1, &B_I_d2, &validArr}) v->assign(W * H, T(0));
    }
    void bind(void** p) override {
        // sqrt(Param(...)) is evaluated in opt_float on the float parameter (shape_from_shading.t:4-6)
        w_p = std::sqrt((T) * (const float*)p[0]); w_s = std::sqrt((T) * (const float*)p[1]); w_g = std::sqrt((T) * (const float*)p[2]);
        f_x = (T) * (const float*)p[3]; f_y = (T) * (const float*)p[4]; u_x = (T) * (const float*)p[5]; u_y = (T) * (const float*)p[6];
        for (int i = 0; i < 9; ++i) L[i] = (T) * (const float*)p[7 + i];
        X = (T*)p[16]; D_i = (const T*)p[17]; Im = (const T*)p[18];
        edgeMaskR = (const uint8_t*)p[19]; edgeMaskC = (const uint8_t*)p[20];
    }
    T* unknownPtr(int) override { return X; }
    long nCentered() const override { return W * H; }
    long rowWidth() const override { return W; }
    bool depthValid(long x, long y) const { return x >= 0 && x < W && y >= 0 && y < H && D_i[y * W + x] > T(0); }
    bool excluded(int, long e) const override { return !(D_i[e] > T(0)); }   // Exclude(Not(DepthValid(0,0))) (:70)
    bool excludedCentered(long e) const override { return !(D_i[e] > T(0)); }
    bool interior(long x, long y) const { return x >= 1 && x <= W - 2 && y >= 1 && y <= H - 2; }   // InBoundsExpanded(0,0,1)

    void precompute() override {
        typedef Dual<T, 3> D;
        for (long y = 0; y < H; ++y)
            for (long x = 0; x < W; ++x) {
                const long e = y * W + x;
                T bi = 0, g0 = 0, g1 = 0, g2 = 0, vl = 0;
                if (interior(x, y)) {
                    if (depthValid(x - 1, y) && depthValid(x, y) && depthValid(x, y - 1)) {   // (:62-66)
                        D d0 = D::var(X[e - 1], 0), d1 = D::var(X[e], 1), d2 = D::var(X[e - W], 2);
                        const T i = (T)x, j = (T)y;
                        // normalAt (:32-43)
                        D n_x = d2 * (d1 - d0) / f_y;
                        D n_y = d0 * (d1 - d2) / f_x;
                        D n_z = (n_x * (u_x - i) / f_x) + (n_y * (u_y - j) / f_y) - (d0 * d2 / (f_x * f_y));
                        D sq = n_x * n_x + n_y * n_y + n_z * n_z;
                        D inv = select(sq.v > T(0), T(1) / sqrt(sq), D(T(1)));
                        n_x = inv * n_x; n_y = inv * n_y; n_z = inv * n_z;
                        // B (:45-54)
                        D Bv = D(L[0]) + L[1] * n_y + L[2] * n_z + L[3] * n_x + L[4] * n_x * n_y + L[5] * n_y * n_z +
                               L[6] * (-(n_x * n_x) - n_y * n_y + T(2) * n_z * n_z) + L[7] * n_z * n_x + L[8] * (n_x * n_x - n_y * n_y);
                        // I (:56-58)
                        const T Iv = Im[e] * T(0.5) + T(0.25) * (Im[e - 1] + Im[e - W]);
                        D r = Bv - Iv;
                        bi = r.v; g0 = r.d[0]; g1 = r.d[1]; g2 = r.d[2];
                    }
                    // valid (:82-87)
                    bool v = depthValid(x, y) && depthValid(x, y - 1) && depthValid(x, y + 1) && depthValid(x - 1, y) && depthValid(x + 1, y);
                    const T thr = T(0.01);
                    v = v && std::fabs(X[e] - X[e - W]) < thr && std::fabs(X[e] - X[e + W]) < thr && std::fabs(X[e] - X[e - 1]) < thr && std::fabs(X[e] - X[e + 1]) < thr;
                    vl = v ? T(1) : T(0);
                }
                B_I[e] = bi; B_I_d0[e] = g0; B_I_d1[e] = g1; B_I_d2[e] = g2; validArr[e] = vl;
            }
    }

    int evalCentered(long e, Inst<T>* out) const override {
        const long x = e % W, y = e / W;
        int k = 0;
        {   // fitting term (:73-74)
            Inst<T>& I = out[k++];
            const bool v = D_i[e] > T(0);
            I.n = 1; I.idx[0] = e; I.val = v ? w_p * (X[e] - D_i[e]) : T(0); I.dv[0] = v ? w_p : T(0);
        }
        const bool in1 = interior(x, y);
        {   // shading terms (:77-80); supports through the gradient images of B_I
            Inst<T>& Hh = out[k++];
            Hh.n = 5; Hh.val = 0; for (int u = 0; u < 5; ++u) { Hh.idx[u] = -1; Hh.dv[u] = 0; }
            Inst<T>& Vv = out[k++];
            Vv.n = 5; Vv.val = 0; for (int u = 0; u < 5; ++u) { Vv.idx[u] = -1; Vv.dv[u] = 0; }
            if (in1) {
                const T mr = (T)edgeMaskR[e], mc = (T)edgeMaskC[e];
                const long er = e + 1, ed = e + W;
                Hh.val = w_g * ((B_I[e] - B_I[er]) * mr);
                Hh.idx[0] = e;         Hh.dv[0] = w_g * mr * (B_I_d1[e] - B_I_d0[er]);
                Hh.idx[1] = e - 1;     Hh.dv[1] = w_g * mr * B_I_d0[e];
                Hh.idx[2] = e - W;     Hh.dv[2] = w_g * mr * B_I_d2[e];
                Hh.idx[3] = er;        Hh.dv[3] = -(w_g * mr * B_I_d1[er]);
                Hh.idx[4] = er - W;    Hh.dv[4] = -(w_g * mr * B_I_d2[er]);
                Vv.val = w_g * ((B_I[e] - B_I[ed]) * mc);
                Vv.idx[0] = e;         Vv.dv[0] = w_g * mc * (B_I_d1[e] - B_I_d2[ed]);
                Vv.idx[1] = e - 1;     Vv.dv[1] = w_g * mc * B_I_d0[e];
                Vv.idx[2] = e - W;     Vv.dv[2] = w_g * mc * B_I_d2[e];
                Vv.idx[3] = ed;        Vv.dv[3] = -(w_g * mc * B_I_d1[ed]);
                Vv.idx[4] = ed - 1;    Vv.dv[4] = -(w_g * mc * B_I_d0[ed]);
            }
        }
        {   // regularisation (:83-90): 4 p(0,0) - (p(-1,0)+p(0,-1)+p(1,0)+p(0,1)),  p(off) = ((i-u_x)/f_x d, (j-u_y)/f_y d, d)
            const bool v = in1 && validArr[e] == T(1);
            const long nb[5] = {e, e - 1, e - W, e + 1, e + W};
            const long ox[5] = {0, -1, 0, 1, 0}, oy[5] = {0, 0, -1, 0, 1};
            for (int c = 0; c < 3; ++c) {
                Inst<T>& I = out[k++];
                I.n = 5; I.val = 0;
                for (int u = 0; u < 5; ++u) { I.idx[u] = v ? nb[u] : -1; I.dv[u] = 0; }
                if (!v) continue;
                T acc = 0;
                for (int u = 0; u < 5; ++u) {
                    const T i = (T)(x + ox[u]), j = (T)(y + oy[u]);
                    const T coef = (c == 0) ? (i - u_x) / f_x : (c == 1) ? (j - u_y) / f_y : T(1);
                    const T wgt = (u == 0) ? T(4) : T(-1);
                    acc += wgt * (coef * X[nb[u]]);
                    I.dv[u] = w_s * wgt * coef;
                }
                I.val = w_s * acc;
            }
        }
        return k;
    }
};

}  // namespace oracle
