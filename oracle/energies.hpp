// TEST INFRASTRUCTURE ONLY (CPU oracle) -- never linked into, imported by, or called from the product path.
//
// The residual templates of the energy files this build supports, written once against dual numbers so
// value and partials come out together (the oracle's analogue of Opt differentiating the .t symbolically).
// Each struct cites the reference .t it restates.  Conventions (reference API/src/o.t):
//   * a residual whose stencil leaves the image is 0 unless the energy tests InBounds itself, in which
//     case the energy's own Select decides and out-of-range loads return 0 (o.t:1895-1936, 570-576);
//   * boolean factors select, they do not multiply (ad.t:683-699), so inf/NaN in a deselected branch is harmless;
//   * Param scalars are read from HOST pointers with their declared C type (float stays float in double
//     mode), arrays are opt_floatN AoS, x fastest (util.t:664-692, o.t:376-387).
#pragma once
#include "dual.hpp"
#include "solver.hpp"

namespace oracle {

// ------------------------------------------------------------------------------------------------
// examples/image_warping/image_warping.t:1-23
template <class T>
struct ImageWarping : Energy<T> {
    long W, H;
    T *Offset = nullptr, *Angle = nullptr; const T *UrShape = nullptr, *Constraints = nullptr, *Mask = nullptr;
    T w_fit = 0, w_reg = 0;
    ImageWarping(const unsigned* dims) : W(dims[0]), H(dims[1]) {
        this->usePreconditioner = true;                                   // image_warping.t:10
        this->addUnknown(W * H, 2); this->addUnknown(W * H, 1);           // Offset, Angle (:2-3)
    }
    void bind(void** p) override {
        Offset = (T*)p[0]; Angle = (T*)p[1]; UrShape = (const T*)p[2]; Constraints = (const T*)p[3]; Mask = (const T*)p[4];
        w_fit = (T) * (const float*)p[5]; w_reg = (T) * (const float*)p[6];
    }
    T* unknownPtr(int img) override { return img == 0 ? Offset : Angle; }
    long nCentered() const override { return W * H; }
    long rowWidth() const override { return W; }
    bool excluded(int, long e) const override { return Mask[e] != T(0); }   // Exclude(Not(eq(Mask(0,0),0))) (:11)
    bool excludedCentered(long e) const override { return Mask[e] != T(0); }
    int evalCentered(long e, Inst<T>* out) const override {
        typedef Dual<T, 5> D;
        const long x = e % W, y = e / W;
        const long offO = this->unkOffset[0], offA = this->unkOffset[1];
        static const int dirs[4][2] = {{1, 0}, {-1, 0}, {0, 1}, {0, -1}};   // Stencil (:14)
        int k = 0;
        for (int n = 0; n < 4; ++n) {
            const long nx = x + dirs[n][0], ny = y + dirs[n][1];
            const bool inb = nx >= 0 && nx < W && ny >= 0 && ny < H;
            const long ne = ny * W + nx;
            const T maskN = inb ? Mask[ne] : T(0);
            const bool valid = inb && maskN == T(0) && Mask[e] == T(0);        // (:17)
            D Ocx = D::var(Offset[2 * e], 0), Ocy = D::var(Offset[2 * e + 1], 1);
            D Onx = D::var(inb ? Offset[2 * ne] : T(0), 2), Ony = D::var(inb ? Offset[2 * ne + 1] : T(0), 3);
            D a = D::var(Angle[e], 4);
            const T dUx = UrShape[2 * e] - (inb ? UrShape[2 * ne] : T(0));
            const T dUy = UrShape[2 * e + 1] - (inb ? UrShape[2 * ne + 1] : T(0));
            D ca = cos(a), sa = sin(a);                                        // Rotate2D, lib.t:92-96
            D rx = ca * dUx - sa * dUy, ry = sa * dUx + ca * dUy;
            D ex = ((Ocx - Onx) - rx) * w_reg, ey = ((Ocy - Ony) - ry) * w_reg;   // (:15-16)
            D res[2] = {select(valid, ex, D(T(0))), select(valid, ey, D(T(0)))};
            for (int c = 0; c < 2; ++c) {
                Inst<T>& I = out[k++];
                I.n = 5;
                I.idx[0] = offO + 2 * e; I.idx[1] = offO + 2 * e + 1;
                I.idx[2] = inb ? offO + 2 * ne : -1; I.idx[3] = inb ? offO + 2 * ne + 1 : -1;
                I.idx[4] = offA + e;
                I.val = res[c].v;
                for (int u = 0; u < 5; ++u) I.dv[u] = res[c].d[u];
            }
        }
        const bool fitValid = Constraints[2 * e] >= T(0) && Constraints[2 * e + 1] >= T(0);   // All(greatereq(C,0)) (:22)
        for (int c = 0; c < 2; ++c) {                                                          // (:21-23)
            Inst<T>& I = out[k++];
            I.n = 1; I.idx[0] = offO + 2 * e + c;
            I.val = fitValid ? w_fit * (Offset[2 * e + c] - Constraints[2 * e + c]) : T(0);
            I.dv[0] = fitValid ? w_fit : T(0);
        }
        return k;
    }
};

// ------------------------------------------------------------------------------------------------
// examples/poisson_image_editing/poisson_image_editing.t:1-13
template <class T>
struct Poisson : Energy<T> {
    long W, H;
    T* X = nullptr; const T *Tg = nullptr, *M = nullptr;
    Poisson(const unsigned* dims) : W(dims[0]), H(dims[1]) {
        this->usePreconditioner = false;                                  // poisson_image_editing.t:5
        this->addUnknown(W * H, 4);
    }
    void bind(void** p) override { X = (T*)p[0]; Tg = (const T*)p[1]; M = (const T*)p[2]; }
    T* unknownPtr(int) override { return X; }
    long nCentered() const override { return W * H; }
    long rowWidth() const override { return W; }
    bool excluded(int, long e) const override { return M[e] != T(0); }    // (:8)
    bool excludedCentered(long e) const override { return M[e] != T(0); }
    int evalCentered(long e, Inst<T>* out) const override {
        const long x = e % W, y = e / W;
        static const int dirs[4][2] = {{1, 0}, {-1, 0}, {0, 1}, {0, -1}};
        int k = 0;
        for (int n = 0; n < 4; ++n) {
            const long nx = x + dirs[n][0], ny = y + dirs[n][1];
            const bool inb = nx >= 0 && nx < W && ny >= 0 && ny < H;
            const long ne = ny * W + nx;
            for (int c = 0; c < 4; ++c) {
                Inst<T>& I = out[k++];
                I.n = 2; I.idx[0] = 4 * e + c; I.idx[1] = inb ? 4 * ne + c : -1;
                if (inb) {   // Select(InBounds(x,y), (X-Xn)-(T-Tn), 0) (:10-12)
                    I.val = (X[4 * e + c] - X[4 * ne + c]) - (Tg[4 * e + c] - Tg[4 * ne + c]);
                    I.dv[0] = T(1); I.dv[1] = T(-1);
                } else { I.val = 0; I.dv[0] = 0; I.dv[1] = 0; }
            }
        }
        return k;
    }
};

// ------------------------------------------------------------------------------------------------
// tests/minimal/laplacian.t:1-7 (also tests/create_delete_cycle).  No InBounds in the energy, so a
// residual reaching outside the image is zero (o.t:1930-1933).  No UsePreconditioner call -> false.
template <class T>
struct Laplacian : Energy<T> {
    long W, H;
    T* X = nullptr; const T* A = nullptr;
    Laplacian(const unsigned* dims) : W(dims[0]), H(dims[1]) { this->addUnknown(W * H, 1); }
    void bind(void** p) override { X = (T*)p[0]; A = (const T*)p[1]; }
    T* unknownPtr(int) override { return X; }
    long nCentered() const override { return W * H; }
    long rowWidth() const override { return W; }
    int evalCentered(long e, Inst<T>* out) const override {
        const long x = e % W, y = e / W;
        Inst<T>& F = out[0];
        F.n = 1; F.idx[0] = e; F.val = T(0.2) * (X[e] - A[e]); F.dv[0] = T(0.2);
        const bool inx = x + 1 < W, iny = y + 1 < H;
        Inst<T>& Rx = out[1];
        Rx.n = 2; Rx.idx[0] = e; Rx.idx[1] = inx ? e + 1 : -1;
        Rx.val = inx ? X[e] - X[e + 1] : T(0); Rx.dv[0] = inx ? T(1) : T(0); Rx.dv[1] = inx ? T(-1) : T(0);
        Inst<T>& Ry = out[2];
        Ry.n = 2; Ry.idx[0] = e; Ry.idx[1] = iny ? e + W : -1;
        Ry.val = iny ? X[e] - X[e + W] : T(0); Ry.dv[0] = iny ? T(1) : T(0); Ry.dv[1] = iny ? T(-1) : T(0);
        return 3;
    }
};

// ------------------------------------------------------------------------------------------------
// tests/minimal_graph_only/curveFitting.t:1-9 -- the reference's only known-answer test
// (main.cpp:43-61, 88-90: unknowns (99.7,101.6) -> (100,102)).
template <class T>
struct CurveFitting : Energy<T> {
    long N, U;
    T* funcParams = nullptr; const T* data = nullptr; long nE = 0; const int *dIdx = nullptr, *pIdx = nullptr;
    CurveFitting(const unsigned* dims) : N(dims[0]), U(dims[1]) {
        this->usePreconditioner = true; this->usesGraph = true;
        this->addUnknown(U, 2);
    }
    void bind(void** p) override {
        funcParams = (T*)p[0]; data = (const T*)p[1];
        nE = *(const int*)p[2]; dIdx = (const int*)p[3]; pIdx = (const int*)p[4];   // Graph("G",2,"d",{N},3,"p",{U},4)
    }
    T* unknownPtr(int) override { return funcParams; }
    long nCentered() const override { return U; }   // zero-valued stand-in residual (o.t:1972-1982)
    int evalCentered(long, Inst<T>*) const override { return 0; }
    long nEdges() const override { return nE; }
    int evalEdge(long e, Inst<T>* out) const override {
        typedef Dual<T, 2> D;
        const long d = dIdx[e], pp = pIdx[e];
        const T x = data[2 * d], y = data[2 * d + 1];
        D a = D::var(funcParams[2 * pp], 0), b = D::var(funcParams[2 * pp + 1], 1);
        D res = y - (a * cos(b * x) + b * sin(a * x));   // curveFitting.t:9
        Inst<T>& I = out[0];
        I.n = 2; I.idx[0] = 2 * pp; I.idx[1] = 2 * pp + 1; I.val = res.v; I.dv[0] = res.d[0]; I.dv[1] = res.d[1];
        return 1;
    }
};

// ------------------------------------------------------------------------------------------------
// examples/arap_mesh_deformation/arap_mesh_deformation.t:1-18
template <class T>
struct Arap : Energy<T> {
    long N;
    T *Offset = nullptr, *Angle = nullptr; const T *UrShape = nullptr, *Constraints = nullptr;
    T w_fit = 0, w_reg = 0; long nE = 0; const int *v0 = nullptr, *v1 = nullptr;
    Arap(const unsigned* dims) : N(dims[0]) {
        this->usePreconditioner = true; this->usesGraph = true;            // (:9)
        this->addUnknown(N, 3); this->addUnknown(N, 3);                    // Offset, Angle (:4-5)
    }
    void bind(void** p) override {
        w_fit = (T) * (const float*)p[0]; w_reg = (T) * (const float*)p[1];
        Offset = (T*)p[2]; Angle = (T*)p[3]; UrShape = (const T*)p[4]; Constraints = (const T*)p[5];
        nE = *(const int*)p[6]; v0 = (const int*)p[7]; v1 = (const int*)p[8];
    }
    T* unknownPtr(int img) override { return img == 0 ? Offset : Angle; }
    long nCentered() const override { return N; }
    int evalCentered(long e, Inst<T>* out) const override {   // fitting (:12-14)
        const bool valid = Constraints[3 * e] >= T(-999999.9);
        for (int c = 0; c < 3; ++c) {
            Inst<T>& I = out[c];
            I.n = 1; I.idx[0] = this->unkOffset[0] + 3 * e + c;
            I.val = valid ? w_fit * (Offset[3 * e + c] - Constraints[3 * e + c]) : T(0);
            I.dv[0] = valid ? w_fit : T(0);
        }
        return 3;
    }
    long nEdges() const override { return nE; }
    int evalEdge(long e, Inst<T>* out) const override {       // regularisation (:17-18)
        typedef Dual<T, 9> D;
        const long a0 = v0[e], a1 = v1[e];
        D O0[3], O1[3], A[3];
        for (int c = 0; c < 3; ++c) {
            O0[c] = D::var(Offset[3 * a0 + c], c);
            O1[c] = D::var(Offset[3 * a1 + c], 3 + c);
            A[c] = D::var(Angle[3 * a0 + c], 6 + c);
        }
        T v[3];
        for (int c = 0; c < 3; ++c) v[c] = UrShape[3 * a0 + c] - UrShape[3 * a1 + c];
        // Rotate3D, lib.t:77-91
        D CosAlpha = cos(A[0]), CosBeta = cos(A[1]), CosGamma = cos(A[2]);
        D SinAlpha = sin(A[0]), SinBeta = sin(A[1]), SinGamma = sin(A[2]);
        D m[9] = {CosGamma * CosBeta,
                  -(SinGamma * CosAlpha) + CosGamma * SinBeta * SinAlpha,
                  SinGamma * SinAlpha + CosGamma * SinBeta * CosAlpha,
                  SinGamma * CosBeta,
                  CosGamma * CosAlpha + SinGamma * SinBeta * SinAlpha,
                  -(CosGamma * SinAlpha) + SinGamma * SinBeta * CosAlpha,
                  -SinBeta,
                  CosBeta * SinAlpha,
                  CosBeta * CosAlpha};
        for (int c = 0; c < 3; ++c) {
            D rot = m[3 * c] * v[0] + m[3 * c + 1] * v[1] + m[3 * c + 2] * v[2];
            D res = ((O0[c] - O1[c]) - rot) * w_reg;
            Inst<T>& I = out[c];
            I.n = 9;
            for (int u = 0; u < 3; ++u) {
                I.idx[u] = this->unkOffset[0] + 3 * a0 + u;
                I.idx[3 + u] = this->unkOffset[0] + 3 * a1 + u;
                I.idx[6 + u] = this->unkOffset[1] + 3 * a0 + u;
            }
            I.val = res.v;
            for (int u = 0; u < 9; ++u) I.dv[u] = res.d[u];
        }
        return 3;
    }
};

}  // namespace oracle
