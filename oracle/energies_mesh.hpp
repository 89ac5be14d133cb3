// TEST INFRASTRUCTURE ONLY (CPU oracle) -- never linked into, imported by, or called from the product path.
//
// The three mesh examples of the reference beyond ARAP, restated per residual (vertex residuals in evalCentered, hyperedge
// residuals in evalEdge) with dual numbers where the residual is not linear.  Conventions as in energies.hpp.
#pragma once
#include "dual.hpp"
#include "solver.hpp"

namespace oracle {

// ------------------------------------------------------------------------------------------------
// examples/cotangent_mesh_smoothing/cotangent_mesh_smoothing.t:1-33
template <class T>
struct CotangentSmoothing : Energy<T> {
    long N; int nE = 0;
    T* X = nullptr; const T* A = nullptr; const int* v[4] = {nullptr, nullptr, nullptr, nullptr};
    T w_fit = 0, w_reg = 0;
    CotangentSmoothing(const unsigned* dims) : N(dims[0]) {
        this->usePreconditioner = true; this->usesGraph = true;                  // (:12)
        this->addUnknown(N, 3);                                                  // X (:5)
    }
    void bind(void** p) override {
        w_fit = (T) * (const float*)p[0]; w_reg = (T) * (const float*)p[1]; X = (T*)p[2]; A = (const T*)p[3];
        nE = *(const int*)p[4]; for (int j = 0; j < 4; ++j) v[j] = (const int*)p[5 + j];
    }
    T* unknownPtr(int) override { return X; }
    long nCentered() const override { return N; }
    long nEdges() const override { return nE; }
    int evalCentered(long e, Inst<T>* out) const override {                       // Energy(w_fitSqrt*(X(0) - A(0))) (:22)
        for (int c = 0; c < 3; ++c) { Inst<T>& F = out[c]; F.n = 1; F.idx[0] = 3 * e + c; F.val = w_fit * (X[3 * e + c] - A[3 * e + c]); F.dv[0] = w_fit; }
        return 3;
    }
    typedef Dual<T, 12> D;
    static D dot(const D* a, const D* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
    static void unit(const D* p, const D* q, D* o) {                              // normalize (lib.t:54-56)
        D d[3] = {p[0] - q[0], p[1] - q[1], p[2] - q[2]};
        const D len = sqrt(dot(d, d));
        for (int c = 0; c < 3; ++c) o[c] = d[c] / len;
    }
    static D cot(const D* a, const D* b) {                                        // (:14-20)
        const D ab = dot(a, b);
        D disc = dot(a, a) * dot(b, b) - ab * ab;
        disc = select(disc.v > T(0), disc, D(T(0.0001)));
        return ab / sqrt(disc);
    }
    int evalEdge(long e, Inst<T>* out) const override {
        D x[4][3];
        for (int j = 0; j < 4; ++j) for (int c = 0; c < 3; ++c) x[j][c] = D::var(X[3 * (long)v[j][e] + c], 3 * j + c);
        D a[3], b[3], c[3], d[3];
        unit(x[0], x[2], a); unit(x[1], x[2], b); unit(x[0], x[3], c); unit(x[1], x[3], d);      // (:24-27)
        D w = (cot(a, b) + cot(c, d)) * T(0.5);                                                   // (:30)
        w = sqrt(select(w.v > T(0), w, D(T(0.0001))));                                            // (:31)
        for (int k = 0; k < 3; ++k) {                                                             // Energy(w_regSqrt*w*(X(G.v1) - X(G.v0))) (:32)
            const D res = (w * (x[1][k] - x[0][k])) * w_reg;
            Inst<T>& R = out[k];
            R.n = 12;
            for (int j = 0; j < 4; ++j) for (int q = 0; q < 3; ++q) R.idx[3 * j + q] = 3 * (long)v[j][e] + q;
            R.val = res.v;
            for (int q = 0; q < 12; ++q) R.dv[q] = res.d[q];
        }
        return 3;
    }
};

// ------------------------------------------------------------------------------------------------
// examples/embedded_mesh_deformation/embedded_mesh_deformation.t:1-31
template <class T>
struct EmbeddedDeformation : Energy<T> {
    long N; int nE = 0;
    T *Offset = nullptr, *Rot = nullptr; const T *Ur = nullptr, *Cons = nullptr; const int *v0 = nullptr, *v1 = nullptr;
    T w_fit = 0, w_reg = 0, w_rot = 0;
    EmbeddedDeformation(const unsigned* dims) : N(dims[0]) {
        this->usePreconditioner = true; this->usesGraph = true;                  // (:11)
        this->addUnknown(N, 3); this->addUnknown(N, 9);                          // Offset, RotMatrix (:6-7)
    }
    void bind(void** p) override {
        w_fit = (T) * (const float*)p[0]; w_reg = (T) * (const float*)p[1]; w_rot = (T) * (const float*)p[2];
        Offset = (T*)p[3]; Rot = (T*)p[4]; Ur = (const T*)p[5]; Cons = (const T*)p[6];
        nE = *(const int*)p[7]; v0 = (const int*)p[8]; v1 = (const int*)p[9];
    }
    T* unknownPtr(int img) override { return img == 0 ? Offset : Rot; }
    long nCentered() const override { return N; }
    long nEdges() const override { return nE; }
    int evalCentered(long e, Inst<T>* out) const override {
        const long offR = this->unkOffset[1];
        int k = 0;
        const bool valid = Cons[3 * e] >= T(-999999.9);                           // (:15)
        for (int c = 0; c < 3; ++c) {                                             // (:14-16)
            Inst<T>& F = out[k++]; F.n = 1; F.idx[0] = 3 * e + c;
            F.val = valid ? w_fit * (Offset[3 * e + c] - Cons[3 * e + c]) : T(0); F.dv[0] = valid ? w_fit : T(0);
        }
        const T* R = Rot + 9 * e;
        const int col[3][3] = {{0, 3, 6}, {1, 4, 7}, {2, 5, 8}};                  // c0, c1, c2 (:19-22)
        const int pairs[3][2] = {{0, 1}, {0, 2}, {1, 2}};
        for (int q = 0; q < 3; ++q) {                                             // w_rot * Dot3(ci, cj) (:23-25)
            const int* a = col[pairs[q][0]]; const int* b = col[pairs[q][1]];
            Inst<T>& I = out[k++]; I.n = 6; I.val = 0;
            for (int m = 0; m < 3; ++m) { I.val += R[a[m]] * R[b[m]]; I.idx[m] = offR + 9 * e + a[m]; I.dv[m] = w_rot * R[b[m]]; I.idx[3 + m] = offR + 9 * e + b[m]; I.dv[3 + m] = w_rot * R[a[m]]; }
            I.val *= w_rot;
        }
        for (int q = 0; q < 3; ++q) {                                             // w_rot * (Dot3(ci, ci) - 1) (:26-28)
            const int* a = col[q];
            Inst<T>& I = out[k++]; I.n = 3; T s = 0;
            for (int m = 0; m < 3; ++m) { s += R[a[m]] * R[a[m]]; I.idx[m] = offR + 9 * e + a[m]; I.dv[m] = w_rot * T(2) * R[a[m]]; }
            I.val = w_rot * (s - T(1));
        }
        return k;
    }
    int evalEdge(long e, Inst<T>* out) const override {                           // (:30-32), Matrix3x3Mul lib.t:39-44
        const long a = v0[e], b = v1[e], offR = this->unkOffset[1];
        const T u[3] = {Ur[3 * b] - Ur[3 * a], Ur[3 * b + 1] - Ur[3 * a + 1], Ur[3 * b + 2] - Ur[3 * a + 2]};
        for (int c = 0; c < 3; ++c) {
            const T* row = Rot + 9 * a + 3 * c;
            Inst<T>& R = out[c];
            R.n = 5;
            R.idx[0] = 3 * b + c; R.dv[0] = w_reg; R.idx[1] = 3 * a + c; R.dv[1] = -w_reg;
            for (int m = 0; m < 3; ++m) { R.idx[2 + m] = offR + 9 * a + 3 * c + m; R.dv[2 + m] = -w_reg * u[m]; }
            R.val = w_reg * ((Offset[3 * b + c] - Offset[3 * a + c]) - (row[0] * u[0] + row[1] * u[1] + row[2] * u[2]));
        }
        return 3;
    }
};

// ------------------------------------------------------------------------------------------------
// examples/robust_nonrigid_alignment/robust_nonrigid_alignment.t:1-27
template <class T>
struct RobustAlignment : Energy<T> {
    long N; int nE = 0;
    T *Offset = nullptr, *Angle = nullptr, *Rw = nullptr; const T *Ur = nullptr, *Cons = nullptr, *Nrm = nullptr; const int *v0 = nullptr, *v1 = nullptr;
    T w_fit = 0, w_reg = 0;
    RobustAlignment(const unsigned* dims) : N(dims[0]) {
        this->usePreconditioner = true; this->usesGraph = true;                  // (:13)
        this->addUnknown(N, 3); this->addUnknown(N, 3); this->addUnknown(N, 1);  // Offset, Angle, RobustWeights (:6-8)
    }
    void bind(void** p) override {
        w_fit = (T) * (const float*)p[0]; w_reg = (T) * (const float*)p[1];
        Offset = (T*)p[2]; Angle = (T*)p[3]; Rw = (T*)p[4]; Ur = (const T*)p[5]; Cons = (const T*)p[6]; Nrm = (const T*)p[7];
        nE = *(const int*)p[8]; v0 = (const int*)p[9]; v1 = (const int*)p[10];
    }
    T* unknownPtr(int img) override { return img == 0 ? Offset : img == 1 ? Angle : Rw; }
    long nCentered() const override { return N; }
    long nEdges() const override { return nE; }
    int evalCentered(long e, Inst<T>* out) const override {
        const long offW = this->unkOffset[2];
        const T rw = Rw[e];
        T nd = 0;
        for (int c = 0; c < 3; ++c) nd += Nrm[3 * e + c] * (Offset[3 * e + c] - Cons[3 * e + c]);       // ConstraintNormals(0):dot(Offset(0) - Constraints(0)) (:17)
        int k = 0;
        for (int c = 0; c < 3; ++c) {          // greatereq on the float3 gives one condition per component; the scalar e_fit is selected by each (:18-19, ad.t:327-349)
            const bool valid = Cons[3 * e + c] >= T(-999999.9);
            Inst<T>& F = out[k++];
            F.n = 4;
            for (int m = 0; m < 3; ++m) { F.idx[m] = 3 * e + m; F.dv[m] = valid ? w_fit * rw * Nrm[3 * e + m] : T(0); }
            F.idx[3] = offW + e; F.dv[3] = valid ? w_fit * nd : T(0);
            F.val = valid ? w_fit * (rw * nd) : T(0);
        }
        for (int c = 0; c < 3; ++c) {          // w_conf (1 - rw^2), w_conf = 0.1 (:5, 22-24)
            const bool valid = Cons[3 * e + c] >= T(-999999.9);
            Inst<T>& F = out[k++];
            F.n = 1; F.idx[0] = offW + e;
            F.val = valid ? T(0.1) * (T(1) - rw * rw) : T(0); F.dv[0] = valid ? T(0.1) * (-T(2) * rw) : T(0);
        }
        return k;
    }
    int evalEdge(long e, Inst<T>* out) const override {                           // ARAP with Rotate3D (:26-27; lib.t:77-91)
        typedef Dual<T, 9> D;
        const long a = v0[e], b = v1[e], offA = this->unkOffset[1];
        D oa[3], ob[3], an[3]; T u[3];
        for (int c = 0; c < 3; ++c) { oa[c] = D::var(Offset[3 * a + c], c); ob[c] = D::var(Offset[3 * b + c], 3 + c); an[c] = D::var(Angle[3 * a + c], 6 + c); u[c] = Ur[3 * a + c] - Ur[3 * b + c]; }
        const D ca = cos(an[0]), cb = cos(an[1]), cg = cos(an[2]), sa = sin(an[0]), sb = sin(an[1]), sg = sin(an[2]);
        const D m[9] = {cg * cb, -sg * ca + cg * sb * sa, sg * sa + cg * sb * ca, sg * cb, cg * ca + sg * sb * sa, -cg * sa + sg * sb * ca, -sb, cb * sa, cb * ca};
        for (int c = 0; c < 3; ++c) {
            const D res = ((oa[c] - ob[c]) - (m[3 * c] * u[0] + m[3 * c + 1] * u[1] + m[3 * c + 2] * u[2])) * w_reg;
            Inst<T>& R = out[c];
            R.n = 9;
            for (int q = 0; q < 3; ++q) { R.idx[q] = 3 * a + q; R.idx[3 + q] = 3 * b + q; R.idx[6 + q] = offA + 3 * a + q; }
            R.val = res.v;
            for (int q = 0; q < 9; ++q) R.dv[q] = res.d[q];
        }
        return 3;
    }
};

}  // namespace oracle
