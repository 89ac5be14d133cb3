// TEST INFRASTRUCTURE ONLY (CPU oracle) -- never linked into, imported by, or called from the product path.
//
// Three further image examples of the reference, restated per residual with explicit values and
// partials (hand-derived where the residual is linear, dual numbers where it is not).  Conventions as in energies.hpp.
#pragma once
#include "dual.hpp"
#include "solver.hpp"

namespace oracle {

// ------------------------------------------------------------------------------------------------
// examples/optical_flow/optical_flow.t:1-19
template <class T>
struct OpticalFlow : Energy<T> {
    long W, H;
    T* X = nullptr; const T *I = nullptr, *Ihat = nullptr, *Idx = nullptr, *Idy = nullptr;
    T w_fit = 0, w_reg = 0;
    OpticalFlow(const unsigned* dims) : W(dims[0]), H(dims[1]) {
        this->usePreconditioner = false;                                    // optical_flow.t:12
        this->addUnknown(W * H, 2);                                         // X (:4)
    }
    void bind(void** p) override {
        w_fit = (T) * (const float*)p[0]; w_reg = (T) * (const float*)p[1];
        X = (T*)p[2]; I = (const T*)p[3]; Ihat = (const T*)p[4]; Idx = (const T*)p[5]; Idy = (const T*)p[6];
    }
    T* unknownPtr(int) override { return X; }
    long nCentered() const override { return W * H; }
    long rowWidth() const override { return W; }
    T get(const T* im, long x, long y) const { return (x >= 0 && x < W && y >= 0 && y < H) ? im[y * W + x] : T(0); }   // o.t:570-576
    T sample(const T* im, T x, T y) const {                                                                            // Image:sample, o.t:578-589
        const long x0 = (long)std::floor(x), x1 = (long)std::ceil(x), y0 = (long)std::floor(y), y1 = (long)std::ceil(y);
        const T xn = x - T(x0), yn = y - T(y0);
        const T top = (T(1) - xn) * get(im, x0, y0) + xn * get(im, x1, y0);
        const T bot = (T(1) - xn) * get(im, x0, y1) + xn * get(im, x1, y1);
        return (T(1) - yn) * top + yn * bot;
    }
    int evalCentered(long e, Inst<T>* out) const override {
        const long x = e % W, y = e / W;
        int k = 0;
        {   // e_fit = w_fit (I(0,0) - I_hat(i + X.x, j + X.y))  (:14-15); d I_hat = (I_hat_dx, I_hat_dy) sampled at the same point (o.t:2494-2498)
            const T px = T(x) + X[2 * e], py = T(y) + X[2 * e + 1];
            Inst<T>& F = out[k++];
            F.n = 2; F.idx[0] = 2 * e; F.idx[1] = 2 * e + 1;
            F.val = w_fit * (I[e] - sample(Ihat, px, py));
            F.dv[0] = -w_fit * sample(Idx, px, py); F.dv[1] = -w_fit * sample(Idy, px, py);
        }
        static const int dirs[4][2] = {{1, 0}, {-1, 0}, {0, 1}, {0, -1}};                 // (:17)
        for (int n = 0; n < 4; ++n) {
            const long nx = x + dirs[n][0], ny = y + dirs[n][1];
            const bool inb = nx >= 0 && nx < W && ny >= 0 && ny < H;
            const long ne = ny * W + nx;
            for (int c = 0; c < 2; ++c) {                                                  // Select(InBounds(nx,ny), w_reg (X - Xn), 0) (:18-19)
                Inst<T>& R = out[k++];
                R.n = 2; R.idx[0] = 2 * e + c; R.idx[1] = inb ? 2 * ne + c : -1;
                R.val = inb ? w_reg * (X[2 * e + c] - X[2 * ne + c]) : T(0);
                R.dv[0] = inb ? w_reg : T(0); R.dv[1] = inb ? -w_reg : T(0);
            }
        }
        return k;
    }
};

// ------------------------------------------------------------------------------------------------
// examples/intrinsic_image_decomposition/intrinsic_image_decomposition.t:1-31
template <class T>
struct IntrinsicImage : Energy<T> {
    long W, H;
    T *r = nullptr, *s = nullptr; const T* target = nullptr;
    T w_fit = 0, w_regA = 0, w_regS = 0, pNorm = 0;
    IntrinsicImage(const unsigned* dims) : W(dims[0]), H(dims[1]) {
        this->addUnknown(W * H, 3); this->addUnknown(W * H, 1);              // r (:6), s (:9); no UsePreconditioner call -> false
    }
    void bind(void** p) override {
        w_fit = (T) * (const float*)p[0]; w_regA = (T) * (const float*)p[1]; w_regS = (T) * (const float*)p[2];
        pNorm = *(const T*)p[3];                                              // Param("pNorm", opt_float, 3) (:5)
        r = (T*)p[4]; target = (const T*)p[5]; s = (T*)p[6];
    }
    T* unknownPtr(int img) override { return img == 0 ? r : s; }
    long nCentered() const override { return W * H; }
    long rowWidth() const override { return W; }
    int evalCentered(long e, Inst<T>* out) const override {
        const long x = e % W, y = e / W, offS = this->unkOffset[1];
        static const int dirs[4][2] = {{1, 0}, {-1, 0}, {0, 1}, {0, -1}};
        int k = 0;
        for (int n = 0; n < 4; ++n) {                                           // albedo: L_p(diff, diff_const, pNorm) (:12-21, lib.t:106-114)
            const long nx = x + dirs[n][0], ny = y + dirs[n][1];
            const bool inb = nx >= 0 && nx < W && ny >= 0 && ny < H;
            const long ne = ny * W + nx;
            T d2 = 0;
            for (int c = 0; c < 3; ++c) { const T d = r[3 * e + c] - (inb ? r[3 * ne + c] : T(0)); d2 += d * d; }
            const T sqrtC = std::sqrt(std::pow(std::sqrt(d2) + T(0.0000001), pNorm - T(2)));   // the ComputedArray: constant w.r.t. the unknowns
            for (int c = 0; c < 3; ++c) {
                Inst<T>& R = out[k++];
                R.n = 2; R.idx[0] = 3 * e + c; R.idx[1] = inb ? 3 * ne + c : -1;
                R.val = inb ? w_regA * (sqrtC * (r[3 * e + c] - r[3 * ne + c])) : T(0);
                R.dv[0] = inb ? w_regA * sqrtC : T(0); R.dv[1] = inb ? -(w_regA * sqrtC) : T(0);
            }
        }
        for (int n = 0; n < 4; ++n) {                                           // shading (:24-28)
            const long nx = x + dirs[n][0], ny = y + dirs[n][1];
            const bool inb = nx >= 0 && nx < W && ny >= 0 && ny < H;
            const long ne = ny * W + nx;
            Inst<T>& R = out[k++];
            R.n = 2; R.idx[0] = offS + e; R.idx[1] = inb ? offS + ne : -1;
            R.val = inb ? w_regS * (s[e] - s[ne]) : T(0);
            R.dv[0] = inb ? w_regS : T(0); R.dv[1] = inb ? -w_regS : T(0);
        }
        for (int c = 0; c < 3; ++c) {                                           // fit: r + s - i (:30-31)
            Inst<T>& F = out[k++];
            F.n = 2; F.idx[0] = 3 * e + c; F.idx[1] = offS + e;
            F.val = w_fit * (r[3 * e + c] + s[e] - target[3 * e + c]);
            F.dv[0] = w_fit; F.dv[1] = w_fit;
        }
        return k;
    }
};

// ------------------------------------------------------------------------------------------------
// examples/volumetric_mesh_deformation/volumetric_mesh_deformation.t:1-20
template <class T>
struct VolumetricMesh : Energy<T> {
    long W, H, Dz;
    T *Offset = nullptr, *Angle = nullptr; const T *Ur = nullptr, *Cons = nullptr;
    T w_fit = 0, w_reg = 0;
    VolumetricMesh(const unsigned* dims) : W(dims[0]), H(dims[1]), Dz(dims[2]) {
        this->usePreconditioner = true;                                       // (:9)
        this->addUnknown(W * H * Dz, 3); this->addUnknown(W * H * Dz, 3);     // Offset, Angle (:3-4)
    }
    void bind(void** p) override {
        Offset = (T*)p[0]; Angle = (T*)p[1]; Ur = (const T*)p[2]; Cons = (const T*)p[3];
        w_fit = (T) * (const float*)p[4]; w_reg = (T) * (const float*)p[5];
    }
    T* unknownPtr(int img) override { return img == 0 ? Offset : Angle; }
    long nCentered() const override { return W * H * Dz; }
    int evalCentered(long e, Inst<T>* out) const override {
        typedef Dual<T, 9> D;
        const long x = e % W, y = (e / W) % H, z = e / (W * H), offA = this->unkOffset[1];
        int k = 0;
        const bool valid = Cons[3 * e] >= T(-999999.9);                        // greatereq(Constraints(0,0,0)(0), -999999.9) (:13)
        for (int c = 0; c < 3; ++c) {                                          // (:12-14)
            Inst<T>& F = out[k++];
            F.n = 1; F.idx[0] = 3 * e + c;
            F.val = valid ? w_fit * (Offset[3 * e + c] - Cons[3 * e + c]) : T(0);
            F.dv[0] = valid ? w_fit : T(0);
        }
        static const int dirs[6][3] = {{1, 0, 0}, {-1, 0, 0}, {0, 1, 0}, {0, -1, 0}, {0, 0, 1}, {0, 0, -1}};   // (:17)
        for (int n = 0; n < 6; ++n) {
            const long nx = x + dirs[n][0], ny = y + dirs[n][1], nz = z + dirs[n][2];
            const bool inb = nx >= 0 && nx < W && ny >= 0 && ny < H && nz >= 0 && nz < Dz;
            const long ne = (nz * H + ny) * W + nx;
            D oc[3], on[3], a[3]; T u[3];
            for (int c = 0; c < 3; ++c) {
                oc[c] = D::var(Offset[3 * e + c], c); on[c] = D::var(inb ? Offset[3 * ne + c] : T(0), 3 + c); a[c] = D::var(Angle[3 * e + c], 6 + c);
                u[c] = Ur[3 * e + c] - (inb ? Ur[3 * ne + c] : T(0));
            }
            // Rotate3D (lib.t:77-91)
            const D ca = cos(a[0]), cb = cos(a[1]), cg = cos(a[2]), sa = sin(a[0]), sb = sin(a[1]), sg = sin(a[2]);
            const D m[9] = {cg * cb, -sg * ca + cg * sb * sa, sg * sa + cg * sb * ca,
                            sg * cb, cg * ca + sg * sb * sa, -cg * sa + sg * sb * ca,
                            -sb, cb * sa, cb * ca};
            for (int c = 0; c < 3; ++c) {                                      // (:18-20)
                const D rot = m[3 * c] * u[0] + m[3 * c + 1] * u[1] + m[3 * c + 2] * u[2];
                const D res = select(inb, ((oc[c] - on[c]) - rot) * w_reg, D(T(0)));
                Inst<T>& R = out[k++];
                R.n = 9;
                for (int q = 0; q < 3; ++q) { R.idx[q] = 3 * e + q; R.idx[3 + q] = inb ? 3 * ne + q : -1; R.idx[6 + q] = offA + 3 * e + q; }
                R.val = res.v;
                for (int q = 0; q < 9; ++q) R.dv[q] = res.d[q];
            }
        }
        return k;
    }
};

}  // namespace oracle
