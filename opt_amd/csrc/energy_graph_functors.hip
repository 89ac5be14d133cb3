// The mesh energies served by the graph functor engine (graph_engine.h): cotangent_mesh_smoothing, embedded_mesh_deformation,
// robust_nonrigid_alignment -- the three graph examples of the reference beyond ARAP.  Each functor restates the residuals of its
// reference .t once, against a scalar type S that the engine instantiates as T or as a dual number.
#include "graph_engine.h"

namespace optamd {
namespace {

template <class S> __device__ __forceinline__ S dot3(const S* a, const S* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }

// ------------------------------------------------------------------------------------------------------------------
// examples/cotangent_mesh_smoothing/cotangent_mesh_smoothing.t:1-33.  X float3 per vertex; fit w_fit (X - A); per half-edge
// (v0 current, v1 neighbour, v2 / v3 the previous / next neighbour in v0's ring) the cotangent-weighted Laplacian term
// w_reg sqrt(w) (X_v1 - X_v0), w = (cot(a,b) + cot(c,d)) / 2 with the guards of the .t (:15-19, :29-31); the weight is
// differentiated like everything else.  UsePreconditioner(true).
template <class T>
struct CotangentG {
    static constexpr int NIMG = 1, K = 3, V = 4, RV = 3, RE = 3;
    static constexpr __host__ __device__ int imgOf(int) { return 0; }
    static constexpr __host__ __device__ int chOf(int k) { return k; }
    static constexpr __host__ __device__ int channels(int) { return 3; }
    static constexpr __host__ __device__ bool edgeDepends(int, int, int) { return true; }     // the weight couples every component of all four vertices
    static int unknownParam(int) { return 2; }
    long N; int nE; const int* vidx[V]; const T* X[NIMG];
    const T* A; T w_fit, w_reg;
    void bindParams(void** p) {
        w_fit = (T) * (const float*)p[0]; w_reg = (T) * (const float*)p[1]; X[0] = (const T*)p[2]; A = (const T*)p[3];
        nE = *(const int*)p[4]; for (int j = 0; j < V; ++j) vidx[j] = (const int*)p[5 + j];     // Graph("G", 4, "v0", {N}, 5, ... "v3", {N}, 8)
    }
    template <class S, class C> __device__ __forceinline__ void vertexResiduals(const C& Xc, long v, S* r) const {
#pragma unroll
        for (int c = 0; c < 3; ++c) r[c] = w_fit * (Xc(c) - A[3 * v + c]);
    }
    template <class S> __device__ __forceinline__ static void normalized(const S* p, const S* q, S* out) {      // normalize(p - q), lib.t:54-56
        S d[3] = {p[0] - q[0], p[1] - q[1], p[2] - q[2]};
        const S len = sqrt(dot3(d, d));
        out[0] = d[0] / len; out[1] = d[1] / len; out[2] = d[2] / len;
    }
    template <class S> __device__ __forceinline__ static S cot(const S* u, const S* w) {                        // cotangent_mesh_smoothing.t:15-20
        const S ab = dot3(u, w);
        S disc = dot3(u, u) * dot3(w, w) - ab * ab;
        disc = valueOf(disc) > T(0) ? disc : S(T(0.0001));
        return ab / sqrt(disc);
    }
    template <class S, class C> __device__ __forceinline__ void edgeResiduals(const C& Xc, long, S* r) const {
        S x[4][3];
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int c = 0; c < 3; ++c) x[j][c] = Xc(j, c);
        S a[3], b[3], cc[3], d[3];
        normalized(x[0], x[2], a); normalized(x[1], x[2], b); normalized(x[0], x[3], cc); normalized(x[1], x[3], d);
        S w = T(0.5) * (cot(a, b) + cot(cc, d));
        w = sqrt(valueOf(w) > T(0) ? w : S(T(0.0001)));
#pragma unroll
        for (int c = 0; c < 3; ++c) r[c] = w_reg * (w * (x[1][c] - x[0][c]));
    }
};

// ------------------------------------------------------------------------------------------------------------------
// examples/embedded_mesh_deformation/embedded_mesh_deformation.t:1-31.  Unknowns Offset (float3) and RotMatrix (float9, row-major)
// per node; fit where Constraints.x >= -999999.9; six orthonormality residuals on the columns of RotMatrix; per half-edge
// (Offset_v1 - Offset_v0) - RotMatrix_v0 (UrShape_v1 - UrShape_v0)  (Matrix3x3Mul, lib.t:39-44).  UsePreconditioner(true).
template <class T>
struct EmbeddedG {
    static constexpr int NIMG = 2, K = 12, V = 2, RV = 9, RE = 3;
    static constexpr __host__ __device__ int imgOf(int k) { return k < 3 ? 0 : 1; }
    static constexpr __host__ __device__ int chOf(int k) { return k < 3 ? k : k - 3; }
    static constexpr __host__ __device__ int channels(int img) { return img == 0 ? 3 : 9; }
    // residual ri depends on Offset_ri of both vertices and on row ri of RotMatrix(v0)
    static constexpr __host__ __device__ bool edgeDepends(int ri, int j, int k) { return k < 3 ? k == ri : (j == 0 && (k - 3) / 3 == ri); }
    static int unknownParam(int img) { return img == 0 ? 3 : 4; }
    long N; int nE; const int* vidx[V]; const T* X[NIMG];
    const T *Ur, *Cons; T w_fit, w_reg, w_rot;
    void bindParams(void** p) {
        w_fit = (T) * (const float*)p[0]; w_reg = (T) * (const float*)p[1]; w_rot = (T) * (const float*)p[2];
        X[0] = (const T*)p[3]; X[1] = (const T*)p[4]; Ur = (const T*)p[5]; Cons = (const T*)p[6];
        nE = *(const int*)p[7]; vidx[0] = (const int*)p[8]; vidx[1] = (const int*)p[9];
    }
    template <class S, class C> __device__ __forceinline__ void vertexResiduals(const C& Xc, long v, S* r) const {
        const bool valid = Cons[3 * v] >= T(-999999.9);
#pragma unroll
        for (int c = 0; c < 3; ++c) { const S e = w_fit * (Xc(c) - Cons[3 * v + c]); r[c] = valid ? e : S(T(0)); }
        S R[9];
#pragma unroll
        for (int i = 0; i < 9; ++i) R[i] = Xc(3 + i);
        const S c0[3] = {R[0], R[3], R[6]}, c1[3] = {R[1], R[4], R[7]}, c2[3] = {R[2], R[5], R[8]};
        r[3] = w_rot * dot3(c0, c1); r[4] = w_rot * dot3(c0, c2); r[5] = w_rot * dot3(c1, c2);
        r[6] = w_rot * (dot3(c0, c0) - T(1)); r[7] = w_rot * (dot3(c1, c1) - T(1)); r[8] = w_rot * (dot3(c2, c2) - T(1));
    }
    template <class S, class C> __device__ __forceinline__ void edgeResiduals(const C& Xc, long e, S* r) const {
        const long a = vidx[0][e], b = vidx[1][e];
        const T u[3] = {Ur[3 * b] - Ur[3 * a], Ur[3 * b + 1] - Ur[3 * a + 1], Ur[3 * b + 2] - Ur[3 * a + 2]};
#pragma unroll
        for (int c = 0; c < 3; ++c)
            r[c] = w_reg * ((Xc(1, c) - Xc(0, c)) - (Xc(0, 3 + 3 * c) * u[0] + Xc(0, 4 + 3 * c) * u[1] + Xc(0, 5 + 3 * c) * u[2]));
    }
};

// ------------------------------------------------------------------------------------------------------------------
// examples/robust_nonrigid_alignment/robust_nonrigid_alignment.t:1-27.  Unknowns Offset, Angle (float3) and RobustWeights (float).
// greatereq(Constraints(0), -999999.9) is a 3-vector of conditions (operators broadcast over vectors, ad.t:327-349), so the
// scalar point-to-plane term and the weight penalty each appear three times, gated by the x, y and z test; w_conf = 0.1.
// Per half-edge the ARAP term with Rotate3D (lib.t:77-91).  UsePreconditioner(true).
template <class T>
struct RobustG {
    static constexpr int NIMG = 3, K = 7, V = 2, RV = 6, RE = 3;
    static constexpr __host__ __device__ int imgOf(int k) { return k < 3 ? 0 : k < 6 ? 1 : 2; }
    static constexpr __host__ __device__ int chOf(int k) { return k < 3 ? k : k < 6 ? k - 3 : 0; }
    static constexpr __host__ __device__ int channels(int img) { return img == 2 ? 1 : 3; }
    static constexpr __host__ __device__ bool edgeDepends(int ri, int j, int k) { return k < 3 ? k == ri : (j == 0 && k < 6); }
    static int unknownParam(int img) { return 2 + img; }
    long N; int nE; const int* vidx[V]; const T* X[NIMG];
    const T *Ur, *Cons, *Nrm; T w_fit, w_reg;
    void bindParams(void** p) {
        w_fit = (T) * (const float*)p[0]; w_reg = (T) * (const float*)p[1];
        X[0] = (const T*)p[2]; X[1] = (const T*)p[3]; X[2] = (const T*)p[4]; Ur = (const T*)p[5]; Cons = (const T*)p[6]; Nrm = (const T*)p[7];
        nE = *(const int*)p[8]; vidx[0] = (const int*)p[9]; vidx[1] = (const int*)p[10];
    }
    template <class S, class C> __device__ __forceinline__ void vertexResiduals(const C& Xc, long v, S* r) const {
        const S rw = Xc(6);
        const S fit = rw * (Nrm[3 * v] * (Xc(0) - Cons[3 * v]) + Nrm[3 * v + 1] * (Xc(1) - Cons[3 * v + 1]) + Nrm[3 * v + 2] * (Xc(2) - Cons[3 * v + 2]));
        const S conf = T(1) - rw * rw;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const bool valid = Cons[3 * v + c] >= T(-999999.9);
            r[c] = valid ? w_fit * fit : S(T(0));
            r[3 + c] = valid ? T(0.1) * conf : S(T(0));
        }
    }
    template <class S, class C> __device__ __forceinline__ void edgeResiduals(const C& Xc, long e, S* r) const {
        const long a = vidx[0][e], b = vidx[1][e];
        const T u0 = Ur[3 * a] - Ur[3 * b], u1 = Ur[3 * a + 1] - Ur[3 * b + 1], u2 = Ur[3 * a + 2] - Ur[3 * b + 2];
        const S al = Xc(0, 3), be = Xc(0, 4), ga = Xc(0, 5);
        const S ca = cos(al), cb = cos(be), cg = cos(ga), sa = sin(al), sb = sin(be), sg = sin(ga);
        const S m0 = cg * cb, m1 = -sg * ca + cg * sb * sa, m2 = sg * sa + cg * sb * ca;
        const S m3 = sg * cb, m4 = cg * ca + sg * sb * sa, m5 = -cg * sa + sg * sb * ca;
        const S m6 = -sb, m7 = cb * sa, m8 = cb * ca;
        r[0] = w_reg * ((Xc(0, 0) - Xc(1, 0)) - (m0 * u0 + m1 * u1 + m2 * u2));
        r[1] = w_reg * ((Xc(0, 1) - Xc(1, 1)) - (m3 * u0 + m4 * u1 + m5 * u2));
        r[2] = w_reg * ((Xc(0, 2) - Xc(1, 2)) - (m6 * u0 + m7 * u1 + m8 * u2));
    }
};

template <class T> EnergyOps<T>* makeCot(const unsigned* dims) { return new GraphOps<T, CotangentG<T>>(dims, true); }
template <class T> EnergyOps<T>* makeEmb(const unsigned* dims) { return new GraphOps<T, EmbeddedG<T>>(dims, true); }
template <class T> EnergyOps<T>* makeRob(const unsigned* dims) { return new GraphOps<T, RobustG<T>>(dims, true); }

}  // namespace

EnergyInfo cotangentInfo() {
    EnergyInfo e;
    e.name = "cotangent_mesh_smoothing"; e.nDims = 1; e.usePreconditioner = true; e.floatOnly = false;
    e.params = {{ParamDecl::kScalar, "w_fit", "float", 0}, {ParamDecl::kScalar, "w_reg", "float", 1}, {ParamDecl::kUnknown, "X", "opt_float3", 2},
                {ParamDecl::kArray, "A", "opt_float3", 3}, {ParamDecl::kGraphCount, "G", "int", 4}, {ParamDecl::kGraphIndex, "G.v0", "int", 5},
                {ParamDecl::kGraphIndex, "G.v1", "int", 6}, {ParamDecl::kGraphIndex, "G.v2", "int", 7}, {ParamDecl::kGraphIndex, "G.v3", "int", 8}};
    e.makeFloat = makeCot<float>; e.makeDouble = makeCot<double>;
    return e;
}
EnergyInfo embeddedInfo() {
    EnergyInfo e;
    e.name = "embedded_mesh_deformation"; e.nDims = 1; e.usePreconditioner = true; e.floatOnly = false;
    e.params = {{ParamDecl::kScalar, "w_fitSqrt", "float", 0}, {ParamDecl::kScalar, "w_regSqrt", "float", 1}, {ParamDecl::kScalar, "w_rotSqrt", "float", 2},
                {ParamDecl::kUnknown, "Offset", "opt_float3", 3}, {ParamDecl::kUnknown, "RotMatrix", "opt_float9", 4}, {ParamDecl::kArray, "UrShape", "opt_float3", 5},
                {ParamDecl::kArray, "Constraints", "opt_float3", 6}, {ParamDecl::kGraphCount, "G", "int", 7}, {ParamDecl::kGraphIndex, "G.v0", "int", 8},
                {ParamDecl::kGraphIndex, "G.v1", "int", 9}};
    e.makeFloat = makeEmb<float>; e.makeDouble = makeEmb<double>;
    return e;
}
EnergyInfo robustInfo() {
    EnergyInfo e;
    e.name = "robust_nonrigid_alignment"; e.nDims = 1; e.usePreconditioner = true; e.floatOnly = false;
    e.params = {{ParamDecl::kScalar, "w_fitSqrt", "float", 0}, {ParamDecl::kScalar, "w_regSqrt", "float", 1}, {ParamDecl::kUnknown, "Offset", "opt_float3", 2},
                {ParamDecl::kUnknown, "Angle", "opt_float3", 3}, {ParamDecl::kUnknown, "RobustWeights", "opt_float", 4}, {ParamDecl::kArray, "UrShape", "opt_float3", 5},
                {ParamDecl::kArray, "Constraints", "opt_float3", 6}, {ParamDecl::kArray, "ConstraintNormals", "opt_float3", 7},
                {ParamDecl::kGraphCount, "G", "int", 8}, {ParamDecl::kGraphIndex, "G.v0", "int", 9}, {ParamDecl::kGraphIndex, "G.v1", "int", 10}};
    e.makeFloat = makeRob<float>; e.makeDouble = makeRob<double>;
    return e;
}

}  // namespace optamd
