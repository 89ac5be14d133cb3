// TEST INFRASTRUCTURE ONLY (CPU oracle) -- never linked into, imported by, or called from the product path.
//
// Forward-mode dual numbers: the oracle's stand-in for Opt's symbolic differentiation
// (reference API/src/ad.t:612-660 `Exp:d`, op partials ad.t:747-797).  A residual written once
// against Dual<T,N> yields its value and its partials w.r.t. the N unknowns of its support,
// which is exactly what the reference generator consumes (o.t:2029-2316).
#pragma once
#include <cmath>

namespace oracle {

template <class T, int N>
struct Dual {
    T v;
    T d[N];
    Dual() : v(0) { for (int i = 0; i < N; ++i) d[i] = 0; }
    Dual(T c) : v(c) { for (int i = 0; i < N; ++i) d[i] = 0; }
    static Dual var(T value, int slot) { Dual r(value); r.d[slot] = T(1); return r; }
};

template <class T, int N> Dual<T,N> operator+(const Dual<T,N>& a, const Dual<T,N>& b) { Dual<T,N> r; r.v = a.v + b.v; for (int i = 0; i < N; ++i) r.d[i] = a.d[i] + b.d[i]; return r; }
template <class T, int N> Dual<T,N> operator-(const Dual<T,N>& a, const Dual<T,N>& b) { Dual<T,N> r; r.v = a.v - b.v; for (int i = 0; i < N; ++i) r.d[i] = a.d[i] - b.d[i]; return r; }
template <class T, int N> Dual<T,N> operator-(const Dual<T,N>& a) { Dual<T,N> r; r.v = -a.v; for (int i = 0; i < N; ++i) r.d[i] = -a.d[i]; return r; }
template <class T, int N> Dual<T,N> operator*(const Dual<T,N>& a, const Dual<T,N>& b) { Dual<T,N> r; r.v = a.v * b.v; for (int i = 0; i < N; ++i) r.d[i] = a.d[i] * b.v + a.v * b.d[i]; return r; }
template <class T, int N> Dual<T,N> operator/(const Dual<T,N>& a, const Dual<T,N>& b) { Dual<T,N> r; T inv = T(1) / b.v; r.v = a.v * inv; for (int i = 0; i < N; ++i) r.d[i] = (a.d[i] - r.v * b.d[i]) * inv; return r; }

template <class T, int N> Dual<T,N> operator+(const Dual<T,N>& a, T b) { Dual<T,N> r = a; r.v += b; return r; }
template <class T, int N> Dual<T,N> operator+(T a, const Dual<T,N>& b) { return b + a; }
template <class T, int N> Dual<T,N> operator-(const Dual<T,N>& a, T b) { Dual<T,N> r = a; r.v -= b; return r; }
template <class T, int N> Dual<T,N> operator-(T a, const Dual<T,N>& b) { return (-b) + a; }
template <class T, int N> Dual<T,N> operator*(const Dual<T,N>& a, T b) { Dual<T,N> r; r.v = a.v * b; for (int i = 0; i < N; ++i) r.d[i] = a.d[i] * b; return r; }
template <class T, int N> Dual<T,N> operator*(T a, const Dual<T,N>& b) { return b * a; }
template <class T, int N> Dual<T,N> operator/(const Dual<T,N>& a, T b) { return a * (T(1) / b); }
template <class T, int N> Dual<T,N> operator/(T a, const Dual<T,N>& b) { return Dual<T,N>(a) / b; }

// ---- which float sin / cos?  The reference calls libdevice's __nv_sinf / __nv_cosf (util.t:162-171), documented to 2 ulp; this restatement calls the host's
// libm, the HIP product ocml's -- three implementations that agree to an ulp or two and are not bitwise equal.  trigSeed() != 0 emulates "another
// implementation within 1 ulp": the correctly rounded value (computed in double), nudged by one ulp up or down for a seeded quarter of the arguments whose result is
// not exact (a hash of the argument's bits, so the same angle gives the same value throughout a run).  Every seed is a legal elementwise variant of the reference, the way every
// seed of reductionMode 1 is a legal order of its atomics (tests/golden/make_trig_variants.py).  Double precision is left alone.
inline unsigned& trigSeed() { static unsigned s = 0; return s; }
inline float trigVariant(float x, double exact) {
    float r = (float)exact;
    if ((double)r == exact) return r;      // an exact result (sin 0 = 0, cos 0 = 1: every angle of the example's initial guess) is exact in every implementation
    unsigned b; __builtin_memcpy(&b, &x, 4);
    unsigned h = (b ^ (trigSeed() * 0x9E3779B9u)) * 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
    if ((h & 3u) == 0u) r = std::nextafter(r, (h & 4u) ? 2.0f : -2.0f);
    return r;
}
inline float osin(float x) { return trigSeed() ? trigVariant(x, std::sin((double)x)) : std::sin(x); }
inline float ocos(float x) { return trigSeed() ? trigVariant(x, std::cos((double)x)) : std::cos(x); }
inline double osin(double x) { return std::sin(x); }
inline double ocos(double x) { return std::cos(x); }

// ad.t:795 sin -> cos ; ad.t:787 cos -> -sin ; ad.t:797 sqrt -> 1/(2 sqrt)
template <class T, int N> Dual<T,N> sin(const Dual<T,N>& a) { Dual<T,N> r; r.v = osin(a.v); T c = ocos(a.v); for (int i = 0; i < N; ++i) r.d[i] = c * a.d[i]; return r; }
template <class T, int N> Dual<T,N> cos(const Dual<T,N>& a) { Dual<T,N> r; r.v = ocos(a.v); T s = -osin(a.v); for (int i = 0; i < N; ++i) r.d[i] = s * a.d[i]; return r; }
template <class T, int N> Dual<T,N> sqrt(const Dual<T,N>& a) { Dual<T,N> r; r.v = std::sqrt(a.v); T k = T(1) / (T(2) * r.v); for (int i = 0; i < N; ++i) r.d[i] = k * a.d[i]; return r; }

// ad.t:765-775: select(c,a,b) picks a branch; its partials are (0, c, not c) -> the chosen branch's partials.
template <class T, int N> Dual<T,N> select(bool c, const Dual<T,N>& a, const Dual<T,N>& b) { return c ? a : b; }

}  // namespace oracle
