// Device primitives of the image_warping kernel sets (energy_image_warping.hip: the streaming kernels; iw_onchip.h: the on-chip linear solve).
// Energy: reference examples/image_warping/image_warping.t:1-23; derivation in the header of energy_image_warping.hip.
#pragma once
#include "common.h"
#include <cstdint>

namespace optamd {
namespace {

template <class T> struct V2 { T x, y; };

// flag byte per pixel (iw_flags / iw_bindMarch): bit0 pixel exists and Mask == 0; bit1 fit constraint valid; bits 2-4 number of active 4-neighbours
constexpr uint8_t kActive = 1, kFit = 2;
constexpr int kCountShift = 2;       // bits 2..4: number of active 4-neighbours (0..4) of an active pixel

template <class T>
struct Q {            // one pixel of a row held in registers
    T ox, oy, a;      // the vector (p_{k-1} or p_k)
    T c, s;           // cos/sin of the pixel's angle
    T ux, uy;         // UrShape (dead on a unit lattice)
    T on;             // 1 if the pixel exists and is not excluded, else 0
    T fw;             // w_fit^2 if its fit residual is on, else 0
};
// the two residuals shared by centre c and its neighbour n in direction (DX, DY) (see iw_pair)
template <int DX, int DY, bool LATTICE, class T>
__device__ __forceinline__ void iw_pairQ(const Q<T>& c, const Q<T>& n, T& ax, T& ay, T& aa) {
    T Dcx, Dcy, Dnx, Dny;
    if (LATTICE) {       // U_c - U_n = -(DX, DY)
        Dcx = DX ? T(DX) * c.s : T(DY) * c.c;   Dcy = DX ? T(-DX) * c.c : T(DY) * c.s;
        Dnx = DX ? T(-DX) * n.s : T(-DY) * n.c; Dny = DX ? T(DX) * n.c : T(-DY) * n.s;
    } else {
        const T ux = c.ux - n.ux, uy = c.uy - n.uy;
        Dcx = -c.s * ux - c.c * uy; Dcy = c.c * ux - c.s * uy;
        Dnx = n.s * ux + n.c * uy;  Dny = -n.c * ux + n.s * uy;
    }
    const T dx = c.ox - n.ox, dy = c.oy - n.oy;
    const T jcx = dx - Dcx * c.a, jcy = dy - Dcy * c.a;
    const T jnx = -dx - Dnx * n.a, jny = -dy - Dny * n.a;
    ax += n.on * (jcx - jnx); ay += n.on * (jcy - jny);
    aa -= n.on * (Dcx * jcx + Dcy * jcy);
}
// ---- each pair of residuals is formed ONCE (round 3) -----------------------------------------------------------------------------------------------
// iw_pairQ evaluates, for centre c and neighbour n, the residual centred at c towards n and the one centred at n towards c -- and the pixel n, when it is the
// centre, evaluates the same two residuals again from its side: jc' = jn, jn' = jc, D_{n,-d} = Dn bit for bit (a - b = -(b - a) and a product keeps its value
// when both factors change sign).  So the pair is formed once, by the end that comes first in lane / sweep order, which also leaves what the other end needs:
//   jc' - jn' = -(jc - jn)   and   D_{n,-d} . jc' = Dn . jn.
// The right-hand pair of lane x is the left-hand pair of lane x + 1 (three DPP moves instead of the neighbour's three vector fields and 13 VALU instructions); the
// pair towards the next row of the march is the pair towards the previous row one trip later (three registers per stencil evaluation).  Accumulation order and
// every accumulated value are those of iw_pairQ: the result is the same bits, ~34 of ~250 VALU instructions per pixel-row less -- which pays where the kernel is
// issue-bound (2048^2, slabs: profiles/r03l_iteration_kernel_sq_counters.md), not at 4096^2.
#ifndef IW_SHARE_PAIRS
#define IW_SHARE_PAIRS 1
#endif
template <class T> struct PairOut { T dx, dy, tn; };      // (jc - jn).x, (jc - jn).y, Dn . jn
// m2 (general UrShape only): receives this pair's term of diag(J^T J) of the Angle unknown -- (w D_x)^2 + (w D_y)^2, iw_evalJTF's `Pa` -- for the centre ([0]) and for
// the neighbour ([1]); summed over a pixel's four pairs that is what its Jacobi preconditioner M_a inverts, so the iteration kernel needs no M_a from memory.
template <int DX, int DY, bool LATTICE, class T>
__device__ __forceinline__ PairOut<T> iw_pairFull(const Q<T>& c, const Q<T>& n, T& ax, T& ay, T& aa, T* m2 = nullptr, T w = T(0)) {
    T Dcx, Dcy, Dnx, Dny;
    if (LATTICE) {       // U_c - U_n = -(DX, DY)
        Dcx = DX ? T(DX) * c.s : T(DY) * c.c;   Dcy = DX ? T(-DX) * c.c : T(DY) * c.s;
        Dnx = DX ? T(-DX) * n.s : T(-DY) * n.c; Dny = DX ? T(DX) * n.c : T(-DY) * n.s;
    } else {
        const T ux = c.ux - n.ux, uy = c.uy - n.uy;
        Dcx = -c.s * ux - c.c * uy; Dcy = c.c * ux - c.s * uy;
        Dnx = n.s * ux + n.c * uy;  Dny = -n.c * ux + n.s * uy;
        if (m2) { m2[0] = (w * Dcx) * (w * Dcx) + (w * Dcy) * (w * Dcy); m2[1] = (w * Dnx) * (w * Dnx) + (w * Dny) * (w * Dny); }
    }
    const T dx = c.ox - n.ox, dy = c.oy - n.oy;
    const T jcx = dx - Dcx * c.a, jcy = dy - Dcy * c.a;
    const T jnx = -dx - Dnx * n.a, jny = -dy - Dny * n.a;
    PairOut<T> o;
    o.dx = jcx - jnx; o.dy = jcy - jny;
    o.tn = Dnx * jnx + Dny * jny;
    ax += n.on * o.dx; ay += n.on * o.dy;
    aa -= n.on * (Dcx * jcx + Dcy * jcy);
    return o;
}
// the same pair seen from its far end: `o` is what the neighbour's evaluation left, nOn that neighbour's activity
template <class T>
__device__ __forceinline__ void iw_pairInherited(const PairOut<T>& o, T nOn, T& ax, T& ay, T& aa) {
    ax -= nOn * o.dx; ay -= nOn * o.dy;
    aa -= nOn * o.tn;
}
template <class T>
struct IWArgs {
    int W, H;                 // local image (incl. ghost rows in slab mode)
    int yBegin, yEnd;         // owned rows
    int gy0, Hg;              // global row of local row 0, global height
    const T* Offset; const T* Angle; const T* UrShape; const T* Constraints; const T* Mask;
    T w_fit, w_reg;
    uint8_t* flags;           // bit0: pixel exists and Mask == 0 ; bit1: fit constraint valid ; bits 2-4: number of active 4-neighbours
    T* cs;                    // (cos a, sin a) per pixel
};


// A real register copy the compiler cannot fold.  The marching kernels pass some loaded fields (cos/sin, U, M) through
// unchanged for three rows; left to itself the compiler keeps them in the registers the load wrote, has to rotate the
// prefetch buffers with v_movs at the loop back-edge, and a v_mov of a register whose load is still in flight costs an
// s_waitcnt there -- the prefetch drains every trip.  Copying once, where the data is consumed anyway, frees the raw
// registers so the next prefetch lands in the same ones and the back-edge carries no waits.
__device__ __forceinline__ float regCopy(float v) { float r; asm("v_mov_b32 %0, %1" : "=v"(r) : "v"(v)); return r; }
__device__ __forceinline__ int regCopy(int v) { int r; asm("v_mov_b32 %0, %1" : "=v"(r) : "v"(v)); return r; }
__device__ __forceinline__ double regCopy(double v) { return __hiloint2double(regCopy(__double2hiint(v)), regCopy(__double2loint(v))); }


// Row addressing through buffer descriptors: element (row, x) of an array is  descriptor(base)  +  soffset = row * W * size (+ the offset of the Angle
// part), one SALU product shared by all arrays of a row  +  voffset = x * size, a per-lane constant of the whole launch.  A load or store then needs no
// address VALU at all.  Byte offsets are 32-bit: a kernel takes this form only while the solver vector stays below 4 GiB.
typedef unsigned int iw_u2 __attribute__((ext_vector_type(2)));
typedef unsigned int iw_u4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t iw_rsrc(const void* base) { return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, -1 /* 2^32 - 1 bytes */, 0x00020000); }
__device__ __forceinline__ V2<float> bufLd2(__amdgpu_buffer_rsrc_t r, unsigned v, unsigned so, const float*) { const iw_u2 w = __builtin_amdgcn_raw_buffer_load_b64(r, (int)v, (int)so, 0); return V2<float>{__uint_as_float(w.x), __uint_as_float(w.y)}; }
__device__ __forceinline__ V2<double> bufLd2(__amdgpu_buffer_rsrc_t r, unsigned v, unsigned so, const double*) { const iw_u4 w = __builtin_amdgcn_raw_buffer_load_b128(r, (int)v, (int)so, 0); V2<double> o; __builtin_memcpy(&o, &w, 16); return o; }
__device__ __forceinline__ float bufLd1(__amdgpu_buffer_rsrc_t r, unsigned v, unsigned so, const float*) { return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, (int)v, (int)so, 0)); }
__device__ __forceinline__ double bufLd1(__amdgpu_buffer_rsrc_t r, unsigned v, unsigned so, const double*) { const iw_u2 w = __builtin_amdgcn_raw_buffer_load_b64(r, (int)v, (int)so, 0); double o; __builtin_memcpy(&o, &w, 8); return o; }
__device__ __forceinline__ void bufSt2(__amdgpu_buffer_rsrc_t r, unsigned v, unsigned so, float x, float y) { __builtin_amdgcn_raw_buffer_store_b64(iw_u2{__float_as_uint(x), __float_as_uint(y)}, r, (int)v, (int)so, 0); }
__device__ __forceinline__ void bufSt2(__amdgpu_buffer_rsrc_t r, unsigned v, unsigned so, double x, double y) { const V2<double> o{x, y}; iw_u4 w; __builtin_memcpy(&w, &o, 16); __builtin_amdgcn_raw_buffer_store_b128(w, r, (int)v, (int)so, 0); }
__device__ __forceinline__ void bufSt1(__amdgpu_buffer_rsrc_t r, unsigned v, unsigned so, float x) { __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(x), r, (int)v, (int)so, 0); }
__device__ __forceinline__ void bufSt1(__amdgpu_buffer_rsrc_t r, unsigned v, unsigned so, double x) { iw_u2 w; __builtin_memcpy(&w, &x, 8); __builtin_amdgcn_raw_buffer_store_b64(w, r, (int)v, (int)so, 0); }
}  // namespace
}  // namespace optamd
