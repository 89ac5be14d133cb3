"""CORROBORATION ONLY -- NOT A PIN.  Runs in the build container only (skipped wherever /root/reference is not mounted).

The reference ships a hand-written CUDA comparator for image_warping whose per-variable maths is plain C++ in a header
(examples/image_warping/src/WarpingSolverEquations.h:9-349: evalFDevice, evalMinusJTFDevice, applyJTJDevice).  This test compiles that
header WHERE IT LIES (nothing of it is copied into the repo) with g++, against a throw-away stand-in for <cuda_runtime.h> written into the
test's tmp directory (vector types and empty qualifiers -- which is exactly why this can never count as a build of the reference and pins
nothing under the task's rules), and checks the relations SURVEY.md section 7 step 1 predicts at Mask == 0:
    comparator F          == 2 * Opt cost            (the comparator sums w r^2, Opt 1/2 sum r^2 with sqrt weights)
    comparator -J^T F     == 2 * oracle r0 = -2 J^T F
    comparator J^T J p    == 2 * oracle J^T J p
It corroborates that the oracle's residuals, Jacobian blocks and sign conventions are the ones the reference authors wrote by hand.
"""
import os
import subprocess

import numpy as np
import pytest

from opt_amd import workloads as wl
from helpers import oracle_solver, rel_err

REF_HDR_DIR = "/root/reference/examples/image_warping/src"

STANDIN = r'''
#pragma once
#include <cmath>
#include <cstring>
#include <limits>
#define __device__
#define __host__
#define __inline__ inline
#define __forceinline__ inline
#define __shared__
#define __global__
struct float2 { float x, y; float2() {} float2(float a, float b) : x(a), y(b) {} };
struct float3 { float x, y, z; };
struct float4 { float x, y, z, w; };
struct int2 { int x, y; }; struct int3 { int x, y, z; }; struct int4 { int x, y, z, w; };
struct uint2 { unsigned x, y; }; struct uint3 { unsigned x, y, z; }; struct uint4 { unsigned x, y, z, w; };
struct uchar4 { unsigned char x, y, z, w; };
inline float2 make_float2(float x, float y) { return float2(x, y); }
inline float3 make_float3(float x, float y, float z) { float3 r; r.x = x; r.y = y; r.z = z; return r; }
inline float4 make_float4(float x, float y, float z, float w) { float4 r; r.x = x; r.y = y; r.z = z; r.w = w; return r; }
inline int2 make_int2(int x, int y) { int2 r; r.x = x; r.y = y; return r; }
inline int3 make_int3(int x, int y, int z) { int3 r; r.x = x; r.y = y; r.z = z; return r; }
inline int4 make_int4(int x, int y, int z, int w) { int4 r; r.x = x; r.y = y; r.z = z; r.w = w; return r; }
inline uint2 make_uint2(unsigned x, unsigned y) { uint2 r; r.x = x; r.y = y; return r; }
inline uint3 make_uint3(unsigned x, unsigned y, unsigned z) { uint3 r; r.x = x; r.y = y; r.z = z; return r; }
inline uchar4 make_uchar4(unsigned char x, unsigned char y, unsigned char z, unsigned char w) { uchar4 r; r.x = x; r.y = y; r.z = z; r.w = w; return r; }
typedef int cudaError; typedef int cudaError_t;
enum { cudaSuccess = 0, cudaMemcpyDeviceToHost = 2, cudaMemcpyHostToDevice = 1 };
inline const char* cudaGetErrorString(int) { return "stand-in"; }
inline int cudaMemcpy(void* d, const void* s, size_t n, int) { memcpy(d, s, n); return 0; }
inline void __syncthreads() {}
inline float __shfl_down(float v, int, int) { return v; }
inline float __int_as_float(int i) { float f; memcpy(&f, &i, 4); return f; }
'''

DRIVER = r'''
#include "%(hdr)s/WarpingSolverState.h"
#include "%(hdr)s/WarpingSolverParameters.h"
#include "%(hdr)s/WarpingSolverEquations.h"
#include <cstdio>
#include <vector>
float bucket[2048];
int main(int argc, char** argv) {
    FILE* f = fopen(argv[1], "rb");
    int W, H; float wf, wr;
    fread(&W, 4, 1, f); fread(&H, 4, 1, f); fread(&wf, 4, 1, f); fread(&wr, 4, 1, f);
    const int N = W * H;
    std::vector<float2> x(N), ur(N), con(N), p(N), delta(N), pre(N); std::vector<float> A(N), mask(N), pA(N), deltaA(N), preA(N);
    fread(x.data(), 8, N, f); fread(A.data(), 4, N, f); fread(ur.data(), 8, N, f); fread(con.data(), 8, N, f); fread(mask.data(), 4, N, f);
    fread(p.data(), 8, N, f); fread(pA.data(), 4, N, f); fclose(f);
    SolverInput in; in.N = N; in.width = W; in.height = H; in.d_constraints = con.data();
    SolverState st; memset(&st, 0, sizeof st);
    st.d_x = x.data(); st.d_A = A.data(); st.d_urshape = ur.data(); st.d_mask = mask.data(); st.d_p = p.data(); st.d_pA = pA.data();
    st.d_delta = delta.data(); st.d_deltaA = deltaA.data(); st.d_precondioner = pre.data(); st.d_precondionerA = preA.data();
    SolverParameters pr; pr.weightFitting = wf; pr.weightRegularizer = wr; pr.nNonLinearIterations = 1; pr.nLinIterations = 1;
    FILE* o = fopen(argv[2], "wb");
    double F = 0;
    for (int i = 0; i < N; ++i) F += (double)evalFDevice(i, in, st, pr);
    fwrite(&F, 8, 1, o);
    for (int i = 0; i < N; ++i) { float bA; float2 b = evalMinusJTFDevice(i, in, st, pr, bA); float v[3] = {b.x, b.y, bA}; fwrite(v, 4, 3, o); }
    for (int i = 0; i < N; ++i) { float bA; float2 b = applyJTJDevice(i, in, st, pr, bA); float v[3] = {b.x, b.y, bA}; fwrite(v, 4, 3, o); }
    fclose(o);
    return 0;
}
'''


@pytest.mark.skipif(not os.path.isdir(REF_HDR_DIR), reason="reference checkout not mounted (build container only)")
def test_oracle_agrees_with_the_hand_written_comparator_maths(oracle_lib, tmp_path):
    (tmp_path / "cuda_runtime.h").write_text(STANDIN)
    (tmp_path / "drv.cpp").write_text(DRIVER % {"hdr": REF_HDR_DIR})
    exe = tmp_path / "drv"
    subprocess.check_call(["g++", "-std=c++14", "-O1", "-w", "-fpermissive", f"-I{tmp_path}", f"-I{REF_HDR_DIR}", str(tmp_path / "drv.cpp"), "-o", str(exe)])
    W, H = 23, 17
    P = wl.image_warping(W, H, random_state=19, mask_fraction=0.0, perturb=0.4)          # Mask == 0: the exact 2x relations hold
    rng = np.random.default_rng(4)
    v = rng.standard_normal(3 * W * H).astype(np.float32)
    blob = tmp_path / "in.bin"
    with open(blob, "wb") as f:
        f.write(np.array([W, H], dtype=np.int32).tobytes())
        f.write(np.array([float(P.params[5]) ** 2, float(P.params[6]) ** 2], dtype=np.float32).tobytes())   # weightFitting = w_fitSqrt^2 (CombinedSolver.h:126-130)
        for a in (P.params[0], P.params[1], P.params[2], P.params[3], P.params[4]):
            f.write(np.ascontiguousarray(a, dtype=np.float32).tobytes())
        f.write(v[:2 * W * H].tobytes()); f.write(v[2 * W * H:].tobytes())
    out = tmp_path / "out.bin"
    subprocess.check_call([str(exe), str(blob), str(out)])
    raw = open(out, "rb").read()
    F = np.frombuffer(raw[:8], dtype=np.float64)[0]
    rest = np.frombuffer(raw[8:], dtype=np.float32).reshape(2, W * H, 3)
    to_flat = lambda a: np.concatenate([a[:, :2].reshape(-1), a[:, 2]])                   # comparator per-variable (x, y, angle) -> Opt's [O x N | a x N]
    o = oracle_solver(oracle_lib, P)
    cost = o.eval_cost(P.params)
    jtf, _ = o.eval_jtf(P.params)
    Av = o.apply_jtj(P.params, v)
    assert abs(F - 2.0 * cost) <= 2e-5 * abs(F)
    assert rel_err(to_flat(rest[0]), -2.0 * jtf) < 2e-5
    assert rel_err(to_flat(rest[1]), 2.0 * Av) < 2e-5
    o.close()
