// shape_from_shading through the C API in double precision with the LM solver (BASELINE config 3), driven like the reference
// example (examples/shape_from_shading/src/main.cpp:27-38, SFSSolverInput.h:22-66): 60 nonlinear x 10 linear iterations.
// Input: either the reference's fixture set -- pass the prefix of `<prefix>_targetIntensity.imagedump`,
// `_targetDepth`, `_initialUnknown`, `_maskEdgeMap` and `<prefix>.SFSSolverParameters` -- or, without arguments, a
// procedural surface lit with the fixture's spherical-harmonics coefficients.
//   usage: sfs_example [prefix | -] [width=512] [energy.t] [height=width]      (640 x 480: the size of the reference's fixture, examples/data/shape_from_shading)
#include "common.h"
#include <cmath>
#include <cstring>

struct Image { int w = 0, h = 0, c = 0, dtype = 0; std::vector<float> f; std::vector<unsigned char> u; };
static bool readImagedump(const std::string& path, Image& im) {     // int32 w,h,channels,datatype{0 float32,1 uint8} + payload (API/src/im.t:7-15)
    FILE* fh = fopen(path.c_str(), "rb");
    if (!fh) return false;
    int hdr[4];
    if (fread(hdr, 4, 4, fh) != 4) { fclose(fh); return false; }
    im.w = hdr[0]; im.h = hdr[1]; im.c = hdr[2]; im.dtype = hdr[3];
    const size_t n = (size_t)im.w * im.h * im.c;
    bool ok;
    if (im.dtype == 0) { im.f.resize(n); ok = fread(im.f.data(), 4, n, fh) == n; for (auto& v : im.f) if (std::isinf(v)) v = v > 0 ? 3.4e38f : -10000.0f; }
    else { im.u.resize(n); ok = fread(im.u.data(), 1, n, fh) == n; }
    fclose(fh);
    return ok;
}
static std::vector<double> shade(const std::vector<double>& d, int W, int H, double fx, double fy, double ux, double uy, const float* L) {
    std::vector<double> B((size_t)W * H, 0.0);
    for (int j = 1; j < H; ++j) for (int i = 1; i < W; ++i) {
        const double d1 = d[(size_t)j * W + i], d0 = d[(size_t)j * W + i - 1], d2 = d[(size_t)(j - 1) * W + i];
        double nx = d2 * (d1 - d0) / fy, ny = d0 * (d1 - d2) / fx, nz = nx * (ux - i) / fx + ny * (uy - j) / fy - d0 * d2 / (fx * fy);
        const double sq = nx * nx + ny * ny + nz * nz, inv = sq > 0 ? 1.0 / std::sqrt(sq) : 1.0;
        nx *= inv; ny *= inv; nz *= inv;
        B[(size_t)j * W + i] = L[0] + L[1] * ny + L[2] * nz + L[3] * nx + L[4] * nx * ny + L[5] * ny * nz + L[6] * (-nx * nx - ny * ny + 2 * nz * nz) + L[7] * nz * nx + L[8] * (nx * nx - ny * ny);
    }
    return B;
}

int main(int argc, char** argv) {
    const std::string prefix = argc > 1 ? argv[1] : "-";
    int W = argc > 2 ? atoi(argv[2]) : 512, H = W;
    const std::string energy = argc > 3 ? argv[3] : "opt_amd/energies/shape_from_shading.t";
    if (argc > 4) H = atoi(argv[4]);
    float wts[3] = {100.f, 100.f, 1.f}, fx = 574.0529f, fy = 574.0528f, ux = 320.f, uy = 240.f;
    float L[9] = {0.6908f, 0.0446f, 0.0181f, -0.1773f, -0.0407f, 0.1447f, 0.0239f, -0.2466f, 0.0058f};
    std::vector<double> X, D, Im;
    std::vector<unsigned char> edgeR, edgeC;
    if (prefix != "-") {
        Image inten, depth, init, edges;
        if (!readImagedump(prefix + "_targetIntensity.imagedump", inten) || !readImagedump(prefix + "_targetDepth.imagedump", depth) ||
            !readImagedump(prefix + "_initialUnknown.imagedump", init) || !readImagedump(prefix + "_maskEdgeMap.imagedump", edges)) { fprintf(stderr, "cannot read fixture set %s\n", prefix.c_str()); return 4; }
        W = depth.w; H = depth.h;
        X.assign(init.f.begin(), init.f.end()); D.assign(depth.f.begin(), depth.f.end()); Im.assign(inten.f.begin(), inten.f.end());
        edgeR.assign(edges.u.begin(), edges.u.begin() + (size_t)W * H); edgeC.assign(edges.u.begin() + (size_t)W * H, edges.u.begin() + 2 * (size_t)W * H);
        FILE* fh = fopen((prefix + ".SFSSolverParameters").c_str(), "rb");       // TerraSolverParameters.h:7-44
        float blob[40] = {0};
        if (!fh || fread(blob, 1, 160, fh) < 156) { fprintf(stderr, "cannot read parameters\n"); return 4; }
        fclose(fh);
        wts[0] = blob[0]; wts[1] = blob[1]; wts[2] = blob[3]; fx = blob[7]; fy = blob[8]; ux = blob[9]; uy = blob[10];
        memcpy(L, blob + 27, sizeof(L));
    } else {
        fx *= W / 640.f; fy *= W / 640.f; ux = W / 2.f; uy = H / 2.f;
        std::vector<double> detail((size_t)W * H);
        D.resize((size_t)W * H); X.resize((size_t)W * H);
        unsigned s = 7u;
        for (int j = 0; j < H; ++j) for (int i = 0; i < W; ++i) {
            const size_t k = (size_t)j * W + i;
            D[k] = 0.45 + 0.05 * std::sin(i * 6.0 / W) * std::cos(j * 5.0 / H);
            detail[k] = D[k] + 0.002 * std::sin(i * 0.7) * std::sin(j * 0.9);
            s = s * 1664525u + 1013904223u;
            X[k] = D[k] + 1e-3 * ((double)(s >> 8) / (1u << 24) - 0.5);
        }
        Im = shade(detail, W, H, fx, fy, ux, uy, L);
        edgeR.assign((size_t)W * H, 1); edgeC.assign((size_t)W * H, 1);
    }
    dumpInput("sfs_X", X); dumpInput("sfs_D", D); dumpInput("sfs_Im", Im); dumpInput("sfs_edgeR", edgeR); dumpInput("sfs_edgeC", edgeC);
    dumpInput("sfs_scalars", std::vector<float>{wts[0], wts[1], wts[2], fx, fy, ux, uy, L[0], L[1], L[2], L[3], L[4], L[5], L[6], L[7], L[8]});
    DeviceBuffer<double> dX(X), dD(D), dIm(Im);
    DeviceBuffer<unsigned char> dR(edgeR), dC(edgeC);
    Opt_InitializationParameters ip = {};
    ip.doublePrecision = 1;
    Opt_State* state = Opt_NewState(ip);
    if (!state) return 2;
    unsigned int dims[] = {(unsigned)W, (unsigned)H};
    Opt_Problem* problem = Opt_ProblemDefine(state, energy.c_str(), "LMGPU");
    Opt_Plan* plan = Opt_ProblemPlan(state, problem, dims);
    if (!plan) return 3;
    int nonLinearIter = 60, linearIter = 10;
    Opt_SetSolverParameter(state, plan, "nIterations", &nonLinearIter);
    Opt_SetSolverParameter(state, plan, "lIterations", &linearIter);
    void* params[21] = {&wts[0], &wts[1], &wts[2], &fx, &fy, &ux, &uy, &L[0], &L[1], &L[2], &L[3], &L[4], &L[5], &L[6], &L[7], &L[8],
                        dX.ptr, dD.ptr, dIm.ptr, dR.ptr, dC.ptr};
    std::vector<SolverIteration> iters, none;
    std::cout << "//////////// (Opt(LM)) ///////////////" << std::endl;
    profiledSolve(state, plan, params, iters);
    const double finalCost = Opt_ProblemCurrentCost(state, plan);
    Opt_PlanFree(state, plan); Opt_ProblemDelete(state, problem);
    saveSolverResults("results_double.csv", none, iters, true);
    reportFinalCosts("Shape From Shading", false, 0.0, true, finalCost);
    double ms = 0; for (auto& it : iters) ms += it.timeInMS;
    std::cout << std::fixed << std::setprecision(2) << W << "x" << H << ", " << iters.size() - 1 << " LM steps, total solver time: " << ms << " ms" << std::endl;
    return (std::isfinite(finalCost) && finalCost < iters[0].cost) ? 0 : 1;
}
