-- Feature-preserving mesh smoothing with a cotangent-weighted Laplacian (Meyer et al.) in Opt's energy DSL.
-- problemparams layout:
--   [0] w_fit   float (host)      sqrt of the data weight
--   [1] w_reg   float (host)      sqrt of the smoothness weight
--   [2] X       opt_float3[N]     unknown vertex positions
--   [3] A       opt_float3[N]     input vertex positions
--   [4] G       int (host)        number of hyperedges;  [5..8] v0, v1, v2, v3: int[G]
--        v0 = a vertex, v1 = one of its ring neighbours, v2 / v3 = the neighbours before / after v1 in the ring
local N = Dim("N", 0)
local w_fit = Param("w_fit", float, 0)
local w_reg = Param("w_reg", float, 1)
local X = Unknown("X", opt_float3, {N}, 2)
local A = Array("A", opt_float3, {N}, 3)
local G = Graph("G", 4, "v0", {N}, 5, "v1", {N}, 6, "v2", {N}, 7, "v3", {N}, 8)

UsePreconditioner(true)

-- stay near the input
Energy(w_fit * (X(0) - A(0)))

-- cotangent of the angle between two unit vectors, guarded against degenerate triangles
local function cotangent(p, q)
    local c = Dot3(p, q)
    local s2 = Dot3(p, p) * Dot3(q, q) - c * c
    s2 = Select(greater(s2, 0.0), s2, 0.0001)
    return c / Sqrt(s2)
end

-- the two angles opposite the edge (v0, v1), seen from v2 and from v3
local alpha = cotangent(normalize(X(G.v0) - X(G.v2)), normalize(X(G.v1) - X(G.v2)))
local beta = cotangent(normalize(X(G.v0) - X(G.v3)), normalize(X(G.v1) - X(G.v3)))
local weight = 0.5 * (alpha + beta)
weight = Sqrt(Select(greater(weight, 0.0), weight, 0.0001))
Energy(w_reg * weight * (X(G.v1) - X(G.v0)))
