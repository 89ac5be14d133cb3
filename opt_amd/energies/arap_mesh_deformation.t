-- As-rigid-as-possible mesh deformation in Opt's energy DSL: per-vertex translation (Offset) and rotation
-- (Angle, Euler angles), edges of the mesh as a graph of directed half-edges (v0 -> v1).
-- problemparams:
--   [0] w_fitSqrt host float      [1] w_regSqrt host float
--   [2] Offset opt_float3[N] unknown (device)     [3] Angle opt_float3[N] unknown (device)
--   [4] UrShape opt_float3[N] rest pose (device)  [5] Constraints opt_float3[N], x < -999999.9 = unconstrained (device)
--   [6] number of half-edges (HOST int*)          [7] v0 per edge (device int*)    [8] v1 per edge (device int*)
local N = opt.Dim("N", 0)
local w_fitSqrt   = Param("w_fitSqrt", float, 0)
local w_regSqrt   = Param("w_regSqrt", float, 1)
local Offset      = Unknown("Offset", opt_float3, {N}, 2)
local Angle       = Unknown("Angle", opt_float3, {N}, 3)
local UrShape     = Array("UrShape", opt_float3, {N}, 4)
local Constraints = Array("Constraints", opt_float3, {N}, 5)
local G = Graph("G", 6, "v0", {N}, 7, "v1", {N}, 8)
UsePreconditioner(true)

-- handle vertices are pulled to their targets
local pinned = greatereq(Constraints(0,0), -999999.9)
Energy(Select(pinned, w_fitSqrt * (Offset(0) - Constraints(0)), 0))

-- every half-edge should be the rest edge rotated by its head vertex's rotation
local restEdge     = UrShape(G.v0) - UrShape(G.v1)
local deformedEdge = Offset(G.v0) - Offset(G.v1)
Energy(w_regSqrt * (deformedEdge - Rotate3D(Angle(G.v0), restEdge)))
