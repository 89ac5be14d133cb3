-- Non-rigid surface alignment with per-correspondence robust weights (Zollhoefer et al.).  problemparams layout:
--   [0] w_fitSqrt, [1] w_regSqrt   float (host)
--   [2] Offset             opt_float3[N]   unknown deformed positions
--   [3] Angle              opt_float3[N]   unknown per-vertex Euler angles
--   [4] RobustWeights      opt_float [N]   unknown confidence of each correspondence
--   [5] UrShape            opt_float3[N]   source positions
--   [6] Constraints        opt_float3[N]   corresponding target points; components < -999999.9 mark "no correspondence"
--   [7] ConstraintNormals  opt_float3[N]   target normals
--   [8] G                  int (host)      number of half-edges;  [9] v0, [10] v1: int[G]
local N = Dim("N", 0)
local w_fitSqrt = Param("w_fitSqrt", float, 0)
local w_regSqrt = Param("w_regSqrt", float, 1)
local w_confSqrt = 0.1
local Offset = Unknown("Offset", opt_float3, {N}, 2)
local Angle = Unknown("Angle", opt_float3, {N}, 3)
local RobustWeights = Unknown("RobustWeights", opt_float, {N}, 4)
local UrShape = Array("UrShape", opt_float3, {N}, 5)
local Constraints = Array("Constraints", opt_float3, {N}, 6)
local ConstraintNormals = Array("ConstraintNormals", opt_float3, {N}, 7)
local G = Graph("G", 8, "v0", {N}, 9, "v1", {N}, 10)

UsePreconditioner(true)

local confidence = RobustWeights(0)
local has_target = greatereq(Constraints(0), -999999.9)

-- point-to-plane distance to the target, scaled by the confidence
local distance = confidence * ConstraintNormals(0):dot(Offset(0) - Constraints(0))
Energy(w_fitSqrt * Select(has_target, distance, 0.0))

-- confidences should stay near one
Energy(w_confSqrt * Select(has_target, 1 - confidence * confidence, 0.0))

-- as-rigid-as-possible regulariser
local rotated = Rotate3D(Angle(G.v0), UrShape(G.v0) - UrShape(G.v1))
Energy(w_regSqrt * ((Offset(G.v0) - Offset(G.v1)) - rotated))
