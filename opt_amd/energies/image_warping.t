-- 2-D as-rigid-as-possible image warp, written in Opt's energy DSL.
-- This file is an INPUT of the C API (Opt_ProblemDefine): it fixes which slot of `problemparams`
-- carries what.  The MI355X backend reads only the declarations below and runs its registered,
-- hand-written HIP kernel set for the energy named by the file stem ("image_warping").
--
-- problemparams layout:
--   [0] Offset       device  opt_float2[W*H]   unknown: warped position of every pixel
--   [1] Angle        device  opt_float [W*H]   unknown: per-pixel rotation
--   [2] UrShape      device  opt_float2[W*H]   rest position of every pixel
--   [3] Constraints  device  opt_float2[W*H]   target position, (-1,-1) = unconstrained
--   [4] Mask         device  opt_float [W*H]   0 = part of the shape, anything else = excluded
--   [5] w_fitSqrt    host    float             sqrt of the fitting weight
--   [6] w_regSqrt    host    float             sqrt of the rigidity weight
local W, H = Dim("W", 0), Dim("H", 1)

local Offset      = Unknown("Offset", opt_float2, {W,H}, 0)
local Angle       = Unknown("Angle", opt_float, {W,H}, 1)
local UrShape     = Array("UrShape", opt_float2, {W,H}, 2)
local Constraints = Array("Constraints", opt_float2, {W,H}, 3)
local Mask        = Array("Mask", opt_float, {W,H}, 4)
local w_fitSqrt   = Param("w_fitSqrt", float, 5)
local w_regSqrt   = Param("w_regSqrt", float, 6)

UsePreconditioner(true)

-- masked-out pixels are not unknowns
local inShape = eq(Mask(0,0), 0)
Exclude(Not(inShape))

-- rigidity: every pixel's 4-neighbourhood should move like a rotation of its rest configuration
local neighbours = { {1,0}, {-1,0}, {0,1}, {0,-1} }
for dx, dy in Stencil(neighbours) do
    local restEdge   = UrShape(0,0) - UrShape(dx,dy)
    local warpedEdge = Offset(0,0) - Offset(dx,dy)
    local residual   = w_regSqrt * (warpedEdge - Rotate2D(Angle(0,0), restEdge))
    local usable     = InBounds(dx,dy) * eq(Mask(dx,dy), 0) * inShape
    Energy(Select(usable, residual, 0))
end

-- fitting: constrained pixels are pulled to their targets
local hasTarget = All(greatereq(Constraints(0,0), 0))
Energy(w_fitSqrt * Select(hasTarget, Offset(0,0) - Constraints(0,0), 0.0))
