-- As-rigid-as-possible deformation of a regular 3-D lattice.  problemparams layout:
--   [0] Offset       opt_float3[W*H*D]   unknown: deformed lattice positions
--   [1] Angle        opt_float3[W*H*D]   unknown: per-node Euler angles
--   [2] UrShape      opt_float3[W*H*D]   rest positions
--   [3] Constraints  opt_float3[W*H*D]   target positions; x < -999999.9 marks an unconstrained node
--   [4] w_fitSqrt    float (host)
--   [5] w_regSqrt    float (host)
local W, H, D = Dim("W", 0), Dim("H", 1), Dim("D", 2)
local Offset = Unknown("Offset", opt_float3, {W,H,D}, 0)
local Angle = Unknown("Angle", opt_float3, {W,H,D}, 1)
local UrShape = Array("UrShape", opt_float3, {W,H,D}, 2)
local Constraints = Array("Constraints", opt_float3, {W,H,D}, 3)
local w_fitSqrt = Param("w_fitSqrt", float, 4)
local w_regSqrt = Param("w_regSqrt", float, 5)

UsePreconditioner(true)

-- handles
local constrained = greatereq(Constraints(0,0,0)(0), -999999.9)
Energy(Select(constrained, w_fitSqrt * (Offset(0,0,0) - Constraints(0,0,0)), 0))

-- every lattice edge should be the rotated rest edge
for dx, dy, dz in Stencil { {1,0,0}, {-1,0,0}, {0,1,0}, {0,-1,0}, {0,0,1}, {0,0,-1} } do
    local edge = (Offset(0,0,0) - Offset(dx,dy,dz)) - Rotate3D(Angle(0,0,0), UrShape(0,0,0) - UrShape(dx,dy,dz))
    Energy(w_regSqrt * Select(InBounds(0,0,0), Select(InBounds(dx,dy,dz), edge, 0.0), 0.0))
end
