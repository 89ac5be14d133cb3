-- Embedded deformation (Sumner et al.): one affine frame per graph node, kept near a rotation.  problemparams layout:
--   [0] w_fitSqrt, [1] w_regSqrt, [2] w_rotSqrt   float (host)
--   [3] Offset       opt_float3[N]   unknown node positions
--   [4] RotMatrix    opt_float9[N]   unknown 3x3 frames, row-major
--   [5] UrShape      opt_float3[N]   rest positions
--   [6] Constraints  opt_float3[N]   handle targets; x < -999999.9 marks a free node
--   [7] G            int (host)      number of half-edges;  [8] v0, [9] v1: int[G]
local N = Dim("N", 0)
local w_fitSqrt = Param("w_fitSqrt", float, 0)
local w_regSqrt = Param("w_regSqrt", float, 1)
local w_rotSqrt = Param("w_rotSqrt", float, 2)
local Offset = Unknown("Offset", opt_float3, {N}, 3)
local RotMatrix = Unknown("RotMatrix", opt_float9, {N}, 4)
local UrShape = Image("UrShape", opt_float3, {N}, 5)
local Constraints = Image("Constraints", opt_float3, {N}, 6)
local G = Graph("G", 7, "v0", {N}, 8, "v1", {N}, 9)

UsePreconditioner(true)

-- handles
local pinned = greatereq(Constraints(0)(0), -999999.9)
Energy(Select(pinned, w_fitSqrt * (Offset(0) - Constraints(0)), 0))

-- the frame's columns should be orthonormal
local M = RotMatrix(0)
local col = { Vector(M(0), M(3), M(6)), Vector(M(1), M(4), M(7)), Vector(M(2), M(5), M(8)) }
Energy(w_rotSqrt * Dot3(col[1], col[2]))
Energy(w_rotSqrt * Dot3(col[1], col[3]))
Energy(w_rotSqrt * Dot3(col[2], col[3]))
Energy(w_rotSqrt * (Dot3(col[1], col[1]) - 1))
Energy(w_rotSqrt * (Dot3(col[2], col[2]) - 1))
Energy(w_rotSqrt * (Dot3(col[3], col[3]) - 1))

-- a node's frame should carry its rest edges onto the deformed ones
local predicted = Matrix3x3Mul(RotMatrix(G.v0), UrShape(G.v1) - UrShape(G.v0))
Energy(w_regSqrt * ((Offset(G.v1) - Offset(G.v0)) - predicted))
