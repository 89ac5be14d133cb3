-- Shading-based depth refinement in Opt's energy DSL (one scalar unknown per pixel: the refined depth X).
-- problemparams layout:
--   [0..2]  w_p, w_s, w_g   host float : fitting / smoothness / shading weights (the energy uses their square roots)
--   [3..6]  f_x, f_y, u_x, u_y host float : camera intrinsics
--   [7..15] L_1 .. L_9      host float : spherical-harmonics lighting coefficients
--   [16] X  opt_float[W*H] device, unknown   [17] D_i opt_float[W*H] input depth (<= 0: no data)   [18] Im opt_float[W*H] intensity
--   [19] edgeMaskR uint8[W*H]   [20] edgeMaskC uint8[W*H] : 0 suppresses the horizontal / vertical shading-gradient term
local DEPTH_JUMP = 0.01
local W, H = Dim("W", 0), Dim("H", 1)

local w_p = sqrt(Param("w_p", float, 0))
local w_s = sqrt(Param("w_s", float, 1))
local w_g = sqrt(Param("w_g", float, 2))
local f_x = Param("f_x", float, 3)
local f_y = Param("f_y", float, 4)
local u_x = Param("u_x", float, 5)
local u_y = Param("u_y", float, 6)
local L = {}
for k = 1, 9 do L[k] = Param("L_" .. k, float, 6 + k) end

local X         = Unknown("X", opt_float, {W,H}, 16)
local D_i       = Array("D_i", opt_float, {W,H}, 17)
local Im        = Array("Im", opt_float, {W,H}, 18)
local edgeMaskR = Array("edgeMaskR", uint8, {W,H}, 19)
local edgeMaskC = Array("edgeMaskC", uint8, {W,H}, 20)

local px, py = Index(0), Index(1)
local function hasDepth(x, y) return greater(D_i(x,y), 0) end

-- back-projected point of the pixel at offset (ox,oy)
local function point(ox, oy)
    local d = X(ox,oy)
    return Vector(((ox + px - u_x) / f_x) * d, ((oy + py - u_y) / f_y) * d, d)
end

-- unit normal from the depth at (ox,oy) and its left / upper neighbours
local function normal(ox, oy)
    local i, j = ox + px, oy + py
    local nx = X(ox, oy - 1) * (X(ox, oy) - X(ox - 1, oy)) / f_y
    local ny = X(ox - 1, oy) * (X(ox, oy) - X(ox, oy - 1)) / f_x
    local nz = (nx * (u_x - i) / f_x) + (ny * (u_y - j) / f_y) - (X(ox - 1, oy) * X(ox, oy - 1) / (f_x * f_y))
    local len2 = nx*nx + ny*ny + nz*nz
    local inv = Select(greater(len2, 0.0), 1.0 / sqrt(len2), 1.0)
    return inv * Vector(nx, ny, nz)
end

local function shading(ox, oy)
    local n = normal(ox, oy)
    local nx, ny, nz = n[0], n[1], n[2]
    return L[1] + L[2]*ny + L[3]*nz + L[4]*nx + L[5]*nx*ny + L[6]*ny*nz + L[7]*(-nx*nx - ny*ny + 2*nz*nz) + L[8]*nz*nx + L[9]*(nx*nx - ny*ny)
end

local function intensity(ox, oy)
    return Im(ox,oy)*0.5 + 0.25*(Im(ox - 1, oy) + Im(ox, oy - 1))
end

-- shading error, stored (with its derivatives) as a computed image
local function shadingError(x, y)
    local usable = hasDepth(x - 1, y) * hasDepth(x, y) * hasDepth(x, y - 1)
    return Select(InBoundsExpanded(0,0,1) * usable, shading(x,y) - intensity(x,y), 0)
end
local B_I = ComputedArray("B_I", {W,H}, shadingError(0,0))

Exclude(Not(hasDepth(0,0)))

-- stay close to the measured depth
Energy(Select(hasDepth(0,0), w_p * (X(0,0) - D_i(0,0)), 0))

-- gradients of the shading error should vanish (except across masked edges)
Energy(Select(InBoundsExpanded(0,0,1), w_g * ((B_I(0,0) - B_I(1,0)) * edgeMaskR(0,0)), 0))
Energy(Select(InBoundsExpanded(0,0,1), w_g * ((B_I(0,0) - B_I(0,1)) * edgeMaskC(0,0)), 0))

-- Laplacian smoothness of the back-projected surface where the depth is continuous
local function smooth(x, y) return less(abs(X(0,0) - X(x,y)), DEPTH_JUMP) end
local regular = hasDepth(0,0) * hasDepth(0,-1) * hasDepth(0,1) * hasDepth(-1,0) * hasDepth(1,0) *
                smooth(0,-1) * smooth(0,1) * smooth(-1,0) * smooth(1,0) * InBoundsExpanded(0,0,1)
local regularImage = ComputedArray("valid", {W,H}, regular)
Energy(Select(eq(regularImage(0,0), 1), w_s * (4.0*point(0,0) - (point(-1,0) + point(0,-1) + point(1,0) + point(0,1))), 0))
