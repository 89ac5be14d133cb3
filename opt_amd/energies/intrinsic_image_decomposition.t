-- Intrinsic image decomposition: log-image i = log-albedo r + log-shading s, sparse (L_p, p < 1) albedo gradients,
-- smooth shading.  problemparams layout:
--   [0] w_fitSqrt         float (host)
--   [1] w_regSqrtAlbedo   float (host)
--   [2] w_regSqrtShading  float (host)
--   [3] pNorm             opt_float (host)   exponent p of the albedo prior
--   [4] r                 opt_float3[W*H]    unknown log-albedo (also read as a constant, "r_const", for the L_p weights)
--   [5] i                 opt_float3[W*H]    log input image
--   [6] s                 opt_float [W*H]    unknown log-shading
local W, H = Dim("W", 0), Dim("H", 1)
local w_fitSqrt = Param("w_fitSqrt", float, 0)
local w_regSqrtAlbedo = Param("w_regSqrtAlbedo", float, 1)
local w_regSqrtShading = Param("w_regSqrtShading", float, 2)
local pNorm = Param("pNorm", opt_float, 3)
local r = Unknown("r", opt_float3, {W,H}, 4)
local r_const = Array("r_const", opt_float3, {W,H}, 4)
local i = Array("i", opt_float3, {W,H}, 5)
local s = Unknown("s", opt_float, {W,H}, 6)

local neighbours = { {1,0}, {-1,0}, {0,1}, {0,-1} }

-- albedo prior: iteratively re-weighted least squares form of |grad r|^p; the weight is frozen per nonlinear iteration
for dx, dy in Stencil(neighbours) do
    local lp = L_p(r(0,0) - r(dx,dy), r_const(0,0) - r_const(dx,dy), pNorm, {W,H})
    Energy(w_regSqrtAlbedo * Select(InBounds(0,0), Select(InBounds(dx,dy), lp, 0), 0))
end

-- shading prior
for dx, dy in Stencil(neighbours) do
    Energy(w_regSqrtShading * Select(InBounds(0,0), Select(InBounds(dx,dy), s(0,0) - s(dx,dy), 0), 0))
end

-- reconstruction
Energy(w_fitSqrt * (r(0,0) + s(0,0) - i(0,0)))
