-- Dense optical flow (brightness constancy + first-order smoothness) in Opt's energy DSL.
-- problemparams layout:
--   [0] w_fit     float (host)        sqrt of the data weight
--   [1] w_reg     float (host)        sqrt of the smoothness weight
--   [2] X         opt_float2[W*H]     unknown flow (u, v) per pixel
--   [3] I         opt_float [W*H]     source frame
--   [4] I_hat     opt_float [W*H]     target frame (sampled bilinearly at pixel + flow)
--   [5] I_hat_dx  opt_float [W*H]     its x-derivative image   (partials of the sample operator)
--   [6] I_hat_dy  opt_float [W*H]     its y-derivative image
local W, H = Dim("W", 0), Dim("H", 1)
local w_fit = Param("w_fit", float, 0)
local w_reg = Param("w_reg", float, 1)
local X = Unknown("X", opt_float2, {W,H}, 2)
local I = Array("I", opt_float, {W,H}, 3)
local T_im = Array("I_hat", opt_float, {W,H}, 4)
local T_dx = Array("I_hat_dx", opt_float, {W,H}, 5)
local T_dy = Array("I_hat_dy", opt_float, {W,H}, 6)
local target = SampledImage(T_im, T_dx, T_dy)
local i, j = Index(0), Index(1)

UsePreconditioner(false)

-- data term: the target frame, looked up where the flow points, should show what the source frame shows here
Energy(w_fit * (I(0,0) - target(i + X(0,0,0), j + X(0,0,1))))

-- smoothness: neighbouring pixels move alike
for dx, dy in Stencil { {1,0}, {-1,0}, {0,1}, {0,-1} } do
    Energy(Select(InBounds(dx,dy), w_reg * (X(0,0) - X(dx,dy)), 0))
end
