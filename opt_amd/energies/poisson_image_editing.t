-- Gradient-domain ("Poisson") image blending in Opt's energy DSL.
-- problemparams layout (all device pointers, row-major, channels interleaved):
--   [0] X   opt_float4[W*H]   unknown, initialised to the base image
--   [1] T   opt_float4[W*H]   image whose gradients are pasted in
--   [2] M   opt_float [W*H]   0 = pixel is solved for, anything else = pixel keeps its value
local W, H = Dim("W", 0), Dim("H", 1)
local X = Unknown("X", opt_float4, {W,H}, 0)
local T = Array("T", opt_float4, {W,H}, 1)
local M = Array("M", opt_float, {W,H}, 2)

UsePreconditioner(false)

-- only pixels inside the pasted region are unknowns
Exclude(Not(eq(M(0,0), 0)))

-- match the gradient of X to the gradient of T along every edge of the 4-neighbourhood
local offsets = { {1,0}, {-1,0}, {0,1}, {0,-1} }
for dx, dy in Stencil(offsets) do
    local gradX = X(0,0) - X(dx,dy)
    local gradT = T(0,0) - T(dx,dy)
    Energy(Select(InBounds(dx,dy), gradX - gradT, 0))
end
