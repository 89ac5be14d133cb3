-- Graph-only energy: fit y = a*cos(b*x) + b*sin(a*x) to N samples; the two parameters live in a one-element
-- unknown array and every sample is a hyperedge {sample, parameters}.
-- problemparams: [0] funcParams opt_float2[U] (device, unknown)   [1] data opt_float2[N] (device)
--                [2] edge count (HOST int*)   [3] "d": sample index per edge (device int*)   [4] "p": parameter index per edge (device int*)
N, U = Dim("N", 0), Dim("U", 1)
funcParams = Unknown("funcParams", opt_float2, {U}, 0)
data       = Image("data", opt_float2, {N}, 1)
local G = Graph("G", 2, "d", {N}, 3, "p", {U}, 4)
UsePreconditioner(true)

local sample = data(G.d)
local x, y = sample(0), sample(1)
local params = funcParams(G.p)
local a, b = params(0), params(1)
Energy(y - (a*cos(b*x) + b*sin(a*x)))
