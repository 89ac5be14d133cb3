-- Smallest image energy: stay close to a target image A while keeping the forward differences of X small.
-- problemparams: [0] X float[W*H] unknown, [1] A float[W*H] target (both device pointers).
W, H = Dim("W", 0), Dim("H", 1)
X = Unknown("X", float, {W,H}, 0)
A = Array("A", float, {W,H}, 1)
local fitWeight = .2
Energy(fitWeight * (X(0,0) - A(0,0)))   -- data term
Energy(X(0,0) - X(1,0))                 -- horizontal smoothness
Energy(X(0,0) - X(0,1))                 -- vertical smoothness
