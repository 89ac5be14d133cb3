#!/bin/bash
mkdir -p gpurun_out/r05a
for rows in 2 4 8; do
  echo "=== rows $rows"; OPT_AMD_ONCHIP_ROWS=$rows python -u tools/round5/dbg_lm.py LMGPU 517 33 0 9 2 3669 2>&1 | tail -n 7
done
echo "=== streaming"; OPT_AMD_ONCHIP=0 python -u tools/round5/dbg_lm.py LMGPU 517 33 0 9 2 3669 2>&1 | tail -n 7
echo "=== rows 8 no mask"; OPT_AMD_ONCHIP_ROWS=8 python -u tools/round5/dbg_lm.py LMGPU 517 33 0 9 2 3669 0 2>&1 | tail -n 7
echo "=== rows 8 period 10"; OPT_AMD_ONCHIP_ROWS=8 python -u tools/round5/dbg_lm.py LMGPU 517 33 0 9 10 3669 2>&1 | tail -n 7
echo "=== rows 8 double"; OPT_AMD_ONCHIP_ROWS=8 python -u tools/round5/dbg_lm.py LMGPU 517 33 1 9 2 3669 2>&1 | tail -n 7
echo "=== rows 8 300x40"; OPT_AMD_ONCHIP_ROWS=8 python -u tools/round5/dbg_lm.py LMGPU 300 40 0 9 2 3669 2>&1 | tail -n 7
