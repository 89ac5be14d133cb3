#!/bin/bash
# round 6, final call: the whole GPU suite on the final tree (bars in force), smoke, then every artifact of profiles/r06_*
out=gpurun_out/r06z; mkdir -p $out
timeout 1800 python -m pytest tests -m gpu -q > $out/gpu_suite.txt 2>&1
grep -n "passed\|failed\|^FAILED\|^ERROR" $out/gpu_suite.txt | tail -20
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $out/smoke.txt 2>&1; tail -n 2 $out/smoke.txt
bash tools/round6/artifacts.sh r06 > $out/artifacts.log 2>&1
tail -n 30 $out/artifacts.log
