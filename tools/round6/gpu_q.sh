#!/bin/bash
out=gpurun_out/r06q; mkdir -p $out; rm -f $out/parity_log.jsonl
export OPT_PARITY_LOG=$PWD/$out/parity_log.jsonl
timeout 1800 python -m pytest tests -m gpu -q > $out/gpu_suite.txt 2>&1
unset OPT_PARITY_LOG
grep -n "passed\|failed\|^FAILED\|^ERROR" $out/gpu_suite.txt | tail -10
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
