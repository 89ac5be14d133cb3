#!/bin/bash
mkdir -p gpurun_out/r05c
timeout 900 python -m pytest tests/test_onchip_sfs_gpu.py tests/test_onchip_stencil_gpu.py tests/test_cpp_callers_gpu.py -q -m gpu --maxfail=5 -p no:cacheprovider > gpurun_out/r05c/pytest2.txt 2>&1; echo "rc=$?" >> gpurun_out/r05c/pytest2.txt
tail -n 3 gpurun_out/r05c/pytest2.txt
OPT_AMD_CONFIG=config1 python tools/bench_configs.py 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.readline()); print('config1 wall ms', r['wall_s']*1e3, r['kernel_avg_us'])"
OPT_AMD_CONFIG=config3 python tools/bench_configs.py 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.readline()); print('config3 wall ms', r['wall_s']*1e3)"
