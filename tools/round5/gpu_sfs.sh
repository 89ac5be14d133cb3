#!/bin/bash
# round 5: on-chip shape_from_shading -- first light, then the parity tests of the new path
mkdir -p gpurun_out/r05s
timeout 600 python -u tools/round5/dbg_sfs.py > gpurun_out/r05s/dbg.txt 2>&1; echo "rc=$?" >> gpurun_out/r05s/dbg.txt
grep -c OK gpurun_out/r05s/dbg.txt; grep -c BAD gpurun_out/r05s/dbg.txt; grep BAD gpurun_out/r05s/dbg.txt | head -n 20; tail -n 30 gpurun_out/r05s/dbg.txt
timeout 900 python -m pytest tests/test_onchip_sfs_gpu.py -q -m gpu --maxfail=12 -p no:cacheprovider > gpurun_out/r05s/pytest.txt 2>&1; echo "rc=$?" >> gpurun_out/r05s/pytest.txt
tail -n 30 gpurun_out/r05s/pytest.txt
