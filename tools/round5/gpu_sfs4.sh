#!/bin/bash
bash tools/round5/gpu_sfs3.sh | grep "rows=6 waves=4\|rows=10\|rc=" | awk 'NR%3==1'
unset OPT_AMD_LIB OPT_AMD_ONCHIP_PROFILE
bash tools/round5/gpu_sfs.sh | grep "flag=1 R=None\|passed\|failed\|^[0-9]*$\|BAD"
