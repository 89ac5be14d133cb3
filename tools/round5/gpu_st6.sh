#!/bin/bash
bash tools/round5/gpu_st.sh | tail -n 4
bash tools/round5/gpu_st2.sh | grep "intrinsic\|optical_flow 1024" | cut -c1-260
