#!/bin/bash
bash tools/round5/gpu_st.sh | tail -n 6
bash tools/round5/gpu_st2.sh | grep "onchip=1"
