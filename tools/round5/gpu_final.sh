#!/bin/bash
# round 5: the final full GPU suite + smoke, then everything profiles/r05_* is made of
bash tools/round5/gpu_full.sh | tail -n 4
bash tools/round5/artifacts.sh r05 | tail -n 12
