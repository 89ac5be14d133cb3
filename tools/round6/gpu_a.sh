#!/bin/bash
# round 6, call A: the box's ceilings (float4 copy, cooperative launch cost) and the round-5 tree's bench line on this round's box
out=gpurun_out/r06a; mkdir -p $out
tools/bin/mb_coop > $out/mb_coop.txt 2>&1
python bench.py > $out/bench.json 2> $out/bench.err
tail -n 30 $out/mb_coop.txt; tail -c 1500 $out/bench.json
