#!/bin/bash
# first GPU call of round 2: new parity tests + peer communicator + bench
out=$GRAFT_REPO_ROOT/gpurun_out/r02a
mkdir -p $out
cd $GRAFT_REPO_ROOT
nproc > $out/nproc.txt
timeout 900 python -m pytest tests/test_steady_state_gpu.py -q -m gpu --timeout 300 -k "not frozen_oracle" --durations=15 > $out/steady.log 2>&1
tail -25 $out/steady.log
timeout 600 python -m pytest tests/test_peer_comm_gpu.py tests/test_slab_gpu.py -q -m gpu --timeout 200 --durations=5 > $out/peer.log 2>&1
tail -25 $out/peer.log
timeout 600 python -m pytest tests -q -m gpu --timeout 300 --deselect tests/test_steady_state_gpu.py --deselect tests/test_peer_comm_gpu.py --deselect tests/test_slab_gpu.py > $out/rest.log 2>&1
tail -5 $out/rest.log
timeout 600 python bench.py > $out/bench.json 2> $out/bench.err
tail -c 3000 $out/bench.json; tail -5 $out/bench.err
