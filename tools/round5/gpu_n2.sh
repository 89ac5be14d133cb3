#!/bin/bash
mkdir -p gpurun_out/r05n
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --share-gpu --steps 2 --warmup 1 > gpurun_out/r05n/n2.json 2> gpurun_out/r05n/n2.err; echo "rc=$?"
tail -c 1500 gpurun_out/r05n/n2.json; tail -n 5 gpurun_out/r05n/n2.err
