#!/bin/bash
# N-rank bench on ONE GPU (functional evidence of the multi-rank path with the final tree: pre-flight, both communicator legs, JSON line): 2, 4, 8 ranks at 4096^2 and 8 ranks at 8192^2
cd /root/repo
mkdir -p gpurun_out/r04m
for n in 2 4 8; do
  timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $((29500 + n)) bench.py --gpus $n --share-gpu --steps 2 --warmup 1 --liters 50 --no-cpu-baseline \
    > gpurun_out/r04m/bench_${n}ranks_shared_gpu_4096.out 2> gpurun_out/r04m/bench_${n}ranks_shared_gpu_4096.err
  grep "^{" gpurun_out/r04m/bench_${n}ranks_shared_gpu_4096.out | tail -1 > gpurun_out/r04m/bench_${n}ranks_shared_gpu_4096.json
  python -c "
import json; d=json.load(open('gpurun_out/r04m/bench_${n}ranks_shared_gpu_4096.json')); print($n, d.get('value'), d.get('comm_ranks'), d.get('error'), (d.get('rccl_leg') or {}).get('status'), d.get('kernel_src_sha16'))"
done
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29520 bench.py --gpus 8 --share-gpu --size 8192 --steps 1 --warmup 1 --liters 30 --no-cpu-baseline \
  > gpurun_out/r04m/bench_8ranks_shared_gpu_8192.out 2> gpurun_out/r04m/bench_8ranks_shared_gpu_8192.err
grep "^{" gpurun_out/r04m/bench_8ranks_shared_gpu_8192.out | tail -1 > gpurun_out/r04m/bench_8ranks_shared_gpu_8192.json
python -c "
import json; d=json.load(open('gpurun_out/r04m/bench_8ranks_shared_gpu_8192.json')); print(8192, d.get('value'), d.get('comm_ranks'), d.get('error'))"
