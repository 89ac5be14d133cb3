#!/bin/bash
# PCG iteration rate of the metric workload (image_warping, GN, float) over image sizes:  tools/size_sweep.sh > table.md
cd $GRAFT_REPO_ROOT
echo "| image | PCG it/s | iteration kernel us (hipEvents) | Gpixel-iterations/s | us per iteration (wall) |"
echo "|---|---|---|---|---|"
for s in 512 1024 2048 4096 8192; do
  timeout 300 python bench.py --size $s --steps 2 --warmup 1 --liters 200 --no-cpu-baseline 2>/dev/null | S=$s python -c "
import sys, json, os
s = int(os.environ['S'])
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']
print('| %dx%d | %.0f | %.1f | %.1f | %.1f |' % (s, s, d['value'], r['avg_kernel_ms'] * 1e3, d['value'] * s * s / 1e9, 1e6 / d['value']))"
done
