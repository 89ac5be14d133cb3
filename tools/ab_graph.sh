#!/bin/bash
# gather vs scatter mode of the graph functor engine on the three mesh examples: tools/ab_graph.sh
cd $GRAFT_REPO_ROOT
for cfg in cotangent embedded robust; do
  for m in 1 0; do
    echo "== $cfg gather=$m"
    OPT_AMD_GRAPH_GATHER=$m OPT_AMD_CONFIG="$cfg" timeout 120 python tools/bench_configs.py 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    l = l.strip()
    if l.startswith('{'):
        d = json.loads(l); k = d['kernel_avg_us']
        print('wall %.1f ms  cost %.6g  ' % (d['wall_s'] * 1e3, d['cost_final']), {n: k[n] for n in k if 'Step1' in n or 'Init1' in n or 'Incidence' in n})
"
  done
done
