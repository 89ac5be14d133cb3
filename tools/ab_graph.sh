#!/bin/bash
# graph functor engine on the three mesh examples: gather vs scatter mode, or library variants (names under opt_amd/lib/):
#   tools/ab_graph.sh                 gather=1 / gather=0 with libOpt.so
#   tools/ab_graph.sh libA.so libB.so gather mode with each library
cd $GRAFT_REPO_ROOT
show='
import sys, json
for l in sys.stdin:
    l = l.strip()
    if l.startswith("{"):
        d = json.loads(l); k = d["kernel_avg_us"]
        print("wall %.1f ms  cost %.6g  " % (d["wall_s"] * 1e3, d["cost_final"]), {n: k[n] for n in k if "Step1" in n or "Init1" in n or "Incidence" in n})
'
for cfg in cotangent embedded robust; do
  if [ $# -gt 0 ]; then
    for lib in "$@"; do echo "== $cfg $lib"; OPT_AMD_LIB=$GRAFT_REPO_ROOT/opt_amd/lib/$lib OPT_AMD_CONFIG="$cfg" timeout 120 python tools/bench_configs.py 2>/dev/null | python -c "$show"; done
  else
    for m in 1 0; do echo "== $cfg gather=$m"; OPT_AMD_GRAPH_GATHER=$m OPT_AMD_CONFIG="$cfg" timeout 120 python tools/bench_configs.py 2>/dev/null | python -c "$show"; done
  fi
done
