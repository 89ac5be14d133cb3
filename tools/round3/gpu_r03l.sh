#!/bin/bash
# SQ counters of the image_warping iteration kernel at 4096^2 and on a 1/8 slab
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r03l; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
C="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_BUSY_CYCLES"
timeout 200 rocprofv3 --pmc $C -f csv -d $O/sq4096 -o p -- python bench.py --steps 1 --warmup 0 --liters 30 --no-cpu-baseline --no-extras > $O/sq4096.log 2>&1
timeout 200 rocprofv3 --pmc SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_INSTS_SMEM -f csv -d $O/sq4096b -o p -- python bench.py --steps 1 --warmup 0 --liters 30 --no-cpu-baseline --no-extras > $O/sq4096b.log 2>&1
timeout 200 rocprofv3 --pmc $C -f csv -d $O/sq2048 -o p -- python bench.py --size 2048 --steps 1 --warmup 0 --liters 30 --no-cpu-baseline --no-extras > $O/sq2048.log 2>&1
python - <<'PY'
import csv, glob, collections
for tag in ("sq4096", "sq4096b", "sq2048"):
    files = glob.glob(f"gpurun_out/r03l/{tag}/**/*counter_collection.csv", recursive=True)
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for f in files:
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "iw_pcgIter2" not in k: continue
            k = k[k.index("iw_pcgIter2"):].split("(")[0]
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            cnt[(k, r["Counter_Name"])] += 1
    for k, v in agg.items():
        print(tag, k, {c: round(x / cnt[(k, c)]) for c, x in v.items()})
PY
