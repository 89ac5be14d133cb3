#!/usr/bin/env python
"""Back in the container: copy what tools/round6/artifacts.sh left under gpurun_out/<tag>/ into profiles/ (tracked).  Multi-rank outputs carry launcher chatter on stdout:
only their JSON line is kept."""
import json, os, shutil, sys
tag = sys.argv[1] if len(sys.argv) > 1 else "r06"
src, dst = os.path.join("gpurun_out", tag), "profiles"
plain = {"bench.json": f"{tag}_bench.json", f"{tag}_configs.json": None, f"{tag}_kernel_stats.csv": None, f"{tag}_summary.md": None, f"{tag}_traffic.json": None, f"{tag}_rocm_smi.txt": None,
         f"{tag}_config1_kernel_stats.csv": None, f"{tag}_config3_kernel_stats.csv": None, f"{tag}_config4_kernel_stats.csv": None, f"{tag}_poisson_image_editing_kernel_stats.csv": None,
         "slab_overhead.txt": f"{tag}_slab_overhead.txt", "horizon_parity.md": f"{tag}_horizon_parity.md", "horizon_parity.json": f"{tag}_horizon_parity.json",
         "config_horizon.json": f"{tag}_config_horizon.json", "config_horizon.txt": f"{tag}_config_horizon.txt", "onchip_kernel_stats.csv": f"{tag}_onchip_kernel_stats.csv", "configs.json": f"{tag}_configs_raw.json"}
for a, b in plain.items():
    if os.path.exists(os.path.join(src, a)):
        shutil.copy(os.path.join(src, a), os.path.join(dst, b or a))
for a, b in {"dry_1.json": f"{tag}_dry_1.json", "dry_2_4096.json": f"{tag}_dry_2_4096.json", "dry_8_4096.json": f"{tag}_dry_8_4096.json", "bench_8ranks_shared_gpu_4096.json": f"{tag}_bench_8ranks_shared_gpu_4096.json"}.items():
    p = os.path.join(src, a)
    if os.path.exists(p):
        lines = [l for l in open(p) if l.startswith("{")]
        if lines:
            json.loads(lines[-1])
            open(os.path.join(dst, b), "w").write(lines[-1])
b = json.load(open(os.path.join(dst, f"{tag}_bench.json")))
print(b["value"], b["roofline"]["frac"], b["roofline"].get("hbm_frac"), b["roofline"].get("frac_of_box_copy_float4"), b["kernel_src_sha16"], json.load(open(os.path.join(dst, f"{tag}_traffic.json")))["kernel_src_sha16"])
