#!/usr/bin/env python
"""VGPR / SGPR / scratch usage of the kernels in a gfx950 assembly listing (hipcc --save-temps): tools/kernel_regs.py <file.s> [name filter ...]"""
import re
import subprocess
import sys


def main():
    txt = open(sys.argv[1]).read()
    filt = sys.argv[2:]
    rows = []
    for m in re.finditer(r"\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel", txt, re.S):
        name, body = m.group(1), m.group(2)
        v = re.search(r"\.amdhsa_next_free_vgpr (\d+)", body).group(1)
        a = re.search(r"\.amdhsa_accum_offset (\d+)", body)
        sc = re.search(r"\.amdhsa_private_segment_fixed_size (\d+)", body).group(1)
        rows.append((name, int(v), int(a.group(1)) if a else 0, int(sc)))
    names = subprocess.run(["c++filt"] + [r[0] for r in rows], capture_output=True, text=True).stdout.splitlines()
    for (n, v, a, sc), d in zip(rows, names):
        d = d.replace("optamd::(anonymous namespace)::", "")
        d = re.sub(r"\(.*", "", d)
        if filt and not any(f in d for f in filt):
            continue
        print(f"vgpr+agpr {v:4d} (arch vgpr {a:4d})  scratch {sc:5d} B  {d}")


if __name__ == "__main__":
    main()
