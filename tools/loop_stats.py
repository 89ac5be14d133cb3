#!/usr/bin/env python
"""Instruction mix of the largest loop of a kernel in a gfx950 assembly listing: tools/loop_stats.py <file.s> <mangled-name substring>"""
import collections
import re
import sys


def main():
    lines = open(sys.argv[1]).read().split("\n")
    key = sys.argv[2]
    start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\S*" + re.escape(key) + r"\S*:", l))
    end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
    body = lines[start:end]
    labels = {m.group(1): i for i, l in enumerate(body) for m in [re.match(r"^(\.LBB\d+_\d+):", l)] if m}
    loops = []
    for i, l in enumerate(body):
        m = re.search(r"s_c?branch\w* (\.LBB\d+_\d+)", l)
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            loops.append((labels[m.group(1)], i))
    print(lines[start][:80], "loops:", [(a, b, b - a) for a, b in loops])
    a, b = max(loops, key=lambda t: t[1] - t[0])
    c = collections.Counter()
    for l in body[a:b]:
        l = l.strip()
        if not l or l[0] in ".;":
            continue
        op = l.split()[0]
        if op.startswith("v_"):
            c["valu"] += 1
            if "f64" in op: c["f64"] += 1
            if "dpp" in l: c["dpp"] += 1
            if op.startswith(("v_div", "v_rcp", "v_sqrt", "v_rsq")): c["div/rcp"] += 1
            if op.startswith("v_mov") and "dpp" not in l: c["mov"] += 1
            if op.startswith("v_cndmask"): c["cndmask"] += 1
        elif op.startswith("s_waitcnt"): c["waitcnt"] += 1
        elif op.startswith("s_"): c["salu"] += 1
        elif op.startswith(("global_load", "buffer_load")): c["load"] += 1
        elif op.startswith(("global_store", "buffer_store")): c["store"] += 1
        elif op.startswith("scratch"): c["scratch"] += 1
        elif op.startswith("ds_"): c["lds"] += 1
    print(dict(c))
    print("waits:", [l.strip() for l in body[a:b] if "s_waitcnt" in l])


if __name__ == "__main__":
    main()
