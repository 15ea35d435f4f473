#!/usr/bin/env python3
"""Static instruction counts of one kernel of a `-save-temps -gline-tables-only` build, by source line
and by basic block.  usage: tools/isa_by_line.py file.s kernel_substring [--blocks]
(compile with the product flags + `-gline-tables-only -save-temps`)"""
import collections
import re
import sys


def main():
    path, key = sys.argv[1], sys.argv[2]
    show_blocks = "--blocks" in sys.argv
    files = {}
    lines = open(path).read().split("\n")
    start = end = None
    for n, l in enumerate(lines):
        m = re.match(r"\s*\.file\s+(\d+)\s+\"[^\"]*\"\s+\"([^\"]+)\"", l)
        if m:
            files[int(m.group(1))] = m.group(2)
        if start is None and re.match(r"^_Z\w*%s\w*:" % key, l):
            start = n
        if start is not None and end is None and ".end_amdhsa_kernel" in l and n > start:
            end = n
    by_line = collections.defaultdict(collections.Counter)
    blocks = []
    cur = None
    loc = (0, 0)
    kinds = collections.Counter()
    for l in lines[start:end]:
        s = l.strip()
        m = re.match(r"\.loc\s+(\d+)\s+(\d+)", s)
        if m:
            loc = (int(m.group(1)), int(m.group(2)))
            continue
        m = re.match(r"^(\.LBB\w+):", s)
        if m:
            cur = [m.group(1), collections.Counter(), collections.Counter(), []]
            blocks.append(cur)
            continue
        if not s or s.startswith((".", ";", "//")) or s.endswith(":"):
            continue
        op = s.split()[0]
        if op.startswith("v_readlane") or op.startswith("v_writelane"):
            k = "lane"
        elif op.startswith("v_"):
            k = "valu"
        elif op.startswith("s_"):
            k = "salu"
        elif op.startswith("ds_"):
            k = "lds"
        elif op.startswith(("global_", "buffer_", "flat_", "scratch_")):
            k = "vmem"
        else:
            k = "other"
        kinds[k] += 1
        by_line[loc][k] += 1
        if cur is not None:
            cur[1][k] += 1
            cur[2][loc] += 1
            if op.startswith(("s_cbranch", "s_branch")):
                cur[3].append(s.split()[-1])
    print("kernel lines", start, end, dict(kinds))
    if show_blocks:
        for name, c, locs, br in blocks:
            if sum(c.values()) < 8:
                continue
            top = ", ".join(f"{files.get(f, f)}:{ln}x{k}" for (f, ln), k in locs.most_common(3))
            print(f"{name:12s} valu {c['valu']:4d} lane {c['lane']:3d} salu {c['salu']:4d} lds {c['lds']:3d} vmem {c['vmem']:3d} -> {' '.join(br)}   [{top}]")
    else:
        rows = sorted(by_line.items(), key=lambda kv: -sum(kv[1].values()))
        for (f, ln), c in rows[:70]:
            print(f"{files.get(f, f)}:{ln:5d}  valu {c['valu']:4d} lane {c['lane']:3d} salu {c['salu']:4d} lds {c['lds']:3d} vmem {c['vmem']:3d}")


main()
