#!/usr/bin/env python3
"""Issue-weighted instruction count of a basic-block range of a kernel's assembly.

usage: tools/isa_weights.py file.s <first label or line> <last label or line>
Weights are the measured issue costs of tools/valu_ceiling.hip (profiles/r03/valu_ceiling.json) in units of
a full-rate wave64 VALU instruction (one per 2 clocks and SIMD): 1 = v_add/sub/mul/fma/fmac_f32,
v_add/sub_u32, v_and/or/xor_b32, v_lshrrev_b32, v_mov_b32 on VGPR / inline-constant / literal operands;
2 = everything else (conversions, compares, v_cndmask, min/max, three-operand integer forms, anything with
an SGPR operand, DPP, readlane); 4 = v_sqrt / v_rcp / v_rsq / v_exp / v_log / v_sin / v_cos.
"""
import re
import sys

FULL = {"v_add_f32", "v_sub_f32", "v_subrev_f32", "v_mul_f32", "v_fma_f32", "v_fmac_f32", "v_add_u32", "v_sub_u32",
        "v_subrev_u32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_lshrrev_b32", "v_mov_b32", "v_add_u16", "v_sub_u16"}
QUARTER = ("v_sqrt_f32", "v_rcp_f32", "v_rsq_f32", "v_exp_f32", "v_log_f32", "v_sin_f32", "v_cos_f32")


def weight(line):
    m = re.match(r"\s*(v_\w+)\s+(.*)", line)
    if not m:
        return None
    op, args = m.group(1), m.group(2).split(";")[0]
    base = re.sub(r"_(e32|e64|sdwa|dpp)$", "", op)
    if base.startswith(QUARTER):
        return 4
    if op.endswith("_dpp") or op.endswith("_sdwa"):
        return 2
    srcs = args.split(",")[1:]
    scalar = any(re.search(r"\b(s\d+|s\[\d+:\d+\]|vcc|exec|m0)\b", a) for a in srcs)
    if base in FULL and not scalar:
        return 1
    return 2


def main():
    path, a, b = sys.argv[1], sys.argv[2], sys.argv[3]
    lines = open(path).read().split("\n")

    def find(x, start=0):
        if x.isdigit():
            return int(x) - 1
        for n in range(start, len(lines)):
            if lines[n].startswith(x + ":"):
                return n
        raise SystemExit(f"label {x} not found")
    i0 = find(a)
    i1 = find(b, i0)
    tot = {"valu": 0, "weighted": 0, "salu": 0, "lds": 0, "vmem": 0}
    kinds = {}
    for l in lines[i0:i1 + 1]:
        s = l.strip()
        if not s or s.startswith((";", ".", "//")) or s.endswith(":"):
            continue
        op = s.split()[0]
        w = weight(s)
        if w is not None:
            tot["valu"] += 1
            tot["weighted"] += w
            kinds[(op, w)] = kinds.get((op, w), 0) + 1
        elif op.startswith("s_"):
            tot["salu"] += 1
        elif op.startswith("ds_"):
            tot["lds"] += 1
        elif op.startswith(("global_", "buffer_", "flat_", "scratch_")):
            tot["vmem"] += 1
    print(tot)
    for (op, w), c in sorted(kinds.items(), key=lambda kv: -kv[1] * kv[0][1]):
        print(f"  {op:28s} x{c:3d}  weight {w}")


if __name__ == "__main__":
    main()
