#!/usr/bin/env python3
"""Writes the measured numbers of DESIGN.md section 8 from the files under profiles/<tag>/ (what
tools/refresh_profiles.sh left there), between the markers `<!-- measured:begin -->` / `<!-- measured:end -->`: the
table and the figures quoted around it come from ONE source, a refresh is `refresh_profiles.sh`, this script, a commit.
usage: python tools/design_numbers.py [tag]"""
import csv
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r05"
prev_tag = sys.argv[2] if len(sys.argv) > 2 else "r%02d" % (int(tag[1:]) - 1)
P = os.path.join(ROOT, "profiles", tag)

def read_table(path):
    out = {}
    for line in open(path):
        m = re.match(r"(.+?)\s+ms/step ([0-9.]+)\s+value ([0-9.e+]+)\s+k4_ms ([0-9.]+)(?:\s+creep k4_ms ([0-9.]+))?", line)
        if m:
            out[m.group(1).strip()] = (float(m.group(2)), float(m.group(3)), float(m.group(4)))
    return out


table = read_table(os.path.join(P, "bench_table.txt"))
prev = read_table(os.path.join(ROOT, "profiles", prev_tag, "bench_table.txt"))


def was(*names):
    """the previous round's figures for the same rows (its own box)"""
    parts = []
    for name in names:
        key = name if name in prev else next((k for k in prev if k.startswith(name)), None)
        if key is None:
            parts.append("-")
        else:
            ms, val, k4 = prev[key]
            parts.append("%.4g ms / %s / %.4g us" % (ms, ("%.3g" % val).replace("e+09", "e9").replace("e+10", "e10"), k4 * 1e3))
    return "; ".join(parts)

d = json.loads(open(os.path.join(P, "bench_default.json")).read().strip().splitlines()[-1])
tr = json.loads(open(os.path.join(P, "bench_under_trace.json")).read().strip().splitlines()[-1])
traffic = json.load(open(os.path.join(P, "k4_traffic.json")))
stats = {}
for row in csv.DictReader(open(os.path.join(P, "kernel_stats.csv"))):
    for key in ("k_step_tiled", "k_scatter_and_bands", "k_scan_onepass", "k_scan_totals", "k_scan_apply"):
        if key in row["Name"]:
            stats[key] = float(row["AverageNs"]) / 1e3
keep = {}
for line in open(os.path.join(P, "kept_windows_in_kernel_builder.txt")):
    m = re.match(r"(\w+) agents (\d+) shadow (\d) slack (\d+): ms/step ([0-9.]+)", line)
    if m:
        keep[(m.group(1), int(m.group(2)), int(m.group(3)), int(m.group(4)))] = float(m.group(5)) * 1e3
small = {}
scene = None
for line in open(os.path.join(P, "trace_small_125k.txt")):
    if line.startswith("== "):
        scene = line[3:].strip()
    m = re.match(r"(\S+).*median\s+([0-9.]+)", line)
    if m and scene:
        small[(scene, m.group(1)[:14])] = float(m.group(2))
scenes = {v["workload"]: v for v in json.load(open(os.path.join(P, "scene_stats.json"))).values()}


def row(name):
    ms, val, k4 = table[name]
    return ms, val, k4


def g(v):  # 6.66e9
    return ("%.3g" % v).replace("e+09", "e9").replace("e+10", "e10").replace("e+06", "e6").replace("e+05", "e5")


R = d["roofline"]
L = []
L.append("| workload (1M agents, dt 0.05, Zanlungo(1,1,0,0.4,2,0.2), eyesight 2.0 m, cell 2.0 m unless said) | ms/step | agent-steps/s | `k_step_tiled` | round 4 (`profiles/" + prev_tag + "/`, its own box) |")
L.append("|---|---|---|---|---|")
L.append(f"| **default: the uniform 2.5 /m^2 crowd walking at 1.3 m/s** (`--workload walk`, §5; `profiles/{tag}/bench_default.json`) | {d['ms_per_step']:.4f} | **{g(d['value'])}** | {R['kernel_ms'] * 1e3:.1f} us | {was('default')} |")
ms, val, k4 = row("creep")
L.append(f"| the same crowd standing, creeping counter-flow only, non-zero forces (`--workload creep`; `value_full_force` of the default line is this scene, timed in the same run: {g(d['value_full_force'])}) | {ms:.4f} | {g(val)} | {k4 * 1e3:.1f} us | {was('creep')} |")
ms, val, k4 = row("random")
L.append(f"| same density, randomly thinned lattice: neighbour counts scatter like a real crowd's, forces of every size (`--workload random`; `scattered_scene` of the default line: {g(d['scattered_scene']['value'])}) | {ms:.4f} | {g(val)} | {k4 * 1e3:.1f} us | {was('random')} |")
a, b = row("walk_4M"), row("walk_16M")
L.append(f"| the walking crowd with 4M / 16M agents on the one GPU | {a[0]:.3f} / {b[0]:.2f} | {g(a[1])} / {g(b[1])} | {a[2]:.3f} / {b[2]:.2f} ms | {was('walk_4M', 'walk_16M')} |")
a, b = row("walk_125k"), row("walk_100k")
L.append(f"| **... with 125k / 100k agents** (configs[2]'s share of one of 8 GPUs / configs[1]; kept windows, one-launch scan: §4) | **{a[0]:.4f} / {b[0]:.4f}** | **{g(a[1])} / {g(b[1])}** | {a[2] * 1e3:.1f} / {b[2] * 1e3:.1f} us (with the builder workgroups in the launch) | {was('walk_125k', 'walk_100k')} |")
a = row("e1c1")
L.append(f"| eyesight 1.0 m, cell 1.0 m (K ~ 8) | {a[0]:.4f} | {g(a[1])} | {a[2] * 1e3:.1f} us | {was('e1c1')} |")
a, b = row("c1e2"), row("c4e2")
L.append(f"| other cell sizes at eyesight 2.0 m: cell 1.0 m (5 x 5 cells) / cell 4.0 m (40 agents per cell) | {a[0]:.3f} / {b[0]:.3f} | {g(a[1])} / {g(b[1])} | {a[2] * 1e3:.1f} / {b[2] * 1e3:.1f} us | {was('c1e2', 'c4e2')} |")
a = row("hotspots_1M")
L.append(f"| half background, half Gaussian hotspots (`--workload hotspots`, up to 4.9 agents/m^2, `CS_CFG_DENSE`) | {a[0]:.3f} | {g(a[1])} | {a[2] * 1e3:.0f} us | {was('hotspots_1M')} |")
a = row("hotspots_4M")
L.append(f"| BASELINE configs[4] itself: 4M agents, half in hotspots, one GPU | {a[0]:.3f} | {g(a[1])} | {a[2]:.2f} ms | {was('hotspots_4M')} |")
a = row("stream")
L.append(f"| 850k agents sustained by 25,000 source-sinks (`--workload stream`, BASELINE configs[3]): spawn kernel + sink test + compaction every step, fire-and-forget (host sync every 16 steps) | {a[0]:.3f} | {g(a[1])} | {a[2] * 1e3:.1f} us | {was('stream')} |")
a = row("stream_route")
L.append(f"| the same stream with device route followers (`--planner route`) | {a[0]:.3f} | {g(a[1])} | {a[2] * 1e3:.1f} us | {was('stream_route')} |")
a = row("readback")
L.append(f"| a frame of all agents to pinned host memory every step (`--readback`) | {a[0]:.2f} | {g(a[1])} ({a[1] * 32 / 1e9:.0f} GB/s over PCIe) | — | {was('readback')} |")
L.append(f"| CPU baseline (`cpu_baseline`): the reference-shaped oracle port, f64, 1 thread, default scene, {d['cpu_baseline']['sample'].split(' of ')[0]} | — | {g(d['cpu_baseline']['value'])} | | at 100k agents: 3.5e5 |")
L.append(f"| CPU, the same arithmetic on cell-sorted arrays with OpenMP (16 threads, 1M agents; not the reference's shape) | — | {g(d['cpu_baseline_openmp']['value'])} | | |")
L.append("")
scan = stats.get("k_scan_onepass")
L.append(f"Per step on one GPU (default, `profiles/{tag}/kernel_stats.csv`, tracer attached): K4 {stats['k_step_tiled']:.1f} us "
         f"({tr['roofline']['kernel_ms'] * 1e3:.1f} by the bench's own events in that run, {R['kernel_ms'] * 1e3:.1f} untraced), the scan (one launch, ticketed tiles since round 5) "
         f"{scan:.1f} us, scatter + band builder (one launch) {stats['k_scatter_and_bands']:.1f} us traced (13-14 untraced).  K4 per launch "
         f"(`profiles/{tag}/pmc*_summary.txt`): {traffic['valu_wave_insts_per_launch']:.3g} VALU, {traffic['salu_wave_insts_per_launch']:.3g} SALU, "
         f"{traffic['lds_wave_insts_per_launch']:.3g} LDS wave-instructions, FETCH_SIZE {traffic['FETCH_SIZE_KiB_per_launch'] / 1024:.1f} MiB (x 2 on gfx950) + WRITE_SIZE "
         f"{traffic['WRITE_SIZE_KiB_per_launch'] / 1024:.1f} MiB = {traffic['traffic_bytes_per_launch'] / 1e6:.1f} MB against {R['algorithmic_bytes_per_launch'] / 1e6:.1f} MB algorithmic "
         f"(`roofline.traffic`); `roofline.frac` {R['frac']:.3f}, `hbm_read_frac` {R['hbm_read_frac']:.3f}, `valu_issue_frac` {R['valu_issue_frac']:.2f} of the measured "
         f"full-rate ceiling.")
w, c, r = scenes["walk"], scenes["creep"], scenes["random"]
L.append(f"Scene statistics from the kernel's trip counters (`profiles/{tag}/scene_stats.json`, quoted by the bench line): walk / creep "
         f"{w['mean_neighbours']:.1f} neighbours in sight, {100 * w['finite_ttc_frac']:.0f} % / {100 * c['finite_ttc_frac']:.0f} % of the agents with a finite time to collision, "
         f"{w['mean_forward_neighbours']:.1f} neighbours with right of way, {w['filter_trips_per_wave']:.1f} filter / {w['ttc_trips_per_wave']:.1f} time-to-collision / "
         f"{w['force_trips_per_wave']:.1f} force trips per wave, force lane use {w['force_lane_use']:.2f}; random: {r['mean_neighbours']:.1f}, {100 * r['finite_ttc_frac']:.0f} %, "
         f"{r['mean_forward_neighbours']:.1f}, {r['filter_trips_per_wave']:.1f} / {r['ttc_trips_per_wave']:.1f} / {r['force_trips_per_wave']:.1f}, lane use {r['force_lane_use']:.2f}, "
         f"{100 * r['waves_beyond_lds_rows_frac']:.0f} % of the waves beyond their LDS rows.")
a = row("walk_125k")
k = lambda s, n, sh: keep[(s, n, sh, 10 if sh else 0)]  # noqa: E731
L.append(f"The 125k-agent step (`bench.py --agents 125000`, walking crowd): **{a[0] * 1e3:.1f} us = {g(a[1])} agent-steps/s** (round 4: 45.4, round 3: 52.0, round 2: 56.6); "
         f"`tools/keep_bench.sh` (`profiles/{tag}/kept_windows_in_kernel_builder.txt`, shorter runs; kept windows against the builder in the scatter launch): "
         f"{k('walk', 125000, 1):.1f} against {k('walk', 125000, 0):.1f} us, {k('walk', 62500, 1):.1f} against {k('walk', 62500, 0):.1f} at 62.5k agents, "
         f"{k('random', 125000, 1):.1f} against {k('random', 125000, 0):.1f} (random), {k('creep', 125000, 1):.1f} against {k('creep', 125000, 0):.1f} (creep).  "
         f"The kernels of such a step (`profiles/{tag}/trace_small_125k.txt`, medians): scan {small[('walk', 'k_scan_onepass')]:.1f} (one launch), "
         f"scatter {small[('walk', 'k_scatter_and_')]:.1f} , neighbour pass {small[('walk', 'void')]:.1f} us, and nothing else.")
block = "\n".join(L)
path = os.path.join(ROOT, "DESIGN.md")
s = open(path).read()
a, b = "<!-- measured:begin -->", "<!-- measured:end -->"
assert a in s and b in s, "markers missing in DESIGN.md"
s = s[:s.index(a) + len(a)] + "\n" + block + "\n" + s[s.index(b):]
open(path, "w").write(s)
print("DESIGN.md section 8 rewritten from", P)
