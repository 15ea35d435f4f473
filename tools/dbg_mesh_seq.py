import os, sys, math
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
from rmf_crowdsim_amd import (CrowdSimError, LocationHash2D, NoLocalPlan, SeededPoissonCrowd, Simulation, SourceSink,
                              StubHighLevelPlan, Zanlungo, scenes)
from rmf_crowdsim_amd.tiles import LocalTileMesh
seed = int(sys.argv[1]); split = sys.argv[2] == "1"
if split: os.environ["CS_TILE_SPLIT"] = "1"
rng = np.random.default_rng(15000 + seed)
cell = float(rng.choice([1.0, 2.0, 2.5])); side = float(rng.choice([40.0, 60.0, 80.0]))
grid = dict(width=side, height=side, cell_size=cell, offset=(float(rng.uniform(-5, 5)), float(rng.uniform(-5, 5))))
off = np.array(grid["offset"]); eyes = float(rng.choice([1.0, 2.0, 3.0])); halo = math.ceil(eyes / cell - 1e-9)
tiles = [(2, 2), (3, 1), (1, 3), (2, 3), (4, 2), (1, 2)][int(rng.integers(0, 6))]
ncell = int(side / cell)
if min(ncell // tiles[0], ncell // tiles[1]) < 2 * halo + 2: tiles = (2, 1)
w = rng.normal(side / 2, side / 6, size=(2000, 2)).clip(1, side - 1) + off if rng.random() < 0.5 else None
single = Simulation(LocationHash2D(**grid))
phases = int(rng.choice([1, 2]))
mesh = LocalTileMesh(LocationHash2D(**grid), tiles, halo_cells=halo, weights=w, phases=1 if split else phases)
kin = rng.random() < 0.5
lp = NoLocalPlan() if kin else Zanlungo(*scenes.METRIC_ZANLUNGO)
speed, dt = (1.0, 0.1) if kin else (0.002, 0.05)
print("cell", cell, "side", side, "eyes", eyes, "halo", halo, "tiles", tiles, "weighted", w is not None, "kin", kin, "cuts", mesh.layout.__dict__.get("cuts_x", None), flush=True)
lattice = scenes.jittered_lattice(4000, 0.6, (8.0 + off[0], 8.0 + off[1]), 0.15, seed)
lattice = lattice[(lattice[:, 0] < side - 8 + off[0]) & (lattice[:, 1] < side - 8 + off[1])]
used = 0
def check(tag):
    for e in [single] + mesh.engines: e.synchronize()
    a, b = single.read_agents(), mesh.read_agents()
    if len(a) != len(b) or a.tobytes() != b.tobytes():
        print("MISMATCH after", tag, "len", len(a), len(b), flush=True)
        if len(a) == len(b):
            bad = np.nonzero((a["x"] != b["x"]) | (a["y"] != b["y"]) | (a["vx"] != b["vx"]) | (a["vy"] != b["vy"]) | (a["id"] != b["id"]))[0]
            print("differing agents", len(bad), a[bad[:5]], b[bad[:5]])
            print("cells", ((a["x"][bad[:5]] - off[0]) / cell), ((a["y"][bad[:5]] - off[1]) / cell))
        else:
            sa, sb = set(a["id"].tolist()), set(b["id"].tolist())
            print("only single", sorted(sa - sb)[:10], "only mesh", sorted(sb - sa)[:10])
            miss = sorted(sa - sb)[:5]
            for m in miss:
                r = a[a["id"] == m][0]; print("missing", r, "cell", (r["x"] - off[0]) / cell, (r["y"] - off[1]) / cell)
        print("layout", [mesh.layout.rect(*mesh.layout.coords(i)) for i in range(mesh.layout.n_tiles)])
        sys.exit(1)
for k in range(60):
    op = rng.random()
    if op < 0.2:
        n = int(rng.choice([1, 10, 200]))
        if kin:
            pts = rng.uniform(8.0, side - 8.0, size=(n, 2)) + off
            if rng.random() < 0.3: pts[:, 0] = np.round((pts[:, 0] - off[0]) / cell) * cell + off[0] + rng.uniform(-0.05, 0.05, size=n)
        else:
            pts = lattice[used:used + n]; used += n
            if not len(pts): continue
        v = (float(rng.uniform(-1, 1)) * speed, float(rng.uniform(-1, 1)) * speed)
        single.add_agents(pts, StubHighLevelPlan(v), lp, eyes); mesh.add_agents(pts, StubHighLevelPlan(v), lp, eyes)
        check(f"op {k} add {n}")
    elif op < 0.3:
        if len(single):
            a = single.read_agents(); vic = int(a["id"][int(rng.integers(0, len(a)))])
            single.remove_agents(vic); mesh.remove_agents(vic)
            check(f"op {k} remove {vic}")
    elif op < 0.35 and kin:
        src = rng.uniform(8.0, side - 8.0, size=2) + off; dst = rng.uniform(8.0, side - 8.0, size=2) + off
        d = dst - src; v = d / max(np.linalg.norm(d), 1e-9) * speed
        for t in (single, mesh):
            t.add_source_sink(SourceSink(tuple(src), 0.8, SeededPoissonCrowd(4.0, 100 + k), StubHighLevelPlan(tuple(v)), lp, [tuple(dst)], False, eyes))
        print("op", k, "sink", src, dst, flush=True)
    else:
        rep = bool(rng.random() < 0.3)
        single.step(dt, report=rep); mesh.step(dt, report=rep)
        check(f"op {k} step report={rep}")
print("all equal")
