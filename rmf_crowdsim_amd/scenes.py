"""Seeded synthetic scenes for parity tests and bench.py.

All randomness comes from splitmix64(seed, index), so a scene is a pure function of its
arguments on any host.
"""
import math

import numpy as np

MASK = (1 << 64) - 1


def _splitmix64(x):
    x = (x + np.uint64(0x9E3779B97F4A7C15)) & np.uint64(MASK)
    x = ((x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & np.uint64(MASK)
    x = ((x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & np.uint64(MASK)
    return x ^ (x >> np.uint64(31))


def uniform01(seed, index):
    """Counter-based uniform in [0, 1) for integer `index` (array)."""
    with np.errstate(over="ignore"):
        r = _splitmix64(np.asarray(index, dtype=np.uint64) ^ _splitmix64(np.uint64(seed)))
    return (r >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def jittered_lattice(n, spacing, origin, jitter_frac, seed, columns=None):
    """First n sites (row-major) of a square lattice, each jittered by +-jitter_frac*spacing."""
    side = columns or int(math.ceil(math.sqrt(n)))
    k = np.arange(n, dtype=np.uint64)
    ix = (k % np.uint64(side)).astype(np.float64)
    iy = (k // np.uint64(side)).astype(np.float64)
    jx = (uniform01(seed, 2 * k) * 2.0 - 1.0) * jitter_frac * spacing
    jy = (uniform01(seed, 2 * k + np.uint64(1)) * 2.0 - 1.0) * jitter_frac * spacing
    return np.stack([origin[0] + (ix + 0.5) * spacing + jx,
                     origin[1] + (iy + 0.5) * spacing + jy], axis=1)


# ---- config 1: the visualiser's parameters, 256 agents ------------------------------
VIZ_GRID = dict(width=1000.0, height=1000.0, cell_size=20.0, offset=(-500.0, -500.0))  # main.rs:65
VIZ_ZANLUNGO = (1.0, 1.0, 0.0, 40.0, 2.0, 20.0)                                        # main.rs:78-80
VIZ_EYESIGHT = 100.0                                                                   # main.rs:86
VIZ_SPEED = (0.0, 10.0)                                                                # main.rs:76


def viz_scene(n=256, seed=1, spacing=30.0):
    """Counter-flow of n agents with the visualiser's planner parameters.

    Ids are assigned in array order; even ids (which the id-parity planner sends towards
    -y, main.rs:26-29) start in the upper half plane and odd ids in the lower one, so
    the two streams meet and nobody reaches a grid edge within 1000 steps of 0.05 s.
    """
    half = n // 2
    cols = int(math.ceil(math.sqrt(half)))
    pts = np.zeros((n, 2))
    up = jittered_lattice(half, spacing, (-cols * spacing / 2.0, 0.0), 0.2, seed, columns=cols)
    dn = jittered_lattice(n - half, spacing, (-cols * spacing / 2.0 + spacing / 2.0, 0.0), 0.2,
                          seed + 1, columns=cols)
    pts[0::2] = up
    pts[1::2, 0] = dn[:, 0]
    pts[1::2, 1] = -dn[:, 1]
    return pts


# ---- the literal visualiser scene --------------------------------------------------
VIZ3_POSITIONS = [(100.0, 100.0), (100.0, -100.0), (60.0, 100.0)]  # main.rs:70-74


# ---- configs 2-3: metric-scale uniform crowd ---------------------------------------
# The reference supplies no metric-scale parameters; these are the visualiser's, scaled so
# that agent_radius = 0.2 m (R 20 -> 0.2, D 40 -> 0.4, eyesight 100 -> 1.0 ... 2.0 m).
METRIC_ZANLUNGO = (1.0, 1.0, 0.0, 0.4, 2.0, 0.2)
METRIC_DENSITY = 2.5  # agents / m^2: 100k agents on 200 m x 200 m (BASELINE.json configs[1])
WALK_SPEED = 1.3      # m/s
# Long runs use a creeping counter-flow: the reference's Zanlungo force grows like 1/t_i and
# a 2.5 agents/m^2 crowd at walking speed drives t_i -> 0 within a few steps (forces clamp
# at 1e15, agents leave the grid and `step` returns "Index out of bounds": the f64 oracle
# shows this too, DESIGN.md "Scenes").  At 1 mm/s nobody can close the lattice gap in 1000
# steps, yet every agent has finite t_i and non-zero forces, so every branch of the kernel
# runs at its steady-state rate.
CREEP_SPEED = 0.001


def uniform_crowd(n, seed=7, density=METRIC_DENSITY, cell_size=2.0, margin=10.0, room=0.0):
    """n agents on a jittered lattice at `density`, in a square grid with `margin` metres of free
    cells around the population and `room` more metres on the high sides (for a crowd that
    walks).  Returns (positions, grid kwargs, extent, group) where group[k] in {0, 1} is a
    checkerboard over lattice sites."""
    spacing = 1.0 / math.sqrt(density)
    side = int(math.ceil(math.sqrt(n)))
    extent = side * spacing
    cells = int(math.ceil((extent + 2 * margin + room) / cell_size))
    width = cells * cell_size
    pts = jittered_lattice(n, spacing, (margin, margin), 0.2, seed, columns=side)
    k = np.arange(n)
    group = ((k % side) + (k // side)) % 2
    grid = dict(width=width, height=width, cell_size=cell_size, offset=(0.0, 0.0))
    return pts, grid, extent, group


def hotspot_crowd(n, seed=7, cell_size=2.0, margin=10.0, sigma=5.0, per_hotspot=800, site=0.45,
                  hot_fraction=0.5):
    """BASELINE.json configs[4]: half of the n agents as a uniform
    background of METRIC_DENSITY / 2, half in Gaussian hotspots (sigma 5 m, `per_hotspot` agents
    each).  Sites of a fine jittered lattice (spacing `site` > agent radius, so nobody overlaps)
    are kept with probability density(x) * site^2; exactly the n sites with the smallest
    u / p survive.  The lattice caps the density at 1 / site^2 = 4.9 agents/m^2: the reference's
    Zanlungo model leaves the grid within ten steps when neighbours stand 0.32 m apart, even at
    creeping speed (the f64 oracle does too).  Returns (positions, grid kwargs, extent, group)
    like uniform_crowd."""
    bg_density = METRIC_DENSITY * (1.0 - hot_fraction)
    extent = math.sqrt(n / METRIC_DENSITY)
    side = int(math.ceil(extent / site))
    n_hot = max(1, int(round(n * hot_fraction / per_hotspot))) if hot_fraction > 0.0 else 0
    rho = np.full((side, side), bg_density, dtype=np.float32)  # [iy, ix]
    hk = np.arange(n_hot, dtype=np.uint64)
    hx = (0.1 + 0.8 * uniform01(seed + 101, 2 * hk)) * extent
    hy = (0.1 + 0.8 * uniform01(seed + 101, 2 * hk + np.uint64(1))) * extent
    reach = int(math.ceil(4.0 * sigma / site))
    peak = (n * hot_fraction / max(n_hot, 1)) / (2.0 * math.pi * sigma * sigma)
    for cx, cy in zip(hx, hy):
        ix0, iy0 = int(cx / site), int(cy / site)
        xs = np.arange(max(ix0 - reach, 0), min(ix0 + reach + 1, side))
        ys = np.arange(max(iy0 - reach, 0), min(iy0 + reach + 1, side))
        gx = np.exp(-(((xs + 0.5) * site - cx) ** 2) / (2.0 * sigma * sigma))
        gy = np.exp(-(((ys + 0.5) * site - cy) ** 2) / (2.0 * sigma * sigma))
        rho[ys[0]:ys[-1] + 1, xs[0]:xs[-1] + 1] += (peak * np.outer(gy, gx)).astype(np.float32)
    p = np.minimum(rho.reshape(-1).astype(np.float64) * site * site, 1.0)
    k = np.arange(side * side, dtype=np.uint64)
    with np.errstate(divide="ignore"):
        key = uniform01(seed + 202, k) / p
    keep = np.argpartition(key, n - 1)[:n]
    keep.sort()
    kk = keep.astype(np.uint64)
    ix = (kk % np.uint64(side)).astype(np.float64)
    iy = (kk // np.uint64(side)).astype(np.float64)
    jx = (uniform01(seed, 2 * kk) * 2.0 - 1.0) * 0.15 * site
    jy = (uniform01(seed, 2 * kk + np.uint64(1)) * 2.0 - 1.0) * 0.15 * site
    pts = np.stack([margin + (ix + 0.5) * site + jx, margin + (iy + 0.5) * site + jy], axis=1)
    cells = int(math.ceil((extent + 2 * margin) / cell_size))
    grid = dict(width=cells * cell_size, height=cells * cell_size, cell_size=cell_size, offset=(0.0, 0.0))
    group = ((keep % side) + (keep // side)) % 2
    return pts, grid, extent, group


def random_crowd(n, seed=7, cell_size=2.0, margin=10.0, site=0.45):
    """METRIC_DENSITY agents/m^2 without the regularity of uniform_crowd: every site of a fine
    lattice is kept with the same probability (about one in two), so the number of neighbours in
    sight scatters like in a real crowd (sigma ~ 4 around 31 at eyesight 2 m) instead of sitting
    at the lattice value."""
    return hotspot_crowd(n, seed=seed, cell_size=cell_size, margin=margin, site=site, hot_fraction=0.0)


def add_counterflow(sim, pts, group, speed, local_planner, eyesight, axis=1):
    """Two interleaved streams: checkerboard group 0 walks +axis, group 1 walks -axis.
    Group 0 is added first, so its agents get the smaller ids (and yield, zanlungo.rs:173-198).
    Returns the agent ids in the order of `pts`."""
    from .simulation import StubHighLevelPlan
    v = [0.0, 0.0]
    v[axis] = speed
    ids = np.zeros(len(pts), dtype=np.int64)
    ids[group == 0] = sim.add_agents(pts[group == 0], StubHighLevelPlan(tuple(v)), local_planner,
                                     eyesight)
    v[axis] = -speed
    ids[group == 1] = sim.add_agents(pts[group == 1], StubHighLevelPlan(tuple(v)), local_planner,
                                     eyesight)
    return ids


def add_walking_crowd(sim, pts, group, local_planner, eyesight, walk=WALK_SPEED, creep=CREEP_SPEED):
    """The creeping counter-flow carried along at walking speed: every agent walks +x at `walk`
    (1.3 m/s: 6.5 cm per step of 0.05 s, so ~3 % of the agents change cell every step and the
    re-binning, the histogram and the scatter do real work), group 0 drifts +y and group 1 -y at
    `creep` on top.  Relative velocities, hence times to collision and neighbour lists, are those of
    the creeping scene.  The reference's force term looks |v_i| * t_i ahead (zanlungo.rs:109-111
    with the neighbour's velocity blended to zero), hundreds of metres here, so exp(-(dist - 2R)/D)
    underflows and v = v_pref exactly, in f64 as in f32: a crowd that walks for as long as the grid
    lasts.  (A head-on counter-flow AT walking speed does not exist in this model for any
    agent_scale: walkers pass through each other, t_i = 0, 0/0)
    Returns the agent ids in the order of `pts`."""
    from .simulation import StubHighLevelPlan
    ids = np.zeros(len(pts), dtype=np.int64)
    ids[group == 0] = sim.add_agents(pts[group == 0], StubHighLevelPlan((walk, creep)), local_planner, eyesight)
    ids[group == 1] = sim.add_agents(pts[group == 1], StubHighLevelPlan((walk, -creep)), local_planner, eyesight)
    return ids


# ---- config 4: a crowd fed by source-sinks --------------------------------------------
def stream_lanes(n_agents, lane_length=16.0, lane_gap=1.0, release_gap=0.4, cell_size=2.0, margin=10.0):
    """Source-sink lanes whose steady state holds ~n_agents: parallel lanes `lane_gap` apart,
    alternating direction, each `lane_length` long (an agent is released whenever the previous
    one is `release_gap` = the reference's hard-coded 0.4 m away, lib.rs:212-217).
    Returns (list of (source, waypoint, velocity), grid kwargs, steps to fill at 1.3 m/s, dt 0.05)."""
    per_lane = lane_length / release_gap
    n_lanes = int(math.ceil(n_agents / per_lane))
    cols = int(math.ceil(math.sqrt(n_lanes * lane_length / lane_gap) / lane_length * lane_length / lane_gap))
    cols = max(1, int(math.ceil(math.sqrt(n_lanes * lane_length * lane_gap) / lane_gap)))
    rows = int(math.ceil(n_lanes / cols))
    width = cols * lane_gap + 2 * margin
    height = rows * (lane_length + 2.0) + 2 * margin
    side = int(math.ceil(max(width, height) / cell_size)) * cell_size
    lanes = []
    for k in range(n_lanes):
        c, r = k % cols, k // cols
        x = margin + (c + 0.5) * lane_gap
        y0 = margin + r * (lane_length + 2.0) + 1.0
        if c % 2 == 0:
            lanes.append(((x, y0), (x, y0 + lane_length), (0.0, WALK_SPEED)))
        else:
            lanes.append(((x, y0 + lane_length), (x, y0), (0.0, -WALK_SPEED)))
    grid = dict(width=side, height=side, cell_size=cell_size, offset=(0.0, 0.0))
    fill_steps = int(lane_length / (WALK_SPEED * 0.05)) + 20
    return lanes, grid, fill_steps
