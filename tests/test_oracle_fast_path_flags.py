"""CPU checks of the two test-only switches of the oracle's cell-sorted path (oracle_fast_steps_ex): they are what the
f32 leg of the long three-way GPU comparisons stands on (tests/test_gpu_north_star.py), so they are pinned against the
plain f64 path here, without a GPU."""
import numpy as np

from oracle_sim import fast_steps
from rmf_crowdsim_amd import scenes

CROSSING_ZANLUNGO = (0.3, 1.0, 0.0, 0.4, 2.0, 0.2)


def _crossing(n, seed=5):
    spacing = 1.0 / np.sqrt(0.3)
    side = int(np.ceil(np.sqrt(n)))
    pts = scenes.jittered_lattice(n, spacing, (40.0, 40.0), 0.25, seed)
    k = np.arange(n)
    group = ((k % side) + (k // side)) % 2
    th = np.radians(30.0)
    v = scenes.WALK_SPEED
    pref = np.where(group[:, None] == 0, np.array([v, 0.0]), np.array([v * np.cos(th), v * np.sin(th)]))
    size = float(np.ceil(side * spacing + 140.0))
    return pts, pref, dict(width=size, height=size, cell_size=2.0, offset=(0.0, 0.0)), side * spacing


def test_cell_relative_positions_change_nothing_in_f64_and_the_guard_changes_nothing_at_all():
    """Flag 2 (positions kept in f64, every update computed relative to the agent's own cell) is the reference's
    arithmetic up to the rounding of a translation: in the f64 build it follows the plain path to 1e-10 of the extent over
    300 steps of the crossing flows (forces of walking magnitude).  Flag 1 (pairs with |rel_vel|^2 < 1e-30 outside the
    collision distance never collide) does not change a bit of a run the f64 path's own flaw does not strike."""
    n, steps = 6000, 300
    pts, pref, grid, extent = _crossing(n)
    struck = np.zeros(n, dtype=np.uint8)
    plain, v_plain, sec = fast_steps(pts, pref, CROSSING_ZANLUNGO, 2.0, grid, 0.05, steps, threads=8, spurious=struck)
    assert sec > 0 and not struck.any() and np.isfinite(plain).all()
    force = np.hypot(*(v_plain - pref).T)
    guarded, v_guarded, sec = fast_steps(pts, pref, CROSSING_ZANLUNGO, 2.0, grid, 0.05, steps, threads=8, guarded=True)
    assert sec > 0 and guarded.tobytes() == plain.tobytes() and v_guarded.tobytes() == v_plain.tobytes()
    rel, v_rel, sec = fast_steps(pts, pref, CROSSING_ZANLUNGO, 2.0, grid, 0.05, steps, threads=8, cell_relative=True)
    d = np.hypot(*(rel - plain).T) / extent
    print(f"f64 per-cell vs plain f64, {steps} steps: max {d.max():.2e}; strongest force now {force.max():.2f} m/s")
    assert sec > 0 and d.max() <= 1e-10


def test_the_f32_leg_of_the_three_way_runs():
    """The f32 build with both flags: an independent 32-bit implementation in the engine's precision class.  On the
    crossing flows it stays within 1e-6 of the extent of the f64 path for 99 % of the agents over 300 steps (the rest are
    the model's discontinuities: dodges decided a step apart), where global f32 coordinates are 100 times further off."""
    n, steps = 6000, 300
    pts, pref, grid, extent = _crossing(n)
    f64, _, _ = fast_steps(pts, pref, CROSSING_ZANLUNGO, 2.0, grid, 0.05, steps, threads=8)
    leg, _, sec = fast_steps(pts, pref, CROSSING_ZANLUNGO, 2.0, grid, 0.05, steps, threads=8, kind="f32", guarded=True,
                             cell_relative=True)
    assert sec > 0 and np.isfinite(leg).all()
    d = np.hypot(*(leg - f64).T) / extent
    glob, _, sec_g = fast_steps(pts, pref, CROSSING_ZANLUNGO, 2.0, grid, 0.05, steps, threads=8, kind="f32", guarded=True)
    dg = np.hypot(*(glob - f64).T) / extent
    print(f"f32 per-cell vs f64: p99 {np.quantile(d, 0.99):.2e} max {d.max():.2e}; f32 global coordinates vs f64: "
          f"median {np.median(dg):.2e} (run {'completed' if sec_g > 0 else 'left the grid'})")
    assert np.quantile(d, 0.99) <= 1e-6 and np.median(d) <= 1e-7
    assert np.median(dg) > 20 * np.median(d)
