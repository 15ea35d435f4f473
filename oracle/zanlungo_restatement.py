"""TEST INFRASTRUCTURE ONLY -- a SECOND, independent restatement of the reference's Zanlungo planner.

Written from /root/reference/rmf_crowdsim/src/local_planners/zanlungo.rs alone (function by
function, same order, same names), NOT from oracle/crowdstep_oracle.cpp.  The reference holds no
test vector for compute_agent_force / right_of_way_vel / slerp / compute_tti /
get_desired_velocity ("parity unpinned" at that boundary), so the C++ oracle is
pinned there only by a reading of the Rust.  Two readings made separately, in two languages,
that agree on tens of thousands of random inputs (tests/test_zanlungo_restatement.py) make a
misreading much less likely; they do not replace running the reference, which this pipeline
cannot do (no rustc / cargo).

All arithmetic is IEEE binary64 through numpy scalars, so x/0 -> inf and 0/0 -> NaN as in Rust
(plain Python floats would raise).  Only tests/ may import this module.
"""
import numpy as np

f64 = np.float64
INF = f64(np.inf)


class Agent:
    """pub struct Agent, lib.rs:46-65: the fields the planner reads."""
    __slots__ = ("agent_id", "position", "velocity", "preferred_vel")

    def __init__(self, agent_id, position, velocity, preferred_vel):
        self.agent_id = int(agent_id)
        self.position = (f64(position[0]), f64(position[1]))
        self.velocity = (f64(velocity[0]), f64(velocity[1]))
        self.preferred_vel = (f64(preferred_vel[0]), f64(preferred_vel[1]))


# nalgebra Vector2<f64> pieces used by the planner
def _add(a, b): return (a[0] + b[0], a[1] + b[1])
def _sub(a, b): return (a[0] - b[0], a[1] - b[1])
def _scale(a, s): return (a[0] * s, a[1] * s)
def _neg(a): return (-a[0], -a[1])
def _dot(a, b): return a[0] * b[0] + a[1] * b[1]
def _norm_squared(a): return a[0] * a[0] + a[1] * a[1]
def _norm(a): return np.sqrt(_norm_squared(a))


def _normalize(a):
    n = _norm(a)
    return (a[0] / n, a[1] / n)


def slerp(t, p0, p1, sin_theta):
    """zanlungo.rs:23-28"""
    theta = np.arcsin(sin_theta)
    t0 = np.sin((f64(1) - t) * theta) / sin_theta
    t1 = np.sin(t * theta) / sin_theta
    return _add(_scale(p0, t0), _scale(p1, t1))


class Zanlungo:
    def __init__(self, agent_scale, obstacle_scale, reaction_time, force_distance, agent_mass, agent_radius):
        """zanlungo.rs:31-48"""
        self.agent_scale = f64(agent_scale)
        self.obstacle_scale = f64(obstacle_scale)
        self.reaction_time = f64(reaction_time)
        self.force_distance = f64(force_distance)
        self.agent_mass = f64(agent_mass)
        self.agent_radius = f64(agent_radius)
        self.agent_priorities = {}  # created empty, no setter in the reference

    def time_to_collision(self, rel_vel, rel_pos):
        """zanlungo.rs:49-74"""
        a = _norm_squared(rel_vel)
        b = f64(2) * _dot(rel_vel, rel_pos)
        c = _norm_squared(rel_pos) - self.agent_radius * self.agent_radius
        discriminant = b * b - f64(4) * a * c
        if discriminant < 0:
            return INF
        t0 = (-b - np.sqrt(discriminant)) / (f64(2) * a)
        t1 = (-b + np.sqrt(discriminant)) / (f64(2) * a)
        if (t0 < 0 and t1 > 0) or (t1 < 0 and t0 > 0):
            return f64(0)
        if t0 < t1 and t0 > 0:
            return t0
        elif t1 > 0:
            return t1
        else:
            return INF

    def compute_tti(self, current_agent, nearby_agents):
        """zanlungo.rs:76-91"""
        t_i = INF
        for n in nearby_agents:
            rel_vel = _sub(n.velocity, current_agent.velocity)
            rel_pos = _sub(n.position, current_agent.position)
            col_time = self.time_to_collision(rel_vel, rel_pos)
            if col_time < t_i:
                t_i = col_time
        return t_i

    def right_of_way_vel(self, agent_id, agent_vel, self_pref_vel, other_vel, other_pref_vel, other_priority):
        """zanlungo.rs:173-198"""
        def_priority = f64(agent_id)
        self_priority = self.agent_priorities.get(agent_id, def_priority)
        right_of_way = self_priority - other_priority
        # f64::clamp(-1, 1): NaN stays NaN
        if right_of_way < -1:
            right_of_way = f64(-1)
        if right_of_way > 1:
            right_of_way = f64(1)
        if right_of_way < 0:
            r_2 = np.sqrt(-right_of_way)
            other_adjusted_vel = _add(other_vel, _scale(_sub(other_pref_vel, other_vel), r_2))
            return -r_2, agent_vel, other_adjusted_vel
        elif right_of_way > 0:
            r_2 = np.sqrt(right_of_way)
            vel = _add(agent_vel, _scale(_sub(self_pref_vel, agent_vel), r_2))
            return r_2, vel, other_vel
        else:
            return f64(0), agent_vel, other_vel

    def compute_agent_force(self, agent, other_agent, t_i):
        """zanlungo.rs:93-170"""
        t_i = f64(t_i)
        def_priority = f64(other_agent.agent_id)
        other_priority = self.agent_priorities.get(other_agent.agent_id, def_priority)
        weight, my_vel, other_vel = self.right_of_way_vel(
            agent.agent_id, agent.velocity, agent.preferred_vel, other_agent.velocity,
            other_agent.preferred_vel, other_priority)
        weight = f64(1) - weight
        fut_pos = _add(agent.position, _scale(my_vel, t_i))
        other_future_pos = _add(other_agent.position, _scale(other_vel, t_i))
        d_ij = _sub(fut_pos, other_future_pos)
        dist = _norm(d_ij)
        if weight > 1:
            pref_speed = _norm(other_agent.preferred_vel)
            interpolate = True
            perp_dir = (f64(0), f64(0))
            if pref_speed < 0.0001:
                curr_rel_pos = _sub(agent.position, other_agent.position)
                perp_dir = (-curr_rel_pos[1], curr_rel_pos[0])
                if _dot(perp_dir, agent.velocity) < 0:
                    perp_dir = _neg(perp_dir)
            else:
                pref_dir = other_agent.preferred_vel
                if _dot(pref_dir, d_ij) > 0:
                    perp_dir = (-pref_dir[1], pref_dir[0])
                    if _dot(perp_dir, d_ij) < 0:
                        perp_dir = _neg(perp_dir)
                else:
                    interpolate = False
            if interpolate:
                sin_theta = perp_dir[0] * d_ij[1] - perp_dir[1] * d_ij[0]
                if sin_theta < 0:
                    sin_theta = -sin_theta
                if sin_theta > 1:
                    sin_theta = f64(1)
                d_ij = slerp(weight - f64(1), d_ij, perp_dir, sin_theta)
        if dist > _norm(_sub(fut_pos, other_future_pos)):
            return (f64(0), f64(0))
        d_ij_normalized = _normalize(d_ij)
        surface_dist = dist - self.agent_radius * f64(2)
        magnitude = weight * self.agent_scale * _norm(_sub(my_vel, other_vel)) / t_i
        if magnitude >= 1e15:
            magnitude = f64(1e15)
        return _scale(d_ij_normalized, magnitude * np.exp(-surface_dist / self.force_distance))

    def get_desired_velocity(self, agent, nearby_agents, recommended_velocity):
        """zanlungo.rs:201-217"""
        t_i = self.compute_tti(agent, nearby_agents)
        force = (f64(0), f64(0))
        if t_i != INF:
            for nearby_agent in nearby_agents:
                force = _add(force, self.compute_agent_force(agent, nearby_agent, t_i))
        rec = (f64(recommended_velocity[0]), f64(recommended_velocity[1]))
        return _add(rec, _scale(force, f64(1) / self.agent_mass))
