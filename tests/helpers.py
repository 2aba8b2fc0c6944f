"""Shared builders for the parity tests: identical oracle / GPU sims from one seeded parameter set."""
import copy

import numpy as np

from oracle import OracleSim, default_curriculum
from wbc_amd import abi

STATE_TENSORS = ["ROOT_STATES", "DOF_STATE", "TORQUES", "OBS_HISTORY", "ACTION_HISTORY", "ACTIONS", "LAST_ACTIONS",
                 "LAST_DOF_VEL", "LAST_ROOT_VEL", "COMMANDS", "GOAL_STATE", "EPISODE_LENGTH", "EPISODE_SUMS", "METRIC_SUMS",
                 "FORCE_SENSOR", "NET_CONTACT_FORCE", "RIGID_BODY_STATE", "BASE_LIN_VEL", "BASE_ANG_VEL", "TIME_OUT_BUF",
                 "RESET_BUF", "BOX_SLEEP_TIMER", "FEET_AIR_TIME", "LAST_CONTACTS"]


def random_quat(rng):
    """A uniformly random unit quaternion (xyzw)."""
    q = rng.normal(size=4)
    return q / np.linalg.norm(q)


def random_env_params(n, seed=0):
    rng = np.random.default_rng(seed)
    tt = rng.uniform(1, 3, n) / 0.02
    return dict(
        friction=rng.uniform(-0.5, 3.0, n).astype(np.float32),
        base_dmass=rng.uniform(-0.5, 2.5, n).astype(np.float32),
        base_dcom=rng.uniform(-0.15, 0.15, (n, 3)).astype(np.float32),
        gripper_dmass=rng.uniform(0, 0.1, n).astype(np.float32),
        motor_strength=rng.uniform(0.7, 1.3, (n, 18)).astype(np.float32),
        env_origins=np.stack([rng.uniform(-3.75, -3, n), rng.uniform(-115, 115, n), np.zeros(n)], 1).astype(np.float32),
        box_delta_y=rng.uniform(0.1, 0.3, n).astype(np.float32),
        traj_timesteps=tt.astype(np.float32),
        traj_total_timesteps=(tt + rng.uniform(0.5, 2, n) / 0.02).astype(np.float32))


def make_oracle(robot, n, params, precision="f64", seed=1, tcfg=None):
    tc = tcfg if tcfg is not None else robot["tcfg"]
    o = OracleSim(robot["wmodel"], tc, n, seed=seed, precision=precision)
    o.set_curriculum(default_curriculum(robot["cfg"]))
    o.set_env_params(robot_model=robot["model"], **params)
    return o


def make_gpu(robot, n, params, seed=1, tcfg=None):
    import torch
    from wbc_amd.sim import WbcSim
    tc = tcfg if tcfg is not None else robot["tcfg"]
    g = WbcSim(robot["wmodel"], tc, n, torch.device("cuda:0"), seed=seed)
    g.set_curriculum(default_curriculum(robot["cfg"]))
    g.set_env_params(**params)
    return g


def sync_oracle_from_gpu(o, g, names=STATE_TENSORS):
    import torch
    torch.cuda.synchronize()
    for name in names:
        o.set(name, g.tensor(name).detach().cpu().numpy().astype(np.float64))
    o.step_counter = g.step_counter


def random_standing_state(n, tcfg, rng, height=(0.30, 0.45)):
    """Poses around the nominal stance, some airborne, some in ground contact."""
    root = np.zeros((n, 2, 13), dtype=np.float32)
    root[:, 0, 0:2] = rng.uniform(-1, 1, (n, 2))
    root[:, 0, 2] = rng.uniform(height[0], height[1], n)
    ang = rng.uniform(-0.15, 0.15, (n, 3))
    q = np.zeros((n, 4))
    # small-angle quaternion from roll/pitch/yaw
    cr, sr, cp, sp, cy, sy = [f(ang[:, i] / 2) for i in range(3) for f in (np.cos, np.sin)]
    q[:, 3] = cr * cp * cy + sr * sp * sy
    q[:, 0] = sr * cp * cy - cr * sp * sy
    q[:, 1] = cr * sp * cy + sr * cp * sy
    q[:, 2] = cr * cp * sy - sr * sp * cy
    root[:, 0, 3:7] = q
    root[:, 0, 7:10] = rng.uniform(-0.5, 0.5, (n, 3))
    root[:, 0, 10:13] = rng.uniform(-1, 1, (n, 3))
    # the free box actor: near the robot, some resting on / dipping into the ground plane, some in the air, a few spinning
    root[:, 1, 6] = 1
    root[:, 1, 0:2] = root[:, 0, 0:2] + rng.uniform(0.3, 0.8, (n, 2)) * rng.choice([-1.0, 1.0], (n, 2))
    root[:, 1, 2] = rng.uniform(0.045, 0.15, n)
    spin = rng.random(n) < 0.25
    root[spin, 1, 7:10] = rng.uniform(-1, 1, (int(spin.sum()), 3))
    root[spin, 1, 10:13] = rng.uniform(-5, 5, (int(spin.sum()), 3))
    dof = np.zeros((n, 20, 2), dtype=np.float32)
    dof[:, :, 0] = np.array(tcfg.default_dof_pos)[None] + rng.uniform(-0.25, 0.25, (n, 20))
    dof[:, :, 1] = rng.uniform(-2, 2, (n, 20))
    dof[:, 18:, :] = 0
    return root, dof


class GpuAsOracle:
    """The HIP sim (WbcSim, through the C-ABI) behind the OracleSim interface the closed-form physics cases use
    (tests/physics_cases.py): get / set by tensor name, step, simulate, set_heightfield, set_curriculum."""

    def __init__(self, wmodel, tcfg, n, seed=1):
        import torch
        from wbc_amd.sim import WbcSim
        self.torch = torch
        self.g = WbcSim(wmodel, tcfg, n, torch.device("cuda:0"), seed=seed)
        self.n, self.model = n, wmodel

    def get(self, name):
        self.torch.cuda.synchronize()
        return self.g.tensor(name).detach().cpu().numpy().astype(np.float64)

    def set(self, name, value):
        t = self.g.tensor(name)
        v = np.array(np.broadcast_to(np.asarray(value, dtype=np.float64), tuple(t.shape)))
        t.copy_(self.torch.from_numpy(v).to(t.dtype))

    def set_curriculum(self, cur):
        self.g.set_curriculum(cur)

    def set_heightfield(self, heights, hscale=0.0, vscale=0.0, tx=0.0, ty=0.0, tz=0.0):
        self.g.set_heightfield(heights, hscale, vscale, tx, ty, tz)

    def step(self, actions):
        a = self.torch.from_numpy(np.ascontiguousarray(actions, dtype=np.float32).reshape(self.n, 18)).cuda()
        self.g.step(a)

    def simulate(self):
        self.g.simulate()

    def refresh_rigid_body_state(self):
        self.g.refresh_rigid_body_state()

    def close(self):
        self.g.close()
