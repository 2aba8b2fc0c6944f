"""Does the whole loop learn the TASK? ("qualitative task parity: the robot stands / walks under a policy
trained here".) From scratch on the MI355X: 4096 envs, the shipped PPO hyper-parameters and non-RESUME schedules
(widowGo1_config.py:359,366, compressed 5x in time), the fused rollout / update / DAgger kernels, the full collision set.

The shipped reward table is a fine-tuning table (the reference ships RESUME = True, widowGo1_config.py:35). Three of its numbers
keep a policy trained from scratch from ever leaving the ground state "do nothing" (profiles/r03_train_curve_*.jsonl):
  * termination height 0.325 m sits 6 mm under the default stance's 0.331 m, and landing from the 0.42 m spawn height compresses
    the Kp = 50 legs by 6 cm: every episode ends at the first touchdown (z_threshold -> 0.25);
  * survive = 0.2 is less than what a standing robot pays in foot_contacts_z (1e-4 * 4 * 35^2 N^2 = 0.49): dying is cheaper
    than standing (survive -> 2.0);
  * energy_square = -6e-5 and foot_contacts_z = -1e-4 price a trot (tens of W per joint, 70 N per stance foot) at 1-2 per step
    against at most 0.45 from tracking_lin_vel_x_l1: standing still is the optimum (both x 0.1).
With these four numbers changed and NOTHING else -- same reward terms (WG:1352-1469), same command / goal curricula, same
schedules -- the loop learns to land, stand, walk forward at the commanded speed and move the gripper to its goals; the DAgger
updates (every 20th iteration) teach the history encoder, after which the student policy (no privileged observations) behaves
like the teacher. That is what this test asserts."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _ee_sphere_error(env):
    """sum |cart2sphere(EE in the yaw frame) - curr_ee_goal_sphere| (what metric tracking_ee_sphere accumulates, WG:1352-1357)."""
    from wbc_amd.envs import _yaw_quat
    yq = _yaw_quat(env.base_quat)
    rel = env.ee_pos - torch.cat([env.root_states[:, :2], env.z_invariant_offset], dim=1)
    s, c = 2 * yq[:, 3] * -yq[:, 2], 1 - 2 * yq[:, 2] * yq[:, 2]
    loc = torch.stack([c * rel[:, 0] - s * rel[:, 1], s * rel[:, 0] + c * rel[:, 1], rel[:, 2]], 1)
    sph = torch.stack([loc.norm(dim=1), torch.atan2(loc[:, 2], loc[:, :2].norm(dim=1)), torch.atan2(loc[:, 1], loc[:, 0])], 1)
    return (sph - env.curr_ee_goal_sphere).abs().sum(1)


class _Meter:
    """Per-step means over the live envs, accumulated on the device."""
    KEYS = ("resets", "len", "vx_err", "cmd_abs", "vx", "ee_err", "rew", "arm")

    def __init__(self, env):
        self.env, self.raw = env, env.step
        self.acc = {k: torch.zeros((), device=env.device) for k in self.KEYS}
        self.steps = 0
        env.step = self.step

    def reset(self):
        for v in self.acc.values():
            v.zero_()
        self.steps = 0

    def step(self, a):
        env = self.env
        ep_before = env.episode_length_buf.clone()
        out = self.raw(a)
        m = env.reset_buf > 0
        live = ~m
        nl = live.sum().clamp(min=1).float()
        self.acc["resets"] += m.float().mean()
        self.acc["len"] += ((ep_before + 1) * m).sum().float() / m.sum().clamp(min=1).float()
        self.acc["vx_err"] += ((env.commands[:, 0] - env.base_lin_vel[:, 0]).abs() * live).sum() / nl
        self.acc["cmd_abs"] += (env.commands[:, 0].abs() * live).sum() / nl
        self.acc["vx"] += (env.base_lin_vel[:, 0] * live).sum() / nl
        self.acc["ee_err"] += (_ee_sphere_error(env) * live).sum() / nl
        self.acc["rew"] += env.rew_buf.mean()
        self.acc["arm"] += env.arm_rew_buf.mean()
        self.steps += 1
        return out

    def means(self):
        return {k: v.item() / max(self.steps, 1) for k, v in self.acc.items()}


def test_policy_learns_to_walk_and_reach_from_scratch_and_the_student_follows():
    from wbc_amd.config import WidowGo1RoughCfg, WidowGo1RoughCfgPPO, class_to_dict
    from wbc_amd.envs import WidowGo1
    from wbc_amd.rsl_rl.runners import OnPolicyRunner
    cfg = WidowGo1RoughCfg()
    cfg.env.num_envs = 4096
    cfg.terrain.mesh_type = "plane"
    cfg.termination.z_threshold = 0.25                 # the four numbers of the module docstring
    cfg.rewards.scales.survive = 2.0
    cfg.rewards.scales.energy_square = -6e-6
    cfg.rewards.scales.foot_contacts_z = -1e-5
    train = class_to_dict(WidowGo1RoughCfgPPO())
    # the shipped file hard-codes RESUME = True (widowGo1_config.py:35), i.e. the fine-tuning schedules; from scratch it is the other
    # branch of widowGo1_config.py:359,366 -- here on a 5x shorter clock, so that one minute of GPU covers both phases (Advantage
    # Mixing ramp to beta = 1 over 600 iterations, ROA regulariser ramping in from iteration 600)
    train["algorithm"]["mixing_schedule"] = [1.0, 0, 600]
    train["algorithm"]["priv_reg_coef_schedual"] = [0, 0.1, 600, 1400]
    torch.manual_seed(train["seed"])
    env = WidowGo1(cfg, sim_device="cuda:0", seed=train["seed"])
    runner = OnPolicyRunner(env, train, log_dir=None, device="cuda:0")
    meter = _Meter(env)
    runner.learn(30, init_at_random_ep_len=True)
    early = meter.means()                               # a random policy: falls at touchdown, the gripper anywhere
    runner.learn(1370)
    meter.reset()
    runner.learn(40)
    late = meter.means()
    print("training rollouts, first 30 / last 40 of 1440 iterations:", {k: (round(early[k], 4), round(late[k], 4)) for k in early})
    assert early["resets"] > 0.02 and early["ee_err"] > 1.0
    assert late["resets"] < 0.004 and late["len"] > 400                                     # episodes run to their 500-step time-out
    assert late["cmd_abs"] > 0.3 and late["vx"] > 0.6 * late["cmd_abs"]                     # walks forward at most of the commanded speed
    assert late["vx_err"] < 0.6 * late["cmd_abs"]                                           # standing still would score cmd_abs
    assert late["ee_err"] < 0.5                                                             # gripper at its goal (1.4 untrained)
    assert all(torch.isfinite(p).all() for p in runner.alg.actor_critic.parameters())

    def evaluate(hist, steps=250):
        ac = runner.alg.actor_critic
        meter.reset()
        obs = env.get_observations()
        with torch.inference_mode():
            for _ in range(steps):
                obs = env.step(ac.act_inference(obs, hist_encoding=hist))[0]
        return meter.means()
    teacher, student = evaluate(False), evaluate(True)
    print("deterministic rollouts: teacher", {k: round(v, 4) for k, v in teacher.items()}, "student", {k: round(v, 4) for k, v in student.items()})
    # Regularized Online Adaptation: the student (history latent instead of the privileged one) matches the teacher
    assert student["rew"] > 0.9 * teacher["rew"] and student["arm"] > 0.9 * teacher["arm"]
    assert student["resets"] < 0.006 and abs(student["vx_err"] - teacher["vx_err"]) < 0.1 * max(teacher["cmd_abs"], 0.1)
    assert abs(student["ee_err"] - teacher["ee_err"]) < 0.1
