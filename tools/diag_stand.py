import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, os.path.join(ROOT, "deep-whole-body-control_amd"))
import torch
from wbc_amd.config import WidowGo1RoughCfg
from wbc_amd.envs import WidowGo1
torch.manual_seed(0)
cfg = WidowGo1RoughCfg(); cfg.env.num_envs = 256; cfg.terrain.mesh_type = "plane"
cfg.domain_rand.push_robots = False
env = WidowGo1(cfg, sim_device="cuda:0", seed=1)
env.update_command_curriculum()
env.reset()
def euler(q):
    x,y,z,w = q.unbind(-1)
    r = torch.atan2(2*(w*x+y*z), 1-2*(x*x+y*y)); p = torch.asin(torch.clamp(2*(w*y-z*x),-1,1))
    return r,p
for name, act in (("zero", torch.zeros(256,18,device="cuda")),):
    print("policy:", name)
    for t in range(40):
        z0 = env.root_states[:,2].clone(); q0 = env.root_states[:,3:7].clone(); g = env.sim.tensor("GOAL_STATE")[:,9:12].clone()
        env.step(act)
        m = env.reset_buf > 0
        if t < 12 or t % 5 == 0:
            print(t, "z mean %.3f min %.3f" % (env.root_states[:,2].mean().item(), env.root_states[:,2].min().item()), "resets", int(m.sum()),
                  "timeouts", int(env.time_out_buf.sum()), "contacts", (env.force_sensor_tensor.norm(dim=-1) > 1.5).float().mean().item())
# what fires: replicate termination test on pre-reset state is not available after the step; instead run manual loop with thresholds disabled
cfg2 = WidowGo1RoughCfg(); cfg2.env.num_envs = 256; cfg2.terrain.mesh_type = "plane"; cfg2.domain_rand.push_robots = False
cfg2.termination.z_threshold = -1.0
env2 = WidowGo1(cfg2, sim_device="cuda:0", seed=1)
env2.update_command_curriculum(); env2.reset()
print("z threshold disabled:")
for t in range(60):
    env2.step(torch.zeros(256,18,device="cuda"))
    r,p = euler(env2.root_states[:,3:7])
    if t < 10 or t % 10 == 0:
        print(t, "z mean %.3f min %.3f max %.3f" % (env2.root_states[:,2].mean().item(), env2.root_states[:,2].min().item(), env2.root_states[:,2].max().item()),
              "roll|max %.3f pitch|max %.3f" % (r.abs().max().item(), p.abs().max().item()), "resets", int((env2.reset_buf>0).sum()),
              "q thigh/calf FL %.3f %.3f" % (env2.dof_pos[0,1].item(), env2.dof_pos[0,2].item()))
