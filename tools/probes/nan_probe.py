"""Order-dependence probe: run a 'polluter' (other kernels that leave their data in LDS / recycled device memory), then reset + render a
2-scene six-object environment and report NaNs in state and depth.  usage: UR5SIM_LIB=... python tools/probes/nan_probe.py <polluter>"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mujoco_rl_ur5_amd.model import load_model
from mujoco_rl_ur5_amd.native import BatchSim
from mujoco_rl_ur5_amd.envs import GraspEnv
pol = sys.argv[1] if len(sys.argv) > 1 else "none"
if pol in ("torch", "all"):
    x = torch.full((64, 64, 128, 128), float("nan"), device="cuda")
    w = torch.randn(64, 64, 3, 3, device="cuda")
    y = torch.nn.functional.conv2d(x, w); torch.cuda.synchronize(); del x, y, w
if pol in ("many", "all"):
    mm = load_model("/UR5+gripper/UR5gripper_2_finger_many_objects.xml")
    big = BatchSim(mm, 256); big.reset(20 + np.arange(256, dtype=np.uint64), 1, 40.0); big.close()
if pol in ("it1", "all"):
    m = load_model("it1_4box")
    s = BatchSim(m, 4096); s.reset(20 + np.arange(4096, dtype=np.uint64), 1, 100.0); s.close()
if pol in ("fill", "all"):
    bufs = [torch.full((1 << 26,), float("nan"), dtype=torch.float64, device="cuda") for _ in range(8)]
    torch.cuda.synchronize(); del bufs; torch.cuda.empty_cache()
m2 = load_model("/UR5+gripper/UR5gripper_2_finger.xml")
env = GraspEnv(file=m2, show_obs=False, n_envs=2, observation="render")
st0 = env.sim.get_state()
obs = env.reset()
st = env.sim.get_state()
d = np.asarray(obs["depth"], dtype=np.float64)
print(pol, os.environ.get("UR5SIM_LIB", "default").split("/")[-1], "nan qpos before/after reset:", int(np.isnan(st0["qpos"]).sum()), int(np.isnan(st["qpos"]).sum()),
      "nan depth per scene:", np.isnan(d).reshape(2, -1).sum(axis=1).tolist(), "status:", env.sim.counters()["status"].tolist())
