#!/usr/bin/env python3
"""GPU side of the pile divergence-time statistic (round-4 verdict 3c): the kept scenes of tools/gpu_many_dump.py replayed on the HIP pile kernel with STEP CAPS
(include/ur5sim_test.h ur5_set_step_cap_dev): one copy of every scene per checkpoint, frozen after that many physics steps of its grasp attempt, all in one launch.
Writes qpos [scene, checkpoint, nq] next to the states; the CPU side (tools/pile_divergence_time.py) replays the same attempts on the oracle and its rounding twins.
    python tools/gpu_many_divergence.py <states.npz> [out.npz] [limit]"""
import json, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from mujoco_rl_ur5_amd.model import load_model
from mujoco_rl_ur5_amd.native import BatchSim

CHECKPOINTS = [5, 10, 20, 30, 40, 60, 80, 100, 120, 160, 240, 320, 400, 480, 560, 640, 800, 1000, 1200, 1500, 1800]
src = sys.argv[1]
out = sys.argv[2] if len(sys.argv) > 2 else src.replace(".npz", "_divergence_gpu.npz")
D = np.load(src)
n = min(int(sys.argv[3]), len(D["sel"])) if len(sys.argv) > 3 else len(D["sel"])
m = load_model("/UR5+gripper/UR5gripper_2_finger_many_objects.xml")
K = len(CHECKPOINTS)
rep = lambda a: np.repeat(a[:n], K, axis=0)                                    # scene-major: copies of scene e are rows e K .. e K + K - 1
sim = BatchSim(m, n * K)
sim.set_state(qpos=rep(D["qpos"]), qvel=rep(D["qvel"]), warmstart=rep(D["warmstart"]), pid=rep(D["pid"]))
sim.set_ctrl(rep(D["ctrl"]))
caps = torch.tensor(np.tile(CHECKPOINTS, n), dtype=torch.int32, device="cuda")
sim.set_step_cap_dev(caps.data_ptr())
rew, ps, pr = sim.grasp_attempt(rep(D["acts"]), rot=rep(D["rots"]), check_mode=0)
torch.cuda.synchronize()
c = sim.counters()
q = sim.get_state()["qpos"].reshape(n, K, m.nq)
steps = c["total_steps"].reshape(n, K)
# a scene whose attempt ends before a cap simply ended: its step count is the attempt's, below the cap
sim.set_step_cap_dev(None)
np.savez_compressed(out, checkpoints=np.array(CHECKPOINTS), qpos=q, steps_taken=steps, sel=D["sel"][:n])
reached = (steps == np.array(CHECKPOINTS)[None, :])
print(json.dumps(dict(scenes=n, checkpoints=CHECKPOINTS, copies=n * K, kernel_ms=sim.last_launch_ms(), copies_that_reached_their_cap=int(reached.sum()),
                      scenes_whose_attempt_is_longer_than_the_last_checkpoint=int(reached[:, -1].sum()), status_nonzero=int((c["status"] != 0).sum()), out=out)))
