#!/usr/bin/env python3
"""Per-phase cycle breakdown of the engine (needs tools/libur5sim_prof.so = csrc/ur5sim.hip built with -DUR5_PROFILE)."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from mujoco_rl_ur5_amd.model import load_model
from mujoco_rl_ur5_amd.native import BatchSim
NAMES = ["kin", "crb", "vel", "broad", "narrow", "rows", "newton_init", "images", "linesearch", "grad+G", "H_asm", "chol", "solve", "integrate", "pid", "ik"]
n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
MANY = len(sys.argv) > 2 and sys.argv[2] == "many"        # the 40-object pile (many-object engine variant)
SIX = len(sys.argv) > 2 and sys.argv[2] == "six"          # the six-object scene (NV = 44 instantiation of the wavefront-per-scene engine)
m = load_model("/UR5+gripper/UR5gripper_2_finger_many_objects.xml" if MANY else ("/UR5+gripper/UR5gripper_2_finger.xml" if SIX else "it1_4box"))
sim = BatchSim(m, n, lib_path=os.environ.get("UR5_PROF_LIB", os.path.join(os.path.dirname(os.path.abspath(__file__)), "libur5sim_prof.so")))
sim.lib.ur5_profile_read.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
def read():
    out = np.zeros((n, 26)); sim.lib.ur5_profile_read(sim._h, out.ctypes.data_as(C.POINTER(C.c_double))); return out
seeds = np.arange(n, dtype=np.uint64) + 20
sim.reset(seeds, 1, 1000.0)
c0 = sim.counters(); p = read()
steps = c0["total_steps"].astype(float)
print("settle: kernel %.1f ms, %d steps/env; cycles per step by phase (mean over envs):" % (sim.last_launch_ms(), steps[0]))
for k, nm in enumerate(NAMES): print("  %-12s %10.0f" % (nm, (p[:, k] / steps).mean()))
def life(p):
    t0 = p[:, 16].min()
    st, en = (p[:, 16] - t0) / 1e5, (p[:, 17] - t0) / 1e5    # ms
    return "waves start %.2f..%.2f ms, end %.2f..%.2f ms, lifetime mean %.2f ms (s_memrealtime)" % (st.min(), st.max(), en.min(), en.max(), (en - st).mean())
print("  %-12s %10.0f  (%s; kernel %.2f ms)" % ("sum", (p[:, :16].sum(1) / steps).mean(), life(p), sim.last_launch_ms()))
lt = (p[:, 17] - p[:, 16]) / 1e5
print("  lifetime percentiles (ms) 10/50/90/99/max: %s; steps 10/50/90/max: %s; us per step 10/50/90/max: %s" % (
    np.percentile(lt, [10, 50, 90, 99, 100]).round(1).tolist(), np.percentile(steps, [10, 50, 90, 100]).tolist(),
    np.percentile(lt * 1e3 / steps, [10, 50, 90, 100]).round(1).tolist()))
st = sim.get_state(); acts = np.zeros((n, 3))
for e in range(n):
    objs = st["qpos"][e][8:].reshape(-1, 7); k = e % (40 if MANY else (6 if SIX else 4))
    acts[e] = [objs[k, 0], objs[k, 1], max(0.9, objs[k, 2])] if (MANY or SIX) else [objs[k, 0], -0.6 + objs[k, 1], 0.91]   # free joints hold world coordinates
rew, ps, pr = sim.grasp_attempt(acts, rot=0, check_mode=0, table_height=0.89 if MANY else 0.91)
c1 = sim.counters(); p = read(); steps = (c1["total_steps"] - c0["total_steps"]).astype(float)
print("grasp: kernel %.1f ms, mean %d steps/env, success %.2f" % (sim.last_launch_ms(), steps.mean(), rew.mean()))
for k, nm in enumerate(NAMES): print("  %-12s %10.0f" % (nm, (p[:, k] / steps).mean()))
print("  sub-intervals x0..x6 (cycles/step): %s" % (p[:, 18:25] / steps[:, None]).mean(0).round(0).tolist())
if os.environ.get("UR5_PROFILE_LEVELS"):   # library built with -DUR5_PROFILE_LEVELS: x0..x5 are parts of chol / solve (they are NOT taken out of those two lines), x7 counts level passes
    x = (p[:, 18:26] / steps[:, None]).mean(0)
    it = (c1["solver_iters"] - c0["solver_iters"]).sum() / steps.sum()
    print("  level loops, cycles/step: factorisation = blocks factored at once %.0f + panel rows (A1, forward substitution inside) incl. barrier %.0f + write-back / trailing update incl. barrier %.0f"
          " + terminal blocks (factor and solves) %.0f; solves = uncoupled blocks + forward sweep of factor-reusing iterations %.0f + backward sweep %.0f" % (x[1], x[2], x[3], x[4], x[5], x[0]))
    print("  %.2f level passes per step = %.2f per Newton iteration (%.2f iterations/step): %.0f cycles per pass for A1, %.0f for the trailing update" % (x[7], x[7] / it, it, x[2] / x[7], x[3] / x[7]))
npairs = np.floor(p[:, 25]); nverts = (p[:, 25] - npairs) * 1e9     # profile build: S.prof[PF_X7] += 1 + 1e-9 * (hull vertices of the pair) per cooperative MPR pair
if not MANY: print("  cooperative MPR: %.3f pairs per step, %.1f hull vertices per pair (one support call scans that many)" % ((npairs / steps).mean(), nverts.sum() / max(npairs.sum(), 1)))
print("  %-12s %10.0f  (%s; kernel %.2f ms)" % ("sum", (p[:, :16].sum(1) / steps).mean() + (p[:, 18:25].sum(1) / steps).mean(), life(p), sim.last_launch_ms()))
lt = (p[:, 17] - p[:, 16]) / 1e5
print("  lifetime percentiles (ms) 10/50/90/99/max: %s; steps 10/50/90/max: %s; us per step 10/50/90/max: %s" % (
    np.percentile(lt, [10, 50, 90, 99, 100]).round(1).tolist(), np.percentile(steps, [10, 50, 90, 100]).tolist(),
    np.percentile(lt * 1e3 / steps, [10, 50, 90, 100]).round(1).tolist()))
