"""The reference's only recorded time series of simulator state: media/plot_1.png, the joint-angle plot that
MJ_Controller.move_group_to_joint_target(group="Arm", plot=True) saves (MujocoController.py:303-304,338-339,639-705; README.md:120-123).
tools/gen_golden_plot1.py digitised it into tests/golden/plot_1.json: 21 samples (steps 20..420) of the six arm joints moving from
qpos 0 to the plotted target under the PID loop, tolerance 0.05.

What it pins (reference-held, not oracle-derived): PID gains and output limits, the actuator model, joint damping / armature, link inertias
and the gravity load of the arm -- together they set the ramp rate of every joint and the step at which it settles.

Reading accuracy is 7-9 mrad per pixel. The plotted code version is older than the shipped one (it sampled every 20th step, today every
2nd) and its gripper is unknown, so the bars are: every sample within 0.045 rad on a 2 rad move, every ramp rate within 4 %, the move ends
in the same 20-step sampling interval."""
import json
import os

import numpy as np
import pytest

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "plot_1.json")))
NAMES = list(GOLD["joints"])
TARGET = np.array([GOLD["joints"][n]["target"] for n in NAMES])
REF = np.array([[np.nan if v is None else v for v in GOLD["joints"][n]["q"]] for n in NAMES]).T      # [sample, joint]
STEPS = np.array(GOLD["sample_steps"])
ARM = 0b111111


def _check_against_the_plot(q, total_steps):
    """q[sample, 6]: arm angles at the reference's sampling points (before sim.step() number `step`, i.e. after step - 1 steps)."""
    assert np.nanmax(np.abs(q - REF)) < 0.045, np.nanmax(np.abs(q - REF), axis=0)
    i0, i1 = 4, 10                                                    # steps 100..220: every joint is on its rate-limited ramp
    for j in range(5):                                                # wrist_3 does not move
        ref_rate = (REF[i1, j] - REF[i0, j]) / (STEPS[i1] - STEPS[i0])
        rate = (q[i1, j] - q[i0, j]) / (STEPS[i1] - STEPS[i0])
        assert abs(rate / ref_rate - 1) < 0.04, (NAMES[j], rate, ref_rate)
    for j in (0, 1, 2, 4):                                            # first sample inside the tolerance band: same sample +- 1
        tol_in = lambda a: int(np.argmax(np.abs(np.nan_to_num(a, nan=TARGET[j]) - TARGET[j]) < 0.05))
        assert abs(tol_in(q[:, j]) - tol_in(REF[:, j])) <= 1, NAMES[j]
    assert np.all(np.abs(q[:, 3] - TARGET[3]) > 0.05)                 # wrist_1 is the last joint in: no sample of it inside the band
    assert np.abs(q[:, 5]).max() < 2e-3 and np.nanmax(np.abs(REF[:, 5])) < 2e-3
    assert STEPS[-1] < total_steps <= STEPS[-1] + 20, total_steps     # a sample at 420, none at 440


def test_oracle_reproduces_the_recorded_arm_trajectory(model_2f):
    from oracle.oracle import Oracle
    m = model_2f
    o = Oracle(m)
    q0 = o.get_state()["qpos"].copy()
    q0[:6] = 0.0
    o.set_state(qpos=q0, qvel=np.zeros(m.nv), warmstart=np.zeros(m.nv))
    res, steps, ps, pq = o.move_group_plot(ARM, TARGET, 0.05, 10000, every=20)
    assert res == 0 and list(ps) == list(STEPS)
    _check_against_the_plot(pq, steps)


def _engine_trajectory(BatchSim, model, **kw):
    """One launch: scene k stops after STEPS[k] - 1 steps (its max_steps), the last scene runs to the tolerance."""
    n = len(STEPS) + 1
    sim = BatchSim(model, n, **kw)
    q0 = np.array(model.qpos0, dtype=np.float64)
    q0[:6] = 0.0
    sim.set_state(qpos=q0, qvel=np.zeros(model.nv), warmstart=np.zeros(model.nv))
    mx = np.concatenate([STEPS - 1, [10000]]).astype(np.int32)
    res, steps = sim.move_group(ARM, TARGET, 0.05, mx)
    q = sim.get_state()["qpos"][:, :6]
    assert res[-1] == 0
    return q[:-1], int(steps[-1])


def test_engine_source_reproduces_the_recorded_arm_trajectory(model_it1, emul_lib):
    """The same engine source as the HIP build (lane emulation on the CPU), and the same answer as the oracle to 1e-6."""
    from mujoco_rl_ur5_amd.native import BatchSim
    from oracle.oracle import Oracle
    q, total = _engine_trajectory(BatchSim, model_it1, lib_path=emul_lib)
    _check_against_the_plot(q, total)
    o = Oracle(model_it1)
    q0 = np.array(model_it1.qpos0, dtype=np.float64)
    q0[:6] = 0.0
    o.set_state(qpos=q0, qvel=np.zeros(model_it1.nv), warmstart=np.zeros(model_it1.nv))
    res, steps, ps, pq = o.move_group_plot(ARM, TARGET, 0.05, 10000, every=20)
    assert steps == total and np.abs(pq - q).max() < 1e-6


@pytest.mark.gpu
def test_hip_engine_reproduces_the_recorded_arm_trajectory(model_it1):
    from mujoco_rl_ur5_amd.native import BatchSim
    q, total = _engine_trajectory(BatchSim, model_it1)
    _check_against_the_plot(q, total)


# ---------------------------------------------------------------------------------------------- the plot's dashed lines are an ikpy output
# media/plot_1.png was saved by a move_ee call: its dashed target lines are what ikpy [3P] returned for that call (MujocoController.py:498-500,
# 509: angles[1:-2] become the arm targets). D3 replaces ikpy (scipy optimiser from the zero pose) by a Levenberg-Marquardt iteration from the home
# pose; this is the one reference-held IK answer: the forward kinematics of the plotted angles, fed back through the IK, must return the plotted
# angles -- same branch of the UR5's eight -- to well inside the 7-9 mrad digitisation (measured: 0.3 / 0.1 / 0.8 / 3.3 / 1.9 mrad).
GRIPPER_CENTRE = np.array([0.0, -0.005, 0.16])                        # MujocoController.py:493


def _ee_target_of_the_plotted_pose(model):
    """World xyz that move_ee must have been called with: ee_link position at the plotted angles minus the gripper-centre offset (:487-493)."""
    from oracle.oracle import Oracle
    o = Oracle(model)
    q = o.get_state()["qpos"].copy()
    q[:6] = TARGET
    o.set_state(qpos=q, qvel=np.zeros(model.nv), warmstart=np.zeros(model.nv))
    o.forward()
    return o.body_xpos()[model.body_name2id("ee_link")] - GRIPPER_CENTRE


def test_oracle_ik_returns_the_plotted_ikpy_angles(model_2f):
    from oracle.oracle import Oracle
    ok, q5 = Oracle(model_2f).ik(_ee_target_of_the_plotted_pose(model_2f))
    assert ok and np.abs(q5 - TARGET[:5]).max() < 0.01, q5 - TARGET[:5]


def test_engine_source_ik_returns_the_plotted_ikpy_angles(model_it1, emul_lib):
    from mujoco_rl_ur5_amd.native import BatchSim
    q5, res = BatchSim(model_it1, 2, lib_path=emul_lib).ik(_ee_target_of_the_plotted_pose(model_it1))
    assert (res == 0).all() and np.abs(q5 - TARGET[:5]).max() < 0.01, q5 - TARGET[:5]


@pytest.mark.gpu
def test_hip_ik_returns_the_plotted_ikpy_angles(model_it1):
    from mujoco_rl_ur5_amd.native import BatchSim
    from oracle.oracle import Oracle
    x = _ee_target_of_the_plotted_pose(model_it1)
    q5, res = BatchSim(model_it1, 4).ik(x)
    assert (res == 0).all() and np.abs(q5 - TARGET[:5]).max() < 0.01, q5 - TARGET[:5]
    assert np.abs(q5 - Oracle(model_it1).ik(x)[1]).max() < 1e-9
