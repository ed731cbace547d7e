"""Keeps the reference's gym id (gym_grasper/__init__.py:4-7 of the reference): ``gym.make("gym_grasper:Grasper-v0")``.

When ``gym`` is installed the id is registered with the batched MI355X environment as its entry point; without gym use
``mujoco_rl_ur5_amd.envs.make("gym_grasper:Grasper-v0", ...)``.
"""
try:  # pragma: no cover - gym is not part of this image
    from gym.envs.registration import register

    register(id="Grasper-v0", entry_point="mujoco_rl_ur5_amd.envs:GraspEnv")
except Exception:  # gym missing or already registered
    pass
