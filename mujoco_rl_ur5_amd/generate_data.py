"""``Offline RL/generate_data.py`` of the reference on the batched simulator.

The reference's script (:14-132) runs its DQN agent for 100 episodes x 50 steps and, next to the usual push-then-learn, files every transition as
(raw observation the action was chosen in, flat action index, reward), FILE_SIZE = 12 per ``Data/grasping_data_{k}.pt`` (:21, :69-90). Here one
``BatchedGraspAgent.round()`` is one such step for every scene of the rank: the observations are rendered, the Q network evaluated, the grasps simulated and
the replay pushed on the GPU; this loop only moves the round's transitions into ``dataset.GraspingDataWriter`` (scene order inside a round, rounds in time
order -- what N interleaved runs of the reference's loop would write). Files are read by the reference's own ``grasping_dataset.Grasping_Dataset``
(tests/test_dataset.py).

    python -m mujoco_rl_ur5_amd.generate_data --n-envs 256 --episodes 2 --steps 5 --out Data
"""
from __future__ import annotations

import argparse
from collections import defaultdict

from .dataset import FILE_SIZE, GraspingDataWriter


def generate_data(agent, directory="Data", episodes=100, steps=50, file_size=FILE_SIZE, learn=True, verbose=False):
    """Run ``episodes`` x ``steps`` rounds of ``agent`` (generate_data.py:40-110) and write every transition. Returns (files, reward counter)."""
    writer = GraspingDataWriter(directory, file_size)
    reward_counter = defaultdict(int)                                        # :38, :78
    for episode in range(1, episodes + 1):
        agent.env.reset()                                                    # :41
        for step in range(1, steps + 1):                                     # :52
            out = agent.round(learn=learn, return_observation=True)          # :64-72 epsilon_greedy, env.step; :104-113 push, learn
            writer.add(out["observation"], out["action"], out["reward"])     # :42-45, :66-69, :73-77
            for r in out["reward"].tolist():
                reward_counter[str(int(r))] += 1
            if verbose:
                print(f"EPISODE {episode} STEP {step}: {agent.N} transitions, success {float(out['reward'].float().mean()):.3f}, "
                      f"epsilon {out['epsilon']:.3f}, files {writer.number_saved}")
    writer.flush()
    return writer.files, dict(reward_counter)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n-envs", type=int, default=64)
    ap.add_argument("--episodes", type=int, default=2)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--out", default="Data")
    ap.add_argument("--seed", type=int, default=122)                         # generate_data.py:16
    ap.add_argument("--load-path", default=None, help="checkpoint of the reference's agent (:18, checkpoint['model_state_dict'])")
    a = ap.parse_args()
    from .agent import BatchedGraspAgent
    agent = BatchedGraspAgent(n_envs=a.n_envs, seed=a.seed, load_path=a.load_path, mem_size=100)   # :23-29
    files, counter = generate_data(agent, a.out, a.episodes, a.steps, verbose=True)
    print(f"{len(files)} files in {a.out}; reward counter {counter}")


if __name__ == "__main__":
    main()
