#!/usr/bin/env python3
"""Golden vectors for mujoco_rl_ur5_amd/qnet.py from the reference's own Modules.py (run in the build container, where
/root/reference exists; the GPU box only sees the committed JSON).

Modules.py imports torchvision.transforms and prettytable at module level (neither is installed here, neither is used by the
network classes), so both are stubbed before the import. Weights: torch.manual_seed(0) before construction -- qnet.py creates its
layers in the same order, so the same seed gives the same initial weights and the outputs can be compared directly.
"""
import json
import os
import sys
import types

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
for name in ("torchvision", "torchvision.transforms", "prettytable"):
    sys.modules[name] = types.ModuleType(name)
sys.modules["torchvision"].transforms = sys.modules["torchvision.transforms"]
sys.modules["prettytable"].PrettyTable = object
sys.path.insert(0, REF)
import Modules as R  # noqa: E402

out = {}
for tag, make in (("MULTIDISCRETE_RESNET_6", lambda: R.MULTIDISCRETE_RESNET(6)), ("RESNET", R.RESNET), ("POLICY_RESNET", R.POLICY_RESNET)):
    torch.manual_seed(0)
    net = make().eval()
    g = torch.Generator().manual_seed(1)
    x = torch.randn(2, 4, 40, 40, generator=g)
    with torch.no_grad():
        y = net(x)
    flat = y.reshape(-1)
    idx = torch.linspace(0, flat.numel() - 1, 24).long()
    out[tag] = {"keys": [[k, list(v.shape)] for k, v in net.state_dict().items()], "n_params": sum(p.numel() for p in net.parameters()),
                "out_shape": list(y.shape), "sum": float(flat.double().sum()), "abs_sum": float(flat.double().abs().sum()),
                "sample_idx": idx.tolist(), "sample": [float(v) for v in flat[idx]]}
# train-mode forward (batch-norm batch statistics), the mode learn() runs in
torch.manual_seed(0)
net = R.MULTIDISCRETE_RESNET(6).train()
g = torch.Generator().manual_seed(2)
x = torch.randn(3, 4, 40, 40, generator=g)
y = net(x)
out["train_mode"] = {"sum": float(y.double().sum()), "abs_sum": float(y.double().abs().sum())}
# ReplayBuffer semantics (Modules.py:28-55): ring overwrite, the most recent transition is always part of a sample
buf = R.ReplayBuffer(5, simple=True)
for i in range(8):
    buf.push(i, 10 * i, i % 2)
out["replay"] = {"len": len(buf), "position": buf.position, "stored_states": [t.state for t in buf.memory],
                 "sample_last": [buf.sample(3)[-1].state for _ in range(4)]}
# one optimiser step of Grasp_Agent.learn() (Grasping_Agent_multidiscrete.py:388-446, GAMMA = 0, BATCH_SIZE 12, Adam lr 1e-3 / weight decay 2e-5
# :27-38,153-156) on a fixed batch, with the reference's network: loss before the step, outputs after it
import torch.nn.functional as F  # noqa: E402
torch.manual_seed(0)
net = R.MULTIDISCRETE_RESNET(6).train()
opt = torch.optim.Adam(net.parameters(), lr=0.001, weight_decay=0.00002)
g = torch.Generator().manual_seed(3)
state = torch.rand(12, 4, 40, 40, generator=g)
action = torch.randint(0, 6 * 40 * 40, (12, 1), generator=g)
reward = torch.randint(0, 2, (12, 1), generator=g)
q_pred = net(state).view(12, -1).gather(1, action)
loss = F.binary_cross_entropy(q_pred, reward.float())
loss.backward()
opt.step()
opt.zero_grad()
net.eval()
with torch.no_grad():
    y2 = net(state[:2])
out["learn_step"] = {"loss": float(loss), "q_pred": [float(v) for v in q_pred.detach().reshape(-1)],
                     "after_sum": float(y2.double().sum()), "after_abs_sum": float(y2.double().abs().sum())}
# the reference's learning CADENCE on a fixed transition stream (Grasping_Agent_multidiscrete.py:551-556: memory.push of ONE transition, then learn()):
# the reference's own ReplayBuffer (python `random` seeded with 20 in its constructor, the newest transition always in the batch) and network;
# learn() starts once 2 * BATCH_SIZE transitions are stored (:396-398). rgb values are multiples of 1/255 (what ToTensor of a uint8 image gives).
torch.manual_seed(0)
net = R.MULTIDISCRETE_RESNET(6).train()
opt = torch.optim.Adam(net.parameters(), lr=0.001, weight_decay=0.00002)
g = torch.Generator().manual_seed(4)
NSEQ, HS = 36, 24
rgb = torch.randint(0, 256, (NSEQ, 3, HS, HS), generator=g).float() / 255.0
dep = torch.rand(NSEQ, 1, HS, HS, generator=g)
states = torch.cat((rgb, dep), dim=1)
actions = torch.randint(0, 6 * HS * HS, (NSEQ, 1), generator=g)
rewards = torch.randint(0, 2, (NSEQ, 1), generator=g)
buf = R.ReplayBuffer(30, simple=True)            # smaller than the stream: the ring wraps
seq_losses = []
for i in range(NSEQ):
    buf.push(states[i:i + 1], actions[i:i + 1], rewards[i:i + 1])
    if len(buf) < 24:
        continue
    batch = R.simple_Transition(*zip(*buf.sample(12)))
    q_pred = net(torch.cat(batch.state)).view(12, -1).gather(1, torch.cat(batch.action))
    loss = F.binary_cross_entropy(q_pred, torch.cat(batch.reward).float())
    loss.backward()
    opt.step()
    opt.zero_grad()
    seq_losses.append(float(loss))
net.eval()
with torch.no_grad():
    y3 = net(states[:2])
out["learn_sequence"] = {"n": NSEQ, "size": HS, "mem": 30, "losses": seq_losses, "after_sum": float(y3.double().sum()), "after_abs_sum": float(y3.double().abs().sum())}
with open(os.path.join(ROOT, "tests", "golden", "qnet_reference.json"), "w") as f:
    json.dump(out, f, indent=1)
print({k: (v.get("n_params"), v.get("out_shape")) for k, v in out.items() if "keys" in v})
