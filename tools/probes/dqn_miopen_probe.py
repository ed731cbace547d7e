import sys, time, os, json
sys.path.insert(0, '/root/repo')
import torch
torch.backends.cudnn.benchmark = (sys.argv[1] == "1")
import bench
dev = torch.device("cuda", 0)
t0 = time.time()
r = bench.dqn_sub_result(torch, dev, 0, 512, 2, 1)
print("benchmark", sys.argv[1], "wall", round(time.time() - t0, 1), {k: (round(v, 3) if isinstance(v, float) else v) for k, v in r.items() if k != "workload"})
