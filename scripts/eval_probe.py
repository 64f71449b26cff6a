import json, os, sys, time
sys.path.insert(0, "/root/repo")
import torch
import bench
dev = torch.device("cuda:0")
r = bench.eval_render_leg(dev, reps=3, profile_path="gpurun_out/eval_profile.txt")
print(json.dumps(r, indent=1))
