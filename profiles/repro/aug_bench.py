import sys, os, json, torch
sys.path.insert(0, os.getcwd())
import bench
print(json.dumps(bench.augment_extra(torch.device("cuda:0")), indent=1))
