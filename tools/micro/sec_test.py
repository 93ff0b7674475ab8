import sys, json, torch
sys.path.insert(0, '.')
import bench
print(json.dumps(bench.secondary_configs(torch.device('cuda:0'))["configs[4] eval_pairs pairs/s"]))
