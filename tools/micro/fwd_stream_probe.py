"""configs[1] in the reference's loop shape (one pair per pass, pair after pair): forward_clouds_stream with 1 .. 4 execution
streams and helper depths 2 / 3.  python3 tools/micro/fwd_stream_probe.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from gcl_amd import synthetic
from gcl_amd.model import load_model
from gcl_amd.scripts.test_kitti import forward_clouds, forward_clouds_stream
dev = torch.device("cuda:0")
torch.manual_seed(0); np.random.seed(0)
torch.set_num_threads(max(1, min(16, len(os.sched_getaffinity(0)))))
model = load_model("ResUNetBN2C")(1, 32, bn_momentum=0.05, normalize_feature=True, conv1_kernel_size=5, D=3).to(dev).eval()
pairs = [synthetic.make_eval_pair(100 + s, baseline=15.0 + 5.0 * (s % 6)) for s in range(8)]
dpairs = [{k: (v.to(dev) if isinstance(v, torch.Tensor) else v) for k, v in p.items()} for p in pairs]
nv = sum(len(d["sinput0_C"]) + len(d["sinput1_C"]) for d in dpairs)
def timed(fn):
    fn(); torch.cuda.synchronize()
    t0, n = time.perf_counter(), 0
    while time.perf_counter() - t0 < 2.0:
        fn(); n += 1
    torch.cuda.synchronize()
    return nv * n / (time.perf_counter() - t0) / 1e6
with torch.no_grad(), torch.cuda.device(dev):
    def serial():
        for d in dpairs:
            forward_clouds(model, [(d[f"sinput{k}_F"], d[f"sinput{k}_C"]) for k in (0, 1)])
    print(f"one call per pair: {timed(serial):.1f} M voxels/s", flush=True)
    for depth in (2, 3):
        for ns in (1, 2, 3, 4):
            def run():
                for _ in forward_clouds_stream(model, ([(d[f"sinput{k}_F"], d[f"sinput{k}_C"]) for k in (0, 1)] for d in dpairs),
                                               device=dev, depth=depth, exec_streams=ns):
                    pass
            print(f"depth {depth} exec_streams {ns}: {timed(run):.1f} M voxels/s", flush=True)
