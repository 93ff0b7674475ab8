"""Control for tests/test_gpu_trajectory.py, oracle only (CPU, ~2 min): how far apart two runs of the SAME six reference
steps end up when they differ by float32 rounding, or by a 1e-7 relative perturbation of the initial parameters in fp64.
This is the sensitivity of the training dynamics themselves (lib/colocation_trainer.py:843-887 at lr 0.1 from a random
initialisation) and the reason the GPU test re-synchronises the oracle at every step instead of comparing end points.

    python tests/trajectory_control.py >> profiles/r06_trajectory.log
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import test_gpu_trajectory as T                      # noqa: E402
from oracle import me_oracle as O                    # noqa: E402


def main():
    torch.set_num_threads(max(1, min(16, os.cpu_count() or 1)))
    st0 = O.random_state(0)
    base, fin = T._oracle_free_run(st0)
    print(f"control: fp64 oracle, six free-running steps: loss triples {base.round(5).tolist()}")
    for name, kw in (("oracle in float32", dict(dtype=torch.float32)),
                     ("fp64 oracle, initial parameters x (1 + 1e-7 N(0,1))", dict(perturb=1e-7))):
        tr, st = T._oracle_free_run(st0, **kw)
        err = (np.abs(tr - base) / np.maximum(np.abs(base), 1e-3)).max(1)
        d = {k: T.rel_l2(st[k], fin[k]) for k in fin}
        worst = {c: max((e for k, e in d.items() if T._cls(k) == c)) for c in ("kernel", "bn", "running")}
        print(f"control: {name} vs fp64 oracle: loss-triple deviation per step {[float(f'{e:.2e}') for e in err]}; "
              f"after step {T.STEPS}: worst kernel / bn / running-stat distance "
              f"{worst['kernel']:.2e} / {worst['bn']:.2e} / {worst['running']:.2e}")


if __name__ == "__main__":
    main()
