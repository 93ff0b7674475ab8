"""``MinkowskiEngine.MinkowskiFunctional`` -- only ``relu`` is used by the reference (36 call sites)."""
import torch


def relu(x):
    """``MEF.relu(SparseTensor) -> SparseTensor``.  A tensor that is already the output of a fused
    BN(+residual)+ReLU kernel is returned as is (relu is idempotent; model/resunet.py:181 re-applies it)."""
    from .core import SparseTensor
    if getattr(x, "_nonneg", False):
        return x
    from . import ops
    out = SparseTensor(ops.relu(x.F), coordinate_map_key=x.coordinate_map_key,
                       coordinate_manager=x.coordinate_manager)
    out._nonneg = True
    return out
