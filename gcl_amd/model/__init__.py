"""Model registry with the reference's ``load_model(name)`` entry (model/__init__.py:20-34)."""
from gcl_amd.model import resunet

MODELS = {n: getattr(resunet, n) for n in dir(resunet) if n.startswith("ResUNet")}


def load_model(name):
    if name not in MODELS:
        raise ValueError(f"Invalid model index. Options are: {sorted(MODELS)}")
    return MODELS[name]
