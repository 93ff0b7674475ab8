"""The C-ABI library loads on a CPU-only box and exports exactly what include/gcl_amd.h declares
(no compute is called here: there is no GPU)."""
import ctypes
import os
import re

import pytest

from gcl_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_functions():
    src = open(os.path.join(ROOT, "include", "gcl_amd.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return set(re.findall(r"\b(gcl_[a-z0-9_]+)\s*\(", src))


def test_library_builds_and_exports_every_header_symbol():
    _lib.build()
    lib = ctypes.CDLL(_lib.LIB_PATH)
    names = _header_functions()
    assert len(names) >= 25
    for n in names:
        assert hasattr(lib, n), f"{n} declared in gcl_amd.h but not exported"
    assert names == set(_lib.SIGNATURES), "python binding table and header differ"


def test_header_argument_counts_match_binding_table():
    src = open(os.path.join(ROOT, "include", "gcl_amd.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    for name, (_, args) in _lib.SIGNATURES.items():
        m = re.search(r"\b" + name + r"\s*\(([^;]*?)\)\s*;", src, flags=re.S)
        assert m, name
        params = m.group(1).strip()
        n = 0 if params in ("", "void") else params.count(",") + 1
        assert n == len(args), (name, n, len(args))


def test_version_and_error_string_without_gpu():
    lib = _lib.load()
    assert lib.gcl_version() >= 100
    assert isinstance(lib.gcl_last_error(), bytes)
    # argument validation happens before any HIP call, so it is testable without a GPU
    rc = lib.gcl_pack_weights(None, 27, 32, 32, 0, 0, None, None, None)
    assert rc == -1 and b"null" in lib.gcl_last_error()
    p8 = ctypes.c_void_p(8)
    rc = lib.gcl_conv_fwd(p8, 10, 0, p8, 0, None, None, None, None, None, 10, 27, 32, 32, None, p8, None, 0, None)
    assert rc == -1 and b"neighbour table" in lib.gcl_last_error()
    # generic shapes (Cin / Cout no multiple of 32, K > 27) are accepted by the same entries; what they cannot take
    # (fused BN statistics, plane images) is rejected before any launch
    rc = lib.gcl_conv_fwd(p8, 10, 0, p8, 3, None, None, p8, None, None, 10, 27, 48, 32, None, p8, p8, 0, None)
    assert rc == -1 and b"multiples of 32" in lib.gcl_last_error()
    rc = lib.gcl_conv_fwd(p8, 10, 1, p8, 4, None, None, p8, None, None, 10, 27, 64, 16, None, p8, None, 0, None)
    assert rc == -1 and b"plane images" in lib.gcl_last_error()
    assert lib.gcl_pack_weights_bytes(1, 64, 16, 4) == 64 * 16 * 4          # fp32 W_eff for the generic kernels
    assert lib.gcl_pack_weights_bytes(125, 32, 32, 4) == 125 * 32 * 32 * 4
    rc = lib.gcl_stem_fwd(p8, p8, p8, 10, 27, 1, 48, p8, None, None, None)
    assert rc == -1 and b"multiple of 32" in lib.gcl_last_error()
    rc = lib.gcl_conv_fwd(p8, 10, 0, p8, 3, None, None, p8, p8, None, 10, 27, 32, 32, None, p8, None, 0, None)
    assert rc == -1 and b"go together" in lib.gcl_last_error()
    rc = lib.gcl_conv_fwd(p8, 10, 0, p8, 1, None, None, p8, None, None, 10, 27, 32, 32, None, p8, None, 0, None)
    assert rc == -1 and b"prec" in lib.gcl_last_error()
    assert lib.gcl_pack_weights_bytes(27, 64, 64, 0) == 27 * 64 * 64 * 4
    assert lib.gcl_pack_weights_bytes(27, 64, 64, 3) == 27 * 64 * 64 * 6
    assert lib.gcl_pack_weights_bytes(27, 64, 64, 4) == 27 * 64 * 64 * 4
    rc = lib.gcl_conv_fwd(p8, 10, 0, p8, 4, None, None, p8, None, None, 10, 27, 32, 32, None, p8, None, 0, None)
    assert rc == -1 and b"gcl_amax" in lib.gcl_last_error()
    # round 5's entries: sizes are host arithmetic, the argument checks come before any launch
    assert lib.gcl_nn_rowmin_scratch_len(0, 5) == 0 and lib.gcl_nn_rowmin_scratch_len(5000, 5000) >= 2500 * 2 * 64
    rc = lib.gcl_nn_rowmin(p8, None, 10, p8, None, 10, 32, 0, None, p8, p8, None)
    assert rc == -1 and b"scratch" in lib.gcl_last_error()
    rc = lib.gcl_nn_rowmin(p8, None, 10, p8, None, 10, 48, 0, p8, p8, p8, None)
    assert rc == -1 and b"16, 32 or 64" in lib.gcl_last_error()
    assert lib.gcl_sc2_register_scratch_bytes(0) == 0 and lib.gcl_sc2_register_scratch_bytes(8193) == 0
    need = lib.gcl_sc2_register_scratch_bytes(5000)
    assert need > lib.gcl_sc2_confidence_scratch_bytes(5000) > 8 * 5000 * 625 * 8
    rc = lib.gcl_sc2_register(p8, p8, 5000, 0.1, 20, 0.6, 1000, 30, 20, 0.6, 1.2, 20, None, p8, p8, p8, p8, p8, p8, p8, p8, None)
    assert rc == -1 and b"null" in lib.gcl_last_error()
    rc = lib.gcl_sc2_register(p8, p8, 9000, 0.1, 20, 0.6, 1000, 30, 20, 0.6, 1.2, 20, p8, p8, p8, p8, p8, p8, p8, p8, p8, None)
    assert rc == -1 and b"n_seeds" in lib.gcl_last_error()


def test_product_path_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import gcl_amd.MinkowskiEngine as ME
    C = torch.zeros((4, 4), dtype=torch.int32)
    C[:, 1] = torch.arange(4)
    with pytest.raises(RuntimeError, match="no CPU backend|needs an AMD GPU"):
        ME.SparseTensor(torch.ones(4, 1), coordinates=C)
    from gcl_amd.lib.eval import find_nn_gpu
    with pytest.raises(RuntimeError, match="GPU"):
        find_nn_gpu(torch.randn(4, 32), torch.randn(4, 32))
