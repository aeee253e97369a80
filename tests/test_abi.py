"""The C-ABI library: builds for gfx950, loads, and exports every symbol include/basq_hip.h declares.
No compute call is made (no GPU here)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib_path():
    from basq_amd import _build

    if not os.path.exists(_build.LIB):
        try:
            _build.build(verbose=False)
        except Exception as e:                       # pragma: no cover
            pytest.fail(f"could not build the HIP library: {e}")
    return _build.LIB


def _declared():
    header = open(os.path.join(ROOT, "include", "basq_hip.h")).read()
    names = set(re.findall(r"\b(basq_[a-z0-9_]+)\s*\(", header))
    return names


def test_header_symbols_exported(lib_path):
    lib = ctypes.CDLL(lib_path)
    missing = [n for n in sorted(_declared()) if not hasattr(lib, n)]
    assert not missing, f"library does not export {missing}"


def test_binding_covers_header(lib_path):
    from basq_amd import _lib

    assert _declared() == set(_lib.SIGNATURES), "ctypes prototypes and include/basq_hip.h disagree"
    lib = _lib.load()
    assert lib.basq_abi_version() == _lib.ABI_VERSION
    assert lib.basq_strerror(0) == b"ok" and lib.basq_strerror(-1) == b"invalid argument"
    assert [lib.basq_kp(d) for d in (1, 2, 3, 10, 32, 38)] == [4, 4, 8, 12, 36, 40]
    assert lib.basq_kp(0) < 0 and lib.basq_kp(39) < 0


def test_argument_validation_without_gpu(lib_path):
    """Entry points reject bad arguments before touching the device."""
    from basq_amd import _lib

    lib = _lib.load()
    spec = _lib.KernelSpecC(0, 10, 2.0, 1.0)
    assert lib.basq_pack_points_f64(ctypes.byref(spec), None, 5, None, 0, None, None) == -1
    assert lib.basq_blocksum_f64(ctypes.byref(spec), None, 1, None, None, None, 1, 0, 0, 1, 1, 0, 0, None, None, None) == -1
    assert lib.basq_blocksum_f64(ctypes.byref(spec), 1, 1, 1, 1, None, 10, 0, 8, 4, 2, 4, 3, 1, 1, None) == -1   # class0 + n_chunks > class_mod
    assert lib.basq_blocksum_f64(ctypes.byref(spec), 1, 1, 1, 1, None, 10, 0, 8, 4, 2, 4, 0, 1, 1, None) == -1   # classes: no ragged tail
    assert lib.basq_regroup_classes_f64(None, 1, 4, 2, None, None, None, None, None) == -1
    assert lib.basq_regroup_classes_f64(1, 1, 4, 3, 1, 1, 1, 1, None) == -1                                       # C must be even
    bad = _lib.KernelSpecC(7, 10, 2.0, 1.0)
    assert lib.basq_gram_f64(ctypes.byref(bad), None, 1, None, 1, None, 1, None) == -1
    assert lib.basq_car_eliminate_f64(None, None, 2000, 10, None, None, None, None, None, None) == -1
    assert lib.basq_nullspace_f64(None, 10, 20, None, None, None, None, None, None) == -1
    assert lib.basq_nullspace_f64(1, 20, 20, 1, 1, 1, None, None, None) == -1    # needs s < M (checked before any launch)
    assert lib.basq_nullspace_f64(1, 10, 2000, 1, 1, 1, None, None, None) == -1  # M <= 1024
    # ABI 15: the epochs' message columns
    assert lib.basq_epoch_turn_f64(None, 4, 1, None, 1, 10, 20, None, None, None, None, None, None, None, None) == -1
    assert lib.basq_epoch_turn_f64(1, 3, 1, 1, 1, 10, 20, 1, 1, 1, 1, 1, 1, 1, None) == -1            # C must be even (and >= 2)
    assert lib.basq_epoch_turn_f64(1, 4, 1, 1, 1, 10, 21, 1, 1, 1, 1, 1, 1, 1, None) == -1            # S must be even
    assert lib.basq_epoch_turn_f64(1, 4, -1, 1, 1, 10, 20, 1, 1, 1, 1, 1, 1, 1, None) == -1
    ptrs = (ctypes.c_void_p * 9)(*([1] * 9))
    assert lib.basq_reweight_compact_rounds_f64(1, 1, 1, None, 1, 9, ptrs, ptrs, ptrs, ptrs, 10, 4, 4, 10, 2, 1, 1, 1, None, None) == -1   # <= 8 rounds
    assert lib.basq_reweight_compact_rounds_f64(1, 1, 1, None, 1, 0, ptrs, ptrs, ptrs, ptrs, 10, 4, 4, 10, 2, 1, 1, 1, None, None) == -1
    assert lib.basq_reweight_compact_rounds_f64(1, 1, 1, 1, 1, 2, ptrs, ptrs, ptrs, ptrs, 10, 4, 4, 10, 2, 1, 1, 1, None, None) == -1       # wx without wx_out
    nul = (ctypes.c_void_p * 2)(1, None)
    assert lib.basq_reweight_compact_rounds_f64(1, 1, 1, None, 1, 2, nul, ptrs, ptrs, ptrs, 10, 4, 4, 10, 2, 1, 1, 1, None, None) == -1    # a round without its outcome
    assert lib.basq_blocksum_geo_f64(ctypes.byref(spec), 1, 1, 1, 1, None, 1, 5, 4, 2, 0, 0, 1, 1, None) == -1   # mode 5 needs the classes
    assert lib.basq_blocksum_geo_f64(ctypes.byref(spec), 1, 1, 1, 1, None, 1, 2, 4, 2, 4, 0, 1, 1, None) == -1   # classes: modes 1 and 5 only
    assert lib.basq_blocksum_geo_f64(ctypes.byref(spec), 1, 1, 1, 1, None, 1, 6, 4, 2, 0, 0, 1, 1, None) == -1
    assert lib.basq_reduction_ws_doubles(100, 200) == 0                    # one CU: no workspace
    assert lib.basq_reduction_ws_doubles(200, 400) > 0                     # 4-CU cluster: ring + flags


def test_product_has_no_cpu_path():
    import torch

    import basq_amd
    from basq_amd._lib import BasqHipError

    with pytest.raises(BasqHipError):
        basq_amd.recombination(torch.zeros(10, 2), torch.zeros(5, 2), 3, basq_amd.kernels.StationaryKernel("rbf", 1.0),
                               torch.device("cpu"))
    with pytest.raises(BasqHipError):     # an opaque callable is accepted (chunked dense path) -- but never on the CPU
        basq_amd.recombination(torch.zeros(10, 2), torch.zeros(5, 2), 3, lambda a, b: a @ b.T, torch.device("cpu"))
    with pytest.raises(TypeError):
        basq_amd.recombination(torch.zeros(10, 2), torch.zeros(5, 2), 3, 3.0, torch.device("cpu"))


def test_product_does_not_import_oracle():
    """Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may touch oracle/."""
    pkg = os.path.join(ROOT, "basq_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, re.M), f"{f} imports the oracle"
                assert "tests.cpu_stand_in" not in src
