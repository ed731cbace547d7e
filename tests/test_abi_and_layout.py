"""Host-side checks that need no GPU: the C-ABI library loads and exports every symbol of include/ur5sim.h, fails loudly
without a device, and the product never touches the oracle."""
import ctypes as C
import os
import re
import subprocess

import pytest

from conftest import ROOT
from mujoco_rl_ur5_amd import native


def _declared_symbols(header="ur5sim.h"):
    src = open(os.path.join(ROOT, "include", header)).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(ur5_[a-z0-9_]+)\s*\(", src)))


def test_header_and_binding_agree():
    assert _declared_symbols() == sorted(native.EXPORTS)
    assert _declared_symbols("ur5sim_test.h") == sorted(native.TEST_EXPORTS)      # the test hooks live in their own header


def test_library_exports_every_declared_symbol():
    import shutil
    if shutil.which("hipcc"):      # rebuilds only when a source is newer than the library
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "mujoco_rl_ur5_amd", "csrc"), "-s", "libur5sim.so"])
    lib = C.CDLL(native.DEFAULT_LIB)
    for sym in _declared_symbols() + _declared_symbols("ur5sim_test.h"):
        assert hasattr(lib, sym), sym


def test_no_cpu_fallback(model_it1):
    """Without a GPU ur5_create must refuse (UR5_ERR_NOGPU) -- never compute on the host."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible here")
    with pytest.raises(RuntimeError, match="no HIP device|no CPU fallback"):
        native.BatchSim(model_it1, 2)


def test_missing_library_is_loud(tmp_path):
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        native.load(str(tmp_path / "libur5sim.so"))


def test_product_never_references_the_oracle():
    pkg = os.path.join(ROOT, "mujoco_rl_ur5_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp")) or f == "Makefile":
                txt = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", txt, flags=re.M), f
                assert "libur5_oracle" not in txt and "tests/emul/_build" not in txt and "libur5sim_emul" not in txt, f
    head = open(os.path.join(ROOT, "oracle", "ur5_oracle.cpp")).read(3000)
    assert "TEST INFRASTRUCTURE ONLY" in head and "PARITY UNPINNED" in head


def test_oracle_library_loads():
    from oracle import oracle
    lib = oracle.lib()
    for sym in ("ur5o_create", "ur5o_step", "ur5o_grasp_attempt", "ur5o_move_group"):
        assert hasattr(lib, sym)
