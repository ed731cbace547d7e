import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

EMUL_DIR = os.path.join(ROOT, "tests", "emul")
EMUL_LIB = os.path.join(EMUL_DIR, "_build", "libur5sim_emul.so")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """`gpu` tests need the MI355X and the built HIP library: skip (not fail) them on a box without either."""
    import torch
    lib = os.path.join(ROOT, "mujoco_rl_ur5_amd", "csrc", "libur5sim.so")
    why = None
    if not torch.cuda.is_available():
        why = "no GPU visible"
    elif not os.path.exists(lib):
        why = "mujoco_rl_ur5_amd/csrc/libur5sim.so is not built"
    if why:
        skip = pytest.mark.skip(reason=why)
        for it in items:
            if "gpu" in it.keywords:
                it.add_marker(skip)


def _engine_sources():
    """Every file of the engine source the test builds compile: the header, its per-phase parts (ur5_engine_*.inc) and the host side."""
    import glob
    d = os.path.join(ROOT, "mujoco_rl_ur5_amd", "csrc")
    return [os.path.join(d, f) for f in ("ur5_engine.h", "ur5sim_host.h", "ur5_devmodel.h", "ur5_raster.h", "ur5_many_names.h")] + sorted(glob.glob(os.path.join(d, "ur5_engine_*.inc")))


def build_emul(flags=(), name="libur5sim_emul.so"):
    """Test-only lane-emulation build of the engine source (see tests/emul/ur5sim_emul.cpp)."""
    lib = os.path.join(os.path.dirname(EMUL_LIB), name)
    srcs = [os.path.join(EMUL_DIR, f) for f in ("ur5sim_emul.cpp", "ur5sim_emul_many.cpp")]
    deps = srcs + _engine_sources()
    if not os.path.exists(lib) or os.path.getmtime(lib) < max(os.path.getmtime(s) for s in deps):
        os.makedirs(os.path.dirname(lib), exist_ok=True)
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", *flags, "-o", lib] + srcs)
    return lib


def build_simt():
    """Test-only host build of the engine's DEVICE code path, one fibre per lane of a wavefront (tests/emul/ur5sim_simt.cpp)."""
    # UR5_SIMT_FLAGS="-DUR5_PANEL_SLOT=16" pytest tests/test_engine_simt.py ... checks a build option of the engine on the wavefront emulation before it costs GPU time
    extra = os.environ.get("UR5_SIMT_FLAGS", "").split()
    tag = "".join(c if c.isalnum() else "_" for c in "".join(extra))
    lib = os.path.join(os.path.dirname(EMUL_LIB), f"libur5sim_simt{('_' + tag) if tag else ''}.so")
    srcs = [os.path.join(EMUL_DIR, f) for f in ("ur5sim_simt.cpp", "ur5sim_simt_many.cpp")]
    deps = srcs + [os.path.join(EMUL_DIR, "ur5_simt_shim.h")] + _engine_sources()
    if not os.path.exists(lib) or os.path.getmtime(lib) < max(os.path.getmtime(s) for s in deps):
        os.makedirs(os.path.dirname(lib), exist_ok=True)
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-Wno-unknown-pragmas", "-I" + EMUL_DIR, *extra, "-o", lib] + srcs)
    return lib


@pytest.fixture(scope="session")
def emul_lib():
    return build_emul()


@pytest.fixture(scope="session")
def simt_lib():
    import platform
    if platform.machine() != "x86_64":
        pytest.skip("the fibre switch of the SIMT test build is x86-64 assembly (ucontext fallback is too slow for the suite)")
    return build_simt()


@pytest.fixture(scope="session")
def model_it1():
    from mujoco_rl_ur5_amd.model import load_model
    return load_model("it1_4box")


@pytest.fixture(scope="session")
def model_2f():
    from mujoco_rl_ur5_amd.model import load_model
    return load_model("/UR5+gripper/UR5gripper_2_finger.xml")


def aimed_actions(qpos, nobj, table_z=0.91, first_id=0):
    """World xyz above object (env % nobj) for every env: the synthetic action rule of SURVEY.md section 8d, config 2."""
    import numpy as np
    n = qpos.shape[0]
    acts = np.zeros((n, 3))
    for e in range(n):
        objs = qpos[e][8:].reshape(-1, 7)
        k = (first_id + e) % nobj
        acts[e] = [objs[k, 0], -0.6 + objs[k, 1], table_z]
    return acts
