"""Test configuration: registers the `gpu` marker and shared fixtures.

`-m "not gpu"` tests run on CPU (oracle vs golden vectors, host logic, C-ABI symbol check,
world_size-2 gloo tests); `-m gpu` tests are the HIP-vs-oracle parity tests proper.
"""

import json
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parents[1]
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

GOLDEN = ROOT / "tests" / "golden"
REFERENCE = Path("/root/reference")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run through gpurun)")
    _install_scratch_poison()


sys.path.insert(0, str(ROOT / "tests" / "tools"))
from build_tools import POISON_SO, build_poison_tool  # noqa: E402,F401  (tests/tools/build_tools.py)


def _install_scratch_poison():
    """On a GPU box every q_phase / q_solve launch of the suite is preceded by a kernel that fills the scratch memory AND the
    vector registers' spill-carrier lanes of every wavefront slot with a pattern (tests/tools/poison_scratch.hip): a kernel that
    reloads a register spill slot before storing to it in the same launch then computes with the pattern instead of with what
    the previous launch left behind -- the parity tests fail at once instead of by chance (round 3 found two
    such shapes).  Default pattern: a quiet NaN (0x7FC00000); STAC_TEST_POISON=<hex> picks another, STAC_TEST_POISON=off none."""
    import os

    pat = os.environ.get("STAC_TEST_POISON", "7FC00000")
    if pat.lower() in ("off", "none", ""):
        return
    import torch

    if torch.cuda.device_count() == 0:
        return  # CPU box: nothing to poison (the -m "not gpu" run)
    import ctypes

    from stac_mjx_amd import engine as eng_mod

    lib = ctypes.CDLL(str(build_poison_tool()))
    lib.poison_scratch.argtypes = [ctypes.c_void_p, ctypes.c_uint32]
    value = int(pat, 16)

    def wrap(name):
        orig = getattr(eng_mod.Engine, name)

        def poisoned(self, *a, **k):
            assert lib.poison_scratch(ctypes.c_void_p(torch.cuda.current_stream().cuda_stream), value) == 0
            return orig(self, *a, **k)

        setattr(eng_mod.Engine, name, poisoned)

    for name in ("q_phase", "q_solve"):
        wrap(name)


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def rodent_cfg():
    with open(GOLDEN / "rodent_model_cfg.json") as fh:
        return json.load(fh)


@pytest.fixture(scope="session")
def rodent_setup(rodent_cfg):
    from stac_mjx_amd.fit_model import finish_fit_setup
    from stac_mjx_amd.mjcf import ModelTables

    tables = ModelTables.load(GOLDEN / "rodent_tables.npz")
    return finish_fit_setup(tables, rodent_cfg, list(rodent_cfg["KEYPOINT_MODEL_PAIRS"].keys()))


@pytest.fixture(scope="session")
def rodent_setup_legacy(rodent_cfg):
    from stac_mjx_amd.fit_model import finish_fit_setup
    from stac_mjx_amd.mjcf import ModelTables

    tables = ModelTables.load(GOLDEN / "rodent_tables_legacy.npz")
    return finish_fit_setup(tables, rodent_cfg, list(rodent_cfg["KEYPOINT_MODEL_PAIRS"].keys()))


@pytest.fixture(scope="session")
def fly_setup():
    from stac_mjx_amd.fit_model import finish_fit_setup
    from stac_mjx_amd.mjcf import ModelTables

    with open(GOLDEN / "fly_model_cfg.json") as fh:
        cfg = json.load(fh)
    tables = ModelTables.load(GOLDEN / "fly_tables.npz")
    return finish_fit_setup(tables, cfg, list(cfg["KEYPOINT_MODEL_PAIRS"].keys()))


@pytest.fixture(scope="session")
def mouse_setup():
    from stac_mjx_amd.fit_model import finish_fit_setup
    from stac_mjx_amd.mjcf import ModelTables

    with open(GOLDEN / "mouse_model_cfg.json") as fh:
        cfg = json.load(fh)
    tables = ModelTables.load(GOLDEN / "mouse_tables.npz")
    return finish_fit_setup(tables, cfg, list(cfg["KEYPOINT_MODEL_PAIRS"].keys()))


@pytest.fixture(scope="session")
def demo_viz():
    with np.load(GOLDEN / "demo_viz_golden.npz") as d:
        return {k: d[k] for k in d.files}


@pytest.fixture(scope="session")
def rodent_mocap():
    return np.load(GOLDEN / "rodent_mocap_1000.npy")


# Toy model of the reference's m_opt known-answer tests (tests/unit/test_m_opt.py:17-37):
# three bodies in a chain, hinges about z, x, y, one marker site per body.
MINIMAL_XML = """
<mujoco>
  <worldbody>
    <body name="b1" pos="1 0 0">
      <joint name="j1" type="hinge" axis="0 0 1"/>
      <site name="s1" pos="0.1 0.2 0.3"/>
      <body name="b2" pos="0 1 0">
        <joint name="j2" type="hinge" axis="1 0 0"/>
        <site name="s2" pos="0.4 0.5 0.6"/>
        <body name="b3" pos="0 0 1">
          <joint name="j3" type="hinge" axis="0 1 0"/>
          <site name="s3" pos="0.15 0.25 0.35"/>
        </body>
      </body>
    </body>
  </worldbody>
</mujoco>
"""


@pytest.fixture(scope="session")
def toy_tables():
    from stac_mjx_amd.mjcf import compile_mjcf

    return compile_mjcf(MINIMAL_XML, from_string=True)
