"""Host logic on CPU: config, io, batching, PRNG, seam protocol, C-ABI symbols (no GPU compute)."""

import os
import re
import types

import numpy as np
import pytest
import torch
import yaml

from conftest import REFERENCE, ROOT


# ---- C ABI -----------------------------------------------------------------------------------------------
def test_library_exports_every_declared_symbol():
    from stac_mjx_amd.build import build_extension
    from stac_mjx_amd.engine import ABI_SYMBOLS, load_library

    build_extension()
    header = (ROOT / "include" / "stac_hip.h").read_text()
    declared = set(re.findall(r"\b(stac_[a-z0-9_]+)\s*\(", header))
    assert declared == set(ABI_SYMBOLS), declared ^ set(ABI_SYMBOLS)
    lib = load_library()
    for s in declared:
        assert hasattr(lib, s), s
    assert lib.stac_abi_version() == 3
    assert lib.stac_device_count() >= 0


def test_engine_fails_loudly_without_gpu(rodent_setup):
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from stac_mjx_amd.engine import Engine, StacHipError

    with pytest.raises(StacHipError):
        Engine(rodent_setup.tables, rodent_setup.lb, rodent_setup.ub)


def test_product_never_imports_oracle():
    for p in (ROOT / "stac_mjx_amd").rglob("*.py"):
        txt = p.read_text()
        assert "import oracle" not in txt and "from oracle" not in txt, p
    for p in (ROOT / "stac_mjx_amd" / "csrc").iterdir():
        if p.suffix in (".hip", ".hpp", ".cpp", ".h"):
            assert '#include "../../oracle' not in p.read_text() and "stac_oracle.h" not in p.read_text(), p


# ---- PRNG ------------------------------------------------------------------------------------------------
def test_threefry_known_answer_vectors():
    """Random123 / JAX test-suite KATs."""
    from stac_mjx_amd.prng import threefry2x32

    def one(key, ctr):
        a, b = threefry2x32(key, np.array([ctr[0]], np.uint32), np.array([ctr[1]], np.uint32))
        return int(a[0]), int(b[0])

    assert one((0, 0), (0, 0)) == (0x6B200159, 0x99BA4EFE)
    assert one((0xFFFFFFFF, 0xFFFFFFFF), (0xFFFFFFFF, 0xFFFFFFFF)) == (0x1CB996FC, 0xBB002BE7)
    assert one((0x13198A2E, 0x03707344), (0x243F6A88, 0x85A308D3)) == (0xC4923A9C, 0x483DF7A0)


def test_permutation_is_a_permutation_and_stable():
    from stac_mjx_amd.prng import permutation, prng_key, sample_time_indices

    p = permutation(prng_key(0), 1000)
    assert sorted(p.tolist()) == list(range(1000))
    assert p[:10].tolist() == [166, 872, 474, 336, 210, 769, 0, 475, 36, 835]  # value predicted in SURVEY.md A3
    assert permutation(prng_key(0), 10).tolist() == [0, 1, 8, 5, 6, 4, 3, 2, 7, 9]
    assert sorted(sample_time_indices(10, 100).tolist()) == list(range(10))  # n <= N_SAMPLE_FRAMES: all frames
    assert len(sample_time_indices(5000, 100)) == 100


# ---- config ------------------------------------------------------------------------------------------------
def _write_cfg(tmp_path, rodent_cfg, n_fit_frames=42):
    (tmp_path / "stac").mkdir()
    (tmp_path / "model").mkdir()
    (tmp_path / "config.yaml").write_text("defaults:\n  - stac: s1\n  - model: m1\n  - _self_\n")
    stac = dict(fit_offsets_path="fit.h5", ik_only_path="ik.h5", data_path="d.mat", continuous=False,
                n_fit_frames=n_fit_frames, skip_fit_offsets=False, skip_ik_only=False, infer_qvels=False,
                n_frames_per_clip=1, mujoco=dict(solver="newton", iterations=1, ls_iterations=4))
    (tmp_path / "stac" / "s1.yaml").write_text(yaml.safe_dump(stac))
    (tmp_path / "stac" / "s2.yaml").write_text(yaml.safe_dump({**stac, "n_fit_frames": 7}))
    (tmp_path / "model" / "m1.yaml").write_text(yaml.safe_dump(rodent_cfg))
    return tmp_path


def test_compose_config_defaults_overrides_and_schema(tmp_path, rodent_cfg):
    from stac_mjx_amd.config import ConfigError, compose_config, load_configs

    d = _write_cfg(tmp_path, rodent_cfg)
    cfg = load_configs(d)
    assert cfg.stac.n_fit_frames == 42 and cfg.model.N_ITER_Q == 400 and cfg.model.FTOL == 1e-4
    assert cfg.model.MARKER_SIZE == 0.005 and cfg.stac.mujoco.solver == "newton"
    assert list(cfg.model.KEYPOINT_MODEL_PAIRS)[0] == "AnkleL"  # key order preserved
    cfg2 = compose_config(d, overrides=["stac=s2", "stac.n_frames_per_clip=250", "model.FTOL=0.005",
                                        "hydra/job_logging=disabled"])
    assert cfg2.stac.n_fit_frames == 7 and cfg2.stac.n_frames_per_clip == 250 and cfg2.model.FTOL == 0.005
    assert "ROOT_OPTIMIZATION_KEYPOINT" in cfg.model and cfg.model.get("NOPE", 3) == 3
    with pytest.raises(ConfigError):
        compose_config(d, overrides=["stac.bogus_key=1"])
    with pytest.raises(ConfigError):
        compose_config(d, config_name="missing")
    rt = yaml.safe_load(cfg.to_yaml())
    assert rt["model"]["SCALE_FACTOR"] == 0.9


@pytest.mark.skipif(not REFERENCE.exists(), reason="reference checkout not present")
def test_reference_config_dirs_load():
    from stac_mjx_amd.config import compose_config

    cfg = compose_config(REFERENCE / "configs", "config")
    assert cfg.stac.n_fit_frames == 10 and cfg.model.MJCF_PATH == "models/rodent.xml"
    cfgt = compose_config(REFERENCE / "tests" / "configs", "config")
    assert cfgt.stac.n_fit_frames == 42 and cfgt.stac.n_frames_per_clip == 1  # tests/unit/test_config.py


# ---- io -------------------------------------------------------------------------------------------------------
def test_load_data_mat_ordering_and_scaling(tmp_path, rodent_cfg):
    """Restates tests/test_io.py: sorted keypoint order == KEYPOINT_MODEL_PAIRS key order, shape (F, 3K)."""
    import scipy.io as spio

    from stac_mjx_amd.config import validate_config
    from stac_mjx_amd.io import load_data

    names = rodent_cfg["KP_NAMES"]
    rng = np.random.default_rng(0)
    pred = rng.normal(0, 100, (12, 3, 23))
    spio.savemat(tmp_path / "d.mat", {"pred": pred})
    stac = dict(fit_offsets_path="fit.h5", ik_only_path="ik.h5", data_path="d.mat", continuous=False, n_fit_frames=4,
                skip_fit_offsets=False, skip_ik_only=True, infer_qvels=False, n_frames_per_clip=1,
                mujoco=dict(solver="newton", iterations=1, ls_iterations=4))
    cfg = validate_config({"model": rodent_cfg, "stac": stac})
    kp, sorted_names = load_data(cfg, base_path=tmp_path)
    assert kp.shape == (12, 69) and kp.dtype == np.float32
    assert sorted_names == list(rodent_cfg["KEYPOINT_MODEL_PAIRS"].keys())
    k = sorted_names.index("SpineL")
    np.testing.assert_allclose(kp[3, 3 * k: 3 * k + 3], pred[3, :, names.index("SpineL")] * 1e-3, rtol=1e-6)
    cfg.model.KP_NAMES = names[:-1]
    with pytest.raises(ValueError):
        load_data(cfg, base_path=tmp_path)
    cfg.stac.data_path = "d.txt"
    with pytest.raises(ValueError):
        load_data(cfg, base_path=tmp_path)


def test_committed_mocap_fixture_matches_reference_mat(rodent_mocap):
    if not REFERENCE.exists():
        pytest.skip("reference checkout not present")
    import scipy.io as spio

    pred = spio.loadmat(REFERENCE / "tests/data/test_rodent_mocap_1000_frames.mat")["pred"]
    assert rodent_mocap.shape == (1000, 69)
    # column 0..2 = AnkleL = index 17 of KP_NAMES
    np.testing.assert_allclose(rodent_mocap[5, :3], (pred[5, :, 17] * 1e-3).astype(np.float32))


def test_save_and_load_stac_data_roundtrip(tmp_path, rodent_cfg):
    from stac_mjx_amd.config import validate_config
    from stac_mjx_amd.io import StacData, load_stac_data, save_data_to_h5

    stac = dict(fit_offsets_path="fit.h5", ik_only_path="ik.h5", data_path="d.mat", continuous=False, n_fit_frames=4,
                skip_fit_offsets=False, skip_ik_only=True, infer_qvels=False, n_frames_per_clip=1,
                mujoco=dict(solver="newton", iterations=1, ls_iterations=4))
    cfg = validate_config({"model": rodent_cfg, "stac": stac})
    d = StacData(qpos=np.ones((2, 74), np.float32), xpos=np.zeros((2, 67, 3), np.float32), xquat=np.zeros((2, 67, 4), np.float32),
                 marker_sites=np.zeros((2, 23, 3), np.float32), offsets=np.full((23, 3), 0.5, np.float32),
                 kp_data=np.zeros((2, 69), np.float32), names_qpos=["a"] * 74, names_xpos=["b"] * 67, kp_names=["c"] * 23)
    p = save_data_to_h5(config=cfg, file_path=tmp_path / "fit.h5", **d.as_dict())
    cfg2, d2 = load_stac_data(tmp_path / "fit.h5")
    assert p.exists() and cfg2.stac.n_fit_frames == 4 and d2.names_qpos == ["a"] * 74
    np.testing.assert_array_equal(d2.offsets, d.offsets)
    np.testing.assert_array_equal(d2.qpos, d.qpos)


# ---- batching (tests/unit/test_utils_math.py shapes) -----------------------------------------------------------------
def test_batch_kp_data_shapes_and_content():
    from stac_mjx_amd.utils import batch_kp_data, handle_edge_effects

    kp = np.arange(8 * 6, dtype=np.float32).reshape(8, 6)
    b = batch_kp_data(kp, 4)
    assert b.shape == (2, 4, 6)
    np.testing.assert_array_equal(b[1, 0], kp[4])
    kp2 = np.arange(60 * 6, dtype=np.float32).reshape(60, 6)
    c = batch_kp_data(kp2, 20, continuous=True)
    assert c.shape == (3, 30, 6)
    np.testing.assert_array_equal(c[0], kp2[:30])
    np.testing.assert_array_equal(c[2, :20], kp2[40:60])
    np.testing.assert_array_equal(c[2, 20:], kp2[40:50])  # wrap padding of the last window
    assert batch_kp_data(kp2[:59], 20).shape == (2, 20, 6)  # trailing frames dropped
    data = types.SimpleNamespace(qpos=c.copy(), kp_data=c.copy(), xpos=c.copy(), xquat=c.copy(), marker_sites=c.copy())
    for k in vars(data):
        setattr(data, k, getattr(data, k).reshape(-1, 6))
    out = handle_edge_effects(data, 20)
    assert out.qpos.shape == (60, 6)
    np.testing.assert_allclose(out.qpos[:20], kp2[:20])


# ---- seam protocol (tests/unit/test_compute_stac.py restated) ----------------------------------------------------------
class FakeData:
    def __init__(self, qpos, site_xpos=None, xpos=None, xquat=None):
        self.qpos = qpos
        self.site_xpos = site_xpos if site_xpos is not None else torch.zeros(2, 3)
        self.xpos = xpos if xpos is not None else torch.zeros(2, 3)
        self.xquat = xquat if xquat is not None else torch.zeros(2, 4)

    def replace(self, **kw):
        return FakeData(kw.get("qpos", self.qpos), kw.get("site_xpos", self.site_xpos), kw.get("xpos", self.xpos),
                        kw.get("xquat", self.xquat))


class FakeModel:
    def __init__(self, nq, jnt_type, site_pos):
        self.nq, self.jnt_type, self.site_pos = nq, jnt_type, site_pos


class FakeStacCore:
    def __init__(self):
        self.q_calls = self.m_calls = 0
        self.q0_args = []

    def q_opt(self, *args, **kw):
        self.q_calls += 1
        self.q0_args.append(args[5])
        return args[1], types.SimpleNamespace(params=args[5], state=types.SimpleNamespace(error=0.0))

    def m_opt(self, mjx_model, mjx_data, keypoints, q, initial_offsets, *a, **kw):
        self.m_calls += 1
        return types.SimpleNamespace(params=torch.as_tensor(initial_offsets).reshape(-1, 3), error=0.0)


@pytest.fixture()
def patched_utils(monkeypatch):
    from stac_mjx_amd import utils

    monkeypatch.setattr(utils, "kinematics", lambda model, data: data)
    monkeypatch.setattr(utils, "com_pos", lambda model, data: data)

    def set_site_pos(model, offsets, site_idxs=None):
        model.site_pos = offsets
        return model

    monkeypatch.setattr(utils, "set_site_pos", set_site_pos)
    return utils


def test_root_optimization_calls_q_opt_twice_and_seeds_root(patched_utils):
    from stac_mjx_amd import compute_stac

    core = FakeStacCore()
    model = FakeModel(7, np.array([0]), torch.zeros(2, 3))
    data = FakeData(torch.tensor([91.0, 92, 93, 4, 5, 6, 7]))
    kp = torch.tensor([[11.0, 12, 13, 21, 22, 23]])
    out = compute_stac.root_optimization(core, model, data, kp, 1, torch.zeros(7), torch.ones(7), torch.tensor([0, 1]),
                                         torch.tensor([True, True]))
    assert isinstance(out, FakeData) and core.q_calls == 2
    exp = torch.tensor([21.0, 22, 23, 4, 5, 6, 7])
    assert torch.allclose(core.q0_args[0], exp) and torch.allclose(core.q0_args[1], exp)


def test_offset_optimization_updates_site_pos(patched_utils):
    from stac_mjx_amd import compute_stac

    core = FakeStacCore()
    model = FakeModel(7, np.array([0]), torch.zeros(2, 3))
    offsets = torch.zeros(2, 3)
    model, data, off = compute_stac.offset_optimization(core, model, FakeData(torch.zeros(7)), torch.zeros(4, 6), offsets,
                                                        torch.zeros(4, 7), 2, torch.zeros(2, 3), torch.tensor([0, 1]), 0.0)
    assert core.m_calls == 1 and torch.allclose(off, offsets) and torch.allclose(model.site_pos, offsets)


def test_pose_optimization_runs_all_frames(patched_utils):
    from stac_mjx_amd import compute_stac

    core = FakeStacCore()
    model = FakeModel(7, np.array([0]), torch.zeros(2, 3))
    res = compute_stac.pose_optimization(core, model, FakeData(torch.zeros(7)), torch.zeros(2, 6), torch.zeros(7),
                                         torch.ones(7), torch.tensor([0, 1]), [])
    _, qposes, _, _, marker_sites, _, frame_error = res
    assert qposes.shape == (2, 7) and len(marker_sites) == 2 and len(frame_error) == 2 and core.q_calls == 2


def test_run_stac_validates_columns_before_touching_the_gpu(tmp_path, rodent_cfg):
    from stac_mjx_amd.config import validate_config
    from stac_mjx_amd.main import run_stac

    stac = dict(fit_offsets_path="fit.h5", ik_only_path="ik.h5", data_path="d.mat", continuous=False, n_fit_frames=4,
                skip_fit_offsets=False, skip_ik_only=True, infer_qvels=False, n_frames_per_clip=1,
                mujoco=dict(solver="newton", iterations=1, ls_iterations=4))
    cfg = validate_config({"model": rodent_cfg, "stac": stac})
    with pytest.raises(ValueError, match="columns"):
        run_stac(cfg, np.zeros((4, 68), np.float32), ["k"] * 23, base_path=tmp_path)


# ---- quaternion / velocity post-processing (tests/unit/test_utils_math.py restated) ------------------------------------
def test_quat_helpers_and_velocity():
    from stac_mjx_amd import utils

    q = np.array([0.5, 0.5, 0.5, 0.5])
    assert np.allclose(utils.quat_mul(np.array([1.0, 0, 0, 0]), q), q)
    assert np.allclose(utils.quat_conj(q), [0.5, -0.5, -0.5, -0.5])
    assert np.allclose(utils.quat_diff(q, q), [1.0, 0, 0, 0])
    ang = np.pi / 2
    assert np.allclose(utils.quat_to_axisangle(np.array([np.cos(ang / 2), np.sin(ang / 2), 0.0, 0.0])), [ang, 0, 0])
    assert np.allclose(utils.quat_to_axisangle(np.array([1.0, 0, 0, 0])), 0.0)
    qpos = np.array([[0.0, 0, 0], [1.0, 2, 3], [2.0, 4, 6]])
    qvel = utils.compute_velocity_from_kinematics(qpos, dt=1.0, freejoint=False, max_qvel=100.0)
    assert qvel.shape == (3, 3) and np.allclose(qvel[0], [1, 2, 3]) and np.allclose(qvel[1], [1, 2, 3]) and np.allclose(qvel[2], 0)
    # free joint: constant yaw rate about z, translation along x, one hinge
    T, dt, w = 6, 0.01, 2.0
    tt = np.arange(T) * dt
    qp = np.zeros((T, 8))
    qp[:, 0] = 3.0 * tt
    qp[:, 3], qp[:, 6] = np.cos(w * tt / 2), np.sin(w * tt / 2)
    qp[:, 7] = 1000.0 * tt  # exceeds max_qvel -> clipped
    v = utils.compute_velocity_from_kinematics(qp, dt=dt, freejoint=True)
    assert v.shape == (T, 7)
    assert np.allclose(v[:-1, 0], 3.0) and np.allclose(v[:-1, 3:6], [0, 0, w], atol=1e-6)
    assert np.allclose(v[:-1, 6], 20.0) and np.allclose(v[-1], 0.0)


# ---- continuous clips: wrap padding and the cross-fade, numerically (stac_mjx/utils.py:350-461) -----------------------
@pytest.mark.parametrize("n_per,n_clips", [(20, 3), (10, 2), (12, 4), (250, 2)])
def test_batch_kp_data_continuous_equals_pad_wrap(n_per, n_clips):
    """The windows are kp[s : s + n + 10] and the LAST one is `jp.pad(..., ((0, 10), (0, 0)), mode="wrap")`
    (utils.py:369-382; numpy's pad has the same wrap semantics): checked row by row.  Clips shorter than the
    overlap leave the second-to-last window short, which the reference's `jp.stack` rejects: ValueError here too."""
    from stac_mjx_amd.utils import CONTINUOUS_BATCH_OVERLAP as OV
    from stac_mjx_amd.utils import batch_kp_data

    rng = np.random.default_rng(n_per * 100 + n_clips)
    kp = rng.normal(size=(n_per * n_clips, 6)).astype(np.float32)
    out = batch_kp_data(kp, n_per, continuous=True)
    assert out.shape == (n_clips, n_per + OV, 6)
    for c in range(n_clips - 1):
        np.testing.assert_array_equal(out[c], kp[c * n_per : c * n_per + n_per + OV])
    last = kp[(n_clips - 1) * n_per :]
    np.testing.assert_array_equal(out[-1], np.pad(last, ((0, OV), (0, 0)), mode="wrap"))
    tk = batch_kp_data(torch.as_tensor(kp), n_per, continuous=True)  # the torch branch does the same
    np.testing.assert_array_equal(tk.numpy(), out)
    with pytest.raises((ValueError, RuntimeError)):
        batch_kp_data(kp[: 4 * 5], 4, continuous=True)


def test_handle_edge_effects_numeric():
    """Frame t of the output is clip t // n's own result, except the first 10 frames of every clip after the first:
    those are the sigmoid cross-fade (1 - m) * [tail of the previous window] + m * [head of this window] with
    m = sigmoid(10 * (linspace(0, 1, 10) - 0.5)) (utils.py:393-461)."""
    from stac_mjx_amd.utils import CONTINUOUS_BATCH_OVERLAP as OV
    from stac_mjx_amd.utils import handle_edge_effects

    n, C = 25, 4
    rng = np.random.default_rng(3)
    fields = {k: rng.normal(size=(C, n + OV) + s) for k, s in
              dict(qpos=(7,), kp_data=(6,), xpos=(5, 3), xquat=(5, 4), marker_sites=(2, 3)).items()}
    data = types.SimpleNamespace(**{k: v.reshape((-1,) + v.shape[2:]).copy() for k, v in fields.items()})
    out = handle_edge_effects(data, n)
    m = 1.0 / (1.0 + np.exp(-10.0 * (np.linspace(0.0, 1.0, OV) - 0.5)))
    for k, v in fields.items():
        o = getattr(out, k)
        assert o.shape == (C * n,) + v.shape[2:]
        for c in range(C):
            for f in range(n):
                want = v[c, f]
                if c > 0 and f < OV:
                    mm = m[f]
                    want = (1.0 - mm) * v[c - 1, n + f] + mm * v[c, f]
                np.testing.assert_allclose(o[c * n + f], want, rtol=1e-12, atol=1e-12)


# ---- packing order, per-call bounds, config rebinding -------------------------------------------------------------------
def _bare_stac(rodent_setup, rodent_cfg, **stac_over):
    """A Stac object without an engine (no GPU): enough for the host-side packing logic."""
    from stac_mjx_amd.config import validate_config
    from stac_mjx_amd.stac import Stac

    stac = dict(fit_offsets_path="fit.h5", ik_only_path="ik.h5", data_path="d.mat", continuous=False, n_fit_frames=4,
                skip_fit_offsets=False, skip_ik_only=False, infer_qvels=False, n_frames_per_clip=3,
                mujoco=dict(solver="newton", iterations=1, ls_iterations=4))
    stac.update(stac_over)
    s = Stac.__new__(Stac)
    s.cfg = validate_config({"model": dict(rodent_cfg), "stac": stac})
    s.setup = rodent_setup
    s._part_names, s._body_names, s._kp_names = rodent_setup.part_names, rodent_setup.tables.body_names, rodent_setup.kp_names
    s._offsets = torch.zeros(23, 3)
    return s


def test_package_data_marker_order_default_and_reference_compat(rodent_setup, rodent_cfg):
    """Default: every field clip-major.  `stac.reference_marker_order: true`: marker_sites in the reference's row
    order, a C-order flatten of the (F, C, K, 3) stack (stac.py:486): row j = frame j // C of clip j % C."""
    C, F = 4, 3
    rng = np.random.default_rng(0)
    res = dict(qpos=torch.as_tensor(rng.normal(size=(C, F, 74))), xpos=torch.zeros(C, F, 67, 3), xquat=torch.zeros(C, F, 67, 4),
               marker_sites=torch.as_tensor(rng.normal(size=(C, F, 23, 3))))
    kp = rng.normal(size=(C, F, 69))
    d0 = _bare_stac(rodent_setup, rodent_cfg)._package_data(res, kp, batched=True)
    d1 = _bare_stac(rodent_setup, rodent_cfg, reference_marker_order=True)._package_data(res, kp, batched=True)
    ms = res["marker_sites"].numpy()
    for j in range(C * F):
        np.testing.assert_array_equal(d0.marker_sites[j], ms[j // F, j % F])   # clip-major, like qpos
        np.testing.assert_array_equal(d1.marker_sites[j], ms[j % C, j // C])   # the reference's frame-major rows
        np.testing.assert_array_equal(d0.qpos[j], res["qpos"].numpy()[j // F, j % F])
    np.testing.assert_array_equal(d1.qpos, d0.qpos)  # only marker_sites is affected (stac.py:483-486)
    # the re-ordering map a consumer of the reference's files applies: rows (f * C + c) -> (c * F + f)
    perm = np.array([(j % F) * C + j // F for j in range(C * F)])
    np.testing.assert_array_equal(d1.marker_sites[perm], d0.marker_sites)
    # unbatched results (fit_offsets' single chain) are not touched by the switch
    res1 = {k: v[0] for k, v in res.items()}
    np.testing.assert_array_equal(_bare_stac(rodent_setup, rodent_cfg, reference_marker_order=True)._package_data(res1, kp[0]).marker_sites, ms[0])


def test_stac_core_q_opt_passes_per_call_bounds_and_checks_site_idxs(rodent_setup):
    """StacCore.q_opt takes lb / ub per call like the reference (stac_core.py:193-235): the engine's own box goes
    down as "no override", a different box is handed to stac_q_solve, a foreign site_idxs raises."""
    from stac_mjx_amd.stac_core import StacCore

    fs = rodent_setup
    calls = []

    class FakeEngine:
        K, nq = 23, 74
        lb, ub = fs.lb.copy(), fs.ub.copy()
        params = types.SimpleNamespace(tol=0.0, maxiter=0)

        def q_solve(self, kp, q0, qs, ks, lb=None, ub=None):
            calls.append((lb, ub))
            return torch.zeros(1, 74), torch.zeros(1, 4), torch.zeros(1, 4, dtype=torch.int32)

    core = StacCore(FakeEngine(), 1e-4, 400, site_idxs=np.arange(21, 44))
    args = (None, None, np.zeros(69, np.float32), np.ones(74, bool), np.ones(69, bool), np.zeros(74, np.float32))
    core.q_opt(*args)
    core.q_opt(*args, torch.as_tensor(fs.lb), torch.as_tensor(fs.ub), np.arange(21, 44))
    tight = np.maximum(fs.lb, -0.1)
    core.q_opt(*args, tight, fs.ub)
    assert calls[0] == (None, None) and calls[1] == (None, None)
    np.testing.assert_array_equal(calls[2][0], tight)
    with pytest.raises(ValueError, match="site_idxs"):
        core.q_opt(*args, None, None, np.arange(20, 43))
    with pytest.raises(ValueError, match="together"):
        core.q_opt(*args, tight, None)


def test_run_stac_rebinds_the_config_stored_with_the_fit(tmp_path, rodent_setup, rodent_cfg, monkeypatch):
    """main.py:111: `cfg, fit_offsets_data = io.load_stac_data(fit_offsets_path)` -- with skip_fit_offsets the fit
    file's config decides continuous / n_frames_per_clip / infer_qvels of the post-processing and is what the ik_only
    file records, while the Stac object keeps the caller's config."""
    from stac_mjx_amd import io, main
    from stac_mjx_amd.config import validate_config

    def mk(**over):
        stac = dict(fit_offsets_path="fit.h5", ik_only_path="ik.h5", data_path="d.mat", continuous=False, n_fit_frames=4,
                    skip_fit_offsets=True, skip_ik_only=False, infer_qvels=False, n_frames_per_clip=4,
                    mujoco=dict(solver="newton", iterations=1, ls_iterations=4))
        stac.update(over)
        return validate_config({"model": dict(rodent_cfg), "stac": stac})

    fit_cfg, caller_cfg = mk(infer_qvels=True, n_frames_per_clip=2), mk()
    z = lambda *s: np.zeros(s, np.float32)
    names = dict(kp_names=rodent_setup.kp_names, names_qpos=rodent_setup.part_names, names_xpos=rodent_setup.tables.body_names)
    io.save_data_to_h5(config=fit_cfg, file_path=tmp_path / "fit.h5", kp_data=z(4, 69), marker_sites=z(4, 23, 3),
                       offsets=np.full((23, 3), 0.5, np.float32), qpos=z(4, 74), xpos=z(4, 67, 3), xquat=z(4, 67, 4),
                       qvel=np.array([]), **names)
    seen = {}

    class DummyStac:
        _timestep, _freejoint = 0.002, True

        def __init__(self, xml, cfg, kp_names, **kw):
            self.cfg = cfg
            seen["stac"] = self

        def ik_only(self, kp, offsets, gather=None):
            seen["offsets"] = np.asarray(offsets)
            q = z(8, 74)
            q[:, 3] = 1.0
            return io.StacData(qpos=q, xpos=z(8, 67, 3), xquat=z(8, 67, 4), marker_sites=z(8, 23, 3), offsets=np.asarray(offsets),
                               kp_data=kp, **names)

    monkeypatch.setattr(main, "Stac", DummyStac)
    _, ik_path = main.run_stac(caller_cfg, z(8, 69), rodent_setup.kp_names, base_path=tmp_path)
    saved_cfg, ik = io.load_stac_data(ik_path)
    assert np.all(seen["offsets"] == 0.5)
    assert seen["stac"].cfg is caller_cfg                       # the Stac object keeps the caller's config
    assert saved_cfg.stac.infer_qvels is True and saved_cfg.stac.n_frames_per_clip == 2   # the file records the fit's
    assert ik.qvel.shape == (8, 73)                             # and the fit's infer_qvels = True was applied


# ---- the .h5 output contract, executed (stac_mjx/io.py:194-278) -------------------------------------------------------
_H5_PY = "/opt/conda/bin/python3.9"  # this image's interpreter that has h5py (python3.10 has none)
_H5_SCRIPT = r"""
import json, sys
sys.path.insert(0, sys.argv[1])
import numpy as np, h5py
from stac_mjx_amd import io
from stac_mjx_amd.config import validate_config
assert io.h5py is not None
mcfg = json.load(open(sys.argv[1] + "/tests/golden/rodent_model_cfg.json"))
cfg = validate_config({"model": mcfg, "stac": dict(fit_offsets_path="fit.h5", ik_only_path="ik.h5", data_path="d.mat",
    continuous=False, n_fit_frames=3, skip_fit_offsets=False, skip_ik_only=False, infer_qvels=True, n_frames_per_clip=3,
    mujoco=dict(solver="newton", iterations=1, ls_iterations=4))})
rng = np.random.default_rng(0)
f32 = lambda *s: rng.normal(size=s).astype(np.float32)
names = dict(kp_names=list(mcfg["KEYPOINT_MODEL_PAIRS"].keys()), names_qpos=["root"] * 7 + ["j%d" % i for i in range(67)],
             names_xpos=["b%d" % i for i in range(67)])
data = dict(kp_data=f32(3, 69), marker_sites=f32(3, 23, 3), offsets=f32(23, 3), qpos=f32(3, 74), xpos=f32(3, 67, 3),
            xquat=f32(3, 67, 4))
out = {}
for tag, qvel in (("fit", np.array([])), ("ik", f32(3, 73))):
    path = io.save_data_to_h5(config=cfg, file_path=sys.argv[2] + "/" + tag + ".h5", qvel=qvel, **names, **data)
    assert str(path).endswith(".h5") and io.resolve_output_path(path) == path
    with h5py.File(path, "r") as f:
        out[tag] = {k: [str(f[k].dtype), list(f[k].shape), f[k].compression] for k in f.keys()}
        cfg_yaml = f["config"][()].decode("utf-8")
    cfg2, back = io.load_stac_data(path)
    assert cfg2.stac.n_frames_per_clip == 3 and cfg2.model.N_ITERS == cfg.model.N_ITERS and "MJCF_PATH" in cfg_yaml
    for k, v in data.items():
        assert np.array_equal(getattr(back, k), v), k
    assert back.kp_names == names["kp_names"] and back.names_qpos == names["names_qpos"] and back.names_xpos == names["names_xpos"]
    assert np.array_equal(back.qvel, qvel)
print(json.dumps(out))
"""


@pytest.mark.skipif(not __import__("os").path.exists(_H5_PY), reason="no interpreter with h5py in this image")
def test_h5_writer_and_reader_executed_with_h5py(tmp_path):
    """N1: the h5py branch of io.save_data_to_h5 / load_stac_data really runs (under the image's python3.9, which has
    h5py 3.3): dataset names, string dtypes and gzip compression are the reference's (io.py:224-236), and the file
    round-trips, also with the empty qvel a fit_offsets file carries."""
    import subprocess

    r = subprocess.run([_H5_PY, "-c", _H5_SCRIPT, str(ROOT), str(tmp_path)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-2000:]
    out = __import__("json").loads(r.stdout.strip().splitlines()[-1])
    want = {"config", "kp_names", "names_qpos", "names_xpos", "kp_data", "marker_sites", "offsets", "qpos", "qvel", "xpos", "xquat"}
    for tag in ("fit", "ik"):
        d = out[tag]
        assert set(d) == want, set(d) ^ want
        assert d["config"][0].startswith("|S") and d["config"][1] == []          # np.bytes_ scalar holding the YAML
        for k in ("kp_names", "names_qpos", "names_xpos"):
            assert d[k][0].startswith("|S") and len(d[k][1]) == 1               # fixed-width byte strings
        for k in ("kp_data", "marker_sites", "offsets", "qpos", "xpos", "xquat"):
            assert d[k][0] == "float32" and d[k][2] == "gzip", (k, d[k])
    assert out["ik"]["qvel"] == ["float32", [3, 73], "gzip"] and out["fit"]["qvel"][1] == [0]


_H5_PAR_SCRIPT = r"""
import sys, time
sys.path.insert(0, sys.argv[1])
import numpy as np, h5py
from stac_mjx_amd import io
rng = np.random.default_rng(1)
N = 4099  # (not a multiple of any chunk length: the last chunk of every dataset is ragged)
f32 = lambda *s: rng.normal(size=s).astype(np.float32)
data = dict(kp_data=f32(N, 69), marker_sites=f32(N, 23, 3), offsets=f32(23, 3), qpos=f32(N, 74), xpos=f32(N, 67, 3), xquat=f32(N, 67, 4))
io._PAR_MIN_BYTES = 1 << 16
path = io.save_data_to_h5(config={"a": 1}, file_path=sys.argv[2] + "/par.h5", qvel=f32(N, 73), kp_names=["k"], names_qpos=["q"],
                          names_xpos=["x"], **data)
with h5py.File(path, "r") as f:
    for k, v in data.items():
        d = f[k]
        assert d.compression == "gzip" and d.dtype == np.float32 and d.shape == v.shape, k
        assert np.array_equal(d[()], v), k                      # HDF5's own gzip filter inflates what the pool deflated
        assert np.array_equal(d[N - 5:], v[N - 5:]) and np.array_equal(d[1000:1003], v[1000:1003])
    assert f["xquat"].chunks[1:] == (67, 4) and f["xquat"].chunks[0] < N   # really chunked by rows, several chunks
print("ok")
"""


@pytest.mark.skipif(not __import__("os").path.exists(_H5_PY), reason="no interpreter with h5py in this image")
def test_h5_parallel_deflate_chunks_are_read_back_by_hdf5(tmp_path):
    """the large datasets of a result file are deflated chunk by chunk on a thread pool and handed to HDF5 with
    write_direct_chunk; HDF5's own filter pipeline must read them back bit for bit (ragged last chunk included)."""
    import subprocess

    r = subprocess.run([_H5_PY, "-c", _H5_PAR_SCRIPT, str(ROOT), str(tmp_path)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stderr[-2000:]


def test_npz_stand_in_is_written_by_parallel_deflate_and_read_by_numpy(tmp_path, monkeypatch):
    """The .npz stand-in (interpreters without h5py): members are ZIP_DEFLATED streams put together from independently
    deflated blocks; numpy and zipfile must read them like any savez_compressed file -- several blocks per member, an
    empty array, a 0-d bytes scalar."""
    import zipfile

    from stac_mjx_amd import io

    monkeypatch.setattr(io, "h5py", None)
    monkeypatch.setattr(io, "_NPZ_BLOCK", 1 << 14)
    rng = np.random.default_rng(2)
    f32 = lambda *s: rng.normal(size=s).astype(np.float32)  # noqa: E731
    data = dict(kp_data=f32(301, 69), marker_sites=f32(301, 23, 3), offsets=f32(23, 3), qpos=f32(301, 74), xpos=f32(301, 67, 3),
                xquat=f32(301, 67, 4))
    out = io.save_data_to_h5(config={"a": 1}, file_path=tmp_path / "r.h5", qvel=np.array([]), kp_names=["k1", "k2"], names_qpos=["q"],
                             names_xpos=["x"], **data)
    assert out.suffix == ".npz" and zipfile.ZipFile(out).testzip() is None
    with np.load(out, allow_pickle=False) as f:
        for k, v in data.items():
            assert np.array_equal(f[k], v), k
        assert f["qvel"].shape == (0,) and bytes(f["config"]) == b"a: 1\n" and f["kp_names"].tolist() == [b"k1", b"k2"]


def test_crc32_of_a_member_from_its_blocks_and_the_writer_threads(monkeypatch):
    """Round 6: every deflate block of the .npz stand-in carries its own CRC-32 and a member's is combined from them (a GF(2) shift
    matrix per block length, cached): equal to zlib's CRC of the whole; the ranks of a node divide its cores among themselves; the
    job window hands results back in order with a bounded number in flight."""
    import zlib
    from concurrent.futures import ThreadPoolExecutor

    from stac_mjx_amd import io

    rng = np.random.default_rng(5)
    for n1, n2 in [(0, 5), (5, 0), (1, 1), (1000, 3), (12345, 1 << 16), (1 << 16, 70001), (3, 1 << 16)]:
        a, b = rng.bytes(n1), rng.bytes(n2)
        assert io._crc32_combine(zlib.crc32(a), zlib.crc32(b), len(b)) == zlib.crc32(a + b), (n1, n2)
    blocks = [rng.bytes(n) for n in (4096, 4096, 4096, 17)]
    crc = zlib.crc32(blocks[0])
    for blk in blocks[1:]:
        crc = io._crc32_combine(crc, zlib.crc32(blk), len(blk))
    assert crc == zlib.crc32(b"".join(blocks)) and 4096 in io._CRC_SHIFT
    monkeypatch.setenv("LOCAL_WORLD_SIZE", "8")
    n8 = io._n_threads()
    monkeypatch.setenv("LOCAL_WORLD_SIZE", "1")
    assert 1 <= n8 <= max(io._n_threads() // 8, 1) and io._n_threads() <= 64
    with ThreadPoolExecutor(4) as pool:
        assert list(io._map_window(pool, lambda i: i * i, range(50), window=3)) == [i * i for i in range(50)]


# ---- library loading guards and the bench launcher (no GPU needed) -------------------------------------------------------
def test_load_library_refuses_a_stale_or_wrong_abi_library(monkeypatch):
    """The .so is git-ignored and travels apart from the sources: a library built from other sources, or one that
    exports another ABI version, must raise instead of being called with today's argument layout."""
    from stac_mjx_amd import engine

    engine.load_library()  # the in-tree build loads
    monkeypatch.setattr(engine, "_LIB", None)
    monkeypatch.setattr(engine, "source_digest", lambda: "0" * 64)
    with pytest.raises(engine.StacHipError, match="other sources"):
        engine.load_library()
    monkeypatch.undo()
    monkeypatch.setattr(engine, "_LIB", None)
    monkeypatch.setattr(engine, "ABI_VERSION", 99)
    with pytest.raises(engine.StacHipError, match="ABI version"):
        engine.load_library()


def test_bench_gpus_n_launches_itself_or_says_why_not():
    """`python bench.py --gpus 2` outside torchrun (how the driver calls it): the parent starts the ranks as child
    processes before touching the GPU; with fewer GPUs than asked it says so and exits non-zero without hanging."""
    import subprocess
    import sys

    from conftest import ROOT

    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2"], capture_output=True, text=True, timeout=300, env=env)
    import torch

    if torch.cuda.device_count() < 2:
        assert r.returncode == 2 and "GPU" in r.stderr and "visible" in r.stderr and r.stdout.strip() == ""
    env["WORLD_SIZE"] = "1"  # a launcher that disagrees with --gpus is an error too, not a silent 1-GPU run
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2"], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode != 0 and "WORLD_SIZE=1" in r.stderr


# ---- the .nwb / .h5 INPUT loaders, executed on the reference's own data (stac_mjx/io.py:127-170) ------------------------
_LOADER_SCRIPT = r"""
import json, sys
sys.path.insert(0, sys.argv[1])
import numpy as np
from stac_mjx_amd import io
from stac_mjx_amd.config import validate_config
ref = sys.argv[2]
def cfg_for(model_json, data_path):
    m = json.load(open(sys.argv[1] + "/tests/golden/" + model_json))
    return m, validate_config({"model": m, "stac": dict(fit_offsets_path="fit.h5", ik_only_path="ik.h5", data_path=data_path,
        continuous=False, n_fit_frames=3, skip_fit_offsets=False, skip_ik_only=False, infer_qvels=True, n_frames_per_clip=3,
        mujoco=dict(solver="newton", iterations=1, ls_iterations=4))})
m, c = cfg_for("rodent_model_cfg.json", "tests/data/test_rodent_mocap_1000_frames.nwb")
nwb, names_nwb = io.load_data(c, ref)
_, c = cfg_for("rodent_model_cfg.json", "tests/data/test_rodent_mocap_1000_frames.mat")
mat, names_mat = io.load_data(c, ref)
mm, c = cfg_for("mouse_model_cfg.json", "tests/data/test_mouse_mocap_3600_frames.h5")
h5, names_h5 = io.load_data(c, ref)
golden = np.load(sys.argv[1] + "/tests/golden/rodent_mocap_1000.npy")
mouse_golden = np.load(sys.argv[1] + "/tests/golden/mouse_mocap_200.npy")
print(json.dumps(dict(
    nwb_shape=list(nwb.shape), mat_shape=list(mat.shape), h5_shape=list(h5.shape), nwb_names=names_nwb, h5_names=names_h5,
    rodent_pairs=list(m["KEYPOINT_MODEL_PAIRS"].keys()), mouse_pairs=list(mm["KEYPOINT_MODEL_PAIRS"].keys()),
    nwb_equals_mat=bool(np.array_equal(nwb, mat)), mat_equals_fixture=bool(np.array_equal(mat, golden)),
    h5_equals_fixture=bool(np.array_equal(h5[:200], mouse_golden)), dtypes=[str(nwb.dtype), str(h5.dtype)],
    h5_finite=bool(np.isfinite(h5).all()))))
"""


@pytest.mark.skipif(not os.path.exists(_H5_PY) or not REFERENCE.exists(),
                    reason="needs the image's h5py interpreter and the reference's test data (build container only)")
def test_nwb_and_h5_input_loaders_on_the_reference_data(tmp_path):
    """N1: `load_nwb` / `load_h5` through `load_data`, on tests/data/test_rodent_mocap_1000_frames.nwb and
    test_mouse_mocap_3600_frames.h5, with the expectations of the reference's tests/test_io.py:93-173: shapes
    (1000, 69) and (3600, 102), keypoint order == KEYPOINT_MODEL_PAIRS key order; and the .nwb result equals the
    .mat one, which equals the committed fixture."""
    import json
    import subprocess

    r = subprocess.run([_H5_PY, "-c", _LOADER_SCRIPT, str(ROOT), str(REFERENCE)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads(r.stdout.strip().splitlines()[-1])
    assert d["nwb_shape"] == [1000, 69] and d["mat_shape"] == [1000, 69] and d["h5_shape"] == [3600, 102]
    assert d["nwb_names"] == d["rodent_pairs"] and d["nwb_names"][:5] == ["AnkleL", "AnkleR", "EarL", "EarR", "ElbowL"]
    assert d["h5_names"] == d["mouse_pairs"] and d["h5_names"][:5] == ["Nose", "Ear_R", "Ear_L", "TTI", "Head"]
    assert d["nwb_equals_mat"] and d["mat_equals_fixture"] and d["h5_equals_fixture"] and d["h5_finite"]
    assert d["dtypes"] == ["float32", "float32"]


# ---- launcher: counting GPUs without touching HIP (bench.py --gpus N) -------------------------------------------
def _fake_kfd(tmp_path, simd_counts):
    for i, n in enumerate(simd_counts):
        d = tmp_path / str(i)
        d.mkdir()
        (d / "properties").write_text(f"cpu_cores_count {0 if n else 16}\nsimd_count {n}\nmem_banks_count 1\n")
    return str(tmp_path)


def test_visible_gpu_count_reads_the_kfd_topology_and_the_visibility_variables(tmp_path):
    from stac_mjx_amd.dist import visible_gpu_count

    root = _fake_kfd(tmp_path, [0, 0, 1024, 1024, 1024, 1024])  # two CPU nodes, four GPUs
    assert visible_gpu_count(root, {}) == 4
    assert visible_gpu_count(root, {"HIP_VISIBLE_DEVICES": "0,2"}) == 2
    assert visible_gpu_count(root, {"ROCR_VISIBLE_DEVICES": "1", "HIP_VISIBLE_DEVICES": "0,1"}) == 1   # the second filters the first
    assert visible_gpu_count(root, {"CUDA_VISIBLE_DEVICES": ""}) == 0
    assert visible_gpu_count(root, {"HIP_VISIBLE_DEVICES": "0,7,1"}) == 1                          # stops at the invalid index
    assert visible_gpu_count(root, {"ROCR_VISIBLE_DEVICES": "GPU-abcdef0123456789,GPU-0123"}) == 2  # UUIDs at face value
    assert visible_gpu_count(str(tmp_path / "missing"), {}) == 0


def test_bench_gpus_2_without_two_gpus_exits_2_and_never_imports_hip_state(tmp_path):
    """`python bench.py --gpus 2` on a box with fewer GPUs: exit code 2, a clear message, nothing launched -- decided from
    sysfs, so the parent has not initialised any GPU runtime on the way."""
    import subprocess
    import sys as _sys

    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    env["HIP_VISIBLE_DEVICES"] = "0"  # at most one GPU visible whatever the box holds: the refusal path, deterministically
    r = subprocess.run([_sys.executable, str(ROOT / "bench.py"), "--gpus", "2"], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 2 and "only" in r.stderr and "visible" in r.stderr and r.stdout.strip() == ""
    src = (ROOT / "bench.py").read_text()
    launch = src[src.index("def self_launch"):src.index("def lib_digest")]
    assert "torch.cuda" not in launch and "device_count" not in launch
