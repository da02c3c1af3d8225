"""BASELINE.json configs at (per-GPU) full size on the HIP path: properties that do not need the oracle at that size
(determinism, box constraints, kinematic consistency, fit quality) plus sampled clips / frames bit-equal to the oracle."""

import json

import numpy as np
import pytest
import torch

from conftest import GOLDEN
from helpers import marker_error_mm, oracle_fit_offsets
from test_gpu_stac import _cfg

pytestmark = pytest.mark.gpu


def _oracle(fs, **kw):
    from oracle import Oracle

    return Oracle(fs.tables, **kw)


def test_fit_offsets_frame_sampling_path_bit_exact(rodent_setup, rodent_cfg, rodent_mocap):
    """n_fit_frames > N_SAMPLE_FRAMES: the offset phase uses the PRNGKey(0) permutation sample (compute_stac.py:136-140)
    -- 100 of 120 frames here -- on the GPU exactly as in the oracle-driven restatement."""
    from stac_mjx_amd.prng import sample_time_indices
    from stac_mjx_amd.stac import Stac

    kp = rodent_mocap[:120]
    cfg = _cfg(rodent_cfg, n_fit_frames=120)
    cfg.model.N_ITERS = 1
    idx = sample_time_indices(120, int(rodent_cfg["N_SAMPLE_FRAMES"]))
    assert len(idx) == 100 and len(set(idx.tolist())) == 100 and idx.min() >= 0 and idx.max() < 120 and not np.array_equal(idx, np.arange(100))
    data = Stac(None, cfg, rodent_setup.kp_names, setup=rodent_setup, verbose=False).fit_offsets(kp)
    ref_off, ref = oracle_fit_offsets(rodent_setup, rodent_cfg, kp, 1)
    np.testing.assert_array_equal(data.offsets, ref_off)
    np.testing.assert_array_equal(data.qpos, ref["qpos"])


def test_config3_full_fit_then_100k_ik_only(rodent_setup, rodent_cfg, rodent_mocap):
    """BASELINE configs[2]: fit_offsets on the real 1000-frame recording (N_ITERS = 6, one warm-started chain, the
    frame sample of the offset phase drawn from 1000 frames), then ik_only of 100 000 frames in clips of 250."""
    from stac_mjx_amd.stac import Stac
    from stac_mjx_amd.synth import synth_keypoints

    fs = rodent_setup
    cfg = _cfg(rodent_cfg, n_fit_frames=1000, n_frames_per_clip=250)
    assert int(cfg.model.N_ITERS) == 6
    stac = Stac(None, cfg, fs.kp_names, setup=fs, verbose=False)
    fit = stac.fit_offsets(rodent_mocap)
    assert fit.qpos.shape == (1000, 74) and np.isfinite(fit.qpos).all() and np.isfinite(fit.offsets).all()
    # the calibration moved the offsets and fits the recording to 1-2 mm (the reference's legacy fit: 0.92 mm on 50 frames)
    moved = np.linalg.norm(fit.offsets - fs.tables.site_pos, axis=-1) * 1e3
    assert moved.max() > 5.0
    assert marker_error_mm(fit.marker_sites, rodent_mocap) < 1.8
    # box constraints hold wherever they are finite; marker-less subtrees never move
    fin = np.isfinite(fs.lb) & np.isfinite(fs.ub)
    assert (fit.qpos[:, fin] >= fs.lb[fin] - 1e-6).all() and (fit.qpos[:, fin] <= fs.ub[fin] + 1e-6).all()
    # kinematic consistency, bit for bit: the recorded markers / bodies are FK(qpos) under the fitted offsets
    orc = _oracle(fs, tol=1e-4, maxiter=400)
    orc.set_site_pos(fit.offsets)
    for t in (0, 123, 999):
        r = orc.fk(fit.qpos[t])
        np.testing.assert_array_equal(fit.marker_sites[t], r["site_xpos"])
        np.testing.assert_array_equal(fit.xpos[t], r["xpos"])
    # the last pose pass is a fixed point of re-running the chain's LAST frame from its own answer's predecessor
    last = orc.pose_optimization(rodent_mocap[999:1000], fit.qpos[998], fs.lb, fs.ub, fs.part_masks)
    np.testing.assert_array_equal(last["qpos"][0], fit.qpos[999])

    # ---- 100 000 frames of ik_only at the reference's default chaining ------------------------------------------------
    fk = lambda q: stac.engine.fk(q, want=("site_xpos",))["site_xpos"].cpu().numpy()
    kp_long, _ = synth_keypoints(fs, fk, 400, 250, seed=7, noise_seed=8)
    kp_long = kp_long.reshape(-1, 69)
    ik = stac.ik_only(kp_long, fit.offsets)
    assert ik.qpos.shape == (100000, 74) and np.isfinite(ik.qpos).all()
    assert (ik.qpos[:, fin] >= fs.lb[fin] - 1e-6).all() and (ik.qpos[:, fin] <= fs.ub[fin] + 1e-6).all()
    assert marker_error_mm(ik.marker_sites, kp_long) < 2.0  # 1 mm keypoint noise
    np.testing.assert_array_equal(ik.kp_data, kp_long)
    # three sampled clips, bit for bit against the oracle (root optimisation + 250 warm-started frames each)
    clips = kp_long.reshape(400, 250, 69)
    sel = [0, 137, 399]
    ref = orc.ik_clips(clips[sel], fs.lb, fs.ub, fs.part_masks, fs.trunk_kps, fs.root_kp_idx, fs.root_dims)
    got = ik.qpos.reshape(400, 250, 74)
    for i, c in enumerate(sel):
        np.testing.assert_array_equal(got[c], ref["qpos"][i])
        np.testing.assert_array_equal(ik.marker_sites.reshape(400, 250, 23, 3)[c], ref["marker_sites"][i])


def test_config5_fruitfly_one_gpu_share(fly_setup):
    """BASELINE configs[4]: the fruit fly (fruitfly_force_free.xml, tethered config: nq = 43, K = 30, oriented bodies, no
    root optimisation), one GPU's share of the 200 000 frames = 25 000 frames in clips of 250."""
    from stac_mjx_amd.engine import Engine
    from stac_mjx_amd.synth import synth_keypoints, synth_offsets

    fs = fly_setup
    with open(GOLDEN / "fly_model_cfg.json") as fh:
        mcfg = json.load(fh)
    tol, maxiter = float(mcfg["FTOL"]), int(mcfg["N_ITER_Q"])
    eng = Engine(fs.tables, fs.lb, fs.ub, tol=tol, maxiter=maxiter)
    off = synth_offsets(fs)
    eng.set_site_pos(off)
    fk = lambda q: eng.fk(q, want=("site_xpos",))["site_xpos"].cpu().numpy()
    kp, _ = synth_keypoints(fs, fk, 100, 250, seed=11, noise_seed=12)
    args = dict(part_masks=fs.part_masks, trunk_kps=fs.trunk_kps, root_kp_idx=max(fs.root_kp_idx, 0), root_dims=fs.root_dims,
                do_root_opt=fs.do_root_opt)
    res = eng.q_phase(kp, **args)
    res2 = eng.q_phase(kp, **args)
    q = res["qpos"].cpu().numpy()
    assert q.shape == (100, 250, 43) and np.isfinite(q).all()
    for k in ("qpos", "frame_error", "counters", "marker_sites"):
        assert torch.equal(res[k], res2[k]), k  # run-to-run determinism
    fin = np.isfinite(fs.lb) & np.isfinite(fs.ub)
    assert (q[..., fin] >= fs.lb[fin] - 1e-6).all() and (q[..., fin] <= fs.ub[fin] + 1e-6).all()
    # the fly model is ~0.5 length units long; FTOL = 5e-3 stops the solves early: residual 6e-3 units (noise 1e-3)
    assert marker_error_mm(res["marker_sites"].cpu().numpy(), kp) < 8.0
    orc = _oracle(fs, tol=tol, maxiter=maxiter)
    orc.set_site_pos(off)
    sel = [0, 57]
    ref = orc.ik_clips(kp[sel], fs.lb, fs.ub, fs.part_masks, fs.trunk_kps, max(fs.root_kp_idx, 0), fs.root_dims,
                       do_root_opt=fs.do_root_opt)
    for i, c in enumerate(sel):
        np.testing.assert_array_equal(q[c], ref["qpos"][i])
        np.testing.assert_array_equal(res["xquat"][c].cpu().numpy(), ref["xquat"][i])
        np.testing.assert_array_equal(res["counters"][c].cpu().numpy().astype(np.uint32), ref["counters"][i])


def test_config4_one_gpu_share_125k_frames(rodent_setup, rodent_cfg):
    """BASELINE configs[3] (1 M frames of ik_only over 8 GPUs at the reference's default chaining): ONE GPU's share,
    125 000 frames = 500 clips x 250 frames, at full size on the HIP path -- run-to-run determinism, box constraints,
    kinematic consistency of the recorded markers, fit quality, and three sampled clips bit-equal to the oracle."""
    from stac_mjx_amd.engine import Engine
    from stac_mjx_amd.synth import synth_keypoints, synth_offsets

    fs = rodent_setup
    tol, maxiter = float(rodent_cfg["FTOL"]), int(rodent_cfg["N_ITER_Q"])
    eng = Engine(fs.tables, fs.lb, fs.ub, tol=tol, maxiter=maxiter)
    off = synth_offsets(fs)
    eng.set_site_pos(off)
    fk = lambda q: eng.fk(q, want=("site_xpos",))["site_xpos"].cpu().numpy()
    kp, _ = synth_keypoints(fs, fk, 500, 250, seed=21, noise_seed=22)
    kp_d = torch.as_tensor(kp).to(eng.device)
    args = dict(part_masks=fs.part_masks, trunk_kps=fs.trunk_kps, root_kp_idx=fs.root_kp_idx, root_dims=fs.root_dims,
                do_root_opt=fs.do_root_opt, want_bodies=False)
    res = eng.q_phase(kp_d, **args)
    q = res["qpos"].cpu().numpy()
    assert q.shape == (500, 250, 74) and np.isfinite(q).all()
    res2 = eng.q_phase(kp_d, **args)
    for k in ("qpos", "frame_error", "counters", "marker_sites", "carry_qpos"):
        assert torch.equal(res[k], res2[k]), k
    fin = np.isfinite(fs.lb) & np.isfinite(fs.ub)
    assert (q[..., fin] >= fs.lb[fin] - 1e-6).all() and (q[..., fin] <= fs.ub[fin] + 1e-6).all()
    np.testing.assert_allclose(np.linalg.norm(q[..., 3:7], axis=-1), 1.0, atol=1e-5)  # written-back unit quaternions
    ms = res["marker_sites"].cpu().numpy()
    assert marker_error_mm(ms, kp) < 2.0  # 1 mm keypoint noise
    assert (res["frame_error"].cpu().numpy() > 0).all() and (res["counters"][..., 3].cpu().numpy() == 6).all()
    # recorded markers are FK(recorded qpos) on a strided sample of all 125 000 poses (the public stac_fk normalises the
    # stored unit quaternion once more, which may move the last bit: 1e-6 m, not tolerance 0; the sampled clips below
    # are compared with the oracle bit for bit)
    flat_q = res["qpos"].reshape(-1, 74)[::997]
    again = eng.fk(flat_q, want=("site_xpos",))["site_xpos"]
    assert torch.allclose(again, res["marker_sites"].reshape(-1, 23, 3)[::997], rtol=0, atol=1e-6)
    orc = _oracle(fs, tol=tol, maxiter=maxiter)
    orc.set_site_pos(off)
    sel = [0, 251, 499]
    ref = orc.ik_clips(kp[sel], fs.lb, fs.ub, fs.part_masks, fs.trunk_kps, fs.root_kp_idx, fs.root_dims, want_bodies=False)
    for i, c in enumerate(sel):
        np.testing.assert_array_equal(q[c], ref["qpos"][i])
        np.testing.assert_array_equal(ms[c], ref["marker_sites"][i])
        np.testing.assert_array_equal(res["frame_error"][c].cpu().numpy(), ref["frame_error"][i])
        np.testing.assert_array_equal(res["counters"][c].cpu().numpy().astype(np.uint32), ref["counters"][i])


def test_mouse_real_recording_full_solves(mouse_setup):
    """The mouse model (nq = 230, 85 tree levels, K = 34) on REAL frames (the first 200 of the reference's
    tests/data/test_mouse_mocap_3600_frames.h5, tests/golden/mouse_mocap_200.npy) at the config's own N_ITER_Q = 400 and
    FTOL with root optimisation: 8 clips x 2 frames against the oracle at tolerance 0 at three lane widths, then a
    2 000-frame property run (the 200 real frames, ten times with 0.2 mm of seeded jitter)."""
    from stac_mjx_amd.engine import Engine

    fs = mouse_setup
    with open(GOLDEN / "mouse_model_cfg.json") as fh:
        mcfg = json.load(fh)
    tol, maxiter = float(mcfg["FTOL"]), int(mcfg["N_ITER_Q"])
    assert maxiter == 400 and fs.do_root_opt
    real = np.load(GOLDEN / "mouse_mocap_200.npy")
    assert real.shape == (200, 102)
    kp = real[::12][:16].reshape(8, 2, 102)  # 8 clips of 2 frames, spread over the recording
    orc = _oracle(fs, tol=tol, maxiter=maxiter)
    ref = orc.ik_clips(kp, fs.lb, fs.ub, fs.part_masks, fs.trunk_kps, fs.root_kp_idx, fs.root_dims, do_root_opt=True)
    assert ref["counters"][..., 0].max() >= 400  # real data: the full-body solves run to the iteration cap
    args = dict(part_masks=fs.part_masks, trunk_kps=fs.trunk_kps, root_kp_idx=fs.root_kp_idx, root_dims=fs.root_dims,
                do_root_opt=True)
    for lanes in (0, 16, 64):
        eng = Engine(fs.tables, fs.lb, fs.ub, tol=tol, maxiter=maxiter, lanes_per_chain=lanes)
        res = eng.q_phase(kp, **args)
        np.testing.assert_array_equal(res["qpos"].cpu().numpy(), ref["qpos"])
        np.testing.assert_array_equal(res["marker_sites"].cpu().numpy(), ref["marker_sites"])
        np.testing.assert_array_equal(res["xpos"].cpu().numpy(), ref["xpos"])
        np.testing.assert_array_equal(res["frame_error"].cpu().numpy(), ref["frame_error"])
        np.testing.assert_array_equal(res["counters"].cpu().numpy().astype(np.uint32), ref["counters"])
        eng.close()
    # ---- 2 000 frames, one frame per clip: properties at a size the oracle would need minutes for ----------------------
    rng = np.random.default_rng(5)
    big = np.concatenate([real + (rng.normal(0, 2e-4, real.shape).astype(np.float32) if r else 0) for r in range(10)])
    big = big.reshape(2000, 1, 102).astype(np.float32)
    eng = Engine(fs.tables, fs.lb, fs.ub, tol=tol, maxiter=maxiter)
    res = eng.q_phase(big, **args)
    res2 = eng.q_phase(big, **args)
    q = res["qpos"].cpu().numpy()
    assert np.isfinite(q).all()
    for k in ("qpos", "frame_error", "counters", "marker_sites"):
        assert torch.equal(res[k], res2[k]), k
    fin = np.isfinite(fs.lb) & np.isfinite(fs.ub)
    assert (q[..., fin] >= fs.lb[fin] - 1e-6).all() and (q[..., fin] <= fs.ub[fin] + 1e-6).all()
    # the un-jittered first 200 clips are the real frames: those that were also in the 8 x 2 sample start identically
    np.testing.assert_array_equal(q[0, 0], ref["qpos"][0, 0])
    again = eng.fk(res["qpos"].reshape(-1, fs.tables.nq), want=("site_xpos",))["site_xpos"]
    assert torch.allclose(again, res["marker_sites"].reshape(-1, fs.tables.nsite, 3), rtol=0, atol=1e-6)  # see config 4
    # the fit reduces the marker error of the root-only pose by a wide margin on every frame
    e_fit = np.linalg.norm(res["marker_sites"].cpu().numpy() - big.reshape(2000, 1, 34, 3), axis=-1).mean(axis=(1, 2))
    assert np.median(e_fit) < 0.03, np.median(e_fit)  # the oracle reaches 0.017 on the sampled real frames
