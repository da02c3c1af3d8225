"""The reference-shaped host API (Stac / StacCore / run_stac) on the GPU vs oracle-driven restatements."""

import numpy as np
import pytest
import torch

from conftest import GOLDEN, ROOT
from helpers import oracle_fit_offsets as _oracle_fit_offsets

pytestmark = pytest.mark.gpu


def _cfg(rodent_cfg, **stac_over):
    from stac_mjx_amd.config import validate_config

    stac = dict(fit_offsets_path="fit.h5", ik_only_path="ik.h5", data_path="d.mat", continuous=False, n_fit_frames=10,
                skip_fit_offsets=False, skip_ik_only=False, infer_qvels=False, n_frames_per_clip=2,
                mujoco=dict(solver="newton", iterations=1, ls_iterations=4))
    stac.update(stac_over)
    return validate_config({"model": dict(rodent_cfg), "stac": stac})


def test_fit_offsets_config1_bit_exact(rodent_setup, rodent_cfg, rodent_mocap):
    """BASELINE config 1: real mocap, n_fit_frames=10, offset/pose alternation."""
    from stac_mjx_amd.stac import Stac

    cfg = _cfg(rodent_cfg)
    assert int(cfg.model.N_ITERS) == 6  # configs/model/rodent.yaml: the real alternation count
    kp = rodent_mocap[:10]
    stac = Stac(None, cfg, rodent_setup.kp_names, setup=rodent_setup, verbose=False)
    data = stac.fit_offsets(kp)
    ref_off, ref = _oracle_fit_offsets(rodent_setup, rodent_cfg, kp, 6)
    np.testing.assert_array_equal(data.offsets, ref_off)
    np.testing.assert_array_equal(data.qpos, ref["qpos"])
    np.testing.assert_array_equal(data.marker_sites, ref["marker_sites"])
    np.testing.assert_array_equal(data.xquat, ref["xquat"])
    assert np.abs(data.offsets - ref_off).max() <= 1e-4 and np.abs(data.qpos - ref["qpos"]).max() <= 1e-4
    assert data.qpos.shape == (10, 74) and data.xpos.shape == (10, 67, 3) and data.kp_data.shape == (10, 69)
    assert data.names_qpos[0] == "root" and data.names_xpos[1] == "walker"
    # offsets moved, and the fit improved over the initial offsets
    assert np.abs(data.offsets - rodent_setup.tables.site_pos).max() > 1e-3
    err = np.linalg.norm(data.marker_sites - kp.reshape(10, 23, 3), axis=-1).mean()
    assert err < 6e-3


def test_ik_only_clip_major_packing(rodent_setup, rodent_cfg, rodent_mocap):
    from oracle import Oracle
    from stac_mjx_amd.stac import Stac

    cfg = _cfg(rodent_cfg, n_frames_per_clip=2)
    kp = rodent_mocap[300:310]
    off = rodent_setup.tables.site_pos + 0.002
    stac = Stac(None, cfg, rodent_setup.kp_names, setup=rodent_setup, verbose=False)
    data = stac.ik_only(kp, off)
    orc = Oracle(rodent_setup.tables, tol=1e-4, maxiter=400)
    orc.set_site_pos(off)
    fs = rodent_setup
    ref = orc.ik_clips(kp.reshape(5, 2, 69), fs.lb, fs.ub, fs.part_masks, fs.trunk_kps, fs.root_kp_idx, fs.root_dims)
    np.testing.assert_array_equal(data.qpos, ref["qpos"].reshape(10, 74))
    np.testing.assert_array_equal(data.marker_sites, ref["marker_sites"].reshape(10, 23, 3))
    np.testing.assert_array_equal(data.xpos, ref["xpos"].reshape(10, 67, 3))
    np.testing.assert_array_equal(data.offsets, off)
    np.testing.assert_array_equal(data.kp_data, kp)


def test_seam_level_drivers_equal_batched_kernel(rodent_setup, rodent_cfg, rodent_mocap):
    """compute_stac drivers over StacCore.q_opt (one launch per solve) == one stac_q_phase launch."""
    from stac_mjx_amd import compute_stac, utils
    from stac_mjx_amd.stac import Stac
    from stac_mjx_amd.stac_core import DataHandle

    cfg = _cfg(rodent_cfg)
    fs = rodent_setup
    stac = Stac(None, cfg, fs.kp_names, setup=fs, verbose=False)
    kp = torch.as_tensor(rodent_mocap[50:52]).to(stac.engine.device)
    model = stac._model_handle()
    data = utils.kinematics(model, DataHandle(qpos=torch.as_tensor(fs.tables.qpos0).to(stac.engine.device)))
    core = stac.stac_core_obj
    data = compute_stac.root_optimization(core, model, data, kp, fs.root_kp_idx, fs.lb, fs.ub, None, fs.trunk_kps)
    parts = [torch.as_tensor(m) for m in fs.part_masks]
    data, qposes, xposes, xquats, markers, _, errs = compute_stac.pose_optimization(core, model, data, kp, fs.lb, fs.ub, None, parts)
    res = stac.engine.q_phase(kp[None], part_masks=fs.part_masks, trunk_kps=fs.trunk_kps, root_kp_idx=fs.root_kp_idx,
                              root_dims=fs.root_dims, do_root_opt=True)
    assert torch.equal(qposes, res["qpos"][0])
    assert torch.equal(torch.stack(markers), res["marker_sites"][0])
    assert torch.equal(torch.stack(xquats), res["xquat"][0])
    np.testing.assert_array_equal(np.array(errs, np.float32), res["frame_error"][0].cpu().numpy())
    # offset_optimization through the seam == engine.m_opt on the sampled frames
    model2, data2, off = compute_stac.offset_optimization(core, model, data, kp, model.site_pos, qposes, 100,
                                                          torch.as_tensor(fs.is_regularized), None, 1.0)
    assert torch.equal(model2.site_pos, off) and off.shape == (23, 3)


def test_run_stac_end_to_end(tmp_path, rodent_setup, rodent_cfg, rodent_mocap):
    from stac_mjx_amd.io import load_stac_data
    from stac_mjx_amd.main import run_stac

    cfg = _cfg(rodent_cfg, n_fit_frames=4, n_frames_per_clip=2, infer_qvels=True)
    cfg.model.N_ITERS = 1
    kp = rodent_mocap[:8]
    fit_path, ik_path = run_stac(cfg, kp, rodent_setup.kp_names, base_path=tmp_path, setup=rodent_setup)
    _, fit = load_stac_data(fit_path)
    _, ik = load_stac_data(ik_path)
    assert fit.qpos.shape == (4, 74) and ik.qpos.shape == (8, 74) and ik.xquat.shape == (8, 67, 4)
    assert ik.qvel.shape == (8, 73) and np.isfinite(ik.qvel).all() and np.abs(ik.qvel[:, 6:]).max() <= 20.0
    np.testing.assert_array_equal(ik.offsets, fit.offsets)
    with pytest.raises(ValueError, match="n_frames_per_clip"):
        cfg3 = _cfg(rodent_cfg, n_fit_frames=4, n_frames_per_clip=3, skip_fit_offsets=True)
        run_stac(cfg3, kp, rodent_setup.kp_names, base_path=tmp_path, setup=rodent_setup)
    cfg2 = _cfg(rodent_cfg, n_fit_frames=4, skip_ik_only=True)
    cfg2.model.N_ITERS = 1
    assert run_stac(cfg2, kp, rodent_setup.kp_names, base_path=tmp_path, setup=rodent_setup)[1] is None


def test_lm_solver_through_the_config_surface(rodent_setup, rodent_cfg, rodent_mocap):
    """`stac.solver: lm` (engine extension): same API and outputs, better or equal marker fit than PG, far fewer
    iterations; offsets are fitted with the same closed form."""
    from stac_mjx_amd.stac import Stac

    kp = rodent_mocap[:40]
    out = {}
    for solver in ("pg", "lm"):
        cfg = _cfg(rodent_cfg, n_frames_per_clip=1, solver=solver)
        cfg.model.N_ITERS = 1
        stac = Stac(None, cfg, rodent_setup.kp_names, setup=rodent_setup, verbose=False)
        fit = stac.fit_offsets(kp[:10])
        ik = stac.ik_only(kp, fit.offsets)
        err = np.linalg.norm(ik.marker_sites - kp.reshape(40, 23, 3), axis=-1).mean()
        out[solver] = (fit, ik, err)
    assert out["lm"][1].qpos.shape == (40, 74) and np.isfinite(out["lm"][1].qpos).all()
    assert out["lm"][2] <= out["pg"][2] + 2e-4, (out["lm"][2], out["pg"][2])
    assert np.abs(out["lm"][0].offsets - out["pg"][0].offsets).max() < 2e-2  # same calibration up to solver differences


def _oracle_fit_offsets_clips(fs, cfgm, kp, n_iters, fpc):
    """Restatement of the clip-parallel calibration (stac.fit_frames_per_clip): every clip is a root-optimised chain
    carried across the iterations; the offset phase sums its statistics over the sampled frames of all clips."""
    from oracle import Oracle
    from stac_mjx_amd.prng import sample_time_indices

    orc = Oracle(fs.tables, tol=float(cfgm["FTOL"]), maxiter=int(cfgm["N_ITER_Q"]))
    C = kp.shape[0] // fpc
    clips = kp[: C * fpc].reshape(C, fpc, -1)
    offsets = fs.tables.site_pos.copy()
    idx = sample_time_indices(C * fpc, int(cfgm["N_SAMPLE_FRAMES"]))
    carry = None
    for it in range(n_iters + 1):
        out = orc.ik_clips(clips, fs.lb, fs.ub, fs.part_masks, fs.trunk_kps, fs.root_kp_idx, fs.root_dims,
                           do_root_opt=(it == 0), q_init=carry)
        carry = out["qpos"][:, -1]
        if it == n_iters:
            return offsets, out
        offsets, _ = orc.m_opt(clips.reshape(C * fpc, -1)[idx], out["qpos"].reshape(C * fpc, -1)[idx], offsets,
                               fs.is_regularized, float(cfgm["M_REG_COEF"]))
        orc.set_site_pos(offsets)


def test_fit_offsets_clip_parallel_extension(rodent_setup, rodent_cfg, rodent_mocap):
    """stac.fit_frames_per_clip (engine extension): independent clips instead of one serial chain -- same kernels,
    so it must still equal the oracle driven the same way, bit for bit."""
    from stac_mjx_amd.stac import Stac

    cfg = _cfg(rodent_cfg, fit_frames_per_clip=3)
    cfg.model.N_ITERS = 2
    kp = rodent_mocap[100:113]  # 13 frames -> 4 clips of 3, the ragged frame is dropped
    stac = Stac(None, cfg, rodent_setup.kp_names, setup=rodent_setup, verbose=False)
    data = stac.fit_offsets(kp)
    ref_off, ref = _oracle_fit_offsets_clips(rodent_setup, rodent_cfg, kp, 2, 3)
    assert data.qpos.shape == (12, 74) and data.kp_data.shape == (12, 69)
    np.testing.assert_array_equal(data.offsets, ref_off)
    np.testing.assert_array_equal(data.qpos, ref["qpos"].reshape(12, 74))
    np.testing.assert_array_equal(data.marker_sites, ref["marker_sites"].reshape(12, 23, 3))


def _two_rank_fit(rank, port, tmp, kp, fpc, tix, gather):
    import os

    import torch.distributed as dist

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE="2")
    dist.init_process_group("gloo", rank=rank, world_size=2)
    try:
        import json

        from conftest import GOLDEN as G
        from stac_mjx_amd.fit_model import finish_fit_setup
        from stac_mjx_amd.mjcf import ModelTables
        from stac_mjx_amd.stac import Stac

        mcfg = json.load(open(G / "rodent_model_cfg.json"))
        fs = finish_fit_setup(ModelTables.load(G / "rodent_tables.npz"), mcfg, list(mcfg["KEYPOINT_MODEL_PAIRS"].keys()))
        cfg = _cfg(mcfg, fit_frames_per_clip=fpc, gather=gather)
        cfg.model.N_ITERS = 2
        data = Stac(None, cfg, fs.kp_names, setup=fs, verbose=False).fit_offsets(kp, time_indices=tix)
        np.savez(f"{tmp}/rank{rank}.npz", offsets=data.offsets, qpos=data.qpos, markers=data.marker_sites, kp=data.kp_data)
    finally:
        dist.destroy_process_group()


def _spawn_two(tmp_path, kp, fpc, tix=None, gather="rank0"):
    import torch.multiprocessing as mp

    port = 29600 + (int(torch.randint(0, 300, (1,)).item()))
    mp.spawn(_two_rank_fit, args=(port, str(tmp_path), kp, fpc, tix, gather), nprocs=2, join=True)
    return np.load(tmp_path / "rank0.npz"), np.load(tmp_path / "rank1.npz")


def test_fit_offsets_clip_parallel_two_ranks_on_one_gpu(tmp_path, rodent_setup, rodent_cfg, rodent_mocap):
    """The sharded calibration with a real process group (2 ranks sharing cuda:0, gloo): clips split 3 + 2, the
    3K+2 partial sums all-reduced in fixed rank order; both ranks end with the same offsets; rank 0 packages every
    clip, rank 1 keeps its shard (stac.gather = rank0, the default); "all" replicates.  The result agrees with the
    single-process run (the offset sums associate differently: 1e-6)."""
    from stac_mjx_amd.stac import Stac

    kp = rodent_mocap[300:310]  # 5 clips of 2
    cfg = _cfg(rodent_cfg, fit_frames_per_clip=2)
    cfg.model.N_ITERS = 2
    one = Stac(None, cfg, rodent_setup.kp_names, setup=rodent_setup, verbose=False).fit_offsets(kp)
    r0, r1 = _spawn_two(tmp_path, kp, 2)
    np.testing.assert_array_equal(r0["offsets"], r1["offsets"])
    assert r0["qpos"].shape == (10, 74) and r1["qpos"].shape == (4, 74)
    for k in ("qpos", "markers", "kp"):
        np.testing.assert_array_equal(r0[k][6:], r1[k])  # rank 1 owns clips 3-4 = frames 6-9
    np.testing.assert_array_equal(r0["kp"], kp)
    assert np.abs(r0["offsets"] - one.offsets).max() < 1e-5
    assert np.abs(r0["markers"] - one.marker_sites).max() < 1e-3
    a0, a1 = _spawn_two(tmp_path, kp, 2, gather="all")
    for k in ("offsets", "qpos", "markers"):
        np.testing.assert_array_equal(a0[k], a1[k])
        np.testing.assert_array_equal(a0[k], r0[k])


def test_fit_offsets_sharded_rank_without_sampled_frames_or_clips(tmp_path, rodent_setup, rodent_cfg, rodent_mocap):
    """A rank whose shard holds no sampled frame runs the offset phase with T = 0 (zero partial sums, no kernel with
    an empty grid) and a rank that owns no clip at all launches nothing: the job neither hangs nor diverges."""
    from stac_mjx_amd.stac import Stac

    kp = rodent_mocap[300:310]
    tix = np.array([0, 1, 2, 3, 5])  # every sampled frame lives on rank 0 (clips 0-2)
    cfg = _cfg(rodent_cfg, fit_frames_per_clip=2)
    cfg.model.N_ITERS = 2
    one = Stac(None, cfg, rodent_setup.kp_names, setup=rodent_setup, verbose=False).fit_offsets(kp, time_indices=tix)
    r0, r1 = _spawn_two(tmp_path, kp, 2, tix=tix)
    np.testing.assert_array_equal(r0["offsets"], r1["offsets"])
    np.testing.assert_array_equal(r0["offsets"], one.offsets)  # rank 1 adds exact zeros: same bits as one process
    np.testing.assert_array_equal(r0["qpos"], one.qpos)
    # one clip, two ranks: rank 1 has nothing to do
    s0, s1 = _spawn_two(tmp_path, kp[:2], 2)
    one1 = Stac(None, cfg, rodent_setup.kp_names, setup=rodent_setup, verbose=False).fit_offsets(kp[:2])
    np.testing.assert_array_equal(s0["offsets"], one1.offsets)
    np.testing.assert_array_equal(s0["offsets"], s1["offsets"])
    np.testing.assert_array_equal(s0["qpos"], one1.qpos)
    assert s1["qpos"].shape == (0, 74)


def test_ik_only_continuous_clips_and_edge_effects(rodent_setup, rodent_cfg, rodent_mocap):
    """stac.continuous: windows of n_frames_per_clip + 10 frames (the last one wrap-padded, utils.py:369-382) run as
    independent chains, then the overlaps are cross-faded (utils.py:393-461).  HIP == oracle on every window; the
    stitched result equals an independent statement of the cross-fade applied to the oracle's windows."""
    from oracle import Oracle
    from stac_mjx_amd import utils
    from stac_mjx_amd.stac import Stac

    fs = rodent_setup
    n, C, OV = 12, 3, utils.CONTINUOUS_BATCH_OVERLAP
    kp = rodent_mocap[200 : 200 + n * C]
    cfg = _cfg(rodent_cfg, n_frames_per_clip=n, continuous=True)
    cfg.model.N_ITER_Q = 30
    off = fs.tables.site_pos + 0.001
    stac = Stac(None, cfg, fs.kp_names, setup=fs, verbose=False)
    data = stac.ik_only(kp, off)
    win = np.stack([kp[c * n : c * n + n + OV] for c in range(C - 1)] + [np.pad(kp[(C - 1) * n :], ((0, OV), (0, 0)), mode="wrap")])
    orc = Oracle(fs.tables, tol=float(rodent_cfg["FTOL"]), maxiter=30)
    orc.set_site_pos(off)
    ref = orc.ik_clips(win, fs.lb, fs.ub, fs.part_masks, fs.trunk_kps, fs.root_kp_idx, fs.root_dims)
    assert data.qpos.shape == (C * (n + OV), 74)
    np.testing.assert_array_equal(data.qpos, ref["qpos"].reshape(-1, 74))
    np.testing.assert_array_equal(data.marker_sites, ref["marker_sites"].reshape(-1, 23, 3))
    np.testing.assert_array_equal(data.xquat, ref["xquat"].reshape(-1, 67, 4))
    np.testing.assert_array_equal(data.kp_data, win.reshape(-1, 69))
    out = utils.handle_edge_effects(data, n)
    m = 1.0 / (1.0 + np.exp(-10.0 * (np.linspace(0.0, 1.0, OV) - 0.5)))
    for name, r in (("qpos", ref["qpos"]), ("marker_sites", ref["marker_sites"]), ("xpos", ref["xpos"]), ("kp_data", win)):
        o = getattr(out, name)
        assert o.shape[0] == n * C
        for c in range(C):
            for f in range(n):
                want = r[c, f].astype(np.float64)
                if c > 0 and f < OV:
                    want = (1.0 - m[f]) * r[c - 1, n + f] + m[f] * r[c, f]
                np.testing.assert_allclose(o[c * n + f], want, rtol=1e-6, atol=1e-7)
    # the stitched keypoints are the input again wherever both windows saw the same frames (the fade of equal values)
    np.testing.assert_allclose(out.kp_data, kp, rtol=1e-6, atol=1e-7)


def _nccl_fit_worker(rank, port, tmp, kp, fpc):
    import os

    import torch.distributed as dist

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE="2")
    torch.cuda.set_device(rank)
    dist.init_process_group("nccl", rank=rank, world_size=2, device_id=torch.device(f"cuda:{rank}"))
    try:
        import json

        from conftest import GOLDEN as G
        from stac_mjx_amd.fit_model import finish_fit_setup
        from stac_mjx_amd.mjcf import ModelTables
        from stac_mjx_amd.stac import Stac

        mcfg = json.load(open(G / "rodent_model_cfg.json"))
        fs = finish_fit_setup(ModelTables.load(G / "rodent_tables.npz"), mcfg, list(mcfg["KEYPOINT_MODEL_PAIRS"].keys()))
        cfg = _cfg(mcfg, fit_frames_per_clip=fpc, n_frames_per_clip=fpc)
        cfg.model.N_ITERS = 2
        stac = Stac(None, cfg, fs.kp_names, setup=fs, device=f"cuda:{rank}", verbose=False)
        data = stac.fit_offsets(kp)
        ik = stac.ik_only(kp, data.offsets)
        np.savez(f"{tmp}/nccl{rank}.npz", offsets=data.offsets, qpos=data.qpos, ik_qpos=ik.qpos)
    finally:
        dist.destroy_process_group()


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs: one RCCL rank per GPU")
def test_fit_and_ik_two_ranks_over_rccl(tmp_path, rodent_setup, rodent_cfg, rodent_mocap):
    """The product path on two GPUs: one process per GPU, backend "nccl" (= RCCL), the 71-float offset-phase sums
    combined on the device over xGMI, clips sharded, results gathered to rank 0.  Same answers as one process (the
    fixed-order sum associates differently from the single-process sum: 1e-5 on the offsets)."""
    import torch.multiprocessing as mp

    from stac_mjx_amd.stac import Stac

    kp = rodent_mocap[300:312]
    cfg = _cfg(rodent_cfg, fit_frames_per_clip=2, n_frames_per_clip=2)
    cfg.model.N_ITERS = 2
    one_stac = Stac(None, cfg, rodent_setup.kp_names, setup=rodent_setup, verbose=False)
    one = one_stac.fit_offsets(kp)
    port = 29700 + int(torch.randint(0, 200, (1,)).item())
    mp.spawn(_nccl_fit_worker, args=(port, str(tmp_path), kp, 2), nprocs=2, join=True)
    r0, r1 = np.load(tmp_path / "nccl0.npz"), np.load(tmp_path / "nccl1.npz")
    np.testing.assert_array_equal(r0["offsets"], r1["offsets"])
    assert r0["qpos"].shape == (12, 74) and r1["qpos"].shape == (6, 74) and r0["ik_qpos"].shape == (12, 74)
    assert np.abs(r0["offsets"] - one.offsets).max() < 1e-5
    np.testing.assert_array_equal(r0["qpos"][6:], r1["qpos"])


@pytest.mark.parametrize("args", [
    ["--steps", "1", "--warmup", "0", "--frames", "500", "--no-cpu-baseline", "--no-extras"],
    ["--steps", "1", "--warmup", "0", "--frames", "500", "--frames-per-clip", "250", "--no-cpu-baseline", "--no-extras"],
    ["--steps", "1", "--warmup", "0", "--frames", "500", "--frames-per-clip", "250", "--scaling", "strong", "--no-cpu-baseline"],
    ["--mode", "fit", "--steps", "1", "--warmup", "0", "--frames", "60", "--frames-per-clip", "10"],
    ["--mode", "run", "--steps", "1", "--warmup", "0", "--frames", "500", "--frames-per-clip", "250"],
    ["--solver", "lm", "--steps", "1", "--warmup", "0", "--frames", "200", "--no-cpu-baseline"],
    ["--model", "fly", "--steps", "1", "--warmup", "0", "--frames", "200", "--no-cpu-baseline", "--no-extras"],
], ids=["ik", "clips", "strong", "fit", "run", "lm", "fly"])
def test_every_bench_mode_prints_its_line(args):
    """Round 6 (a fit-mode line once named a field of the default mode and only failed on the GPU box, inside the round's collection):
    every mode of bench.py runs at a small size as its own process and prints ONE JSON line with the contract's keys."""
    import json
    import subprocess
    import sys

    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "1", *args], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config"):
        assert k in d, k
    assert d["value"] > 0 and d["n_gpus"] == 1 and d["config"]["workload"]
    if "--mode" not in args:
        assert d["roofline"]["frac"] > 0 and d["config"]["per_rank"][0]["clips"] > 0
        assert ("predicted_value" in d["config"])


@pytest.mark.parametrize("args,clips", [
    (["--frames", "400"], [400, 400]),
    (["--scaling", "strong", "--frames", "1000", "--frames-per-clip", "250"], [2, 2]),
    (["--scaling", "strong", "--frames", "250", "--frames-per-clip", "250"], [1, 0]),
], ids=["weak", "strong", "more-ranks-than-clips"])
def test_bench_two_ranks_on_one_gpu(args, clips):
    """The N > 1 path of bench.py -- what the driver runs on an 8-GPU node -- with two ranks on THIS box's one GPU (test hooks:
    every rank on cuda:0, gloo for the barrier and the reductions): the timing barrier, the max over ranks, `config.per_rank` (one
    all_gather_object), the offset-phase exchange with bitwise-equal offsets, a rank without clips."""
    import json
    import os
    import subprocess
    import sys

    env = dict(os.environ, STAC_BENCH_BACKEND="gloo", STAC_BENCH_SHARE_GPU="1")
    port = 29800 + int(torch.randint(0, 150, (1,)).item())
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(ROOT / "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--no-cpu-baseline", *args]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["value"] > 0 and d["config"]["collective_backend"] == "gloo"
    pr = d["config"]["per_rank"]
    assert [p["rank"] for p in pr] == [0, 1] and [p["clips"] for p in pr] == clips
    assert d["config"]["frames_total"] == sum(p["frames"] for p in pr)
    ex = d["config"]["offset_phase_exchange"]
    assert ex["world_size"] == 2 and ex["offsets_bitwise_equal_across_ranks"] is True and ex["n_floats"] == 71
    assert d["scaling"] == ("strong" if "--scaling" in args else "weak")


@pytest.mark.parametrize("args", [["--mode", "fit", "--frames", "60", "--frames-per-clip", "10"],
                                  ["--mode", "run", "--frames", "500", "--frames-per-clip", "250"]], ids=["fit", "run"])
def test_bench_fit_and_run_modes_two_ranks_on_one_gpu(args):
    """`--mode fit` (clips sharded, the 71-float all-reduce per calibration iteration) and `--mode run` (`Stac.ik_only` sharding one
    global dataset itself) with two ranks on one GPU (same hooks): one line, whole-job accounting."""
    import json
    import os
    import subprocess
    import sys

    env = dict(os.environ, STAC_BENCH_BACKEND="gloo", STAC_BENCH_SHARE_GPU="1")
    port = 29950 + int(torch.randint(0, 40, (1,)).item())
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(ROOT / "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", *args]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["value"] > 0
    if args[1] == "fit":
        assert d["config"]["collective_backend"] == "gloo" and d["config"]["collective_world_size"] == 2
