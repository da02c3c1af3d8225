"""HIP path vs CPU oracle -- the parity tests proper (need a real MI355X: `-m gpu`).

Everything goes through the C ABI (`libstac_hip.so`) via `stac_mjx_amd.engine.Engine`.
Tolerances: the north star asks for qpos / offsets within 1e-4 of the reference.  The kernels are
built to reproduce the float32 oracle operation for operation, so these tests assert EXACT equality
(`assert_array_equal`, tolerance 0) wherever the oracle is defined, which implies the 1e-4 bar;
`TOL_NORTH_STAR` is used only where two different algorithms are compared (golden FK vectors).
"""

import numpy as np
import pytest

from conftest import GOLDEN

pytestmark = pytest.mark.gpu

TOL_NORTH_STAR = 1e-4  # BASELINE.json north_star: "within 1e-4"
TOL_FK_GOLDEN = 5e-7   # stored reference output vs restated FK (float32 epsilon level)


def _engine(fs, **kw):
    from stac_mjx_amd.engine import Engine

    return Engine(fs.tables, fs.lb, fs.ub, **kw)


def _oracle(fs, **kw):
    from oracle import Oracle

    return Oracle(fs.tables, **kw)


def _np(t):
    return t.detach().cpu().numpy()


# ---- FK ------------------------------------------------------------------------------------------------
def test_fk_bit_exact_and_golden(rodent_setup_legacy, demo_viz):
    fs = rodent_setup_legacy
    eng, orc = _engine(fs), _oracle(fs)
    eng.set_site_pos(demo_viz["offsets"])
    orc.set_site_pos(demo_viz["offsets"])
    out = eng.fk(demo_viz["qpos"])
    xpos, xquat, sx, qn = (_np(out[k]) for k in ("xpos", "xquat", "site_xpos", "qpos"))
    assert np.abs(xpos - demo_viz["xpos"]).max() <= TOL_FK_GOLDEN
    assert np.abs(sx - demo_viz["walker_body_sites"]).max() <= TOL_FK_GOLDEN
    for f in range(50):
        r = orc.fk(demo_viz["qpos"][f])
        np.testing.assert_array_equal(xpos[f], r["xpos"])
        np.testing.assert_array_equal(xquat[f], r["xquat"])
        np.testing.assert_array_equal(sx[f], r["site_xpos"])
        np.testing.assert_array_equal(qn[f], r["qpos"])


def test_fk_random_poses_non_unit_quaternions(rodent_setup):
    fs = rodent_setup
    eng, orc = _engine(fs), _oracle(fs)
    rng = np.random.default_rng(5)
    q = fs.tables.qpos0[None] + rng.normal(0, 0.4, (33, 74)).astype(np.float32)
    out = eng.fk(q)
    for f in range(33):
        r = orc.fk(q[f])
        np.testing.assert_array_equal(_np(out["xpos"][f]), r["xpos"])
        np.testing.assert_array_equal(_np(out["xquat"][f]), r["xquat"])
        np.testing.assert_array_equal(_np(out["qpos"][f]), r["qpos"])


@pytest.mark.parametrize("model", ["fly", "mouse", "random_ball_slide", "random_fixed_root"])
def test_fk_output_subsets_ragged_blocks(model, fly_setup, mouse_setup):
    """fk_kernel has one path per set of requested outputs (marker sites after the walk when both body arrays are wanted,
    during it otherwise; the normalised coordinates as a block copy + the quaternion joints' words) and works in blocks of 64
    poses: every subset, on 150 poses (two full blocks and a ragged one) and on a single pose, equals the oracle bit for bit."""
    from oracle import Oracle
    from stac_mjx_amd.engine import Engine

    if model in ("fly", "mouse"):
        fs = fly_setup if model == "fly" else mouse_setup
        t, lb, ub = fs.tables, fs.lb, fs.ub
    else:
        rng = np.random.default_rng(77 if model == "random_ball_slide" else 78)
        t = _random_tables(rng, 37, model == "random_ball_slide", p_slide=0.2, p_ball=0.2 if model == "random_ball_slide" else 0.0)
        lb, ub = np.full(t.nq, -np.inf, np.float32), np.full(t.nq, np.inf, np.float32)
    eng, orc = Engine(t, lb, ub), Oracle(t)
    rng = np.random.default_rng(11)
    q = (np.asarray(t.qpos0, np.float32)[None] + rng.normal(0, 0.4, (150, t.nq))).astype(np.float32)
    ref = [orc.fk(x.copy()) for x in q]
    names = ("qpos", "xpos", "xquat", "site_xpos")
    for want in [names, names[1:], ("site_xpos",), ("xpos", "xquat"), ("xquat", "site_xpos"), ("qpos",)]:
        for n in (150, 1):
            out = eng.fk(q[:n], want=want)
            for k in names:
                if k in want:
                    got = _np(out[k]).reshape(n, -1)
                    exp = np.stack([np.asarray(r[k], np.float32).reshape(-1) for r in ref[:n]])
                    np.testing.assert_array_equal(got, exp, err_msg=f"{model} {want} N={n} {k}")
                else:
                    assert out[k] is None


def test_fk_normalised_coordinates_in_place(rodent_setup):
    """`stac_fk` with qpos_norm_out == qpos (the caller normalises its own array): same result as into a separate array."""
    import torch
    from stac_mjx_amd.engine import _ptr

    fs = rodent_setup
    eng = _engine(fs)
    rng = np.random.default_rng(3)
    q = (fs.tables.qpos0[None] + rng.normal(0, 0.4, (130, 74))).astype(np.float32)
    ref = eng.fk(q)
    qd = torch.from_numpy(q).to(eng.device)
    xpos = torch.empty((130, eng.nbody, 3), dtype=torch.float32, device=eng.device)
    eng._check(eng.lib.stac_fk(eng._h, _ptr(qd), 130, _ptr(qd), _ptr(xpos), None, None, eng._stream()))
    torch.cuda.synchronize()
    assert torch.equal(qd, ref["qpos"]) and torch.equal(xpos, ref["xpos"])
    assert not np.array_equal(_np(qd), q)  # (the free joint's quaternion was not a unit one)


# ---- single solves (the StacCore.q_opt seam) ------------------------------------------------------------
@pytest.mark.parametrize("lanes", [4, 8, 16, 32, 64])
@pytest.mark.parametrize("which", ["all", "part", "root_trunk"])
def test_q_solve_bit_exact(rodent_setup, rodent_mocap, lanes, which):
    fs = rodent_setup
    eng, orc = _engine(fs, lanes_per_chain=lanes), _oracle(fs)
    N = 5
    kp = rodent_mocap[100:100 + N]
    q0 = np.repeat(fs.tables.qpos0[None], N, 0)
    q0[:, :3] = kp[:, 3 * fs.root_kp_idx: 3 * fs.root_kp_idx + 3]
    if which == "all":
        qs, ks = np.ones(74, bool), np.ones(69, bool)
    elif which == "part":
        qs, ks = fs.part_masks[1], np.ones(69, bool)
    else:
        qs, ks = np.arange(74) < 7, np.repeat(fs.trunk_kps, 3)
    params, state, counters = eng.q_solve(kp, q0, qs, ks)
    params, state, counters = _np(params), _np(state), _np(counters)
    for n in range(N):
        x, st = orc.q_opt(kp[n], qs, ks, q0[n], fs.lb, fs.ub)
        assert counters[n].tolist() == [st["iter_num"], st["ls_evals"], st["grad_evals"], 1]
        np.testing.assert_array_equal(params[n], x)
        np.testing.assert_array_equal(state[n], np.array([st["error"], st["stepsize"], st["t"], st["loss"]], np.float32))
        assert np.abs(params[n] - x).max() <= TOL_NORTH_STAR


@pytest.mark.parametrize("seed", range(8))
def test_q_solve_random_masks_bit_exact(rodent_setup, rodent_mocap, seed):
    """Random qs_to_opt / per-coordinate kps_to_opt masks (incl. nothing to optimise and no keypoint weighted), random
    solver settings and starting points: the StacCore.q_opt seam equals the oracle bit for bit."""
    fs = rodent_setup
    rng = np.random.default_rng(seed)
    tol, maxiter, maxls = float(rng.choice([1e-3, 1e-5])), int(rng.integers(1, 25)), int(rng.choice([2, 5, 15]))
    eng, orc = _engine(fs, tol=tol, maxiter=maxiter, maxls=maxls), _oracle(fs, tol=tol, maxiter=maxiter, maxls=maxls)
    N = 3
    kp = rodent_mocap[rng.integers(0, 900):][:N]
    q0 = np.repeat(fs.tables.qpos0[None], N, 0) + rng.normal(0, 0.05, (N, 74)).astype(np.float32)
    q0[:, :3] = kp[:, 3 * fs.root_kp_idx: 3 * fs.root_kp_idx + 3]
    qs = rng.random(74) < rng.choice([0.0, 0.1, 0.5, 1.0])
    ks = rng.random(69) < rng.choice([0.0, 0.3, 1.0])
    params, state, counters = (_np(v) for v in eng.q_solve(kp, q0, qs, ks))
    for n in range(N):
        x, st = orc.q_opt(kp[n], qs, ks, q0[n], fs.lb, fs.ub)
        assert counters[n].tolist() == [st["iter_num"], st["ls_evals"], st["grad_evals"], 1]
        np.testing.assert_array_equal(params[n], x)
        np.testing.assert_array_equal(state[n], np.array([st["error"], st["stepsize"], st["t"], st["loss"]], np.float32))


# ---- the q_phase drivers ---------------------------------------------------------------------------------
def _q_phase_twice(eng, kp, **kw):
    """Run the launch twice on the same engine and require identical outputs: the second launch finds the scratch memory
    (register spills) as the first one left it, so a value read before it is written shows up here (it did: round 3)."""
    a = eng.q_phase(kp, **kw)
    b = eng.q_phase(kp, **kw)
    for k in ("qpos", "frame_error", "counters", "carry_qpos", "marker_sites", "xpos"):
        if a.get(k) is not None:
            assert (a[k] == b[k]).all(), f"second launch differs in {k}"
    return b


def _compare_phase(res, ref, bodies=True):
    np.testing.assert_array_equal(_np(res["counters"]).astype(np.uint32), ref["counters"])
    np.testing.assert_array_equal(_np(res["qpos"]), ref["qpos"])
    np.testing.assert_array_equal(_np(res["frame_error"]), ref["frame_error"])
    np.testing.assert_array_equal(_np(res["marker_sites"]), ref["marker_sites"])
    if bodies:
        np.testing.assert_array_equal(_np(res["xpos"]), ref["xpos"])
        np.testing.assert_array_equal(_np(res["xquat"]), ref["xquat"])
    assert np.abs(_np(res["qpos"]) - ref["qpos"]).max() <= TOL_NORTH_STAR


@pytest.mark.parametrize("lanes", [4, 8, 16, 32, 64])
def test_q_phase_ik_clips_bit_exact(rodent_setup, rodent_mocap, lanes):
    """Stac.ik_only semantics: root optimisation on frame 0 of every clip, then warm-started frames."""
    fs = rodent_setup
    eng, orc = _engine(fs, lanes_per_chain=lanes), _oracle(fs)
    kp = rodent_mocap[:9 * 3].reshape(9, 3, 69)  # 9 clips (ragged vs lanes), 3 frames
    res = _q_phase_twice(eng, kp, part_masks=fs.part_masks, trunk_kps=fs.trunk_kps, root_kp_idx=fs.root_kp_idx,
                         root_dims=fs.root_dims, do_root_opt=True)
    ref = orc.ik_clips(kp, fs.lb, fs.ub, fs.part_masks, fs.trunk_kps, fs.root_kp_idx, fs.root_dims)
    _compare_phase(res, ref)
    np.testing.assert_array_equal(_np(res["carry_qpos"]), ref["qpos"][:, -1])


@pytest.mark.timeout(300)
@pytest.mark.parametrize("lanes,solver", [(16, "pg"), (0, "pg"), (0, "lm")])
def test_q_phase_missing_keypoints_terminate_and_stay_in_their_chain(rodent_setup, rodent_mocap, lanes, solver):
    """Mocap files carry NaN for markers the cameras lost.  The reference has no guard (jaxopt then iterates on NaN); what the
    engine owes is (a) every launch ends -- NaN compares false everywhere, so the stopping test, the line search and the LM
    accept rule all fall through -- and (b) the damage stays in the chain that has the NaN: the other chains of the same
    wavefront are bit-identical to a launch without it, and to the oracle."""
    from stac_mjx_amd.engine import Engine

    fs = rodent_setup
    kp = rodent_mocap[100:100 + 8 * 3].reshape(8, 3, 69).copy()
    args = dict(part_masks=fs.part_masks, trunk_kps=fs.trunk_kps, root_kp_idx=fs.root_kp_idx, root_dims=fs.root_dims, do_root_opt=True)
    eng = Engine(fs.tables, fs.lb, fs.ub, tol=1e-4, maxiter=40, lanes_per_chain=lanes, solver=solver, lm_maxiter=10)
    keys = ("qpos", "frame_error", "counters", "marker_sites")
    clean = {k: _np(v) for k, v in eng.q_phase(kp, **args).items() if k in keys}
    dirty_kp = kp.copy()
    dirty_kp[2, 1, 3 * 5:3 * 5 + 3] = np.nan   # a limb marker, second frame of chain 2
    dirty_kp[5, 0, 3 * fs.root_kp_idx] = np.nan  # the root marker's x, first frame of chain 5 (the root solves start from it)
    dirty = {k: _np(v) for k, v in eng.q_phase(dirty_kp, **args).items() if k in keys}
    keep = [c for c in range(8) if c not in (2, 5)]
    for k in keys:
        np.testing.assert_array_equal(dirty[k][keep], clean[k][keep], err_msg=k)
    np.testing.assert_array_equal(dirty["qpos"][2, 0], clean["qpos"][2, 0])  # the frame before the NaN is untouched too
    if solver == "pg":
        ref = _oracle(fs, maxiter=40).ik_clips(dirty_kp, fs.lb, fs.ub, fs.part_masks, fs.trunk_kps, fs.root_kp_idx, fs.root_dims)
        np.testing.assert_array_equal(dirty["qpos"][keep], ref["qpos"][keep])
        np.testing.assert_array_equal(dirty["counters"][keep].astype(np.uint32), ref["counters"][keep])
        # (what a chain does AFTER its NaN is not compared: where a NaN survives a clip or a select is not part of the contract)
    eng.close()


@pytest.mark.parametrize("flags", ["0", "2", "4"])
@pytest.mark.parametrize("lanes", [8, 16, 32])
def test_q_phase_fk_program_and_level_loop_agree(rodent_setup, fly_setup, rodent_mocap, monkeypatch, flags, lanes):
    """The kernel has several FK implementations: the step program (one lane or four lanes per position; as ONE
    straight-line step when the model allows it, else -- or with STAC_HIP_FLAGS bit 2 -- step by step through flags and
    forms) and the walk over the level tables (bit 1 forces it, and the fly's 6-wide levels on 4-lane groups force it
    too): all equal the oracle."""
    monkeypatch.setenv("STAC_HIP_FLAGS", flags)
    fs = rodent_setup
    eng, orc = _engine(fs, lanes_per_chain=lanes), _oracle(fs)
    kp = rodent_mocap[600:610].reshape(5, 2, 69)
    res = eng.q_phase(kp, part_masks=fs.part_masks, trunk_kps=fs.trunk_kps, root_kp_idx=fs.root_kp_idx,
                      root_dims=fs.root_dims, do_root_opt=True)
    _compare_phase(res, orc.ik_clips(kp, fs.lb, fs.ub, fs.part_masks, fs.trunk_kps, fs.root_kp_idx, fs.root_dims))
    fly = fly_setup  # tethered: no root optimisation
    engf, orcf = _engine(fly, lanes_per_chain=lanes, maxiter=30), _oracle(fly, maxiter=30)
    rng = np.random.default_rng(11)
    qt = fly.tables.qpos0[None] + np.clip(rng.normal(0, 0.1, (4, fly.tables.nq)), -0.2, 0.2).astype(np.float32)
    qt[:, 3:7] = fly.tables.qpos0[3:7]
    kpf = np.stack([orcf.fk(q)["site_xpos"].reshape(-1) for q in qt]).reshape(2, 2, 3 * fly.tables.nsite)
    kpf = kpf + rng.normal(0, 1e-3, kpf.shape).astype(np.float32)
    resf = engf.q_phase(kpf, part_masks=fly.part_masks)
    _compare_phase(resf, orcf.ik_clips(kpf, fly.lb, fly.ub, fly.part_masks, fly.trunk_kps, 0, 7, do_root_opt=False))


@pytest.mark.parametrize("flags", ["0", "2", "4"])
@pytest.mark.parametrize("lanes", [16, 32])
@pytest.mark.parametrize("nodiet", [None, "1"])
def test_q_phase_fk_implementations_agree_on_the_mouse(mouse_setup, monkeypatch, flags, lanes, nodiet):
    """The same for the mouse (86-step program, 181 active joints, start pose outside the box for some ranges, root
    optimisation with root fast trips): program, program through the dispatch, level loop; with and without the LDS diet of
    a program launch (no level tables / body records staged, root program only as far as used).  Launched twice."""
    monkeypatch.setenv("STAC_HIP_FLAGS", flags)
    if nodiet:
        monkeypatch.setenv("STAC_HIP_NODIET", nodiet)
    ms = mouse_setup
    real = np.load(GOLDEN / "mouse_mocap_200.npy")
    kp = real[[3, 40, 77, 120, 160, 199]].reshape(3, 2, 102)
    eng, orc = _engine(ms, lanes_per_chain=lanes, maxiter=30), _oracle(ms, maxiter=30)
    res = _q_phase_twice(eng, kp, part_masks=ms.part_masks, trunk_kps=ms.trunk_kps, root_kp_idx=ms.root_kp_idx,
                         root_dims=ms.root_dims, do_root_opt=ms.do_root_opt)
    _compare_phase(res, orc.ik_clips(kp, ms.lb, ms.ub, ms.part_masks, ms.trunk_kps, ms.root_kp_idx, ms.root_dims,
                                     do_root_opt=ms.do_root_opt))


@pytest.mark.parametrize("wpe,wpb", [("2", "3"), ("3", "10"), ("2", "8"), ("4", "5")])
def test_q_phase_register_cap_variants(rodent_setup, rodent_mocap, monkeypatch, wpe, wpb):
    """The 2- and 3-wavefronts-per-SIMD builds of the 16-lane kernel (the launch shapes of large batches, forced here on a
    small one through the developer overrides), multi-wave workgroups with ragged last blocks: all equal the oracle.  (There
    is no 128-VGPR build at 16 lanes any more -- it did not pass the spill gate: a request for it runs the 2-per-SIMD one.)"""
    monkeypatch.setenv("STAC_HIP_WPE", wpe)
    monkeypatch.setenv("STAC_HIP_WPB", wpb)
    fs = rodent_setup
    eng, orc = _engine(fs, lanes_per_chain=16, maxiter=40), _oracle(fs, maxiter=40)
    kp = rodent_mocap[700:745].reshape(45, 1, 69)  # 45 chains: 12 waves, a partly filled last wave
    res = _q_phase_twice(eng, kp, part_masks=fs.part_masks, trunk_kps=fs.trunk_kps, root_kp_idx=fs.root_kp_idx,
                         root_dims=fs.root_dims, do_root_opt=True)
    _compare_phase(res, orc.ik_clips(kp, fs.lb, fs.ub, fs.part_masks, fs.trunk_kps, fs.root_kp_idx, fs.root_dims))


@pytest.mark.parametrize("frames", [1, 3])
@pytest.mark.parametrize("handoff", ["12", "40"])
def test_q_phase_straggler_handoff(rodent_setup, rodent_mocap, monkeypatch, frames, handoff):
    """Once most chains of a launch are done, the rest move from the throughput kernel to the latency kernel at an
    iteration boundary (solver state through global memory).  Forced here on a small batch: results, residuals and
    counters still equal the oracle bit for bit, whichever kernel finished a chain."""
    monkeypatch.setenv("STAC_HIP_HANDOFF", handoff)
    fs = rodent_setup
    eng, orc = _engine(fs, lanes_per_chain=16, maxiter=60), _oracle(fs, maxiter=60)
    kp = rodent_mocap[500:500 + 44 * frames].reshape(44, frames, 69)
    res = _q_phase_twice(eng, kp, part_masks=fs.part_masks, trunk_kps=fs.trunk_kps, root_kp_idx=fs.root_kp_idx,
                         root_dims=fs.root_dims, do_root_opt=True)
    ref = orc.ik_clips(kp, fs.lb, fs.ub, fs.part_masks, fs.trunk_kps, fs.root_kp_idx, fs.root_dims)
    _compare_phase(res, ref)
    np.testing.assert_array_equal(_np(res["carry_qpos"]), ref["qpos"][:, -1])


@pytest.mark.parametrize("handoff", [None, "10"])
def test_q_phase_chain_queue(rodent_setup, rodent_mocap, monkeypatch, handoff):
    """More chains than chain slots: the grid covers the slots and a group that finishes a chain takes the next
    unstarted one from a device counter (forced here: 16 slots for 50 chains), with and without the straggler
    hand-off on top.  Which group ran which chain must not matter: everything equals the oracle."""
    monkeypatch.setenv("STAC_HIP_QUEUE", "16")
    if handoff:
        monkeypatch.setenv("STAC_HIP_HANDOFF", handoff)
    fs = rodent_setup
    eng, orc = _engine(fs, lanes_per_chain=16, maxiter=40), _oracle(fs, maxiter=40)
    kp = rodent_mocap[100:200].reshape(50, 2, 69)
    res = eng.q_phase(kp, part_masks=fs.part_masks, trunk_kps=fs.trunk_kps, root_kp_idx=fs.root_kp_idx,
                      root_dims=fs.root_dims, do_root_opt=True)
    ref = orc.ik_clips(kp, fs.lb, fs.ub, fs.part_masks, fs.trunk_kps, fs.root_kp_idx, fs.root_dims)
    _compare_phase(res, ref)
    np.testing.assert_array_equal(_np(res["carry_qpos"]), ref["qpos"][:, -1])


def test_q_phase_carried_chain_no_root_opt(rodent_setup, rodent_mocap):
    """fit_offsets semantics: one chain continued across calls with q_init (stac.py:298-311)."""
    fs = rodent_setup
    eng, orc = _engine(fs), _oracle(fs)
    kp = rodent_mocap[:4]
    q, _ = orc.root_optimization(kp, fs.tables.qpos0, fs.lb, fs.ub, fs.trunk_kps, fs.root_kp_idx)
    ref1 = orc.pose_optimization(kp, q, fs.lb, fs.ub, fs.part_masks)
    ref2 = orc.pose_optimization(kp, ref1["carry_qpos"], fs.lb, fs.ub, fs.part_masks)
    r1 = eng.q_phase(kp[None], part_masks=fs.part_masks, q_init=q[None])
    r2 = eng.q_phase(kp[None], part_masks=fs.part_masks, q_init=r1["carry_qpos"])
    np.testing.assert_array_equal(_np(r1["qpos"][0]), ref1["qpos"])
    np.testing.assert_array_equal(_np(r2["qpos"][0]), ref2["qpos"])
    np.testing.assert_array_equal(_np(r2["frame_error"][0]), ref2["frame_error"])
    np.testing.assert_array_equal(_np(r2["xquat"][0]), ref2["xquat"])


def test_q_phase_no_parts_and_tight_iteration_cap(rodent_setup, rodent_mocap):
    fs = rodent_setup
    eng, orc = _engine(fs, maxiter=7, tol=1e-9), _oracle(fs, maxiter=7, tol=1e-9)
    kp = rodent_mocap[200:206].reshape(3, 2, 69)
    res = eng.q_phase(kp, part_masks=[], trunk_kps=fs.trunk_kps, root_kp_idx=fs.root_kp_idx, do_root_opt=True)
    ref = orc.ik_clips(kp, fs.lb, fs.ub, [], fs.trunk_kps, fs.root_kp_idx, fs.root_dims)
    _compare_phase(res, ref)
    assert (ref["counters"][..., 0] == 7).all()


@pytest.mark.parametrize("spec", ["0", "1"])
@pytest.mark.parametrize("maxls", [2, 3, 15])
def test_q_phase_line_search_bounds_and_speculative_mode(rodent_setup, rodent_mocap, monkeypatch, spec, maxls):
    """Small maxls exercises jaxopt's loop bound (a candidate taken WITHOUT evaluation); STAC_HIP_SPEC forces
    the speculative latency kernel (8 lane groups on one chain) on or off -- both must equal the oracle."""
    from oracle import Oracle
    from stac_mjx_amd.engine import Engine

    monkeypatch.setenv("STAC_HIP_SPEC", spec)
    fs = rodent_setup
    eng = Engine(fs.tables, fs.lb, fs.ub, tol=1e-4, maxiter=40, maxls=maxls)
    orc = Oracle(fs.tables, tol=1e-4, maxiter=40, maxls=maxls)
    kp = rodent_mocap[400:406].reshape(3, 2, 69)
    res = eng.q_phase(kp, part_masks=fs.part_masks, trunk_kps=fs.trunk_kps, root_kp_idx=fs.root_kp_idx,
                      root_dims=fs.root_dims, do_root_opt=True)
    ref = orc.ik_clips(kp, fs.lb, fs.ub, fs.part_masks, fs.trunk_kps, fs.root_kp_idx, fs.root_dims)
    _compare_phase(res, ref)


def test_speculative_mode_long_chain_matches_normal_mode(rodent_setup, rodent_mocap, monkeypatch):
    """One 12-frame warm-started chain: speculative kernel == regular kernel == oracle, bit for bit."""
    fs = rodent_setup
    kp = rodent_mocap[:12][None]
    outs = []
    for spec in ("0", "1"):
        monkeypatch.setenv("STAC_HIP_SPEC", spec)
        eng = _engine(fs)
        outs.append(eng.q_phase(kp, part_masks=fs.part_masks, trunk_kps=fs.trunk_kps, root_kp_idx=fs.root_kp_idx,
                                root_dims=fs.root_dims, do_root_opt=True))
    for k in ("qpos", "frame_error", "counters", "marker_sites", "xquat", "carry_qpos"):
        assert (outs[0][k] == outs[1][k]).all(), k
    ref = _oracle(fs).ik_clips(kp, fs.lb, fs.ub, fs.part_masks, fs.trunk_kps, fs.root_kp_idx, fs.root_dims)
    _compare_phase(outs[1], ref)


def test_q_phase_fruitfly_model(fly_setup):
    """Different tree (quat-oriented bodies, nq=43, K=30, 6 part groups, no root optimisation)."""
    fs = fly_setup
    eng, orc = _engine(fs, tol=5e-3, maxiter=60), _oracle(fs, tol=5e-3, maxiter=60)
    rng = np.random.default_rng(11)
    qtrue = fs.tables.qpos0[None] + np.clip(rng.normal(0, 0.15, (6, 43)), -0.3, 0.3).astype(np.float32)
    qtrue[:, 3:7] = fs.tables.qpos0[3:7]
    kp = np.stack([orc.fk(q)["site_xpos"].reshape(-1) for q in qtrue]).reshape(3, 2, 90)
    kp = kp + rng.normal(0, 1e-3, kp.shape).astype(np.float32)
    res = eng.q_phase(kp, part_masks=fs.part_masks)
    ref = orc.ik_clips(kp, fs.lb, fs.ub, fs.part_masks, fs.trunk_kps, 0, 7, do_root_opt=False)
    _compare_phase(res, ref)


@pytest.mark.parametrize("lanes", [0, 16, 64])
def test_q_phase_mouse_model(mouse_setup, lanes):
    """Mouse: nq = 230, 181 active bodies on 85 tree levels, K = 34 (> 32-site loss tree), no part
    groups; lanes = 0 takes the speculative latency kernel (3 chains), 16 / 64 the regular one."""
    fs = mouse_setup
    assert fs.tables.nq == 230 and fs.tables.nsite == 34 and fs.part_masks.shape[0] == 0
    eng, orc = _engine(fs, maxiter=25, lanes_per_chain=lanes), _oracle(fs, maxiter=25)
    rng = np.random.default_rng(4)
    qt = fs.tables.qpos0[None] + np.clip(rng.normal(0, 0.05, (3, 230)), -0.1, 0.1).astype(np.float32)
    qt[:, 3:7] = fs.tables.qpos0[3:7]
    kp = np.stack([orc.fk(q)["site_xpos"].reshape(-1) for q in qt]).reshape(3, 1, 102)
    kp = kp + rng.normal(0, 5e-4, kp.shape).astype(np.float32)
    res = _q_phase_twice(eng, kp, part_masks=fs.part_masks, trunk_kps=fs.trunk_kps, root_kp_idx=fs.root_kp_idx,
                         root_dims=fs.root_dims, do_root_opt=fs.do_root_opt)
    ref = orc.ik_clips(kp, fs.lb, fs.ub, fs.part_masks, fs.trunk_kps, fs.root_kp_idx, fs.root_dims,
                       do_root_opt=fs.do_root_opt)
    _compare_phase(res, ref)


def test_q_phase_ball_and_slide_joints():
    from stac_mjx_amd.fit_model import FitSetup
    from stac_mjx_amd.mjcf import align_joint_dims, compile_mjcf

    xml = """
    <mujoco><compiler angle="radian"/><worldbody>
      <body name="r" pos="0 0 0.5"><freejoint/>
        <site name="s0" pos="0.1 0 0"/><site name="s0b" pos="0 0.1 0.05"/>
        <body name="a" pos="0.2 0 0" quat="0.9 0.1 0.2 0.3"><joint name="ball" type="ball" pos="0.01 0.02 0"/>
          <site name="s1" pos="0 0.1 0"/><site name="s1b" pos="0.1 0.1 0"/>
          <body name="b" pos="0 0.2 0"><joint name="sl" type="slide" axis="1 1 0" range="-0.2 0.2"/>
            <joint name="h" axis="0 1 1" pos="0 0 .1" range="-1 1"/>
            <site name="s2" pos="0.05 0.02 0.1"/><site name="s2b" pos="-0.05 0.1 0"/></body></body></body>
    </worldbody></mujoco>"""
    t = compile_mjcf(xml, from_string=True)
    lb, ub, names = align_joint_dims(t.jnt_type, t.jnt_range, t.jnt_names)
    from oracle import Oracle
    from stac_mjx_amd.engine import Engine

    eng, orc = Engine(t, lb, ub, tol=1e-5, maxiter=80), Oracle(t, tol=1e-5, maxiter=80)
    rng = np.random.default_rng(2)
    qt = t.qpos0[None] + rng.normal(0, 0.1, (4, t.nq)).astype(np.float32)
    kp = np.stack([orc.fk(q)["site_xpos"].reshape(-1) for q in qt]).reshape(2, 2, 18)
    pm = np.zeros((1, t.nq), bool)
    pm[0, 7:] = True
    res = eng.q_phase(kp, part_masks=pm)
    ref = orc.ik_clips(kp, lb, ub, pm, np.ones(6, bool), 0, 7, do_root_opt=False)
    _compare_phase(res, ref)


# ---- offset phase ----------------------------------------------------------------------------------------------
def test_m_phase_bit_exact_and_partial_sums(rodent_setup_legacy, demo_viz):
    fs = rodent_setup_legacy
    eng, orc = _engine(fs), _oracle(fs)
    eng.set_site_pos(demo_viz["offsets"])
    orc.set_site_pos(demo_viz["offsets"])
    q, kp, m0 = demo_viz["qpos"], demo_viz["kp_data"], demo_viz["offsets"]
    part = eng.m_partial(kp, q)
    np.testing.assert_array_equal(_np(part), orc.m_partial(kp, q))
    off, err = eng.m_finish(part, m0, fs.is_regularized, 1.0)
    ref_off, ref_err = orc.m_finish(orc.m_partial(kp, q), m0, fs.is_regularized, 1.0)
    np.testing.assert_array_equal(_np(off), ref_off)
    assert float(err) == np.float32(ref_err)
    assert np.abs(_np(off) - ref_off).max() <= TOL_NORTH_STAR
    # sharded partial sums (what the all-reduce adds up) reproduce the closed form to rounding
    pa, pb = eng.m_partial(kp[:20], q[:20]), eng.m_partial(kp[20:], q[20:])
    off2, _ = eng.m_finish(pa + pb, m0, fs.is_regularized, 1.0)
    assert np.abs(_np(off2) - ref_off).max() <= 1e-6


def test_m_opt_known_answers_through_hip(toy_tables):
    """The reference's tests/unit/test_m_opt.py cases, run through the HIP path."""
    from stac_mjx_amd.engine import Engine
    from stac_mjx_amd.mjcf import align_joint_dims

    t = toy_tables
    lb, ub, _ = align_joint_dims(t.jnt_type, t.jnt_range, t.jnt_names)
    eng = Engine(t, lb, ub)
    gt_a = np.array([[0.1, 0.2, 0.3], [0.4, 0.5, 0.6], [0.15, 0.25, 0.35]], np.float32)
    gt_b = np.array([[0.2, -0.1, 0.4], [0.3, 0.4, -0.2], [-0.1, 0.3, 0.1]], np.float32)

    def keypoints(q, off):
        eng.set_site_pos(off)
        return eng.fk(q, want=("site_xpos",))["site_xpos"].reshape(len(q), -1)

    z = np.zeros((3, 3), np.float32)
    q = np.zeros((5, 3), np.float32)
    p, e = eng.m_opt(keypoints(q, gt_a), q, z, z, 0.0)
    np.testing.assert_allclose(_np(p), gt_a, atol=1e-5)
    assert float(e) < 1e-8
    q = (np.random.RandomState(42).randn(10, 3) * 0.5).astype(np.float32)
    p, _ = eng.m_opt(keypoints(q, gt_a), q, z, z, 0.0)
    np.testing.assert_allclose(_np(p), gt_a, atol=1e-5)
    q = np.zeros((8, 3), np.float32)
    q[:, 0] = np.linspace(0.0, np.pi / 4, 8)
    p, _ = eng.m_opt(keypoints(q, gt_b), q, gt_b, z, 0.0)
    np.testing.assert_allclose(_np(p), gt_b, atol=1e-5)
    q = (np.random.RandomState(99).randn(15, 3) * 1.5).astype(np.float32)
    p, _ = eng.m_opt(keypoints(q, gt_b), q, z, z, 0.0)
    np.testing.assert_allclose(_np(p), gt_b, atol=1e-4)
    q = (np.random.RandomState(42).randn(10, 3) * 0.3).astype(np.float32)
    kp = keypoints(q, gt_a)
    p, _ = eng.m_opt(kp, q, np.full((3, 3), 99.0, np.float32), np.ones((3, 3), np.float32), 0.0)
    np.testing.assert_allclose(_np(p), gt_a, atol=1e-5)
    p, _ = eng.m_opt(kp, q, z, np.ones((3, 3), np.float32), 1e6)
    np.testing.assert_allclose(_np(p), z, atol=1e-3)
    q = np.zeros((10, 3), np.float32)
    gt = np.full((3, 3), 0.5, np.float32)
    kp = keypoints(q, gt)
    is_reg = np.zeros((3, 3), np.float32)
    is_reg[0] = 1.0
    ps, _ = eng.m_opt(kp, q, z, is_reg, 1e4)
    pn, _ = eng.m_opt(kp, q, z, is_reg, 0.0)
    assert np.linalg.norm(_np(ps)[0]) < np.linalg.norm(_np(pn)[0])
    np.testing.assert_allclose(_np(ps)[1:], gt[1:], atol=1e-5)


def test_set_get_site_pos_roundtrip_and_effect(rodent_setup):
    fs = rodent_setup
    eng = _engine(fs)
    np.testing.assert_array_equal(_np(eng.get_site_pos()), fs.tables.site_pos)
    new = fs.tables.site_pos + 0.01
    eng.set_site_pos(new)
    np.testing.assert_array_equal(_np(eng.get_site_pos()), new)
    a = _np(eng.fk(fs.tables.qpos0[None])["site_xpos"])
    orc = _oracle(fs)
    orc.set_site_pos(new)
    np.testing.assert_array_equal(a[0], orc.fk(fs.tables.qpos0)["site_xpos"])


# ---- size-independent properties at BASELINE sizes ----------------------------------------------------------------
def test_full_size_properties_10k_frames(rodent_setup, rodent_mocap):
    """BASELINE config 2 size (10 000 frames, independent chains): properties the domain offers."""
    import torch

    fs = rodent_setup
    eng = _engine(fs)
    kp = np.tile(rodent_mocap, (10, 1)).reshape(10000, 1, 69)
    res = eng.q_phase(kp, part_masks=fs.part_masks, trunk_kps=fs.trunk_kps, root_kp_idx=fs.root_kp_idx,
                      do_root_opt=True)
    torch.cuda.synchronize()
    q = _np(res["qpos"])[:, 0]
    # (1) batch invariance / determinism: the 10 tiled copies of the same frame give identical bits
    np.testing.assert_array_equal(q[:1000], q[9000:])
    # (2) box constraints hold; root quaternion is unit after replace_qs
    assert (q >= fs.lb - 1e-6).all() and (q <= fs.ub + 1e-6).all()
    np.testing.assert_allclose(np.linalg.norm(q[:, 3:7], axis=1), 1.0, atol=1e-5)
    # (3) joints that are not ancestors of any marker never move
    orc = _oracle(fs)
    qr = fs.tables.qpos0 + np.random.default_rng(3).normal(0, 0.1, 74).astype(np.float32)
    _, g = orc.q_loss(qr, kp[0, 0], np.ones(74, bool), np.ones(69, bool), qr)
    np.testing.assert_array_equal(q[:, g == 0], np.broadcast_to(fs.tables.qpos0[g == 0], (10000, 29)))
    # (4) idempotence of the outputs: FK(qpos_out) == markers_out / xpos_out
    #     (stac_fk re-normalises the stored unit quaternion: allow the last ulp)
    fk = eng.fk(res["qpos"].reshape(-1, 74))
    np.testing.assert_allclose(_np(fk["site_xpos"]), _np(res["marker_sites"]).reshape(-1, 23, 3), atol=3e-7, rtol=0)
    np.testing.assert_allclose(_np(fk["xpos"]), _np(res["xpos"]).reshape(-1, 67, 3), atol=3e-7, rtol=0)
    # (5) a sample of frames agrees bit for bit with the oracle
    idx = [0, 123, 999]
    ref = orc.ik_clips(kp[idx], fs.lb, fs.ub, fs.part_masks, fs.trunk_kps, fs.root_kp_idx, fs.root_dims)
    np.testing.assert_array_equal(q[idx], ref["qpos"][:, 0])
    # (6) the fit explains the markers (initial offsets -> cm-level residual, like the oracle)
    err = np.linalg.norm(_np(res["marker_sites"])[:, 0] - kp[:, 0].reshape(-1, 23, 3), axis=-1)
    assert err.mean() < 2e-2


def test_errors_are_loud(rodent_setup):
    from stac_mjx_amd.engine import StacHipError

    fs = rodent_setup
    eng = _engine(fs, maxiter=0)
    with pytest.raises(StacHipError):
        eng.q_phase(np.zeros((1, 1, 69), np.float32), part_masks=[])
    with pytest.raises(ValueError):
        _engine(fs).q_phase(np.zeros((1, 1, 68), np.float32), part_masks=[])


@pytest.mark.parametrize("lanes", [0, 16])
def test_more_than_64_sites(rodent_setup, lanes):
    """70 fit sites: the loss tree leaves the registers (pairwise tree in LDS, same order as the oracle's)."""
    from oracle import Oracle
    from stac_mjx_amd.engine import Engine

    fs = rodent_setup
    rng = np.random.default_rng(70)
    t = fs.tables.copy()
    pick = np.sort(rng.integers(0, 23, 70))
    t.nsite = 70
    t.site_bodyid = fs.tables.site_bodyid[pick].astype(np.int32)
    t.site_pos = (fs.tables.site_pos[pick] + rng.normal(0, 2e-3, (70, 3))).astype(np.float32)
    t.site_names = [f"s{i}" for i in range(70)]
    orc = Oracle(t, tol=1e-4, maxiter=15)
    q = np.tile(t.qpos0, (4, 1)) + rng.normal(0, 0.05, (4, 74)).astype(np.float32)
    kp = np.stack([orc.fk(x.copy())["site_xpos"].reshape(-1) for x in q]).astype(np.float32).reshape(2, 2, 210)
    kp = (kp + rng.normal(0, 1e-3, kp.shape)).astype(np.float32)
    trunk = (rng.random(70) < 0.4).astype(np.uint8)
    eng = Engine(t, fs.lb, fs.ub, tol=1e-4, maxiter=15, lanes_per_chain=lanes)
    res = eng.q_phase(kp, part_masks=fs.part_masks, trunk_kps=trunk, root_kp_idx=3, root_dims=7, do_root_opt=True)
    _compare_phase(res, orc.ik_clips(kp, fs.lb, fs.ub, fs.part_masks, trunk, 3, 7))


# ---- optional LM solver (STAC_SOLVER_LM; not the reference's algorithm) vs its oracle statement ---------------------------------
TOL_LM_MARKERS = 0.0    # since round 5 oracle/stac_oracle.c::q_opt_lm_ws is the LM kernel's operation sequence: bit for bit


@pytest.mark.parametrize("lanes", [16, 32, 64])
def test_lm_q_phase_equals_oracle_lm_bit_for_bit(rodent_setup, rodent_mocap, lanes):
    """The optional LM solver against its CPU statement: the same evaluations, Gauss-Newton entries, L^T D L pivots, steps and
    accept / reject turns -- qpos, stopping residuals and counters bit for bit (until round 4 the oracle ran a dense Cholesky and
    the two were compared in marker space at 0.5 mm)."""
    from oracle import Oracle
    from stac_mjx_amd.engine import Engine

    fs = rodent_setup
    eng = Engine(fs.tables, fs.lb, fs.ub, tol=1e-4, solver="lm", lm_maxiter=40, lanes_per_chain=lanes)
    orc = Oracle(fs.tables, tol=1e-4)
    kp = rodent_mocap[::40][:10].reshape(5, 2, 69)
    res = eng.q_phase(kp, part_masks=fs.part_masks, trunk_kps=fs.trunk_kps, root_kp_idx=fs.root_kp_idx,
                      root_dims=fs.root_dims, do_root_opt=True)
    ref = orc.ik_clips_lm(kp, fs.lb, fs.ub, fs.part_masks, fs.trunk_kps, fs.root_kp_idx, fs.root_dims)
    np.testing.assert_array_equal(_np(res["qpos"]).view(np.uint32), ref["qpos"].view(np.uint32))
    np.testing.assert_array_equal(_np(res["frame_error"]).view(np.uint32), ref["frame_error"].view(np.uint32))
    np.testing.assert_array_equal(_np(res["counters"]).astype(np.uint32), ref["counters"])
    np.testing.assert_array_equal(_np(res["marker_sites"]), ref["marker_sites"])
    pg = orc.ik_clips(kp, fs.lb, fs.ub, fs.part_masks, fs.trunk_kps, fs.root_kp_idx, fs.root_dims)
    mk, q = _np(res["marker_sites"]), _np(res["qpos"])
    tgt = kp.reshape(5, 2, 23, 3)
    e_hip = np.linalg.norm(mk - tgt, axis=-1).mean()
    e_ref = np.linalg.norm(ref["marker_sites"] - tgt, axis=-1).mean()
    e_pg = np.linalg.norm(pg["marker_sites"] - tgt, axis=-1).mean()
    assert np.abs(mk - ref["marker_sites"]).max() <= TOL_LM_MARKERS
    assert abs(e_hip - e_ref) <= 2e-5 and e_hip <= e_pg + 1e-5
    assert (_np(res["frame_error"]) <= 1e-4).all()
    assert (q >= fs.lb - 1e-6).all() and (q <= fs.ub + 1e-6).all()
    # outputs are self-consistent: FK(qpos_out) == markers_out
    fk = eng.fk(res["qpos"].reshape(-1, 74))
    np.testing.assert_allclose(_np(fk["site_xpos"]), mk.reshape(-1, 23, 3), atol=3e-7, rtol=0)
    it = _np(res["counters"])[..., 0].mean()
    assert it < 60, it


@pytest.mark.parametrize("lanes", [16, 64])
def test_lm_fruitfly_bit_exact_and_mouse_refused(fly_setup, mouse_setup, lanes):
    """Round 6: the optional LM solver on BASELINE configs[4]'s model (fruit fly: oriented leg bodies, 43 coordinates, no root
    optimisation, six part groups) equals the oracle's LM bit for bit and fits the markers at least as well as the PG solver; the mouse
    (230 coordinates: 12.5 KB of LDS per chain times its path depth) is beyond the solver's capacity -- LDS per CU / 192 coordinates -- and is REFUSED with an error that says so (never mis-run)."""
    from oracle import Oracle
    from stac_mjx_amd.engine import Engine, StacHipError

    fs = fly_setup
    t = fs.tables
    orc = Oracle(t, tol=1e-4)
    rng = np.random.default_rng(31)
    qt = t.qpos0[None] + np.clip(rng.normal(0, 0.15, (8, t.nq)), -0.3, 0.3).astype(np.float32)
    qt[:, 3:7] = t.qpos0[3:7]
    kp = np.stack([orc.fk(q)["site_xpos"].reshape(-1) for q in qt]).astype(np.float32)
    kp = (kp + rng.normal(0, 1e-3, kp.shape)).astype(np.float32).reshape(4, 2, -1)
    eng = Engine(t, fs.lb, fs.ub, tol=1e-4, solver="lm", lm_maxiter=30, lanes_per_chain=lanes)
    res = eng.q_phase(kp, part_masks=fs.part_masks)
    ref = orc.ik_clips_lm(kp, fs.lb, fs.ub, fs.part_masks, fs.trunk_kps, 0, 7, do_root_opt=False, maxiter=30)
    np.testing.assert_array_equal(_np(res["qpos"]).view(np.uint32), ref["qpos"].view(np.uint32))
    np.testing.assert_array_equal(_np(res["frame_error"]).view(np.uint32), ref["frame_error"].view(np.uint32))
    np.testing.assert_array_equal(_np(res["counters"]).astype(np.uint32), ref["counters"])
    pg = orc.ik_clips(kp, fs.lb, fs.ub, fs.part_masks, fs.trunk_kps, 0, 7, do_root_opt=False)
    tgt = kp.reshape(4, 2, t.nsite, 3)
    assert np.linalg.norm(_np(res["marker_sites"]) - tgt, axis=-1).mean() <= np.linalg.norm(pg["marker_sites"] - tgt, axis=-1).mean() + 1e-5
    eng.close()
    ms = mouse_setup
    engm = Engine(ms.tables, ms.lb, ms.ub, tol=1e-4, solver="lm", lm_maxiter=5, lanes_per_chain=lanes)
    kpm = np.zeros((2, 1, 3 * ms.tables.nsite), np.float32)
    with pytest.raises(StacHipError, match="LM q_phase kernel limits|192 optimised coordinates"):
        engm.q_phase(kpm, part_masks=ms.part_masks, trunk_kps=ms.trunk_kps, root_kp_idx=ms.root_kp_idx, root_dims=ms.root_dims,
                     do_root_opt=ms.do_root_opt)
    engm.close()


@pytest.mark.parametrize("seed,free_root", [(0, True), (1, False), (2, True), (5, False)])
def test_lm_ball_and_slide_models_bit_exact(seed, free_root):
    """Round 5: ball joints in the LM solver (four raw quaternion coordinates per joint, columns in the frame the rotation is applied
    in, the gauge term of every quaternion) -- HIP == the oracle's LM bit for bit on random trees with ball, slide and hinge joints."""
    from oracle import Oracle
    from stac_mjx_amd.engine import Engine
    from test_oracle import _ball_case

    rng, t, lb, ub = _ball_case(seed, free_root)
    assert (t.jnt_type == 1).sum() >= 3
    orc = Oracle(t, tol=1e-4, maxiter=50)
    C, F, K, nq = 9, 2, t.nsite, t.nq
    q = np.tile(t.qpos0, (C * F, 1)) + rng.normal(0, 0.15, (C * F, nq)).astype(np.float32)
    q = np.clip(q, np.where(np.isfinite(lb), lb, -3), np.where(np.isfinite(ub), ub, 3)).astype(np.float32)
    kp = np.stack([orc.fk(x.copy())["site_xpos"].reshape(-1) for x in q]).astype(np.float32)
    kp = (kp + rng.normal(0, 1e-3, kp.shape)).astype(np.float32).reshape(C, F, 3 * K)
    part = (rng.random((2, nq)) < 0.4).astype(np.uint8)
    trunk = (rng.random(K) < 0.6).astype(np.uint8)
    trunk[0] = 1
    ref = orc.ik_clips_lm(kp, lb, ub, part, trunk, 0, 7, do_root_opt=free_root, maxiter=15)
    for lanes in (0, 64):
        eng = Engine(t, lb, ub, tol=1e-4, solver="lm", lm_maxiter=15, lanes_per_chain=lanes)
        res = eng.q_phase(kp, part_masks=part, trunk_kps=trunk, root_kp_idx=0, root_dims=7, do_root_opt=free_root)
        np.testing.assert_array_equal(_np(res["qpos"]).view(np.uint32), ref["qpos"].view(np.uint32))
        np.testing.assert_array_equal(_np(res["frame_error"]).view(np.uint32), ref["frame_error"].view(np.uint32))
        np.testing.assert_array_equal(_np(res["counters"]).astype(np.uint32), ref["counters"])
        eng.close()
    tgt = kp.reshape(C, F, K, 3)
    assert np.linalg.norm(ref["marker_sites"].reshape(C, F, K, 3) - tgt, axis=-1).mean() < 5e-3


def test_lm_chain_queue_same_as_static_assignment(rodent_setup, rodent_mocap, monkeypatch):
    """LM kernel with the chain queue forced (8 slots for 30 chains): which group ran which chain must not matter --
    identical outputs to the launch where every chain has its own slot."""
    from stac_mjx_amd.engine import Engine

    fs = rodent_setup
    kp = rodent_mocap[0:60].reshape(30, 2, 69)
    outs = []
    for q in ("0", "8"):
        monkeypatch.setenv("STAC_HIP_QUEUE", q)
        eng = Engine(fs.tables, fs.lb, fs.ub, tol=1e-4, solver="lm", lm_maxiter=40)
        outs.append(eng.q_phase(kp, part_masks=fs.part_masks, trunk_kps=fs.trunk_kps, root_kp_idx=fs.root_kp_idx,
                                root_dims=fs.root_dims, do_root_opt=True))
    for k in ("qpos", "frame_error", "counters", "marker_sites", "carry_qpos"):
        assert (outs[0][k] == outs[1][k]).all(), k


# ---- random models: the plan builder (levels, positions, stored transforms, step program) on arbitrary trees ------------
def _random_tables(rng, nbody, free_root, p_slide=0.1, p_ball=0.0, max_children_bias=0.6, lean=False, k_max=12, p_oriented=None):
    """A random kinematic tree as ModelTables: depth-first body order, 0-3 joints per body (mostly hinges, some with
    jnt_pos == 0, some slides / balls), random body orientations (some identity), sites on random bodies.
    lean: what the lean kernels' split kinematics take -- only hinges below the (free) root; some bodies with body_pos == 0;
    oriented bodies only on request (p_oriented: round 6 -- one more product of the quaternion pass each)."""
    from stac_mjx_amd.mjcf import JNT_BALL, JNT_FREE, JNT_HINGE, JNT_SLIDE, ModelTables

    parent = [0] * nbody
    for b in range(2, nbody):  # b = 1 is the root body under the world
        # depth-first numbering: the parent must be on the path from the root to body b - 1
        path = [b - 1]
        while path[-1] > 1:
            path.append(parent[path[-1]])
        parent[b] = path[0] if rng.random() < max_children_bias else int(rng.choice(path))
    parent[1] = 0
    depth = [0] * nbody
    for b in range(1, nbody):
        depth[b] = depth[parent[b]] + 1

    def unit(v):
        return v / np.linalg.norm(v)

    body_pos = rng.normal(0, 0.05, (nbody, 3))
    body_quat = np.tile([1.0, 0, 0, 0], (nbody, 1))
    for b in range(1, nbody):
        if (not lean and rng.random() < 0.4) if p_oriented is None else (rng.random() < p_oriented):
            body_quat[b] = unit(rng.normal(0, 1, 4))
        if lean and rng.random() < 0.15:
            body_pos[b] = 0.0
    jt, jadr_q, jbody, jpos, jaxis, jrange, qpos0 = [], [], [], [], [], [], []
    body_jntadr, body_jntnum = [-1] * nbody, [0] * nbody
    nq = 0
    for b in range(1, nbody):
        n = 1 if (b == 1 and free_root) else int(rng.choice([0, 1, 1, 1, 2, 3]))
        if n:
            body_jntadr[b] = len(jt)
        body_jntnum[b] = n
        for i in range(n):
            if b == 1 and free_root:
                ty = JNT_FREE
            else:
                u = rng.random()
                ty = JNT_SLIDE if u < p_slide else (JNT_BALL if u < p_slide + p_ball else JNT_HINGE)
            jt.append(ty)
            jadr_q.append(nq)
            jbody.append(b)
            jpos.append(np.zeros(3) if rng.random() < 0.4 else rng.normal(0, 0.02, 3))
            jaxis.append(unit(rng.normal(0, 1, 3)) if rng.random() < 0.5 else np.eye(3)[rng.integers(3)])
            if ty == JNT_FREE:
                qpos0 += [0, 0, 0.1, 1, 0, 0, 0]
                jrange.append([0, 0])
                nq += 7
            elif ty == JNT_BALL:
                qpos0 += [1, 0, 0, 0]
                jrange.append([0, 0])
                nq += 4
            else:
                qpos0.append(float(rng.normal(0, 0.05)) if rng.random() < 0.3 else 0.0)
                jrange.append([-1.0, 1.2] if ty == JNT_HINGE else [-0.05, 0.05])
                nq += 1
    K = int(rng.integers(3, k_max))
    site_body = np.sort(rng.integers(1, nbody, K)).astype(np.int32)
    return ModelTables(
        nbody=nbody, njnt=len(jt), nq=nq, nsite=K, body_parentid=np.array(parent, np.int32),
        body_pos=body_pos.astype(np.float32), body_quat=body_quat.astype(np.float32),
        body_jntadr=np.array(body_jntadr, np.int32), body_jntnum=np.array(body_jntnum, np.int32),
        body_depth=np.array(depth, np.int32), jnt_type=np.array(jt, np.int32), jnt_qposadr=np.array(jadr_q, np.int32),
        jnt_bodyid=np.array(jbody, np.int32), jnt_pos=np.array(jpos, np.float32).reshape(-1, 3),
        jnt_axis=np.array(jaxis, np.float32).reshape(-1, 3), jnt_range=np.array(jrange, np.float32).reshape(-1, 2),
        qpos0=np.array(qpos0, np.float32), site_bodyid=site_body, site_pos=rng.normal(0, 0.01, (K, 3)).astype(np.float32),
        body_names=[f"b{i}" for i in range(nbody)], jnt_names=[f"j{i}" for i in range(len(jt))],
        site_names=[f"s{i}" for i in range(K)])


@pytest.mark.parametrize("seed", list(range(12)) + [129])
def test_random_models_bit_exact(seed):
    """Random trees (3-40 bodies, chains and bushes, free or fixed root, slides, every third one with ball joints)
    through FK and the q_phase at several lane-group sizes: HIP == oracle bit for bit.  (Seed 129, found by
    tests/fuzz_random_models.py in round 3: nq = 81 at 4 lanes per chain, an instantiation in which the compiler keeps a
    lambda out of line -- its reads of the plan go through flat pointers, which faulted while the plan pointer of a
    program launch pointed in front of the LDS base.)"""
    from oracle import Oracle
    from stac_mjx_amd.engine import Engine
    from stac_mjx_amd.mjcf import JNT_BALL, JNT_FREE, JNT_QPOS_DIMS

    rng = np.random.default_rng(1000 + seed)
    free_root = bool(seed % 2)
    t = _random_tables(rng, int(rng.integers(3, 41)), free_root, p_ball=0.15 if seed % 3 == 0 else 0.0,
                       max_children_bias=0.05 if seed >= 10 else float(rng.choice([0.3, 0.6, 0.9])))  # 10, 11: bushes
    nq, K = t.nq, t.nsite
    if nq == 0:
        pytest.skip("no joints drawn")
    lb, ub = np.full(nq, -np.inf, np.float32), np.full(nq, np.inf, np.float32)
    for j in range(t.njnt):
        a, ty = int(t.jnt_qposadr[j]), int(t.jnt_type[j])
        if ty == JNT_FREE:
            lb[a + 3:a + 7], ub[a + 3:a + 7] = -1, 1
        elif ty == JNT_BALL:
            lb[a:a + 4], ub[a:a + 4] = -1, 1
        else:
            lb[a], ub[a] = min(t.jnt_range[j, 0], 0.0), t.jnt_range[j, 1]
    orc = Oracle(t, tol=1e-5, maxiter=12)
    # poses inside the box, quaternions deliberately not normalised
    q = np.tile(t.qpos0, (6, 1)) + rng.normal(0, 0.2, (6, nq)).astype(np.float32)
    q = np.clip(q, np.where(np.isfinite(lb), lb, -3), np.where(np.isfinite(ub), ub, 3)).astype(np.float32)
    kp = np.stack([orc.fk(x.copy())["site_xpos"].reshape(-1) for x in q]).astype(np.float32)
    kp = (kp + rng.normal(0, 2e-3, kp.shape)).astype(np.float32).reshape(3, 2, 3 * K)
    # two part groups: a random half of the coordinates, and the coordinates of one random joint
    part = np.zeros((2, nq), np.uint8)
    part[0] = rng.random(nq) < 0.5
    jr = int(rng.integers(t.njnt))
    part[1, int(t.jnt_qposadr[jr]):int(t.jnt_qposadr[jr]) + JNT_QPOS_DIMS[int(t.jnt_type[jr])]] = 1
    trunk = (rng.random(K) < 0.6).astype(np.uint8)
    trunk[0] = 1
    do_root = free_root
    ref = orc.ik_clips(kp, lb, ub, part, trunk, 0, 7, do_root_opt=do_root)
    for lanes in (8, 16, 0):
        eng = Engine(t, lb, ub, tol=1e-5, maxiter=12, lanes_per_chain=lanes)
        fk = eng.fk(q)
        for i in range(len(q)):
            o = orc.fk(q[i].copy())
            np.testing.assert_array_equal(_np(fk["xpos"][i]), o["xpos"])
            np.testing.assert_array_equal(_np(fk["site_xpos"][i]), o["site_xpos"])
        res = eng.q_phase(kp, part_masks=part, trunk_kps=trunk, root_kp_idx=0, root_dims=7, do_root_opt=do_root)
        _compare_phase(res, ref)


def _random_lean_case(seed, chains=6, frames=2, maxiter=10, wide=False, p_oriented=None):
    """One random model of the kind the lean kernels take (free root, hinges, no oriented body: split kinematics with
    host-scheduled passes -- restarts, bodies / joints without offset, pruned root programs) through every lean launch site,
    each launched twice, against the oracle at tolerance 0.  Returns how many of the launches ran a lean kernel."""
    from oracle import Oracle
    from stac_mjx_amd.engine import Engine
    from stac_mjx_amd.mjcf import JNT_FREE

    rng = np.random.default_rng((190000 if wide else 90000) + seed + (200000 if p_oriented else 0))
    nbody = int(rng.integers(75, 190)) if wide else int(rng.integers(3, (25, 45, 70)[seed % 3]))
    t = _random_tables(rng, nbody, True, p_slide=0.0, p_ball=0.0, max_children_bias=float(rng.choice([0.3, 0.6, 0.9, 0.97])), lean=True,
                       k_max=int(rng.choice([12, 30, 60]) if wide else rng.choice([6, 12, 30])), p_oriented=p_oriented)
    if wide:
        # the wide lean shapes (round 5: 32 lanes, eight solver registers per lane, two rounds of sites): 97 .. 256 coordinates; the
        # free root's own body may be oriented (a free joint sets its pose: mouse)
        if not 96 < t.nq <= 256:
            pytest.skip("not a model of the wide lean shapes")
        if seed % 2:
            qr = rng.normal(0, 1, 4)
            t.body_quat[1] = (qr / np.linalg.norm(qr)).astype(np.float32)
    elif t.nq > 80:  # (the narrow lean instantiations hold 80 coordinates at 16 lanes, 96 at 32)
        pytest.skip("more coordinates than the narrow lean kernels hold")
    nq, K = t.nq, t.nsite
    lb, ub = np.full(nq, -np.inf, np.float32), np.full(nq, np.inf, np.float32)
    for j in range(t.njnt):
        a, ty = int(t.jnt_qposadr[j]), int(t.jnt_type[j])
        if ty == JNT_FREE:
            lb[a + 3:a + 7], ub[a + 3:a + 7] = -1, 1
        else:
            lb[a], ub[a] = min(t.jnt_range[j, 0], 0.0), t.jnt_range[j, 1]
    tol = float(rng.choice([1e-5, 1e-3]))
    orc = Oracle(t, tol=tol, maxiter=maxiter)
    n = chains * frames
    q = np.tile(t.qpos0, (n, 1)) + rng.normal(0, 0.15, (n, nq)).astype(np.float32)
    q = np.clip(q, np.where(np.isfinite(lb), lb, -3), np.where(np.isfinite(ub), ub, 3)).astype(np.float32)
    kp = np.stack([orc.fk(x.copy())["site_xpos"].reshape(-1) for x in q]).astype(np.float32)
    kp = (kp + rng.normal(0, 2e-3, kp.shape)).astype(np.float32).reshape(chains, frames, 3 * K)
    P = int(rng.integers(0, 4))
    part = np.zeros((P, nq), np.uint8)
    for i in range(P):
        part[i] = rng.random(nq) < rng.choice([0.15, 0.5])
    trunk = (rng.random(K) < 0.6).astype(np.uint8)
    trunk[0] = 1
    kw = dict(part_masks=part, trunk_kps=trunk, root_kp_idx=0, root_dims=7, do_root_opt=True)
    ref = orc.ik_clips(kp, lb, ub, part, trunk, 0, 7, do_root_opt=True)
    lean_runs = 0
    for lanes in ((32, 0) if wide else (16, 32, 0)):
        eng = Engine(t, lb, ub, tol=tol, maxiter=maxiter, lanes_per_chain=lanes)
        res = _q_phase_twice(eng, kp, **kw)
        lean_runs += _last_q_kernel(eng)[3] & 1
        np.testing.assert_array_equal(res["qpos"].cpu().numpy().view(np.uint32), ref["qpos"].view(np.uint32), err_msg=f"seed {seed} lanes {lanes}")
        np.testing.assert_array_equal(res["counters"].cpu().numpy(), ref["counters"])
    return lean_runs


@pytest.mark.parametrize("seed", range(10))
def test_random_lean_models_bit_exact(seed):
    """Round 5: random trees of the lean kind (no generator of the other fuzz tests ever draws one: 40 % oriented bodies) -- the
    host's list scheduler of the split kinematics meets chains, bushes, zero offsets and root programs of every shape."""
    _random_lean_case(seed)


def test_random_wide_lean_models_bit_exact():
    """Round 5: random trees of 97 .. 256 coordinates (split-kinematics programs of a hundred steps and more, two rounds of sites,
    an oriented free-root body in every other one) at 32 lanes and in latency mode, each launched twice, equal to the oracle bit for
    bit -- and most of them in the wide lean shapes <32,8,2,1> / <32,8,2,9> (a chain deeper than 254 products stays generic)."""
    lean_runs = sum(_random_lean_case(seed, chains=(5, 40)[seed % 2], frames=1 + seed % 2, wide=True) for seed in range(6))
    assert lean_runs >= 8, lean_runs  # (two launches per model)


def test_random_lean_models_do_take_the_lean_kernels():
    assert sum(_random_lean_case(s) for s in (100, 101, 102)) >= 6  # (three launches per model)


@pytest.mark.parametrize("seed", range(10))
def test_random_lean_models_with_oriented_bodies_bit_exact(seed):
    """Round 6: an oriented body below the root (body_quat not the identity: the fruit fly's legs) is one more product of the
    quaternion pass, q_parent * body_quat, its right factor a constant that every chain region holds -- random trees in which 15 to
    60 % of the bodies are oriented, through every lean launch site, against the oracle at tolerance 0."""
    _random_lean_case(seed, p_oriented=(0.15, 0.4, 0.6)[seed % 3])


def test_random_oriented_lean_models_do_take_the_lean_kernels():
    assert sum(_random_lean_case(s, p_oriented=0.4) for s in (100, 101, 102)) >= 6  # (three launches per model)


def _random_model_case(seed, nbody_lo, nbody_hi, chains, frames, lanes_list, maxiter=10, q_init=False):
    """One random model through the q_phase at the given lane widths, each launched twice, against the oracle."""
    from oracle import Oracle
    from stac_mjx_amd.engine import Engine, StacHipError
    from stac_mjx_amd.mjcf import JNT_BALL, JNT_FREE, JNT_QPOS_DIMS

    rng = np.random.default_rng(50000 + seed)
    free_root = bool(rng.integers(2))
    t = _random_tables(rng, int(rng.integers(nbody_lo, nbody_hi + 1)), free_root, p_ball=float(rng.choice([0.0, 0.1])),
                       max_children_bias=float(rng.choice([0.05, 0.3, 0.6, 0.9, 0.97])))
    nq, K = t.nq, t.nsite
    if nq == 0:
        pytest.skip("no joints drawn")
    lb, ub = np.full(nq, -np.inf, np.float32), np.full(nq, np.inf, np.float32)
    for j in range(t.njnt):
        a, ty = int(t.jnt_qposadr[j]), int(t.jnt_type[j])
        if ty == JNT_FREE:
            lb[a + 3:a + 7], ub[a + 3:a + 7] = -1, 1
        elif ty == JNT_BALL:
            lb[a:a + 4], ub[a:a + 4] = -1, 1
        else:
            lb[a], ub[a] = min(t.jnt_range[j, 0], 0.0), t.jnt_range[j, 1]
    tol = float(rng.choice([1e-5, 1e-3]))
    orc = Oracle(t, tol=tol, maxiter=maxiter)
    n = chains * frames
    q = np.tile(t.qpos0, (n, 1)) + rng.normal(0, 0.15, (n, nq)).astype(np.float32)
    q = np.clip(q, np.where(np.isfinite(lb), lb, -3), np.where(np.isfinite(ub), ub, 3)).astype(np.float32)
    kp = np.stack([orc.fk(x.copy())["site_xpos"].reshape(-1) for x in q]).astype(np.float32)
    kp = (kp + rng.normal(0, 2e-3, kp.shape)).astype(np.float32).reshape(chains, frames, 3 * K)
    P = int(rng.integers(0, 4))
    part = np.zeros((P, nq), np.uint8)
    for i in range(P):
        part[i] = rng.random(nq) < rng.choice([0.15, 0.5])
    trunk = (rng.random(K) < 0.6).astype(np.uint8)
    trunk[0] = 1
    qi = q.reshape(chains, frames, nq)[:, 0].copy() if q_init else None
    kw = dict(part_masks=part, trunk_kps=trunk, root_kp_idx=0, root_dims=7, do_root_opt=free_root and not q_init)
    ref = orc.ik_clips(kp, lb, ub, part, trunk, 0, 7, do_root_opt=kw["do_root_opt"], q_init=qi)
    ran = 0
    for lanes in lanes_list:
        try:
            eng = Engine(t, lb, ub, tol=tol, maxiter=maxiter, lanes_per_chain=lanes)
            res = _q_phase_twice(eng, kp, q_init=qi, **kw)
        except StacHipError as e:  # a lane width that cannot hold this model (capacity) is refused, not mis-run
            assert "limits" in str(e) or "lanes" in str(e) or "capacity" in str(e).lower(), e
            continue
        _compare_phase(res, ref)
        if lanes == lanes_list[-1]:  # the solver seam (stac_q_solve = StacCore.q_opt) on the same model: random masks, a per-call box
            ns = min(n, 6)
            qs = (rng.random(nq) < 0.6).astype(np.uint8)
            ks = (rng.random(3 * K) < 0.8).astype(np.uint8)
            fin = np.isfinite(lb) & np.isfinite(ub)
            lb2, ub2 = lb.copy(), ub.copy()
            lb2[fin], ub2[fin] = lb[fin] * 0.5, ub[fin] * 0.5
            kpf = kp.reshape(n, 3 * K)[:ns]
            q0s = np.clip(q[:ns], np.where(np.isfinite(lb2), lb2, -3), np.where(np.isfinite(ub2), ub2, 3)).astype(np.float32)
            par, st, cn = eng.q_solve(kpf, q0s, qs, ks, lb=lb2, ub=ub2)
            for i in range(ns):
                xr, sr = orc.q_opt(kpf[i], qs, ks, q0s[i], lb2, ub2)
                np.testing.assert_array_equal(_np(par)[i], xr)
                assert int(_np(cn)[i, 0]) == sr["iter_num"] and float(_np(st)[i, 0]) == np.float32(sr["error"])
        eng.close()
        ran += 1
    assert ran > 0


@pytest.mark.parametrize("seed", range(4))
def test_random_models_large_trees_many_chains(seed):
    """Larger random trees (40-110 bodies: nq up to ~200, the wide instantiations), 70 chains x 2 frames (more chains than a
    workgroup holds; with a carried start pose on odd seeds), every lane width, each launch repeated: HIP == oracle bit for bit."""
    _random_model_case(seed, 40, 110, 70, 2, (8, 16, 32, 64, 0), q_init=bool(seed % 2))


# ---- boundary details ---------------------------------------------------------------------------------------------------
def test_q_solve_per_call_bounds(rodent_setup, rodent_mocap):
    """StacCore.q_opt takes lb / ub per call (stac_core.py:193-235, hyperparams_proj): a box passed to stac_q_solve
    overrides the model's for that call only, equals the oracle with the same box, and the next call without a box
    is back on the model's bounds."""
    fs = rodent_setup
    eng, orc = _engine(fs, maxiter=50), _oracle(fs, maxiter=50)
    kp, q0 = rodent_mocap[20:26], np.tile(fs.tables.qpos0, (6, 1))
    q0[:, :3] = kp[:, 3 * fs.root_kp_idx : 3 * fs.root_kp_idx + 3]
    ones_q, ones_k = np.ones(74, np.uint8), np.ones(69, np.uint8)
    lb2, ub2 = np.maximum(fs.lb, -0.05).astype(np.float32), np.minimum(fs.ub, 0.05).astype(np.float32)
    lb2[:7], ub2[:7] = fs.lb[:7], fs.ub[:7]
    base, _, _ = eng.q_solve(kp, q0, ones_q, ones_k)
    tight, st, cnt = eng.q_solve(kp, q0, ones_q, ones_k, lb=lb2, ub=ub2)
    again, _, _ = eng.q_solve(kp, q0, ones_q, ones_k)
    assert (base == again).all() and not (base == tight).all()
    tight = _np(tight)
    assert (tight[:, 7:] >= -0.05).all() and (tight[:, 7:] <= 0.05).all()
    for i in range(6):
        ref, rst = orc.q_opt(kp[i], ones_q, ones_k, q0[i], lb2, ub2)
        np.testing.assert_array_equal(tight[i], ref)
        assert int(_np(cnt)[i, 0]) == rst["iter_num"] and float(_np(st)[i, 0]) == np.float32(rst["error"])
        ref0, _ = orc.q_opt(kp[i], ones_q, ones_k, q0[i], fs.lb, fs.ub)
        np.testing.assert_array_equal(_np(base)[i], ref0)
    from stac_mjx_amd.engine import StacHipError

    with pytest.raises(StacHipError, match="lb > ub"):
        eng.q_solve(kp, q0, ones_q, ones_k, lb=ub2 + 1.0, ub=ub2)


def test_m_partial_with_zero_frames(rodent_setup):
    """T = 0 (a rank of a sharded fit without sampled frames): zero sums, T = 0, no invalid launch; the closed form on
    it keeps unregularised offsets where they were instead of dividing 0 by 0."""
    fs = rodent_setup
    eng, orc = _engine(fs), _oracle(fs)
    part = _np(eng.m_partial(np.zeros((0, 69), np.float32), np.zeros((0, 74), np.float32)))
    np.testing.assert_array_equal(part, np.zeros(71, np.float32))
    m0 = fs.tables.site_pos
    d = np.zeros((23, 3), np.float32)
    d[:5] = 1.0
    out, err = eng.m_finish(part, m0, d, 1.0)
    ref, rerr = orc.m_finish(part, m0, d, 1.0)
    np.testing.assert_array_equal(_np(out), ref)
    np.testing.assert_array_equal(_np(out), m0)
    assert float(err[0]) == rerr == 0.0


# ---- latency mode over 1, 4 and 8 wavefronts per chain ---------------------------------------------------------------
@pytest.mark.parametrize("specg", ["8", "16", "32", "64"])
def test_latency_mode_wavefronts_per_chain(rodent_setup, fly_setup, mouse_setup, rodent_mocap, monkeypatch, specg):
    """The speculative evaluations of a chain on one wavefront (eight roles of 8 lanes, or four of 16: the shape of the
    straggler kernel and of large clip counts), on four (32 lanes each) or on eight (64 lanes each; roles exchange
    accept flags, losses and two gradients through LDS, two workgroup barriers per trip): warm-started multi-frame
    clips of three models, small line-search bound included -- all equal the oracle bit for bit, and each other."""
    from oracle import Oracle
    from stac_mjx_amd.engine import Engine

    monkeypatch.setenv("STAC_HIP_SPEC", "1")
    monkeypatch.setenv("STAC_HIP_SPECG", specg)
    fs = rodent_setup
    kp = rodent_mocap[:24].reshape(3, 8, 69)
    for maxls in (15, 2):
        eng = Engine(fs.tables, fs.lb, fs.ub, tol=1e-4, maxiter=60, maxls=maxls)
        orc = Oracle(fs.tables, tol=1e-4, maxiter=60, maxls=maxls)
        res = _q_phase_twice(eng, kp, part_masks=fs.part_masks, trunk_kps=fs.trunk_kps, root_kp_idx=fs.root_kp_idx,
                             root_dims=fs.root_dims, do_root_opt=True)
        ref = orc.ik_clips(kp, fs.lb, fs.ub, fs.part_masks, fs.trunk_kps, fs.root_kp_idx, fs.root_dims)
        _compare_phase(res, ref)
        np.testing.assert_array_equal(_np(res["carry_qpos"]), ref["qpos"][:, -1])
    # fruit fly: oriented bodies (16-word FK records), no root optimisation
    fly = fly_setup
    engf, orcf = _engine(fly, tol=5e-3, maxiter=40), _oracle(fly, tol=5e-3, maxiter=40)
    rng = np.random.default_rng(21)
    qt = fly.tables.qpos0[None] + np.clip(rng.normal(0, 0.15, (6, 43)), -0.3, 0.3).astype(np.float32)
    qt[:, 3:7] = fly.tables.qpos0[3:7]
    kpf = np.stack([orcf.fk(q)["site_xpos"].reshape(-1) for q in qt]).reshape(2, 3, 90)
    kpf = kpf + rng.normal(0, 1e-3, kpf.shape).astype(np.float32)
    _compare_phase(_q_phase_twice(engf, kpf, part_masks=fly.part_masks),
                   orcf.ik_clips(kpf, fly.lb, fly.ub, fly.part_masks, fly.trunk_kps, 0, 7, do_root_opt=False))
    # mouse: nq = 230 (four solver registers per lane at 64 lanes), 85 tree levels
    ms = mouse_setup
    engm, orcm = _engine(ms, maxiter=12), _oracle(ms, maxiter=12)
    qm = ms.tables.qpos0[None] + np.clip(rng.normal(0, 0.05, (2, 230)), -0.1, 0.1).astype(np.float32)
    qm[:, 3:7] = ms.tables.qpos0[3:7]
    kpm = np.stack([orcm.fk(q)["site_xpos"].reshape(-1) for q in qm]).reshape(2, 1, 102)
    _compare_phase(_q_phase_twice(engm, kpm, part_masks=ms.part_masks, trunk_kps=ms.trunk_kps, root_kp_idx=ms.root_kp_idx,
                                  root_dims=ms.root_dims, do_root_opt=ms.do_root_opt),
                   orcm.ik_clips(kpm, ms.lb, ms.ub, ms.part_masks, ms.trunk_kps, ms.root_kp_idx, ms.root_dims,
                                 do_root_opt=ms.do_root_opt))


@pytest.mark.parametrize("specg", ["8", "16", "32", "64"])
def test_latency_mode_chain_queue(rodent_setup, rodent_mocap, monkeypatch, specg):
    """More clips than chain slots in latency mode (forced: 4 slots for 11 clips): the roles of a finished clip -- one
    wavefront, or all wavefronts of the workgroup together -- take the next unstarted clip."""
    monkeypatch.setenv("STAC_HIP_SPEC", "1")
    monkeypatch.setenv("STAC_HIP_SPECG", specg)
    monkeypatch.setenv("STAC_HIP_QUEUE", "4")
    fs = rodent_setup
    eng, orc = _engine(fs, maxiter=30), _oracle(fs, maxiter=30)
    kp = rodent_mocap[100:133].reshape(11, 3, 69)
    res = eng.q_phase(kp, part_masks=fs.part_masks, trunk_kps=fs.trunk_kps, root_kp_idx=fs.root_kp_idx,
                      root_dims=fs.root_dims, do_root_opt=True)
    ref = orc.ik_clips(kp, fs.lb, fs.ub, fs.part_masks, fs.trunk_kps, fs.root_kp_idx, fs.root_dims)
    _compare_phase(res, ref)
    np.testing.assert_array_equal(_np(res["carry_qpos"]), ref["qpos"][:, -1])


def test_latency_mode_four_roles_per_chain(rodent_setup, rodent_mocap, monkeypatch):
    """STAC_HIP_SPECR=4 (developer switch; never auto-selected): two candidates + their momentum points per trip, two
    chains per wavefront; iterations that accept neither of the first two candidates take a second trip with candidates
    2 and 3.  Same answers as the oracle, with the chain queue on top."""
    monkeypatch.setenv("STAC_HIP_SPEC", "1")
    monkeypatch.setenv("STAC_HIP_SPECG", "8")
    monkeypatch.setenv("STAC_HIP_SPECR", "4")
    monkeypatch.setenv("STAC_HIP_QUEUE", "4")
    fs = rodent_setup
    kp = rodent_mocap[700:733].reshape(11, 3, 69)
    for maxls in (15, 3):
        eng, orc = _engine(fs, maxiter=40, maxls=maxls), _oracle(fs, maxiter=40, maxls=maxls)
        res = eng.q_phase(kp, part_masks=fs.part_masks, trunk_kps=fs.trunk_kps, root_kp_idx=fs.root_kp_idx,
                          root_dims=fs.root_dims, do_root_opt=True)
        _compare_phase(res, orc.ik_clips(kp, fs.lb, fs.ub, fs.part_masks, fs.trunk_kps, fs.root_kp_idx, fs.root_dims))


# ---- round 3: root fast trips, wave-synchronous root phase, chain order + placement by SIMD load ------------------------------
@pytest.mark.parametrize("lanes", [8, 16, 32])
@pytest.mark.parametrize("nofast", [None, "1"])
def test_root_fast_trips_with_carried_poses_outside_the_box(rodent_setup, rodent_mocap, monkeypatch, lanes, nofast):
    """Root optimisation from given start poses (q_init), some of whose NON-root coordinates lie outside their bounds: for
    those chains the projected gradient moves unmasked coordinates in its first iteration (clip(y) != y), so their terms of
    the solver's norms are not zero and the root fast trip must not be taken (tail_ok); the other chains of the same
    wavefront take it.  Everything equals the oracle, with and without the fast path."""
    if nofast:
        monkeypatch.setenv("STAC_HIP_NOFAST", nofast)
    fs = rodent_setup
    eng, orc = _engine(fs, lanes_per_chain=lanes, maxiter=60), _oracle(fs, maxiter=60)
    kp = rodent_mocap[300:322].reshape(22, 1, 69)
    rng = np.random.default_rng(3)
    q_init = np.tile(fs.tables.qpos0, (22, 1)).astype(np.float32)
    q_init[:, 7:] += rng.normal(0, 0.05, (22, 67)).astype(np.float32)
    q_init = np.clip(q_init, np.where(np.isfinite(fs.lb), fs.lb, -10), np.where(np.isfinite(fs.ub), fs.ub, 10)).astype(np.float32)
    fin = np.flatnonzero(np.isfinite(fs.ub[7:])) + 7
    for c in (1, 6, 7, 13, 21):  # chains with one or two coordinates beyond their upper / lower bound
        q_init[c, fin[(3 * c) % len(fin)]] = fs.ub[fin[(3 * c) % len(fin)]] + 0.25
        q_init[c, fin[(5 * c + 1) % len(fin)]] = fs.lb[fin[(5 * c + 1) % len(fin)]] - 0.1
    res = eng.q_phase(kp, part_masks=fs.part_masks, trunk_kps=fs.trunk_kps, root_kp_idx=fs.root_kp_idx, root_dims=fs.root_dims,
                      do_root_opt=True, q_init=q_init)
    ref = orc.ik_clips(kp, fs.lb, fs.ub, fs.part_masks, fs.trunk_kps, fs.root_kp_idx, fs.root_dims, q_init=q_init)
    _compare_phase(res, ref)


@pytest.mark.parametrize("C", [8400, 4600])
def test_chain_order_and_placement_by_simd_load(rodent_setup, rodent_mocap, monkeypatch, C):
    """A batch large enough (2 100 wavefronts on 1 024 SIMDs: SIMDs with three against SIMDs with two; 1 150: two against
    one) for the launch to order its chains by expected length and let every wavefront pick its chains by the load of its
    SIMD (HW_ID count + bounded spin barrier): no result depends on where a chain runs -- identical to the launch without the
    order, run to run, and equal to the oracle on sampled chains."""
    fs = rodent_setup
    rng = np.random.default_rng(17)
    base = rodent_mocap[rng.integers(0, 1000, C)]
    kp = (base + rng.normal(0, 2e-3, base.shape)).astype(np.float32).reshape(C, 1, 69)
    args = dict(part_masks=fs.part_masks, trunk_kps=fs.trunk_kps, root_kp_idx=fs.root_kp_idx, root_dims=fs.root_dims,
                do_root_opt=True, want_bodies=False)
    eng = _engine(fs, maxiter=20)
    a = eng.q_phase(kp, **args)
    b = eng.q_phase(kp, **args)
    for k in ("qpos", "frame_error", "counters", "marker_sites", "carry_qpos"):
        assert (a[k] == b[k]).all(), k
    monkeypatch.setenv("STAC_HIP_NOORDER", "1")
    plain = _engine(fs, maxiter=20).q_phase(kp, **args)
    for k in ("qpos", "frame_error", "counters", "marker_sites", "carry_qpos"):
        assert (a[k] == plain[k]).all(), k
    sel = [0, 1, 777, 4099, 4100, C - 1]
    ref = _oracle(fs, maxiter=20).ik_clips(kp[sel], fs.lb, fs.ub, fs.part_masks, fs.trunk_kps, fs.root_kp_idx, fs.root_dims, want_bodies=False)
    for i, c in enumerate(sel):
        np.testing.assert_array_equal(_np(a["qpos"][c]), ref["qpos"][i])
        np.testing.assert_array_equal(_np(a["counters"][c]).astype(np.uint32), ref["counters"][i])


# ---- round 4: every shipped instantiation, twice -------------------------------------------------------------------------------
# stac_kernels.hip, STAC_Q_SHAPES / STAC_Q_SPEC_SHAPES and stac_lm.hip: (lanes, registers per lane, register cap / roles).  Each
# case forces one instantiation through the developer switches, launches it twice on the same engine and compares with the
# oracle (PG: bit for bit; LM: finite, repeatable, inside the box, marker error no worse than the PG answer).  The list of
# kernels this suite launches on a GPU is committed as profiles/r04/gpu_suite_kernels.txt; tests/test_isa_hazards.py checks on
# the CPU that every q_phase instantiation of the built library is in it.
def _mid_model(seed=5000, nbody=90, nq_expected=119):
    """A random tree (free root, hinges and slides) with nq = 119: the 16-registers-at-8-lanes, 8-at-16 and 4-at-32 shapes; or
    ("big": seed 5001, 130 bodies) with nq = 179 and few sites, which the LM kernel's wide instantiations hold (the mouse's
    LM matrices do not fit the LDS)."""
    from stac_mjx_amd.mjcf import JNT_FREE

    rng = np.random.default_rng(seed)
    t = _random_tables(rng, nbody, True, p_ball=0.0, max_children_bias=0.6)
    assert t.nq == nq_expected
    lb, ub = np.full(t.nq, -np.inf, np.float32), np.full(t.nq, np.inf, np.float32)
    for j in range(t.njnt):
        a, ty = int(t.jnt_qposadr[j]), int(t.jnt_type[j])
        if ty == JNT_FREE:
            lb[a + 3:a + 7], ub[a + 3:a + 7] = -1, 1
        else:
            lb[a], ub[a] = min(t.jnt_range[j, 0], 0.0), t.jnt_range[j, 1]
    part = np.zeros((2, t.nq), np.uint8)
    part[0] = rng.random(t.nq) < 0.3
    part[1] = rng.random(t.nq) < 0.15
    trunk = (rng.random(t.nsite) < 0.6).astype(np.uint8)
    trunk[0] = 1
    return t, lb, ub, part, trunk, rng


_SHAPE_CASES = [
    # (model, lanes, solver, environment)                                       instantiation
    ("rodent", 8, "pg", {}),                                                    # q<8,10,2,0>
    ("rodent", 16, "pg", {"STAC_HIP_WPE": "2"}),                                # q<16,5,2,1>  (lean)
    ("rodent", 16, "pg", {"STAC_HIP_WPE": "3"}),                                # q<16,5,3,1>
    ("rodent", 16, "pg", {"STAC_HIP_WPE": "2", "STAC_HIP_NOLEAN": "1"}),        # q<16,5,2,0>  (generic)
    ("rodent", 16, "pg", {"STAC_HIP_WPE": "3", "STAC_HIP_NOLEAN": "1"}),        # q<16,5,3,0>
    ("rodent", 32, "pg", {"STAC_HIP_WPE": "2"}),                                # q<32,3,2,1>  (lean)
    ("rodent", 32, "pg", {"STAC_HIP_WPE": "2", "STAC_HIP_NOLEAN": "1"}),        # q<32,3,2,0>  (generic)
    ("rodent", 32, "pg", {"STAC_HIP_WPE": "4", "STAC_HIP_NOLEAN": "1"}),        # q<32,3,4,0>  (generic only: a lean launch keeps its own variants)
    ("rodent", 64, "pg", {"STAC_HIP_WPE": "2"}),                                # q<64,2,2,0>
    ("rodent", 64, "pg", {"STAC_HIP_WPE": "4"}),                                # q<64,2,4,0>
    ("rodent", 0, "pg", {"STAC_HIP_SPEC": "1", "STAC_HIP_SPECG": "8", "STAC_HIP_SPECR": "4"}),   # q<8,10,2,4>
    ("rodent", 0, "pg", {"STAC_HIP_SPEC": "1", "STAC_HIP_SPECG": "8"}),         # q<8,10,2,8>
    ("rodent", 0, "pg", {"STAC_HIP_SPEC": "1", "STAC_HIP_SPECG": "16"}),        # q<16,5,2,5>  (lean)
    ("rodent", 0, "pg", {"STAC_HIP_SPEC": "1", "STAC_HIP_SPECG": "32"}),        # q<32,3,2,9>
    ("rodent", 0, "pg", {"STAC_HIP_SPEC": "1", "STAC_HIP_SPECG": "16", "STAC_HIP_SPECR": "8"}),   # q<16,5,2,9>  (lean: two wavefronts per chain)
    ("rodent", 0, "pg", {"STAC_HIP_SPEC": "1", "STAC_HIP_SPECG": "16", "STAC_HIP_NOLEAN": "1"}),   # q<16,5,2,4>  (generic)
    ("rodent", 0, "pg", {"STAC_HIP_SPEC": "1", "STAC_HIP_SPECG": "32", "STAC_HIP_NOLEAN": "1"}),   # q<32,3,2,8>
    ("rodent", 0, "pg", {"STAC_HIP_SPEC": "1", "STAC_HIP_SPECG": "64"}),        # q<64,2,2,8>
    ("mid", 8, "pg", {}),                                                       # q<8,16,2,0>
    ("mid", 16, "pg", {"STAC_HIP_WPE": "2"}),                                   # q<16,8,2,0>
    ("mid", 16, "pg", {"STAC_HIP_WPE": "3"}),                                   # q<16,8,3,0>
    ("mid", 32, "pg", {"STAC_HIP_WPE": "2"}),                                   # q<32,4,2,0>
    ("mid", 0, "pg", {"STAC_HIP_SPEC": "1", "STAC_HIP_SPECG": "16"}),           # q<16,8,2,4>
    ("mouse", 16, "pg", {}),                                                    # q<16,16,2,0>
    ("mouse", 32, "pg", {}),                                                    # q<32,8,2,1>  (lean: the wide shape, two rounds of sites)
    ("mouse", 32, "pg", {"STAC_HIP_NOLEAN": "1"}),                              # q<32,8,2,0>  (generic)
    ("mouse", 64, "pg", {"STAC_HIP_WPE": "2"}),                                 # q<64,4,2,0>
    ("mouse", 0, "pg", {"STAC_HIP_SPEC": "1", "STAC_HIP_SPECG": "32"}),         # q<32,8,2,9>  (lean)
    ("mouse", 0, "pg", {"STAC_HIP_SPEC": "1", "STAC_HIP_SPECG": "32", "STAC_HIP_NOLEAN": "1"}),   # q<32,8,2,8>  (generic)
    ("mouse", 0, "pg", {"STAC_HIP_SPEC": "1", "STAC_HIP_SPECG": "64"}),         # q<64,4,2,8>
    ("fly", 16, "pg", {"STAC_HIP_WPE": "3"}),                                   # q<16,3,3,1>  (lean: the narrow shapes of models up to 48 / 64 coordinates)
    ("fly", 0, "pg", {"STAC_HIP_SPEC": "1", "STAC_HIP_SPECG": "16"}),           # q<16,3,2,5>
    ("fly", 0, "pg", {"STAC_HIP_SPEC": "1", "STAC_HIP_SPECG": "32"}),           # q<32,2,2,9>
    ("rodent", 16, "lm", {}), ("rodent", 32, "lm", {}), ("rodent", 64, "lm", {}),     # lm<16,5,2> lm<32,3,2> lm<64,2,3>
    ("mid", 16, "lm", {}),                                                      # lm<16,8,2>
    ("big", 16, "lm", {}), ("big", 32, "lm", {}), ("big", 64, "lm", {}),         # lm<16,16,2> lm<32,8,2> lm<64,4,2>
]


def _last_q_kernel(eng):
    import ctypes

    out = (ctypes.c_int32 * 4)()
    eng.lib.stac_debug_last_q_kernel(out)
    return tuple(out)


def test_default_launches_of_the_rodent_take_the_lean_kernels(rodent_setup, rodent_mocap, monkeypatch):
    """The shapes the bench and `ik_only` run the rodent in exist as lean kernels (SPECP bit 0); the host must pick them on its
    own at every one of its three launch sites -- throughput launch, latency launch, hand-off resume -- and must not when
    STAC_HIP_NOLEAN asks for the generic ones (a launch site that forgets the hinge flag falls back silently: -13 %)."""
    from stac_mjx_amd.engine import Engine

    fs = rodent_setup
    kw = dict(part_masks=fs.part_masks, trunk_kps=fs.trunk_kps, root_kp_idx=fs.root_kp_idx, root_dims=fs.root_dims, do_root_opt=True)
    eng = Engine(fs.tables, fs.lb, fs.ub, tol=1e-4, maxiter=6)
    eng.q_phase(np.tile(rodent_mocap[300:306], (1000, 1)).reshape(6000, 1, 69), **kw)  # chain queue + hand-off: the resume launch is last
    assert _last_q_kernel(eng) == (16, 5, 2, 5)
    eng.q_phase(rodent_mocap[300:312].reshape(4, 3, 69), **kw)  # few chains: latency mode, four wavefronts per chain
    assert _last_q_kernel(eng) == (32, 3, 2, 9)
    eng16 = Engine(fs.tables, fs.lb, fs.ub, tol=1e-4, maxiter=6, lanes_per_chain=16)
    eng16.q_phase(rodent_mocap[300:310].reshape(10, 1, 69), **kw)  # throughput kernel on request, no hand-off at this size
    g, nqr, wpe, specp = _last_q_kernel(eng16)
    assert (g, nqr, specp) == (16, 5, 1) and wpe in (2, 3)
    monkeypatch.setenv("STAC_HIP_NOLEAN", "1")
    gen = Engine(fs.tables, fs.lb, fs.ub, tol=1e-4, maxiter=6)
    gen.q_phase(rodent_mocap[300:312].reshape(4, 3, 69), **kw)
    assert _last_q_kernel(gen) == (32, 3, 2, 8)


def test_split_kinematics_and_step_program_agree_bit_for_bit(rodent_setup, rodent_mocap, monkeypatch):
    """Round 5: the lean kernels run the kinematics as three passes (quaternion chain, rotations, position chain:
    PlanHeader::fk3) -- the same operations in the same order per result as the generic kernels' step program.  Every launch site
    (throughput, latency with four wavefronts per chain, one-wavefront latency / hand-off) must give the generic kernels' bits
    (STAC_HIP_NOFK3: no split-kinematics tables, hence no lean kernel), on root passes (pruned programs) and pose passes alike."""
    from stac_mjx_amd.engine import Engine

    fs = rodent_setup
    kw = dict(part_masks=fs.part_masks, trunk_kps=fs.trunk_kps, root_kp_idx=fs.root_kp_idx, root_dims=fs.root_dims, do_root_opt=True)
    cases = [("throughput + hand-off", np.tile(rodent_mocap[300:306], (800, 1)).reshape(4800, 1, 69), {}, (16, 5, 2, 5)),
             ("latency, four wavefronts per chain", rodent_mocap[300:312].reshape(4, 3, 69), {}, (32, 3, 2, 9)),
             ("throughput on request", rodent_mocap[300:340].reshape(20, 2, 69), dict(lanes_per_chain=16), None)]
    lean = {}
    for name, kp, ekw, shape in cases:
        eng = Engine(fs.tables, fs.lb, fs.ub, tol=1e-4, maxiter=8, **ekw)
        lean[name] = eng.q_phase(kp, **kw)
        got = _last_q_kernel(eng)
        assert got[3] & 1, (name, got)  # a lean kernel ran
        if shape is not None:
            assert got == shape, (name, got)
    monkeypatch.setenv("STAC_HIP_NOFK3", "1")
    for name, kp, ekw, shape in cases:
        eng = Engine(fs.tables, fs.lb, fs.ub, tol=1e-4, maxiter=8, **ekw)
        ref = eng.q_phase(kp, **kw)
        assert not (_last_q_kernel(eng)[3] & 1), name  # the generic kernel
        for key in ("qpos", "frame_error", "counters"):
            a, b = lean[name][key].cpu().numpy(), ref[key].cpu().numpy()
            assert np.array_equal(a.view(np.uint32) if a.dtype == np.float32 else a, b.view(np.uint32) if b.dtype == np.float32 else b), (name, key)


def test_solves_longer_than_the_momentum_table(rodent_setup, rodent_mocap, monkeypatch):
    """Round 5: the throughput kernels read FISTA's t_(k+1) and (t_k - 1) / t_(k+1) from a table in the plan for the first 256
    iterations of a solve and compute them beyond it; solves that never reach the tolerance run through both (280 iterations each)."""
    from oracle import Oracle
    from stac_mjx_amd.engine import Engine

    fs = rodent_setup
    monkeypatch.setenv("STAC_HIP_SPEC", "0")
    kp = rodent_mocap[500:506].reshape(3, 2, 69)
    kw = dict(part_masks=fs.part_masks, trunk_kps=fs.trunk_kps, root_kp_idx=fs.root_kp_idx, root_dims=fs.root_dims, do_root_opt=True)
    orc = Oracle(fs.tables, tol=1e-12, maxiter=280)
    ref = orc.ik_clips(kp, fs.lb, fs.ub, fs.part_masks, fs.trunk_kps, fs.root_kp_idx, fs.root_dims)
    assert ref["counters"][..., 0].max() >= 6 * 280  # every solve of a frame ran to the bound
    for lanes, env in ((16, {}), (16, {"STAC_HIP_NOLEAN": "1"}), (32, {})):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        eng = Engine(fs.tables, fs.lb, fs.ub, tol=1e-12, maxiter=280, lanes_per_chain=lanes)
        _compare_phase(_q_phase_twice(eng, kp, **kw), ref)
        eng.close()


def test_default_launches_of_the_mouse_take_the_wide_lean_kernels(mouse_setup):
    """Round 5: a model too wide for the 16-lane lean shapes (mouse: 230 coordinates, an oriented free-root body) runs the split
    kinematics in the 32-lane shapes with eight solver registers per lane, at every batch size, by the host's own choice -- and equals
    the oracle there (the generic kernels spent 79 % of a trip in the step program of its 84-product spine)."""
    from oracle import Oracle
    from stac_mjx_amd.engine import Engine

    fs = mouse_setup
    t = fs.tables
    orc = Oracle(t, tol=1e-4, maxiter=10)
    rng = np.random.default_rng(5)
    qm = t.qpos0[None] + np.clip(rng.normal(0, 0.05, (4, t.nq)), -0.1, 0.1).astype(np.float32)
    qm[:, 3:7] = t.qpos0[3:7]
    kp4 = np.stack([orc.fk(x)["site_xpos"].reshape(-1) for x in qm]).astype(np.float32)
    kw = dict(part_masks=fs.part_masks, trunk_kps=fs.trunk_kps, root_kp_idx=fs.root_kp_idx, root_dims=fs.root_dims, do_root_opt=fs.do_root_opt)
    eng = Engine(t, fs.lb, fs.ub, tol=1e-4, maxiter=10)
    big = eng.q_phase(np.tile(kp4, (750, 1)).reshape(3000, 1, -1), **kw)      # throughput launch
    assert _last_q_kernel(eng)[0] == 32 and _last_q_kernel(eng)[1] == 8 and _last_q_kernel(eng)[3] & 1
    few = eng.q_phase(kp4.reshape(2, 2, -1), **kw)                              # latency launch
    assert _last_q_kernel(eng) == (32, 8, 2, 9)
    ref1 = orc.ik_clips(kp4.reshape(4, 1, -1), fs.lb, fs.ub, fs.part_masks, fs.trunk_kps, fs.root_kp_idx, fs.root_dims, do_root_opt=fs.do_root_opt)
    np.testing.assert_array_equal(_np(big["qpos"])[:4].view(np.uint32), ref1["qpos"].view(np.uint32))
    np.testing.assert_array_equal(_np(big["qpos"])[2996:].view(np.uint32), ref1["qpos"].view(np.uint32))
    _compare_phase(few, orc.ik_clips(kp4.reshape(2, 2, -1), fs.lb, fs.ub, fs.part_masks, fs.trunk_kps, fs.root_kp_idx, fs.root_dims, do_root_opt=fs.do_root_opt))


def test_default_launches_of_the_fly_take_the_lean_kernels(fly_setup, monkeypatch):
    """Round 6: BASELINE configs[4]'s model (fruit fly: 24 oriented leg bodies, a free root that is not optimised -- tethered) runs the
    split kinematics by the host's own choice at every launch site, equals the oracle there, and equals the generic kernels'
    bits (STAC_HIP_NOFK3BQ: no split-kinematics tables for a model with oriented bodies, as before round 6)."""
    from oracle import Oracle
    from stac_mjx_amd.engine import Engine

    fs = fly_setup
    t = fs.tables
    orc = Oracle(t, tol=5e-3, maxiter=12)
    rng = np.random.default_rng(21)
    qt = t.qpos0[None] + np.clip(rng.normal(0, 0.15, (8, t.nq)), -0.3, 0.3).astype(np.float32)
    qt[:, 3:7] = t.qpos0[3:7]
    kp8 = np.stack([orc.fk(q)["site_xpos"].reshape(-1) for q in qt]).astype(np.float32)
    kp8 = (kp8 + rng.normal(0, 1e-3, kp8.shape)).astype(np.float32)
    kw = dict(part_masks=fs.part_masks)
    okw = dict(do_root_opt=False)
    eng = Engine(t, fs.lb, fs.ub, tol=5e-3, maxiter=12)
    big = eng.q_phase(np.tile(kp8, (750, 1)).reshape(6000, 1, -1), **kw)  # throughput launch (+ hand-off)
    g, nqr, wpe, specp = _last_q_kernel(eng)
    assert g == 16 and specp & 1, (g, nqr, wpe, specp)
    few = _q_phase_twice(eng, kp8.reshape(4, 2, -1), **kw)                 # latency launch
    assert _last_q_kernel(eng)[3] & 1 and _last_q_kernel(eng)[3] & ~1, _last_q_kernel(eng)
    ref1 = orc.ik_clips(kp8.reshape(8, 1, -1), fs.lb, fs.ub, fs.part_masks, fs.trunk_kps, 0, 7, **okw)
    np.testing.assert_array_equal(_np(big["qpos"])[:8].view(np.uint32), ref1["qpos"].view(np.uint32))
    np.testing.assert_array_equal(_np(big["qpos"])[5992:].view(np.uint32), ref1["qpos"].view(np.uint32))
    _compare_phase(few, orc.ik_clips(kp8.reshape(4, 2, -1), fs.lb, fs.ub, fs.part_masks, fs.trunk_kps, 0, 7, **okw))
    eng16 = Engine(t, fs.lb, fs.ub, tol=5e-3, maxiter=12, lanes_per_chain=16)
    thr = _q_phase_twice(eng16, kp8.reshape(4, 2, -1), **kw)                # throughput kernel on request
    assert _last_q_kernel(eng16)[0] == 16 and _last_q_kernel(eng16)[3] == 1
    _compare_phase(thr, orc.ik_clips(kp8.reshape(4, 2, -1), fs.lb, fs.ub, fs.part_masks, fs.trunk_kps, 0, 7, **okw))
    monkeypatch.setenv("STAC_HIP_NOFK3BQ", "1")
    gen = Engine(t, fs.lb, fs.ub, tol=5e-3, maxiter=12)
    gbig = gen.q_phase(np.tile(kp8, (750, 1)).reshape(6000, 1, -1), **kw)
    assert not (_last_q_kernel(gen)[3] & 1)
    for key in ("qpos", "frame_error", "counters"):
        a, b = _np(big[key]), _np(gbig[key])
        assert np.array_equal(a.view(np.uint32) if a.dtype == np.float32 else a, b.view(np.uint32) if b.dtype == np.float32 else b), key


@pytest.mark.parametrize("model,lanes,solver,env", _SHAPE_CASES, ids=lambda v: str(v).replace(" ", "") if not isinstance(v, dict) else
                         "-".join(f"{k[9:]}{x}" for k, x in v.items()) or "auto")
def test_every_shipped_instantiation_twice(rodent_setup, mouse_setup, fly_setup, rodent_mocap, monkeypatch, model, lanes, solver, env):
    from oracle import Oracle
    from stac_mjx_amd.engine import Engine

    for k, v in env.items():
        monkeypatch.setenv(k, v)
    maxiter = 14
    if model in ("mid", "big"):
        t, lb, ub, part, trunk, rng = _mid_model() if model == "mid" else _mid_model(5001, 130, 179)
        orc = Oracle(t, tol=1e-5, maxiter=maxiter)
        q = np.tile(t.qpos0, (10, 1)) + rng.normal(0, 0.15, (10, t.nq)).astype(np.float32)
        q = np.clip(q, np.where(np.isfinite(lb), lb, -3), np.where(np.isfinite(ub), ub, 3)).astype(np.float32)
        kp = np.stack([orc.fk(x.copy())["site_xpos"].reshape(-1) for x in q]).astype(np.float32)
        kp = (kp + rng.normal(0, 2e-3, kp.shape)).astype(np.float32).reshape(5, 2, 3 * t.nsite)
        kw = dict(part_masks=part, trunk_kps=trunk, root_kp_idx=0, root_dims=7, do_root_opt=True)
        okw = dict(do_root_opt=True)
        tol = 1e-5
    else:
        fs = rodent_setup if model == "rodent" else (fly_setup if model == "fly" else mouse_setup)
        t, lb, ub, part, trunk, tol = fs.tables, fs.lb, fs.ub, fs.part_masks, fs.trunk_kps, 1e-4
        orc = Oracle(t, tol=tol, maxiter=maxiter)
        if model == "rodent":
            kp = rodent_mocap[300:310].reshape(5, 2, 69)
            do_root = True
        elif model == "fly":  # (tethered: no root optimisation)
            rng = np.random.default_rng(78)
            qm = t.qpos0[None] + np.clip(rng.normal(0, 0.15, (10, t.nq)), -0.3, 0.3).astype(np.float32)
            qm[:, 3:7] = t.qpos0[3:7]
            kp = np.stack([orc.fk(x)["site_xpos"].reshape(-1) for x in qm]).reshape(5, 2, 3 * t.nsite).astype(np.float32)
            do_root = False
        else:
            rng = np.random.default_rng(77)
            qm = t.qpos0[None] + np.clip(rng.normal(0, 0.05, (6, t.nq)), -0.1, 0.1).astype(np.float32)
            qm[:, 3:7] = t.qpos0[3:7]
            kp = np.stack([orc.fk(x)["site_xpos"].reshape(-1) for x in qm]).reshape(3, 2, 3 * t.nsite).astype(np.float32)
            do_root = fs.do_root_opt
        kw = dict(part_masks=part, trunk_kps=trunk, root_kp_idx=max(fs.root_kp_idx, 0), root_dims=fs.root_dims, do_root_opt=do_root)
        okw = dict(do_root_opt=do_root)
    rk, rd = kw["root_kp_idx"], kw["root_dims"]
    if solver == "pg":
        eng = Engine(t, lb, ub, tol=tol, maxiter=maxiter, lanes_per_chain=lanes)
        res = _q_phase_twice(eng, kp, **kw)
        _compare_phase(res, orc.ik_clips(kp, lb, ub, part, trunk, rk, rd, **okw))
    else:
        eng = Engine(t, lb, ub, tol=tol, solver="lm", lm_maxiter=8, lanes_per_chain=lanes)
        a, b = eng.q_phase(kp, **kw), eng.q_phase(kp, **kw)
        for k in ("qpos", "frame_error", "counters", "marker_sites"):
            assert (a[k] == b[k]).all(), k
        qo, mk = _np(a["qpos"]), _np(a["marker_sites"])
        assert np.isfinite(qo).all()
        if model == "rodent":  # (random models: coordinates without a marker below them keep a rest value that may lie outside the box)
            assert (qo >= lb - 1e-6).all() and (qo <= ub + 1e-6).all()
        pg = orc.ik_clips(kp, lb, ub, part, trunk, rk, rd, **okw)
        tgt = kp.reshape(mk.shape)
        assert np.linalg.norm(mk - tgt, axis=-1).mean() <= np.linalg.norm(pg["marker_sites"] - tgt, axis=-1).mean() + 1e-4
    eng.close()
