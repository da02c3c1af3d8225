"""The CPU oracle against the reference's golden vectors and known-answer tests (CPU only).

Pins: FK vs the stored reference output demos/demo_viz.p (tests/golden/demo_viz_golden.npz);
m_opt vs the known-answer cases of the reference's tests/unit/test_m_opt.py:72-225.
q_opt has no reference-side numeric pin: covered by gradient checks, optimality
properties and a self-regression fixture.
"""

import numpy as np
import pytest

from oracle import Oracle


@pytest.fixture(scope="module")
def orc_legacy(rodent_setup_legacy, demo_viz):
    o = Oracle(rodent_setup_legacy.tables)
    o.set_site_pos(demo_viz["offsets"])
    return o


@pytest.fixture(scope="module")
def orc(rodent_setup):
    return Oracle(rodent_setup.tables, tol=1e-4, maxiter=400)


# ---- FK known-answer test (reference's own stored output) ---------------------------------
def test_fk_matches_stored_reference_output(orc_legacy, demo_viz):
    ex = es = 0.0
    for f in range(demo_viz["qpos"].shape[0]):
        r = orc_legacy.fk(demo_viz["qpos"][f])
        ex = max(ex, np.abs(r["xpos"] - demo_viz["xpos"][f]).max())
        es = max(es, np.abs(r["site_xpos"] - demo_viz["walker_body_sites"][f]).max())
    assert ex <= 5e-7 and es <= 5e-7, (ex, es)


def test_fk_f32_vs_f64_twin(rodent_setup_legacy, demo_viz):
    o32, o64 = Oracle(rodent_setup_legacy.tables), Oracle(rodent_setup_legacy.tables, precision="f64")
    r32, r64 = o32.fk(demo_viz["qpos"][7]), o64.fk(demo_viz["qpos"][7])
    np.testing.assert_allclose(r32["xpos"], r64["xpos"], atol=3e-7)
    np.testing.assert_allclose(r32["xquat"], r64["xquat"], atol=5e-7)


def test_fk_normalises_and_writes_back_root_quaternion(orc, rodent_setup):
    q = rodent_setup.tables.qpos0.copy()
    q[3:7] = [2.0, 0.0, 0.0, 0.0]
    r = orc.fk(q)
    np.testing.assert_allclose(r["qpos"][3:7], [1, 0, 0, 0])
    np.testing.assert_allclose(r["xquat"][1], [1, 0, 0, 0])


def test_toy_fk_closed_form(toy_tables):
    o = Oracle(toy_tables)
    r = o.fk(np.array([np.pi / 2, 0.0, 0.0], np.float32))  # b1 turned 90 deg about z
    np.testing.assert_allclose(r["xpos"][1], [1, 0, 0], atol=1e-6)
    np.testing.assert_allclose(r["xpos"][2], [0, 0, 0], atol=1e-6)  # (1,0,0) + Rz(90)(0,1,0)
    np.testing.assert_allclose(r["xpos"][3], [0, 0, 1], atol=1e-6)
    np.testing.assert_allclose(r["site_xpos"][0], [1 - 0.2, 0.1, 0.3], atol=1e-6)


# ---- q_loss and its analytic gradient -----------------------------------------------------------
def _fd_grad(o64, q, kp, qs, ks, h=1e-4):
    g = np.zeros(q.size)
    for i in range(q.size):
        qp, qm = q.copy(), q.copy()
        qp[i] += h
        qm[i] -= h
        g[i] = (o64.q_loss(qp, kp, qs, ks, q, False)[0] - o64.q_loss(qm, kp, qs, ks, q, False)[0]) / (
            float(qp[i]) - float(qm[i]))
    return g


@pytest.mark.parametrize("which", ["all", "part", "trunk"])
def test_gradient_matches_finite_differences(rodent_setup_legacy, demo_viz, which):
    fs = rodent_setup_legacy
    o64, o32 = Oracle(fs.tables, precision="f64"), Oracle(fs.tables)
    for o in (o64, o32):
        o.set_site_pos(demo_viz["offsets"])
    rng = np.random.default_rng(0)
    q = demo_viz["qpos"][3] + rng.normal(0, 0.05, 74).astype(np.float32)
    q[3:7] *= 1.3  # deliberately non-unit root quaternion
    kp = demo_viz["kp_data"][3]
    qs = np.ones(74, bool) if which != "part" else fs.part_masks[0]
    ks = np.ones(69, bool) if which != "trunk" else np.repeat(fs.trunk_kps, 3)
    loss, g = o64.q_loss(q, kp, qs, ks, q)
    gfd = _fd_grad(o64, q, kp, qs, ks) * qs
    assert np.abs(g - gfd).max() <= 5e-6 * max(1.0, np.abs(g).max())
    l32, g32 = o32.q_loss(q, kp, qs, ks, q)
    assert abs(l32 - loss) <= 1e-6 and np.abs(g32 - g).max() <= 5e-6
    if which == "all":
        assert (g != 0).sum() == 45  # only marker-ancestor joints get gradient


def test_gradient_ball_and_slide_joints():
    from stac_mjx_amd.mjcf import compile_mjcf

    xml = """
    <mujoco><compiler angle="radian"/><worldbody>
      <body name="r" pos="0 0 0.5"><freejoint/>
        <site name="s0" pos="0.1 0 0"/>
        <body name="a" pos="0.2 0 0" quat="0.9 0.1 0.2 0.3"><joint name="ball" type="ball" pos="0.01 0.02 0"/>
          <site name="s1" pos="0 0.1 0"/>
          <body name="b" pos="0 0.2 0"><joint name="sl" type="slide" axis="1 1 0"/><joint name="h" axis="0 1 1" pos="0 0 .1"/>
            <site name="s2" pos="0.05 0.02 0.1"/></body></body></body>
    </worldbody></mujoco>"""
    t = compile_mjcf(xml, from_string=True)
    assert t.nq == 7 + 4 + 2
    o64 = Oracle(t, precision="f64")
    rng = np.random.default_rng(1)
    q = t.qpos0 + rng.normal(0, 0.2, t.nq).astype(np.float32)
    kp = rng.normal(0, 0.3, 9).astype(np.float32)
    qs, ks = np.ones(t.nq, bool), np.ones(9, bool)
    _, g = o64.q_loss(q, kp, qs, ks, q)
    gfd = _fd_grad(o64, q, kp, qs, ks, h=1e-4)
    assert np.all(g != 0)
    assert np.abs(g - gfd).max() <= 5e-6 * np.abs(g).max(), np.abs(g - gfd)


def test_make_qs_mask_blocks_gradient_and_values(orc, rodent_setup, rodent_mocap):
    fs = rodent_setup
    q0 = fs.tables.qpos0.copy()
    q = q0 + 0.1
    mask = fs.part_masks[2]
    loss_m, g = orc.q_loss(q, rodent_mocap[0], mask, np.ones(69, bool), q0)
    assert np.all(g[~mask] == 0)
    blended = np.where(mask, q, q0)
    loss_b, _ = orc.q_loss(blended, rodent_mocap[0], np.ones(74, bool), np.ones(69, bool), blended)
    assert loss_m == loss_b


# ---- projected gradient ------------------------------------------------------------------------------
def test_pg_decreases_loss_respects_box_and_reports_state(orc, rodent_setup, rodent_mocap):
    fs = rodent_setup
    q0 = fs.tables.qpos0.copy()
    q0[:3] = rodent_mocap[0, 3 * fs.root_kp_idx: 3 * fs.root_kp_idx + 3]
    allq, allk = np.ones(74, bool), np.ones(69, bool)
    l0, _ = orc.q_loss(q0, rodent_mocap[0], allq, allk, q0)
    x, st = orc.q_opt(rodent_mocap[0], allq, allk, q0, fs.lb, fs.ub)
    assert st["loss"] < 0.05 * l0
    assert np.all(x >= fs.lb) and np.all(x <= fs.ub)
    assert 1 <= st["iter_num"] <= 400 and st["grad_evals"] == 2 * st["iter_num"]
    assert st["ls_evals"] >= st["iter_num"]
    assert st["error"] <= 1e-4 or st["iter_num"] == 400
    # joints that are not ancestors of any marker never move (tail, toes, fingers, jaw)
    qr = q0 + np.random.default_rng(3).normal(0, 0.1, 74).astype(np.float32)
    _, g = orc.q_loss(qr, rodent_mocap[0], allq, allk, qr)
    assert (g == 0).sum() == 29
    assert np.all(x[g == 0] == np.clip(q0, fs.lb, fs.ub)[g == 0])


def test_pg_maxiter_zero_returns_q0(rodent_setup, rodent_mocap):
    o = Oracle(rodent_setup.tables, maxiter=0)
    q0 = rodent_setup.tables.qpos0.copy()
    x, st = o.q_opt(rodent_mocap[0], np.ones(74, bool), np.ones(69, bool), q0, rodent_setup.lb, rodent_setup.ub)
    np.testing.assert_array_equal(x, q0)
    assert st["iter_num"] == 0 and np.isinf(st["error"])


def test_pg_fixed_point_when_already_optimal(toy_tables):
    """Keypoints generated by FK at q* => loss 0, gradient 0, PG stops after one update at q*."""
    o = Oracle(toy_tables, tol=1e-6, maxiter=50)
    qs = np.array([0.3, -0.2, 0.5], np.float32)
    kp = o.fk(qs)["site_xpos"].reshape(-1)
    x, st = o.q_opt(kp, np.ones(3, bool), np.ones(9, bool), qs, -np.ones(3, np.float32) * 7, np.ones(3, np.float32) * 7)
    assert st["iter_num"] == 1 and st["error"] <= 1e-6
    np.testing.assert_allclose(x, qs, atol=1e-6)


def test_pg_recovers_pose_on_toy_model(toy_tables):
    o = Oracle(toy_tables, tol=1e-7, maxiter=2000)
    qs = np.array([0.4, -0.3, 0.2], np.float32)
    kp = o.fk(qs)["site_xpos"].reshape(-1)
    lim = np.full(3, 2 * np.pi, np.float32)
    x, st = o.q_opt(kp, np.ones(3, bool), np.ones(9, bool), np.zeros(3, np.float32), -lim, lim)
    np.testing.assert_allclose(x, qs, atol=1e-3)


# ---- offset phase: the reference's known-answer tests (tests/unit/test_m_opt.py) ------------------------
GT_A = np.array([[0.1, 0.2, 0.3], [0.4, 0.5, 0.6], [0.15, 0.25, 0.35]], np.float32)
GT_B = np.array([[0.2, -0.1, 0.4], [0.3, 0.4, -0.2], [-0.1, 0.3, 0.1]], np.float32)


def _toy_keypoints(o, q_traj, offsets):
    o.set_site_pos(offsets)
    return np.stack([o.fk(q)["site_xpos"].reshape(-1) for q in q_traj])


@pytest.fixture()
def toy(toy_tables):
    return Oracle(toy_tables)


def test_m_opt_identity_pose_recovers_offsets_and_error(toy):
    q = np.zeros((5, 3), np.float32)
    kp = _toy_keypoints(toy, q, GT_A)
    params, err = toy.m_opt(kp, q, np.zeros((3, 3)), np.zeros((3, 3)), 0.0)
    np.testing.assert_allclose(params, GT_A, atol=1e-5)
    assert err < 1e-8


def test_m_opt_varied_random_poses(toy):
    q = (np.random.RandomState(42).randn(10, 3) * 0.5).astype(np.float32)
    kp = _toy_keypoints(toy, q, GT_A)
    params, _ = toy.m_opt(kp, q, np.zeros((3, 3)), np.zeros((3, 3)), 0.0)
    np.testing.assert_allclose(params, GT_A, atol=1e-5)


def test_m_opt_sweeping_single_joint(toy):
    q = np.zeros((8, 3), np.float32)
    q[:, 0] = np.linspace(0.0, np.pi / 4, 8)
    kp = _toy_keypoints(toy, q, GT_B)
    params, _ = toy.m_opt(kp, q, GT_B, np.zeros((3, 3)), 0.0)
    np.testing.assert_allclose(params, GT_B, atol=1e-5)


def test_m_opt_large_rotations(toy):
    q = (np.random.RandomState(99).randn(15, 3) * 1.5).astype(np.float32)
    kp = _toy_keypoints(toy, q, GT_B)
    params, _ = toy.m_opt(kp, q, np.zeros((3, 3)), np.zeros((3, 3)), 0.0)
    np.testing.assert_allclose(params, GT_B, atol=1e-4)


def test_m_opt_reg_coef_zero_vs_strong(toy):
    q = (np.random.RandomState(42).randn(10, 3) * 0.3).astype(np.float32)
    kp = _toy_keypoints(toy, q, GT_A)
    p0, _ = toy.m_opt(kp, q, np.full((3, 3), 99.0), np.ones((3, 3)), 0.0)
    np.testing.assert_allclose(p0, GT_A, atol=1e-5)
    p1, _ = toy.m_opt(kp, q, np.zeros((3, 3)), np.ones((3, 3)), 1e6)
    np.testing.assert_allclose(p1, np.zeros((3, 3)), atol=1e-3)


def test_m_opt_partial_regularization(toy):
    q = np.zeros((10, 3), np.float32)
    gt = np.full((3, 3), 0.5, np.float32)
    kp = _toy_keypoints(toy, q, gt)
    is_reg = np.zeros((3, 3), np.float32)
    is_reg[0] = 1.0
    ps, _ = toy.m_opt(kp, q, np.zeros((3, 3)), is_reg, 1e4)
    pn, _ = toy.m_opt(kp, q, np.zeros((3, 3)), is_reg, 0.0)
    assert np.linalg.norm(ps[0]) < np.linalg.norm(pn[0])
    np.testing.assert_allclose(ps[1:], gt[1:], atol=1e-5)


def test_m_opt_error_equals_direct_objective(orc_legacy, demo_viz):
    """error returned by the closed form == objective evaluated directly at m* (stac_core.py:168-170)."""
    o = orc_legacy
    T = 20
    q, kp = demo_viz["qpos"][:T], demo_viz["kp_data"][:T]
    m0 = demo_viz["offsets"]
    d = np.zeros((23, 3), np.float32)
    d[[8, 9, 17, 18, 19]] = 1.0
    params, err = o.m_opt(kp, q, m0, d, 1.0)
    o2 = Oracle(o.t)
    o2.set_site_pos(params)
    direct = sum(((kp[t].reshape(23, 3) - o2.fk(q[t])["site_xpos"]) ** 2).sum() for t in range(T))
    direct += ((d * (params - m0)) ** 2).sum()
    assert abs(err - direct) <= 2e-4 * max(direct, 1e-3)
    # partial sums split over two "ranks" add up (what the all-reduce relies on)
    pa, pb = o.m_partial(kp[:7], q[:7]), o.m_partial(kp[7:], q[7:])
    p2, e2 = o.m_finish(pa + pb, m0, d, 1.0)
    np.testing.assert_allclose(p2, params, atol=2e-6)


# ---- phase drivers -----------------------------------------------------------------------------------------
def test_oracle_regression_pin(orc, rodent_setup, golden_dir):
    fs = rodent_setup
    with np.load(golden_dir / "oracle_regress.npz") as d:
        out = orc.ik_clips(d["kp"], fs.lb, fs.ub, fs.part_masks, fs.trunk_kps, fs.root_kp_idx, fs.root_dims)
        np.testing.assert_array_equal(out["counters"], d["counters"])
        np.testing.assert_array_equal(out["qpos"], d["qpos"])
        np.testing.assert_array_equal(out["frame_error"], d["frame_error"])


def test_pose_optimization_sequencing(orc, rodent_setup, rodent_mocap):
    """Driver == hand-rolled sequence of q_opt calls with reference sequencing (compute_stac.py:216-267)."""
    fs = rodent_setup
    kp = rodent_mocap[:2]
    q, _ = orc.root_optimization(kp, fs.tables.qpos0, fs.lb, fs.ub, fs.trunk_kps, fs.root_kp_idx)
    out = orc.pose_optimization(kp, q, fs.lb, fs.ub, fs.part_masks)
    allq, allk = np.ones(74, bool), np.ones(69, bool)
    qpos = q.copy()
    for f in range(2):
        x, st = orc.q_opt(kp[f], allq, allk, qpos, fs.lb, fs.ub)
        qpos = orc.fk(x)["qpos"]
        for pm in fs.part_masks:
            x, st = orc.q_opt(kp[f], pm, allk, qpos, fs.lb, fs.ub)
            qpos = orc.fk(np.where(pm, x, qpos))["qpos"]
        np.testing.assert_array_equal(out["qpos"][f], qpos)
        assert out["frame_error"][f] == np.float32(st["error"])  # residual of the LAST solve
        assert out["counters"][f, 3] == 6
    np.testing.assert_array_equal(out["marker_sites"][1], orc.fk(qpos)["site_xpos"])
    # initial (unfitted) offsets: centimetre-level marker error is expected before the offset phase
    err = np.linalg.norm(out["marker_sites"] - kp.reshape(2, 23, 3), axis=-1)
    assert err.mean() < 2e-2


def test_ik_clips_threads_and_clip_independence(orc, rodent_setup, rodent_mocap):
    fs = rodent_setup
    kp = rodent_mocap[:6].reshape(3, 2, 69)
    a = orc.ik_clips(kp, fs.lb, fs.ub, fs.part_masks, fs.trunk_kps, fs.root_kp_idx, nthreads=1)
    b = orc.ik_clips(kp[1:2], fs.lb, fs.ub, fs.part_masks, fs.trunk_kps, fs.root_kp_idx, nthreads=2)
    np.testing.assert_array_equal(a["qpos"][1], b["qpos"][0])
    np.testing.assert_array_equal(a["xpos"][1], b["xpos"][0])


def test_pg_residuals_against_the_run_the_reference_printed(orc, rodent_setup, rodent_mocap):
    """The one printed run of the reference: demos/rodent_demo.ipynb cell 6 ran run_stac on the first 10 frames of the
    same .mat with configs/model/rodent.yaml: `Root optimization ... error of 4.3102e-05`, first pose pass (before any
    offset update) `Mean: 3.5538e-05 / Standard deviation: 9.482e-06` of the per-frame PG residuals.

    tests/tools/jaxopt_variant_sweep.py (table: profiles/r03/jaxopt_variant_sweep.txt) shows that
    (1) that notebook was produced by OLDER reference source (its print strings and its 40 s iterative offset phase do
    not exist in the current compute_stac.py); (2) none of 70 variants of the jaxopt details or of the older host rules
    moves the root residual from 9.5e-5 to 4.3e-5 -- the second root solve is a period-two step-size cycle whose
    low-phase residuals are ..., 1.20e-4, 9.50e-5, 7.6e-5, 6.4e-5, 6.03e-5 (minimum), so 4.31e-5 cannot be the first
    value below FTOL on this trajectory; (3) the gap is CONFINED to the root optimisation and the frame that follows it:
    frames 1-9 of the first pose pass give 3.42e-5 +- 0.78e-5 against the notebook's 3.55e-5 +- 0.95e-5 for all ten.
    These are the tightest bounds the restatement achieves; they are asserted on the mean AND the std."""
    fs = rodent_setup
    kp = rodent_mocap[:10]
    q, st = orc.root_optimization(kp, fs.tables.qpos0, fs.lb, fs.ub, fs.trunk_kps, fs.root_kp_idx)
    assert st["error"] <= 1e-4  # converged, like the reference's (whose value the restatement does not reach: next test)
    out = orc.pose_optimization(kp, q, fs.lb, fs.ub, fs.part_masks)
    e = out["frame_error"].astype(np.float64)
    assert (e <= 1e-4).all()  # every frame's last (head) solve converged, like the reference's
    # all ten frames: mean within 15 %, std within 2.2x (frame 0, straight after the root optimisation, carries the gap)
    assert abs(e.mean() / 3.5538e-05 - 1) <= 0.15, e.mean()
    assert e.std() <= 2.2 * 9.482e-06, e.std()
    # frames 1-9: mean within 5 %, std within 25 % of the notebook's ten-frame figures
    assert abs(e[1:].mean() / 3.5538e-05 - 1) <= 0.05, e[1:].mean()
    assert abs(e[1:].std() / 9.482e-06 - 1) <= 0.25, e[1:].std()


@pytest.mark.xfail(strict=True, reason="KNOWN GAP, not a target: the reference notebook printed a root-optimisation residual of "
                   "4.3102e-05; the restated PG stops at 9.50e-05 after 30 iterations of the second root solve (factor 2.2, unexplained "
                   "by 70 variants: profiles/r03/jaxopt_variant_sweep.txt).  A change that moves the oracle ONTO the reference makes "
                   "this test pass -- then delete the marker")
def test_known_gap_root_residual_of_the_printed_reference_run(orc, rodent_setup, rodent_mocap):
    fs = rodent_setup
    _, st = orc.root_optimization(rodent_mocap[:10], fs.tables.qpos0, fs.lb, fs.ub, fs.trunk_kps, fs.root_kp_idx)
    assert abs(st["error"] / 4.3102e-05 - 1) <= 0.05, (st["iter_num"], st["error"])


def test_regression_root_solve_trajectory_of_the_restatement(orc, rodent_setup, rodent_mocap):
    """Regression pin of the restatement's OWN trajectory (so that an unintended change of the oracle is noticed); this
    value is NOT the reference's -- see the known-gap test above."""
    fs = rodent_setup
    _, st = orc.root_optimization(rodent_mocap[:10], fs.tables.qpos0, fs.lb, fs.ub, fs.trunk_kps, fs.root_kp_idx)
    assert st["iter_num"] == 30 and abs(st["error"] - 9.50e-5) < 1e-6


def test_variant_sweep_baseline_equals_oracle(orc, rodent_setup, rodent_mocap):
    """The numpy twin of q_opt_ws inside tests/tools/jaxopt_variant_sweep.py (the loop the variants modify) runs the
    oracle's trajectory: same iteration counts in the root solves, residuals equal to 1e-3 relative (numpy's dot
    products sum in another order than the oracle's trees)."""
    import sys
    from pathlib import Path

    sys.path.insert(0, str(Path(__file__).resolve().parent / "tools"))
    import jaxopt_variant_sweep as sweep

    r = sweep.run_variant({}, passes=False)
    fs = rodent_setup
    _, st = orc.root_optimization(rodent_mocap[:10], fs.tables.qpos0, fs.lb, fs.ub, fs.trunk_kps, fs.root_kp_idx)
    assert r["root_iters"][-1] == st["iter_num"]
    assert abs(r["root_err"] / st["error"] - 1) < 1e-3
    # a variant that is known to change the trajectory does change it (the switches are live)
    r2 = sweep.run_variant(dict(eps=0.0), passes=False)
    assert r2["root_iters"] != r["root_iters"]


# ---- optional LM solver (not the reference's algorithm): the oracle side ---------------------------------------------------
def test_lm_converges_and_fits_at_least_as_well_as_pg(orc, rodent_setup, rodent_mocap):
    fs = rodent_setup
    kp = rodent_mocap[::50][:12].reshape(12, 1, 69)
    lm = orc.ik_clips_lm(kp, fs.lb, fs.ub, fs.part_masks, fs.trunk_kps, fs.root_kp_idx, fs.root_dims, want_bodies=False)
    pg = orc.ik_clips(kp, fs.lb, fs.ub, fs.part_masks, fs.trunk_kps, fs.root_kp_idx, fs.root_dims, want_bodies=False)
    tgt = kp.reshape(12, 1, 23, 3)
    e_lm = np.linalg.norm(lm["marker_sites"] - tgt, axis=-1).mean()
    e_pg = np.linalg.norm(pg["marker_sites"] - tgt, axis=-1).mean()
    assert e_lm <= e_pg + 1e-5, (e_lm, e_pg)
    assert (lm["frame_error"] <= 1e-4).all()  # converged to the same stopping residual the PG solver uses
    assert lm["counters"][..., 2].mean() < 0.05 * (pg["counters"][..., 1] + pg["counters"][..., 2]).mean()
    q = lm["qpos"][:, 0]
    assert (q >= fs.lb - 1e-6).all() and (q <= fs.ub + 1e-6).all()
    np.testing.assert_allclose(np.linalg.norm(q[:, 3:7], axis=1), 1.0, atol=1e-5)


def test_lm_single_solve_matches_f64_twin_in_marker_space(rodent_setup, rodent_mocap):
    fs = rodent_setup
    o32, o64 = Oracle(fs.tables), Oracle(fs.tables, precision="f64")
    kp = rodent_mocap[10]
    q0 = fs.tables.qpos0.copy()
    q0[:3] = kp[3 * fs.root_kp_idx: 3 * fs.root_kp_idx + 3]
    allq, allk = np.ones(74, bool), np.ones(69, bool)
    x32, s32 = o32.q_opt_lm(kp, allq, allk, q0, fs.lb, fs.ub)
    x64, s64 = o64.q_opt_lm(kp, allq, allk, q0, fs.lb, fs.ub)
    assert s32["error"] <= 1e-4 and s64["error"] <= 1e-4 and s32["iter_num"] < 40
    m32, m64 = o32.fk(x32)["site_xpos"], o64.fk(x64)["site_xpos"]
    assert np.abs(m32 - m64).max() < 5e-4 and abs(s32["loss"] - s64["loss"]) < 1e-3 * s64["loss"]


# ---- the PG driver against an independent transcription of jaxopt's algorithm -----------------------------
def _pg_numpy(orc, kp, qs, ks, q0, lb, ub, tol, maxiter, maxls):
    """ProjectedGradient.run as SURVEY.md A2 states it (FISTA momentum, backtracking line search, box projection, unit-step
    residual), written independently of oracle/stac_oracle.c in numpy float32.  Only the objective (value, gradient) is the
    oracle's; float32 fma is emulated through float64 (the product of two float32 is exact there), sums use the same
    pairwise tree as the oracle so that the two drivers can be compared bit for bit."""
    f32 = np.float32

    def fma(a, b, c):
        return (np.asarray(a, np.float64) * np.asarray(b, np.float64) + np.asarray(c, np.float64)).astype(f32)

    def tree(v):
        n = 1
        while n < v.size:
            n *= 2
        v = np.concatenate([v.astype(f32), np.zeros(n - v.size, f32)])
        while v.size > 1:
            v = v[0::2] + v[1::2]
        return f32(v[0])

    def fun(p, grad):
        loss, g = orc.q_loss(p, kp, qs, ks, q0, grad)
        return f32(loss), g

    clip = lambda v: np.minimum(np.maximum(v, lb), ub).astype(f32)
    x, y = q0.astype(f32).copy(), q0.astype(f32).copy()
    eta_state, t, error = f32(1), f32(1), f32(np.inf)
    it = ls_evals = 0
    eps = f32(1.1920929e-7)
    while it < maxiter and (it == 0 or error > f32(tol)):
        fy, g = fun(y, True)
        eta = eta_state
        cand = clip(fma(-eta, g, y))
        n = 0
        while n < maxls:
            fc, _ = fun(cand, False)
            ls_evals += 1
            d = cand - y
            sq, vd = tree(d * d), tree(d * g)
            if not (eta * (fc - fy) > (eta * vd + f32(0.5) * sq) + eps):
                break
            eta = eta * f32(0.5)
            cand = clip(fma(-eta, g, y))
            n += 1
        eta_next = f32(1) if eta <= f32(1e-6) else eta / f32(0.5)
        tn = f32(0.5) * (f32(1) + np.sqrt(f32(1) + f32(4) * t * t, dtype=f32))
        beta = (t - f32(1)) / tn
        y = fma(beta, cand - x, cand)
        x = cand
        _, gn = fun(x, True)
        d = clip(x - gn) - x
        error = np.sqrt(tree(d * d), dtype=f32)
        eta_state, t = eta_next, tn
        it += 1
    return x, dict(iter_num=it, stepsize=float(eta_state), error=float(error), t=float(t), ls_evals=ls_evals)


@pytest.mark.parametrize("which,maxls", [("all", 15), ("part", 15), ("root", 15), ("all", 2)])
def test_pg_driver_equals_independent_transcription(orc, rodent_setup, rodent_mocap, which, maxls):
    fs = rodent_setup
    o = Oracle(fs.tables, tol=1e-4, maxiter=30, maxls=maxls)
    kp = rodent_mocap[7]
    q0 = fs.tables.qpos0.copy()
    q0[:3] = kp[3 * fs.root_kp_idx: 3 * fs.root_kp_idx + 3]
    qs = {"all": np.ones(74, bool), "part": fs.part_masks[0].astype(bool), "root": np.arange(74) < 7}[which]
    ks = np.repeat(fs.trunk_kps, 3).astype(bool) if which == "root" else np.ones(69, bool)
    x_c, st_c = o.q_opt(kp, qs, ks, q0, fs.lb, fs.ub)
    x_p, st_p = _pg_numpy(o, kp, qs, ks, q0, fs.lb.astype(np.float32), fs.ub.astype(np.float32), 1e-4, 30, maxls)
    np.testing.assert_array_equal(x_c, x_p)
    for k in ("iter_num", "stepsize", "error", "t", "ls_evals"):
        assert st_c[k] == st_p[k], (k, st_c[k], st_p[k])
    assert st_c["iter_num"] > 3


def _ball_case(seed, free_root):
    """A random tree with ball, slide and hinge joints (the generator of the GPU fuzz tests), a pose and targets near it."""
    from test_gpu_parity import _random_tables
    from stac_mjx_amd.mjcf import JNT_BALL, JNT_FREE

    rng = np.random.default_rng(seed)
    t = _random_tables(rng, 24, free_root, p_slide=0.15, p_ball=0.3)
    lb, ub = np.full(t.nq, -np.inf, np.float32), np.full(t.nq, np.inf, np.float32)
    for j in range(t.njnt):
        a, ty = int(t.jnt_qposadr[j]), int(t.jnt_type[j])
        if ty == JNT_FREE: lb[a + 3:a + 7], ub[a + 3:a + 7] = -1, 1
        elif ty == JNT_BALL: lb[a:a + 4], ub[a:a + 4] = -1, 1
        else: lb[a], ub[a] = min(t.jnt_range[j, 0], 0.0), t.jnt_range[j, 1]
    return rng, t, lb, ub


@pytest.mark.parametrize("seed,free_root", [(0, True), (1, False), (2, True), (3, False)])
def test_lm_jacobian_columns_agree_with_the_analytic_gradient(seed, free_root):
    """Round 5 (ball joints in the LM solver): J^T f over lm_jac_col's columns == the gradient q_loss computes, for every coordinate
    of hinge, slide, free and ball joints (a ball's columns: in the frame its rotation is applied in, and back)."""
    rng, t, lb, ub = _ball_case(seed, free_root)
    assert (t.jnt_type == 1).sum() >= 3
    q = np.asarray(t.qpos0, np.float32) + 0.2 * rng.standard_normal(t.nq).astype(np.float32)
    kp = (0.1 * rng.standard_normal(3 * t.nsite)).astype(np.float32)
    assert Oracle(t, precision="f64").lm_jac_check(q, kp) < 1e-7   # (float inputs, double arithmetic)
    assert Oracle(t).lm_jac_check(q, kp) < 5e-6


@pytest.mark.parametrize("seed,free_root", [(0, True), (1, False)])
def test_lm_with_ball_joints_converges_and_fits_at_least_as_well_as_pg(seed, free_root):
    rng, t, lb, ub = _ball_case(seed, free_root)
    orc = Oracle(t, tol=1e-4, maxiter=400)
    n, K = 6, t.nsite
    q = np.tile(t.qpos0, (n, 1)) + rng.normal(0, 0.15, (n, t.nq)).astype(np.float32)
    q = np.clip(q, np.where(np.isfinite(lb), lb, -3), np.where(np.isfinite(ub), ub, 3)).astype(np.float32)
    kp = np.stack([orc.fk(x.copy())["site_xpos"].reshape(-1) for x in q]).astype(np.float32).reshape(n, 1, 3 * K)
    part = np.zeros((0, t.nq), np.uint8)
    trunk = np.ones(K, np.uint8)
    lm = orc.ik_clips_lm(kp, lb, ub, part, trunk, 0, 7, do_root_opt=free_root, want_bodies=False, maxiter=40)
    pg = orc.ik_clips(kp, lb, ub, part, trunk, 0, 7, do_root_opt=free_root, want_bodies=False)
    tgt = kp.reshape(n, 1, K, 3)
    e_lm = np.linalg.norm(lm["marker_sites"] - tgt, axis=-1).mean()
    e_pg = np.linalg.norm(pg["marker_sites"] - tgt, axis=-1).mean()
    assert e_lm <= e_pg + 1e-5 and e_lm < 2e-3, (e_lm, e_pg)
    ql = lm["qpos"][:, 0]
    fin = np.isfinite(lb) & np.isfinite(ub)
    assert (ql[:, fin] >= lb[fin] - 1e-6).all() and (ql[:, fin] <= ub[fin] + 1e-6).all()
    # far fewer evaluations than the projected gradient needs
    assert lm["counters"][..., 2].mean() < 0.25 * (pg["counters"][..., 1] + pg["counters"][..., 2]).mean()
