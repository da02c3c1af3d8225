"""The q_phase pinned to the ONE real reference output in the tree: demos/demo_viz.p (50 rodent frames fitted by
stac-mjx itself: qpos, offsets, marker sites, keypoints; committed as tests/golden/demo_viz_golden.npz).

The reference holds no numeric test of q_opt / pose_optimization (its tests mock q_opt), so this stored fit is the
only MJX + jaxopt result the restated projected gradient can be tied to.  qpos is not identifiable to 1e-4 on the
full body (7 near-null directions, iteration-capped solves), so the pins are, in order of strength:

  1. the stored pose satisfies the ORACLE's stopping rule for the reference's last solve of each frame
     (|| clip(q - grad) - q || <= FTOL over the last part's coordinates): gradient, box projection, masks and
     residual definition agree with what jaxopt evaluated;
  2. the stored pose is a fixed point of the oracle's part solves to <= 1e-4 in the median frame (the north
     star's tolerance), i.e. restarting the reference's last solves at its own answer changes nothing;
  3. the whole driver (root optimisation + 7 warm-started pose passes, compute_stac.py:17-104,170-278,
     stac.py:298-341) fits the same 50 frames to <= the stored fit's mean marker error (0.919 mm) from the second
     pass on;
  4. fit_offsets from the YAML's initial offsets (12.3 mm away) lands within 1.5 mm of the stored offsets and at a
     lower marker error than the stored fit.

The `-m gpu` half runs the same computations through libstac_hip.so and requires them to equal the oracle's
bit for bit, which carries the pins over to the HIP kernels.
"""

import numpy as np
import pytest

from helpers import marker_error_mm, oracle_chain_passes, oracle_fit_offsets, pg_residual

STORED_MEAN_MM = 0.919  # mean marker error of the stored reference fit on its 50 frames
FTOL = 1e-4             # configs/model/rodent.yaml:4.  demo_viz.p does not record the tolerance it was fitted with (SURVEY.md
                        # guessed the older default 5e-3); pin 1 below is what shows it was <= 1e-4: a 5e-3 run would leave
                        # head residuals spread up to 5e-3, the stored ones are <= 1e-4 in 98 % of the frames.
N_PASSES = 7            # N_ITERS = 6 pose passes + the final one (stac.py:298,331)


@pytest.fixture(scope="module")
def orc_stored(rodent_setup_legacy, demo_viz):
    from oracle import Oracle

    o = Oracle(rodent_setup_legacy.tables, tol=FTOL, maxiter=400)
    o.set_site_pos(demo_viz["offsets"])
    return o


def test_stored_fit_quality_is_what_the_survey_measured(demo_viz):
    assert abs(marker_error_mm(demo_viz["walker_body_sites"], demo_viz["kp_data"]) - STORED_MEAN_MM) < 2e-3


def test_stored_pose_satisfies_the_last_solves_stopping_rule(orc_stored, rodent_setup_legacy, demo_viz):
    """Pin 1.  The reference records the pose after the last part solve ("head"); that solve stopped on
    error <= FTOL, so the oracle's residual over the head coordinates, evaluated at the stored pose, must be
    <= FTOL too (up to float32 noise between jax's autodiff and the analytic gradient).  The arms come before
    it and are not touched afterwards: same rule.  The full-body residual is far above FTOL: those solves are
    iteration-capped, as restated."""
    fs = rodent_setup_legacy
    names = list(fs.part_names_cfg) if hasattr(fs, "part_names_cfg") else None
    R = np.array([[pg_residual(orc_stored, fs, demo_viz["qpos"][t], demo_viz["kp_data"][t], m)
                   for m in [np.ones(fs.tables.nq)] + list(fs.part_masks)] for t in range(50)])
    full, r_leg, l_leg, r_arm, l_arm, head = R.T
    assert (head <= FTOL).mean() >= 0.95 and head.max() <= 2e-4, (names, np.quantile(head, [0.5, 0.9, 1.0]))
    assert (r_arm <= FTOL).mean() >= 0.85 and (l_arm <= FTOL).mean() >= 0.75
    assert np.median(r_leg) <= 2e-4 and np.median(l_leg) <= 2e-4 and max(r_leg.max(), l_leg.max()) <= 5e-4
    assert full.min() > 10 * FTOL  # the full-body solve never converged in the reference either


def test_stored_pose_is_a_fixed_point_of_the_part_solves(orc_stored, rodent_setup_legacy, demo_viz):
    """Pin 2.  Restart every part solve of the reference at the reference's own answer: the oracle stops at once
    (median 1-3 iterations of 400) and moves the part's coordinates by <= 1e-4 in the median frame."""
    fs = rodent_setup_legacy
    ones = np.ones(69, np.uint8)
    report = []
    for pi, pm in enumerate(fs.part_masks):
        its, dqs = [], []
        for t in range(50):
            out, st = orc_stored.q_opt(demo_viz["kp_data"][t], pm.astype(np.uint8), ones, demo_viz["qpos"][t], fs.lb, fs.ub)
            its.append(st["iter_num"])
            dqs.append(np.abs((out - demo_viz["qpos"][t]) * pm).max())
        report.append((np.median(its), max(its), np.median(dqs), max(dqs)))
    for med_it, max_it, med_dq, _ in report:
        assert med_it <= 3 and max_it <= 20, report
        assert med_dq <= 5e-4, report
    for med_it, _, med_dq, _ in report[2:]:  # arms and head: the last three solves of a frame
        assert med_it <= 1 and med_dq <= 1e-4, report


def test_driver_fits_the_golden_frames_at_least_as_well_as_the_reference(orc_stored, rodent_setup_legacy, demo_viz, record_property):
    """Pin 3.  compute_stac.py:205-278 driven over the golden frames with the stored offsets."""
    fs = rodent_setup_legacy
    outs = oracle_chain_passes(orc_stored, fs, demo_viz["kp_data"], N_PASSES)
    errs = [marker_error_mm(o["marker_sites"], demo_viz["kp_data"]) for o in outs]
    dq = [np.abs(o["qpos"] - demo_viz["qpos"]) for o in outs]
    for p in range(N_PASSES):
        record_property(f"pass{p}_marker_mm", round(errs[p], 4))
        record_property(f"pass{p}_dq_median_p90_max", [float(f"{v:.3g}") for v in (np.median(dq[p]), np.quantile(dq[p], 0.9), dq[p].max())])
    print("\nmarker error per pass (mm):", np.round(errs, 4), " stored reference fit:", STORED_MEAN_MM)
    print("|qpos - stored| median / p90 / max per pass:", [(float(f"{np.median(d):.2g}"), float(f"{np.quantile(d, .9):.2g}"), float(f"{d.max():.2g}")) for d in dq])
    assert errs[0] <= 1.0                      # first pass from a cold start: 0.957 mm
    assert errs[1] <= STORED_MEAN_MM           # 0.820 mm: at least as good as the reference's own fit
    assert errs[6] <= 0.83 and errs[6] <= errs[1]
    assert all(e2 <= e1 + 1e-3 for e1, e2 in zip(errs[1:], errs[2:]))  # the warm-started passes keep improving
    # joint space: typical agreement 1e-3 rad, weakly determined joints differ -- recorded, bounded
    assert np.median(dq[6]) <= 2e-3 and dq[6].max() <= 0.5
    # the marker-less subtrees (tail, toes, fingers, jaw) never move, in the reference and here
    never = np.all(demo_viz["qpos"] == 0.0, axis=0)
    assert never.sum() >= 25 and np.all(outs[6]["qpos"][:, never] == 0.0)


def test_fit_offsets_on_the_golden_frames_lands_near_the_stored_offsets(rodent_setup_legacy, rodent_cfg, demo_viz):
    """Pin 4.  Stac.fit_offsets (stac.py:253-354, N_ITERS = 6) on the 50 golden frames from the YAML's
    KEYPOINT_INITIAL_OFFSETS.  The stored offsets come from a longer legacy fit (other frames, iterative m-phase),
    so they are approached, not reproduced: from 12.3 mm away to about 1.2 mm, at a marker error below the
    stored fit's."""
    fs = rodent_setup_legacy
    stored = demo_viz["offsets"]
    d0 = np.linalg.norm(fs.tables.site_pos - stored, axis=-1).mean() * 1e3
    hist = []
    off, out = oracle_fit_offsets(fs, rodent_cfg, demo_viz["kp_data"], int(rodent_cfg["N_ITERS"]), history=hist)
    d = [np.linalg.norm(o - stored, axis=-1).mean() * 1e3 for _, o in hist]
    err = marker_error_mm(out["marker_sites"], demo_viz["kp_data"])
    print(f"\noffsets: initial {d0:.2f} mm from the stored ones; after each calibration iteration {np.round(d, 3)}; final marker error {err:.3f} mm")
    assert d0 > 10.0 and d[0] <= 1.6 and d[-1] <= 1.3 and all(b <= a + 1e-3 for a, b in zip(d, d[1:]))
    assert np.linalg.norm(off - stored, axis=-1).max() * 1e3 <= 3.0
    assert err <= STORED_MEAN_MM


# ---------------------------------------------------------------------------------------------------------
# the same through libstac_hip.so: equal to the oracle bit for bit, hence tied to the same pins
# ---------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
def test_gpu_chain_passes_equal_the_pinned_oracle(orc_stored, rodent_setup_legacy, demo_viz):
    import torch

    from stac_mjx_amd.engine import Engine

    fs = rodent_setup_legacy
    eng = Engine(fs.tables, fs.lb, fs.ub, tol=FTOL, maxiter=400)
    eng.set_site_pos(demo_viz["offsets"])
    ref = oracle_chain_passes(orc_stored, fs, demo_viz["kp_data"], N_PASSES)
    kp = torch.as_tensor(demo_viz["kp_data"][None]).to(eng.device)
    carry = None
    for p in range(N_PASSES):
        res = eng.q_phase(kp, part_masks=fs.part_masks, trunk_kps=fs.trunk_kps, root_kp_idx=fs.root_kp_idx,
                          root_dims=fs.root_dims, do_root_opt=(p == 0), q_init=carry)
        carry = res["carry_qpos"]
        np.testing.assert_array_equal(res["qpos"][0].cpu().numpy(), ref[p]["qpos"])
        np.testing.assert_array_equal(res["marker_sites"][0].cpu().numpy(), ref[p]["marker_sites"])
        np.testing.assert_array_equal(res["frame_error"][0].cpu().numpy(), ref[p]["frame_error"])
        np.testing.assert_array_equal(res["counters"][0].cpu().numpy().astype(np.uint32), ref[p]["counters"])
        err = marker_error_mm(res["marker_sites"].cpu().numpy(), demo_viz["kp_data"])
        assert err <= (1.0 if p == 0 else STORED_MEAN_MM)
    assert err <= 0.83


@pytest.mark.gpu
def test_gpu_part_solves_are_fixed_points_at_the_stored_pose(orc_stored, rodent_setup_legacy, demo_viz):
    from stac_mjx_amd.engine import Engine

    fs = rodent_setup_legacy
    eng = Engine(fs.tables, fs.lb, fs.ub, tol=FTOL, maxiter=400)
    eng.set_site_pos(demo_viz["offsets"])
    ones = np.ones(69, np.uint8)
    for pi, pm in enumerate(fs.part_masks):
        params, state, counters = eng.q_solve(demo_viz["kp_data"], demo_viz["qpos"], pm, ones)
        params, counters = params.cpu().numpy(), counters.cpu().numpy()
        for t in (0, 17, 49):
            out, st = orc_stored.q_opt(demo_viz["kp_data"][t], pm.astype(np.uint8), ones, demo_viz["qpos"][t], fs.lb, fs.ub)
            np.testing.assert_array_equal(params[t], out)
            assert counters[t, 0] == st["iter_num"]
        dq = np.abs((params - demo_viz["qpos"]) * pm).max(axis=1)
        assert np.median(counters[:, 0]) <= 3 and np.median(dq) <= 5e-4
        if pi >= 2:
            assert np.median(dq) <= 1e-4


@pytest.mark.gpu
def test_gpu_fit_offsets_on_the_golden_frames(rodent_setup_legacy, rodent_cfg, demo_viz):
    """Stac.fit_offsets with the real N_ITERS = 6 through the HIP engine == the oracle-driven restatement, and
    near the stored offsets / below the stored fit's marker error (pin 4)."""
    from test_gpu_stac import _cfg

    from stac_mjx_amd.stac import Stac

    fs = rodent_setup_legacy
    cfg = _cfg(rodent_cfg, n_fit_frames=50)
    assert int(cfg.model.N_ITERS) == 6
    data = Stac(None, cfg, fs.kp_names, setup=fs, verbose=False).fit_offsets(demo_viz["kp_data"])
    ref_off, ref = oracle_fit_offsets(fs, rodent_cfg, demo_viz["kp_data"], 6)
    np.testing.assert_array_equal(data.offsets, ref_off)
    np.testing.assert_array_equal(data.qpos, ref["qpos"])
    np.testing.assert_array_equal(data.marker_sites, ref["marker_sites"])
    assert np.linalg.norm(data.offsets - demo_viz["offsets"], axis=-1).mean() * 1e3 <= 1.3
    assert marker_error_mm(data.marker_sites, demo_viz["kp_data"]) <= STORED_MEAN_MM
