"""Test-side restatements of the reference's host sequencing, driven by the CPU oracle.

Shared by the CPU pins (tests/test_pin_demo_viz.py) and the GPU parity tests: the GPU path must equal these
bit for bit, and these are pinned -- in marker space -- to the reference's own stored fit (demos/demo_viz.p).
"""

from __future__ import annotations

import numpy as np


def oracle_chain_passes(orc, fs, kp, n_passes, q0=None):
    """stac.py:277-311 with fixed offsets: root optimisation on frame 0, then ``n_passes`` warm-started
    ``pose_optimization`` passes over the same frames, the carried qpos crossing the passes (stac.py:300-301).
    Returns the list of per-pass outputs."""
    q = fs.tables.qpos0 if q0 is None else q0
    if fs.do_root_opt:
        q, _ = orc.root_optimization(kp, q, fs.lb, fs.ub, fs.trunk_kps, fs.root_kp_idx, fs.root_dims)
    outs = []
    for _ in range(n_passes):
        out = orc.pose_optimization(kp, q, fs.lb, fs.ub, fs.part_masks)
        q = out["carry_qpos"]
        outs.append(out)
    return outs


def oracle_fit_offsets(fs, cfgm, kp, n_iters, time_indices=None, history=None):
    """Stac.fit_offsets (stac.py:253-354) driven by the CPU oracle: root optimisation, then ``n_iters`` x
    (pose pass, closed-form offsets regularised toward the previous iterate), then the final pose pass."""
    from oracle import Oracle
    from stac_mjx_amd.prng import sample_time_indices

    orc = Oracle(fs.tables, tol=float(cfgm["FTOL"]), maxiter=int(cfgm["N_ITER_Q"]))
    offsets = fs.tables.site_pos.copy()
    q = fs.tables.qpos0
    if fs.do_root_opt:
        q, _ = orc.root_optimization(kp, q, fs.lb, fs.ub, fs.trunk_kps, fs.root_kp_idx, fs.root_dims)
    idx = sample_time_indices(kp.shape[0], int(cfgm["N_SAMPLE_FRAMES"])) if time_indices is None else np.asarray(time_indices)
    for _ in range(n_iters):
        out = orc.pose_optimization(kp, q, fs.lb, fs.ub, fs.part_masks)
        q = out["carry_qpos"]
        offsets, _ = orc.m_opt(kp[idx], out["qpos"][idx], offsets, fs.is_regularized, float(cfgm["M_REG_COEF"]))
        orc.set_site_pos(offsets)
        if history is not None:
            history.append((out, offsets.copy()))
    out = orc.pose_optimization(kp, q, fs.lb, fs.ub, fs.part_masks)
    return offsets, out


def marker_error_mm(marker_sites, kp):
    """Mean marker-to-keypoint distance in millimetres."""
    m = np.asarray(marker_sites)
    return float(np.linalg.norm(m.reshape(-1, m.shape[-2], 3) - np.asarray(kp).reshape(-1, m.shape[-2], 3), axis=-1).mean() * 1e3)


def pg_residual(orc, fs, q, kp, mask):
    """The stopping residual of jaxopt's ProjectedGradient at q for the coordinates in ``mask``:
    || clip(q - grad) - q ||_2 over the masked coordinates."""
    m = np.asarray(mask).astype(np.uint8)
    _, g = orc.q_loss(q, kp, m, np.ones(3 * fs.tables.nsite, np.uint8), q)
    r = (np.clip(q - g * m, fs.lb, fs.ub) - q) * m
    return float(np.linalg.norm(r))
