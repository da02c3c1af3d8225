"""Phase drivers with the reference's signatures (``stac_mjx/compute_stac.py``).

Two layers:

* seam-level drivers ``root_optimization`` / ``pose_optimization`` / ``offset_optimization`` taking a
  duck-typed ``stac_core_obj`` first -- the reference's L3 functions (compute_stac.py:17-278), one
  ``q_opt`` per solve.  They exist for protocol parity (same call counts, seeding and outputs as
  tests/unit/test_compute_stac.py) and for callers that bring their own core object.
* the batched entry points the engine is built for live in ``stac.py`` (one ``stac_q_phase`` launch
  per phase: the per-frame Python loop of the reference is inside the kernel).
"""

from __future__ import annotations

import time

import numpy as np
import torch

from . import prng, utils
from .mjcf import JNT_SLIDE


def root_optimization(stac_core_obj, mjx_model, mjx_data, kp_data, root_kp_idx, lb, ub, site_idxs, trunk_kps, frame: int = 0):
    """compute_stac.py:17-104: seed root translation from the root keypoint, two root-only solves."""
    root_dims = 4 if int(mjx_model.jnt_type[0]) == JNT_SLIDE else 7
    q0 = mjx_data.qpos.clone()
    root_xyz = torch.as_tensor(kp_data[frame, 3 * root_kp_idx : 3 * root_kp_idx + 3]).to(q0)
    q0[:3] = root_xyz
    qs_to_opt = torch.zeros(q0.shape[0], dtype=torch.bool)
    qs_to_opt[:root_dims] = True
    kps_to_opt = torch.as_tensor(np.repeat(np.asarray(torch.as_tensor(trunk_kps).cpu()), 3))
    mjx_data, res = stac_core_obj.q_opt(mjx_model, mjx_data, kp_data[frame, :], qs_to_opt, kps_to_opt, q0, lb, ub, site_idxs)
    mjx_data = utils.replace_qs(mjx_model, mjx_data, utils.make_qs(q0, qs_to_opt, res.params))
    q0 = mjx_data.qpos.clone()
    q0[:3] = root_xyz
    mjx_data, res = stac_core_obj.q_opt(mjx_model, mjx_data, kp_data[frame, :], qs_to_opt, kps_to_opt, q0, lb, ub, site_idxs)
    mjx_data = utils.replace_qs(mjx_model, mjx_data, utils.make_qs(q0, qs_to_opt, res.params))
    return mjx_data


def offset_optimization(stac_core_obj, mjx_model, mjx_data, kp_data, offsets, q, n_sample_frames, is_regularized,
                        site_idxs, m_reg_coef, time_indices=None):
    """compute_stac.py:107-167.  ``time_indices`` overrides the PRNGKey(0) permutation when given."""
    n = kp_data.shape[0]
    if time_indices is None:
        time_indices = prng.sample_time_indices(n, n_sample_frames, seed=0)
    idx = torch.as_tensor(np.asarray(time_indices), dtype=torch.long)
    kp_t = torch.as_tensor(kp_data)
    q_t = torch.as_tensor(q)
    keypoints = kp_t[idx.to(kp_t.device)]
    qs = q_t[idx.to(q_t.device)]
    res = stac_core_obj.m_opt(mjx_model, mjx_data, keypoints, qs, offsets, is_regularized, m_reg_coef, site_idxs)
    offset_opt_param = res.params
    mjx_model = utils.set_site_pos(mjx_model, offset_opt_param, site_idxs)
    mjx_data = utils.kinematics(mjx_model, mjx_data)
    return mjx_model, mjx_data, offset_opt_param


def pose_optimization(stac_core_obj, mjx_model, mjx_data, kp_data, lb, ub, site_idxs, indiv_parts):
    """compute_stac.py:170-278: per frame one full-body solve + one solve per part, warm-started."""
    s = time.time()
    qposes, xposes, xquats, marker_sites, frame_time, frame_error = [], [], [], [], [], []
    nq = mjx_model.nq
    kps_to_opt = torch.ones(kp_data.shape[1], dtype=torch.bool)
    qs_to_opt = torch.ones(nq, dtype=torch.bool)
    for n_frame in range(kp_data.shape[0]):
        t0 = time.time()
        q0 = mjx_data.qpos.clone()
        mjx_data, res = stac_core_obj.q_opt(mjx_model, mjx_data, kp_data[n_frame, :], qs_to_opt, kps_to_opt, q0, lb, ub, site_idxs)
        mjx_data = utils.replace_qs(mjx_model, mjx_data, res.params)
        for part in indiv_parts:
            q0 = mjx_data.qpos.clone()
            mjx_data, res = stac_core_obj.q_opt(mjx_model, mjx_data, kp_data[n_frame, :], part, kps_to_opt, q0, lb, ub, site_idxs)
            mjx_data = utils.replace_qs(mjx_model, mjx_data, utils.make_qs(q0, part, res.params))
        qposes.append(mjx_data.qpos)
        xposes.append(mjx_data.xpos)
        xquats.append(mjx_data.xquat)
        marker_sites.append(utils.get_site_xpos(mjx_data, site_idxs))
        frame_time.append(time.time() - t0)
        frame_error.append(res.state.error)
    _ = time.time() - s
    return mjx_data, torch.stack(qposes), xposes, xquats, marker_sites, frame_time, frame_error
