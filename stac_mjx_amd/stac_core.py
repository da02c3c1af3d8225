"""The solver seam: ``StacCore.q_opt`` / ``StacCore.m_opt`` backed by the HIP engine.

Mirrors ``stac_mjx/stac_core.py:175-275``: same method names, argument order and result shapes
(``res.params``, ``res.state.error``; ``MOptResult(params, error)``), so the phase drivers and the
reference's protocol tests (``tests/unit/test_compute_stac.py``) read the same.  The MJX model/data
arguments are replaced by light handles (:class:`ModelHandle`, :class:`DataHandle`) because the
model lives on the device inside the engine.
"""

from __future__ import annotations

from dataclasses import dataclass, replace
from typing import NamedTuple

import numpy as np
import torch


class MOptResult(NamedTuple):
    """Result of marker offset optimisation (``stac_core.py:20-24``)."""

    params: torch.Tensor  # [K, 3]
    error: torch.Tensor  # scalar


@dataclass
class PGState:
    """The fields of jaxopt's ``ProxGradState`` the reference reads or prints."""

    iter_num: int
    stepsize: float
    error: float
    t: float
    loss: float
    ls_evals: int = 0
    grad_evals: int = 0


class OptStep(NamedTuple):
    params: torch.Tensor  # [nq]
    state: PGState


@dataclass
class ModelHandle:
    """Stand-in for ``mjx.Model``: the engine owns the tables; ``site_pos`` mirrors the offsets."""

    engine: object
    nq: int
    jnt_type: np.ndarray
    site_pos: torch.Tensor

    def replace(self, **kw):
        return replace(self, **kw)


@dataclass
class DataHandle:
    """Stand-in for ``mjx.Data``: carries qpos and the FK outputs the drivers record."""

    qpos: torch.Tensor
    xpos: torch.Tensor | None = None
    xquat: torch.Tensor | None = None
    site_xpos: torch.Tensor | None = None

    def replace(self, **kw):
        return replace(self, **kw)


class StacCore:
    """Pose (projected gradient) and offset (closed form) optimisation on the GPU engine."""

    def __init__(self, engine, tol: float = 1e-5, n_iter_q: int = 400, site_idxs=None):
        self.engine = engine
        # model site ids of the engine's K fit sites (stac.py:227-235), when the caller knows them
        self.site_idxs = None if site_idxs is None else np.asarray(site_idxs).reshape(-1)
        self.tol = float(tol)
        self.n_iter_q = int(n_iter_q)
        engine.params.tol = self.tol
        engine.params.maxiter = self.n_iter_q

    def q_opt(self, mjx_model, mjx_data, marker_ref_arr, qs_to_opt, kps_to_opt, q0, lb=None, ub=None, site_idxs=None):
        """One ``StacCore.q_opt`` (``stac_core.py:193-235``).  ``lb`` / ``ub`` are the box of THIS call, as in the
        reference (``hyperparams_proj``); None = the bounds the engine was built with.  ``site_idxs`` is fixed at
        engine construction (the engine's K fit sites ARE ``site_idxs``): passing a different index set raises."""
        e = self.engine
        qs = np.asarray(torch.as_tensor(qs_to_opt).cpu()).astype(bool)
        ks = np.asarray(torch.as_tensor(kps_to_opt).cpu()).astype(bool)
        if (lb is None) != (ub is None):
            raise ValueError("lb and ub must be given together")
        if lb is not None:
            lb = np.asarray(torch.as_tensor(lb).cpu(), dtype=np.float32).reshape(-1)
            ub = np.asarray(torch.as_tensor(ub).cpu(), dtype=np.float32).reshape(-1)
            if np.array_equal(lb, e.lb) and np.array_equal(ub, e.ub):
                lb = ub = None  # the engine's own box: nothing to upload
        if site_idxs is not None:
            si = np.asarray(torch.as_tensor(site_idxs).cpu()).reshape(-1)
            if si.size != e.K or (self.site_idxs is not None and not np.array_equal(si, self.site_idxs)):
                raise ValueError("site_idxs differs from the fit sites this engine was built with")
        params, state, counters = e.q_solve(torch.as_tensor(marker_ref_arr).reshape(1, -1), torch.as_tensor(q0).reshape(1, -1), qs, ks,
                                            lb=lb, ub=ub)
        s = state[0].cpu().numpy()
        c = counters[0].cpu().numpy()
        st = PGState(iter_num=int(c[0]), stepsize=float(s[1]), error=float(s[0]), t=float(s[2]), loss=float(s[3]),
                     ls_evals=int(c[1]), grad_evals=int(c[2]))
        return mjx_data, OptStep(params=params[0], state=st)

    def m_opt(self, mjx_model, mjx_data, keypoints, q, initial_offsets, is_regularized, reg_coef, site_idxs=None):
        """``StacCore.m_opt`` (``stac_core.py:237-275``)."""
        off, err = self.engine.m_opt(keypoints, q, initial_offsets, is_regularized, float(reg_coef))
        return MOptResult(params=off, error=err[0])
