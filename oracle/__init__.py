"""CPU oracle for the STAC hot path -- TEST INFRASTRUCTURE ONLY.

ctypes front-end of ``oracle/stac_oracle.c`` (see ``stac_oracle.h`` for what each function
restates and the parity status: FK and m_opt pinned exactly, the q_phase pinned in marker space and at its
stopping rule to the reference's stored fit demos/demo_viz.p -- tests/test_pin_demo_viz.py).

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import this package.  Nothing under ``stac_mjx_amd/`` does.
"""

from __future__ import annotations

import ctypes as C
import subprocess
from pathlib import Path

import numpy as np

_HERE = Path(__file__).resolve().parent

_i32p = C.POINTER(C.c_int32)
_f32p = C.POINTER(C.c_float)
_u8p = C.POINTER(C.c_uint8)
_u32p = C.POINTER(C.c_uint32)


class OrcModel(C.Structure):
    _fields_ = [
        ("nbody", C.c_int32), ("njnt", C.c_int32), ("nq", C.c_int32), ("nsite", C.c_int32),
        ("body_parentid", _i32p), ("body_pos", _f32p), ("body_quat", _f32p),
        ("body_jntadr", _i32p), ("body_jntnum", _i32p),
        ("jnt_type", _i32p), ("jnt_qposadr", _i32p), ("jnt_bodyid", _i32p),
        ("jnt_pos", _f32p), ("jnt_axis", _f32p), ("qpos0", _f32p),
        ("site_bodyid", _i32p), ("site_pos", _f32p),
    ]  # fmt: skip


class OrcPgParams(C.Structure):
    _fields_ = [("tol", C.c_float), ("maxiter", C.c_int32), ("maxls", C.c_int32)]


class OrcLmParams(C.Structure):
    _fields_ = [("tol", C.c_float), ("maxiter", C.c_int32), ("lambda0", C.c_float)]


class OrcPgState(C.Structure):
    _fields_ = [
        ("iter_num", C.c_int32), ("stepsize", C.c_float), ("error", C.c_float), ("t", C.c_float),
        ("ls_evals", C.c_int32), ("grad_evals", C.c_int32), ("loss", C.c_float),
    ]  # fmt: skip

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


def build(force: bool = False) -> None:
    """Compile the oracle's C restatement (gcc, a few seconds)."""
    libs = [_HERE / "libstac_oracle_f32.so", _HERE / "libstac_oracle_f64.so"]
    src = [_HERE / "stac_oracle.c", _HERE / "stac_oracle.h"]
    stale = force or any((not l.exists()) or l.stat().st_mtime < max(s.stat().st_mtime for s in src) for l in libs)
    if stale:
        subprocess.run(["make", "-C", str(_HERE), "-B"], check=True, capture_output=True)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _u8(a):
    return np.ascontiguousarray(np.asarray(a).astype(np.uint8))


def _p(a, t):
    return a.ctypes.data_as(t) if a is not None else None


class Oracle:
    """CPU restatement bound to one model (``ModelTables``-like object with numpy fields)."""

    def __init__(self, tables, *, tol=1e-4, maxiter=400, maxls=15, precision="f32"):
        build()
        self.lib = C.CDLL(str(_HERE / f"libstac_oracle_{precision}.so"))
        self.precision = precision
        self.t = tables
        self.nq, self.nbody, self.njnt, self.K = tables.nq, tables.nbody, tables.njnt, tables.nsite
        self._keep = {}
        m = OrcModel()
        m.nbody, m.njnt, m.nq, m.nsite = tables.nbody, tables.njnt, tables.nq, tables.nsite
        for name, ctype, dt in [
            ("body_parentid", _i32p, np.int32), ("body_pos", _f32p, np.float32), ("body_quat", _f32p, np.float32),
            ("body_jntadr", _i32p, np.int32), ("body_jntnum", _i32p, np.int32), ("jnt_type", _i32p, np.int32),
            ("jnt_qposadr", _i32p, np.int32), ("jnt_bodyid", _i32p, np.int32), ("jnt_pos", _f32p, np.float32),
            ("jnt_axis", _f32p, np.float32), ("qpos0", _f32p, np.float32), ("site_bodyid", _i32p, np.int32),
            ("site_pos", _f32p, np.float32),
        ]:  # fmt: skip
            arr = np.ascontiguousarray(getattr(tables, name), dtype=dt).copy()
            self._keep[name] = arr
            setattr(m, name, arr.ctypes.data_as(ctype))
        self.m = m
        self.params = OrcPgParams(tol, maxiter, maxls)
        L = self.lib
        L.orc_q_loss.restype = C.c_float
        L.orc_q_loss_d.restype = C.c_double
        L.orc_max_threads.restype = C.c_int32

    # -- offsets -------------------------------------------------------------------------
    def set_site_pos(self, offsets):
        self._keep["site_pos"][...] = _f32(offsets).reshape(self.K, 3)

    def get_site_pos(self):
        return self._keep["site_pos"].copy()

    # -- kernels -------------------------------------------------------------------------
    def fk(self, qpos):
        q = _f32(qpos).copy()
        xpos = np.empty((self.nbody, 3), np.float32)
        xquat = np.empty((self.nbody, 4), np.float32)
        xanchor = np.empty((self.njnt, 3), np.float32)
        xaxis = np.empty((self.njnt, 3), np.float32)
        sx = np.empty((self.K, 3), np.float32)
        self.lib.orc_fk(C.byref(self.m), _p(q, _f32p), _p(xpos, _f32p), _p(xquat, _f32p), _p(xanchor, _f32p),
                        _p(xaxis, _f32p), _p(sx, _f32p))
        return dict(qpos=q, xpos=xpos, xquat=xquat, xanchor=xanchor, xaxis=xaxis, site_xpos=sx)

    def q_loss(self, q, kp, qs_to_opt, kps_to_opt, initial_q, with_grad=True):
        q, kp, iq = _f32(q), _f32(kp), _f32(initial_q)
        qs, ks = _u8(qs_to_opt), _u8(kps_to_opt)
        g = np.empty(self.nq, np.float32) if with_grad else None
        loss = self.lib.orc_q_loss_d(C.byref(self.m), _p(q, _f32p), _p(kp, _f32p), _p(qs, _u8p), _p(ks, _u8p),
                                   _p(iq, _f32p), _p(g, _f32p))
        return float(loss), g

    def q_opt(self, kp, qs_to_opt, kps_to_opt, q0, lb, ub):
        kp, q0, lb, ub = _f32(kp), _f32(q0), _f32(lb), _f32(ub)
        qs, ks = _u8(qs_to_opt), _u8(kps_to_opt)
        out = np.empty(self.nq, np.float32)
        st = OrcPgState()
        self.lib.orc_q_opt(C.byref(self.m), C.byref(self.params), _p(kp, _f32p), _p(qs, _u8p), _p(ks, _u8p),
                           _p(q0, _f32p), _p(lb, _f32p), _p(ub, _f32p), _p(out, _f32p), C.byref(st))
        return out, st.as_dict()

    def m_partial(self, keypoints, q):
        kp, q = _f32(keypoints), _f32(q)
        T = kp.shape[0]
        part = np.empty(3 * self.K + 2, np.float32)
        self.lib.orc_m_partial(C.byref(self.m), _p(kp, _f32p), _p(q, _f32p), C.c_int32(T), _p(part, _f32p))
        return part

    def m_finish(self, partial, initial_offsets, is_regularized, reg_coef):
        part, m0, d = _f32(partial), _f32(initial_offsets), _f32(is_regularized)
        out = np.empty((self.K, 3), np.float32)
        err = C.c_float()
        self.lib.orc_m_finish(C.c_int32(self.K), _p(part, _f32p), _p(m0, _f32p), _p(d, _f32p), C.c_float(reg_coef),
                              _p(out, _f32p), C.byref(err))
        return out, float(err.value)

    def m_opt(self, keypoints, q, initial_offsets, is_regularized, reg_coef):
        return self.m_finish(self.m_partial(keypoints, q), initial_offsets, is_regularized, reg_coef)

    # -- drivers -------------------------------------------------------------------------
    def root_optimization(self, kp_clip, qpos, lb, ub, trunk_kps, root_kp_idx, root_dims=7, frame=0):
        kp, q, lb, ub = _f32(kp_clip), _f32(qpos).copy(), _f32(lb), _f32(ub)
        tk = _u8(trunk_kps)
        st = OrcPgState()
        self.lib.orc_root_optimization(C.byref(self.m), C.byref(self.params), _p(kp, _f32p), C.c_int32(frame),
                                       C.c_int32(root_kp_idx), C.c_int32(root_dims), _p(lb, _f32p), _p(ub, _f32p),
                                       _p(tk, _u8p), _p(q, _f32p), C.byref(st))
        return q, st.as_dict()

    def pose_optimization(self, kp_clip, qpos, lb, ub, part_masks):
        kp, q, lb, ub = _f32(kp_clip), _f32(qpos).copy(), _f32(lb), _f32(ub)
        F = kp.shape[0]
        pm = _u8(part_masks).reshape(-1, self.nq) if len(part_masks) else np.zeros((0, self.nq), np.uint8)
        P = pm.shape[0]
        out = dict(
            qpos=np.empty((F, self.nq), np.float32), xpos=np.empty((F, self.nbody, 3), np.float32),
            xquat=np.empty((F, self.nbody, 4), np.float32), marker_sites=np.empty((F, self.K, 3), np.float32),
            frame_error=np.empty(F, np.float32), counters=np.empty((F, 4), np.uint32),
        )  # fmt: skip
        self.lib.orc_pose_optimization(
            C.byref(self.m), C.byref(self.params), _p(kp, _f32p), C.c_int32(F), _p(lb, _f32p), _p(ub, _f32p),
            _p(pm, _u8p) if P else None, C.c_int32(P), _p(q, _f32p), _p(out["qpos"], _f32p), _p(out["xpos"], _f32p),
            _p(out["xquat"], _f32p), _p(out["marker_sites"], _f32p), _p(out["frame_error"], _f32p),
            _p(out["counters"], _u32p))  # fmt: skip
        out["carry_qpos"] = q
        return out

    def ik_clips(self, kp, lb, ub, part_masks, trunk_kps, root_kp_idx, root_dims=7, do_root_opt=True,
                 q_init=None, nthreads=0, want_bodies=True):
        kp, lb, ub = _f32(kp), _f32(lb), _f32(ub)
        Cn, F = kp.shape[0], kp.shape[1]
        pm = _u8(part_masks).reshape(-1, self.nq) if len(part_masks) else np.zeros((0, self.nq), np.uint8)
        P = pm.shape[0]
        tk = _u8(trunk_kps)
        qi = _f32(q_init) if q_init is not None else None
        out = dict(
            qpos=np.empty((Cn, F, self.nq), np.float32),
            xpos=np.empty((Cn, F, self.nbody, 3), np.float32) if want_bodies else None,
            xquat=np.empty((Cn, F, self.nbody, 4), np.float32) if want_bodies else None,
            marker_sites=np.empty((Cn, F, self.K, 3), np.float32),
            frame_error=np.empty((Cn, F), np.float32), counters=np.empty((Cn, F, 4), np.uint32),
        )  # fmt: skip
        self.lib.orc_ik_clips(
            C.byref(self.m), C.byref(self.params), _p(kp, _f32p), C.c_int32(Cn), C.c_int32(F), _p(lb, _f32p),
            _p(ub, _f32p), _p(pm, _u8p) if P else None, C.c_int32(P), _p(tk, _u8p), C.c_int32(root_kp_idx),
            C.c_int32(root_dims), C.c_int32(1 if do_root_opt else 0), _p(qi, _f32p), _p(out["qpos"], _f32p),
            _p(out["xpos"], _f32p), _p(out["xquat"], _f32p), _p(out["marker_sites"], _f32p),
            _p(out["frame_error"], _f32p), _p(out["counters"], _u32p), C.c_int32(nthreads))  # fmt: skip
        return out

    # -- optional LM solver (not the reference's algorithm; see stac_oracle.h) -----------------------------------
    def q_opt_lm(self, kp, qs_to_opt, kps_to_opt, q0, lb, ub, maxiter=40, lambda0=1e-2):
        kp, q0, lb, ub = _f32(kp), _f32(q0), _f32(lb), _f32(ub)
        qs, ks = _u8(qs_to_opt), _u8(kps_to_opt)
        out = np.empty(self.nq, np.float32)
        st = OrcPgState()
        lp = OrcLmParams(self.params.tol, maxiter, lambda0)
        self.lib.orc_q_opt_lm(C.byref(self.m), C.byref(lp), _p(kp, _f32p), _p(qs, _u8p), _p(ks, _u8p), _p(q0, _f32p),
                              _p(lb, _f32p), _p(ub, _f32p), _p(out, _f32p), C.byref(st))
        return out, st.as_dict()

    def lm_jac_check(self, q, kp):
        """max |J^T f - grad| of the LM solver's Jacobian columns against the analytic gradient (stac_oracle.c::orc_lm_jac_check)."""
        q, kp = _f32(q), _f32(kp)
        self.lib.orc_lm_jac_check.restype = C.c_double
        return float(self.lib.orc_lm_jac_check(C.byref(self.m), _p(q, _f32p), _p(kp, _f32p)))

    def ik_clips_lm(self, kp, lb, ub, part_masks, trunk_kps, root_kp_idx, root_dims=7, do_root_opt=True, q_init=None,
                    nthreads=0, want_bodies=True, maxiter=40, lambda0=1e-2):
        kp, lb, ub = _f32(kp), _f32(lb), _f32(ub)
        Cn, F = kp.shape[0], kp.shape[1]
        pm = _u8(part_masks).reshape(-1, self.nq) if len(part_masks) else np.zeros((0, self.nq), np.uint8)
        P = pm.shape[0]
        tk = _u8(trunk_kps)
        qi = _f32(q_init) if q_init is not None else None
        lp = OrcLmParams(self.params.tol, maxiter, lambda0)
        out = dict(
            qpos=np.empty((Cn, F, self.nq), np.float32),
            xpos=np.empty((Cn, F, self.nbody, 3), np.float32) if want_bodies else None,
            xquat=np.empty((Cn, F, self.nbody, 4), np.float32) if want_bodies else None,
            marker_sites=np.empty((Cn, F, self.K, 3), np.float32),
            frame_error=np.empty((Cn, F), np.float32), counters=np.empty((Cn, F, 4), np.uint32),
        )  # fmt: skip
        self.lib.orc_ik_clips_lm(
            C.byref(self.m), C.byref(lp), _p(kp, _f32p), C.c_int32(Cn), C.c_int32(F), _p(lb, _f32p), _p(ub, _f32p),
            _p(pm, _u8p) if P else None, C.c_int32(P), _p(tk, _u8p), C.c_int32(root_kp_idx), C.c_int32(root_dims),
            C.c_int32(1 if do_root_opt else 0), _p(qi, _f32p), _p(out["qpos"], _f32p), _p(out["xpos"], _f32p),
            _p(out["xquat"], _f32p), _p(out["marker_sites"], _f32p), _p(out["frame_error"], _f32p),
            _p(out["counters"], _u32p), C.c_int32(nthreads))  # fmt: skip
        return out

    def max_threads(self):
        return int(self.lib.orc_max_threads())
