"""ctypes binding of ``libstac_hip.so`` (the C ABI of ``include/stac_hip.h``).

Device memory, streams and (in ``dist.py``) the process group come from PyTorch-ROCm; every
compute call goes through the C ABI into hand-written HIP kernels.  There is no CPU or eager
PyTorch fallback: without the built extension or without a GPU the constructor raises.
"""

from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

import numpy as np
import torch

from .build import LIB, STAMP, source_digest

ABI_VERSION = 3  # include/stac_hip.h: STAC_HIP_ABI_VERSION

_i32p = C.POINTER(C.c_int32)
_f32p = C.POINTER(C.c_float)
_u8p = C.POINTER(C.c_uint8)


class StacModelTables(C.Structure):
    _fields_ = [
        ("nbody", C.c_int32), ("njnt", C.c_int32), ("nq", C.c_int32), ("nsite", C.c_int32),
        ("body_parentid", _i32p), ("body_pos", _f32p), ("body_quat", _f32p),
        ("body_jntadr", _i32p), ("body_jntnum", _i32p),
        ("jnt_type", _i32p), ("jnt_qposadr", _i32p), ("jnt_bodyid", _i32p),
        ("jnt_pos", _f32p), ("jnt_axis", _f32p), ("qpos0", _f32p),
        ("site_bodyid", _i32p), ("site_pos", _f32p), ("lb", _f32p), ("ub", _f32p),
    ]  # fmt: skip


class StacQParams(C.Structure):
    _fields_ = [("tol", C.c_float), ("maxiter", C.c_int32), ("maxls", C.c_int32), ("lanes_per_chain", C.c_int32),
                ("solver", C.c_int32), ("lm_lambda0", C.c_float)]


SOLVERS = {"pg": 0, "lm": 1}  # STAC_SOLVER_PG (the reference's algorithm, parity mode) / STAC_SOLVER_LM


class StacHipError(RuntimeError):
    pass


_LIB = None

# every symbol include/stac_hip.h declares
ABI_SYMBOLS = (
    "stac_last_error", "stac_abi_version", "stac_device_count", "stac_model_create", "stac_model_destroy",
    "stac_model_info", "stac_set_site_pos", "stac_get_site_pos", "stac_fk", "stac_q_solve", "stac_q_phase",
    "stac_m_phase_workspace_floats", "stac_m_phase_partial", "stac_m_phase_finish",
)  # fmt: skip


def load_library(path: Path | None = None):
    """Load ``libstac_hip.so``; raises if it has not been built (``__graft_entry__.build()``)."""
    global _LIB
    if _LIB is not None and path is None:
        return _LIB
    p = Path(path) if path else Path(os.environ.get("STAC_HIP_LIB", LIB))  # env: diagnostic builds only
    if not p.exists():
        raise StacHipError(f"HIP extension not built: {p} is missing (run `python -m stac_mjx_amd.build`)")
    if p == LIB and STAMP.exists() and STAMP.read_text().strip() != source_digest():
        # the library is git-ignored and travels separately from the sources: never call a stale one with today's
        # argument layout (python -m stac_mjx_amd.build rebuilds; STAC_HIP_LIB selects a diagnostic build explicitly)
        raise StacHipError(f"{p} was built from other sources than the ones in this tree (stamp {STAMP.name} differs): "
                           "rebuild with `python -m stac_mjx_amd.build`")
    lib = C.CDLL(str(p))
    lib.stac_abi_version.restype = C.c_int32
    got = int(lib.stac_abi_version())
    if got != ABI_VERSION:
        raise StacHipError(f"{p} exports ABI version {got}, this binding needs {ABI_VERSION}: rebuild with "
                           "`python -m stac_mjx_amd.build`")
    lib.stac_last_error.restype = C.c_char_p
    lib.stac_model_create.restype = C.c_void_p
    lib.stac_model_create.argtypes = [C.POINTER(StacModelTables)]
    lib.stac_model_destroy.argtypes = [C.c_void_p]
    lib.stac_m_phase_workspace_floats.restype = C.c_int64
    lib.stac_m_phase_workspace_floats.argtypes = [C.c_void_p, C.c_int32]
    vp = C.c_void_p
    lib.stac_model_info.argtypes = [vp, _i32p]
    lib.stac_set_site_pos.argtypes = [vp, vp, vp]
    lib.stac_get_site_pos.argtypes = [vp, vp, vp]
    lib.stac_fk.argtypes = [vp, vp, C.c_int32, vp, vp, vp, vp, vp]
    lib.stac_q_solve.argtypes = [vp, C.POINTER(StacQParams), vp, vp, _u8p, _u8p, _f32p, _f32p, C.c_int32, vp, vp, vp, vp]
    lib.stac_q_phase.argtypes = [vp, C.POINTER(StacQParams), vp, vp, _u8p, _u8p, C.c_int32, C.c_int32, C.c_int32,
                                 C.c_int32, C.c_int32, C.c_int32, vp, vp, vp, vp, vp, vp, vp, vp]
    lib.stac_m_phase_partial.argtypes = [vp, vp, vp, C.c_int32, vp, vp, vp]
    lib.stac_m_phase_finish.argtypes = [vp, vp, vp, vp, C.c_float, vp, vp, vp]
    if path is None:
        _LIB = lib
    return lib


def _ptr(t: torch.Tensor | None):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def _host_u8(a, n):
    a = np.ascontiguousarray(np.asarray(a).astype(np.uint8)).reshape(-1)
    if a.size != n:
        raise ValueError(f"mask has {a.size} entries, expected {n}")
    return a


class Engine:
    """One compiled model resident on one GPU.

    ``solver="lm"`` (engine extension, not the reference's algorithm) runs at most ``lm_maxiter`` accepted steps per solve:
    the default is 20 since round 3 (it was 40; the steps beyond 20 buy 0.001 mm of marker RMSE on the bench batch and set
    the tail of a launch, ``profiles/r03/lm_cap_sweep.txt``) -- pass ``lm_maxiter=40`` / ``stac.lm_maxiter: 40`` for the old
    behaviour.  ``solver="pg"`` (default, parity mode) is not affected."""

    def __init__(self, tables, lb, ub, *, tol=1e-4, maxiter=400, maxls=15, lanes_per_chain=0, device=None,
                 solver="pg", lm_maxiter=20, lm_lambda0=1e-2):
        self.lib = load_library()
        if not torch.cuda.is_available():
            raise StacHipError("no GPU visible: the STAC engine has no CPU fallback")
        self.device = torch.device(device if device is not None else f"cuda:{torch.cuda.current_device()}")
        torch.cuda.set_device(self.device)
        self.nq, self.nbody, self.njnt, self.K = tables.nq, tables.nbody, tables.njnt, tables.nsite
        self.lb = np.ascontiguousarray(lb, dtype=np.float32).copy()
        self.ub = np.ascontiguousarray(ub, dtype=np.float32).copy()
        self.params = StacQParams(float(tol), int(maxiter), int(maxls), int(lanes_per_chain), SOLVERS["pg"], 0.0)
        # q_phase may use the optional LM solver; q_solve (the StacCore.q_opt seam) always runs the reference's PG
        self.phase_params = StacQParams(float(tol), int(lm_maxiter if solver == "lm" else maxiter), int(maxls),
                                        int(lanes_per_chain), SOLVERS[solver], float(lm_lambda0))
        self.solver = solver
        t = StacModelTables()
        t.nbody, t.njnt, t.nq, t.nsite = tables.nbody, tables.njnt, tables.nq, tables.nsite
        keep = {}
        for name, ctype, dt, src in [
            ("body_parentid", _i32p, np.int32, tables.body_parentid), ("body_pos", _f32p, np.float32, tables.body_pos),
            ("body_quat", _f32p, np.float32, tables.body_quat), ("body_jntadr", _i32p, np.int32, tables.body_jntadr),
            ("body_jntnum", _i32p, np.int32, tables.body_jntnum), ("jnt_type", _i32p, np.int32, tables.jnt_type),
            ("jnt_qposadr", _i32p, np.int32, tables.jnt_qposadr), ("jnt_bodyid", _i32p, np.int32, tables.jnt_bodyid),
            ("jnt_pos", _f32p, np.float32, tables.jnt_pos), ("jnt_axis", _f32p, np.float32, tables.jnt_axis),
            ("qpos0", _f32p, np.float32, tables.qpos0), ("site_bodyid", _i32p, np.int32, tables.site_bodyid),
            ("site_pos", _f32p, np.float32, tables.site_pos), ("lb", _f32p, np.float32, lb), ("ub", _f32p, np.float32, ub),
        ]:  # fmt: skip
            arr = np.ascontiguousarray(src, dtype=dt)
            keep[name] = arr
            setattr(t, name, arr.ctypes.data_as(ctype))
        self._h = self.lib.stac_model_create(C.byref(t))
        if not self._h:
            raise StacHipError("stac_model_create failed: " + self._err())
        info = (C.c_int32 * 8)()
        self._check(self.lib.stac_model_info(self._h, info))
        self.info = dict(zip(("nbody", "njnt", "nq", "K", "n_active_bodies", "n_active_joints", "n_levels", "max_level_width"), info))

    # -- plumbing ---------------------------------------------------------------------------
    def _err(self) -> str:
        return self.lib.stac_last_error().decode("utf-8", "replace")

    def _check(self, rc: int):
        if rc != 0:
            raise StacHipError(f"libstac_hip error {rc}: {self._err()}")

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _dev(self, a, dtype=torch.float32):
        if isinstance(a, torch.Tensor):
            return a.to(device=self.device, dtype=dtype).contiguous()
        return torch.as_tensor(np.ascontiguousarray(a), dtype=dtype).to(self.device)

    def close(self):
        if getattr(self, "_h", None):
            self.lib.stac_model_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_lanes_per_chain(self, g: int):
        self.params.lanes_per_chain = int(g)
        self.phase_params.lanes_per_chain = int(g)

    # -- offsets ------------------------------------------------------------------------------
    def set_site_pos(self, offsets):
        off = self._dev(offsets).reshape(self.K, 3)
        # `off` may be a temporary: its block returns to torch's caching allocator, which hands it out again only to
        # work queued behind this call on the same stream -- no host synchronisation needed
        self._check(self.lib.stac_set_site_pos(self._h, _ptr(off), self._stream()))

    def get_site_pos(self) -> torch.Tensor:
        out = torch.empty((self.K, 3), dtype=torch.float32, device=self.device)
        self._check(self.lib.stac_get_site_pos(self._h, _ptr(out), self._stream()))
        return out

    # -- kernels ------------------------------------------------------------------------------
    def fk(self, qpos, want=("qpos", "xpos", "xquat", "site_xpos")):
        q = self._dev(qpos).reshape(-1, self.nq)
        N = q.shape[0]
        mk = lambda *s: torch.empty(s, dtype=torch.float32, device=self.device)
        out = {
            "qpos": mk(N, self.nq) if "qpos" in want else None,
            "xpos": mk(N, self.nbody, 3) if "xpos" in want else None,
            "xquat": mk(N, self.nbody, 4) if "xquat" in want else None,
            "site_xpos": mk(N, self.K, 3) if "site_xpos" in want else None,
        }
        self._check(self.lib.stac_fk(self._h, _ptr(q), N, _ptr(out["qpos"]), _ptr(out["xpos"]), _ptr(out["xquat"]),
                                     _ptr(out["site_xpos"]), self._stream()))
        return out

    def q_solve(self, kp, q0, qs_to_opt, kps_to_opt, lb=None, ub=None):
        """Batched ``StacCore.q_opt``: kp[N,3K], q0[N,nq] -> params[N,nq], state[N,4], counters[N,4].
        ``lb`` / ``ub`` (host, [nq]) give the box of this call; None = the bounds the engine was built with."""
        kp = self._dev(kp).reshape(-1, 3 * self.K)
        q0 = self._dev(q0).reshape(-1, self.nq)
        N = kp.shape[0]
        qs = _host_u8(qs_to_opt, self.nq)
        ks = _host_u8(kps_to_opt, 3 * self.K)
        params = torch.empty((N, self.nq), dtype=torch.float32, device=self.device)
        state = torch.empty((N, 4), dtype=torch.float32, device=self.device)
        counters = torch.empty((N, 4), dtype=torch.int32, device=self.device)
        if (lb is None) != (ub is None):
            raise ValueError("lb and ub must be given together")
        lbp = ubp = None
        if lb is not None:
            lb_h = np.ascontiguousarray(np.asarray(lb, dtype=np.float32).reshape(-1))
            ub_h = np.ascontiguousarray(np.asarray(ub, dtype=np.float32).reshape(-1))
            if lb_h.size != self.nq or ub_h.size != self.nq:
                raise ValueError(f"lb / ub must have {self.nq} entries")
            lbp, ubp = lb_h.ctypes.data_as(_f32p), ub_h.ctypes.data_as(_f32p)
        self._check(self.lib.stac_q_solve(self._h, C.byref(self.params), _ptr(kp), _ptr(q0), qs.ctypes.data_as(_u8p),
                                          ks.ctypes.data_as(_u8p), lbp, ubp, N, _ptr(params), _ptr(state),
                                          _ptr(counters), self._stream()))
        return params, state, counters

    def q_phase(self, kp, *, part_masks, trunk_kps=None, root_kp_idx=-1, root_dims=7, do_root_opt=False, q_init=None,
                want_bodies=True, want_markers=True, want_carry=True, out=None):
        """The q_phase of C clips x F frames (``stac_q_phase``).  kp: [C,F,3K].  ``want_carry=False``: no ``carry_qpos`` (the pose
        of every clip's last frame, [C,nq]: the warm start of a next pass) -- with one frame per clip it is ``qpos`` over again."""
        kp = self._dev(kp)
        if kp.dim() != 3 or kp.shape[2] != 3 * self.K:
            raise ValueError(f"kp must be [C, F, {3 * self.K}]")
        Cn, F = kp.shape[0], kp.shape[1]
        pm = np.ascontiguousarray(np.asarray(part_masks).astype(np.uint8)).reshape(-1, self.nq) if len(part_masks) else np.zeros((0, self.nq), np.uint8)
        P = pm.shape[0]
        tk = _host_u8(trunk_kps, self.K) if trunk_kps is not None else np.zeros(self.K, np.uint8)
        qi = self._dev(q_init).reshape(Cn, self.nq) if q_init is not None else None
        mk = lambda *s: torch.empty(s, dtype=torch.float32, device=self.device)
        o = out or {}
        res = {
            "qpos": o.get("qpos") if o.get("qpos") is not None else mk(Cn, F, self.nq),
            "frame_error": o.get("frame_error") if o.get("frame_error") is not None else mk(Cn, F),
            "counters": o.get("counters") if o.get("counters") is not None else torch.empty((Cn, F, 4), dtype=torch.int32, device=self.device),
            "carry_qpos": (o.get("carry_qpos") if o.get("carry_qpos") is not None else mk(Cn, self.nq)) if want_carry else None,
            "xpos": (o.get("xpos") if o.get("xpos") is not None else mk(Cn, F, self.nbody, 3)) if want_bodies else None,
            "xquat": (o.get("xquat") if o.get("xquat") is not None else mk(Cn, F, self.nbody, 4)) if want_bodies else None,
            "marker_sites": (o.get("marker_sites") if o.get("marker_sites") is not None else mk(Cn, F, self.K, 3)) if want_markers else None,
        }
        if self.solver == "pg":  # StacCore may have updated tol / maxiter on self.params
            self.phase_params.tol, self.phase_params.maxiter = self.params.tol, self.params.maxiter
        self._check(self.lib.stac_q_phase(
            self._h, C.byref(self.phase_params), _ptr(kp), _ptr(qi), pm.ctypes.data_as(_u8p) if P else None,
            tk.ctypes.data_as(_u8p), Cn, F, P, int(root_kp_idx), int(root_dims), 1 if do_root_opt else 0,
            _ptr(res["qpos"]), _ptr(res["frame_error"]), _ptr(res["counters"]), _ptr(res["carry_qpos"]),
            _ptr(res["xpos"]), _ptr(res["xquat"]), _ptr(res["marker_sites"]), self._stream()))  # fmt: skip
        res["_keepalive"] = (kp, qi)
        return res

    def m_partial(self, kp, q):
        kp = self._dev(kp).reshape(-1, 3 * self.K)
        q = self._dev(q).reshape(-1, self.nq)
        T = kp.shape[0]
        ws_n = int(self.lib.stac_m_phase_workspace_floats(self._h, T))
        ws = torch.empty(max(ws_n, 1), dtype=torch.float32, device=self.device)
        partial = torch.empty(3 * self.K + 2, dtype=torch.float32, device=self.device)
        # the workspace is a temporary of torch's stream-ordered caching allocator (see set_site_pos): no host sync
        self._check(self.lib.stac_m_phase_partial(self._h, _ptr(kp), _ptr(q), T, _ptr(ws), _ptr(partial), self._stream()))
        return partial

    def m_finish(self, partial, initial_offsets, is_regularized, reg_coef):
        partial = self._dev(partial)
        m0 = self._dev(initial_offsets).reshape(self.K, 3)
        d = self._dev(is_regularized).reshape(self.K, 3)
        out = torch.empty((self.K, 3), dtype=torch.float32, device=self.device)
        err = torch.empty(1, dtype=torch.float32, device=self.device)
        self._check(self.lib.stac_m_phase_finish(self._h, _ptr(partial), _ptr(m0), _ptr(d), C.c_float(float(reg_coef)),
                                                 _ptr(out), _ptr(err), self._stream()))
        return out, err

    def m_opt(self, kp, q, initial_offsets, is_regularized, reg_coef):
        return self.m_finish(self.m_partial(kp, q), initial_offsets, is_regularized, reg_coef)
