"""``Stac``: model set-up, ``fit_offsets`` / ``ik_only`` and output packing on the HIP engine.

Mirrors the hot-path part of ``stac_mjx/stac.py`` (:91-503; rendering is out of scope).  The
per-frame Python loops of the reference (``compute_stac.pose_optimization``) run inside one
``stac_q_phase`` kernel launch per phase; sequencing, warm starts, sampling and packing follow the
reference.
"""

from __future__ import annotations

from pathlib import Path

import numpy as np
import torch

from . import dist, prng, utils
from .engine import Engine
from .fit_model import FitSetup, build_fit_setup
from .io import StacData
from .stac_core import ModelHandle, StacCore


class Stac:
    """Skeletal registration on one GPU (or one rank of a multi-GPU job)."""

    def __init__(self, xml_path, cfg, kp_names, *, setup: FitSetup | None = None, device=None, verbose: bool = True):
        self.cfg = cfg
        self._kp_names = list(kp_names)
        self._xml_path = Path(xml_path) if xml_path is not None else None
        self.verbose = verbose
        self.setup = setup if setup is not None else build_fit_setup(self._xml_path, cfg.model, self._kp_names)
        s = self.setup
        self._lb, self._ub, self._part_names = s.lb, s.ub, s.part_names
        self._indiv_parts = s.part_masks
        self._trunk_kps = s.trunk_kps
        self._root_kp_idx = s.root_kp_idx
        self._is_regularized = s.is_regularized
        self._body_names = s.tables.body_names
        self._freejoint, self._slidejoint, self._fixed = s.freejoint, s.slidejoint, s.fixed_root
        stac_cfg = cfg.stac
        # `stac.solver` is an engine extension of the config surface: "pg" (default) = the reference's projected
        # gradient, reproduced exactly; "lm" = optional Levenberg-Marquardt solver (faster, fits the markers at
        # least as well, but does not reproduce the reference's truncated iterates).
        self.engine = Engine(s.tables, s.lb, s.ub, tol=float(cfg.model.FTOL), maxiter=int(cfg.model.N_ITER_Q),
                             lanes_per_chain=int(stac_cfg.get("lanes_per_chain", 0) or 0), device=device,
                             solver=str(stac_cfg.get("solver", "pg") or "pg"),
                             lm_maxiter=int(stac_cfg.get("lm_maxiter", 20) or 20))
        self.stac_core_obj = StacCore(self.engine, float(cfg.model.FTOL), int(cfg.model.N_ITER_Q))
        self._offsets = torch.as_tensor(s.tables.site_pos.copy())
        self.timings = None  # bench.py --mode run: {} collects the phase times of ik_only (see _tick)
        self._t_last = 0.0
        self._timestep = s.tables.timestep

    # -- helpers ------------------------------------------------------------------------------------
    def _log(self, *a):
        if self.verbose:
            print(*a, flush=True)

    def _model_handle(self):
        return ModelHandle(engine=self.engine, nq=self.setup.tables.nq, jnt_type=self.setup.tables.jnt_type,
                           site_pos=self.engine.get_site_pos())

    def _get_error_stats(self, errors):
        e = np.asarray(errors).reshape(-1)
        if e.size == 0:  # a rank of a sharded run that owns no clip
            return e, 0.0, 0.0
        return e, float(np.mean(e)), float(np.std(e))

    def _q_phase(self, kp, *, do_root_opt, q_init=None, want_outputs=True):
        s = self.setup
        return self.engine.q_phase(
            kp, part_masks=s.part_masks, trunk_kps=s.trunk_kps, root_kp_idx=max(s.root_kp_idx, 0),
            root_dims=s.root_dims, do_root_opt=do_root_opt, q_init=q_init, want_bodies=want_outputs,
            want_markers=want_outputs)

    # -- fit_offsets (stac.py:253-354) ------------------------------------------------------------------
    def fit_offsets(self, kp_data, time_indices=None) -> StacData:
        """Alternate pose and offset optimisation.

        Default (the reference, ``stac.py:298-341``): ONE warm-started chain over all frames, carried across the
        calibration iterations -- a serial computation, run on this rank's GPU (replicas only).

        ``stac.fit_frames_per_clip: F`` (engine extension, not the reference's sequencing): the fit frames are cut
        into clips of F frames; every clip is its own chain (root-optimised in the first pass, carried across the
        iterations), the clips are sharded over the ranks and the offset phase all-reduces its 3K + 2 partial sums.

        ``time_indices`` (optional) overrides the PRNGKey(0) frame sample of the offset phase.
        """
        cfgm = self.cfg.model
        eng = self.engine
        fpc = int(self.cfg.stac.get("fit_frames_per_clip", 0) or 0)
        kp_np = np.asarray(kp_data, dtype=np.float32)
        if fpc > 0:
            kp_np = utils.batch_kp_data(kp_np, fpc, continuous=False)  # [C, F, 3K]; a ragged tail is dropped
            if kp_np.shape[0] == 0:
                raise ValueError(f"fit_frames_per_clip = {fpc} exceeds the {len(kp_data)} fit frames")
        else:
            kp_np = kp_np[None]
        n_clips, n_per = kp_np.shape[0], kp_np.shape[1]
        n = n_clips * n_per
        lo, hi = dist.shard_range(n_clips) if fpc > 0 else (0, 1)
        kp = torch.as_tensor(kp_np[lo:hi]).to(eng.device)
        self._offsets = eng.get_site_pos().clone()
        do_root = self.setup.do_root_opt
        if self._root_kp_idx == -1:
            self._log("ROOT_OPTIMIZATION_KEYPOINT not specified, skipping Root Optimization.")
        elif self._fixed:
            self._log("ROOT_OPTIMIZATION_KEYPOINT specified but model has fixed root, skipping Root Optimization")
        n_sample = int(cfgm.N_SAMPLE_FRAMES)
        if time_indices is None:
            time_indices = prng.sample_time_indices(n, n_sample, seed=0)
        tix = np.asarray(time_indices, dtype=np.int64)
        mine = tix[(tix >= lo * n_per) & (tix < hi * n_per)] - lo * n_per  # sampled frames that live on this rank
        idx = torch.as_tensor(mine, dtype=torch.long, device=eng.device)
        is_reg = torch.as_tensor(self._is_regularized).to(eng.device)
        carry = None
        res = None
        for n_iter in range(int(cfgm.N_ITERS) + 1):
            final = n_iter == int(cfgm.N_ITERS)
            self._log("Final pose optimization" if final else f"Calibration iteration: {n_iter + 1}/{cfgm.N_ITERS}")
            # root optimisation happens once, before the first pose pass (stac.py:277-296); the warm start is
            # carried across iterations and into the final pass (stac.py:300-301,331-332)
            res = self._q_phase(kp, do_root_opt=(do_root and n_iter == 0), q_init=carry, want_outputs=final)
            carry = res["carry_qpos"]
            _, mean, std = self._get_error_stats(res["frame_error"].cpu().numpy())
            self._log(f"Mean: {mean}\nStandard deviation: {std}")
            if final:
                break
            # offset phase (compute_stac.py:107-167): regularised toward the PREVIOUS iterate (stac.py:317-328)
            nq = self.setup.tables.nq
            partial = eng.m_partial(kp.reshape(-1, kp.shape[-1])[idx], res["qpos"].reshape(-1, nq)[idx])
            if fpc > 0:
                partial = dist.all_reduce_partial(partial)  # the one data-path collective: 3K + 2 floats
            new_off, err = eng.m_finish(partial, self._offsets, is_reg, float(cfgm.M_REG_COEF))
            self._log(f"Final residual error of {float(err)}")
            eng.set_site_pos(new_off)
            self._offsets = new_off
        if fpc > 0 and dist.is_dist():
            res, kp_np = self._gather(res, kp_np, n_clips, lo, hi)
        return self._package_data(res, kp_np.reshape(kp_np.shape[0] * n_per, kp_np.shape[-1]), batched=fpc > 0)

    # -- ik_only (stac.py:356-454) --------------------------------------------------------------------------
    def ik_only(self, kp_data, offsets, gather=None) -> StacData:
        """Inverse kinematics with fixed offsets; clips are independent chains, sharded over ranks.  ``gather`` (multi-GPU
        result placement: "rank0" | "all" | "none") overrides ``stac.gather`` for this call (run_stac resolves "auto")."""
        eng = self.engine
        tick = self._tick
        tick(None)
        batched = utils.batch_kp_data(np.asarray(kp_data, dtype=np.float32), int(self.cfg.stac.n_frames_per_clip),
                                      continuous=bool(self.cfg.stac.continuous))
        eng.set_site_pos(torch.as_tensor(np.asarray(offsets, dtype=np.float32)).reshape(-1, 3))
        n_clips = batched.shape[0]
        lo, hi = dist.shard_range(n_clips)
        kp = torch.as_tensor(batched[lo:hi]).to(eng.device)
        tick("batch_and_h2d_s")
        if self._root_kp_idx == -1:
            self._log("Missing or invalid ROOT_OPTIMIZATION_KEYPOINT, skipping root_optimization()")
        res = self._q_phase(kp, do_root_opt=self.setup.do_root_opt)
        tick("q_phase_and_fk_kernels_s")
        if dist.is_dist():
            res, batched = self._gather(res, batched, n_clips, lo, hi, mode=gather)
            tick("gather_s")
        _, mean, std = self._get_error_stats(res["frame_error"].cpu().numpy())
        self._log(f"Mean: {mean}\nStandard deviation: {std}")
        self._offsets = eng.get_site_pos()
        data = self._package_data(res, batched, batched=True)
        tick("d2h_and_packing_s")
        return data

    def _tick(self, name):
        """Phase clock of ik_only for `bench.py --mode run` (``self.timings = {}`` switches it on; it synchronises the device
        at every phase boundary, so it is off by default)."""
        if self.timings is None:
            return
        import time

        torch.cuda.synchronize(self.engine.device)
        now = time.perf_counter()
        if name is not None:
            self.timings[name] = self.timings.get(name, 0.0) + now - self._t_last
        self._t_last = now

    def _gather(self, res, kp_clips, n_clips, lo, hi, mode=None):
        """Multi-GPU result placement, ``stac.gather`` (engine extension): "rank0" (default) -- rank 0 packages every
        clip, the other ranks keep (and return) their own shard; "all" -- every rank gets every clip (small runs,
        tests); "none" -- every rank keeps its shard.  Returns (results, the keypoint clips that go with them)."""
        mode = str(mode or self.cfg.stac.get("gather", "rank0") or "rank0")
        if mode == "auto":  # run_stac resolves "auto" by output size before it calls ik_only; direct callers get rank0
            mode = "rank0"
        if mode not in ("rank0", "all", "none"):
            raise ValueError(f"stac.gather must be auto, rank0, all or none, not {mode!r}")
        tensors = {k: v for k, v in res.items() if isinstance(v, torch.Tensor)}
        if mode == "all":
            return {k: dist.all_gather_clips(v, n_clips) for k, v in tensors.items()}, kp_clips
        if mode == "rank0":
            full = {k: dist.gather_clips(v, n_clips, dst=0) for k, v in tensors.items()}
            if dist.world()[0] == 0:
                return full, kp_clips
        return tensors, kp_clips[lo:hi]

    # -- packing (stac.py:456-503) ---------------------------------------------------------------------------
    def _package_data(self, res, kp_data, batched=False) -> StacData:
        """Clip-major flatten of every field.

        The reference flattens ``marker_sites`` frame-major when C > 1 and F > 1 (``stac.py:486``: a C-order
        reshape of the (F, C, K, 3) stack, while qpos / xpos / xquat / kp_data come out clip-major -- SURVEY.md
        A5-7, a reference bug).  Default here: all fields clip-major, row i of every field is the same frame.
        ``stac.reference_marker_order: true`` reproduces the reference's row order of ``marker_sites`` exactly
        (row j = frame j // C of clip j % C) for consumers that undo it themselves."""
        nq, nb, K = self.setup.tables.nq, self.setup.tables.nbody, self.setup.tables.nsite
        qpos = res["qpos"].reshape(-1, nq).cpu().numpy()
        xpos = res["xpos"].reshape(-1, nb, 3).cpu().numpy()
        xquat = res["xquat"].reshape(-1, nb, 4).cpu().numpy()
        ms = res["marker_sites"]
        if batched and ms.dim() == 4 and bool(self.cfg.stac.get("reference_marker_order", False)):
            ms = ms.transpose(0, 1)  # (C, F, K, 3) -> (F, C, K, 3), flattened in C order like stac.py:486
        markers = ms.reshape(-1, K, 3).cpu().numpy()
        offsets = np.asarray(torch.as_tensor(self._offsets).cpu()).reshape(K, 3)
        kp_flat = np.asarray(kp_data).reshape(-1, np.asarray(kp_data).shape[-1])
        return StacData(qpos=qpos, xpos=xpos, xquat=xquat, marker_sites=markers, offsets=offsets,
                        names_qpos=self._part_names, names_xpos=self._body_names, kp_data=kp_flat,
                        kp_names=self._kp_names)
