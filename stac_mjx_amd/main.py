"""User-level pipeline with the reference's signatures (``stac_mjx/main.py``)."""

from __future__ import annotations

import time
from pathlib import Path

import numpy as np

from . import dist, io, utils
from .config import compose_config
from .stac import Stac


def load_configs(config_dir, config_name: str = "config"):
    """``stac_mjx.main.load_configs`` (main.py:18-30)."""
    cfg = compose_config(config_dir, config_name=config_name)
    print("Config loaded and validated.")
    return cfg


def run_stac(cfg, kp_data, kp_names, base_path=None, *, setup=None, device=None):
    """``stac_mjx.main.run_stac`` (main.py:33-139): fit_offsets (unless skipped) then ik_only (unless skipped).

    Returns ``(fit_offsets_path, ik_only_path or None)``.  Raises ``ValueError`` when ``kp_data`` columns
    do not match ``3 * len(kp_names)`` or ``n_frames_per_clip`` does not divide the frame count.
    ``infer_qvels`` runs the reference's finite-difference post-processing on the host.
    """
    base_path = Path.cwd() if base_path is None else Path(base_path)
    kp_data = np.asarray(kp_data)
    expected_cols = len(kp_names) * 3
    if kp_data.shape[1] != expected_cols:
        raise ValueError(
            f"kp_data has {kp_data.shape[1]} columns but expected {expected_cols} ({len(kp_names)} keypoints x 3). "
            "Ensure kp_data is shaped (n_frames, n_keypoints * 3) and that kp_names length matches the number of "
            "keypoints in kp_data.")
    start = time.time()
    fit_offsets_path = base_path / cfg.stac.fit_offsets_path
    ik_only_path = base_path / cfg.stac.ik_only_path
    xml_path = base_path / cfg.model.MJCF_PATH
    stac = Stac(xml_path, cfg, kp_names, setup=setup, device=device)
    if not cfg.stac.skip_ik_only and dist.world()[1] > 1 and str(cfg.stac.get("gather", "auto") or "auto") == "none":
        # an EXPLICIT gather = none cannot serve a continuous run (cross-fades reach across shard borders): say so before any
        # work is done -- the config that decides is the one stored with the fit (the caller's, when this run writes it)
        fit_cfg = cfg if not cfg.stac.skip_fit_offsets else io.load_stac_data(fit_offsets_path)[0]
        if fit_cfg.stac.continuous:
            raise ValueError("stac.gather = none writes per-rank shards, but stac.continuous cross-fades neighbouring "
                             "clips across shard borders: use gather = rank0 (or all, or auto) for continuous runs")

    if not cfg.stac.skip_fit_offsets:
        kps = kp_data[: cfg.stac.n_fit_frames]
        print(f"Running fit. Mocap data shape: {kps.shape}")
        fit_data = stac.fit_offsets(kps)
        if dist.world()[0] == 0:  # multi-GPU: rank 0 holds the gathered result and writes it
            io.save_data_to_h5(config=cfg, file_path=fit_offsets_path, **fit_data.as_dict())
        fit_offsets_path = io.resolve_output_path(fit_offsets_path)
        dist.barrier()
        print(f"saved fit to {fit_offsets_path}", flush=True)
    else:
        print("Skipping fit_offsets. To change this behavior, set cfg.stac.skip_fit_offsets to False.")

    if cfg.stac.skip_ik_only:
        print("Skipping IK-only phase. To change this behavior, set cfg.stac.skip_ik_only to False.")
        return fit_offsets_path, None
    if kp_data.shape[0] % cfg.stac.n_frames_per_clip != 0:
        raise ValueError(
            f"n_frames_per_clip ({cfg.stac.n_frames_per_clip}) must divide evenly with the total number of mocap "
            f"frames({kp_data.shape[0]})")
    print("Running ik_only()")
    # the config stored with the fit replaces the caller's from here on, as in the reference (main.py:111): it decides
    # continuous / n_frames_per_clip / infer_qvels below and is what the ik_only file records (the Stac object
    # itself keeps the caller's config, also as in the reference)
    cfg, fit_data = io.load_stac_data(fit_offsets_path)
    rank, world_size = dist.world()
    F = int(cfg.stac.n_frames_per_clip)
    # where the results go: resolved here (an "auto" becomes rank0 / none by output size; a continuous run always gathers) and
    # handed to ik_only as an argument -- the caller's config object is not touched
    mode = _ik_output_mode(stac.cfg, kp_data.shape[0], stac, continuous=bool(cfg.stac.continuous)) if world_size > 1 else "rank0"
    sharded = world_size > 1 and mode == "none"
    if sharded and cfg.stac.continuous:  # (explicit gather = none with a fit file whose config turned out continuous)
        raise ValueError("stac.gather = none writes per-rank shards, but stac.continuous cross-fades neighbouring "
                         "clips across shard borders: use gather = rank0 (or all, or auto) for continuous runs")
    ik_data = stac.ik_only(kp_data, fit_data.offsets, gather=mode)
    if rank != 0 and not sharded and mode != "all":
        dist.barrier()  # this rank holds its own shard only: rank 0 post-processes and writes the gathered result
        return fit_offsets_path, io.resolve_output_path(ik_only_path)
    if cfg.stac.continuous:
        ik_data = utils.handle_edge_effects(ik_data, F)
    print(f"Final qpos shape: {ik_data.qpos.shape}")
    if cfg.stac.infer_qvels and ik_data.qpos.shape[0]:  # main.py:118-133: per clip of n_frames_per_clip frames
        batched = ik_data.qpos.reshape((-1, F, ik_data.qpos.shape[-1]))
        qvels = [utils.compute_velocity_from_kinematics(c, dt=stac._timestep, freejoint=stac._freejoint) for c in batched]
        ik_data.qvel = np.stack(qvels).reshape(-1, qvels[0].shape[-1])
    if sharded:
        # every rank writes ITS clips under a shard name (never a partial result under the full-run name); rank 0 adds
        # the manifest that io.load_sharded_stac_data() reads the run back through
        n_clips = kp_data.shape[0] // F
        lo, hi = dist.shard_range(n_clips)
        shard = io.shard_path(ik_only_path, rank, world_size)
        io.save_data_to_h5(config=cfg, file_path=shard, **ik_data.as_dict())
        dist.barrier()
        manifest = io.manifest_path(ik_only_path)
        if rank == 0:
            io.write_manifest(manifest, ik_only_path, world_size, n_clips, F)
        dist.barrier()
        print(f"Saved ik_only shard {io.resolve_output_path(shard)} (clips {lo}:{hi}); manifest {manifest}. "
              f"Finished in {(time.time() - start) / 60:.2f} minutes")
        return fit_offsets_path, manifest
    if rank == 0:
        io.save_data_to_h5(config=cfg, file_path=ik_only_path, **ik_data.as_dict())
    ik_only_path = io.resolve_output_path(ik_only_path)
    dist.barrier()
    print(f"Saved ik_only to {ik_only_path}. Finished in {(time.time() - start) / 60:.2f} minutes")
    return fit_offsets_path, ik_only_path


GATHER_AUTO_MAX_BYTES = 1 << 30  # above this much output a multi-GPU run keeps per-rank shard files ("auto")


def _ik_output_mode(cfg, n_frames: int, stac, continuous: bool = False) -> str:
    """``stac.gather`` for the ik_only outputs: "rank0" | "all" | "none" | "auto" (default).  "auto" gathers to rank 0
    while the packed outputs (qpos, xpos, xquat, marker_sites, kp_data; 2 728 B per rodent frame) stay below
    ``stac.gather_max_bytes`` (default 1 GiB) and writes per-rank shards above: a 1 M-frame run is 2.7 GB that a
    padded gather would stage on rank 0's GPU for nothing.  A continuous run (cross-fades across clip borders, done on the
    gathered result) always gathers under "auto", whatever its size."""
    mode = str(cfg.stac.get("gather", "auto") or "auto")
    if mode != "auto":
        return mode
    if continuous:
        return "rank0"
    t = stac.setup.tables
    per_frame = 4 * (t.nq + 7 * t.nbody + 6 * t.nsite)
    limit = int(cfg.stac.get("gather_max_bytes", GATHER_AUTO_MAX_BYTES) or GATHER_AUTO_MAX_BYTES)
    return "rank0" if n_frames * per_frame <= limit else "none"
