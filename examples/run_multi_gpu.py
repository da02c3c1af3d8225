#!/usr/bin/env python3
"""The whole pipeline on N GPUs of one node, one process per GPU, collectives over RCCL (backend "nccl" on ROCm).

    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port 29511 \
        examples/run_multi_gpu.py [--fit-frames 1000] [--fit-frames-per-clip 10] [--ik-frames 100000] [--out DIR]

What runs (BASELINE.json configs[3], scaled by the arguments):

  1. ``run_stac`` with ``stac.fit_frames_per_clip`` (engine extension): the fit frames are cut into clips, the clips
     are sharded over the ranks, and every calibration iteration combines the 3K + 2 offset-phase sums
     (``stac_mjx/stac_core.py:157-160`` summed over ranks) with ONE collective on the device buffers
     (``dist.all_reduce_partial``: all-gather + fixed-order sum, bitwise reproducible).  That is the only data-path
     collective of the engine.
  2. ``ik_only`` of a long synthetic recording: contiguous blocks of clips per rank, no communication; the results
     are gathered to rank 0 (``stac.gather = rank0``), which writes the output file; ``--gather none`` (or ``auto`` above
     1 GiB of outputs) makes every rank write its own shard file plus a manifest instead.

The process group is created with ``device_id`` right after ``torch.cuda.set_device`` and before any kernel, copy or
allocation on the GPU.  Data: synthetic motion
(stac_mjx_amd/synth.py; no dataset can be fetched here), 1 mm keypoint noise, marker offsets perturbed by 2 mm.
"""

import argparse
import json
import os
import sys
import tempfile
import time
from pathlib import Path

import numpy as np
import torch
import torch.distributed as tdist

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--fit-frames", type=int, default=1000)
    ap.add_argument("--fit-frames-per-clip", type=int, default=10)
    ap.add_argument("--ik-frames", type=int, default=100000)
    ap.add_argument("--frames-per-clip", type=int, default=250)
    ap.add_argument("--n-iters", type=int, default=6)
    ap.add_argument("--out", default=None)
    ap.add_argument("--gather", default="rank0", choices=["rank0", "all", "none", "auto"])
    ap.add_argument("--backend", default="nccl", help="nccl = RCCL over xGMI (GPUs); gloo for a CPU-side smoke run of the plumbing")
    args = ap.parse_args()

    rank, local_rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("LOCAL_RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29511")
    torch.cuda.set_device(local_rank)
    if args.backend == "nccl":
        tdist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(f"cuda:{local_rank}"))
    else:
        tdist.init_process_group(args.backend, rank=rank, world_size=world)

    from stac_mjx_amd import dist, io
    from stac_mjx_amd.config import validate_config
    from stac_mjx_amd.fit_model import finish_fit_setup
    from stac_mjx_amd.main import run_stac
    from stac_mjx_amd.mjcf import ModelTables
    from stac_mjx_amd.synth import synth_keypoints

    g = ROOT / "tests" / "golden"
    mcfg = json.load(open(g / "rodent_model_cfg.json"))
    mcfg["N_ITERS"] = args.n_iters
    kp_names = list(mcfg["KEYPOINT_MODEL_PAIRS"].keys())
    fs = finish_fit_setup(ModelTables.load(g / "rodent_tables.npz"), mcfg, kp_names)
    out_dir = Path(args.out) if args.out else Path(tempfile.gettempdir()) / "stac_multi_gpu"
    if rank == 0:
        out_dir.mkdir(parents=True, exist_ok=True)
    dist.barrier()

    # ---- data: one synthetic recording (seeded: the same on every rank), generated with perturbed marker offsets that
    #      the calibration has to find; run_stac fits the first n_fit_frames and then tracks everything ------------------
    from stac_mjx_amd.engine import Engine
    from stac_mjx_amd.synth import synth_offsets

    gen = Engine(fs.tables, fs.lb, fs.ub, device=f"cuda:{local_rank}")
    gen.set_site_pos(synth_offsets(fs))
    fk = lambda q: gen.fk(q, want=("site_xpos",))["site_xpos"].cpu().numpy()
    C = (args.fit_frames + args.ik_frames + args.frames_per_clip - 1) // args.frames_per_clip
    kp_all, _ = synth_keypoints(fs, fk, C, args.frames_per_clip, seed=7, noise_seed=8)
    kp_all = kp_all.reshape(-1, kp_all.shape[-1])
    gen.close()

    cfg = validate_config({"model": dict(mcfg), "stac": dict(
        fit_offsets_path="fit_offsets.h5", ik_only_path="ik_only.h5", data_path="-", continuous=False,
        n_fit_frames=args.fit_frames, skip_fit_offsets=False, skip_ik_only=False, infer_qvels=False,
        n_frames_per_clip=args.frames_per_clip, fit_frames_per_clip=args.fit_frames_per_clip, gather=args.gather,
        mujoco=dict(solver="newton", iterations=1, ls_iterations=4))})
    torch.cuda.synchronize()
    dist.barrier()
    t0 = time.perf_counter()
    fit_path, ik_path = run_stac(cfg, kp_all, kp_names, base_path=out_dir, setup=fs, device=f"cuda:{local_rank}")
    torch.cuda.synchronize()
    dist.barrier()
    t_all = time.perf_counter() - t0
    if rank == 0:
        _, fit = io.load_stac_data(fit_path)
        _, ik = io.load_sharded_stac_data(ik_path) if str(ik_path).endswith(".manifest.json") else io.load_stac_data(ik_path)
        err = np.linalg.norm(ik.marker_sites - ik.kp_data.reshape(len(ik.kp_data), -1, 3), axis=-1).mean() * 1e3
        print(json.dumps({
            "world_size": world, "backend": tdist.get_backend(), "fit_frames": int(fit.qpos.shape[0]),
            "fit_frames_per_clip": args.fit_frames_per_clip, "ik_frames": int(ik.qpos.shape[0]),
            "n_frames_per_clip": args.frames_per_clip, "seconds_total": t_all,
            "offset_phase_collective": f"{args.n_iters} x all-reduce of {3 * fs.tables.nsite + 2} floats",
            "ik_mean_marker_error_mm": float(err), "fit_file": str(fit_path), "ik_file": str(ik_path)}))
    tdist.destroy_process_group()


if __name__ == "__main__":
    main()
