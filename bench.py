#!/usr/bin/env python3
"""bench.py -- frames/s of the STAC q_phase on MI355X (BASELINE.json metric), one JSON line on rank 0.

A "step" is one pass of the hot path (`stac_q_phase`: root optimisation + full-body solve + part
solves per frame, PG parity mode, FTOL/N_ITER_Q of configs/model/rodent.yaml) over one batch of
synthetic keypoints that is already resident in HBM.  Workload = BASELINE.json configs[1]: rodent,
23 keypoints, 10 000 synthetic frames, q_phase only, on every GPU (weak scaling: clips are
independent, each rank fits its own shard; no collective on the data path).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--frames 10000] [--frames-per-clip 1]

For N > 1 either launch it under torch.distributed.run yourself (one rank per GPU) or just pass --gpus N: without
WORLD_SIZE in the environment the script starts `python -m torch.distributed.run --nproc-per-node N bench.py ...` as a
child process BEFORE anything touches the GPU and relays rank 0's JSON line.  RCCL carries the timing barrier and the
max-over-ranks reduction (and, in --mode fit, the 71-float offset-phase all-reduce).
"""

from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md chip table: 8.0 TB/s spec (6.29 TB/s measured copy)
FP32_VALU_PEAK_TFLOPS = 157.3
VALU_ISSUE_CEILING = 0.429  # wave64 VALU instructions per cycle per SIMD, measured (profiles/tools/valu_issue_micro.hip)


def load_setup(model="rodent"):
    from stac_mjx_amd.fit_model import finish_fit_setup
    from stac_mjx_amd.mjcf import ModelTables

    g = ROOT / "tests" / "golden"
    cfg = json.load(open(g / f"{model}_model_cfg.json"))
    fs = finish_fit_setup(ModelTables.load(g / f"{model}_tables.npz"), cfg, list(cfg["KEYPOINT_MODEL_PAIRS"].keys()))
    return fs, cfg


def job_shape(scaling: str, frames: int, frames_per_clip: int, rank: int, world: int) -> dict:
    """Which clips a rank of the job solves.  weak: every rank brings `frames` frames (the default bench: per-GPU work fixed);
    strong: `frames` frames in the WHOLE job (BASELINE configs[3]: 1 M frames at 250 per clip = 4 000 clips over the ranks).
    Clips go to ranks in contiguous blocks (dist.shard_range, what Stac.ik_only does: stac_mjx/stac.py:405-440).
    `value` of a line = frames_total x steps / max-over-ranks time (job_value)."""
    from stac_mjx_amd.dist import shard_range

    F = int(frames_per_clip)
    c_total = frames // F if scaling == "strong" else world * (frames // F)
    lo, hi = shard_range(c_total, rank, world)
    return {"scaling": scaling, "F": F, "clips_total": c_total, "lo": lo, "hi": hi, "clips_rank": hi - lo,
            "frames_rank": (hi - lo) * F, "frames_total": c_total * F}


def job_value(shape: dict, steps: int, elapsed_max_over_ranks: float) -> float:
    return shape["frames_total"] * steps / elapsed_max_over_ranks


def rank_record(rank: int, shape: dict, kernel, kernel_ms: float, elapsed_s: float) -> dict:
    """What one rank did in the timed region: its clips, the q_phase instantiation its LAST launch ran (<G, NQR, WPE, SPECP> from
    stac_debug_last_q_kernel: a rank whose share falls into the latency regime runs another kernel than a full GPU does), the
    mean HIP-event time of a step on its stream and its own wall time (the line's `value` only carries the slowest rank's)."""
    return {"rank": int(rank), "clips": int(shape["clips_rank"]), "frames": int(shape["frames_rank"]),
            "kernel": "q_phase_kernel<%d,%d,%d,%d>" % tuple(int(x) for x in kernel), "kernel_ms": float(kernel_ms),
            "elapsed_s": float(elapsed_s)}


def gather_rank_records(rec: dict, dist) -> list:
    """Every rank's record on every rank, in rank order (one all_gather_object: control plane, after the timed region)."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return [rec]
    out = [None] * dist.get_world_size()
    dist.all_gather_object(out, rec)
    return sorted(out, key=lambda r: r["rank"])


def predicted_value(model: str, shape: dict, world: int, table_path=None):
    """The whole-job rate this line SHOULD show if every GPU ran its share at the committed single-GPU rate of that share
    (profiles/single_gpu_rates.json: measured on one MI355X at the per-rank chain counts; log-linear between its rows).  The
    slowest rank has the most clips (dist.shard_range: the first one).  None when the table has no rows for this kind of job."""
    import math

    try:
        rows = [r for r in json.load(open(table_path or ROOT / "profiles" / "single_gpu_rates.json"))["rows"]
                if r["model"] == model and r["frames_per_clip"] == shape["F"]]
    except (FileNotFoundError, KeyError, ValueError):
        return None
    rows.sort(key=lambda r: r["chains"])
    clips = -(-shape["clips_total"] // world)  # the largest share
    if not rows or clips <= 0:
        return None
    if clips <= rows[0]["chains"]:
        rate = rows[0]["frames_per_s"] * clips / rows[0]["chains"]  # below the table: a GPU this empty runs chains side by side
    elif clips >= rows[-1]["chains"]:
        rate = rows[-1]["frames_per_s"]
    else:
        hi = next(i for i, r in enumerate(rows) if r["chains"] >= clips)
        a, b = rows[hi - 1], rows[hi]
        w = (math.log(clips) - math.log(a["chains"])) / (math.log(b["chains"]) - math.log(a["chains"]))
        rate = math.exp((1 - w) * math.log(a["frames_per_s"]) + w * math.log(b["frames_per_s"]))
    return shape["frames_total"] / (clips * shape["F"] / rate)  # job frames over the slowest rank's predicted time


def cpu_baseline(fs, cfg, kp_host, target_s=15.0):
    """The oracle (CPU restatement, kind='port') on a bounded sample of the same workload."""
    from oracle import Oracle

    orc = Oracle(fs.tables, tol=float(cfg["FTOL"]), maxiter=int(cfg["N_ITER_Q"]))
    cores = orc.max_threads()
    C, F = kp_host.shape[0], kp_host.shape[1]

    def run(n):
        t0 = time.perf_counter()
        orc.ik_clips(kp_host[:n], fs.lb, fs.ub, fs.part_masks, fs.trunk_kps, max(fs.root_kp_idx, 0), fs.root_dims,
                     do_root_opt=fs.do_root_opt, want_bodies=False)
        return time.perf_counter() - t0

    n0 = min(C, max(cores * 2, 1))
    t0 = run(n0)
    n = int(min(C, max(n0, n0 * target_s / max(t0, 1e-3))))
    t = run(n)
    return {"value": n * F / t, "unit": "frames/s", "cores": cores, "kind": "port",
            "sample": f"oracle/stac_oracle.c (f32 PG restatement; gcc -O3 -mfma -ffp-contract=off, OpenMP schedule(dynamic) over "
                      f"clips, {cores} threads) on the first {n} clips x {F} frames of the same synthetic batch, {t:.1f} s"}


def run_fit_mode(args, rank, local_rank, world, dist):
    """--mode fit: Stac.fit_offsets with stac.fit_frames_per_clip (engine extension): the fit frames are cut into clips,
    the clips are sharded over the ranks (one process per GPU) and every calibration iteration all-reduces the 3K + 2
    offset-phase sums over RCCL -- the one data-path collective of the engine.  Weak scaling: every rank
    brings `--frames` frames.  A step = one whole fit (N_ITERS pose passes + offset phases, then the final pose pass)."""
    from stac_mjx_amd.config import validate_config
    from stac_mjx_amd.engine import Engine
    from stac_mjx_amd.stac import Stac
    from stac_mjx_amd.synth import synth_keypoints, synth_offsets

    fs, mcfg = load_setup("rodent")
    F = args.frames_per_clip if args.frames_per_clip > 1 else 10
    C = args.frames // F
    cfg = validate_config({"model": dict(mcfg), "stac": dict(
        fit_offsets_path="fit.h5", ik_only_path="ik.h5", data_path="synthetic", continuous=False, n_fit_frames=world * C * F,
        skip_fit_offsets=False, skip_ik_only=True, infer_qvels=False, n_frames_per_clip=F, fit_frames_per_clip=F, gather="none",
        mujoco=dict(solver="newton", iterations=1, ls_iterations=4))})
    stac = Stac(None, cfg, fs.kp_names, setup=fs, device=f"cuda:{local_rank}", verbose=False)
    eng = stac.engine
    gen = Engine(fs.tables, fs.lb, fs.ub, device=f"cuda:{local_rank}")
    gen.set_site_pos(synth_offsets(fs))  # keypoints come from perturbed offsets: the fit has something to find
    fk = lambda q: gen.fk(q, want=("site_xpos",))["site_xpos"].cpu().numpy()
    kp_host, _ = synth_keypoints(fs, fk, world * C, F, seed=3, noise_seed=4)  # the same global batch on every rank
    kp = kp_host.reshape(world * C * F, -1)
    n_pass = int(mcfg["N_ITERS"]) + 1
    off0 = fs.tables.site_pos.copy()

    def step():
        eng.set_site_pos(off0)
        return stac.fit_offsets(kp)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        data = step()
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if dist:
        tmax = torch.tensor([elapsed], device=eng.device)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    solves = world * C * F * n_pass * args.steps
    d_off = float(np.linalg.norm(data.offsets - synth_offsets(fs), axis=-1).mean() * 1e3)
    line = {
        "metric": "frame-solves/sec STAC fit_offsets (rodent, 23 kp), " + ("one chain" if C == 1 else "clip-parallel calibration"),
        "value": solves / elapsed, "unit": "frame-solves/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {
            "workload": f"Stac.fit_offsets, stac.fit_frames_per_clip={F}: {C * F} frames per GPU as {C} clips, N_ITERS={n_pass - 1} "
                        f"(+ final pass), PG parity mode; "
                        + ("one warm-started chain per GPU = the reference's own sequencing (stac.py:298-311)" if C == 1
                           else "NOT the reference's single-chain sequencing (engine extension)"),
            "frames_per_gpu": C * F, "fit_frames_per_clip": F,
            "parallelism": f"clips sharded over {world} GPU(s); offset phase: one all-reduce of {3 * fs.tables.nsite + 2} floats per "
                           f"calibration iteration ({n_pass - 1} per fit)",
            "collective_backend": (dist.get_backend() if dist else None), "collective_world_size": world,
            "mean_offset_error_mm_vs_generating_offsets": d_off,
        },
    }
    if rank == 0:
        print(json.dumps(line))


def run_run_mode(args, rank, local_rank, world, dist):
    """--mode run: the PRODUCT path end to end -- `Stac.ik_only` (stac_mjx/stac.py:356-454) on synthetic frames with every
    output (qpos, xpos, xquat, marker_sites; 2 728 B per rodent frame), then `io.save_data_to_h5`
    (stac_mjx/io.py:194-237).  Reports where the wall time goes: batching + upload, the q_phase and FK kernels, download +
    packing, the file write; and the FK output pass on its own (HIP events) against the HBM roofline of its 2 448 + 296
    algorithmic bytes per pose.  `value` = frames/s of ik_only itself (file write excluded and reported beside it)."""
    import tempfile

    from stac_mjx_amd import io as sio
    from stac_mjx_amd.config import validate_config
    from stac_mjx_amd.stac import Stac
    from stac_mjx_amd.synth import synth_keypoints, synth_offsets

    fs, mcfg = load_setup("rodent")
    F = args.frames_per_clip if args.frames_per_clip > 1 else 250
    C = max(args.frames // F, 1)
    cfg = validate_config({"model": dict(mcfg), "stac": dict(
        fit_offsets_path="fit.h5", ik_only_path="ik.h5", data_path="synthetic", continuous=False, n_fit_frames=F,
        skip_fit_offsets=True, skip_ik_only=False, infer_qvels=False, n_frames_per_clip=F, gather="none",
        mujoco=dict(solver="newton", iterations=1, ls_iterations=4))})
    stac = Stac(None, cfg, fs.kp_names, setup=fs, device=f"cuda:{local_rank}", verbose=False)
    eng = stac.engine
    offsets = synth_offsets(fs)
    eng.set_site_pos(offsets)
    fk = lambda q: eng.fk(q, want=("site_xpos",))["site_xpos"].cpu().numpy()
    # ONE global dataset of world x C clips, the same on every rank: `Stac.ik_only` shards the clips itself
    # (dist.shard_range), so every rank solves C of them -- weak scaling: `--frames` frames per GPU
    from stac_mjx_amd import dist as sdist

    shape = job_shape("weak", C * F, F, rank, world)
    kp_glob, _ = synth_keypoints(fs, fk, shape["clips_total"], F, seed=11, noise_seed=12)
    kp_flat = kp_glob.reshape(shape["frames_total"], -1)
    lo, hi = shape["lo"], shape["hi"]
    assert (lo, hi) == sdist.shard_range(shape["clips_total"])  # (what ik_only will take)
    kp_host = kp_glob[lo:hi]  # this rank's clips (kernel-only timing below)
    nq, nb, K = fs.tables.nq, fs.tables.nbody, fs.tables.nsite
    for _ in range(args.warmup):
        stac.ik_only(kp_flat, offsets)
    phases, walls, writes = {}, [], []
    data = None
    for _ in range(args.steps):
        stac.timings = {}
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        data = stac.ik_only(kp_flat, offsets)
        torch.cuda.synchronize()
        walls.append(time.perf_counter() - t0)
        for k, v in stac.timings.items():
            phases[k] = phases.get(k, 0.0) + v / args.steps
        stac.timings = None
    with tempfile.TemporaryDirectory() as td:
        t0 = time.perf_counter()
        out = sio.save_data_to_h5(cfg, data.kp_names, data.names_qpos, data.names_xpos, data.kp_data, data.marker_sites, data.offsets,
                                  data.qpos, data.xpos, data.xquat, np.array([]), Path(td) / "ik_only.h5")
        write_s = time.perf_counter() - t0
        written = (out.name, out.stat().st_size)
    # the kernels on their own (HIP events on the launch stream): q_phase without outputs, then the FK output pass over its result
    kp = torch.as_tensor(kp_host).to(eng.device)
    kw = dict(part_masks=fs.part_masks, trunk_kps=fs.trunk_kps, root_kp_idx=max(fs.root_kp_idx, 0), root_dims=fs.root_dims,
              do_root_opt=fs.do_root_opt)
    res = eng.q_phase(kp, want_bodies=False, want_markers=False, want_carry=False, **kw)
    e = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    e[0].record()
    res = eng.q_phase(kp, want_bodies=False, want_markers=False, want_carry=False, out=res, **kw)
    e[1].record()
    qflat = res["qpos"].reshape(-1, nq)
    fko = eng.fk(qflat)
    fk_ms = []
    for _ in range(5):
        e[2].record()
        fko = eng.fk(qflat)
        e[3].record()
        torch.cuda.synchronize()
        fk_ms.append(e[2].elapsed_time(e[3]))
    q_ms, fk_ms = e[0].elapsed_time(e[1]), float(np.median(fk_ms))
    fk_bytes = C * F * (4 * nq * 2 + 12 * nb + 16 * nb + 12 * K)  # qpos in, normalised qpos + xpos + xquat + marker sites out
    wall = float(np.mean(walls))
    if dist:  # the slowest rank's wall is the job's
        tmax = torch.tensor([wall], device=eng.device)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        wall = float(tmax.item())
    frames = (hi - lo) * F  # frames this rank solved (every rank the same number)
    assert data.qpos.shape[0] == frames, (data.qpos.shape, frames)
    gpu_side = phases.get("q_phase_and_fk_kernels_s", 0.0)
    line = {
        "metric": "frames/sec Stac.ik_only end to end (rodent, 23 kp, all outputs)", "value": job_value(shape, 1, wall), "unit": "frames/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * wall, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {
            "workload": f"Stac.ik_only: {frames} synthetic rodent frames per GPU as {C} clips of {F} (reference default chaining), "
                        f"qpos + xpos + xquat + marker_sites = {4 * nq + 28 * nb + 12 * K + 4} B per frame out, PG parity mode",
            "frames_per_gpu": frames, "n_frames_per_clip": F,
            "phases_s": {**{k: round(v, 4) for k, v in phases.items()}, "ik_only_wall_s": round(wall, 4),
                         "save_data_to_h5_s": round(write_s, 3)},
            "file_written": {"name": written[0], "bytes": written[1],
                             "note": "gzip datasets like the reference (io.py:224-236); .npz stand-in where this interpreter has no h5py"},
            "kernels_ms": {"q_phase_no_outputs": round(q_ms, 3), "fk_output_pass": round(fk_ms, 4)},
            "q_phase_share_of_gpu_side_time": q_ms * 1e-3 / gpu_side if gpu_side else None,
            "fk_output_pass": {"algorithmic_bytes": fk_bytes, "achieved_GBps": fk_bytes / (fk_ms * 1e-3) / 1e9,
                               "frac_of_hbm_peak": fk_bytes / (fk_ms * 1e-3) / 1e9 / HBM_PEAK_GBS},
            "d2h_bytes": frames * (4 * nq + 28 * nb + 12 * K + 4),
        },
        "roofline": {"bound": "hbm", "achieved": frames * (12 * K + 4 * nq + 28 * nb + 12 * K + 4) / gpu_side / 1e9 if gpu_side else None,
                     "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": (frames * (12 * K + 4 * nq + 28 * nb + 12 * K + 4) / gpu_side / 1e9 / HBM_PEAK_GBS) if gpu_side else None,
                     "traffic": None, "kernel": "stac::q_phase_kernel + stac::fk_kernel (ik_only, all outputs)",
                     "algorithmic_bytes_per_frame": 12 * K + 4 * nq + 28 * nb + 12 * K + 4},
    }
    if rank == 0:
        print(json.dumps(line))


def self_launch(n_gpus: int) -> int:
    """`python bench.py --gpus N` outside torchrun: start the N ranks as CHILD processes (never re-exec a process that
    may have touched the GPU; the devices are counted from the KFD topology in sysfs, not through torch or HIP) and return
    their exit code.  Rank 0 of the children prints the JSON line on the inherited stdout."""
    import socket
    import subprocess

    from stac_mjx_amd.dist import visible_gpu_count

    visible = visible_gpu_count()  # KFD topology + *_VISIBLE_DEVICES: never torch / HIP in the launcher parent
    if visible < n_gpus:
        print(f"bench.py: --gpus {n_gpus} requested but only {visible} GPU{'s' if visible != 1 else ''} visible "
              f"on this node; nothing launched", file=sys.stderr)
        return 2
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL needs it on this pool
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), str(Path(__file__).resolve()), *sys.argv[1:]]
    return subprocess.run(cmd, env=env).returncode


def lib_digest() -> str:
    """What libstac_hip.so was built from (profiles/*.json entries are tied to it)."""
    from stac_mjx_amd.build import STAMP

    return STAMP.read_text().strip() if STAMP.exists() else ""


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--frames", type=int, default=10000, help="frames per GPU per step (--scaling strong: frames of the WHOLE job)")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak = --frames frames on every GPU (the default bench); strong = --frames frames in total, their clips "
                         "split over the ranks (BASELINE configs[3]: --scaling strong --frames 1000000 --frames-per-clip 250)")
    ap.add_argument("--frames-per-clip", type=int, default=1)
    ap.add_argument("--lanes", type=int, default=0, help="lanes of a wavefront per chain (0 = auto)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the clips250 and LM legs (profiling runs)")
    ap.add_argument("--mode", default="ik", choices=["ik", "fit", "run"],
                    help="ik = q_phase only (the BASELINE metric); fit = offset/pose alternation on clips sharded over the "
                         "ranks with the offset-phase all-reduce over RCCL (stac.fit_frames_per_clip), reported as frame-solves/s; "
                         "run = Stac.ik_only end to end with every output + the result file, phase by phase")
    ap.add_argument("--solver", default="pg", choices=["pg", "lm"],
                    help="pg = the reference's projected gradient (parity mode, the BASELINE metric); lm = optional fast solver")
    ap.add_argument("--lm-maxiter", type=int, default=20, help="--solver lm: accepted LM steps per solve (engine default 20)")
    ap.add_argument("--model", default="rodent", choices=["rodent", "fly", "mouse"],
                    help="rodent = the BASELINE metric; fly / mouse = other fixtures (reported under config, not the headline)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(self_launch(args.gpus))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch one rank per GPU "
                         f"(python -m torch.distributed.run --nproc-per-node {args.gpus} bench.py --gpus {args.gpus} ...)")
    # Test hooks (tests/test_gpu_stac.py: the N > 1 code path of this script on a ONE-GPU box): STAC_BENCH_SHARE_GPU=1 puts every rank on
    # cuda:0, STAC_BENCH_BACKEND=gloo carries the barrier / reductions (RCCL refuses two ranks on one device).  Never set by the driver.
    backend = os.environ.get("STAC_BENCH_BACKEND", "nccl")
    if os.environ.get("STAC_BENCH_SHARE_GPU"):
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1 or "TORCHELASTIC_RUN_ID" in os.environ:  # under torch.distributed.run: RCCL barrier + max-over-ranks timing
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local_rank}"))
        else:
            dist.init_process_group(backend)

    if args.mode in ("fit", "run"):
        (run_fit_mode if args.mode == "fit" else run_run_mode)(args, rank, local_rank, world, dist)
        if dist:
            dist.destroy_process_group()
        return

    from stac_mjx_amd.engine import Engine
    from stac_mjx_amd.synth import synth_keypoints, synth_offsets

    fs, cfg = load_setup(args.model)
    F = args.frames_per_clip
    strong = args.scaling == "strong"
    if F < 1 or args.frames % F:
        raise SystemExit(f"bench.py: --frames {args.frames} is not a multiple of --frames-per-clip {F} (the job would silently be "
                         f"{args.frames // max(F, 1) * max(F, 1)} frames)")
    shape = job_shape(args.scaling, args.frames, F, rank, world)
    C_total, lo, hi, C = shape["clips_total"], shape["lo"], shape["hi"], shape["clips_rank"]
    eng = Engine(fs.tables, fs.lb, fs.ub, tol=float(cfg["FTOL"]), maxiter=int(cfg["N_ITER_Q"]), lanes_per_chain=args.lanes,
                 device=f"cuda:{local_rank}", solver=args.solver, lm_maxiter=args.lm_maxiter)
    # synthetic batch (seeded per rank), generated with the engine's own FK, then offsets fixed
    eng.set_site_pos(synth_offsets(fs))
    fk = lambda q: eng.fk(q, want=("site_xpos",))["site_xpos"].cpu().numpy()
    if strong:
        # one global dataset whatever the number of ranks: blocks of kBlock clips, block b seeded by b; a rank generates the
        # blocks its shard touches (uploaded block by block: 1 M frames are 276 MB of keypoints)
        kBlock = 500
        parts = []
        for b in (range(lo // kBlock, (hi - 1) // kBlock + 1) if hi > lo else ()):  # (a rank without clips -- more ranks than clips -- generates nothing)
            nb_ = min(kBlock, C_total - b * kBlock)
            blk, _ = synth_keypoints(fs, fk, nb_, F, seed=100 + 2 * b, noise_seed=101 + 2 * b)
            parts.append(torch.as_tensor(blk[max(lo - b * kBlock, 0):max(min(hi - b * kBlock, nb_), 0)]).to(eng.device))
        kp = torch.cat(parts) if parts else torch.empty((0, F, 3 * fs.tables.nsite), device=eng.device)
        kp_host = kp[:2000].cpu().numpy()  # (cpu_baseline's sample)
        assert kp.shape[0] == C
    else:
        kp_host, _ = synth_keypoints(fs, fk, C, F, seed=1000 * rank, noise_seed=1000 * rank + 1)
        kp = torch.as_tensor(kp_host).to(eng.device)
    out = None

    def step():
        nonlocal out
        if C == 0:  # (a rank without clips still takes part in the barriers and the reductions)
            return
        out = eng.q_phase(kp, part_masks=fs.part_masks, trunk_kps=fs.trunk_kps, root_kp_idx=fs.root_kp_idx,
                          root_dims=fs.root_dims, do_root_opt=fs.do_root_opt, want_bodies=False, want_markers=False, want_carry=False,
                          out=out)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for i in range(args.steps):
        ev[i][0].record()  # same stream the kernel is launched on (torch's current stream)
        step()
        ev[i][1].record()
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    elapsed_rank = elapsed
    if dist:
        tmax = torch.tensor([elapsed], device=eng.device)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    kern_ms = float(np.mean([a.elapsed_time(b) for a, b in ev])) if C > 0 else 0.0
    import ctypes

    last_kernel = (ctypes.c_int32 * 4)()
    eng.lib.stac_debug_last_q_kernel(last_kernel)
    per_rank = gather_rank_records(rank_record(rank, shape, tuple(last_kernel), kern_ms, elapsed_rank), dist)

    frames_step = C * F  # this rank's frames per step
    value = job_value(shape, args.steps, elapsed)  # whole-job frames over the slowest rank's time
    if C > 0:
        cnt = out["counters"].to(torch.float64).sum(dim=(0, 1)).cpu().numpy()
        err = torch.linalg.norm((eng.fk(out["qpos"].reshape(-1, fs.tables.nq), want=("site_xpos",))["site_xpos"]
                                 - kp.reshape(-1, fs.tables.nsite, 3)), dim=-1)
    else:
        cnt, err = np.zeros(4), torch.zeros(1, device=eng.device)
    # algorithmic bytes of the q_phase kernel: keypoints in, qpos + residual out (576 B/frame)
    bytes_frame = 12 * fs.tables.nsite + 4 * fs.tables.nq + 4
    kern_ms = max(kern_ms, 1e-9)
    achieved = frames_step * bytes_frame / (kern_ms * 1e-3) / 1e9
    # algorithmic flops (full-tree constants): value+grad 17.9 kflop, loss 15.9 kflop
    flops = cnt[2] * 17.9e3 + cnt[1] * 15.9e3
    traffic, traffic_note = None, "no rocprofv3 PMC entry for this workload in profiles/traffic.json"
    try:  # HBM bytes per launch measured offline with rocprofv3 PMC passes on this same workload (profiles/)
        for ent in json.load(open(ROOT / "profiles" / "traffic.json"))["entries"]:
            if ent["frames"] == frames_step and ent["frames_per_clip"] == F and (args.lanes == 0) == (ent["lanes"] == "auto") \
                    and ent.get("model", "rodent") == args.model and ent.get("solver", "pg") == args.solver:
                if ent.get("lib_digest", "") == lib_digest():
                    traffic, traffic_note = ent["bytes_per_launch"], ent.get("source", "")
                else:  # the counters were collected on another build of the kernels: not this launch's traffic
                    traffic_note = f"stale: {ent.get('source', '')} was measured on another build of libstac_hip.so"
    except (FileNotFoundError, KeyError):
        pass
    line = {
        "metric": "frames/sec STAC pose-fit (rodent, 23 kp)" if args.model == "rodent" else f"frames/sec STAC pose-fit ({args.model}, {fs.tables.nsite} kp)",
        "value": value, "unit": "frames/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
        "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {
            "workload": (f"BASELINE configs[3]: rodent.xml, {C_total * F} synthetic frames in total as {C_total} clips split over "
                         f"{world} GPU(s) (strong scaling; this rank: {C} clips), " if strong and args.model == "rodent" else
                         f"{'BASELINE configs[1]: rodent.xml' if args.model == 'rodent' else args.model + ' model'}, ")
                        + f"{fs.tables.nsite} keypoints, {frames_step} synthetic frames per GPU, "
                        f"q_phase only ({'root opt + ' if fs.do_root_opt else ''}full + {len(fs.part_masks)} part PG solves per frame), n_frames_per_clip={F} "
                        f"({C} independent chains), FTOL={cfg['FTOL']}, N_ITER_Q={cfg['N_ITER_Q']}, "
                        + ("solver=pg (parity mode)" if args.solver == "pg" else f"solver=lm, at most {args.lm_maxiter} steps per solve (NOT the reference's algorithm; marker-space quality only)"),
            "frames_per_gpu": frames_step, "frames_total": C_total * F, "n_frames_per_clip": F, "lanes_per_chain": args.lanes or "auto",
            "parallelism": f"clips sharded over {world} GPU(s) (dist.shard_range: contiguous blocks), no data-path collective",
            "collective_backend": (dist.get_backend() if dist else None), "collective_world_size": world,
            # every rank's share and what it ran (a SCALE record can be diagnosed from the line alone), and the rate the committed
            # single-GPU table predicts for this job (profiles/single_gpu_rates.json)
            "per_rank": per_rank, "predicted_value": predicted_value(args.model, shape, world) if args.solver == "pg" else None,
            "iters_per_frame": cnt[0] / max(frames_step, 1), "ls_evals_per_frame": cnt[1] / max(frames_step, 1),
            "grad_evals_per_frame": cnt[2] / max(frames_step, 1),
            "marker_rmse_mm": float(torch.sqrt((err ** 2).mean()).item() * 1e3),
            "valu_tflops_algorithmic": flops / (kern_ms * 1e-3) / 1e12,
            "valu_frac_of_fp32_peak": flops / (kern_ms * 1e-3) / 1e12 / FP32_VALU_PEAK_TFLOPS,
        },
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_note,
                     "kernel": "stac::q_phase_kernel",
                     "algorithmic_bytes_per_launch": frames_step * bytes_frame,
                     "kernel_ms": kern_ms, "algorithmic_bytes_per_frame": bytes_frame},
    }
    # VALU view of the same launch (the binding resource; informative next to the required HBM roofline): issued VALU
    # wave-instructions per step from the committed rocprofv3 PMC pass of this workload (profiles/), against the
    # algorithmic work of the evaluations the kernel's own counters report
    try:
        for ent in json.load(open(ROOT / "profiles" / "valu.json"))["entries"]:
            if ent["frames"] == frames_step and ent["frames_per_clip"] == F and args.model == ent.get("model", "rodent") and args.solver == "pg":
                if ent.get("lib_digest", "") != lib_digest():
                    # the instruction count belongs to another build of the kernels: do not mix it with this run's time
                    line["roofline_valu"] = {"stale": True, "source": ent["source"]}
                    continue
                evals = float(cnt[1] + cnt[2])  # logical q_loss evaluations of the step (line-search + value-and-gradient)
                lane_slots = ent["sq_insts_valu"] * 64.0
                line["roofline_valu"] = {
                    "sq_insts_valu_per_step": ent["sq_insts_valu"], "source": ent["source"],
                    "logical_evaluations_per_step": evals, "lane_slots_per_evaluation": lane_slots / evals,
                    "algorithmic_flops_per_step": flops, "algorithmic_flop_per_lane_slot": flops / lane_slots,
                    "useful_lane_fraction": (flops / 2.0) / lane_slots,  # one FMA = 2 flop per lane-slot at best
                    # ONE busy figure: issued wave64 VALU instructions against what the SIMDs can issue -- the measured ceiling
                    # of 0.429 instructions per cycle and SIMD (2.33 cycles each; profiles/tools/valu_issue_micro.hip,
                    # profiles/r04/valu_issue_micro.txt: eight independent v_fma_f32 chains at 2-4 wavefronts per SIMD; a lone
                    # wavefront issues one per 4.58 cycles whatever its dependences), 1024 SIMDs at the nominal 2.4 GHz
                    "valu_issue_busy_frac": ent["sq_insts_valu"] / (VALU_ISSUE_CEILING * 1024 * 2.4e9 * kern_ms * 1e-3),
                    "valu_issue_ceiling_insts_per_cycle_per_simd": VALU_ISSUE_CEILING,
                    "sq_active_inst_valu_raw": ent.get("sq_active_inst_valu"),  # raw counter (quad-cycles), not a busy figure
                    "note": "valu_issue_busy_frac = SQ_INSTS_VALU / (0.429 x SIMD-cycles of the launch): 1.0 = every SIMD issuing "
                            "VALU back to back from several wavefronts"}
    except (FileNotFoundError, KeyError):
        pass
    if world > 1 and args.solver == "pg":
        # The q_phase shards without a collective, so a scaling line of this mode would prove only the timing barrier.  The one
        # data-path exchange of the engine -- the offset phase's 3K + 2 partial sums -- is therefore exercised
        # on the same ranks: this rank's sums over its own first frames, all-reduced and finished on every rank.  Never `value`.
        from stac_mjx_amd.dist import offset_phase_exchange_probe

        n_s = min(frames_step, 100)
        q_s = out["qpos"].reshape(-1, fs.tables.nq)[:n_s] if C > 0 else torch.empty((0, fs.tables.nq), device=eng.device)
        part = eng.m_partial(kp.reshape(-1, kp.shape[-1])[:n_s], q_s)  # (no frames on this rank: sums of zeros, T = 0)
        off_now = eng.get_site_pos()
        is_reg = torch.ones_like(off_now)
        probe = offset_phase_exchange_probe(part, lambda red: eng.m_finish(red, off_now, is_reg, 1.0)[0])
        line["config"]["offset_phase_exchange"] = dict(probe, note="one all-reduce (all-gather + fixed-order sum) of the offset "
                                                       "phase's partial sums per calibration iteration; --mode fit runs the whole calibration")
    if rank == 0 and world == 1 and args.solver == "pg" and args.model == "rodent" and F == 1 and not args.no_extras and not strong:
        # BASELINE configs[1] with the reference's DEFAULT chaining (configs/stac/stac.yaml:9, n_frames_per_clip = 250):
        # the same number of frames as 40 warm-started clips -- latency mode, one chain per workgroup.  Never `value`.
        Fc = 250
        Cc = max(frames_step // Fc, 1)
        kpc_host, _ = synth_keypoints(fs, fk, Cc, Fc, seed=7, noise_seed=8)
        kpc = torch.as_tensor(kpc_host).to(eng.device)
        oc = None
        cev = []
        for i in range(2):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            oc = eng.q_phase(kpc, part_masks=fs.part_masks, trunk_kps=fs.trunk_kps, root_kp_idx=fs.root_kp_idx,
                             root_dims=fs.root_dims, do_root_opt=fs.do_root_opt, want_bodies=False, want_markers=False, want_carry=False, out=oc)
            e1.record()
            cev.append((e0, e1))
        torch.cuda.synchronize()
        c_ms = float(cev[1][0].elapsed_time(cev[1][1]))
        ccnt = oc["counters"].to(torch.float64).sum(dim=(0, 1)).cpu().numpy()
        cerr = torch.linalg.norm((eng.fk(oc["qpos"].reshape(-1, fs.tables.nq), want=("site_xpos",))["site_xpos"]
                                  - kpc.reshape(-1, fs.tables.nsite, 3)), dim=-1)
        line["config"]["clips250"] = {
            "workload": f"{Cc * Fc} frames as {Cc} warm-started clips of {Fc} frames (reference default n_frames_per_clip)",
            "frames_per_s": Cc * Fc / (c_ms * 1e-3), "kernel_ms": c_ms,
            "iters_per_frame": ccnt[0] / (Cc * Fc), "us_per_pg_iteration": c_ms * 1e3 / (ccnt[0] / Cc),
            "marker_rmse_mm": float(torch.sqrt((cerr ** 2).mean()).item() * 1e3),
            "note": "each clip is one serial chain of 250 x ~410 PG iterations: speculative latency mode, 4 wavefronts per chain"}
    if rank == 0 and world == 1 and args.solver == "pg" and args.model == "rodent" and not args.no_extras and not strong:
        # the north star words the q_phase as a Levenberg-Marquardt update; the reference runs projected gradient (the
        # `value` above, parity mode).  The optional LM solver on the same resident batch, for the record -- never `value`.
        lm = Engine(fs.tables, fs.lb, fs.ub, tol=float(cfg["FTOL"]), maxiter=int(cfg["N_ITER_Q"]), device=f"cuda:{local_rank}",
                    solver="lm", lm_maxiter=args.lm_maxiter)
        lm.set_site_pos(synth_offsets(fs))
        lo = None
        lev = []
        for i in range(1 + args.steps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            lo = lm.q_phase(kp, part_masks=fs.part_masks, trunk_kps=fs.trunk_kps, root_kp_idx=fs.root_kp_idx,
                            root_dims=fs.root_dims, do_root_opt=fs.do_root_opt, want_bodies=False, want_markers=False, want_carry=False, out=lo)
            e1.record()
            lev.append((e0, e1))
        torch.cuda.synchronize()
        lm_ms = float(np.mean([a.elapsed_time(b) for a, b in lev[1:]]))
        lerr = torch.linalg.norm((eng.fk(lo["qpos"].reshape(-1, fs.tables.nq), want=("site_xpos",))["site_xpos"]
                                  - kp.reshape(-1, fs.tables.nsite, 3)), dim=-1)
        line["config"]["lm_solver_same_batch"] = {
            "frames_per_s": frames_step / (lm_ms * 1e-3), "kernel_ms": lm_ms,
            "marker_rmse_mm": float(torch.sqrt((lerr ** 2).mean()).item() * 1e3),
            "lm_maxiter": args.lm_maxiter,
            "note": "stac_q_params.solver = STAC_SOLVER_LM; not the reference's algorithm, judged in marker space only"}
    if rank == 0 and world == 1 and not args.no_cpu_baseline and args.solver == "pg" and C > 0:
        line["cpu_baseline"] = cpu_baseline(fs, cfg, kp_host)
    if rank == 0:
        print(json.dumps(line))
    if dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
