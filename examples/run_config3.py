#!/usr/bin/env python3
"""BASELINE.json configs[2]: full rodent fit (offset_phase + q_phase alternation on the real 1000-frame mocap
fixture, N_ITERS = 6) followed by a 100 000-frame ik_only (n_frames_per_clip = 250) on one MI355X.

    python examples/run_config3.py [--fit-frames 1000] [--ik-frames 100000]

Prints timings, the marker error before/after the offset fit and the iterations spent.  Data: the committed
fixture tests/golden/rodent_mocap_1000.npy (reference's tests/data/*.mat through load_data semantics) for the
fit; synthetic motion (stac_mjx_amd/synth.py) generated with the FITTED offsets for the long ik_only.
"""

import argparse
import json
import sys
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))

from stac_mjx_amd.config import validate_config  # noqa: E402
from stac_mjx_amd.fit_model import finish_fit_setup  # noqa: E402
from stac_mjx_amd.mjcf import ModelTables  # noqa: E402
from stac_mjx_amd.stac import Stac  # noqa: E402
from stac_mjx_amd.synth import synth_keypoints  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--fit-frames", type=int, default=1000)
    ap.add_argument("--ik-frames", type=int, default=100000)
    ap.add_argument("--n-iters", type=int, default=6)
    ap.add_argument("--fit-frames-per-clip", type=int, default=0,
                    help="0 = the reference's single warm-started chain; F > 0 = independent clips of F fit frames (engine extension)")
    args = ap.parse_args()
    g = ROOT / "tests" / "golden"
    mcfg = json.load(open(g / "rodent_model_cfg.json"))
    mcfg["N_ITERS"] = args.n_iters
    stac_cfg = dict(fit_offsets_path="fit.h5", ik_only_path="ik.h5", data_path="-", continuous=False,
                    n_fit_frames=args.fit_frames, skip_fit_offsets=False, skip_ik_only=False, infer_qvels=False,
                    n_frames_per_clip=250, mujoco=dict(solver="newton", iterations=1, ls_iterations=4),
                    fit_frames_per_clip=args.fit_frames_per_clip)
    cfg = validate_config({"model": mcfg, "stac": stac_cfg})
    kp_names = list(mcfg["KEYPOINT_MODEL_PAIRS"].keys())
    fs = finish_fit_setup(ModelTables.load(g / "rodent_tables.npz"), mcfg, kp_names)
    stac = Stac(None, cfg, kp_names, setup=fs, verbose=False)
    kp = np.load(g / "rodent_mocap_1000.npy")[: args.fit_frames]

    def marker_err(data, kpd):
        return float(np.linalg.norm(data.marker_sites - kpd.reshape(len(kpd), -1, 3), axis=-1).mean() * 1e3)

    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fit = stac.fit_offsets(kp)
    torch.cuda.synchronize()
    t_fit = time.perf_counter() - t0
    print(f"fit_offsets: {args.fit_frames} frames x ({args.n_iters} + 1) pose passes in {t_fit:.1f} s "
          f"({args.fit_frames * (args.n_iters + 1) / t_fit:.0f} frame-solves/s, "
          f"{'ONE warm-started chain' if not args.fit_frames_per_clip else str(args.fit_frames // args.fit_frames_per_clip) + ' independent clips'}); "
          f"mean marker error {marker_err(fit, kp):.2f} mm; max |offset change| "
          f"{np.abs(fit.offsets - fs.tables.site_pos).max() * 1e3:.1f} mm")

    # long ik_only on synthetic motion consistent with the fitted offsets
    stac.engine.set_site_pos(fit.offsets)
    fk = lambda q: stac.engine.fk(q, want=("site_xpos",))["site_xpos"].cpu().numpy()
    C = args.ik_frames // 250
    kp_long, _ = synth_keypoints(fs, fk, C, 250, seed=7, noise_seed=8)
    kp_long = kp_long.reshape(-1, kp_long.shape[-1])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ik = stac.ik_only(kp_long, fit.offsets)
    torch.cuda.synchronize()
    t_ik = time.perf_counter() - t0
    print(f"ik_only: {len(kp_long)} frames in {C} clips of 250 in {t_ik:.2f} s = {len(kp_long) / t_ik:.0f} frames/s "
          f"(incl. host packaging); mean marker error {marker_err(ik, kp_long):.2f} mm (1 mm synthetic noise)")


if __name__ == "__main__":
    main()
