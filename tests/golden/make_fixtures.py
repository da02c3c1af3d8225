#!/usr/bin/env python3
"""Generate the committed fixtures under tests/golden/ from the read-only reference checkout.

Run in the build container (needs /root/reference; never on the GPU box):

    python tests/golden/make_fixtures.py [--reference /root/reference]

Fixtures are DATA only (inputs / expected outputs / compiled model tables), no reference source:

  demo_viz_golden.npz       the reference's own stored fit output demos/demo_viz.p (a legacy pickle of
                            jax arrays, read without jax): qpos[50,74], xpos[50,67,3],
                            walker_body_sites[50,23,3], offsets[23,3], kp_data[50,69] + name lists.
                            This is the FK known-answer test.
  rodent_tables.npz         ModelTables compiled by stac_mjx_amd.mjcf from models/rodent.xml with the
                            keypoint sites and SCALE_FACTOR of configs/model/rodent.yaml (current
                            rescale.py rule).
  rodent_tables_legacy.npz  same with the scaling rule of the build that produced demo_viz.p.
  rodent_model_cfg.json     the `model` config group (configs/model/rodent.yaml) as JSON.
  rodent_mocap_1000.npy     tests/data/test_rodent_mocap_1000_frames.mat through load_data semantics:
                            float32 [1000, 69] metres, KEYPOINT_MODEL_PAIRS order.
  fly_tables.npz / fly_model_cfg.json
                            fruitfly_force_free.xml + configs/model/fly_tethered.yaml (BASELINE config 5).
  mouse_tables.npz / mouse_model_cfg.json
                            mouse_with_meshes.xml + configs/model/mouse.yaml (nq = 230, K = 34).
  mouse_mocap_200.npy       the first 200 frames of tests/data/test_mouse_mocap_3600_frames.h5 through load_data
                            semantics (configs/model/mouse.yaml): float32 [200, 102], KEYPOINT_MODEL_PAIRS order.  Read
                            under /opt/conda/bin/python3.9 (the interpreter of this image that has h5py).
  oracle_regress.npz        outputs of THIS repo's oracle on a few real frames (regression pin of the
                            oracle itself; not a reference pin).
"""

from __future__ import annotations

import argparse
import json
import pickle
import sys
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
ROOT = HERE.parents[1]
sys.path.insert(0, str(ROOT))

from stac_mjx_amd.config import compose_config  # noqa: E402
from stac_mjx_amd.fit_model import build_fit_setup  # noqa: E402
from stac_mjx_amd.io import load_data  # noqa: E402


class _JaxFreeUnpickler(pickle.Unpickler):
    """demo_viz.p holds jax arrays; rebuild them as numpy arrays without importing jax."""

    def find_class(self, module, name):
        if module == "jax._src.array" and name == "_reconstruct_array":
            def rec(fun, args, arr_state, aval_state):
                v = fun(*args)
                v.__setstate__(arr_state)
                return v
            return rec
        if module.startswith("numpy.core"):
            module = module.replace("numpy.core", "numpy._core")
        return super().find_class(module, name)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reference", default="/root/reference")
    args = ap.parse_args()
    ref = Path(args.reference)

    # 1. stored reference output -------------------------------------------------------------
    with open(ref / "demos" / "demo_viz.p", "rb") as fh:
        d = _JaxFreeUnpickler(fh).load()
    np.savez_compressed(
        HERE / "demo_viz_golden.npz",
        qpos=np.asarray(d["qpos"], np.float32),
        xpos=np.asarray(d["xpos"], np.float32),
        walker_body_sites=np.asarray(d["walker_body_sites"], np.float32),
        offsets=np.asarray(d["offsets"], np.float32).reshape(-1, 3),
        kp_data=np.asarray(d["kp_data"], np.float32),
        names_qpos=np.array(d["names_qpos"], dtype=np.str_),
        names_xpos=np.array(d["names_xpos"], dtype=np.str_),
        kp_names=np.array(d["kp_names"], dtype=np.str_),
    )

    # 2. rodent tables + config ---------------------------------------------------------------
    cfg = compose_config(ref / "configs", "config")
    kp_names = list(cfg.model.KEYPOINT_MODEL_PAIRS.keys())
    xml = ref / cfg.model.MJCF_PATH
    build_fit_setup(xml, cfg.model, kp_names).tables.save(HERE / "rodent_tables.npz")
    build_fit_setup(xml, cfg.model, kp_names, legacy_joint_pos_scale=True).tables.save(HERE / "rodent_tables_legacy.npz")
    with open(HERE / "rodent_model_cfg.json", "w") as fh:
        json.dump(cfg.model.to_dict(), fh, indent=1)

    # 3. real mocap through load_data ----------------------------------------------------------
    kp, names = load_data(cfg, base_path=ref)
    assert names == kp_names and kp.shape == (1000, 69) and kp.dtype == np.float32
    np.save(HERE / "rodent_mocap_1000.npy", kp)

    # 4. fruit fly (tethered) ------------------------------------------------------------------
    import yaml

    fly = yaml.safe_load(open(ref / "configs" / "model" / "fly_tethered.yaml"))
    fly_names = list(fly["KEYPOINT_MODEL_PAIRS"].keys())
    fs = build_fit_setup(ref / fly["MJCF_PATH"], fly, fly_names)
    fs.tables.save(HERE / "fly_tables.npz")
    with open(HERE / "fly_model_cfg.json", "w") as fh:
        json.dump(fly, fh, indent=1)

    # 4b. mouse (nq = 230, 225 bodies, K = 34: stresses level count, nq capacity and the > 32-site loss tree)
    mouse = yaml.safe_load(open(ref / "configs" / "model" / "mouse.yaml"))
    mouse_names = list(mouse["KEYPOINT_MODEL_PAIRS"].keys())
    build_fit_setup(ref / mouse["MJCF_PATH"], mouse, mouse_names).tables.save(HERE / "mouse_tables.npz")
    with open(HERE / "mouse_model_cfg.json", "w") as fh:
        json.dump(mouse, fh, indent=1)

    # 4c. real mouse mocap (h5py lives in the image's python3.9 only: run load_data there)
    import subprocess

    h5py_python = "/opt/conda/bin/python3.9"
    code = (
        "import sys, json; sys.path.insert(0, sys.argv[1]); import numpy as np\n"
        "from stac_mjx_amd import io; from stac_mjx_amd.config import validate_config\n"
        "m = json.load(open(sys.argv[1] + '/tests/golden/mouse_model_cfg.json'))\n"
        "cfg = validate_config({'model': m, 'stac': dict(fit_offsets_path='f.h5', ik_only_path='i.h5', continuous=False,\n"
        "    data_path='tests/data/test_mouse_mocap_3600_frames.h5', n_fit_frames=1, skip_fit_offsets=False,\n"
        "    skip_ik_only=False, infer_qvels=False, n_frames_per_clip=1, mujoco=dict(solver='newton', iterations=1, ls_iterations=4))})\n"
        "kp, names = io.load_data(cfg, sys.argv[2])\n"
        "assert kp.shape == (3600, 102) and names == list(m['KEYPOINT_MODEL_PAIRS'].keys())\n"
        "np.save(sys.argv[3], kp[:200])\n")
    subprocess.run([h5py_python, "-c", code, str(ROOT), str(ref), str(HERE / "mouse_mocap_200.npy")], check=True)

    # 5. oracle regression pin -------------------------------------------------------------------
    from oracle import Oracle

    fsr = build_fit_setup(xml, cfg.model, kp_names)
    orc = Oracle(fsr.tables, tol=float(cfg.model.FTOL), maxiter=int(cfg.model.N_ITER_Q))
    clips = kp[:8].reshape(4, 2, 69)
    out = orc.ik_clips(clips, fsr.lb, fsr.ub, fsr.part_masks, fsr.trunk_kps, fsr.root_kp_idx, fsr.root_dims)
    np.savez_compressed(HERE / "oracle_regress.npz", kp=clips, qpos=out["qpos"], marker_sites=out["marker_sites"],
                        frame_error=out["frame_error"], counters=out["counters"])
    print("fixtures written to", HERE)


if __name__ == "__main__":
    main()
