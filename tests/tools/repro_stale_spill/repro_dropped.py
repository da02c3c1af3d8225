"""Round-3 defect reproduction: the dropped latency shape q_phase_kernel<8,16,2,8> (built from the round-3 sources with
different compiler flags: tests/tools/repro_stale_spill/build_repro.sh) on random models with more than 80 coordinates, each launched three
times on one engine; optional poison of scratch + vector registers before every launch.
usage: STAC_HIP_LIB=build/libr3_A.so python tests/tools/repro_stale_spill/repro_dropped.py [nseeds] [poison hex|none]"""
import ctypes, os, sys
import numpy as np
os.environ["STAC_HIP_SPEC"] = "1"; os.environ["STAC_HIP_SPECG"] = "8"
os.environ["STAC_TEST_POISON"] = "off"
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import torch
import test_gpu_parity as T
from oracle import Oracle
from stac_mjx_amd.engine import Engine
from stac_mjx_amd.mjcf import JNT_FREE, JNT_BALL
nseeds = int(sys.argv[1]) if len(sys.argv) > 1 else 40
pat = sys.argv[2] if len(sys.argv) > 2 else "none"
P = None
if pat != "none":
    P = ctypes.CDLL(os.path.abspath("tests/tools/libpoison.so")); P.poison_scratch.argtypes = [ctypes.c_void_p, ctypes.c_uint32]
bad_oracle = bad_repeat = ran = 0
for seed in range(nseeds):
    rng = np.random.default_rng(9000 + seed)
    t = T._random_tables(rng, int(rng.integers(70, 111)), True, p_ball=0.0, max_children_bias=float(rng.choice([0.3, 0.6, 0.9])))
    if not (80 < t.nq <= 128):
        continue
    nq, K = t.nq, t.nsite
    lb, ub = np.full(nq, -np.inf, np.float32), np.full(nq, np.inf, np.float32)
    for j in range(t.njnt):
        a, ty = int(t.jnt_qposadr[j]), int(t.jnt_type[j])
        if ty == JNT_FREE: lb[a + 3:a + 7], ub[a + 3:a + 7] = -1, 1
        else: lb[a], ub[a] = min(t.jnt_range[j, 0], 0.0), t.jnt_range[j, 1]
    orc = Oracle(t, tol=1e-5, maxiter=10)
    C, F = 7, 2
    q = np.tile(t.qpos0, (C * F, 1)) + rng.normal(0, 0.15, (C * F, nq)).astype(np.float32)
    q = np.clip(q, np.where(np.isfinite(lb), lb, -3), np.where(np.isfinite(ub), ub, 3)).astype(np.float32)
    kp = np.stack([orc.fk(x.copy())["site_xpos"].reshape(-1) for x in q]).astype(np.float32).reshape(C, F, 3 * K)
    part = np.zeros((1, nq), np.uint8); part[0] = rng.random(nq) < 0.3
    trunk = np.ones(K, np.uint8)
    ref = orc.ik_clips(kp, lb, ub, part, trunk, 0, 7, do_root_opt=True)
    eng = Engine(t, lb, ub, tol=1e-5, maxiter=10)
    outs = []
    for rep in range(3):
        if P: assert P.poison_scratch(ctypes.c_void_p(torch.cuda.current_stream().cuda_stream), int(pat, 16)) == 0
        r = eng.q_phase(kp, part_masks=part, trunk_kps=trunk, root_kp_idx=0, root_dims=7, do_root_opt=True)
        outs.append((r["qpos"].cpu().numpy().copy(), r["counters"].cpu().numpy().copy()))
    ran += 1
    eq_o = [bool((o[0] == ref["qpos"]).all() and (o[1].astype(np.uint32) == ref["counters"]).all()) for o in outs]
    eq_r = [bool((outs[0][0] == o[0]).all() and (outs[0][1] == o[1]).all()) for o in outs[1:]]
    if not all(eq_o): bad_oracle += 1
    if not all(eq_r): bad_repeat += 1
    if not all(eq_o) or not all(eq_r):
        print(f"seed {seed} nq {nq}: equals oracle per launch {eq_o}, launches 2,3 equal launch 1 {eq_r}", flush=True)
    eng.close()
print(f"{os.environ.get('STAC_HIP_LIB')} poison={pat}: ran {ran} models, {bad_oracle} differ from the oracle, {bad_repeat} differ between launches")
