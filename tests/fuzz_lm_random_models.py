"""Not collected by pytest: `python tests/fuzz_lm_random_models.py N` on a GPU box.  The optional LM solver (stac.solver: lm) on N random
models (5 / 40 / 130 chains x 1-2 frames, launched twice): no fault, finite, repeatable, inside the box wherever the start pose is,
and -- since round 5, when oracle/stac_oracle.c::q_opt_lm_ws became the kernel's operation sequence -- EQUAL to the oracle's LM bit for
bit (qpos, residuals, counters).  Since the same round ball joints are solved too (four raw quaternion coordinates per joint): a
refusal can only be a capacity limit."""
import sys, numpy as np, torch
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import test_gpu_parity as T
from oracle import Oracle
from stac_mjx_amd.engine import Engine, StacHipError
from stac_mjx_amd.mjcf import JNT_BALL, JNT_FREE
bad = 0; ran = 0; refused = 0; far = 0
for seed in range(int(sys.argv[1])):
    rng = np.random.default_rng(70000 + seed)
    free_root = bool(rng.integers(2))
    t = T._random_tables(rng, int(rng.integers(3, 90)), free_root, p_ball=float(rng.choice([0.0, 0.1, 0.3])), max_children_bias=float(rng.choice([0.05, 0.3, 0.6, 0.9])))
    nq, K = t.nq, t.nsite
    if nq == 0: continue
    lb, ub = np.full(nq, -np.inf, np.float32), np.full(nq, np.inf, np.float32)
    for j in range(t.njnt):
        a, ty = int(t.jnt_qposadr[j]), int(t.jnt_type[j])
        if ty == JNT_FREE: lb[a + 3:a + 7], ub[a + 3:a + 7] = -1, 1
        elif ty == JNT_BALL: lb[a:a + 4], ub[a:a + 4] = -1, 1
        else: lb[a], ub[a] = min(t.jnt_range[j, 0], 0.0), t.jnt_range[j, 1]
    orc = Oracle(t, tol=1e-4, maxiter=50)
    C, F = int((5, 40, 130)[seed % 3]), 1 + seed % 2
    n = C * F
    q = np.tile(t.qpos0, (n, 1)) + rng.normal(0, 0.15, (n, nq)).astype(np.float32)
    q = np.clip(q, np.where(np.isfinite(lb), lb, -3), np.where(np.isfinite(ub), ub, 3)).astype(np.float32)
    kp = np.stack([orc.fk(x.copy())["site_xpos"].reshape(-1) for x in q]).astype(np.float32)
    kp = (kp + rng.normal(0, 1e-3, kp.shape)).astype(np.float32).reshape(C, F, 3 * K)
    P = int(rng.integers(0, 3))
    part = np.zeros((P, nq), np.uint8)
    for i in range(P): part[i] = rng.random(nq) < 0.4
    trunk = (rng.random(K) < 0.6).astype(np.uint8); trunk[0] = 1
    try:
        eng = Engine(t, lb, ub, tol=1e-4, solver="lm", lm_maxiter=15)
        kw = dict(part_masks=part, trunk_kps=trunk, root_kp_idx=0, root_dims=7, do_root_opt=free_root)
        r1 = eng.q_phase(kp, **kw); r2 = eng.q_phase(kp, **kw)
        torch.cuda.synchronize()
    except StacHipError as e:
        refused += 1; print("seed", seed, "refused:", str(e)[:100]); continue
    ran += 1
    qq = r1["qpos"].cpu().numpy()
    fin = np.isfinite(lb) & np.isfinite(ub) & (t.qpos0 >= lb) & (t.qpos0 <= ub)  # (LM leaves coordinates alone that move no fit site)
    c_fin = bool(np.isfinite(qq).all()); c_rep = bool((r1["qpos"] == r2["qpos"]).all())
    c_box = bool((qq[..., fin] >= lb[fin] - 1e-6).all() and (qq[..., fin] <= ub[fin] + 1e-6).all())
    ok = c_fin and c_rep and c_box
    if not ok:
        viol = np.maximum(lb[fin] - qq[..., fin], qq[..., fin] - ub[fin]).max()
        print("   finite", c_fin, "repeatable", c_rep, "in box", c_box, "worst violation %.3g" % viol, "types", sorted(set(int(x) for x in t.jnt_type)))
    ms = r1["marker_sites"].cpu().numpy().reshape(C, F, K, 3)
    err = np.sqrt(((ms - kp.reshape(C, F, K, 3)) ** 2).sum(-1).mean())
    ref = orc.ik_clips_lm(kp, lb, ub, part, trunk, 0, 7, do_root_opt=free_root, maxiter=15)
    eref = np.sqrt(((ref["marker_sites"].reshape(C, F, K, 3) - kp.reshape(C, F, K, 3)) ** 2).sum(-1).mean())
    dm = np.abs(ms - ref["marker_sites"].reshape(C, F, K, 3)).max()
    exact = (np.array_equal(qq.view(np.uint32), ref["qpos"].view(np.uint32))
             and np.array_equal(r1["counters"].cpu().numpy().astype(np.uint32), ref["counters"])
             and np.array_equal(r1["frame_error"].cpu().numpy().view(np.uint32), ref["frame_error"].view(np.uint32)))
    if not exact: far += 1
    if not ok or not exact:
        bad += 1; print("seed", seed, "BAD ok", ok, "exact", exact, "rmse hip %.4g oracle %.4g maxdiff %.3g nq %d K %d" % (err, eref, dm, nq, K))
    eng.close()
print("ran", ran, "refused", refused, "bad", bad, "not bit-identical to the oracle's LM", far)
