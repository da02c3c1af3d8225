#!/usr/bin/env python3
"""Variant sweep: why does the restated ProjectedGradient not reproduce the reference notebook's printed run?

TEST INFRASTRUCTURE (CPU only; drives the oracle's loss/gradient, never the product path).

`/root/reference/demos/rodent_demo.ipynb` (cell 6 output) is the one real run of the reference in the tree:
run_stac on frames 0-9 of tests/data/test_rodent_mocap_1000_frames.mat with configs/model/rodent.yaml.  It printed

    Root optimization ... error of 4.3102449126308784e-05
    pose pass 1 (before any offset update):  Mean 3.5538312658900395e-05   Standard deviation 9.482042514719069e-06

The oracle (oracle/stac_oracle.c, `q_opt_ws` + drivers) gives 9.50e-5 after 30 iterations / 4.07e-5 +- 2.08e-5 on the same
frames and config, stably.  This script re-runs `root_optimization` + the first `pose_optimization` pass under
every variant of (1) the from-memory jaxopt details of SURVEY.md A2 and (2) the host-side rules that differ between the
current reference source and the OLDER source the notebook was demonstrably produced with (its print strings --
"Optimizing first 7 qposes for root", "starting offset optimization", "Final error of", 40 s iterative offset
phases, config key N_ITER_M -- do not exist in the current stac_mjx/compute_stac.py:49-104,142-165), and prints a table
against the notebook's numbers.

The solver loop below is a numpy float32 twin of `q_opt_ws` (same operation order except the reduction trees, which
numpy sums pairwise in its own order); loss and gradient come from the oracle (`orc_q_loss`), so FK and the analytic
gradient are the pinned ones.  The unmodified twin reproduces the C oracle's counters and residuals (checked by
tests/test_oracle.py::test_variant_sweep_baseline_equals_oracle).

Usage:  python tests/tools/jaxopt_variant_sweep.py [--jobs 8] [--out profiles/r03/jaxopt_variant_sweep.txt] [--quick]
Follows: stac_mjx/stac_core.py:66-99, stac_mjx/compute_stac.py:17-104,170-278, demos/rodent_demo.ipynb:102-107.
"""

from __future__ import annotations

import argparse
import itertools
import json
import math
import sys
from concurrent.futures import ProcessPoolExecutor
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

GOLDEN = ROOT / "tests" / "golden"

# what the notebook printed (demos/rodent_demo.ipynb:102-107)
NB_ROOT = 4.3102449126308784e-05
NB_MEAN = 3.5538312658900395e-05
NB_STD = 9.482042514719069e-06

F32 = np.float32
EPS32 = F32(1.1920929e-7)

# ---------------------------------------------------------------------------------------------------------------------
# variant description: every key has the value the oracle uses as its default
DEFAULT = dict(
    # --- jaxopt details ---
    eps="f32",            # slack of the sufficient-decrease test: "f32" (1.19e-7) | "f64" (2.2e-16) | 0 | any float
    eps_side="rhs",       # "rhs": lhs > rhs + eps   | "lhs": lhs + eps > rhs   | "scaled": eps multiplied by the step size
    ls_cmp=">",           # ">" | ">="
    ls_form="scaled",     # "scaled": eta (f_c - f_y) > eta <d,g> + 0.5 |d|^2   | "classic": f_c > f_y + <d,g> + |d|^2 / (2 eta)
    grow="div",           # next step size: "div" eta/0.5 (reset to 1 when eta <= reset_thresh) | "keep" | "one" (restart at 1 every iteration)
    reset_thresh=1e-6,
    err_at="x_unit",      # stopping residual: "x_unit" |clip(x'-g(x'))-x'|  | "x_step" same with the accepted step size
                          # | "x_step_scaled" the former divided by the step size | "y_unit" |clip(y-g(y))-y| (no extra gradient)
    project_x0=False,     # clip q0 into the box before the first iteration
    maxls=15,
    momentum="t",         # "t": beta = (t-1)/t_next | "tnext": beta = (t_next-1)/t_next | "k": beta = k/(k+3)
    accel=True,
    stop_cmp=">",         # loop while error > tol | ">="
    maxiter=400,
    stepsize0=1.0,
    # --- host rules (current source vs what the older source is known or suspected to have done) ---
    root_quat_bounds="pm1",   # "pm1": [-1,1] on the raw root quaternion (stac.py:54-88) | "inf": +-inf on all seven root coordinates
    lb_min0=True,             # lb = min(lb, 0)  (stac.py:88)
    tables="current",         # "current" rescale rule | "legacy" dm_control rule (demo_viz.p's)
    reseed_second=True,       # second root solve starts from the keypoint again (compute_stac.py:80-81)
    root_seed="3idx",         # q0[:3] = kp[3 idx : 3 idx + 3] | "idx": kp[idx : idx+3] | "none": keep qpos0's position
    root_mask="trunk",        # keypoints of the root solves: "trunk" | "all"
    root_passes=2,
    tol=1e-4,
    quat_grad="tangent",      # "tangent": (I - q^q^T)/|q| as autodiff through normalize gives
    writeback_norm=True,      # FK writes the normalised root quaternion back into qpos (mjx kinematics)
)


def describe(over):
    return ", ".join(f"{k}={v}" for k, v in over.items()) or "baseline (oracle as committed)"


# ---------------------------------------------------------------------------------------------------------------------
class Env:
    """Model, data and oracle handles (one per process)."""

    _cache = {}

    def __init__(self, tables_kind):
        from oracle import Oracle
        from stac_mjx_amd.fit_model import finish_fit_setup
        from stac_mjx_amd.mjcf import ModelTables

        with open(GOLDEN / "rodent_model_cfg.json") as fh:
            cfg = json.load(fh)
        name = "rodent_tables.npz" if tables_kind == "current" else "rodent_tables_legacy.npz"
        tables = ModelTables.load(GOLDEN / name)
        self.fs = finish_fit_setup(tables, cfg, list(cfg["KEYPOINT_MODEL_PAIRS"].keys()))
        self.orc = Oracle(tables)
        self.kp = np.load(GOLDEN / "rodent_mocap_1000.npy")[:10].astype(np.float32)

    @classmethod
    def get(cls, kind):
        if kind not in cls._cache:
            cls._cache[kind] = cls(kind)
        return cls._cache[kind]


def _eps(v):
    e = v["eps"]
    if e == "f32":
        return EPS32
    if e == "f64":
        return F32(2.220446049250313e-16)
    return F32(e)


def pg_solve(env, v, kp, qs, ks, q0, lb, ub):
    """numpy float32 twin of oracle/stac_oracle.c::q_opt_ws with the variant switches of DEFAULT."""
    orc = env.orc
    qs8, ks8 = qs.astype(np.uint8), ks.astype(np.uint8)
    q0 = q0.astype(F32)

    def vg(p):
        l, g = orc.q_loss(p, kp, qs8, ks8, q0, with_grad=True)
        return F32(l), g

    def val(p):
        l, _ = orc.q_loss(p, kp, qs8, ks8, q0, with_grad=False)
        return F32(l)

    clip = lambda a: np.minimum(np.maximum(a, lb), ub)  # noqa: E731
    eps = _eps(v)
    x = clip(q0) if v["project_x0"] else q0.copy()
    y = x.copy()
    step, t, err = F32(v["stepsize0"]), F32(1), F32(np.inf)
    it = ls_evals = 0
    tol = F32(v["tol"])
    half = F32(0.5)
    while it < v["maxiter"]:
        src = y if v["accel"] else x
        f, g = vg(src)
        eta = F32(1) if v["grow"] == "one" else step
        cand = clip(src - eta * g)
        n = 0
        while n < v["maxls"]:
            fc = val(cand)
            ls_evals += 1
            d = cand - src
            sq = F32(np.dot(d, d))
            vd = F32(np.dot(d, g))
            if v["ls_form"] == "scaled":
                lhs = eta * (fc - f)
                rhs = eta * vd + half * sq
                e = eps * eta if v["eps_side"] == "scaled" else eps
            else:
                lhs = fc
                rhs = f + vd + sq / (F32(2) * eta)
                e = eps
            if v["eps_side"] == "lhs":
                bad = (lhs + e > rhs) if v["ls_cmp"] == ">" else (lhs + e >= rhs)
            else:
                bad = (lhs > rhs + e) if v["ls_cmp"] == ">" else (lhs >= rhs + e)
            if not bad:
                break
            eta = eta * half
            cand = clip(src - eta * g)
            n += 1
        if v["grow"] == "div":
            nstep = F32(1) if eta <= F32(v["reset_thresh"]) else eta / half
        else:
            nstep = eta
        if v["accel"]:
            tn = half * (F32(1) + np.sqrt(F32(1) + F32(4) * t * t))
            if v["momentum"] == "t":
                beta = (t - F32(1)) / tn
            elif v["momentum"] == "tnext":
                beta = (tn - F32(1)) / tn
            else:
                beta = F32(it) / F32(it + 3)
            y = (cand + beta * (cand - x)).astype(F32)
            t = tn
        xprev, x = x, cand
        if v["err_at"] == "y_unit":
            r = clip(src - g) - src
            err = F32(np.sqrt(np.dot(r, r)))
        else:
            _, gn = vg(x)
            s = F32(1) if v["err_at"] == "x_unit" else eta
            r = clip(x - s * gn) - x
            err = F32(np.sqrt(np.dot(r, r)))
            if v["err_at"] == "x_step_scaled":
                err = err / s
        step = nstep
        it += 1
        if not ((err > tol) if v["stop_cmp"] == ">" else (err >= tol)):
            break
    return x, dict(iter_num=it, error=float(err), ls_evals=ls_evals, stepsize=float(step))


def writeback(env, v, q):
    """replace_qs (utils.py:147-169): FK writes the normalised free-joint quaternion back."""
    return env.orc.fk(q)["qpos"] if v["writeback_norm"] else q.astype(F32)


def run_variant(over, passes=True):
    v = dict(DEFAULT)
    v.update(over)
    env = Env.get(v["tables"])
    fs, kp = env.fs, env.kp
    lb, ub = fs.lb.copy(), fs.ub.copy()
    if not v["lb_min0"]:
        # undo lb = min(lb, 0): rodent hinges take their range's lower end (every rodent joint but the root has a range)
        t = fs.tables
        for j in range(1, t.njnt):
            r = t.jnt_range[j]
            if not (r[0] == 0 and r[1] == 0):
                lb[int(t.jnt_qposadr[j])] = r[0]
    if v["root_quat_bounds"] == "inf":
        lb[:7], ub[:7] = -np.inf, np.inf
    nq, K = fs.tables.nq, fs.tables.nsite
    q = fs.tables.qpos0.astype(F32).copy()

    # ---- root_optimization (compute_stac.py:17-104) ----
    ridx = fs.root_kp_idx
    if v["root_seed"] == "3idx":
        root_xyz = kp[0, 3 * ridx : 3 * ridx + 3]
    elif v["root_seed"] == "idx":
        root_xyz = kp[0, ridx : ridx + 3]
    else:
        root_xyz = None
    qs = np.zeros(nq, bool)
    qs[: fs.root_dims] = True
    ks = np.repeat(fs.trunk_kps, 3) if v["root_mask"] == "trunk" else np.ones(3 * K, bool)
    st = dict(error=float("nan"), iter_num=0)
    root_iters = []
    for p in range(v["root_passes"]):
        q0 = q.copy()
        if root_xyz is not None and (p == 0 or v["reseed_second"]):
            q0[:3] = root_xyz
        x, st = pg_solve(env, v, kp[0], qs, ks, q0, lb, ub)
        q = writeback(env, v, np.where(qs, x, q0))
        root_iters.append(st["iter_num"])
    res = dict(root_err=st["error"], root_iters=root_iters)
    if not passes:
        return res

    # ---- pose_optimization, first pass (compute_stac.py:170-278) ----
    allq, allk = np.ones(nq, bool), np.ones(3 * K, bool)
    errs, its = [], []
    for f in range(kp.shape[0]):
        x, st = pg_solve(env, v, kp[f], allq, allk, q, lb, ub)
        q = writeback(env, v, x)
        its.append(st["iter_num"])
        for pm in fs.part_masks:
            q0 = q.copy()
            x, st = pg_solve(env, v, kp[f], pm, allk, q0, lb, ub)
            q = writeback(env, v, np.where(pm, x, q0))
        errs.append(st["error"])
    errs = np.asarray(errs, np.float64)
    res.update(mean=float(errs.mean()), std=float(errs.std()), full_iters=its, errs=errs.tolist())
    return res


def score(r):
    """Sum of |log ratio| to the notebook's three numbers (0 = reproduces them)."""
    s = abs(math.log(r["root_err"] / NB_ROOT))
    if "mean" in r:
        s += abs(math.log(r["mean"] / NB_MEAN)) + abs(math.log(max(r["std"], 1e-12) / NB_STD))
    return s


# ---------------------------------------------------------------------------------------------------------------------
def single_variants():
    """One deviation from the oracle at a time."""
    out = [dict()]
    out += [dict(eps=e) for e in (0.0, "f64", 1e-6, 1e-5)]
    out += [dict(eps_side=s) for s in ("lhs", "scaled")]
    out += [dict(ls_cmp=">=")]
    out += [dict(ls_form="classic"), dict(ls_form="classic", eps=0.0)]
    out += [dict(grow="keep"), dict(grow="one"), dict(reset_thresh=1e-3), dict(reset_thresh=0.0)]
    out += [dict(err_at=e) for e in ("x_step", "x_step_scaled", "y_unit")]
    out += [dict(project_x0=True)]
    out += [dict(maxls=m) for m in (14, 16, 30)]
    out += [dict(momentum=m) for m in ("tnext", "k")]
    out += [dict(accel=False)]
    out += [dict(stop_cmp=">=")]
    out += [dict(maxiter=m) for m in (250, 1000)]
    out += [dict(stepsize0=s) for s in (0.5, 2.0)]
    out += [dict(root_quat_bounds="inf")]
    out += [dict(lb_min0=False)]
    out += [dict(tables="legacy")]
    out += [dict(reseed_second=False)]
    out += [dict(root_seed=s) for s in ("idx", "none")]
    out += [dict(root_mask="all")]
    out += [dict(root_passes=p) for p in (1, 3)]
    out += [dict(writeback_norm=False)]
    out += [dict(tol=t) for t in (5e-5, 1e-5)]
    return out


def pair_variants():
    """Older-source host rules combined with each other and with the likeliest solver details."""
    host = [dict(root_quat_bounds="inf"), dict(tables="legacy"), dict(root_quat_bounds="inf", tables="legacy")]
    solver = [dict(), dict(eps=0.0), dict(err_at="x_step"), dict(err_at="y_unit"), dict(grow="keep"), dict(maxiter=250),
              dict(reseed_second=False), dict(root_mask="all"), dict(lb_min0=False), dict(ls_form="classic")]
    out = []
    for h, s in itertools.product(host, solver):
        o = dict(h)
        o.update(s)
        if o not in out and len(o) > 1:
            out.append(o)
    return out


def _work(over):
    try:
        return over, run_variant(over)
    except Exception as ex:  # a variant that diverges (NaN) is a result, not a crash
        return over, dict(root_err=float("nan"), root_iters=[], mean=float("nan"), std=float("nan"), error=repr(ex))


def perturbation_spread(n=8, rel=1e-6):
    """Spread of the baseline under 1e-6 relative perturbations of the keypoints (what 'lands within' means)."""
    env = Env.get("current")
    base = env.kp.copy()
    rng = np.random.default_rng(0)
    rows = []
    for _ in range(n):
        env.kp = (base * (1 + rel * rng.standard_normal(base.shape))).astype(np.float32)
        rows.append(run_variant({}))
    env.kp = base
    return rows


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("--jobs", type=int, default=8)
    ap.add_argument("--out", type=str, default=None)
    ap.add_argument("--quick", action="store_true", help="single-deviation variants only")
    ap.add_argument("--spread", type=int, default=6, help="perturbed baseline runs (0 = skip)")
    args = ap.parse_args(argv)

    variants = single_variants() + ([] if args.quick else pair_variants())
    with ProcessPoolExecutor(max_workers=args.jobs) as ex:
        results = list(ex.map(_work, variants, chunksize=1))

    lines = []
    P = lines.append
    P("jaxopt / host-rule variant sweep against demos/rodent_demo.ipynb cell 6 (frames 0-9, rodent.yaml)")
    P(f"notebook: root residual {NB_ROOT:.4e}   pass-1 mean {NB_MEAN:.4e}   std {NB_STD:.4e}")
    P("")
    P(f"{'root err':>10} {'root it':>9} {'p1 mean':>10} {'p1 std':>10} {'full it':>8} {'score':>6}  variant")
    for over, r in sorted(results, key=lambda t: (score(t[1]) if np.isfinite(t[1]['root_err']) else 1e9)):
        fi = int(np.mean(r.get("full_iters", [0]))) if r.get("full_iters") else 0
        sc = score(r) if np.isfinite(r["root_err"]) else float("nan")
        P(f"{r['root_err']:10.3e} {str(r['root_iters']):>9} {r.get('mean', float('nan')):10.3e} "
          f"{r.get('std', float('nan')):10.3e} {fi:8d} {sc:6.2f}  {describe(over)}")
    if args.spread:
        rows = perturbation_spread(args.spread)
        re_ = [r["root_err"] for r in rows]
        me = [r["mean"] for r in rows]
        sd = [r["std"] for r in rows]
        P("")
        P(f"baseline under 1e-6 relative keypoint perturbations ({len(rows)} runs): root {min(re_):.3e}..{max(re_):.3e}, "
          f"mean {min(me):.3e}..{max(me):.3e}, std {min(sd):.3e}..{max(sd):.3e}")
    text = "\n".join(lines)
    print(text)
    if args.out:
        Path(args.out).parent.mkdir(parents=True, exist_ok=True)
        Path(args.out).write_text(text + "\n")
    return results


if __name__ == "__main__":
    main()
