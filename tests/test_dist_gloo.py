"""N > 1 path on CPU: world_size-2 gloo process group (sharding, gather order, the one all-reduce)."""

import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from conftest import GOLDEN, ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, str(ROOT))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as tdist

    from oracle import Oracle
    from stac_mjx_amd import dist
    from stac_mjx_amd.mjcf import ModelTables

    tdist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        assert dist.world() == (rank, world)
        # shard ranges tile [0, n) contiguously, earlier ranks take the remainder
        for n in (0, 1, 5, 8, 4001):
            los = [dist.shard_range(n, r, world) for r in range(world)]
            assert los[0][0] == 0 and los[-1][1] == n and all(a[1] == b[0] for a, b in zip(los, los[1:]))
        # ragged all-gather keeps clip order
        n_total = 5
        lo, hi = dist.shard_range(n_total)
        local = torch.arange(lo, hi, dtype=torch.float32)[:, None].repeat(1, 3)
        full = dist.all_gather_clips(local, n_total)
        assert torch.equal(full[:, 0], torch.arange(n_total, dtype=torch.float32))
        # rank-0 gather (what Stac uses by default): rank 0 gets every clip in order, the others nothing
        g0 = dist.gather_clips(local, n_total, dst=0)
        assert (g0 is None) if rank else torch.equal(g0, full)
        # more ranks than clips: the last rank owns nothing, the collectives still line up
        lo1, hi1 = dist.shard_range(1)
        one = torch.full((hi1 - lo1, 2), 7.0)
        g1 = dist.gather_clips(one, 1, dst=0)
        assert (g1 is None) if rank else (g1.shape == (1, 2) and float(g1[0, 0]) == 7.0)
        assert dist.all_gather_clips(one, 1).shape == (1, 2)
        # offset phase: per-rank partial sums -> one all-reduce -> same closed form on every rank
        t = ModelTables.load(GOLDEN / "rodent_tables_legacy.npz")
        with np.load(GOLDEN / "demo_viz_golden.npz") as d:
            qpos, kp, off = d["qpos"], d["kp_data"], d["offsets"]
        orc = Oracle(t)
        orc.set_site_pos(off)
        lo, hi = dist.shard_range(50)
        part = torch.as_tensor(orc.m_partial(kp[lo:hi], qpos[lo:hi]))
        for det in (True, False):
            red = dist.all_reduce_partial(part, deterministic=det)
            assert red.shape == (71,) and float(red[70]) == 50.0
            ref = orc.m_partial(kp, qpos)
            np.testing.assert_allclose(red.numpy(), ref, rtol=2e-5, atol=1e-6)
        is_reg = np.zeros((23, 3), np.float32)
        params, err = orc.m_finish(dist.all_reduce_partial(part).numpy(), off, is_reg, 1.0)
        ref_params, _ = orc.m_opt(kp, qpos, off, is_reg, 1.0)
        np.testing.assert_allclose(params, ref_params, atol=1e-6)
        gathered = [None] * world
        tdist.all_gather_object(gathered, params.tobytes())
        assert gathered[0] == gathered[1]  # identical offsets on every rank (bitwise)
        # the sub-record bench.py adds to a multi-GPU line (config.offset_phase_exchange): timing + the bitwise contract
        fin = lambda red: torch.as_tensor(orc.m_finish(red.numpy(), off, is_reg, 1.0)[0])  # noqa: E731
        probe = dist.offset_phase_exchange_probe(part, fin, repeats=3)
        assert probe["world_size"] == world and probe["backend"] == "gloo" and probe["n_floats"] == 71
        assert probe["us_per_exchange"] > 0 and probe["offsets_bitwise_equal_across_ranks"] is True
        # ... and it notices ranks that disagree
        bad = dist.offset_phase_exchange_probe(part, lambda red: fin(red) + float(rank), repeats=1)
        assert bad["offsets_bitwise_equal_across_ranks"] is False
        q.put((rank, "ok"))
    except Exception as e:  # pragma: no cover
        q.put((rank, repr(e)))
    finally:
        tdist.destroy_process_group()


def test_two_rank_gloo_sharding_and_allreduce():
    import oracle

    oracle.build()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(0, "ok"), (1, "ok")], res


def test_single_process_helpers_are_identity():
    from stac_mjx_amd import dist

    assert dist.world() == (0, 1) and dist.shard_range(7) == (0, 7)
    t = torch.arange(6.0)
    assert torch.equal(dist.all_reduce_partial(t), t) and torch.equal(dist.all_gather_clips(t, 6), t)


# ---- per-rank shard files instead of a padded gather (run_stac, stac.gather = none / auto) -----------------------------
def _shard_worker(rank, world, port, tmp, mode, q):
    sys.path.insert(0, str(ROOT))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import json

    import torch.distributed as tdist

    from stac_mjx_amd import dist, io, main
    from stac_mjx_amd.config import validate_config
    from stac_mjx_amd.fit_model import finish_fit_setup
    from stac_mjx_amd.mjcf import ModelTables

    tdist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        mcfg = json.load(open(GOLDEN / "rodent_model_cfg.json"))
        fs = finish_fit_setup(ModelTables.load(GOLDEN / "rodent_tables.npz"), mcfg, list(mcfg["KEYPOINT_MODEL_PAIRS"].keys()))
        F, n_clips = 2, 5  # ragged: rank 0 owns 3 clips, rank 1 owns 2
        stac_cfg = dict(fit_offsets_path="fit.h5", ik_only_path="ik.h5", data_path="d.mat", continuous=False, n_fit_frames=4,
                        skip_fit_offsets=True, skip_ik_only=False, infer_qvels=True, n_frames_per_clip=F,
                        mujoco=dict(solver="newton", iterations=1, ls_iterations=4))
        stac_cfg.update(mode)
        cfg = validate_config({"model": dict(mcfg), "stac": stac_cfg})
        z = lambda *s: np.zeros(s, np.float32)  # noqa: E731
        names = dict(kp_names=fs.kp_names, names_qpos=fs.part_names, names_xpos=fs.tables.body_names)
        if rank == 0:
            io.save_data_to_h5(config=cfg, file_path=f"{tmp}/fit.h5", kp_data=z(4, 69), marker_sites=z(4, 23, 3),
                               offsets=np.full((23, 3), 0.25, np.float32), qpos=z(4, 74), xpos=z(4, 67, 3),
                               xquat=z(4, 67, 4), qvel=np.array([]), **names)
        dist.barrier()
        kp_all = np.arange(n_clips * F * 69, dtype=np.float32).reshape(n_clips * F, 69)

        class ShardStac:  # what Stac.ik_only returns under gather = none: this rank's clips only
            _timestep, _freejoint = 0.002, True

            def __init__(self, xml, cfg, kp_names, **kw):
                self.cfg = cfg
                self.setup = fs

            def ik_only(self, kp, offsets, gather=None):
                mode_now = str(gather or self.cfg.stac.get("gather", "rank0"))
                lo, hi = dist.shard_range(n_clips) if mode_now == "none" else (0, n_clips)
                n = (hi - lo) * F
                qp = z(n, 74)
                qp[:, 3] = 1.0
                qp[:, 0] = np.arange(lo * F, hi * F)  # frame id in the root x coordinate
                return io.StacData(qpos=qp, xpos=z(n, 67, 3), xquat=z(n, 67, 4), marker_sites=z(n, 23, 3),
                                   offsets=np.asarray(offsets), kp_data=kp[lo * F : hi * F], **names)

        main.Stac = ShardStac
        try:
            _, ik_path = main.run_stac(cfg, kp_all, fs.kp_names, base_path=tmp)
            q.put((rank, str(ik_path)))
        except ValueError as e:  # (a refused configuration: reported, both ranks refuse alike)
            q.put((rank, "ValueError: " + str(e)))
    finally:
        tdist.destroy_process_group()


def _run_sharded(tmp_path, mode):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_shard_worker, args=(r, 2, port, str(tmp_path), mode, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = dict(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    return out


def test_run_stac_writes_per_rank_shards_and_a_manifest(tmp_path):
    """stac.gather = none: no rank writes a partial result under the full-run name; every rank writes its clip block
    (ragged: 3 + 2 clips), rank 0 the manifest; io.load_sharded_stac_data gives the frames back in run order."""
    import json

    from stac_mjx_amd import io

    paths = _run_sharded(tmp_path, dict(gather="none"))
    assert paths[0] == paths[1] and paths[0].endswith("ik.manifest.json")
    assert not (tmp_path / "ik.h5").exists() and not (tmp_path / "ik.npz").exists()
    doc = json.loads((tmp_path / "ik.manifest.json").read_text())
    assert [(s["clip_lo"], s["clip_hi"]) for s in doc["shards"]] == [(0, 3), (3, 5)]
    for s in doc["shards"]:
        assert (tmp_path / s["file"]).exists() and f"rank{s['rank']:03d}-of-002" in s["file"]
    cfg, data = io.load_sharded_stac_data(paths[0])
    np.testing.assert_array_equal(data.qpos[:, 0], np.arange(10))
    np.testing.assert_array_equal(data.kp_data, np.arange(10 * 69, dtype=np.float32).reshape(10, 69))
    assert data.qvel.shape == (10, 73) and np.all(data.offsets == 0.25)


def test_run_stac_auto_gather_switches_on_output_size(tmp_path):
    """"auto" (the default): small runs gather to rank 0 and write ONE file; above stac.gather_max_bytes, shards."""
    small = _run_sharded(tmp_path / "a", dict()) if (tmp_path / "a").mkdir() is None else None
    assert small[0].endswith(("ik.h5", "ik.npz")) and not (tmp_path / "a" / "ik.manifest.json").exists()
    big = _run_sharded(tmp_path / "b", dict(gather_max_bytes=1000)) if (tmp_path / "b").mkdir() is None else None
    assert big[0].endswith("ik.manifest.json")


def test_run_stac_auto_gather_keeps_a_continuous_run_on_rank0(tmp_path):
    """"auto" never shards a continuous run (its cross-fades reach across shard borders): even above
    stac.gather_max_bytes it resolves to rank0; an EXPLICIT gather = none on a continuous run is refused on every rank
    before any work is done."""
    import types

    from stac_mjx_amd import main
    from stac_mjx_amd.config import ConfigNode

    stac = types.SimpleNamespace(setup=types.SimpleNamespace(tables=types.SimpleNamespace(nq=74, nbody=67, nsite=23)))
    cfg = ConfigNode({"stac": {"gather": "auto", "gather_max_bytes": 1000}})
    assert main._ik_output_mode(cfg, 10, stac) == "none" and main._ik_output_mode(cfg, 10, stac, continuous=True) == "rank0"
    assert main._ik_output_mode(ConfigNode({"stac": {"gather": "none"}}), 10, stac, continuous=True) == "none"  # explicit stays
    out = _run_sharded(tmp_path, dict(gather="none", continuous=True))
    assert all(v.startswith("ValueError: stac.gather = none") for v in out.values()), out
    assert not list(tmp_path.glob("ik*"))  # refused before anything was written


def _bench_shape_worker(rank, world, port, q):
    sys.path.insert(0, str(ROOT))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as tdist

    import bench
    from stac_mjx_amd import dist

    tdist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        out = {}
        for name, (scaling, frames, F) in {"cfg3": ("strong", 1_000_000, 250), "odd": ("strong", 1001 * 250, 250),
                                           "weak": ("weak", 10_000, 1), "run": ("weak", 100_000, 250)}.items():
            sh = bench.job_shape(scaling, frames, F, rank, world)
            # the rank's block is the one Stac.ik_only takes under the live process group (no explicit rank / world)
            assert (sh["lo"], sh["hi"]) == dist.shard_range(sh["clips_total"])
            # max-over-ranks time, like bench.py: the job's value is the same number on every rank
            t = torch.tensor([1.0 + rank])
            tdist.all_reduce(t, op=tdist.ReduceOp.MAX)
            out[name] = dict(sh, value=bench.job_value(sh, 3, float(t.item())))
            # round 6: what each rank did travels in the line (config.per_rank), next to the rate the committed single-GPU table predicts
            rec = bench.rank_record(rank, sh, (16, 5, 3, 1) if rank == 0 else (32, 3, 2, 9), 10.0 + rank, 1.0 + rank)
            out[name]["per_rank"] = bench.gather_rank_records(rec, tdist)
            out[name]["predicted"] = bench.predicted_value("rodent", sh, world)
        one = bench.job_shape("strong", 250, 250, rank, world)  # more ranks than clips: the last rank owns nothing
        out["one_clip"] = dict(one, per_rank=bench.gather_rank_records(bench.rank_record(rank, one, (32, 3, 2, 9), 0.0, 0.5), tdist))
        q.put((rank, out, None))
    except Exception as exc:  # noqa: BLE001
        q.put((rank, None, repr(exc)))
    finally:
        tdist.destroy_process_group()


def test_bench_strong_and_weak_frame_accounting_two_ranks():
    import bench

    """bench.py's sharding arithmetic under a live world-size-2 group (ADVICE r4: the run-mode line counted frames N times;
    VERDICT r4 #4: a strong-scaling mode on BASELINE configs[3]): the ranks' blocks tile the clips, `frames_total` is the
    job's, `value` = frames_total x steps / the slowest rank's time and is identical on every rank."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_bench_shape_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = dict()
    for _ in ps:
        rank, out, err = q.get(timeout=120)
        assert err is None, err
        res[rank] = out
    for p in ps:
        p.join(timeout=60)
    for name in ("cfg3", "odd", "weak", "run"):
        a, b = res[0][name], res[1][name]
        assert a["lo"] == 0 and a["hi"] == b["lo"] and b["hi"] == a["clips_total"] == b["clips_total"]
        assert a["frames_rank"] + b["frames_rank"] == a["frames_total"] == b["frames_total"]
        assert a["value"] == b["value"] == a["frames_total"] * 3 / 2.0
    assert res[0]["cfg3"]["frames_total"] == 1_000_000 and res[0]["cfg3"]["clips_rank"] == 2000 and res[0]["cfg3"]["scaling"] == "strong"
    assert res[0]["odd"]["clips_rank"] == 501 and res[1]["odd"]["clips_rank"] == 500  # earlier ranks take the remainder
    assert res[0]["weak"]["frames_total"] == 20_000 and res[0]["weak"]["frames_rank"] == res[1]["weak"]["frames_rank"] == 10_000
    assert res[0]["run"]["frames_total"] == 200_000 and res[1]["run"]["clips_rank"] == 400
    # per-rank records: the same list on every rank, in rank order, each rank's own clips / kernel / times
    for name in ("cfg3", "odd", "weak", "run"):
        pr = res[0][name]["per_rank"]
        assert pr == res[1][name]["per_rank"] and [r["rank"] for r in pr] == [0, 1]
        assert [r["clips"] for r in pr] == [res[0][name]["clips_rank"], res[1][name]["clips_rank"]]
        assert pr[0]["kernel"] == "q_phase_kernel<16,5,3,1>" and pr[1]["kernel"] == "q_phase_kernel<32,3,2,9>"
        assert [r["kernel_ms"] for r in pr] == [10.0, 11.0] and [r["elapsed_s"] for r in pr] == [1.0, 2.0]
    assert [r["clips"] for r in res[0]["one_clip"]["per_rank"]] == [1, 0]
    # predicted rate: two ranks of 2 000 clips each run at the table's 2 000-chain rate -> twice that; the weak default twice the
    # 10 000-chain rate; a job whose shape the table does not hold has no prediction
    import json

    rows = {(r["frames_per_clip"], r["chains"]): r["frames_per_s"] for r in json.load(open(ROOT / "profiles" / "single_gpu_rates.json"))["rows"]
            if r["model"] == "rodent"}
    assert res[0]["cfg3"]["predicted"] == res[1]["cfg3"]["predicted"] == pytest.approx(2 * rows[(250, 2000)])
    assert res[0]["weak"]["predicted"] == pytest.approx(2 * rows[(1, 10000)])
    ch = sorted(c for (f, c) in rows if f == 250)
    rate = lambda n: float(np.exp(np.interp(np.log(n), np.log(ch), np.log([rows[(250, c)] for c in ch]))))  # log-linear between the rows
    assert res[0]["run"]["predicted"] == pytest.approx(2 * rate(400), rel=1e-6)  # 400 clips per rank
    # the odd job: the slowest rank has 501 of the 1 001 clips -> the job's frames over that rank's predicted time
    assert res[0]["odd"]["predicted"] == pytest.approx(1001 * 250 / (501 * 250 / rate(501)), rel=1e-6)
    assert bench.predicted_value("rodent", dict(res[0]["cfg3"], F=7), 2) is None
    # the JSON line carries the mode it ran in
    src = (ROOT / "bench.py").read_text()
    assert '"per_rank": per_rank' in src and '"predicted_value"' in src
    assert '"scaling": args.scaling' in src and "BASELINE configs[3]" in src
