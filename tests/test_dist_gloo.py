"""N > 1 path on CPU: world_size-2 gloo process group (sharding, gather order, the one all-reduce)."""

import os
import socket
import sys

import numpy as np
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
