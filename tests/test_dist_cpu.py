"""Multi-process CPU tests (gloo, world_size 2) of the image-parallel plumbing: shard partition, bucketed weight
broadcast (C1) and result gather/merge (C2)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import fie_amd  # noqa: F401
from fie_amd import dist as fdist


def test_shard_is_a_partition():
    items = list(range(23))
    for world in (1, 2, 3, 8):
        parts = [fdist.shard(items, r, world) for r in range(world)]
        assert sorted(sum(parts, [])) == items
        assert max(map(len, parts)) - min(map(len, parts)) <= 1
    assert fdist.shard([], 1, 2) == []


def test_merge_results_orders_rows():
    a = dict(processed=2, skipped=1, failed=0, total_time=1.5, rows=[dict(index=0), dict(index=2)])
    b = dict(processed=1, skipped=0, failed=1, total_time=0.5, rows=[dict(index=1)])
    m = fdist.merge_results([a, b])
    assert (m["processed"], m["skipped"], m["failed"], m["total_time"]) == (3, 1, 1, 2.0)
    assert [r["index"] for r in m["rows"]] == [0, 1, 2]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    r, _, w = fdist.init(backend="gloo")
    assert (r, w) == (rank, world)
    g = torch.Generator().manual_seed(1 if rank == 0 else 99)
    sds = {"unet": {"a.weight": torch.randn(37, 5, generator=g).half(), "b.bias": torch.randn(11, generator=g).half()},
           "vae": {"c.weight": torch.randn(3, 3, 3, 3, generator=g), "d": torch.randn(1000, generator=g).half()}}
    fdist.broadcast_state_dicts(sds, src=0, bucket_bytes=512)
    g0 = torch.Generator().manual_seed(1)
    assert torch.equal(sds["unet"]["a.weight"], torch.randn(37, 5, generator=g0).half())
    assert torch.equal(sds["unet"]["b.bias"], torch.randn(11, generator=g0).half())
    assert torch.equal(sds["vae"]["c.weight"], torch.randn(3, 3, 3, 3, generator=g0))
    entries = [(i, f"id{i}", {}) for i in range(7)]
    mine = fdist.shard(entries, rank, world)
    res = dict(processed=len(mine), skipped=0, failed=rank, total_time=0.25 * len(mine),
               rows=[dict(index=i, image_id=k) for i, k, _ in mine])
    gathered = fdist.gather_results(res)
    if rank == 0:
        m = fdist.merge_results(gathered)
        out.put((m["processed"], m["failed"], [r["index"] for r in m["rows"]]))
    else:
        assert gathered is None
    fdist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_gloo_world2_broadcast_and_gather():
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    got = out.get(timeout=100)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert got == (7, 1, list(range(7)))


def _bcast_worker(rank, world, port, out):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    fdist.init(backend="gloo")
    from fie_amd import stack
    cfgs, sds = stack.broadcast_stack("tiny", True, device="cpu", dtype=torch.float16, seed=77)
    _, ref = stack.synthetic_stack("tiny", True, device="cpu", dtype=torch.float16, seed=77)
    ok = all(torch.equal(sds[k][n], ref[k][n]) for k in ref for n in ref[k])
    out.put((rank, ok, sum(len(v) for v in sds.values())))
    fdist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(180)
def test_gloo_world2_weight_broadcast_of_a_full_stack():
    """C1 end to end on the tiny stack: rank 1 starts from uninitialised buffers and ends bit-identical to rank 0's weights."""
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_bcast_worker, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted(out.get(timeout=150) for _ in range(2))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert [g[:2] for g in got] == [(0, True), (1, True)] and got[0][2] == got[1][2] > 1000
