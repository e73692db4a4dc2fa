"""Image-parallel execution of the PIE-Bench batch over the GPUs of one node (new capability -- the
reference's run_batch.py:176-219 is a serial single-GPU loop).

One process per GPU (`torch.distributed`, backend "nccl" = RCCL over xGMI; "gloo" in the CPU tests).  Each
`(image, prompt)` edit is independent, so the data path has NO collective: rank r takes entries r, r+W, r+2W, ...
of the filtered list.  Two collectives exist outside the hot loop:
  C1  broadcast_state_dicts  rank 0's fp16 weights -> all ranks, as a few large flat buckets (ring broadcast is bound by
                             one ~153 GB/s xGMI link: bucket size only has to amortise launch latency)
  C2  gather_results         per-rank (processed, skipped, failed, total_time, rows) -> rank 0 for the reference's summary
"""
import os

import torch
import torch.distributed as dist


def env_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def init(backend=None):
    """Initialise the default process group from the launcher's env (no-op for a single process)."""
    rank, local, world = env_world()
    if world > 1 and torch.cuda.is_available():
        # every launch of this process (torch's and the HIP library's) targets its GPU; a rehearsal with more ranks than
        # GPUs (gloo on a one-GPU box) folds the ranks onto the devices that exist
        torch.cuda.set_device(local % torch.cuda.device_count())
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        backend = backend or os.environ.get("FIE_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        kw = {"device_id": torch.device("cuda", local)} if backend == "nccl" else {}
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return rank, local, world


def shard(entries, rank, world):
    """Static round-robin over the filtered entry list (balances PIE-Bench's 10 category directories)."""
    return list(entries)[rank::world]


def broadcast_state_dicts(sds, src=0, device=None, bucket_bytes=1 << 30):
    """C1.  `sds`: {component: {name: tensor}} with identical keys/shapes/dtypes on every rank (non-src ranks pass
    tensors of the right shape whose values are overwritten).  Tensors are packed into flat buckets of up to
    `bucket_bytes` so the broadcast is a handful of large messages instead of thousands of small ones."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return sds
    flat = [(c, n) for c in sorted(sds) for n in sorted(sds[c])]
    bucket, size = [], 0

    def flush():
        nonlocal bucket, size
        if not bucket:
            return
        dt = sds[bucket[0][0]][bucket[0][1]].dtype
        buf = torch.cat([sds[c][n].reshape(-1).to(device or sds[c][n].device) for c, n in bucket])
        if dist.get_backend() == "gloo" and buf.is_cuda:      # gloo moves host memory: stage through the CPU
            host = buf.cpu()
            dist.broadcast(host, src=src)
            buf.copy_(host)
        else:
            dist.broadcast(buf, src=src)
        off = 0
        for c, n in bucket:
            t = sds[c][n]
            sds[c][n] = buf[off:off + t.numel()].view(t.shape).to(t.device, dt).clone()
            off += t.numel()
        bucket, size = [], 0

    for c, n in flat:
        t = sds[c][n]
        if bucket and (t.dtype != sds[bucket[0][0]][bucket[0][1]].dtype or size + t.numel() * t.element_size() > bucket_bytes):
            flush()
        bucket.append((c, n))
        size += t.numel() * t.element_size()
    flush()
    return sds


def gather_results(local_result, dst=0):
    """C2.  Gathers one picklable dict per rank on `dst` (list on dst, None elsewhere)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return [local_result]
    out = [None] * dist.get_world_size() if dist.get_rank() == dst else None
    dist.gather_object(local_result, out, dst=dst)
    return out


def merge_results(results):
    """Sum the counters and concatenate the per-image rows (sorted back into entry order)."""
    tot = dict(processed=0, skipped=0, failed=0, total_time=0.0, rows=[])
    for r in results:
        for k in ("processed", "skipped", "failed", "total_time"):
            tot[k] += r[k]
        tot["rows"] += r["rows"]
    tot["rows"].sort(key=lambda row: row["index"])
    return tot


def barrier():
    if dist.is_initialized():
        dist.barrier()
