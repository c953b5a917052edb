"""Process-group glue for the scoring path (one process per GPU, torch.distributed over RCCL).

The reference shards query rows in contiguous blocks and merges the -100-filled [N, N] matrices with
all_reduce(SUM) (retrieval_utils.py:213-215, 233-235, 252-262; backend 'nccl' at util/misc.py:225).
Here each rank contributes only its own [step, N] row block to ONE all_gather per matrix (504 KB per
rank at N=1000, W=8 instead of 4 MB all-reduced), and the result equals the single-process matrix for
any world size.  `compat_offset=True` reproduces the reference's W-dependent offsets instead.
"""
from __future__ import annotations

import os
from typing import Tuple


def _dist():
    import torch.distributed as dist
    return dist


def is_dist_avail_and_initialized() -> bool:           # util/misc.py:170-175
    d = _dist()
    return d.is_available() and d.is_initialized()


def get_world_size() -> int:                           # util/misc.py:178-181
    return _dist().get_world_size() if is_dist_avail_and_initialized() else 1


def get_rank() -> int:                                 # util/misc.py:184-187
    return _dist().get_rank() if is_dist_avail_and_initialized() else 0


def is_main_process() -> bool:
    return get_rank() == 0


def init_distributed_mode(backend: str = None) -> Tuple[int, int, int]:
    """util/misc.py:199-229: env:// rendezvous from RANK / WORLD_SIZE / LOCAL_RANK; returns (rank, world, local_rank).
    backend defaults to 'nccl' (= RCCL on ROCm) when a GPU is visible, 'gloo' otherwise."""
    import datetime
    import torch
    if "RANK" not in os.environ or "WORLD_SIZE" not in os.environ:
        return 0, 1, 0
    rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ.get("LOCAL_RANK", 0))
    if backend is None:
        backend = os.environ.get("BLIM_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
    if "BLIM_FORCE_DEVICE" in os.environ:                 # test aid: several ranks on one GPU (gloo backend only)
        local = int(os.environ["BLIM_FORCE_DEVICE"])
    if backend == "nccl":
        torch.cuda.set_device(local)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29500")
    if not _dist().is_initialized():
        _dist().init_process_group(backend=backend, init_method="env://", world_size=world, rank=rank,
                                   timeout=datetime.timedelta(seconds=7200))
    _dist().barrier()
    return rank, world, local


def force_collective() -> bool:
    """BLIM_FORCE_COLLECTIVE=1 with a process group up: run every collective branch of the multi-GPU flow even at world size 1 -- the merge of the
    score blocks, the prior-vector gather, the clip-feature gather, bench.py's score-row all_gather -- so that a ONE-GPU box executes the same RCCL
    calls (backend 'nccl', device tensors) the 8-GPU job makes, with results that must equal the non-collective path bit for bit
    (tests/test_main_driver.py).  The reference's counterpart: util/misc.py:199-229 + retrieval_utils.py:252-262 run under torchrun even with one process."""
    return os.environ.get("BLIM_FORCE_COLLECTIVE", "0") == "1" and is_dist_avail_and_initialized()


def row_block(n: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous row block of `rank`: step = n // W + 1 (retrieval_utils.py:213-215)."""
    step = n // world + 1
    start = min(n, rank * step)
    return start, min(n, start + step)


def merge_row_blocks(S, block: Tuple[int, int], world: int, compat_offset: bool = False):
    """S: [N, M] f32 tensor whose rows block[0]:block[1] this rank computed (others -100).
    Returns the merged matrix on every rank."""
    import torch
    d = _dist()
    if (world == 1 and not force_collective()) or not is_dist_avail_and_initialized():
        return S
    N, M = S.shape
    step = N // world + 1
    mine = torch.full((step, M), -100.0, dtype=S.dtype, device=S.device)
    s, e = block
    if e > s:
        mine[: e - s] = S[s:e]
    gathered = [torch.empty_like(mine) for _ in range(world)]
    d.all_gather(gathered, mine)
    out = torch.cat(gathered, dim=0)[:N].contiguous()
    if compat_offset:
        # all_reduce(SUM) of W matrices that hold -100 wherever the rank did not compute (retrieval_utils.py:252-262)
        out = torch.where(out == -100.0, out * world, out - 100.0 * (world - 1))
    return out


def merge_row_blocks_many(mats, blocks, world: int, compat_offset: bool = False):
    """Same as merge_row_blocks for several matrices at once: ONE all-gather of the concatenated row blocks (the <= 6 score
    matrices of an evaluation travel together).  mats[i]: [N_i, M_i]; blocks[i]: this rank's row range."""
    import torch
    d = _dist()
    if (world == 1 and not force_collective()) or not is_dist_avail_and_initialized() or not mats:
        return list(mats)
    parts, shapes = [], []
    for S, (s, e) in zip(mats, blocks):
        N, M = S.shape
        step = N // world + 1
        mine = torch.full((step, M), -100.0, dtype=torch.float32, device=S.device)
        if e > s:
            mine[: e - s] = S[s:e]
        parts.append(mine.reshape(-1)); shapes.append((N, M, step))
    flat = torch.cat(parts)
    gathered = [torch.empty_like(flat) for _ in range(world)]
    d.all_gather(gathered, flat)
    outs, off = [], 0
    for (N, M, step) in shapes:
        out = torch.cat([g[off: off + step * M].reshape(step, M) for g in gathered], dim=0)[:N].contiguous()
        if compat_offset:
            out = torch.where(out == -100.0, out * world, out - 100.0 * (world - 1))
        outs.append(out); off += step * M
    return outs


def local_world_size(world: int = 1) -> int:
    """Ranks sharing THIS node's CPUs: LOCAL_WORLD_SIZE (torchrun sets it), else min(world, GPUs of the node)."""
    v = os.environ.get("LOCAL_WORLD_SIZE")
    if v and v.isdigit() and int(v) > 0:
        return int(v)
    try:
        import torch
        n = torch.cuda.device_count()
    except Exception:
        n = 0
    return max(1, min(world, n) if n > 0 else world)


def usable_cpus() -> int:
    """CPUs this process may use: the affinity mask and the cgroup CPU quota (v2: cpu.max; v1: cpu.cfs_quota_us / cpu.cfs_period_us)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    try:
        q = open("/sys/fs/cgroup/cpu.max").read().split()
        if q[0] != "max":
            n = min(n, max(1, int(q[0]) // int(q[1])))
    except (OSError, ValueError, IndexError):
        try:
            quota = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if quota > 0 and period > 0:
                n = min(n, max(1, quota // period))
        except (OSError, ValueError):
            pass
    return max(1, n)


def host_threads(world: int = 1, cap: int = 8) -> int:
    """Intra-op threads for the host-side torch ops of the scoring drivers (top-k of the first-stage matrices, padding, stacking):
    the CPUs this process may use (`usable_cpus`) shared by the ranks of THIS NODE (`local_world_size`: on a multi-node job the global world
    size would throttle every rank to one thread), at most `cap`.  torch's default is one thread per LOGICAL CPU of the machine; inside a
    16-CPU quota on a 256-thread host that made a 64 MB torch.stack take 170 ms (measured on the GPU boxes), and eight ranks each starting 256
    threads is worse."""
    return max(1, min(cap, usable_cpus() // local_world_size(world)))


def limit_host_threads(world: int = 1) -> int:
    import torch
    n = host_threads(world)
    torch.set_num_threads(n)
    return n
