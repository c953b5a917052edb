"""CPU, 2 processes over gloo: the row-block all-gather equals the single-process matrix (W=1-equivalent),
and compat mode reproduces the reference's all_reduce(SUM) of -100-filled matrices (retrieval_utils.py:252-262)."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from blim_amd import distributed as D


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, n, m, out_q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    D.init_distributed_mode(backend="gloo")
    full = torch.from_numpy(np.random.RandomState(0).randn(n, m).astype(np.float32) - 5.0)
    s, e = D.row_block(n, world, rank)
    mine = torch.full((n, m), -100.0)
    mine[s:e] = full[s:e]
    merged = D.merge_row_blocks(mine.clone(), (s, e), world)
    compat = D.merge_row_blocks(mine.clone(), (s, e), world, compat_offset=True)
    ref = mine.clone()
    dist.all_reduce(ref, op=dist.ReduceOp.SUM)          # what the reference does
    out_q.put((rank, torch.equal(merged, full), torch.allclose(compat, ref, atol=1e-4), D.get_world_size(), D.get_rank()))
    dist.barrier()
    dist.destroy_process_group()


def test_all_gather_row_blocks_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, 11, 7, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, eq, compat_ok, w, r in res:
        assert eq and compat_ok and w == 2 and r == rank
