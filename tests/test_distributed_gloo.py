"""CPU, 2 processes over gloo: the row-block all-gather equals the single-process matrix (W=1-equivalent),
and compat mode reproduces the reference's all_reduce(SUM) of -100-filled matrices (retrieval_utils.py:252-262)."""
import os
import socket

import numpy as np
import pytest
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
    # several matrices of different shapes in ONE all-gather
    full2 = torch.from_numpy(np.random.RandomState(1).randn(m, n + 2).astype(np.float32) - 5.0)
    s2, e2 = D.row_block(m, world, rank)
    mine2 = torch.full((m, n + 2), -100.0); mine2[s2:e2] = full2[s2:e2]
    many = D.merge_row_blocks_many([mine.clone(), mine2.clone()], [(s, e), (s2, e2)], world)
    many_c = D.merge_row_blocks_many([mine.clone(), mine2.clone()], [(s, e), (s2, e2)], world, compat_offset=True)
    ref2 = mine2.clone(); dist.all_reduce(ref2, op=dist.ReduceOp.SUM)
    ok_many = torch.equal(many[0], full) and torch.equal(many[1], full2) and torch.allclose(many_c[0], ref, atol=1e-4) and torch.allclose(many_c[1], ref2, atol=1e-4)
    out_q.put((rank, torch.equal(merged, full) and ok_many, torch.allclose(compat, ref, atol=1e-4), D.get_world_size(), D.get_rank()))
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


# ----------------------------------------------------------------------------- evaluation() control flow at W = 2
class _FakeScorer:
    """Deterministic stand-in for PairScorer: a score is a pure function of (pass kind, video, text), like the engine's."""

    def __init__(self):
        self.calls = []

    def _f(self, pairs, salt, video_matters=True):
        p = np.asarray(pairs, dtype=np.float64)
        v = p[:, 0] if video_matters else 0.0
        return (-(1.0 + salt) - 0.37 * np.sin(1.3 * v + 0.7 * p[:, 1] + salt) - 0.01 * p[:, 1]).astype(np.float32)

    def vtg(self, pairs, cpn=False):
        self.calls.append(("vtg", bool(cpn), len(pairs)))
        return self._f(pairs, 5.0 if cpn else 0.0, video_matters=not cpn)

    def tvg(self, pairs, cpn=False):
        self.calls.append(("tvg", bool(cpn), len(pairs)))
        return self._f(pairs, 9.0 if cpn else 2.0)


class _Loader:
    def __init__(self, n, bs=4):
        import types
        self.n, self.bs = n, bs
        self.dataset = types.SimpleNamespace(video_vocab=torch.zeros(n, 4, 8), tvg_prefix_length=3)

    def __iter__(self):
        one = lambda: torch.ones(3, dtype=torch.long)
        for s in range(0, self.n, self.bs):
            k = min(self.bs, self.n - s)
            yield {"video": [torch.zeros(4, 2, 8) for _ in range(k)], "vtg_ids": [one() for _ in range(k)], "vtg_labels": [one() for _ in range(k)],
                   "vtg_masks": [one() for _ in range(k)], "tvg_ids": [one() for _ in range(k)], "tvg_labels": [one() for _ in range(k)],
                   "tvg_masks": [one() for _ in range(k)], "tvg_video_labels": torch.arange(s, s + k)}


def _eval_args(n, scorer):
    import types
    rs = np.random.RandomState(3)
    sims = rs.randn(n, n).astype(np.float32) + 3 * np.eye(n, dtype=np.float32)
    return types.SimpleNamespace(topk=3, batch_size_eval=4, num_clips=4, cpn=True, resume="ckpt", eval=True, dataset="MSRVTT",
                                 iv2_scores={"v2t": torch.from_numpy(sims), "t2v": torch.from_numpy(sims.T.copy())}, _scorer=scorer)


def _run_eval(n):
    import types
    from blim_amd import retrieval_utils as RU
    scorer = _FakeScorer()
    model = types.SimpleNamespace(eval=lambda: None, module=types.SimpleNamespace(set_tvg_prefix_length=lambda k: None))
    tok = types.SimpleNamespace(pad_token_id=0)
    t2v, v2t = RU.evaluation(model, _Loader(n), torch.device("cpu"), tok, _eval_args(n, scorer))
    return t2v, v2t, scorer.calls


def _eval_worker(rank, world, port, n, out_q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    D.init_distributed_mode(backend="gloo")
    t2v, v2t, calls = _run_eval(n)
    out_q.put((rank, {k: v for k, v in t2v.items()}, {k: v for k, v in v2t.items()}, calls))
    dist.barrier()
    dist.destroy_process_group()


def test_evaluation_world2_equals_single_process_and_shards_the_prior_over_texts():
    n = 11
    ref_t2v, ref_v2t, ref_calls = _run_eval(n)                       # W = 1 in this process
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_eval_worker, args=(r, 2, port, n, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, t2v, v2t, calls in res:
        for k in ref_t2v:
            assert np.array_equal(t2v[k], ref_t2v[k]), ("t2v", k, rank)
        for k in ref_v2t:
            assert np.array_equal(v2t[k], ref_v2t[k]), ("v2t", k, rank)
        # the v2t prior pass scored this rank's TEXT block once (6 and 5 texts), not rows x topk pairs
        prior_calls = [c for c in calls if c[0] == "vtg" and c[1]]
        assert prior_calls == [("vtg", True, 6 if rank == 0 else 5)]
    assert ("vtg", True, n) in ref_calls                               # single process: the prior of every text, once
    # pooled pairs: one VTG call and one TVG call over the union of both directions' pairs (41 of 2 x 33 requests here)
    assert [c[:2] for c in ref_calls] == [("vtg", False), ("tvg", False), ("vtg", True), ("tvg", True)] and ref_calls[0][2] == ref_calls[1][2] < 2 * n * 3


@pytest.mark.parametrize("n,world", [(11, 8), (19, 4)], ids=["11-items-8-ranks", "19-items-4-ranks"])
def test_evaluation_with_more_ranks_than_the_row_blocks_fill(n, world):
    """step = N // W + 1 (retrieval_utils.py:213-215) leaves the last ranks short or EMPTY (N = 11, W = 8: blocks of 2 2 2 2 2 1 0 0 rows): every rank still
    joins every collective and ends with the single-process matrices, bit for bit."""
    ref_t2v, ref_v2t, _ = _run_eval(n)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_eval_worker, args=(r, world, port, n, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(r[0] for r in res) == list(range(world))
    for rank, t2v, v2t, calls in res:
        for k in ref_t2v:
            assert np.array_equal(t2v[k], ref_t2v[k]), ("t2v", k, rank)
        for k in ref_v2t:
            assert np.array_equal(v2t[k], ref_v2t[k]), ("v2t", k, rank)


def _run_eval_with(n, **kw):
    import types
    from blim_amd import retrieval_utils as RU
    scorer = _FakeScorer()
    args = _eval_args(n, scorer)
    for k, v in kw.items():
        setattr(args, k, v)
    model = types.SimpleNamespace(eval=lambda: None, module=types.SimpleNamespace(set_tvg_prefix_length=lambda k: None))
    t2v, v2t = RU.evaluation(model, _Loader(n), torch.device("cpu"), types.SimpleNamespace(pad_token_id=0), args)
    return t2v, v2t, scorer.calls, args._eval_stats


def test_cross_direction_dedup_gives_the_same_matrices_with_fewer_scored_pairs():
    """v2t.candidate_likelihood[j, i] and t2v.query_likelihood[i, j] are the same number (likewise the two TVG matrices): the t2v
    passes copy what the v2t matrices hold and score only the rest; with dense candidates (topk >= N) they score nothing."""
    n = 11
    a, b = _run_eval_with(n), _run_eval_with(n, dedup=False)
    for x, y in ((a[0], b[0]), (a[1], b[1])):
        assert set(x) == set(y)
        for k in x:
            assert np.array_equal(x[k], y[k]), k
    assert b[3]["pairs_scored"] == b[3]["pairs_requested"] == 6 * n * 3
    assert a[3]["pairs_requested"] == 6 * n * 3 and a[3]["pairs_scored"] < b[3]["pairs_scored"]
    dense = _run_eval_with(n, topk=n)
    assert [c for c in dense[2]] == [("vtg", False, n * n), ("tvg", False, n * n), ("vtg", True, n), ("tvg", True, n * n)]   # every pair once
    ref = _run_eval_with(n, topk=n, dedup=False)
    for k in ref[0]:
        assert np.array_equal(dense[0][k], ref[0][k]), k
    # shard emulation: rank 1 of 2 scores only the pairs it owns (VTG: its videos, TVG: its texts); whatever it holds equals the full result,
    # and the VTG matrix of its video rows is complete
    sh = _run_eval_with(n, shard=(2, 1))
    s, e = D.row_block(n, 2, 1)
    for full_d, part_d in ((a[0], sh[0]), (a[1], sh[1])):
        for k in full_d:
            if k != "internvideo2":
                m = part_d[k] != -100.0
                assert m.any() and np.array_equal(part_d[k][m], full_d[k][m]), k
    assert np.array_equal(sh[1]["candidate_likelihood"][s:e], a[1]["candidate_likelihood"][s:e])
    assert sh[3]["pairs_scored"] < a[3]["pairs_scored"]


def test_pair_ownership_with_unequal_numbers_of_videos_and_texts():
    """Nv != Nt (DiDeMo / ActivityNet style): pooled pair ownership and the reference's six row-sharded passes give the same matrices, also for
    an emulated rank whose blocks are cut differently on the two axes."""
    import types
    from blim_amd import retrieval_utils as RU
    nv, nt = 9, 13
    rs = np.random.RandomState(11)
    sims = rs.randn(nv, nt).astype(np.float32)

    def run(**kw):
        scorer = _FakeScorer()
        args = types.SimpleNamespace(topk=4, batch_size_eval=4, num_clips=4, cpn=True, resume="ckpt", eval=True, dataset="MSRVTT",
                                     iv2_scores={"v2t": torch.from_numpy(sims), "t2v": torch.from_numpy(sims.T.copy())}, _scorer=scorer, **kw)
        model = types.SimpleNamespace(eval=lambda: None, module=types.SimpleNamespace(set_tvg_prefix_length=lambda k: None))
        t2v, v2t = RU.evaluation(model, _Loader(max(nv, nt)), torch.device("cpu"), types.SimpleNamespace(pad_token_id=0), args)
        return t2v, v2t, args._eval_stats

    a, b = run(), run(dedup=False)
    for x, y in ((a[0], b[0]), (a[1], b[1])):
        for k in x:
            assert x[k].shape == y[k].shape and np.array_equal(x[k], y[k]), k
    assert a[1]["candidate_likelihood"].shape == (nv, nt) and a[0]["query_likelihood"].shape == (nt, nv)
    assert a[2]["pairs_scored"] < b[2]["pairs_scored"]
    sh = run(shard=(3, 2))
    for full_d, part_d in ((a[0], sh[0]), (a[1], sh[1])):
        for k in full_d:
            if k != "internvideo2":
                m = part_d[k] != -100.0
                assert np.array_equal(part_d[k][m], full_d[k][m]), k


def _forced_worker(port, n, out_q):
    """World size 1 with a process group up and BLIM_FORCE_COLLECTIVE=1: the collective branches run (what the -m gpu tests do over RCCL on a one-GPU box)."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", BLIM_FORCE_COLLECTIVE="1")
    D.init_distributed_mode(backend="gloo")
    assert D.force_collective()
    calls = []
    orig = dist.all_gather
    dist.all_gather = lambda *a, **k: (calls.append(1), orig(*a, **k))[1]
    t2v, v2t, _ = _run_eval(n)
    full = torch.from_numpy(np.random.RandomState(0).randn(n, 5).astype(np.float32) - 5.0)
    merged = D.merge_row_blocks(full.clone(), (0, n), 1)
    out_q.put(({k: v for k, v in t2v.items()}, {k: v for k, v in v2t.items()}, len(calls), torch.equal(merged, full)))
    dist.barrier()
    dist.destroy_process_group()


def test_forced_collectives_at_world_size_one_change_no_bit():
    """distributed.force_collective: at W = 1 every collective of the evaluation runs (the score-block all-gather, the text-sharded prior's gather) and
    the matrices equal the run without a process group bit for bit."""
    n = 11
    assert not D.force_collective()                                  # no process group in this process
    ref_t2v, ref_v2t, _ = _run_eval(n)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_forced_worker, args=(_free_port(), n, q))
    p.start()
    t2v, v2t, n_gathers, merged_ok = q.get(timeout=180)
    p.join(timeout=60)
    assert p.exitcode == 0 and merged_ok
    assert n_gathers >= 3                                            # merge of the blocks + the prior vector + merge_row_blocks above
    for k in ref_t2v:
        assert np.array_equal(t2v[k], ref_t2v[k]), ("t2v", k)
    for k in ref_v2t:
        assert np.array_equal(v2t[k], ref_v2t[k]), ("v2t", k)


# ----------------------------------------------------------------------------- sharded calibration sample (`--vtg_precise` / `--tvg_precise auto` at W > 1)
def _cal_worker(rank, world, port, out_q):
    import types
    from blim_amd import retrieval_utils as RU
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    D.init_distributed_mode(backend="gloo")
    pairs = np.stack([np.repeat(np.arange(9), 5), np.arange(45) % 7], axis=1)           # 9 queries x 5 candidates
    mine, share = RU.PairScorer._my_block(pairs, (world, rank))
    dev_all = np.abs(np.random.RandomState(3).randn(len(pairs))) * 1e-4                  # the deviation "of pair k" (what a rank would measure on its block)
    k0 = int(np.nonzero((pairs == mine[0]).all(axis=1))[0][0])
    got = RU.PairScorer._gather_dev(types.SimpleNamespace(device=torch.device("cpu")), dev_all[k0:k0 + len(mine)], share)
    out_q.put((rank, len(mine), sorted(got.tolist()) == sorted(dev_all.tolist()), RU.predicted_max_deviation(got, 48000)))
    dist.barrier()
    dist.destroy_process_group()


def test_calibration_sample_is_split_over_the_ranks_and_gathered():
    """PairScorer.calibrate_*'s `share`: each rank scores one contiguous block of the sample; every rank then holds ALL deviations (same decision everywhere)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_cal_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert [r[1] for r in res] == [23, 22] and all(r[2] for r in res) and res[0][3] == res[1][3]
    from blim_amd import retrieval_utils as RU
    pairs = np.zeros((5, 2), np.int64)
    assert RU.PairScorer._my_block(pairs, None)[1] is None and RU.PairScorer._my_block(pairs, (1, 0))[1] is None      # no process group: the whole sample, no gather
