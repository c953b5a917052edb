"""CPU: host-side mirror (padding, recall, ensemble, sharding, packing plans) and the C-ABI library's symbols."""
import collections
import os
import types

import numpy as np
import pytest
import torch

from blim_amd import distributed as D
from blim_amd import engine as eng
from blim_amd import retrieval_utils as RU
from blim_amd import synth
from blim_amd import training_utils as TU
from oracle import blim_oracle as O

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def test_library_exports_every_declared_symbol():
    lib = eng.load_library()
    syms = eng.declared_symbols()
    assert len(syms) >= 25
    missing = [s for s in syms if not hasattr(lib, s)]
    assert not missing, missing
    assert lib.blim_abi_version() == 9          # v9: blim_gemm_f16_lo6 + blim_f6_tiles_bytes, options precise_lo6 / masked_query_zero in, precise_qk / precise_act / precise_lo8 out; v8: blim_load_adapter (LoRA adapters kept apart); v7: blim_batch.own_start (segmented sequences); v6: blim_train_step (one merged pass); v5: blim_train_*; v4: blim_vision_*, option "precise"
    assert lib.blim_timing_num_classes() >= 8


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the no-GPU failure mode")
def test_engine_fails_loudly_without_gpu():
    with pytest.raises(eng.BlimError):
        eng.Engine(synth.ModelDims(num_layers=1))
    # and at the C level
    lib = eng.load_library()
    import ctypes as C
    cfg = eng.Config(152064, 3584, 18944, 1, 28, 4, 1024, 4, 128, 1, 1e-6, 1e6)
    h = C.c_void_p()
    assert lib.blim_create(C.byref(cfg), C.byref(h)) < 0
    assert b"no HIP device" in lib.blim_last_error() or b"failed" in lib.blim_last_error()


def test_padding_ids_matches_golden():
    g = np.load(os.path.join(GOLD, "tiny.npz"))
    from oracle.gen_golden import CASES
    spec = CASES["tiny"]
    prob = synth.make_problem(spec["pseed"], spec["n"], synth.ModelDims(**spec["dims"]), tok_per_clip=spec["tok_per_clip"], text_len=spec["text_len"])
    tok = types.SimpleNamespace(pad_token_id=synth.PAD_ID)
    T = lambda rows: [torch.from_numpy(r) for r in rows]
    ids, lab, msk = RU.padding_ids(T(prob.vtg_ids), T(prob.vtg_labels), T(prob.vtg_masks), tok)
    assert np.array_equal(ids.numpy(), g["pad_vtg_ids"]) and np.array_equal(lab.numpy(), g["pad_vtg_labels"]) and np.array_equal(msk.numpy(), g["pad_vtg_masks"])
    # ragged / single-row edge cases
    ids, lab, msk = RU.padding_ids([torch.tensor([5])], [torch.tensor([-100])], [torch.tensor([1])], tok)
    assert ids.shape == (1, 1)


def test_get_recall_matches_golden():
    g = np.load(os.path.join(GOLD, "tiny.npz"))
    ids = {i: i for i in range(50)}
    rec = TU.get_recall(g["recall_t2v"], g["recall_v2t"], ids, ids)
    assert [rec[k] for k in g["recall_keys"]] == list(g["recall_vals"])
    bz = g["recall_v2t"].copy(); bz[3, 4] = 0.0
    rec0 = TU.get_recall(g["recall_t2v"], bz, ids, ids)
    assert [rec0[k] for k in g["recall_keys"]] == list(g["recall_zero_vals"])
    assert rec0["v2t_r1"] == 0.0 and rec0["t2v_r1"] == rec["t2v_r1"]


def test_combine_matches_oracle():
    rs = np.random.RandomState(0)
    n = 12
    mk = lambda: (rs.randn(n, n) - 5).astype(np.float32)
    t2v = {k: mk() for k in ("candidate_likelihood", "candidate_prior", "query_likelihood", "internvideo2")}
    v2t = {k: mk() for k in ("candidate_likelihood", "candidate_prior", "query_likelihood", "internvideo2")}
    for finetuned in (True, False):
        args = types.SimpleNamespace(cpn=True, alpha=[0.4, 0.8], c=[0.3, 0.6, 0.9, 0.7], resume="x" if finetuned else "", eval=True)
        res = TU.combine_and_rank(t2v, v2t, args, n)
        cpn_t2v, cpn_v2t, bt, bv = O.combine_scores(t2v, v2t, args.alpha, args.c, True, finetuned)
        assert res["blim"] == O.get_recall(bt, bv)
        assert res["cpn_candidate_likelihood"] == O.get_recall(cpn_t2v, cpn_v2t)
        if not finetuned:
            assert res["cpn_candidate_likelihood"]["t2v_r1"] == 0.0   # np.zeros placeholder -> zero sentinel


def test_row_block_is_the_reference_partition():
    for n, w in ((1000, 8), (1000, 1), (7, 8), (4917, 8), (16, 4)):
        blocks = [D.row_block(n, w, r) for r in range(w)]
        step = n // w + 1
        assert blocks[0] == (0, min(n, step))
        covered = sorted(i for s, e in blocks for i in range(s, e))
        assert covered == list(range(n))
    assert D.row_block(1000, 8, 7) == (882, 1000) and D.row_block(1000, 8, 0) == (0, 126)


def test_packed_batch_blocks():
    b = eng.PackedBatch(np.arange(100), np.ones(100), np.array([0, 40, 45]), np.array([40, 5, 55]), np.array([0, 0, 0]), np.array([0, 40, 40]), device="cpu")
    assert b.n_blocks == 2 + 1 + 2
    assert b.blk_seq.tolist() == [0, 0, 1, 2, 2] and b.blk_q0.tolist() == [0, 32, 0, 0, 32]
    st = b.struct()
    assert st.n_tokens == 100 and st.n_seqs == 3


class _FakeModel:
    """Planning needs only .project / .device / .dims / .tvg_prefix_length."""

    def __init__(self, dims, tp):
        self.dims, self.device, self.tvg_prefix_length, self.engine, self.dtype = dims, torch.device("cpu"), tp, None, torch.float16

    def project(self, feat, tvg, cache=True):
        clips, T, _ = feat.shape
        return torch.zeros((clips if tvg else clips * T, self.dims.hidden_size), dtype=torch.float16)


def _scorer(n=6, layout=True, limit=None):
    dims = synth.ModelDims(vocab_size=151700, hidden_size=256, intermediate_size=512, num_layers=2, num_heads=2, num_kv_heads=1, mm_hidden_size=64)
    prob = synth.make_problem(5, n, dims, tok_per_clip=8, text_len=(3, 9), reference_layout=layout)
    tok = types.SimpleNamespace(pad_token_id=synth.PAD_ID)
    T = lambda rows: [torch.from_numpy(r) for r in rows]
    vtg = RU.padding_ids(T(prob.vtg_ids), T(prob.vtg_labels), T(prob.vtg_masks), tok)
    tvg = RU.padding_ids(T(prob.tvg_ids), T(prob.tvg_labels), T(prob.tvg_masks), tok)
    fake = _FakeModel(dims, prob.tvg_prefix_length)
    fake.tokenizer_model_max_length = limit
    sc = RU.PairScorer(fake, vtg[0], vtg[2], vtg[1], tvg[0], tvg[2], tvg[1],
                       [torch.from_numpy(v) for v in prob.video], None, torch.from_numpy(prob.tvg_video_labels), dims.num_clips, max_tokens=4096)
    return sc, prob


def test_vtg_plan_shares_the_video_prefix():
    sc, prob = _scorer()
    pairs = np.array([[0, 0], [0, 1], [0, 2], [3, 1]])
    (plan,) = sc.plan_vtg(pairs)
    nv = 4 * 8
    pre_post = len(prob.vtg_ids[0]) - 1 - (len(prob.vtg_labels[0]) - int((prob.vtg_labels[0] == -100).sum()))
    resp = [int((prob.vtg_labels[i] != -100).sum()) for i in range(6)]
    expect = 2 * (pre_post + nv) + (resp[0] - 1) + (resp[1] - 1) + (resp[2] - 1) + (resp[1] - 1)
    assert plan.n_tokens == expect                       # two prefixes (videos 0 and 3), four suffixes
    assert plan.n_pairs == 4 and plan.n_rows == resp[0] + resp[1] + resp[2] + resp[1]
    assert plan.batch.n_seqs == 2 + 4
    pl = plan.batch.pfx_len.numpy()
    assert sorted(pl.tolist()) == [0, 0] + [pre_post + nv] * 4
    # labels are the response tokens, rows start at the last prefix token
    rows = plan.rows.numpy(); rs = plan.row_start.numpy()
    first_rows = rows[rs[:-1]]
    assert set(first_rows.tolist()) <= {pre_post + nv - 1, plan.batch.seq_start.numpy()[np.nonzero(pl == 0)[0][1]] + pre_post + nv - 1}


def test_vtg_plan_cuts_rows_at_tokenizer_model_max_length():
    """modeling_videochat_flash.py:452-457 in the fused planner: a spliced row longer than the limit loses its last response tokens (labels and
    body alike); shorter rows are untouched; a limit that leaves no response token, or cuts a TVG row, is refused."""
    sc0, prob = _scorer()
    full = [len(prob.vtg_ids[i]) - 1 + 4 * 8 for i in range(6)]                    # spliced row lengths: 63 .. 69
    resp = [int((prob.vtg_labels[i] != -100).sum()) for i in range(6)]
    limit = 64
    sc, _ = _scorer(limit=limit)
    pairs = np.array([[i, i] for i in range(6)])
    (p0,), (p1,) = sc0.plan_vtg(pairs), sc.plan_vtg(pairs)
    kept = [r - max(0, f - limit) for r, f in zip(resp, full)]
    assert kept != resp and min(kept) >= 1
    assert p0.n_rows == sum(resp) and p1.n_rows == sum(kept)
    assert p0.n_tokens - p1.n_tokens == sum(resp) - sum(kept)
    rs0, rs1 = p0.row_start.numpy(), p1.row_start.numpy()
    l0, l1 = p0.labels.numpy(), p1.labels.numpy()
    for k in range(6):                                                              # plans keep the pair order of the request here (one video per pair)
        assert np.array_equal(l1[rs1[k]:rs1[k + 1]], l0[rs0[k]:rs0[k] + kept[k]])
    (c1,) = sc.plan_vtg(pairs, cpn=True)                                            # the prior's rows are cut at the same place
    assert c1.n_rows == sum(kept)
    with pytest.raises(ValueError, match="leaves no response token"):
        _scorer(limit=min(full) - max(resp))[0].plan_vtg(pairs)
    with pytest.raises(ValueError, match="TVG row"):
        _scorer(limit=30)


def test_vtg_cpn_plan_scores_each_text_once():
    sc, prob = _scorer()
    pairs = np.array([[0, 1], [2, 1], [3, 1], [4, 5]])
    (plan,) = sc.plan_vtg(pairs, cpn=True)
    assert plan.n_pairs == 2                            # texts 1 and 5
    fan = sorted(len(o) for o in plan.out_index)
    assert fan == [1, 3]
    # the (masked) video tokens are not even packed; positions skip over them
    pos = plan.batch.positions.numpy()
    assert pos.max() >= 4 * 8 + 14


def test_tvg_plan_needs_three_tokens_per_pair():
    sc, prob = _scorer()
    pairs = np.array([[0, 2], [1, 2], [5, 2]])
    (plan,) = sc.plan_tvg(pairs)
    prompt = len(prob.tvg_ids[2]) - 3
    assert plan.n_tokens == prompt + 3 * 3 and plan.n_rows == 12 and plan.n_pairs == 3
    (planc,) = sc.plan_tvg(np.array([[0, 2], [1, 2], [0, 3]]), cpn=True)
    # prior = f(prompt length, video): texts 2 and 3 differ in length here, so 3 scored pairs; prefix = tvg_prefix_length tokens
    assert planc.n_pairs == len({(len(prob.tvg_ids[i]), j) for j, i in [(0, 2), (1, 2), (0, 3)]})
    assert planc.batch.seq_len.numpy()[0] == prob.tvg_prefix_length


def test_tvg_plan_packs_the_candidates_of_a_text_into_segmented_sequences():
    """100 candidate videos of one text: ONE prompt sequence + merged sequences of at most 256 own tokens whose 3-token segments carry their own
    first-key index (blim_batch.own_start); the rows of pair m are the prompt's last token + its three clip tokens."""
    sc, prob = _scorer(n=100)
    pairs = np.array([[j, 2] for j in range(100)])
    (plan,) = sc.plan_tvg(pairs)
    b = plan.batch
    prompt = len(prob.tvg_ids[2]) - 3
    assert plan.n_tokens == prompt + 300 and plan.n_pairs == 100 and b.n_seqs == 3            # prompt + 85 x 3 + 15 x 3 tokens
    sl = sorted(b.seq_len.numpy().tolist())
    assert sl == sorted([prompt, 255, 45])
    own = b.own_start.numpy()
    assert np.array_equal(own[:prompt], np.zeros(prompt, np.int32))
    assert np.array_equal(own[prompt:prompt + 255], np.repeat(np.arange(85) * 3, 3)) and np.array_equal(own[prompt + 255:], np.repeat(np.arange(15) * 3, 3))
    assert np.array_equal(b.pfx_len.numpy()[1:], [prompt, prompt]) and np.array_equal(b.pfx_start.numpy()[1:], [0, 0])
    rows = plan.rows.numpy().reshape(100, 4)
    assert (rows[:, 0] == prompt - 1).all() and np.array_equal(rows[:, 1:].reshape(-1), prompt + np.arange(300))
    pos = b.positions.numpy()
    assert np.array_equal(pos[prompt:], np.tile(prompt + np.arange(3), 100))
    # a plan with one candidate per text has no segments at all: plain causal sequences, own_start stays NULL
    (plain,) = sc.plan_tvg(np.array([[0, 1], [1, 2]]))
    assert plain.batch.own_start is None


def test_several_passes_planned_into_the_same_engine_calls():
    """iter_vtg_jobs / iter_tvg_jobs: a likelihood pass and a prior in ONE plan -- the same sequences as the two separate plans, back to back (token counts, positions,
    key visibility add up; output slots of the second job follow the first's); a job list that does not fit one call splits and still covers every slot once."""
    sc, prob = _scorer()
    for kind, first, second in (("vtg", np.array([[0, 0], [0, 1], [3, 1], [2, 4]]), np.array([[0, 0], [0, 1], [0, 4]])),
                                ("tvg", np.array([[0, 2], [1, 2], [5, 2], [4, 3]]), np.array([[0, 2], [1, 2], [0, 3]]))):
        plan_of = sc.plan_vtg if kind == "vtg" else sc.plan_tvg
        jobs_of = sc.iter_vtg_jobs if kind == "vtg" else sc.iter_tvg_jobs
        (a,), (b,) = plan_of(first, False), plan_of(second, True)
        (m,) = list(jobs_of([(first, False), (second, True)]))
        assert m.n_tokens == a.n_tokens + b.n_tokens and m.n_rows == a.n_rows + b.n_rows and m.n_pairs == a.n_pairs + b.n_pairs
        assert np.array_equal(m.batch.positions.numpy(), np.concatenate([a.batch.positions.numpy(), b.batch.positions.numpy()]))
        assert np.array_equal(m.batch.seq_len.numpy(), np.concatenate([a.batch.seq_len.numpy(), b.batch.seq_len.numpy()]))
        assert np.array_equal(m.batch.pfx_start.numpy()[: a.batch.n_seqs], a.batch.pfx_start.numpy())
        assert np.array_equal(m.batch.pfx_start.numpy()[a.batch.n_seqs:], b.batch.pfx_start.numpy() + np.where(b.batch.pfx_len.numpy() > 0, a.n_tokens, 0))
        slots = sorted(int(x) for o in m.out_index for x in o)
        assert slots == list(range(len(first) + len(second)))
        assert sorted(int(x) for o in m.out_index[a.n_pairs:] for x in o) == list(range(len(first), len(first) + len(second)))
    sc.max_tokens = 120
    pairs = np.array([[j, i] for j in range(6) for i in range(6)])
    plans = list(sc.iter_tvg_jobs([(pairs, False), (pairs, True)]))
    assert len(plans) > 1 and sorted(int(x) for p in plans for o in p.out_index for x in o) == list(range(72))


def test_plans_split_at_the_token_budget():
    sc, prob = _scorer()
    sc.max_tokens = 150
    pairs = np.array([[j, i] for j in range(6) for i in range(6)])
    plans = sc.plan_vtg(pairs)
    assert len(plans) > 1 and sum(p.n_pairs for p in plans) == 36
    covered = sorted(int(o[0]) for p in plans for o in p.out_index)
    assert covered == list(range(36))


def test_vtg_cpn_without_any_prompt_token_is_rejected():
    """Headline-shaped rows ([<image>][text], no ChatML header): with the video keys masked nothing is visible in front of the
    response, so the planner refuses instead of pointing the first label row in front of the packed batch."""
    sc, prob = _scorer(layout=False)
    with pytest.raises(ValueError, match="at least one prompt token"):
        sc.plan_vtg(np.array([[0, 1], [2, 1]]), cpn=True)
    (plan,) = sc.plan_vtg(np.array([[0, 1], [2, 1]]))              # the likelihood pass itself is fine
    assert plan.n_pairs == 2 and int(plan.rows.numpy().min()) >= 0


def test_packed_batch_rejects_positions_beyond_the_rope_table():
    pos = np.arange(40, dtype=np.int32)
    b = eng.PackedBatch(pos, np.ones(40, np.uint8), np.array([0], np.int32), np.array([40], np.int32), device="cpu")
    assert b.max_position == 39
    b.struct(40)                                         # fits
    with pytest.raises(eng.BlimError, match="RoPE table"):
        b.struct(39)


# ----------------------------------------------------------------------------- planner invariants on random inputs (CPU)
def _check_plan(plan, n_req_seen):
    b = plan.batch
    T = plan.n_tokens
    ss, sl = b.seq_start.numpy(), b.seq_len.numpy()
    ps, pl = b.pfx_start.numpy(), b.pfx_len.numpy()
    pos = b.positions.numpy()
    # sequences tile the packed token range without gaps or overlaps
    order = np.argsort(ss)
    assert ss[order][0] == 0 and np.array_equal(ss[order][1:], (ss[order] + sl[order])[:-1]) and ss[order][-1] + sl[order][-1] == T
    ost = b.own_start.numpy() if getattr(b, "own_start", None) is not None else np.zeros(T, np.int32)
    for s in range(b.n_seqs):
        own = pos[ss[s]: ss[s] + sl[s]]
        seg = ost[ss[s]: ss[s] + sl[s]]                                       # segmented sequences (TVG: one segment per candidate video of a text)
        idx = np.arange(sl[s])
        assert np.all(seg <= idx) and np.all(np.diff(seg) >= 0)
        starts = np.unique(seg)
        assert np.array_equal(seg[starts], starts)                            # a segment's first token names itself
        for a, e in zip(starts, list(starts[1:]) + [sl[s]]):
            assert np.all(np.diff(own[a:e]) >= 1)                             # positions increase inside a segment
        if pl[s] > 0:                                                        # a prefix is another sequence's tokens, entirely in front of the own tokens
            assert 0 <= ps[s] and ps[s] + pl[s] <= T
            assert pos[ps[s]: ps[s] + pl[s]].max() < own.min()
    rows = plan.rows.numpy()
    assert rows.min() >= 0 and rows.max() < T
    if plan.kind == "vtg":
        rs = plan.row_start.numpy()
        assert rs[0] == 0 and rs[-1] == len(rows) == plan.n_rows and np.all(np.diff(rs) >= 1)
        assert len(plan.labels.numpy()) == plan.n_rows
    else:
        assert len(rows) == plan.n_pairs * 4 and len(plan.labels.numpy()) == plan.n_pairs
    # 32-query blocks cover every sequence
    bq = collections.Counter(b.blk_seq.numpy().tolist())
    assert all(bq[s] == (sl[s] + 31) // 32 for s in range(b.n_seqs))
    for outs in plan.out_index:
        for o in np.atleast_1d(outs):
            n_req_seen[int(o)] += 1


@pytest.mark.parametrize("seed", range(12))
def test_planner_invariants_on_random_pair_lists(seed):
    rs = np.random.RandomState(seed)
    n = int(rs.randint(1, 8))
    sc, prob = _scorer(n=n, layout=bool(seed % 4))
    sc.max_tokens = int(rs.choice([90, 200, 700, 4096]))
    P = int(rs.randint(1, 40))
    pairs = np.stack([rs.randint(0, n, P), rs.randint(0, n, P)], axis=1)          # duplicates allowed
    for kind, cpn in (("vtg", False), ("vtg", True), ("tvg", False), ("tvg", True)):
        if kind == "vtg" and cpn and not (seed % 4):
            continue                                                              # headline rows: the VTG prior is rejected (tested above)
        plans = sc.plan_vtg(pairs, cpn) if kind == "vtg" else sc.plan_tvg(pairs, cpn)
        seen = collections.Counter()
        for p in plans:
            _check_plan(p, seen)
        assert sorted(seen) == list(range(P)) and all(v == 1 for v in seen.values()), (kind, cpn)   # every requested pair answered exactly once


def test_bench_launcher_command_plumbing():
    """bench.py --gpus N without a launcher: the child command is torch.distributed.run with one process per GPU, a 127.0.0.1
    rendezvous and the user's own flags passed through."""
    import importlib.util
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_for_test", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
    argv = ["--gpus", "8", "--steps", "20", "--warmup", "3"]
    cmd = bench.launcher_command(8, argv, port=29777)
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and "--nproc-per-node=8" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29777"
    i = cmd.index(os.path.join(root, "bench.py"))
    assert cmd[i + 1:] == argv
    port = int(bench.launcher_command(2, [])[bench.launcher_command(2, []).index("--master-port") + 1])
    assert 1024 < port < 65536


def test_coefficient_sweeps_match_the_reference_when_present():
    """calculate_score / calculate_cpn_score (training_utils.py:106-140): brute-force property here; equality with the reference's own
    functions when /root/reference is importable (build container only)."""
    from blim_amd import training_utils as TU
    rs = np.random.RandomState(4)
    n = 40
    a, b = rs.randn(n, n).astype(np.float32) + 1.5 * np.eye(n, dtype=np.float32), rs.randn(n, n).astype(np.float32) + 0.5 * np.eye(n, dtype=np.float32)
    c, d = rs.randn(n, n).astype(np.float32) + 1.0 * np.eye(n, dtype=np.float32), rs.randn(n, n).astype(np.float32)
    ids = {i: i for i in range(n)}
    t2v, v2t, ct, cv = TU.calculate_score(a, b, c, d, ids, ids)
    grid = [round(float(x), 1) for x in np.linspace(0, 1, 11)]
    r1 = lambda m: TU.get_recall(m, m, ids, ids)["t2v_r1"]
    assert ct in grid and r1(ct * a + (1 - ct) * c) == max(r1(x * a + (1 - x) * c) for x in np.linspace(0, 1, 11))
    assert np.allclose(t2v, ct * a + (1 - ct) * c) and np.allclose(v2t, cv * b + (1 - cv) * d)
    p2t, p2v, pt, pv = TU.calculate_cpn_score(a, b, c, d, ids, ids)
    assert np.allclose(p2t, a - pt * c) and np.allclose(p2v, b - pv * d)
    from oracle import ref_harness
    if ref_harness.available():
        R = ref_harness.load().TU
        for mine, ref in ((TU.calculate_score(a, b, c, d, ids, ids), R.calculate_score(a, b, c, d, ids, ids)),
                          (TU.calculate_cpn_score(a, b, c, d, ids, ids), R.calculate_cpn_score(a, b, c, d, ids, ids))):
            assert mine[2:] == ref[2:] and np.array_equal(mine[0], ref[0]) and np.array_equal(mine[1], ref[1])


def test_host_thread_cap_is_sane(monkeypatch):
    """distributed.host_threads: torch intra-op threads for the drivers = CPUs this process may use / ranks on THIS node, between 1 and the cap."""
    from blim_amd import distributed as D
    monkeypatch.delenv("LOCAL_WORLD_SIZE", raising=False)
    n1 = D.host_threads(1)
    monkeypatch.setenv("LOCAL_WORLD_SIZE", "8")
    n8 = D.host_threads(8)
    assert 1 <= n8 <= n1 <= 8 and D.host_threads(1, cap=2) <= 2
    # a 4-node x 8-GPU job: the division is by the node's 8 ranks, not by the global 32 (which throttled every rank to one thread)
    assert D.host_threads(32) == n8 and D.local_world_size(32) == 8
    monkeypatch.setenv("LOCAL_WORLD_SIZE", str(10 ** 6))
    assert D.host_threads(10 ** 6) == 1
    assert D.usable_cpus() >= 1


def test_chunked_projection_follows_the_order_of_first_use():
    """PairScorer.video_feat projects a CHUNK on a miss: the requested video plus the next ones the running pass will ask for (expect()), at most
    feat_chunk at a time, every video exactly once."""
    sc, prob = _scorer(n=12)
    calls = []
    orig = sc.m.project
    sc.m.project = lambda feat, tvg, cache=True: (calls.append(int(feat.reshape(-1)[0] * 0) + len(calls)), orig(feat, tvg, cache))[1]
    sc.feat_chunk = 5
    pairs = np.array([[j, i] for i in (3, 1) for j in (7, 2, 9, 2, 11, 0, 4)])
    sc.plan_tvg(pairs)
    wanted = [2, 7, 9, 11, 0, 4]                                  # text 1 first (lexsort by text, then video): videos 0 2 4 7 9 11; all distinct videos once
    assert len(calls) == 6 and sorted(k[0] for k in sc._vfeat) == sorted(set(wanted))
    n0 = len(calls)
    sc.plan_tvg(pairs)                                            # second pass: everything cached
    assert len(calls) == n0


def test_driver_refuses_checkpoint_configs_outside_the_scoring_splice(tmp_path):
    """blim_amd.main.dims_from_config: the splice the path builds is the reference's `video_image` / `spatial_*pad*` / no-newline branch
    (modeling_videochat_flash.py:209-243); token compression and any other merge type change the rows and are refused, not approximated."""
    import json
    from blim_amd.main import dims_from_config
    base = {"vocab_size": 152064, "hidden_size": 3584, "intermediate_size": 18944, "num_hidden_layers": 28, "num_attention_heads": 28, "num_key_value_heads": 4,
            "vision_encode_type": "video_image", "mm_patch_merge_type": "spatial_nopad", "mm_newline_position": "nothing"}

    def write(**kw):
        c = dict(base, **kw)
        json.dump({k: v for k, v in c.items() if v is not None}, open(tmp_path / "config.json", "w"))
        return str(tmp_path)

    d = dims_from_config(write(), 4)
    assert (d.hidden_size, d.num_layers, d.num_kv_heads, d.mm_hidden_size, d.num_clips) == (3584, 28, 4, 1024, 4)
    for bad in (dict(mm_llm_compress=True), dict(vision_encode_type=None), dict(mm_patch_merge_type="flat"), dict(mm_newline_position="one_token"),
                dict(frame_aspect_ratio="anyres_max_9")):
        with pytest.raises(NotImplementedError):
            dims_from_config(write(**bad), 4)


def test_executed_flops_formula_and_calibration_pairs():
    """retrieval_utils.executed_flops: the per-token constants of SURVEY.md section 8a on the token counts launched, the compensated modes' doubled GEMMs,
    the last layer's pruned rows; calibration_pairs: the same pairs on every rank, inside the top-k."""
    import torch
    from blim_amd import retrieval_utils as RU
    d = synth.ModelDims()
    F, qkv, head = 466092032, 2 * 3584 * (3584 + 2 * 512), 1089994752
    tok, rows = 32560, 28160
    assert RU.executed_flops(d, tok, rows, "vtg", None, prune=False) == 28 * F * tok + head * rows
    assert RU.executed_flops(d, tok, rows, "vtg", None) == 28 * F * tok + head * rows - (F - qkv) * (tok - rows)
    assert RU.executed_flops(d, tok, tok, "vtg", None) == 28 * F * tok + head * tok                    # nothing to prune when every row is read
    assert RU.executed_flops(d, tok, rows, "vtg", "full", prune=False) == 2 * (28 * F * tok + head * rows)
    assert RU.executed_flops(d, tok, rows, "vtg", "attn", prune=False) == 28 * (F + qkv + 2 * 3584 * 3584) * tok + 2 * head * rows
    tvg_head = (2 * 3584 * 1024 + 2 * 1024 * 1000) * 400                                               # visual head + vocabulary product: three-term when compensated
    assert RU.executed_flops(d, 1000, 400, "tvg", "full", n_vocab=1000, prune=False) == 2 * 28 * F * 1000 + 3 * tvg_head
    mlp_gu, mlp_d = 4 * 3584 * 18944, 2 * 3584 * 18944
    assert RU.executed_flops(d, 1000, 400, "tvg", "attn", n_vocab=1000, prune=False) == 28 * (2 * F - mlp_d - mlp_gu) * 1000 + 3 * tvg_head      # TVG "attn": the MLP branch walks K once
    assert RU.executed_flops(d, 1000, 400, "tvg", None, n_vocab=1000, prune=False) == 28 * F * 1000 + tvg_head
    assert RU.TVG_MODES == ("attn", "full") and RU.VTG_MODES == ("none", "full")                         # round 5: the intermediate modes (qk, qkx, act0, VTG attn) are gone
    # the e2m3 share (engine option "precise_lo6"): the second walk over K of the compensated decoder GEMMs and of lm_head; never in the plain mode
    assert RU.lo6_pass_flops(d, tok, rows, "vtg", None) == 0
    assert RU.lo6_pass_flops(d, tok, rows, "vtg", "full", prune=False) == RU.executed_flops(d, tok, rows, "vtg", "full", prune=False) / 2
    assert RU.lo6_pass_flops(d, tok, rows, "vtg", "attn", prune=False) == 28 * (qkv + 2 * 3584 * 3584) * tok + head * rows
    assert RU.lo6_pass_flops(d, 1000, 400, "tvg", "full", prune=False) == 28 * F * 1000                   # the TVG head's three-term products are 16-bit GEMMs
    assert RU.lo6_pass_flops(d, tok, rows, "vtg", "full") == RU.executed_flops(d, tok, rows, "vtg", "full") - RU.executed_flops(d, tok, rows, "vtg", None)
    sims = torch.from_numpy(np.random.RandomState(0).randn(40, 50).astype(np.float32))
    p = RU.calibration_pairs(sims, topk=5)
    assert p.shape == (16 * 5, 2) and len(np.unique(p[:, 0])) == 16 and p[:, 0].min() == 0 and p[:, 0].max() == 39
    top = sims.topk(5, dim=1).indices.numpy()
    assert all(c in top[q] for q, c in p)
    assert np.array_equal(p, RU.calibration_pairs(sims, topk=5))                                        # deterministic: the same on every rank
    assert RU.calibration_pairs(sims[:3], topk=64).shape == (3 * 16, 2)                                 # at most 16 per query, at most Nt


def test_predicted_max_deviation_extrapolates_the_samples_tail():
    """retrieval_utils.predicted_max_deviation (`--vtg_precise` / `--tvg_precise auto`): a 256-entry sample of a log-normal law with the tail measured on weights with
    massive activations (sigma_log 0.9) predicts the largest of 48,000 entries within its sampling spread, far above the sample's own maximum; Gaussian deviations are
    over- not under-estimated; with no more entries than the sample the sample maximum is returned."""
    import math
    from statistics import NormalDist
    from blim_amd import retrieval_utils as RU
    rng = np.random.RandomState(0)
    med, sig, N = 1e-4, 0.9, 48000
    true_max = med * math.exp(sig * NormalDist().inv_cdf(1 - 1 / N))              # 3.9e-3
    preds = []
    for _ in range(40):
        x = med * np.exp(sig * rng.randn(256))
        p = RU.predicted_max_deviation(x, N)
        assert p >= x.max()
        preds.append(p)
    assert 0.7 * true_max < np.median(preds) < 1.4 * true_max and min(preds) > 0.4 * true_max, (true_max, np.median(preds), min(preds))
    for _ in range(20):
        x = np.abs(rng.randn(256)) * 5e-5                                          # half-normal: the true largest of 48,000 is ~ 4.3 x rms = 2.1e-4
        p = RU.predicted_max_deviation(x, N)
        assert 2.1e-4 < p < 7e-4, p                                                # conservative (~ 8 x rms), still inside the bar
    x = np.abs(rng.randn(256)) * 5e-5
    assert RU.predicted_max_deviation(x, 256) == x.max() and RU.predicted_max_deviation(x, None) == x.max()
    assert RU.predicted_max_deviation(np.array([0.0, np.nan]), N) == 0.0
    assert RU.predicted_max_deviation(x[:10], N) == x[:10].max()                    # too small a sample to fit: its maximum



def test_calibrate_second_pass_decides_between_the_two_second_passes_of_a_bf16_engine():
    """calibration.CalibrationMixin.calibrate_second_pass with a stand-in scorer: the e2m3 form is measured against the bf16 one through the same stages as the VTG modes
    (accepted when quiet, rejected when its deviations are outside the bar), the engine is left in the chosen form, and other engines have nothing to measure."""
    from blim_amd import retrieval_utils as RU
    from blim_amd.calibration import CalibrationMixin

    class S(CalibrationMixin):
        def __init__(self, dtype, scale):
            self.opts, self.scale, self.resolved = [], scale, []
            self.engine = types.SimpleNamespace(dtype=dtype, lo6=False, can_precise=True, set_option=lambda k, v: (self.opts.append((k, v)), setattr(self.engine, "lo6", bool(v))))
            self.m = types.SimpleNamespace(resolve_second_pass=lambda mode: (self.resolved.append(mode), self.engine.set_option("precise_lo6", int(mode == "e2m3"))))

        def set_vtg_mode(self, mode):
            self.mode = mode

        def vtg(self, pairs, cpn=False):
            base = -2.0 - 0.01 * (pairs[:, 1] % 5)
            noise = np.random.RandomState(7).randn(100000)[(pairs[:, 0] * 131 + pairs[:, 1]) % 100000]
            return base * (1.0 + (self.scale * noise if self.engine.lo6 else 0.0))

    sims = np.random.RandomState(1).randn(300, 300).astype(np.float32)
    first, confirm = RU.calibration_pairs(sims, 16, n_queries=32, per_query=8), RU.calibration_pairs(sims, 16, n_queries=256, per_query=8)
    quiet = S("bf16", 2e-5)
    chosen, table = quiet.calibrate_second_pass(first, n_eval=48000, confirm_pairs=confirm)
    assert chosen == "e2m3" and quiet.resolved == ["e2m3"] and quiet.engine.lo6 and quiet.mode == "full" and table["e2m3"]["max"] < 1e-4
    loud = S("bf16", 6e-4)
    chosen, table = loud.calibrate_second_pass(first, n_eval=48000, confirm_pairs=confirm)
    assert chosen == "16bit" and not loud.engine.lo6 and table["e2m3"]["max"] > 1e-3
    f16 = S("f16", 1.0); f16.engine.lo6 = True
    assert f16.calibrate_second_pass(first, n_eval=48000) == ("e2m3", {}) and f16.opts == []


def test_second_pass_request_and_resolution():
    """BlimModel.second_pass (round 6): "e2m3" | "16bit" switch the engine option at once; "auto" is a request on bf16 engines (the parity form runs until evaluation()
    has measured, and again after any weight change) and simply means e2m3 on fp16 engines."""
    from blim_amd.modeling import BlimModel
    opts = []
    mk = lambda dtype, lo6: types.SimpleNamespace(weights_version=0, can_precise=True, dtype=dtype, lo6=lo6,
                                                    set_option=lambda k, v: (opts.append((k, v)), setattr(m.engine, "lo6", bool(v)) if k == "precise_lo6" else None))
    m = BlimModel.__new__(BlimModel)
    m.engine = mk("bf16", False)
    m._vtg_request, m._tvg_request, m._vtg_resolved, m._tvg_resolved = "full", "full", None, None
    assert m.second_pass == "16bit" and m.second_pass_resolved()
    m.second_pass = "e2m3"
    assert opts[-1] == ("precise_lo6", 1) and m.second_pass == "e2m3"
    m.second_pass = "auto"
    assert opts[-1] == ("precise_lo6", 0) and m.second_pass == "auto" and not m.second_pass_resolved()
    m.resolve_second_pass("e2m3")
    assert opts[-1] == ("precise_lo6", 1) and m.second_pass == "auto" and m.second_pass_resolved()
    m.engine.weights_version += 1
    assert not m.second_pass_resolved()
    m.engine = mk("f16", True)
    m.second_pass = "auto"
    assert m.second_pass == "e2m3" and m.second_pass_resolved()
    with pytest.raises(ValueError):
        m.second_pass = "fp4"


class _FakeCalScorer:
    """Stand-in for PairScorer under calibration.CalibrationMixin: vtg() returns a per-pair "true" score, perturbed in the plain mode by a seeded per-pair deviation."""

    def __init__(self, dev_of_pair):
        from blim_amd.calibration import CalibrationMixin
        self.__class__ = type("Fake", (_FakeCalScorer, CalibrationMixin), {})
        self.dev_of_pair, self.mode, self.engine, self.m = dev_of_pair, None, types.SimpleNamespace(can_precise=True), types.SimpleNamespace()
        self.calls = []

    def set_vtg_mode(self, mode):
        self.mode = None if mode in (None, "none") else mode

    def vtg(self, pairs, cpn=False):
        self.calls.append((self.mode, len(pairs)))
        base = -1.0 - 0.001 * (pairs[:, 0] % 7)
        return (base if self.mode == "full" else base * (1.0 + self.dev_of_pair(pairs))).astype(np.float64)


def test_auto_confirmation_sample_before_an_extrapolated_reject():
    """calibration.CalibrationMixin._decide (round 6).  A 256-pair sample whose own statistics are inside the bar but whose log-normal tail fit,
    read off at the 472,000 entries of an ActivityNet-sized evaluation, is not: Gaussian-like deviations (rms 7e-5: the largest of 472,000 is ~ 3.5e-4) used to be sent
    to the compensated mode by the 2.2 x overshoot of that fit.  Now a confirmation sample (2,048 pairs) is measured first and the tail read from its top eighth:
    Gaussian-like deviations are ACCEPTED, a genuinely heavy tail (log-normal, sigma 0.9, median 1e-4: the heavy7b law, largest of 48,000 = 3.9e-3) stays REJECTED, a
    sample that is outside the bar by itself is rejected without the second stage, and an emulated rank (`adopt`) measures the job's stages but takes the job's decision."""
    from blim_amd import retrieval_utils as RU
    rng = np.random.RandomState(5)
    sims = rng.randn(600, 600).astype(np.float32)
    first = RU.calibration_pairs(sims, 16, n_queries=32, per_query=8)
    confirm = RU.calibration_pairs(sims, 16, n_queries=256, per_query=8)
    assert len(first) == 256 and len(confirm) == 2048
    table_of = lambda law: {(int(a), int(b)): v for (a, b), v in zip(confirm, law(len(confirm)))} | {(int(a), int(b)): v for (a, b), v in zip(first, law(len(first)))}
    mk = lambda tab: _FakeCalScorer(lambda pairs: np.array([tab[(int(a), int(b))] for a, b in pairs]))
    # (1) Gaussian-like, rms 7e-5
    gauss = table_of(lambda n: np.abs(rng.randn(n)) * 7e-5)
    sc = mk(gauss)
    chosen, table = sc.calibrate_vtg(first, n_eval=472000, confirm_pairs=confirm)
    e = table["none"]
    assert e["max"] < 1e-3 and 4.5 * e["rms"] < 1e-3 and e["pred"] > 8e-4, e                      # stage 1 alone would have rejected (the false reject)
    assert "confirm" in e and e["confirm"]["n"] >= 2048 and e["confirm"]["pred"] < 8e-4 and e["confirm"]["accepted"], e
    assert chosen == "none" and sc.mode is None
    assert sum(n for m, n in sc.calls if m == "full") == e["confirm"]["n"]                        # every measured pair scored once per mode
    # without a confirmation sample: the old behaviour
    assert mk(gauss).calibrate_vtg(first, n_eval=472000)[0] == "full"
    # the same sample at MSRVTT size (48,000 entries) is accepted by stage 1 alone
    sc = mk(gauss); chosen, table = sc.calibrate_vtg(first, n_eval=48000, confirm_pairs=confirm)
    assert chosen == "none" and "confirm" not in table["none"]
    # (2) heavy tail: rejected, with or without the second stage
    heavy = table_of(lambda n: 1e-4 * np.exp(0.9 * rng.randn(n)))
    sc = mk(heavy); chosen, table = sc.calibrate_vtg(first, n_eval=48000, confirm_pairs=confirm)
    assert chosen == "full" and sc.mode == "full", table
    # (3) a sample outside the bar by itself: no second stage
    bad = table_of(lambda n: np.abs(rng.randn(n)) * 6e-4)
    sc = mk(bad); chosen, table = sc.calibrate_vtg(first, n_eval=472000, confirm_pairs=confirm)
    assert chosen == "full" and "confirm" not in table["none"] and sum(n for m, n in sc.calls if m == "full") == 256
    # (4) an emulated rank 3 of 8: its block of both stages, the job's decision
    sc = mk(gauss); chosen, table = sc.calibrate_vtg(first, n_eval=472000, confirm_pairs=confirm, share=(8, 3), adopt=("none", True))
    assert chosen == "none" and table["none"]["adopted"] == "none" and sum(n for m, n in sc.calls if m == "full") == 256 // 8 + (len(confirm) - len({tuple(p) for p in first} & {tuple(p) for p in confirm})) // 8
    sc = mk(gauss); chosen, table = sc.calibrate_vtg(first, n_eval=472000, confirm_pairs=confirm, share=(8, 3), adopt=("full", False))
    assert chosen == "full" and sum(n for m, n in sc.calls if m == "full") == 32


def test_auto_numeric_modes_are_requests_and_their_resolution_follows_the_weights():
    """BlimModel keeps what the user ASKED for (vtg_precise / tvg_precise: none | full | auto, attn | full | auto) apart from what `auto` RESOLVED to, and the resolution
    stands only while the engine's weights and adapters are the ones it was measured on (the first evaluation() used to overwrite "auto" with its choice, and
    the training loop's validation scored every later epoch's adapters in the mode measured on the first)."""
    import types
    from blim_amd.modeling import BlimModel
    m = BlimModel.__new__(BlimModel)                                       # no engine needed for this logic: a stand-in with the two attributes it reads
    m.engine = types.SimpleNamespace(weights_version=0, can_precise=True, dtype="f16")
    m._vtg_request, m._tvg_request, m._vtg_resolved, m._tvg_resolved = None, "full", None, None
    assert m.vtg_mode() is None and m.tvg_mode() == "full"
    m.vtg_precise, m.tvg_precise = "auto", "auto"
    assert m.vtg_mode() == "auto" and m.tvg_mode() == "full" and not m.tvg_resolved()          # unresolved: VTG says so, TVG runs fully compensated
    m.resolve_vtg("none"); m.resolve_tvg("attn")
    assert m.vtg_precise == "auto" and m.tvg_precise == "auto"                                  # the requests are untouched
    assert m.vtg_mode() is None and m.tvg_mode() == "attn" and m.tvg_resolved()
    m.engine.weights_version += 1                                                               # a weight or an adapter was (re)loaded
    assert m.vtg_mode() == "auto" and m.tvg_mode() == "full" and not m.tvg_resolved()
    m.resolve_vtg("full")
    assert m.vtg_mode() == "full"
    m.vtg_precise = "none"
    assert m.vtg_precise is None and m.vtg_mode() is None
    for bad in ("qk", "qkx", "attn", "act0"):
        with pytest.raises(ValueError):
            m.vtg_precise = bad
    with pytest.raises(ValueError):
        m.tvg_precise = "act0"
