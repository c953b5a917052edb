"""GPU (-m gpu): seeded random problem shapes through the fused PairScorer AND the literal reference-shaped API, each compared
entry by entry with the numpy oracle's restatement of compute_*_scores_x (oracle/blim_oracle.py; itself pinned to the reference
by tests/golden).  Covers the ragged edge cases the golden fixtures do not: a single video/text, top-k larger than N, one-token
captions, one token per clip, batch size 1, headline-shaped rows without a prompt, paragraph-length captions."""
import types

import numpy as np
import pytest
import torch

from blim_amd import retrieval_utils as RU
from blim_amd import synth
from blim_amd.modeling import BlimModel, DDPLike
from oracle import blim_oracle as O

pytestmark = pytest.mark.gpu
D = dict(vocab_size=151700, hidden_size=256, intermediate_size=512, num_layers=2, num_heads=2, num_kv_heads=1, mm_hidden_size=64)
PASSES = [("v2t", "vtg", False), ("v2t", "vtg", True), ("v2t", "tvg", False), ("t2v", "vtg", False), ("t2v", "tvg", False), ("t2v", "tvg", True)]
#        seed  n  tok/clip  text_len  topk  bs  reference_layout
CASES = [(101, 1, 8, (3, 6), 4, 3, True),        # one video, one text; top-k > N
         (102, 5, 1, (1, 1), 2, 1, True),        # one token per clip, one-token captions, batch size 1
         (103, 4, 3, (1, 12), 7, 5, True),       # top-k > N, ragged lengths
         (104, 6, 8, (2, 9), 3, 2, True),
         (105, 3, 5, (4, 4), 3, 3, False),       # headline-shaped rows (no prompt tokens): VTG prior undefined -> skipped
         (106, 7, 2, (1, 5), 5, 4, True),
         (107, 3, 40, (33, 70), 3, 2, True),     # captions across the 32- and 64-token query-block boundaries, 160-token video prefix
         (108, 2, 64, (60, 64), 2, 2, True),     # reference-sized rows: 256 video tokens
         (109, 9, 4, (1, 33), 9, 4, True),       # dense: every candidate of every query
         (110, 4, 8, (31, 33), 4, 3, False),     # headline-shaped, suffixes straddling one query block
         (111, 2, 64, (250, 420), 2, 1, True)]   # paragraph-length captions (DiDeMo / ActivityNet rows of ~700 tokens): bodies of many query blocks and key tiles


@pytest.fixture(scope="module")
def models():
    dims = synth.ModelDims(**D)
    w = synth.synthetic_weights(dims, 9)
    model = BlimModel(dims, max_positions=1024, dtype="f16")
    model.engine.load_weights(w)
    om = O.OracleModel(O.OracleConfig(**D), w)
    yield dims, model, om
    model.engine.close()


@pytest.mark.parametrize("seed,n,tpc,tl,topk,bs,layout", CASES)
def test_random_shapes_fused_and_literal_vs_oracle(models, seed, n, tpc, tl, topk, bs, layout):
    dims, model, om = models
    prob = synth.make_problem(seed, n, dims, tok_per_clip=tpc, text_len=tl, reference_layout=layout)
    model.set_tvg_prefix_length(prob.tvg_prefix_length); om.set_tvg_prefix_length(prob.tvg_prefix_length)
    model.clear_cache()
    tok = types.SimpleNamespace(pad_token_id=synth.PAD_ID)
    Tt = lambda rows: [torch.from_numpy(r) for r in rows]
    vtg = RU.padding_ids(Tt(prob.vtg_ids), Tt(prob.vtg_labels), Tt(prob.vtg_masks), tok)
    tvg = RU.padding_ids(Tt(prob.tvg_ids), Tt(prob.tvg_labels), Tt(prob.tvg_masks), tok)
    ov = O.padding_ids(prob.vtg_ids, prob.vtg_labels, prob.vtg_masks, synth.PAD_ID)
    ot = O.padding_ids(prob.tvg_ids, prob.tvg_labels, prob.tvg_masks, synth.PAD_ID)
    video = [torch.from_numpy(v) for v in prob.video]
    vocab, vlab = torch.from_numpy(prob.video_vocab), torch.from_numpy(prob.tvg_video_labels)
    ddp, dev = DDPLike(model), model.device
    args = types.SimpleNamespace(topk=topk, batch_size_eval=bs, num_clips=dims.num_clips)
    scorer = RU.PairScorer(ddp, vtg[0], vtg[2], vtg[1], tvg[0], tvg[2], tvg[1], video, vocab, vlab, dims.num_clips, max_tokens=600)
    for direction, ft, cpn in PASSES:
        if not layout and ft == "vtg" and cpn:
            with pytest.raises(ValueError):
                scorer.vtg(np.array([[0, 0]]), True)
            continue
        qv = direction == "v2t"
        sims = prob.v2t_sims if qv else prob.t2v_sims
        o_ids, o_lab, o_msk = ov if ft == "vtg" else ot
        fn_o = O.compute_v2t_scores_x if qv else O.compute_t2v_scores_x
        want = fn_o(np.full((n, n), -100.0, np.float32), sims, 0, o_ids, o_msk, o_lab, prob.video, prob.video_vocab, prob.tvg_video_labels,
                    om, topk, bs, dims.num_clips, ft, cpn)
        m = want != -100.0
        # fused
        pairs = RU._topk_pairs(torch.from_numpy(sims), 0, topk, qv)
        sc = scorer.vtg(pairs, cpn) if ft == "vtg" else scorer.tvg(pairs, cpn)
        S = np.full((n, n), -100.0, np.float32)
        r, c = (pairs[:, 0], pairs[:, 1]) if qv else (pairs[:, 1], pairs[:, 0])
        S[r, c] = sc
        assert np.array_equal(S != -100.0, m), (direction, ft, cpn)
        np.testing.assert_allclose(S[m], want[m], rtol=1e-3, err_msg=f"fused {direction} {ft} cpn={cpn}")
        # literal API (the reference's per-batch control flow)
        ids, lab, msk = vtg if ft == "vtg" else tvg
        fn = RU.compute_v2t_scores_x if qv else RU.compute_t2v_scores_x
        L = fn(torch.full((n, n), -100.0, device=dev), torch.from_numpy(sims), 0, ids, msk, lab, video, vocab.to(dev), vlab, ddp, dev, args,
               forward_type=ft, cpn=cpn).cpu().numpy()
        assert np.array_equal(L != -100.0, m), (direction, ft, cpn)
        np.testing.assert_allclose(L[m], want[m], rtol=1e-3, err_msg=f"literal {direction} {ft} cpn={cpn}")


@pytest.mark.parametrize("heads,kv,hidden,inter,dtype", [(2, 2, 256, 512, "f16"), (6, 2, 768, 1280, "f16"), (8, 1, 1024, 768, "bf16"), (3, 3, 384, 640, "bf16")],
                         ids=["mha-2", "gqa-3to1", "mqa-8to1-bf16", "mha-3-bf16"])
def test_other_head_groupings_and_widths(heads, kv, hidden, inter, dtype):
    """The fixtures cover 2:1 (tiny), 4:1 (deep) and 7:1 (7B) query-to-key-value head ratios; the attention kernel is instantiated per ratio and the GEMM
    tiles meet other edge shapes at other widths: plain multi-head, 3:1, 8:1 (one K/V head), an odd head count; fp16 and the compensated bf16 engine."""
    D2 = dict(D, num_heads=heads, num_kv_heads=kv, hidden_size=hidden, intermediate_size=inter)
    dims = synth.ModelDims(**D2)
    w = synth.synthetic_weights(dims, 9)
    model = BlimModel(dims, max_positions=512, dtype=dtype)
    try:
        model.engine.load_weights(w)
        test_random_shapes_fused_and_literal_vs_oracle((dims, model, O.OracleModel(O.OracleConfig(**D2), w)), 401, 4, 8, (3, 40), 3, 2, True)
    finally:
        model.engine.close()


@pytest.mark.parametrize("clips", [1, 2, 8])
def test_num_clips_other_than_four(clips):
    """args.num_clips (retrieval_utils.py:99: positions p + arange(C) - (C + 1)) is 4 in every golden fixture; the engine, the planner's merged TVG sequences
    (C - 1 tokens per segment) and the oracle follow the general rule -- one, two and eight clips per video, fused and literal against the oracle."""
    D2 = dict(D, num_clips=clips)
    dims = synth.ModelDims(**D2)
    w = synth.synthetic_weights(dims, 9)
    model = BlimModel(dims, max_positions=512, dtype="f16")
    try:
        model.engine.load_weights(w)
        om = O.OracleModel(O.OracleConfig(**D2), w)
        for case in [(301, 5, 8, (3, 12), 4, 3, True), (302, 3, 5, (4, 9), 3, 2, False)]:
            test_random_shapes_fused_and_literal_vs_oracle((dims, model, om), *case)
    finally:
        model.engine.close()


@pytest.mark.parametrize("n,topk,bs", [(48, 48, 16), (36, 32, 16)], ids=["dense-48", "top32-of-36"])
def test_evaluation_dense_and_top32_vs_oracle(models, n, topk, bs):
    """BASELINE configs 3 / 4 in miniature: evaluation() with dense candidates (k = N = 48) and with top-32 + CPN, all six matrices
    against the oracle's restatement of the reference loops; with and without the cross-direction de-duplication.  (The oracle is the
    slow side: 151,700-wide lm_head per label row in numpy; short captions keep the test under a minute.)"""
    dims, model, om = models
    prob = synth.make_problem(300 + n, n, dims, tok_per_clip=4, text_len=(1, 4))
    model.set_tvg_prefix_length(prob.tvg_prefix_length); om.set_tvg_prefix_length(prob.tvg_prefix_length)
    model.clear_cache()
    ov = O.padding_ids(prob.vtg_ids, prob.vtg_labels, prob.vtg_masks, synth.PAD_ID)
    ot = O.padding_ids(prob.tvg_ids, prob.tvg_labels, prob.tvg_masks, synth.PAD_ID)
    want = {}
    for direction, ft, cpn in PASSES:
        qv = direction == "v2t"
        if topk >= n and not qv and not cpn:
            # dense: the reference's own fixtures hold bit-identical numbers in v2t[j, i] and t2v[i, j] (tests/golden/full7b.npz:
            # S_v2t_vtg[0, 0] == S_t2v_vtg[0, 0]); the oracle's transposed matrix stands for the second loop
            want[(direction, ft, cpn)] = want[("v2t", ft, False)].T.copy()
            continue
        o_ids, o_lab, o_msk = ov if ft == "vtg" else ot
        fn_o = O.compute_v2t_scores_x if qv else O.compute_t2v_scores_x
        want[(direction, ft, cpn)] = fn_o(np.full((n, n), -100.0, np.float32), prob.v2t_sims if qv else prob.t2v_sims, 0, o_ids, o_msk, o_lab, prob.video,
                                          prob.video_vocab, prob.tvg_video_labels, om, topk, bs, dims.num_clips, ft, cpn)
    key = {("v2t", "vtg", False): ("v2t", "candidate_likelihood"), ("v2t", "vtg", True): ("v2t", "candidate_prior"), ("v2t", "tvg", False): ("v2t", "query_likelihood"),
           ("t2v", "vtg", False): ("t2v", "query_likelihood"), ("t2v", "tvg", False): ("t2v", "candidate_likelihood"), ("t2v", "tvg", True): ("t2v", "candidate_prior")}
    tok = types.SimpleNamespace(pad_token_id=synth.PAD_ID)
    res = {}
    for dedup in (True, False):
        args = types.SimpleNamespace(topk=topk, batch_size_eval=bs, num_clips=dims.num_clips, cpn=True, resume="ckpt", eval=True, dataset="SYNTH", dedup=dedup,
                                     max_tokens=2048, iv2_scores={"v2t": torch.from_numpy(prob.v2t_sims), "t2v": torch.from_numpy(prob.t2v_sims)})
        t2v, v2t = RU.evaluation(DDPLike(model), synth.ProblemLoader(prob, 16), model.device, tok, args)
        res[dedup] = (t2v, v2t, args._eval_stats)
        for k, (d, name) in key.items():
            S = (v2t if d == "v2t" else t2v)[name]
            W = want[k]
            assert np.array_equal(S != -100.0, W != -100.0), (k, dedup)
            m = W != -100.0
            np.testing.assert_allclose(S[m], W[m], rtol=1e-3, err_msg=f"{k} dedup={dedup}")
    assert res[False][2]["pairs_scored"] == res[False][2]["pairs_requested"] == 6 * n * min(topk, n)
    assert res[True][2]["pairs_scored"] < res[False][2]["pairs_scored"]
    if topk >= n:                                           # dense: every (video, text) pair once for VTG, once for TVG, n priors + the t2v prior
        assert res[True][2]["pairs_scored"] == 3 * n * n + n
    for a, b in zip(res[True][:2], res[False][:2]):
        for k in a:
            np.testing.assert_allclose(a[k], b[k], rtol=1e-5)
