"""Soak run of tests/test_gpu_fuzz.py's comparison on RANDOM case parameters (fused PairScorer + literal API vs the numpy oracle): problem size, tokens per
clip, caption lengths, top-k, batch size, layout, the planner's token budget (forces plan splits inside a text's candidate group and inside merged TVG
sequences) and, for a third of the cases, config.tokenizer_model_max_length.   python tools/fuzz_more.py [cases] [first seed]"""
import os, sys, types, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from blim_amd import retrieval_utils as RU, synth
from blim_amd.modeling import BlimModel, DDPLike
from oracle import blim_oracle as O

D = dict(vocab_size=151700, hidden_size=256, intermediate_size=512, num_layers=2, num_heads=2, num_kv_heads=1, mm_hidden_size=64)
if os.environ.get("BLIM_FUZZ_DIMS") == "h512":      # wide enough for the RMSNorm kernel that writes e2m3 tiles (H > 256), 2 I a multiple of 256 (SwiGLU-written tiles), 9 and 4 second-pass K-steps
    D = dict(vocab_size=151700, hidden_size=512, intermediate_size=1152, num_layers=2, num_heads=4, num_kv_heads=2, mm_hidden_size=64)
PASSES = [("v2t", "vtg", False), ("v2t", "vtg", True), ("v2t", "tvg", False), ("t2v", "vtg", False), ("t2v", "tvg", False), ("t2v", "tvg", True)]
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 5000
dims = synth.ModelDims(**D)
w = synth.synthetic_weights(dims, 9)
model = BlimModel(dims, max_positions=1024, dtype=os.environ.get("BLIM_DTYPE", "f16"))
model.engine.load_weights(w)
om = O.OracleModel(O.OracleConfig(**D), w)
tok = types.SimpleNamespace(pad_token_id=synth.PAD_ID)
worst_all, t00 = 0.0, time.time()
for case in range(n_cases):
    rs = np.random.RandomState(seed0 + case)
    n = int(rs.randint(1, 11)); tpc = int(rs.choice([1, 2, 3, 5, 8, 16, 33, 64])); lo = int(rs.randint(1, 20)); hi = lo + int(rs.randint(0, 40))
    topk = int(rs.randint(1, n + 3)); bs = int(rs.randint(1, 6)); layout = bool(rs.rand() < 0.8)
    prob = synth.make_problem(seed0 + case, n, dims, tok_per_clip=tpc, text_len=(lo, hi), reference_layout=layout)
    full = [len(x) - 1 + 4 * tpc for x in prob.vtg_ids]
    resp = [int((np.asarray(x) != -100).sum()) for x in prob.vtg_labels]
    limit = None
    if rs.rand() < 0.33:                                  # cut some rows, keep at least one response token of every row and every TVG row whole
        lo_ok = max(max(f - r + 1 for f, r in zip(full, resp)), max(len(x) - 1 + dims.num_clips for x in prob.tvg_ids))
        if lo_ok < max(full):
            limit = int(rs.randint(lo_ok, max(full) + 1))
    longest = max(max(full), max(len(x) + 3 for x in prob.tvg_ids))
    max_tokens = int(rs.choice([longest + 8, 2 * longest, 600, 4096]))
    model.tokenizer_model_max_length = limit; om.tokenizer_model_max_length = limit
    model.set_tvg_prefix_length(prob.tvg_prefix_length); om.set_tvg_prefix_length(prob.tvg_prefix_length)
    model.clear_cache()
    Tt = lambda rows: [torch.from_numpy(r) for r in rows]
    vtg = RU.padding_ids(Tt(prob.vtg_ids), Tt(prob.vtg_labels), Tt(prob.vtg_masks), tok)
    tvg = RU.padding_ids(Tt(prob.tvg_ids), Tt(prob.tvg_labels), Tt(prob.tvg_masks), tok)
    ov = O.padding_ids(prob.vtg_ids, prob.vtg_labels, prob.vtg_masks, synth.PAD_ID)
    ot = O.padding_ids(prob.tvg_ids, prob.tvg_labels, prob.tvg_masks, synth.PAD_ID)
    video = [torch.from_numpy(v) for v in prob.video]
    vocab, vlab = torch.from_numpy(prob.video_vocab), torch.from_numpy(prob.tvg_video_labels)
    ddp, dev = DDPLike(model), model.device
    args = types.SimpleNamespace(topk=topk, batch_size_eval=bs, num_clips=dims.num_clips)
    scorer = RU.PairScorer(ddp, vtg[0], vtg[2], vtg[1], tvg[0], tvg[2], tvg[1], video, vocab, vlab, dims.num_clips, max_tokens=max_tokens)
    worst = 0.0
    for direction, ft, cpn in PASSES:
        if not layout and ft == "vtg" and cpn:
            continue
        qv = direction == "v2t"
        sims = prob.v2t_sims if qv else prob.t2v_sims
        o_ids, o_lab, o_msk = ov if ft == "vtg" else ot
        fn_o = O.compute_v2t_scores_x if qv else O.compute_t2v_scores_x
        want = fn_o(np.full((n, n), -100.0, np.float32), sims, 0, o_ids, o_msk, o_lab, prob.video, prob.video_vocab, prob.tvg_video_labels, om, topk, bs, dims.num_clips, ft, cpn)
        m = want != -100.0
        pairs = RU._topk_pairs(torch.from_numpy(sims), 0, topk, qv)
        sc = scorer.vtg(pairs, cpn) if ft == "vtg" else scorer.tvg(pairs, cpn)
        S = np.full((n, n), -100.0, np.float32)
        r, c = (pairs[:, 0], pairs[:, 1]) if qv else (pairs[:, 1], pairs[:, 0])
        S[r, c] = sc
        ids, lab, msk = vtg if ft == "vtg" else tvg
        fn = RU.compute_v2t_scores_x if qv else RU.compute_t2v_scores_x
        Lm = fn(torch.full((n, n), -100.0, device=dev), torch.from_numpy(sims), 0, ids, msk, lab, video, vocab.to(dev), vlab, ddp, dev, args, forward_type=ft, cpn=cpn).cpu().numpy()
        for name, G in (("fused", S), ("literal", Lm)):
            assert np.array_equal(G != -100.0, m), (case, name, direction, ft, cpn)
            assert np.isfinite(G[m]).all(), (case, name, direction, ft, cpn)
            rel = float((np.abs(G[m] - want[m]) / np.maximum(np.abs(want[m]), 1e-4)).max())     # one-entry vocabularies score exactly 0
            worst = max(worst, rel)
            assert rel < 1e-3, (case, name, direction, ft, cpn, rel, dict(n=n, tpc=tpc, tl=(lo, hi), topk=topk, bs=bs, layout=layout, limit=limit, max_tokens=max_tokens))
    worst_all = max(worst_all, worst)
    print(f"case {case} (seed {seed0 + case}): n={n} tok/clip={tpc} text_len=({lo},{hi}) topk={topk} bs={bs} layout={layout} limit={limit} max_tokens={max_tokens}: worst {worst:.2e}", flush=True)
print(f"{n_cases} cases ok, worst relative deviation {worst_all:.2e}, {time.time() - t00:.0f} s")
model.engine.close()
