"""CPU: the numpy oracle against the golden vectors recorded from the REFERENCE's own code (oracle/gen_golden.py)."""
import os

import numpy as np
import pytest

from blim_amd import synth
from oracle import blim_oracle as O
from oracle import synth_np
from oracle.gen_golden import CASES

GOLD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def tiny():
    g = np.load(os.path.join(GOLD, "tiny.npz"))
    spec = CASES["tiny"]
    dims = synth.ModelDims(**spec["dims"])
    cfg = O.OracleConfig(**spec["dims"])
    w = O.synthetic_weights(cfg, spec["wseed"])
    prob = synth.make_problem(spec["pseed"], spec["n"], dims, tok_per_clip=spec["tok_per_clip"], text_len=spec["text_len"])
    m = O.OracleModel(cfg, w)
    m.set_tvg_prefix_length(prob.tvg_prefix_length)
    vtg = O.padding_ids(prob.vtg_ids, prob.vtg_labels, prob.vtg_masks, synth.PAD_ID)
    tvg = O.padding_ids(prob.tvg_ids, prob.tvg_labels, prob.tvg_masks, synth.PAD_ID)
    return dict(g=g, spec=spec, dims=dims, cfg=cfg, w=w, prob=prob, m=m, vtg=vtg, tvg=tvg)


def test_synth_statements_agree():
    a = synth.tensor(3, "layers.0.q_proj.w", (64, 32), 0.02)
    b = synth_np.tensor(3, "layers.0.q_proj.w", (64, 32), 0.02)
    assert np.array_equal(a, b)
    assert abs(float(synth.bell_f32(1, "x", 200000, 0.02).std()) - 0.02) < 2e-4
    assert np.array_equal(synth.bf16_bits(a), synth_np.to_bf16_bits(b))


def test_padding_ids(tiny):
    g = tiny["g"]
    for k, a in zip(("ids", "labels", "masks"), tiny["vtg"]):
        assert np.array_equal(a, g[f"pad_vtg_{k}"])
    for k, a in zip(("ids", "labels", "masks"), tiny["tvg"]):
        assert np.array_equal(a, g[f"pad_tvg_{k}"])


@pytest.mark.parametrize("kind", ["vtg", "tvg"])
def test_prepare_and_forward(tiny, kind):
    g, m, prob = tiny["g"], tiny["m"], tiny["prob"]
    ids, lab, msk = tiny[kind]
    sel = [0, 1, 2]
    mask, cpn, emb, lab2 = m.prepare_inputs_labels_for_multimodal(ids[sel], msk[sel], lab[sel], [prob.video[i] for i in sel], tvg=(kind == "tvg"))
    assert np.array_equal(mask, g[f"prep_{kind}_mask"])
    assert np.array_equal(cpn, g[f"prep_{kind}_cpn_mask"])
    assert np.array_equal(lab2, g[f"prep_{kind}_labels"])
    np.testing.assert_allclose(emb, g[f"prep_{kind}_embeds"], atol=1e-6)
    for tag, mm in (("", mask), ("_cpn", cpn)):
        h = m.forward_hidden(emb, mm)
        np.testing.assert_allclose(h, g[f"fwd_{kind}{tag}_hidden"], atol=2e-5)
        if kind == "vtg":
            np.testing.assert_allclose(m.label_logprobs(h, lab2), g[f"fwd_{kind}{tag}_score"], rtol=1e-5)
            if tag == "":
                logits, _ = m.forward(emb[:1], mm[:1])
                np.testing.assert_allclose(logits[0, g["fwd_vtg_logits_row0_pos"]][:, ::997], g["fwd_vtg_logits_row0_sub"], atol=2e-5)
                np.testing.assert_allclose(O.vtg_criterion(logits, lab2[:1]), g["fwd_vtg_score"][:1], rtol=1e-5)


PASSES = [("v2t_vtg", "v2t", "vtg", False), ("v2t_vtg_cpn", "v2t", "vtg", True), ("v2t_tvg", "v2t", "tvg", False),
          ("t2v_vtg", "t2v", "vtg", False), ("t2v_tvg", "t2v", "tvg", False), ("t2v_tvg_cpn", "t2v", "tvg", True)]


@pytest.mark.parametrize("name,direction,ftype,cpn", PASSES)
def test_six_passes(tiny, name, direction, ftype, cpn):
    g, m, prob, spec, dims = tiny["g"], tiny["m"], tiny["prob"], tiny["spec"], tiny["dims"]
    ids, lab, msk = tiny[ftype]
    n = spec["n"]
    fn = O.compute_v2t_scores_x if direction == "v2t" else O.compute_t2v_scores_x
    sims = prob.v2t_sims if direction == "v2t" else prob.t2v_sims
    S = fn(np.full((n, n), -100.0, dtype=np.float32), sims, 0, ids, msk, lab, prob.video, prob.video_vocab, prob.tvg_video_labels, m,
           spec["topk"], spec["bs"], dims.num_clips, ftype, cpn)
    G = g[f"S_{name}"]
    assert np.array_equal(S == -100.0, G == -100.0)
    np.testing.assert_allclose(S, G, rtol=1e-5)


def test_row_truncation_at_tokenizer_model_max_length(tiny):
    """modeling_videochat_flash.py:452-457: spliced rows longer than config.tokenizer_model_max_length lose their tail -- here 1 - 5 response
    tokens of four VTG rows (oracle/gen_golden_truncate.py); the prepared tensors and all six passes against the reference's."""
    g = np.load(os.path.join(GOLD, "truncate.npz"))
    prob, spec, dims = tiny["prob"], tiny["spec"], tiny["dims"]
    m = O.OracleModel(tiny["cfg"], tiny["w"])
    m.set_tvg_prefix_length(prob.tvg_prefix_length)
    m.tokenizer_model_max_length = int(g["limit"])
    ids, lab, msk = tiny["vtg"]
    mask, cpn, emb, lab2 = m.prepare_inputs_labels_for_multimodal(ids, msk, lab, prob.video, tvg=False)
    assert emb.shape[1] == int(g["limit"])
    assert np.array_equal(mask, g["prep_vtg_mask"]) and np.array_equal(cpn, g["prep_vtg_cpn_mask"]) and np.array_equal(lab2, g["prep_vtg_labels"])
    assert (lab2 != -100).sum() < (np.asarray(lab) != -100).sum()                      # the limit did cut labels
    np.testing.assert_allclose(emb, g["prep_vtg_embeds"], atol=1e-6)
    for tag, mm in (("", mask), ("_cpn", cpn)):
        np.testing.assert_allclose(m.label_logprobs(m.forward_hidden(emb, mm), lab2), g[f"fwd_vtg{tag}_score"], rtol=1e-5)
    n = spec["n"]
    for name, direction, ftype, c in PASSES:
        i_, l_, m_ = tiny[ftype]
        fn = O.compute_v2t_scores_x if direction == "v2t" else O.compute_t2v_scores_x
        sims = prob.v2t_sims if direction == "v2t" else prob.t2v_sims
        S = fn(np.full((n, n), -100.0, dtype=np.float32), sims, 0, i_, m_, l_, prob.video, prob.video_vocab, prob.tvg_video_labels, m,
               spec["topk"], spec["bs"], dims.num_clips, ftype, c)
        G = g[f"S_{name}"]
        assert np.array_equal(S == -100.0, G == -100.0), name
        np.testing.assert_allclose(S, G, rtol=2e-5, err_msg=name)
    # the limit changes the VTG scores of the cut rows and nothing else
    t = tiny["g"]
    assert not np.allclose(g["S_v2t_vtg"], t["S_v2t_vtg"], rtol=1e-4) and np.array_equal(g["S_v2t_tvg"], t["S_v2t_tvg"])


def test_criteria(tiny):
    g = tiny["g"]
    np.testing.assert_allclose(O.vtg_criterion(g["crit_vtg_logits"], g["crit_vtg_labels"]), g["crit_vtg_out"], rtol=1e-5)
    np.testing.assert_allclose(O.tvg_criterion(g["crit_tvg_logits"], g["crit_tvg_labels"]), g["crit_tvg_out"], rtol=1e-5)


def test_recall(tiny):
    g = tiny["g"]
    rec = O.get_recall(g["recall_t2v"], g["recall_v2t"])
    assert [rec[k] for k in g["recall_keys"]] == list(g["recall_vals"])
    bz = g["recall_v2t"].copy(); bz[3, 4] = 0.0
    rec0 = O.get_recall(g["recall_t2v"], bz)
    assert [rec0[k] for k in g["recall_keys"]] == list(g["recall_zero_vals"])


def test_cpn_prior_is_query_independent(tiny):
    """v2t VTG-CPN scores depend on the text only (what the fused path exploits)."""
    G = tiny["g"]["S_v2t_vtg_cpn"]
    for col in range(G.shape[1]):
        vals = G[:, col][G[:, col] != -100.0]
        if len(vals) > 1:
            assert np.ptp(vals) < 1e-5


@pytest.mark.needs_reference
def test_reference_importable_and_agrees_on_one_layer(tiny):
    """Build container only: run the reference itself on one ragged batch and compare with the oracle."""
    import torch
    from oracle import ref_harness
    m, prob, cfg, w = tiny["m"], tiny["prob"], tiny["cfg"], tiny["w"]
    ref = ref_harness.build_model(cfg, w)
    ids, lab, msk = tiny["vtg"]
    sel = [1, 4]
    mask, cpn, emb, lab2 = m.prepare_inputs_labels_for_multimodal(ids[sel], msk[sel], lab[sel], [prob.video[i] for i in sel])
    T = lambda a: torch.from_numpy(np.asarray(a))
    with torch.no_grad():
        # the reference's own sequence assembly (it also sets model.llm_compress_layer_list, which forward reads)
        r = ref.prepare_inputs_labels_for_multimodal(T(ids[sel]), None, T(msk[sel]), None, T(lab[sel]), [T(prob.video[i]) for i in sel],
                                                     ["video"] * 2, image_sizes=None, video_feature=True, cpn=True)
        (_, _, (m_r, c_r), _, e_r, l_r) = r
        assert np.array_equal(m_r.numpy(), mask) and np.array_equal(c_r.numpy(), cpn) and np.array_equal(l_r.numpy(), lab2)
        np.testing.assert_allclose(e_r.numpy(), emb, atol=1e-6)
        out = ref(inputs_embeds=e_r, attention_mask=m_r)
    np.testing.assert_allclose(m.forward_hidden(emb, mask), out.hidden_states.numpy(), atol=2e-5)


def test_oracle_at_depth_28_layers_h1024():
    """The restatement at DEPTH: 28 layers (H=1024) against tests/golden/deep.npz, which the reference itself produced
    (oracle/gen_golden.py --case deep): one reference-shaped pass and the headline SYN rows (96 video + 32 text tokens, top-16)."""
    from oracle.gen_golden import LazyWeights, problem_of
    g = np.load(os.path.join(GOLD, "deep.npz"))
    spec = CASES["deep"]
    dims = synth.ModelDims(**spec["dims"])
    cfg = O.OracleConfig(**spec["dims"])
    m = O.OracleModel(cfg, dict(LazyWeights(dims, spec["wseed"]).items()))
    for prefix, sp, prob, checks in (("S_", spec, problem_of(spec, dims), [("t2v_tvg_cpn", "t2v", "tvg", True, 2)]),
                                      ("SYN_", spec["syn"], problem_of(spec, dims, spec["syn"]), [("v2t_vtg", "v2t", "vtg", False, 1)])):
        m.set_tvg_prefix_length(prob.tvg_prefix_length)
        vtg = O.padding_ids(prob.vtg_ids, prob.vtg_labels, prob.vtg_masks, synth.PAD_ID)
        tvg = O.padding_ids(prob.tvg_ids, prob.tvg_labels, prob.tvg_masks, synth.PAD_ID)
        n = sp["n"]
        for name, direction, ftype, cpn, q in checks:
            ids, lab, msk = vtg if ftype == "vtg" else tvg
            fn = O.compute_v2t_scores_x if direction == "v2t" else O.compute_t2v_scores_x
            sims = (prob.v2t_sims if direction == "v2t" else prob.t2v_sims)[:q]
            S = fn(np.full((n, n), -100.0, dtype=np.float32), sims, 0, ids, msk, lab, prob.video, prob.video_vocab, prob.tvg_video_labels, m,
                   sp["topk"], sp["bs"], dims.num_clips, ftype, cpn)
            G = g[f"{prefix}{name}"]
            assert np.array_equal(S[:q] == -100.0, G[:q] == -100.0)
            np.testing.assert_allclose(S[:q], G[:q], rtol=2e-5)


def test_oracle_on_heavy_tailed_weights_28_layers():
    """The restatement under a trained checkpoint's dynamic ranges (oracle/gen_golden_heavy.py: heavy-tailed norm weights, q / k biases of order one,
    residual channels at 1,190 against an rms of 17) against what the reference computed on the same weights: one VTG row and one TVG-prior row."""
    from oracle.gen_golden_heavy import SPEC, heavy_weights
    g = np.load(os.path.join(GOLD, "heavy.npz"))
    assert float(g["resid_absmax_per_layer"].max()) > 1000 and float(g["resid_rms_per_layer"].max()) < 20
    dims = synth.ModelDims(**SPEC["dims"])
    m = O.OracleModel(O.OracleConfig(**SPEC["dims"]), heavy_weights(dims, SPEC["wseed"]))
    prob = synth.make_problem(SPEC["pseed"], SPEC["n"], dims, tok_per_clip=SPEC["tok_per_clip"], text_len=SPEC["text_len"])
    m.set_tvg_prefix_length(prob.tvg_prefix_length)
    vtg = O.padding_ids(prob.vtg_ids, prob.vtg_labels, prob.vtg_masks, synth.PAD_ID)
    tvg = O.padding_ids(prob.tvg_ids, prob.tvg_labels, prob.tvg_masks, synth.PAD_ID)
    n = SPEC["n"]
    for name, direction, ftype, cpn in (("v2t_vtg", "v2t", "vtg", False), ("t2v_tvg_cpn", "t2v", "tvg", True)):
        ids, lab, msk = vtg if ftype == "vtg" else tvg
        fn = O.compute_v2t_scores_x if direction == "v2t" else O.compute_t2v_scores_x
        sims = (prob.v2t_sims if direction == "v2t" else prob.t2v_sims)[:1]
        S = fn(np.full((n, n), -100.0, dtype=np.float32), sims, 0, ids, msk, lab, prob.video, prob.video_vocab, prob.tvg_video_labels, m,
               SPEC["topk"], SPEC["bs"], dims.num_clips, ftype, cpn)
        G = g[f"S_{name}"]
        assert np.array_equal(S[:1] == -100.0, G[:1] == -100.0)
        np.testing.assert_allclose(S[:1], G[:1], rtol=5e-5, err_msg=name)


def test_torch_port_of_the_layer_agrees_with_the_numpy_oracle(tiny):
    """oracle/torch_port.py (the threaded CPU statement bench.py times) == oracle/blim_oracle.py on one layer + the LSE head."""
    import torch
    from oracle import torch_port as TP
    m, prob, cfg = tiny["m"], tiny["prob"], tiny["cfg"]
    ids, lab, msk = tiny["vtg"]
    sel = [0, 3, 5]
    mask, cpn, emb, lab2 = m.prepare_inputs_labels_for_multimodal(ids[sel], msk[sel], lab[sel], [prob.video[i] for i in sel])
    L = emb.shape[1]
    w = {k: torch.from_numpy(v) for k, v in m.w.items()}
    cos, sin = O.rope_tables(cfg.head_dim, cfg.rope_theta, L)
    tc, ts = TP.rope_tables(cfg.head_dim, cfg.rope_theta, L)
    np.testing.assert_allclose(tc.numpy(), cos, atol=1e-5); np.testing.assert_allclose(ts.numpy(), sin, atol=1e-5)     # f32 angles up to ~60 rad
    for mm in (mask, cpn):
        want = m.decoder_layer(0, emb, O.additive_mask(mm, L), cos, sin)
        got = TP.decoder_layer(torch.from_numpy(emb), w, "layers.0.", TP.additive_mask(torch.from_numpy(mm), L), tc, ts, cfg.num_heads, cfg.num_kv_heads, cfg.rms_eps)
        valid = mask.astype(bool)
        np.testing.assert_allclose(got.numpy()[valid], want[valid], atol=2e-5)
    rows = torch.from_numpy(want[0, :5])
    labels = torch.tensor([5, 17, 151644, 3, 99])
    lp = TP.label_logprobs(rows, w["lm_head"], labels).numpy()
    ref = O.log_softmax(want[0, :5] @ m.w["lm_head"].T)[np.arange(5), labels.numpy()]
    np.testing.assert_allclose(lp, ref, rtol=1e-5)


def test_bench_plan_fixture_covers_the_benched_problem():
    """tests/golden/full7b_bench.npz (the reference's loops on the problem bench.py's first plan is built from): four query rows per direction, each
    with exactly the top-16 candidates of the synthetic first-stage matrices, finite log-likelihoods; and the v2t / t2v VTG matrices agree where both
    directions score the same (video, text) pair -- the same forward in the reference."""
    import torch
    g = np.load(os.path.join(GOLD, "full7b_bench.npz"))
    dims = synth.ModelDims()
    prob = synth.make_problem(1000, 55, dims, tok_per_clip=24, text_len=(32, 32), reference_layout=False)
    for name, sims in (("SYN_v2t_vtg", prob.v2t_sims), ("SYN_t2v_vtg", prob.t2v_sims), ("SYN_v2t_tvg", prob.v2t_sims), ("SYN_t2v_tvg", prob.t2v_sims),
                       ("SYN_t2v_tvg_cpn", prob.t2v_sims)):
        S = g[name]
        assert S.shape == (55, 55) and (S[4:] == -100.0).all()
        top = torch.from_numpy(sims[:4]).topk(16, dim=1).indices.numpy()
        for q in range(4):
            assert sorted(np.nonzero(S[q] != -100.0)[0].tolist()) == sorted(top[q].tolist())
            assert np.isfinite(S[q, top[q]]).all() and (S[q, top[q]] < 0).all()
    a, b = g["SYN_v2t_vtg"], g["SYN_t2v_vtg"].T                       # both indexed (video, text)
    both = (a != -100.0) & (b != -100.0)
    assert both.sum() >= 1 and np.allclose(a[both], b[both], rtol=1e-6)


def test_lora_adapted_scoring_six_passes():
    """tests/golden/lora_tiny.npz = the reference's own loops on a model with NON-ZERO LoRA adapters kept apart (oracle/gen_golden_lora.py: the
    `--eval --resume` flow of main.py:96-105, 125-128).  The oracle on W + (alpha / r) B A merged in fp32 must give the same six matrices:
    pins the merge rule (scaling, A / B orientation, which modules are adapted, visual_head from the resume file, tvg_mlp = copy of mlp)."""
    import lora_fixture as LF
    spec, g, dims, prob = LF.load_case("lora_tiny")
    w = LF.merged_fp32(LF.base_weights_host(spec, dims), LF.trainable_of(spec, dims))
    m = O.OracleModel(O.OracleConfig(**spec["dims"]), w)
    m.set_tvg_prefix_length(prob.tvg_prefix_length)
    vtg = O.padding_ids(prob.vtg_ids, prob.vtg_labels, prob.vtg_masks, synth.PAD_ID)
    tvg = O.padding_ids(prob.tvg_ids, prob.tvg_labels, prob.tvg_masks, synth.PAD_ID)
    n = spec["n"]
    base = np.load(os.path.join(GOLD, "tiny.npz"))        # the same problem on the base weights (tvg_mlp differs there: VTG passes only)
    for name, direction, ftype, c in LF.PASSES:
        ids, lab, msk = vtg if ftype == "vtg" else tvg
        fn = O.compute_v2t_scores_x if direction == "v2t" else O.compute_t2v_scores_x
        sims = prob.v2t_sims if direction == "v2t" else prob.t2v_sims
        S = fn(np.full((n, n), -100.0, dtype=np.float32), sims, 0, ids, msk, lab, prob.video, prob.video_vocab, prob.tvg_video_labels, m,
               spec["topk"], spec["bs"], dims.num_clips, ftype, c)
        G = g[f"S_{name}"]
        assert np.array_equal(S == -100.0, G == -100.0), name
        np.testing.assert_allclose(S, G, rtol=2e-5, err_msg=name)
        if ftype == "vtg":      # the adapters are not a no-op: the scores moved
            on = G != -100.0
            assert np.max(np.abs(G[on] - base[f"S_{name}"][on]) / np.abs(G[on])) > 1e-3, name
