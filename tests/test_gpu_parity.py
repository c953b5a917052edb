"""GPU (-m gpu): the HIP engine, called through the C ABI, against the numpy oracle and the golden vectors recorded
from the reference.

Tolerances.  Scores: 1e-3 relative per entry (BASELINE.json north_star) for every pass kind at every size, including all 28 layers of
the real 7B configuration (test_depth_*), in BOTH 16-bit engine dtypes on the fused path evaluation() runs:
* fp16 -- the engine's default, the reference's own autocast dtype and the dtype bench.py runs in: plain 16-bit VTG calls, TVG calls
  (scores ~10x smaller in magnitude) in the compensated mode (hi + lo activations, engine option "precise");
* bf16 -- the dtype BASELINE.json's configurations name: 8-bit mantissas miss the bar when plain (1.0 - 1.7e-3 VTG, 7 - 9e-3 TVG at 7B
  depth), so since round 3 bf16 engines run every call compensated (hi + lo bf16 = 16 significant bits against exact bf16 weights:
  VTG 2e-6, TVG <= 7e-4 at 7B depth; modeling.py: vtg_precise = "full").
The literal reference-shaped API on a bf16 engine hands its spliced embeddings out as one float32 [B, L, H] tensor (hi + lo formed in the compensated
mode; a bf16 tensor between prepare_inputs_labels_for_multimodal and forward() would by itself put ~1e-3 on the scores), so it is held to the same 1e-3.
Intermediate 16-bit tensors: 2e-2 of the tensor's max."""
import os
import types

import numpy as np
import pytest
import torch

from blim_amd import engine as eng
from blim_amd import retrieval_utils as RU
from blim_amd import synth
from blim_amd import training_utils as TU
from blim_amd.modeling import BlimModel, DDPLike
from oracle import blim_oracle as O
from oracle.gen_golden import CASES

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SCORE_RTOL = 1e-3


def h16(x, dtype):
    """float32 numpy -> torch tensor of the engine's 16-bit compute dtype (round to nearest even)."""
    if dtype == torch.bfloat16:
        return torch.from_numpy(synth.bf16_bits(np.asarray(x, np.float32)).view(np.int16)).view(torch.bfloat16)
    return torch.from_numpy(np.asarray(x, np.float32)).to(torch.float16)


DTYPES = ["f16", "bf16"]


def score_rtol(dtype: str, pass_name: str, literal: bool = False) -> float:
    return SCORE_RTOL          # one bar for both 16-bit dtypes, fused and literal paths (round 3)


def relmax(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))


def _build(case, layers=None, device_synth=False, dtype="f16"):
    spec = CASES[case]
    d = dict(spec["dims"])
    if layers is not None:
        d["num_layers"] = layers
    dims = synth.ModelDims(**d)
    model = BlimModel(dims, max_positions=1024, dtype=dtype)
    w = None
    if device_synth:
        model.engine.init_synthetic_weights(spec["wseed"])
    else:
        w = synth.synthetic_weights(dims, spec["wseed"])
        model.engine.load_weights(w)
    prob = synth.make_problem(spec["pseed"], spec["n"], dims, tok_per_clip=spec["tok_per_clip"], text_len=spec["text_len"])
    model.set_tvg_prefix_length(prob.tvg_prefix_length)
    return types.SimpleNamespace(spec=spec, dims=dims, model=model, w=w, prob=prob, d=d, dtype=dtype, case=case)


@pytest.fixture(scope="module", params=DTYPES)
def tiny(request):
    t = _build("tiny", dtype=request.param)
    yield t
    t.model.engine.close()


@pytest.fixture(scope="module", params=DTYPES)
def tiny1(request):
    t = _build("tiny", layers=1, dtype=request.param)
    yield t
    t.model.engine.close()


@pytest.fixture(scope="module", params=DTYPES)
def wide(request):
    t = _build("wide", device_synth=True, dtype=request.param)
    yield t
    t.model.engine.close()


def test_native_library_is_the_thing_under_test():
    lib = eng.load_library()
    assert os.path.samefile(lib._name, eng.LIB_PATH)
    assert torch.cuda.is_available()


def test_device_fill_is_bit_exact_with_the_numpy_rule():
    for n, std, mean in ((1000, 0.02, 0.0), (4099, 0.1, 1.0), (7, 1.0, 0.0)):
        out = torch.empty(n, dtype=torch.bfloat16, device="cuda")
        eng.fill_bell_bf16(out, 7, "layers.3.q_proj.w", std, mean)
        got = out.view(torch.int16).cpu().numpy().view(np.uint16)
        assert np.array_equal(got, synth.bf16_bits(synth.bell_f32(7, "layers.3.q_proj.w", n, std, mean)))


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16], ids=DTYPES)
@pytest.mark.parametrize("M,N,K", [(256, 256, 64), (1, 4, 64), (300, 500, 128), (1000, 260, 256), (513, 1028, 3584), (2048, 512, 18944)])
def test_gemm_vs_numpy(M, N, K, dtype):
    rs = np.random.RandomState(M + N)
    a = synth.bf16_round(rs.randn(M, K).astype(np.float32)); w = synth.bf16_round(rs.randn(N, K).astype(np.float32) * 0.05)
    got = eng.gemm_bf16(h16(a, dtype).cuda(), h16(w, dtype).cuda()).float().cpu().numpy()
    want = a @ w.T
    assert relmax(got, want) < (6e-3 if dtype == torch.bfloat16 else 1e-3)        # one 16-bit rounding of the output


def test_gemm_row_chunking_beyond_4gib():
    """An A operand of >= 4 GiB (32-bit operand offsets inside the kernel) is processed as row chunks: rows on both sides of the
    chunk boundary and the last rows agree with a GEMM on just those rows."""
    M, N, K = 66000, 512, 32768                      # 66000 x 65536 B = 4.03 GiB; chunk boundary at row 65280
    g = torch.Generator(device="cuda").manual_seed(1)
    a = (torch.randn((M, K), generator=g, device="cuda", dtype=torch.float32) * 0.5).to(torch.bfloat16)
    w = (torch.randn((N, K), generator=g, device="cuda", dtype=torch.float32) * 0.05).to(torch.bfloat16)
    out = eng.gemm_bf16(a, w)
    for lo, hi in ((0, 256), (65280 - 128, 65280 + 128), (M - 300, M)):
        part = eng.gemm_bf16(a[lo:hi].contiguous(), w)
        assert torch.equal(out[lo:hi], part), (lo, hi)
        ref = a[lo:hi].float() @ w.float().T
        assert relmax(part.float().cpu().numpy(), ref.cpu().numpy()) < 6e-3


def test_engine_batch_beyond_113k_tokens(wide):
    """--max_tokens above ~113 k (the [T, I] SwiGLU output reaches 4 GiB at 7B width): one packed batch of 120,000 tokens; the
    sequences that straddle the 4-GiB row boundary give the hidden states they give in a small batch."""
    t = wide
    E = t.model.engine
    Ls, n_seq = 100, 1200
    T = Ls * n_seq
    g = torch.Generator(device="cuda").manual_seed(5)
    emb = (torch.randn((T, t.dims.hidden_size), generator=g, device="cuda") * 0.02).to(E.torch_dtype)
    mk = lambda n: eng.PackedBatch(np.tile(np.arange(Ls, dtype=np.int32), n), np.ones(Ls * n, np.uint8), np.arange(n, dtype=np.int32) * Ls, np.full(n, Ls, np.int32))
    big, _ = E.decode(mk(n_seq), emb)
    for s0 in (0, 1132, 1133, n_seq - 1):             # 4 GiB / (18944 * 2 B) = row 113,359 -> inside sequence 1133
        small, _ = E.decode(mk(1), emb[s0 * Ls:(s0 + 1) * Ls].contiguous())
        assert torch.isfinite(small.float()).all()
        assert torch.equal(big[s0 * Ls:(s0 + 1) * Ls], small), s0


def test_layer_stages_against_oracle(tiny1):
    """QKV+bias+RoPE, attention (key mask incl. CPN), SwiGLU and both residual GEMMs, one stage at a time."""
    t = tiny1
    ocfg = O.OracleConfig(**t.d)
    om = O.OracleModel(ocfg, t.w); om.set_tvg_prefix_length(t.prob.tvg_prefix_length)
    vtg = O.padding_ids(t.prob.vtg_ids, t.prob.vtg_labels, t.prob.vtg_masks, synth.PAD_ID)
    sel = [0, 1, 2]
    mask, cpn, emb, lab = om.prepare_inputs_labels_for_multimodal(vtg[0][sel], vtg[2][sel], vtg[1][sel], [t.prob.video[i] for i in sel])
    B, L, H = emb.shape
    E = t.model.engine
    e_t = h16(emb, t.model.dtype).cuda()
    nq, nk = t.dims.num_heads * 128, t.dims.num_kv_heads * 128
    for mm in (mask, cpn):
        parts = {}
        cos, sin = O.rope_tables(ocfg.head_dim, ocfg.rope_theta, L)
        x1 = om.decoder_layer(0, h16(emb, t.model.dtype).float().numpy(), O.additive_mask(mm, L), cos, sin, parts)
        for tr in (1, 0):
            E.set_option("attn_tr_read", tr)
            _, hd = E.forward(e_t, torch.from_numpy(mm.astype(np.uint8)).cuda(), want_logits=False, want_hidden=True)
            valid = mask.astype(bool)
            qkv = E.debug_read("qkv", (B * L, nq + 2 * nk), t.model.dtype).float().cpu().numpy().reshape(B, L, -1)
            assert relmax(qkv[..., :nq][valid], parts["q"][valid]) < 2e-2
            assert relmax(qkv[..., nq:nq + nk][valid], parts["k"][valid]) < 2e-2
            assert relmax(qkv[..., nq + nk:][valid], parts["v"][valid]) < 2e-2
            at = E.debug_read("attn", (B * L, H), t.model.dtype).float().cpu().numpy().reshape(B, L, H)
            assert relmax(at[valid], parts["attn"][valid]) < 2e-2
            ac = E.debug_read("act", (B * L, t.dims.intermediate_size), t.model.dtype).float().cpu().numpy().reshape(B, L, -1)
            assert relmax(ac[valid], parts["act"][valid]) < 2e-2
            rs = E.debug_read("resid", (B * L, H), torch.float32).cpu().numpy().reshape(B, L, H)
            assert relmax(rs[valid], x1[valid]) < 1e-2
            assert relmax(hd.cpu().numpy()[valid], O.rms_norm(x1, om.w["final_norm"], ocfg.rms_eps)[valid]) < 1e-2
    E.set_option("attn_tr_read", 1)


def test_projector_and_sequence_assembly(tiny):
    t = tiny
    g = np.load(os.path.join(GOLD, "tiny.npz"))
    T = lambda a: torch.from_numpy(np.asarray(a)).cuda()
    sel = [0, 1, 2]
    for kind in ("vtg", "tvg"):
        r = t.model.prepare_inputs_labels_for_multimodal(T(g[f"pad_{kind}_ids"][sel]), None, T(g[f"pad_{kind}_masks"][sel]), None,
                                                         T(g[f"pad_{kind}_labels"][sel]), [T(t.prob.video[i]) for i in sel], ["video"] * 3,
                                                         image_sizes=None, video_feature=True, tvg=(kind == "tvg"), cpn=True)
        (none0, pos, (m_t, c_t), pkv, e_t, l_t) = r
        assert none0 is None and pos is None and pkv is None
        assert np.array_equal(m_t.cpu().numpy(), g[f"prep_{kind}_mask"]) and m_t.dtype == torch.long
        assert np.array_equal(c_t.cpu().numpy(), g[f"prep_{kind}_cpn_mask"])
        assert np.array_equal(l_t.cpu().numpy(), g[f"prep_{kind}_labels"])
        assert relmax(e_t.float().cpu().numpy(), g[f"prep_{kind}_embeds"]) < 1e-2
        # cpn=False returns the plain mask, not the pair (modeling_videochat_flash.py:512-515)
        r2 = t.model.prepare_inputs_labels_for_multimodal(T(g[f"pad_{kind}_ids"][sel]), None, T(g[f"pad_{kind}_masks"][sel]), None,
                                                          T(g[f"pad_{kind}_labels"][sel]), [T(t.prob.video[i]) for i in sel], ["video"] * 3,
                                                          video_feature=True, tvg=(kind == "tvg"))
        assert torch.equal(r2[2], m_t)


def test_forward_surface_and_hidden_states(tiny):
    t = tiny
    g = np.load(os.path.join(GOLD, "tiny.npz"))
    ddp = DDPLike(t.model).eval()
    for kind in ("vtg", "tvg"):
        emb = h16(g[f"prep_{kind}_embeds"], t.model.dtype).cuda()
        valid = g[f"prep_{kind}_mask"].astype(bool)
        for tag, mk in (("", f"prep_{kind}_mask"), ("_cpn", f"prep_{kind}_cpn_mask")):
            out = ddp(inputs_embeds=emb, attention_mask=torch.from_numpy(g[mk]).cuda())
            assert out.logits.dtype == torch.float32 and out.logits.shape == (3, emb.shape[1], t.dims.vocab_size)
            assert out.hidden_states.shape == emb.shape
            assert relmax(out.hidden_states.cpu().numpy()[valid], g[f"fwd_{kind}{tag}_hidden"][valid]) < 1e-2
            if kind == "vtg":
                sc = RU.vtg_criterion(out.logits, torch.from_numpy(g["prep_vtg_labels"]).cuda()).cpu().numpy()
                np.testing.assert_allclose(sc, g[f"fwd_vtg{tag}_score"], rtol=SCORE_RTOL)
                if tag == "":
                    pos = torch.from_numpy(g["fwd_vtg_logits_row0_pos"]).cuda()
                    assert relmax(out.logits[0, pos][:, ::997].cpu().numpy(), g["fwd_vtg_logits_row0_sub"]) < 1e-2
    with pytest.raises(NotImplementedError):
        t.model(input_ids=torch.zeros((1, 4), dtype=torch.long))
    with pytest.raises(NotImplementedError):
        t.model(inputs_embeds=emb, labels=torch.zeros((3, 4), dtype=torch.long))


def test_gather_rows_outside_the_batch_poison_the_score_instead_of_faulting(tiny):
    """C-ABI robustness: a label-row index outside [0, n_tokens) must not read wild memory; its output row is NaN."""
    t = tiny
    L, Hd = 12, t.dims.hidden_size
    batch = eng.PackedBatch(np.arange(L, dtype=np.int32), np.ones(L, np.uint8), np.array([0], np.int32), np.array([L], np.int32))
    emb = (torch.randn((L, Hd), device="cuda") * 0.02).to(t.model.engine.torch_dtype)
    rows = torch.tensor([0, -1, L - 1, L, 1 << 30], dtype=torch.int32, device="cuda")
    hid, _ = t.model.engine.decode(batch, emb, out_rows=rows)
    torch.cuda.synchronize()
    bad = torch.isnan(hid.float()).all(dim=1).cpu().numpy()
    assert bad.tolist() == [False, True, False, True, True]
    ref, _ = t.model.engine.decode(batch, emb)
    assert torch.equal(hid[0], ref[0]) and torch.equal(hid[2], ref[L - 1])


def test_criteria_on_golden_logits():
    g = np.load(os.path.join(GOLD, "tiny.npz"))
    sc = RU.vtg_criterion(torch.from_numpy(g["crit_vtg_logits"]).cuda(), torch.from_numpy(g["crit_vtg_labels"]).cuda()).cpu().numpy()
    np.testing.assert_allclose(sc, g["crit_vtg_out"], rtol=1e-5)
    sc = RU.tvg_criterion(torch.from_numpy(g["crit_tvg_logits"]).cuda(), torch.from_numpy(g["crit_tvg_labels"]).cuda()).cpu().numpy()
    np.testing.assert_allclose(sc, g["crit_tvg_out"], rtol=1e-5)


PASSES = [("v2t_vtg", True, "vtg", False), ("v2t_vtg_cpn", True, "vtg", True), ("v2t_tvg", True, "tvg", False),
          ("t2v_vtg", False, "vtg", False), ("t2v_tvg", False, "tvg", False), ("t2v_tvg_cpn", False, "tvg", True)]


def _six_passes(t, literal, prob=None, spec=None, names=None, max_tokens=4096):
    """Scores `names` (default: all six pass kinds) for the first spec['queries'] query rows of `prob` through the literal
    reference-shaped API or the fused PairScorer; returns {pass name: [n, n] matrix, -100 where nothing was computed}."""
    prob = prob or t.prob
    spec = spec or t.spec
    ddp = DDPLike(t.model)
    dev = t.model.device
    t.model.set_tvg_prefix_length(prob.tvg_prefix_length)
    tok = types.SimpleNamespace(pad_token_id=synth.PAD_ID)
    Tt = lambda rows: [torch.from_numpy(r) for r in rows]
    vtg = RU.padding_ids(Tt(prob.vtg_ids), Tt(prob.vtg_labels), Tt(prob.vtg_masks), tok)
    tvg = RU.padding_ids(Tt(prob.tvg_ids), Tt(prob.tvg_labels), Tt(prob.tvg_masks), tok)
    video = [torch.from_numpy(v) for v in prob.video]
    vocab = torch.from_numpy(prob.video_vocab); vlab = torch.from_numpy(prob.tvg_video_labels)
    n, q = spec["n"], spec.get("queries", spec["n"])
    args = types.SimpleNamespace(topk=spec["topk"], batch_size_eval=spec["bs"], num_clips=t.dims.num_clips)
    out = {}
    scorer = None if literal else RU.PairScorer(ddp, vtg[0], vtg[2], vtg[1], tvg[0], tvg[2], tvg[1], video, vocab, vlab, t.dims.num_clips, max_tokens=max_tokens)
    for name, qv, ft, cpn in PASSES:
        if names is not None and name not in names:
            continue
        sims = torch.from_numpy(prob.v2t_sims if qv else prob.t2v_sims)[:q]
        S = torch.full((n, n), -100.0, device=dev)
        if literal:
            fn = RU.compute_v2t_scores_x if qv else RU.compute_t2v_scores_x
            ids, lab, msk = vtg if ft == "vtg" else tvg
            S = fn(S, sims, 0, ids, msk, lab, video, vocab.to(dev), vlab, ddp, dev, args, forward_type=ft, cpn=cpn)
        else:
            pairs = RU._topk_pairs(sims, 0, args.topk, qv)
            sc = scorer.vtg(pairs, cpn) if ft == "vtg" else scorer.tvg(pairs, cpn)
            r, c = (pairs[:, 0], pairs[:, 1]) if qv else (pairs[:, 1], pairs[:, 0])
            S[torch.from_numpy(r).to(dev), torch.from_numpy(c).to(dev)] = torch.from_numpy(sc).to(dev)
        out[name] = S.cpu().numpy()
    return out


def _resolve_auto(t, prob=None, spec=None):
    """`--vtg_precise auto` as evaluation() resolves it: PairScorer.calibrate_vtg on the calibration pairs of the problem -> (mode, table)."""
    prob = prob or t.prob
    spec = spec or t.spec
    t.model.set_tvg_prefix_length(prob.tvg_prefix_length)
    tok = types.SimpleNamespace(pad_token_id=synth.PAD_ID)
    Tt = lambda rows: [torch.from_numpy(r) for r in rows]
    vtg = RU.padding_ids(Tt(prob.vtg_ids), Tt(prob.vtg_labels), Tt(prob.vtg_masks), tok)
    tvg = RU.padding_ids(Tt(prob.tvg_ids), Tt(prob.tvg_labels), Tt(prob.tvg_masks), tok)
    t.model.vtg_precise = "auto"
    sc = RU.PairScorer(DDPLike(t.model), vtg[0], vtg[2], vtg[1], tvg[0], tvg[2], tvg[1], [torch.from_numpy(v) for v in prob.video], torch.from_numpy(prob.video_vocab),
                       torch.from_numpy(prob.tvg_video_labels), t.dims.num_clips)
    cal = RU.calibration_pairs(torch.from_numpy(prob.v2t_sims), spec["topk"], n_queries=32, per_query=8)
    chosen, table = sc.calibrate_vtg(cal, n_eval=3 * len(cal))          # as evaluation() does: the three VTG-type passes' entries (on these small fixtures the sample is one whole pass)
    assert (t.model.vtg_mode() or "none") == chosen and t.model.vtg_precise == "auto"          # the request stays "auto"; what it resolved to is kept beside it
    return chosen, table


def _resolve_tvg_auto(t, prob=None, spec=None):
    """`--tvg_precise auto` as evaluation() resolves it (PairScorer.calibrate_tvg) -> (mode, table); t.model.tvg_precise is left at the chosen mode."""
    prob = prob or t.prob
    spec = spec or t.spec
    t.model.set_tvg_prefix_length(prob.tvg_prefix_length)
    tok = types.SimpleNamespace(pad_token_id=synth.PAD_ID)
    Tt = lambda rows: [torch.from_numpy(r) for r in rows]
    vtg = RU.padding_ids(Tt(prob.vtg_ids), Tt(prob.vtg_labels), Tt(prob.vtg_masks), tok)
    tvg = RU.padding_ids(Tt(prob.tvg_ids), Tt(prob.tvg_labels), Tt(prob.tvg_masks), tok)
    t.model.tvg_precise = "auto"
    sc = RU.PairScorer(DDPLike(t.model), vtg[0], vtg[2], vtg[1], tvg[0], tvg[2], tvg[1], [torch.from_numpy(v) for v in prob.video], torch.from_numpy(prob.video_vocab),
                       torch.from_numpy(prob.tvg_video_labels), t.dims.num_clips)
    assert sc.tvg_mode == "full"                                           # unresolved auto runs fully compensated
    tp = RU.calibration_pairs(torch.from_numpy(prob.t2v_sims), spec["topk"], n_queries=64, per_query=4)
    chosen, table = sc.calibrate_tvg(np.stack([tp[:, 1], tp[:, 0]], axis=1), n_eval=3 * len(tp))
    assert t.model.tvg_mode() == chosen and t.model.tvg_precise == "auto"
    t.model.tvg_precise = chosen                                           # the callers go on with explicit requests
    return chosen, table


TVG_PASSES = ("t2v_tvg", "t2v_tvg_cpn", "v2t_tvg")


def _worst_rel(got, g, prefix="S_"):
    """{pass: worst relative deviation over the computed entries}; the computed-entry pattern must equal the golden one."""
    worst = {}
    for name, S in got.items():
        G = g[f"{prefix}{name}"]
        assert np.array_equal(S == -100.0, G == -100.0), name
        m = G != -100.0
        assert np.isfinite(S[m]).all(), name
        worst[name] = float((np.abs(S[m].astype(np.float64) - G[m]) / np.abs(G[m])).max())
    return worst


def _check_passes(got, g, t, literal=False):
    for name, S in got.items():
        G = g[f"S_{name}"]
        assert np.array_equal(S == -100.0, G == -100.0), name          # same entries computed (top-k, leftover batch)
        m = G != -100.0
        rtol = score_rtol(t.dtype, name, literal)
        np.testing.assert_allclose(S[m], G[m], rtol=rtol, err_msg=f"{name} [{t.dtype}]")


@pytest.mark.parametrize("literal", [True, False], ids=["literal-api", "fused-pairscorer"])
def test_six_passes_tiny_vs_reference_golden(tiny, literal):
    g = np.load(os.path.join(GOLD, "tiny.npz"))
    got = _six_passes(tiny, literal)
    _check_passes(got, g, tiny, literal)
    # R@k of the full BLiM ensemble: identical to the reference's matrices'
    n = tiny.spec["n"]
    args = types.SimpleNamespace(cpn=True, alpha=[0.4, 0.8], c=[0.3, 0.6, 0.9, 0.7], resume="ckpt", eval=True)
    mk = lambda src, pre: ({"candidate_likelihood": src[f"{pre}t2v_tvg"], "candidate_prior": src[f"{pre}t2v_tvg_cpn"], "query_likelihood": src[f"{pre}t2v_vtg"],
                            "internvideo2": tiny.prob.t2v_sims},
                           {"candidate_likelihood": src[f"{pre}v2t_vtg"], "candidate_prior": src[f"{pre}v2t_vtg_cpn"], "query_likelihood": src[f"{pre}v2t_tvg"],
                            "internvideo2": tiny.prob.v2t_sims})
    assert TU.combine_and_rank(*mk(got, ""), args, n) == TU.combine_and_rank(*mk(g, "S_"), args, n)


@pytest.mark.parametrize("literal", [True, False], ids=["literal-api", "fused-pairscorer"])
def test_rows_cut_at_tokenizer_model_max_length_vs_reference_golden(tiny, literal):
    """modeling_videochat_flash.py:452-457: with config.tokenizer_model_max_length = 64 four of the six VTG rows lose 1 - 5 response tokens
    (oracle/gen_golden_truncate.py ran the reference with that limit).  Prepared masks / labels bit-equal, all six passes within the score bar,
    through the literal API and through the fused planner (which cuts the response before it packs it)."""
    g = np.load(os.path.join(GOLD, "truncate.npz"))
    t = tiny
    t.model.tokenizer_model_max_length = int(g["limit"])
    try:
        if literal:
            g0 = np.load(os.path.join(GOLD, "tiny.npz"))
            T = lambda a: torch.from_numpy(np.asarray(a)).cuda()
            n = t.spec["n"]
            r = t.model.prepare_inputs_labels_for_multimodal(T(g0["pad_vtg_ids"]), None, T(g0["pad_vtg_masks"]), None, T(g0["pad_vtg_labels"]),
                                                             [T(v) for v in t.prob.video], ["video"] * n, image_sizes=None, video_feature=True, tvg=False, cpn=True)
            (_, _, (m_t, c_t), _, e_t, l_t) = r
            assert e_t.shape[1] == int(g["limit"])
            assert np.array_equal(m_t.cpu().numpy(), g["prep_vtg_mask"]) and np.array_equal(c_t.cpu().numpy(), g["prep_vtg_cpn_mask"])
            assert np.array_equal(l_t.cpu().numpy(), g["prep_vtg_labels"])
            assert relmax(e_t.float().cpu().numpy(), g["prep_vtg_embeds"]) < 1e-2
        got = _six_passes(t, literal)
        _check_passes(got, g, t, literal)
        base = np.load(os.path.join(GOLD, "tiny.npz"))
        assert not np.allclose(got["v2t_vtg"], base["S_v2t_vtg"], rtol=5e-3)            # the limit was in force (1.1e-2 on the cut rows)
    finally:
        t.model.tokenizer_model_max_length = None


@pytest.mark.parametrize("literal", [False, True], ids=["fused-pairscorer", "literal-api"])
def test_six_passes_7b_width_vs_reference_golden(wide, literal):
    """Qwen2-7B width (H=3584, 28/4 heads, I=18944, V=152064), one layer; weights generated ON DEVICE from the seed."""
    g = np.load(os.path.join(GOLD, "wide.npz"))
    _check_passes(_six_passes(wide, literal), g, wide, literal)


@pytest.mark.parametrize("dtype", DTYPES)
def test_full_size_properties_7b(dtype):
    """Size-independent properties at the full 28-layer 7B configuration: a pair's score does not depend on what else is in
    the batch (bitwise), nor on the order of the pairs; shared-prefix scoring equals per-pair scoring (1e-3)."""
    dims = synth.ModelDims()
    model = BlimModel(dims, max_positions=1024, dtype=dtype)
    model.engine.init_synthetic_weights(0)
    prob = synth.make_problem(21, 6, dims, tok_per_clip=24, text_len=(5, 32), reference_layout=True)
    model.set_tvg_prefix_length(prob.tvg_prefix_length)
    tok = types.SimpleNamespace(pad_token_id=synth.PAD_ID)
    Tt = lambda rows: [torch.from_numpy(r) for r in rows]
    vtg = RU.padding_ids(Tt(prob.vtg_ids), Tt(prob.vtg_labels), Tt(prob.vtg_masks), tok)
    tvg = RU.padding_ids(Tt(prob.tvg_ids), Tt(prob.tvg_labels), Tt(prob.tvg_masks), tok)
    sc = RU.PairScorer(DDPLike(model), vtg[0], vtg[2], vtg[1], tvg[0], tvg[2], tvg[1], [torch.from_numpy(v) for v in prob.video],
                       torch.from_numpy(prob.video_vocab), torch.from_numpy(prob.tvg_video_labels), dims.num_clips)
    pairs = np.array([[j, i] for j in range(3) for i in range(6)])
    full = sc.vtg(pairs)
    assert np.isfinite(full).all() and (full < 0).all()
    perm = np.random.RandomState(0).permutation(len(pairs))
    assert np.array_equal(sc.vtg(pairs[perm]), full[perm])                       # order invariance, bitwise
    alone = np.array([sc.vtg(pairs[k:k + 1])[0] for k in (0, 7, 17)])
    np.testing.assert_allclose(alone, full[[0, 7, 17]], rtol=SCORE_RTOL)          # batch-composition invariance
    tv = sc.tvg(pairs)
    assert np.array_equal(sc.tvg(pairs[perm]), tv[perm])
    # the v2t prior does not depend on the video
    pr = sc.vtg(pairs, cpn=True).reshape(3, 6)
    assert np.ptp(pr, axis=0).max() == 0.0
    # shared-prefix scoring == the literal per-pair forward (no sharing), through the reference-shaped API
    ddp = DDPLike(model)
    dev = model.device
    j, i = 1, 4
    r = model.prepare_inputs_labels_for_multimodal(vtg[0][[i]].to(dev), None, vtg[2][[i]].to(dev), None, vtg[1][[i]].to(dev),
                                                   [torch.from_numpy(prob.video[j]).to(dev)], ["video"], video_feature=True, cpn=True)
    out = ddp(inputs_embeds=r[4], attention_mask=r[2][0])
    lit = RU.vtg_criterion(out.logits, r[5]).cpu().numpy()[0]
    # (not bitwise: the literal row is one causal segment, the fused one a prefix + own segment, so the 32-key softmax tiles fall differently)
    np.testing.assert_allclose(full[j * 6 + i], lit, rtol=score_rtol(dtype, "vtg", literal=True))
    alone_all = np.array([sc.vtg(pairs[k:k + 1])[0] for k in range(len(pairs))])
    assert np.array_equal(alone_all, full)                                        # batch composition: bitwise
    model.engine.close()


# ----------------------------------------------------------------------------- parity at depth (28 layers)
# tests/golden/{deep,full7b}.npz: the reference itself run in fp32 on CPU at 28 layers (oracle/gen_golden.py) -- H=1024 and the
# real Qwen2-7B configuration (weight seed 0 = bench.py's weights), reference-shaped ragged rows for all six pass kinds plus
# BASELINE.json's headline rows (SYN: 96 video + 32 text tokens, top-16).  Every computed score is compared per entry.


def _depth_case(case, dtype, capsys, literal_too=True):
    from oracle.gen_golden import problem_of
    t = _build(case, device_synth=True, dtype=dtype)
    g = np.load(os.path.join(GOLD, f"{case}.npz"))
    res = {}
    try:
        for literal in ([False, True] if literal_too else [False]):
            tag = "literal" if literal else "fused"
            res[tag] = _worst_rel(_six_passes(t, literal), g)
            if "syn" in t.spec:
                sprob = problem_of(t.spec, t.dims, t.spec["syn"])
                w = _worst_rel(_six_passes(t, literal, prob=sprob, spec=t.spec["syn"], names=t.spec["syn"]["passes"], max_tokens=1 << 16), g, "SYN_")
                res[tag].update({"SYN_" + k: v for k, v in w.items()})
        # final-norm hidden state of one ragged batch, every 16th column (relative to the tensor's max)
        t.model.set_tvg_prefix_length(t.prob.tvg_prefix_length)
        T = lambda a: torch.from_numpy(np.asarray(a)).cuda()
        sel = [0, 1, 2]
        hid = {}
        for kind in ("vtg", "tvg"):
            r = t.model.prepare_inputs_labels_for_multimodal(T(g[f"pad_{kind}_ids"][sel]), None, T(g[f"pad_{kind}_masks"][sel]), None,
                                                             T(g[f"pad_{kind}_labels"][sel]), [T(t.prob.video[i]) for i in sel], ["video"] * 3,
                                                             video_feature=True, tvg=(kind == "tvg"), cpn=True)
            assert np.array_equal(r[2][0].cpu().numpy(), g[f"prep_{kind}_mask"]) and np.array_equal(r[5].cpu().numpy(), g[f"prep_{kind}_labels"])
            valid = g[f"prep_{kind}_mask"].astype(bool)
            for tag, mm in (("", r[2][0]), ("_cpn", r[2][1])):
                out = t.model(inputs_embeds=r[4], attention_mask=mm, want_logits=False)
                hid[f"{kind}{tag}"] = relmax(out.hidden_states.cpu().numpy()[..., ::16][valid], g[f"fwd_{kind}{tag}_hidden_sub16"][valid])
        if dtype == "f16":      # `--vtg_precise auto` (the driver's default): on these weights the plain fp16 VTG calls are kept
            chosen, table = _resolve_auto(t)
            with capsys.disabled():
                print(f"\n[{case} f16] vtg_precise auto (max / rms vs the fully compensated mode): " + ", ".join(f"{k} {v['max']:.1e} / {v['rms']:.1e}" for k, v in table.items()) + f" -> {chosen}")
            assert chosen == "none", (case, table)
            t.model.vtg_precise = None
        # `--tvg_precise auto`: how much of the TVG calls' MLP branch needs compensating, measured; the TVG passes re-run in the chosen mode (both paths) against the golden
        tchosen, ttable = _resolve_tvg_auto(t)
        with capsys.disabled():
            print(f"\n[{case} {dtype}] tvg_precise auto (max / rms vs the fully compensated mode): " + ", ".join(f"{k} {v['max']:.1e} / {v['rms']:.1e}" for k, v in ttable.items()) + f" -> {tchosen}")
        if tchosen != "full":
            for literal in ([False, True] if literal_too else [False]):
                res[("literal" if literal else "fused") + f"-tvg-{tchosen}"] = _worst_rel(_six_passes(t, literal, names=TVG_PASSES), g)
        if dtype == "f16":
            assert tchosen in ("attn", "full"), (case, ttable)            # N(0, 0.02^2) weights: `attn` on the headline-shaped problem (7e-5); on the short ragged rows of
                                                                          # the fixtures it reads 6e-4 max / 2.4e-4 rms -- 4.5 x rms is at the bar, so either outcome is legitimate
        t.model.tvg_precise = "full"
    finally:
        t.model.engine.close()
    with capsys.disabled():
        for tag, w in res.items():
            print(f"\n[{case} {dtype} {tag}] worst relative score deviation vs the fp32 reference, 28 layers: " + ", ".join(f"{k} {v:.2e}" for k, v in w.items()))
        print(f"[{case} {dtype}] final hidden state, max abs error / max: " + ", ".join(f"{k} {v:.2e}" for k, v in hid.items()))
    return res, hid


@pytest.mark.parametrize("dtype", DTYPES)
def test_depth_28_layers_h1024_vs_reference_golden(dtype, capsys):
    res, hid = _depth_case("deep", dtype, capsys)
    for tag, w in res.items():
        for k, v in w.items():
            assert v < score_rtol(dtype, k, tag == "literal"), (dtype, tag, k, v)
    assert max(hid.values()) < 2e-2


@pytest.mark.parametrize("case", ["full7b", "full7b_ref"])
@pytest.mark.parametrize("dtype", DTYPES)
def test_depth_full_7b_vs_reference_golden(dtype, case, capsys):
    """The real Qwen2-7B configuration, all 28 layers, the weights bench.py runs on; literal and fused paths.  `full7b`: short ragged rows
    + the headline rows; `full7b_ref`: reference-sized rows (256 video tokens, captions of 8-48 tokens)."""
    if not os.path.exists(os.path.join(GOLD, f"{case}.npz")):
        pytest.skip(f"tests/golden/{case}.npz not generated")
    res, hid = _depth_case(case, dtype, capsys)
    for tag, w in res.items():
        for k, v in w.items():
            assert v < score_rtol(dtype, k, tag == "literal"), (dtype, tag, k, v)
    assert max(hid.values()) < 2e-2


@pytest.mark.parametrize("case", ["heavy", "sink"])
@pytest.mark.parametrize("dtype", DTYPES + ["f8"])
def test_heavy_tailed_weights_28_layers_vs_reference_golden(dtype, case, capsys):
    """Weights reshaped towards a trained checkpoint's statistics (oracle/gen_golden_heavy.py: heavy-tailed norm weights with channels at 8 and 1/16, q / k
    biases of order one with +-6 outliers, sharper attention, two 'massive' residual channels 20 - 40 x the stream's rms from layer 2 on), 28 layers at
    H = 1024, the reference in fp32: every fixture before this one had N(0, 0.02^2) weights.  Same score bar, fused and literal paths."""
    from oracle.gen_golden_heavy import CASES as HEAVY_CASES, heavy_weights
    SPEC = HEAVY_CASES[case]      # `sink`: additionally, delimiter tokens whose embeddings carry +-250 in two channels (position-specific massive activations, attention sinks)
    g = np.load(os.path.join(GOLD, f"{case}.npz"))
    dims = synth.ModelDims(**SPEC["dims"])
    model = BlimModel(dims, max_positions=1024, dtype=dtype)
    try:
        model.engine.load_weights(heavy_weights(dims, SPEC["wseed"], sink=SPEC.get("sink", False)))
        prob = synth.make_problem(SPEC["pseed"], SPEC["n"], dims, tok_per_clip=SPEC["tok_per_clip"], text_len=SPEC["text_len"])
        model.set_tvg_prefix_length(prob.tvg_prefix_length)
        t = types.SimpleNamespace(spec=SPEC, dims=dims, model=model, prob=prob, dtype=dtype, case=case)
        plain, auto = None, None
        if case == "sink" and dtype == "f16":
            # massive activations on sink positions: plain fp16 VTG calls -- the library default and the reference's own numerics -- read 3.2e-3 here (reported below, not
            # asserted).  The driver's default is `--vtg_precise auto`: the mode is MEASURED on the loaded weights (PairScorer.calibrate_vtg), must come out as one of
            # the cheap compensated modes on this fixture, and every pass is then held to the same 1e-3 as everywhere else -- no carve-out.
            plain = {tag: _worst_rel(_six_passes(t, lit, names=("v2t_vtg", "t2v_vtg")), g) for tag, lit in (("fused", False), ("literal", True))}
            auto = _resolve_auto(t)
            t.model.vtg_precise = auto[0]
        tauto = None
        if dtype != "f8":
            # `--tvg_precise auto`: measured too; massive residual channels need the MLP branch compensated -- the passes below run in what it chose
            tauto = _resolve_tvg_auto(t)
        res = {tag: _worst_rel(_six_passes(t, literal), g) for tag, literal in ((("fused", False), ("literal", True)) if dtype != "f8" else (("fused", False),))}
    finally:
        model.engine.close()
    if tauto is not None:
        with capsys.disabled():
            print(f"\n[{case} {dtype}] tvg_precise auto (max / rms vs the fully compensated mode): " + ", ".join(f"{k} {v['max']:.1e} / {v['rms']:.1e}" for k, v in tauto[1].items())
                  + f" -> {tauto[0]} (the TVG passes below ran in it)")
        if dtype == "bf16":
            assert tauto[0] == "full", tauto                                # 8-bit mantissas: a plain MLP branch leaves 2e-3
    if auto is not None:
        with capsys.disabled():
            print(f"\n[sink f16] plain fp16 VTG calls (library default), fused: " + ", ".join(f"{k} {v:.2e}" for k, v in plain["fused"].items())
                  + "; vtg_precise auto (max / rms vs the fully compensated mode): " + ", ".join(f"{k} {v['max']:.1e} / {v['rms']:.1e}" for k, v in auto[1].items())
                  + f" -> {auto[0]} (the passes below ran in it)")
        # plain fp16 is rejected on these weights and the passes below hold 1e-3 fully compensated (round 4's intermediate modes -- qk 1.2e-3, qkx 6.6e-4, attn
        # 5.3e-4 on this fixture -- are gone: DESIGN.md section 4)
        assert auto[0] == "full", auto
    with capsys.disabled():
        for tag, w in res.items():
            print(f"\n[{case} {dtype} {tag}] worst relative score deviation vs the fp32 reference, 28 layers, residual |max| {float(g['resid_absmax_per_layer'].max()):.0f} at rms "
                  f"{float(g['resid_rms_per_layer'].max()):.1f}: " + ", ".join(f"{k} {v:.2e}" for k, v in w.items()))
    # the reference's OWN .half() run (main.py:97) against its fp32 run on these weights: the yardstick for a plain 16-bit mode (keys H16_*)
    own = {}
    for k in ("v2t_vtg", "t2v_vtg"):
        if f"H16_{k}" in g.files:
            m = (g[f"S_{k}"] != -100.0) & ~g[f"H16_{k}_nonfinite"]
            own[k] = float((np.abs(g[f"H16_{k}"][m] - g[f"S_{k}"][m]) / np.abs(g[f"S_{k}"][m])).max())
    for tag, w in res.items():
        for k, v in w.items():
            if dtype == "f8":      # reported, non-parity mode: the outliers must not make it WORSE than on N(0, 0.02^2) weights (same bounds as test_depth_fp8_mode_deltas_*)
                assert v < (0.18 if "tvg" in k else 0.08), (dtype, tag, k, v)
            else:
                assert v < score_rtol(dtype, k, tag == "literal"), (dtype, tag, k, v)


@pytest.mark.parametrize("case", ["heavy7b", "sink7b"])
@pytest.mark.parametrize("dtype", DTYPES)
def test_heavy_tailed_weights_full_7b_vs_reference_golden(dtype, case, capsys):
    """The same reshaping on the REAL Qwen2-7B configuration, all 28 layers (oracle/gen_golden_heavy.py --case heavy7b | sink7b: the reference in fp32, 30.5 GB of weights).
    The engine fills its weights on device from the seed and takes the reshaped tensors (norms, q / k biases and weights, layer 2's down_proj; sink7b: the embedding table)
    from the host.  sink7b on fp16: the plain VTG calls hold the bar at this size (7.7e-4; the H = 1024 `sink` case, 3.2e-3, is the harsher one) and the fully
    compensated mode is measured beside them."""
    from oracle.gen_golden_heavy import CASES as HEAVY_CASES, heavy_items
    SPEC7B = HEAVY_CASES[case]
    path = os.path.join(GOLD, f"{case}.npz")
    if not os.path.exists(path):
        pytest.skip(f"tests/golden/{case}.npz not generated")
    g = np.load(path)
    sink = bool(SPEC7B.get("sink", False))
    dims = synth.ModelDims(**SPEC7B["dims"])
    model = BlimModel(dims, max_positions=1024, dtype=dtype)
    extra = {}
    try:
        model.engine.init_synthetic_weights(SPEC7B["wseed"])
        for name, arr in heavy_items(dims, SPEC7B["wseed"], only_changed=True, sink=sink):
            model.engine.load_weight(name, arr)
        prob = synth.make_problem(SPEC7B["pseed"], SPEC7B["n"], dims, tok_per_clip=SPEC7B["tok_per_clip"], text_len=SPEC7B["text_len"])
        model.set_tvg_prefix_length(prob.tvg_prefix_length)
        t = types.SimpleNamespace(spec=SPEC7B, dims=dims, model=model, prob=prob, dtype=dtype, case=case)
        res = {tag: _worst_rel(_six_passes(t, literal), g) for tag, literal in (("fused", False), ("literal", True))}
        # `--tvg_precise auto` on these weights (massive residual channels at 7B width): measured, then the TVG passes re-run in the chosen mode against the golden
        tauto = _resolve_tvg_auto(t)
        if tauto[0] != "full":
            for tag, literal in (("fused", False), ("literal", True)):
                res[f"{tag}-tvg-{tauto[0]}"] = _worst_rel(_six_passes(t, literal, names=TVG_PASSES), g)
        model.tvg_precise = "full"
        if sink and dtype == "f16":
            for mode in ("full",):
                model.vtg_precise = mode
                extra[mode] = _worst_rel(_six_passes(t, False, names=("v2t_vtg", "t2v_vtg", "v2t_vtg_cpn")), g)
    finally:
        model.engine.close()
    with capsys.disabled():
        print(f"\n[{case} {dtype}] tvg_precise auto (max / rms vs the fully compensated mode): " + ", ".join(f"{k} {v['max']:.1e} / {v['rms']:.1e}" for k, v in tauto[1].items()) + f" -> {tauto[0]}")
        for tag, w in res.items():
            print(f"\n[{case} {dtype} {tag}] worst relative score deviation vs the fp32 reference, 28 layers of the 7B configuration, residual |max| "
                  f"{float(g['resid_absmax_per_layer'].max()):.0f} at rms {float(g['resid_rms_per_layer'].max()):.1f}: " + ", ".join(f"{k} {v:.2e}" for k, v in w.items()))
        for mode, w in extra.items():
            print(f"[{case} {dtype} fused, vtg_precise = {mode}] " + ", ".join(f"{k} {v:.2e}" for k, v in w.items()))
    for tag, w in res.items():
        for k, v in w.items():
            assert v < score_rtol(dtype, k, tag == "literal"), (dtype, tag, k, v)      # at this size plain fp16 holds the bar in front of the sinks too (7.7e-4)
    if extra:
        assert max(extra["full"].values()) < 2e-4, extra                          # fully compensated: a decade inside the bar in front of the sinks too


@pytest.mark.parametrize("weights", ["gaussian", "heavy7b"])
def test_auto_modes_hold_the_bar_over_a_whole_evaluation(weights, capsys):
    """`--vtg_precise auto` / `--tvg_precise auto` decide on a 256-pair sample; the bar is per entry of the WHOLE evaluation.  tools/tvg_auto_validate.py runs one six-pass
    evaluation of N = 400 reference-shaped items on the real 7B configuration twice -- every call fully compensated (<= 1e-4 from the fp32 reference on every fixture), then
    with both modes on auto -- and compares all 6 x 6,400 entries: none may differ by more than 1e-3.  (With the max + 4.5 x rms rule alone, heavy7b weights at N = 1,000
    let VTG `qkx` through with 203 of 48,000 entries above the bar, largest 4.8e-3, and TVG `attn` with 5: the deviations' tail is log-normal, not Gaussian --
    retrieval_utils.predicted_max_deviation; profiles/r04_auto_tail_validation.md.)"""
    import json
    import subprocess
    import sys
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "tvg_auto_validate.py"), "--vtg", "--n", "400", "--weights", weights], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads(r.stdout[r.stdout.index("{"):])
    mats = {k: v for k, v in d.items() if isinstance(v, dict) and "entries" in v}
    assert len(mats) == 6 and all(v["entries"] == 6400 for v in mats.values())
    with capsys.disabled():
        print(f"\n[{weights}, N = 400, 7B] auto -> vtg {d['vtg_chosen']}, tvg {d['chosen']}; whole evaluation vs fully compensated: "
              + "; ".join(f"{k} max {v['max']:.1e} rms {v['rms']:.1e} over {v['over_1e-3']}" for k, v in mats.items())
              + f"  ({d['seconds_full']} s fully compensated, {d['seconds_auto']} s auto)")
    assert all(v["over_1e-3"] == 0 for v in mats.values()), mats
    if weights == "gaussian":
        assert d["vtg_chosen"] == "none" and d["chosen"] in ("attn", "full"), d      # the cheap modes are kept where they hold
    else:
        assert d["vtg_chosen"] == "full", d                                        # massive residual channels: plain fp16 leaves 0.5 % of the entries above the bar


@pytest.mark.parametrize("literal", [True, False], ids=["literal-api", "fused-pairscorer"])
def test_masked_query_zero_option_matches_its_restatement(literal):
    """Engine option "masked_query_zero" (default OFF; PARITY-UNPINNED): masked query positions write a zero attention output -- the reference's flash-attention-2
    class (modeling_qwen2_flash.py:526-563: such positions are dropped before flash_attn_varlen_func and zero-padded back), which cannot be imported here.  Checked
    against the oracle's restatement of those lines (OracleModel.masked_query_zero), not against a recorded run: the TVG-CPN prior, whose first gathered row is a
    masked position (modeling_videochat_flash.py:414-417), changes; passes whose scored rows are never masked do not; with the option off every golden is unchanged
    (the rest of this file)."""
    t = _build("tiny", dtype="f16")
    try:
        om = O.OracleModel(O.OracleConfig(**t.d), t.w); om.set_tvg_prefix_length(t.prob.tvg_prefix_length)
        prob, spec = t.prob, t.spec
        tvg = O.padding_ids([np.asarray(r) for r in prob.tvg_ids], [np.asarray(r) for r in prob.tvg_labels], [np.asarray(r) for r in prob.tvg_masks], synth.PAD_ID)
        n = spec["n"]

        def oracle_prior(zero):
            om.masked_query_zero = zero
            S = np.full((n, n), -100.0, np.float32)
            return O.compute_t2v_scores_x(S, prob.t2v_sims[:spec.get("queries", n)], 0, tvg[0], tvg[2], tvg[1], [np.asarray(v) for v in prob.video], prob.video_vocab, prob.tvg_video_labels,
                                          om, spec["topk"], spec["bs"], t.dims.num_clips, "tvg", cpn=True)

        want_off, want_on = oracle_prior(False), oracle_prior(True)
        m = want_off != -100.0
        assert float(np.max(np.abs(want_on[m] - want_off[m]) / np.abs(want_off[m]))) > 1e-2            # the two semantics really differ on this pass
        got_off = _six_passes(t, literal, names=("t2v_tvg_cpn", "t2v_tvg"))
        t.model.masked_query_zero = True
        got_on = _six_passes(t, literal, names=("t2v_tvg_cpn", "t2v_tvg"))
        t.model.masked_query_zero = False
        rel = lambda a, b: float(np.max(np.abs(a[m].astype(np.float64) - b[m]) / np.abs(b[m])))
        assert rel(got_off["t2v_tvg_cpn"], want_off) < SCORE_RTOL and rel(got_on["t2v_tvg_cpn"], want_on) < SCORE_RTOL, (rel(got_off["t2v_tvg_cpn"], want_off), rel(got_on["t2v_tvg_cpn"], want_on))
        assert np.array_equal(got_on["t2v_tvg"], got_off["t2v_tvg"])                                    # no scored row of the likelihood pass is a masked position
        with pytest.raises(eng.BlimError, match="16-bit engine"):
            e8 = eng.Engine(synth.ModelDims(vocab_size=151700, hidden_size=256, intermediate_size=512, num_layers=1, num_heads=2, num_kv_heads=1, mm_hidden_size=64), max_positions=64, dtype="f8")
            try:
                e8.set_option("masked_query_zero", 1)
            finally:
                e8.close()
    finally:
        t.model.engine.close()


def test_benched_step_plan_meets_the_reference_golden(capsys):
    """The batch bench.py times, itself: plan 0 of rank 0 (55 video queries x top-16 texts = 880 pairs, 32,560 packed tokens, real 7B
    configuration, weight seed 0) is built by bench.build_step_plans and run once; `full7b_bench.npz` holds what the REFERENCE's own
    loops return for the first four query rows of each direction on the same problem (64 + 64 VTG entries + the TVG passes).  The 64
    v2t entries are read out of the benched step's output; the other passes go through the same scorer.  1e-3 per entry."""
    import bench
    path = os.path.join(GOLD, "full7b_bench.npz")
    if not os.path.exists(path):
        pytest.skip("tests/golden/full7b_bench.npz not generated")
    g = np.load(path)
    dims = synth.ModelDims()
    model = BlimModel(dims, max_positions=1024, dtype="f16")
    model.engine.init_synthetic_weights(0)
    try:
        ((scorer, plan, prob, pairs),) = bench.build_step_plans(model, 0, 1, 55, 16)
        assert plan.n_pairs == 880 and plan.n_tokens == 32560 and plan.n_rows == 28160       # the step BENCH_r*.json is quoted on
        model.set_tvg_prefix_length(prob.tvg_prefix_length)
        sc = scorer.run(plan).float().cpu().numpy()
        S = np.full((55, 55), -100.0, np.float32)
        for k, outs in enumerate(plan.out_index):
            S[pairs[outs, 0], pairs[outs, 1]] = sc[k]
        worst = {}
        G = g["SYN_v2t_vtg"]
        m = G != -100.0
        assert m.sum() == 64 and m[:4].sum() == 64 and np.array_equal(m[:4], (S != -100.0)[:4])    # the same top-16 entries of query rows 0-3
        worst["v2t_vtg (inside the benched step)"] = float(np.max(np.abs(S[m] - G[m]) / np.abs(G[m])))
        for name, transpose, kind, cpn in (("t2v_vtg", True, "vtg", False), ("v2t_tvg", False, "tvg", False), ("t2v_tvg", True, "tvg", False),
                                          ("t2v_tvg_cpn", True, "tvg", True)):
            G = g[f"SYN_{name}"]
            q, c = np.nonzero(G != -100.0)
            assert len(q) == 64
            pr = np.stack([c, q], axis=1) if transpose else np.stack([q, c], axis=1)         # (video, text)
            got = scorer.vtg(pr, cpn) if kind == "vtg" else scorer.tvg(pr, cpn)
            worst[name] = float(np.max(np.abs(got - G[q, c]) / np.abs(G[q, c])))
    finally:
        model.engine.close()
    with capsys.disabled():
        print("\n[full7b_bench f16] worst relative deviation vs the fp32 reference on bench.py's own step: " + ", ".join(f"{k} {v:.2e}" for k, v in worst.items()))
    for k, v in worst.items():
        assert v < SCORE_RTOL, (k, v)


def test_compensated_fp16_mode_cuts_the_hidden_state_error(capsys):
    """Engine option "precise" (hi + lo fp16 activations, GEMMs walk K twice; what the TVG calls run in): on the 28-layer H=1024
    golden batch the final hidden state is several times closer to the fp32 reference than in the plain fp16 mode."""
    t = _build("deep", device_synth=True, dtype="f16")
    g = np.load(os.path.join(GOLD, "deep.npz"))
    T = lambda a: torch.from_numpy(np.asarray(a)).cuda()
    sel = [0, 1, 2]
    try:
        r = t.model.prepare_inputs_labels_for_multimodal(T(g["pad_tvg_ids"][sel]), None, T(g["pad_tvg_masks"][sel]), None, T(g["pad_tvg_labels"][sel]),
                                                         [T(t.prob.video[i]) for i in sel], ["video"] * 3, video_feature=True, tvg=True, cpn=True)
        valid = g["prep_tvg_mask"].astype(bool)
        want = g["fwd_tvg_hidden_sub16"][valid]
        err = {}
        for precise in (False, True):
            r[4]._blim_rows = "tvg" if precise else "vtg"     # the row kind travels with the tensor (modeling.py); rows tagged VTG run plain on an fp16 engine
            out = t.model(inputs_embeds=r[4], attention_mask=r[2][0], want_logits=False)
            got = out.hidden_states.cpu().numpy()[..., ::16][valid]
            err[precise] = float(np.sqrt(np.mean((got - want) ** 2)) / np.sqrt(np.mean(want ** 2)))
    finally:
        t.model.engine.close()
    with capsys.disabled():
        print(f"\n[deep f16] final hidden state, relative rms error vs the fp32 reference: plain {err[False]:.2e}, compensated {err[True]:.2e}")
    assert err[True] < 0.5 * err[False]
    with pytest.raises(eng.BlimError, match="16-bit engine"):         # fp16 and bf16 engines have the mode (round 3), fp8 engines do not
        e = eng.Engine(synth.ModelDims(vocab_size=151700, hidden_size=256, intermediate_size=512, num_layers=1, num_heads=2, num_kv_heads=1, mm_hidden_size=64),
                       max_positions=64, dtype="f8")
        try:
            e.set_option("precise", 1)
        finally:
            e.close()


BF16_LO6_BOUND = 1.5e-4     # measured (profiles/r06b_gputest_parity_lines.txt): VTG passes 3.1 - 6.8e-5 at 7B depth; the TVG passes of a bf16 engine keep the bf16 second pass (<= 2.4e-5)


@pytest.mark.parametrize("case", ["deep", "full7b"])
@pytest.mark.parametrize("dtype", DTYPES)
def test_e2m3_second_pass_of_the_compensated_gemms(case, dtype, capsys):
    """bf16 engines (round 6): the same pass as an OPT-IN (`--second_pass e2m3`, option "precise_lo6" = 1) -- their default second pass stays a bf16
    walk over K (1 - 3e-6).  A bf16 value's lo part is 2^-9 of it and e2m3 keeps ~ 5 bits of that: hi + lo carry about what ONE fp16 rounding keeps, at 0.69x the plain
    rate instead of 0.5x; the bounds below are the measured ones with margin, well inside the 1e-3 bar on these weights.

    Engine option "precise_lo6" (default on fp16 engines; gemm.hip phase 2): in the compensated modes the product with the activations' LO parts -- 2^-11 of the
    values -- runs on the e2m3 MFMA inside the same kernel instead of a second fp16 walk over K.  (a) With the option off the fp16 K-twice kernels still meet the
    golden (the path bf16 engines keep); (b) on / off differ in bits but by far less than the bar: the e2m3 pass removes > 95 % of what the lo pass removes at
    all; (c) both meet the golden on every pass with the VTG calls fully compensated."""
    if not os.path.exists(os.path.join(GOLD, f"{case}.npz")):
        pytest.skip(f"tests/golden/{case}.npz not generated")
    t = _build(case, device_synth=True, dtype=dtype)
    g = np.load(os.path.join(GOLD, f"{case}.npz"))
    res, got = {}, {}
    try:
        t.model.vtg_precise = "full"
        for on in (1, 0):
            t.model.engine.set_option("precise_lo6", on)
            got[on] = _six_passes(t, False)
            res[on] = _worst_rel(got[on], g)
        t.model.engine.set_option("precise_lo6", 1)
        lit = _worst_rel(_six_passes(t, True), g)
    finally:
        t.model.engine.close()
    between = {}
    for k in got[1]:
        m = got[0][k] != -100.0
        between[k] = float(np.max(np.abs(got[1][k][m].astype(np.float64) - got[0][k][m]) / np.abs(got[0][k][m])))
    with capsys.disabled():
        print(f"\n[{case} {dtype}, every call fully compensated] vs the fp32 reference: e2m3 second pass " + ", ".join(f"{k} {v:.1e}" for k, v in res[1].items())
              + f"; {dtype} second pass " + ", ".join(f"{k} {v:.1e}" for k, v in res[0].items()) + "; between the two " + ", ".join(f"{k} {v:.1e}" for k, v in between.items())
              + "; literal path (e2m3) " + ", ".join(f"{k} {v:.1e}" for k, v in lit.items()))
    if dtype == "f16":
        assert max(res[0].values()) < 1e-4 and max(res[1].values()) < 2e-4 and max(lit.values()) < 2e-4, (res, lit)        # fully compensated calls sit far inside the bar either way
        assert 0.0 < max(between.values()) < 1e-4, between
    else:
        assert max(res[0].values()) < 1e-4 and max(res[1].values()) < BF16_LO6_BOUND and max(lit.values()) < BF16_LO6_BOUND, (res, lit)
        assert 0.0 < max(between.values()) < BF16_LO6_BOUND, between


def test_producers_write_the_same_e2m3_tiles_as_the_pass_over_their_rows(monkeypatch):
    """Fully compensated fp16 calls: the kernels that PRODUCE a compensated GEMM's input write the lo part of their output straight as that GEMM's e2m3 operand
    tiles -- RMSNorm (QKV and gate | up inputs), the gate | up GEMM's SwiGLU epilogue (down input) -- and the lo halves of
    those rows are neither stored nor re-read.  Same bits as the separate pass over stored lo rows (BLIM_LO6_FUSED_TILES=0): every score of the six passes, 7B
    width (I = 18,944: 148 tile columns), ragged last row tiles, fused and literal."""
    got = {}
    for fused in ("1", "0"):
        monkeypatch.setenv("BLIM_LO6_FUSED_TILES", fused)
        t = _build("wide", layers=2, device_synth=True, dtype="f16")
        try:
            t.model.vtg_precise = "full"; t.model.tvg_precise = "full"
            got[fused] = [_six_passes(t, lit) for lit in (False, True)]
        finally:
            t.model.engine.close()
    for a, b in zip(got["1"], got["0"]):
        for name in a:
            assert np.isfinite(a[name][a[name] != -100]).all()
            np.testing.assert_array_equal(a[name], b[name], err_msg=name)


def test_e2m3_weight_images_follow_the_weights():
    """The e2m3 images the compensated modes' second pass reads (option "precise_lo6") are derived state, rebuilt PER MATRIX when its source is replaced in a live engine.
    A stale image would be a SILENT error of ~2^-11 of the weight change -- so: an engine that has already run compensated calls gets another weight set loaded over the first
    (all tensors, then a single decoder matrix) and must score bit for bit like a fresh engine loaded with the same tensors."""
    spec = CASES["tiny"]
    dims = synth.ModelDims(**spec["dims"])
    prob = synth.make_problem(spec["pseed"], spec["n"], dims, tok_per_clip=spec["tok_per_clip"], text_len=spec["text_len"])
    w1, w2 = synth.synthetic_weights(dims, 11), synth.synthetic_weights(dims, 12)
    w3 = dict(w2); w3["layers.1.down_proj.w"] = w1["layers.1.down_proj.w"]

    def scores(model):
        t = types.SimpleNamespace(spec=spec, dims=dims, model=model, prob=prob, dtype="f16", case="tiny")
        model.set_tvg_prefix_length(prob.tvg_prefix_length)
        model.vtg_precise = "full"
        return _six_passes(t, False)

    live = BlimModel(dims, max_positions=1024, dtype="f16")
    try:
        assert live.engine.lo6
        live.engine.load_weights(w1)
        s1 = scores(live)                                              # builds the copies of w1
        live.engine.load_weights(w2)
        s2 = scores(live)
        live.engine.load_weight("layers.1.down_proj.w", w3["layers.1.down_proj.w"])
        s3 = scores(live)
    finally:
        live.engine.close()
    for w, got in ((w2, s2), (w3, s3)):
        fresh = BlimModel(dims, max_positions=1024, dtype="f16")
        try:
            fresh.engine.load_weights(w)
            want = scores(fresh)
        finally:
            fresh.engine.close()
        for k in want:
            assert np.array_equal(got[k], want[k]), k
    assert not np.array_equal(s1["v2t_vtg"], s2["v2t_vtg"]) and not np.array_equal(s2["v2t_vtg"], s3["v2t_vtg"])


@pytest.mark.parametrize("dtype", DTYPES)
def test_last_layer_pruning_changes_no_bit(dtype):
    """Engine option "prune_last" (default on): a call that names the rows it reads runs the last layer's o_proj / norm / MLP on those rows only.
    Rows are independent in every kernel involved, so every score equals the unpruned one bit for bit -- VTG, TVG and both priors, 7B width,
    3 layers, ragged reference-shaped rows (the shared prefix is most of the tokens: the case the pruning is for)."""
    d = dict(CASES["wide"]["dims"], num_layers=3)
    dims = synth.ModelDims(**d)
    model = BlimModel(dims, max_positions=1024, dtype=dtype)
    model.engine.init_synthetic_weights(5)
    prob = synth.make_problem(31, 6, dims, tok_per_clip=16, text_len=(4, 20))
    model.set_tvg_prefix_length(prob.tvg_prefix_length)
    tok = types.SimpleNamespace(pad_token_id=synth.PAD_ID)
    Tt = lambda rows: [torch.from_numpy(r) for r in rows]
    vtg = RU.padding_ids(Tt(prob.vtg_ids), Tt(prob.vtg_labels), Tt(prob.vtg_masks), tok)
    tvg = RU.padding_ids(Tt(prob.tvg_ids), Tt(prob.tvg_labels), Tt(prob.tvg_masks), tok)
    pairs = np.array([[j, i] for j in range(4) for i in range(6)])
    res = {}
    try:
        for on in (1, 0):
            model.engine.set_option("prune_last", on)
            sc = RU.PairScorer(DDPLike(model), vtg[0], vtg[2], vtg[1], tvg[0], tvg[2], tvg[1], [torch.from_numpy(v) for v in prob.video],
                               torch.from_numpy(prob.video_vocab), torch.from_numpy(prob.tvg_video_labels), dims.num_clips)
            (plan,) = sc.plan_vtg(pairs)
            assert plan.n_rows <= plan.n_tokens - plan.n_tokens // 16             # the pruned path is the one that runs
            res[on] = [sc.vtg(pairs), sc.vtg(pairs, cpn=True), sc.tvg(pairs), sc.tvg(pairs, cpn=True)]
    finally:
        model.engine.close()
    for a, b in zip(res[1], res[0]):
        assert np.isfinite(a).all() and np.array_equal(a.view(np.uint32), b.view(np.uint32))


def test_segmented_tvg_sequences_equal_one_sequence_per_pair():
    """The candidates of a text are packed into ONE sequence whose 3-token segments do not see each other (blim_batch.own_start): 40 candidate
    videos of one text = a 120-token sequence over four attention blocks, segments straddling tile boundaries.  Every pair's score equals
    the score of the pair planned alone (one segment: a plain causal sequence) -- likelihood and prior, 7B width (7 query heads per KV head)."""
    d = dict(CASES["wide"]["dims"], num_layers=2)
    dims = synth.ModelDims(**d)
    model = BlimModel(dims, max_positions=1024, dtype="f16")
    model.engine.init_synthetic_weights(7)
    prob = synth.make_problem(33, 40, dims, tok_per_clip=8, text_len=(4, 20))
    model.set_tvg_prefix_length(prob.tvg_prefix_length)
    tok = types.SimpleNamespace(pad_token_id=synth.PAD_ID)
    Tt = lambda rows: [torch.from_numpy(r) for r in rows]
    vtg = RU.padding_ids(Tt(prob.vtg_ids), Tt(prob.vtg_labels), Tt(prob.vtg_masks), tok)
    tvg = RU.padding_ids(Tt(prob.tvg_ids), Tt(prob.tvg_labels), Tt(prob.tvg_masks), tok)
    try:
        sc = RU.PairScorer(DDPLike(model), vtg[0], vtg[2], vtg[1], tvg[0], tvg[2], tvg[1], [torch.from_numpy(v) for v in prob.video],
                           torch.from_numpy(prob.video_vocab), torch.from_numpy(prob.tvg_video_labels), dims.num_clips)
        pairs = np.array([[j, 3] for j in range(40)] + [[j, 5] for j in range(7)])
        (plan,) = sc.plan_tvg(pairs)
        assert plan.batch.own_start is not None and plan.batch.n_seqs == 4 and int(plan.batch.seq_len.max()) == 120     # 2 prompts + 2 merged sequences
        for cpn in (False, True):
            full = sc.tvg(pairs, cpn)
            alone = np.array([sc.tvg(pairs[k:k + 1], cpn)[0] for k in range(len(pairs))])
            assert np.isfinite(full).all()
            # not bitwise: a segment that straddles a 32-key tile sums its three keys in two steps, and one flipped 16-bit rounding of the (plain, on
            # fp16 engines) SwiGLU output travels on -- measured 5e-6; with every activation compensated (option precise_act = 1) <= 2e-6
            print(f"segmented vs alone, cpn={cpn}: max rel {float(np.max(np.abs(full - alone) / np.abs(alone))):.2e}")
            np.testing.assert_allclose(full, alone, rtol=5e-5)
        # several passes planned into the same engine calls (iter_tvg_jobs: a rank's likelihood pass + its prior, the calibration sample): one call instead of
        # two, every sequence the same as in its own pass -> the same bits
        plans = list(sc.iter_tvg_jobs([(pairs, False), (pairs[:20], True)]))
        assert len(plans) == 1 and plans[0].n_pairs == len(pairs) + 20
        both = sc.score(iter(plans), len(pairs) + 20)
        np.testing.assert_array_equal(both[: len(pairs)], sc.tvg(pairs, False))
        np.testing.assert_array_equal(both[len(pairs):], sc.tvg(pairs[:20], True))
        # ... and the VTG counterpart (a rank's likelihood pass + the prior of its block of texts)
        vp = np.array([[j, 3] for j in range(5)] + [[j, 5] for j in range(3)] + [[7, 9]])
        tp = np.array([[0, 3], [0, 5], [0, 7], [0, 9]])
        plans = list(sc.iter_vtg_jobs([(vp, False), (tp, True)]))
        assert len(plans) == 1
        both = sc.score(iter(plans), len(vp) + len(tp))
        np.testing.assert_array_equal(both[: len(vp)], sc.vtg(vp, False))
        np.testing.assert_array_equal(both[len(vp):], sc.vtg(tp, True))
    finally:
        model.engine.close()


def test_fp16_stores_saturate_instead_of_overflowing():
    """fp16 outputs of the scoring kernels saturate to +-65504 (MODE.FP16_OVFL, csrc/common.hpp) instead of overflowing to inf; NaN and
    true infinities of the inputs pass through.  bf16 outputs have f32's range and are not touched."""
    M, N, K = 256, 256, 64
    a = torch.zeros((M, K), dtype=torch.float16, device="cuda"); w = torch.zeros((N, K), dtype=torch.float16, device="cuda")
    a[:, 0] = 1000.0
    w[0, 0] = 1000.0; w[1, 0] = -1000.0; w[2, 0] = 65.0; w[3, 0] = 65.6           # 1e6, -1e6, 65000 (in range: 64992 in fp16), 65625 (just beyond 65504 + half an ulp)
    a[5, 1] = float("inf"); w[4, 1] = 1.0; a[6, 1] = float("nan")
    out = eng.gemm_bf16(a, w).float().cpu().numpy()
    assert (out[:5, 0] == 65504.0).all() and (out[:5, 1] == -65504.0).all() and (out[:5, 2] == 64992.0).all() and (out[:5, 3] == 65504.0).all()
    # NaN / infinite OPERANDS still poison the result (the mode bit is set around the conversions only: while it is set the fp16 MFMA would read a
    # NaN operand as 0): +inf in, +inf out; NaN in, NaN out
    assert np.isposinf(out[5, 4]) and np.isnan(out[6, 4]) and np.isnan(out[5, 0]) and out[0, 4] == 0.0
    ob = eng.gemm_bf16(a.to(torch.bfloat16), w.to(torch.bfloat16)).float().cpu().numpy()
    assert abs(ob[0, 0] - 1.0e6) < 1.0e4 and abs(ob[0, 1] + 1.0e6) < 1.0e4


def test_activations_beyond_fp16_range(capsys):
    """`saturation.npz` (oracle/gen_golden_saturation.py): the tiny configuration with layer 0's gate / up projections scaled so that
    silu(gate) * up reaches ~7e6 (|gate|, |up| themselves stay in range) -- the shape of the massive activations real checkpoints show.
    The REFERENCE run as main.py:97 runs it (`.half()`) overflows to inf there and returns NaN for every score (recorded in the fixture); its fp32
    run is the truth.  The engine: an fp16 engine saturates the 16-bit store and stays finite (a defined value, not the truth); a bf16
    engine -- f32's range, compensated to 16 significant bits -- reproduces the fp32 reference to 1e-3."""
    from oracle.gen_golden_saturation import DIMS, N, PSEED, TEXT, TOK, TOPK, scaled_weights
    g = np.load(os.path.join(GOLD, "saturation.npz"))
    assert g["fp16_v2t_vtg_nonfinite"][g["fp32_v2t_vtg"] != -100.0].all() and g["fp16_v2t_tvg_nonfinite"][g["fp32_v2t_tvg"] != -100.0].all()    # the reference's fp16 run: all NaN
    assert not g["fp32_v2t_vtg_nonfinite"].any() and not g["fp32_v2t_tvg_nonfinite"].any()
    dims = synth.ModelDims(**DIMS)
    w = scaled_weights(dims)
    prob = synth.make_problem(PSEED, N, dims, tok_per_clip=TOK, text_len=TEXT)
    spec = dict(n=N, topk=TOPK, bs=3)
    res = {}
    for dtype in ("f16", "bf16"):
        model = BlimModel(dims, max_positions=512, dtype=dtype)
        model.engine.load_weights(w)
        t = types.SimpleNamespace(model=model, dims=dims, prob=prob, spec=spec, dtype=dtype)
        try:
            got = _six_passes(t, False, names=("v2t_vtg", "v2t_tvg"))
        finally:
            model.engine.close()
        for name, S in got.items():
            G = g[f"fp32_{name}"]
            m = G != -100.0
            assert np.array_equal(S != -100.0, m)
            assert np.isfinite(S[m]).all(), (dtype, name)                      # never NaN / inf, in either dtype
            res[(dtype, name)] = float(np.max(np.abs(S[m] - G[m]) / np.abs(G[m])))
    with capsys.disabled():
        print("\n[saturation] worst relative deviation from the fp32 reference (its fp16 run: all NaN): " + ", ".join(f"{d} {n} {v:.2e}" for (d, n), v in res.items()))
    assert res[("bf16", "v2t_vtg")] < SCORE_RTOL and res[("bf16", "v2t_tvg")] < SCORE_RTOL


@pytest.mark.parametrize("case", ["deep", "full7b"])
def test_depth_fp8_mode_deltas_vs_reference_golden(case, capsys):
    """fp8 mode against the fp32 REFERENCE at 28 layers (deltas reported -- a separate mode, never the headline, and NOT a parity mode: e4m3's 3-bit
    mantissa leaves ~5 % noise on every GEMM output whatever the scaling; measured over runs at 7B depth: VTG 1.2 - 3.8e-2, TVG 3.8e-2 - 1.2e-1, and
    2 - 4 points of R@1 on a fine-tuned weight set, profiles/r03_modes_trained_weights.md).  Bounds = 1.5 - 2x the worst measured."""
    res, _ = _depth_case(case, "f8", capsys, literal_too=False)
    for k, v in res["fused"].items():
        assert v < (0.18 if "tvg" in k else 0.08), (k, v)


class _SynthDataset:
    """Minimal stand-in for dataloader/base_dataset.py's eval side: what evaluation() reads from it."""

    def __init__(self, prob):
        self.prob = prob
        self.video_vocab = torch.from_numpy(prob.video_vocab)
        self.tvg_prefix_length = prob.tvg_prefix_length

    def __len__(self):
        return len(self.prob.video)


class _SynthLoader:
    """Yields the eval collate format of base_dataset.py:119-163 (lists of 1-D tensors per batch)."""

    def __init__(self, prob, bs):
        self.dataset = _SynthDataset(prob)
        self.prob, self.bs = prob, bs

    def __len__(self):
        return (len(self.prob.video) + self.bs - 1) // self.bs

    def __iter__(self):
        p, T = self.prob, torch.from_numpy
        for s in range(0, len(p.video), self.bs):
            e = min(len(p.video), s + self.bs)
            yield {"video": [T(v) for v in p.video[s:e]],
                   "vtg_ids": [T(x) for x in p.vtg_ids[s:e]], "vtg_labels": [T(x) for x in p.vtg_labels[s:e]], "vtg_masks": [T(x) for x in p.vtg_masks[s:e]],
                   "tvg_ids": [T(x) for x in p.tvg_ids[s:e]], "tvg_labels": [T(x) for x in p.tvg_labels[s:e]], "tvg_masks": [T(x) for x in p.tvg_masks[s:e]],
                   "tvg_video_labels": T(p.tvg_video_labels[s:e])}


@pytest.mark.parametrize("literal", [False, True], ids=["fused", "literal"])
def test_evaluation_and_val_one_epoch_end_to_end(tiny, literal):
    """evaluation() / val_one_epoch() with the reference's argument names on a loader in the reference's collate format:
    all six matrices equal the reference's golden ones, recall dicts identical (fine-tuned mode with CPN)."""
    g = np.load(os.path.join(GOLD, "tiny.npz"))
    t = tiny
    tok = types.SimpleNamespace(pad_token_id=synth.PAD_ID)
    args = types.SimpleNamespace(topk=t.spec["topk"], batch_size_eval=t.spec["bs"], num_clips=t.dims.num_clips, cpn=True, resume="ckpt", eval=True,
                                 dataset="SYNTH", alpha=[0.4, 0.8], c=[0.3, 0.6, 0.9, 0.7], literal=literal,
                                 iv2_scores={"v2t": torch.from_numpy(t.prob.v2t_sims), "t2v": torch.from_numpy(t.prob.t2v_sims)})
    loader = _SynthLoader(t.prob, bs=4)
    ddp = DDPLike(t.model)
    t2v, v2t = RU.evaluation(ddp, loader, t.model.device, tok, args)
    assert set(t2v) == {"candidate_likelihood", "query_likelihood", "internvideo2", "candidate_prior"} and set(v2t) == set(t2v)
    got = {"v2t_vtg": v2t["candidate_likelihood"], "v2t_vtg_cpn": v2t["candidate_prior"], "v2t_tvg": v2t["query_likelihood"],
           "t2v_vtg": t2v["query_likelihood"], "t2v_tvg": t2v["candidate_likelihood"], "t2v_tvg_cpn": t2v["candidate_prior"]}
    _check_passes(got, g, t, literal)
    res = TU.val_one_epoch(ddp, loader, None, t.model.device, 0, None, tokenizer=tok, args=args)
    ref_t2v = {"candidate_likelihood": g["S_t2v_tvg"], "candidate_prior": g["S_t2v_tvg_cpn"], "query_likelihood": g["S_t2v_vtg"], "internvideo2": t.prob.t2v_sims}
    ref_v2t = {"candidate_likelihood": g["S_v2t_vtg"], "candidate_prior": g["S_v2t_vtg_cpn"], "query_likelihood": g["S_v2t_tvg"], "internvideo2": t.prob.v2t_sims}
    assert res == TU.combine_and_rank(ref_t2v, ref_v2t, args, t.spec["n"])
    # zero-shot mode (no --resume): only v2t VTG, its prior and t2v VTG are computed (retrieval_utils.py:227, 242)
    args.resume = ""
    t2v0, v2t0 = RU.evaluation(ddp, loader, t.model.device, tok, args)
    assert set(t2v0) == {"query_likelihood", "internvideo2"} and set(v2t0) == {"candidate_likelihood", "candidate_prior", "internvideo2"}


def test_auto_modes_are_measured_again_when_weights_or_adapters_change():
    """`--vtg_precise / --tvg_precise auto` is a REQUEST: evaluation() measures it on the weights loaded, a second evaluation() on the same weights reuses the answer (no
    calibration calls), and any weight / adapter change -- the training loop's per-epoch validation loads new adapters every epoch, main.py:166 -- makes it unresolved so
    that the next evaluation() measures again.  (the first evaluation used to overwrite "auto" with its choice, and every later epoch's adapters were scored in
    the mode measured on the first epoch's.)"""
    t = _build("tiny", dtype="f16")
    try:
        tok = types.SimpleNamespace(pad_token_id=synth.PAD_ID)
        mk = lambda: types.SimpleNamespace(topk=t.spec["topk"], batch_size_eval=t.spec["bs"], num_clips=t.dims.num_clips, cpn=True, resume="ckpt", eval=True, dataset="SYNTH",
                                           alpha=[0.4, 0.8], c=[0.3, 0.6, 0.9, 0.7], iv2_scores={"v2t": torch.from_numpy(t.prob.v2t_sims), "t2v": torch.from_numpy(t.prob.t2v_sims)})
        loader, ddp = _SynthLoader(t.prob, bs=4), DDPLike(t.model)
        t.model.vtg_precise, t.model.tvg_precise = "auto", "auto"
        a1 = mk(); r1 = RU.evaluation(ddp, loader, t.model.device, tok, a1)
        assert "vtg_precise_table" in a1._eval_stats and "tvg_precise_table" in a1._eval_stats                       # measured
        assert t.model.vtg_precise == "auto" and t.model.tvg_precise == "auto" and t.model.vtg_mode() != "auto" and t.model.tvg_resolved()
        a2 = mk(); r2 = RU.evaluation(ddp, loader, t.model.device, tok, a2)
        assert "vtg_precise_table" not in a2._eval_stats and "tvg_precise_table" not in a2._eval_stats               # same weights: the answers stand
        assert a2._eval_stats["vtg_precise"] == a1._eval_stats["vtg_precise"] and a2._eval_stats["tvg_precise"] == a1._eval_stats["tvg_precise"]
        for d1, d2 in zip(r1, r2):
            assert all(np.array_equal(d1[k], d2[k]) for k in d1)
        H, r = t.dims.hidden_size, 8
        g_ = np.random.RandomState(0)
        t.model.engine.load_adapter("layers.0.o_proj.w", (g_.randn(r, H) * 0.02).astype(np.float32), (g_.randn(H, r) * 0.02).astype(np.float32), r, 32.0)
        assert t.model.vtg_mode() == "auto" and not t.model.tvg_resolved()                                          # new adapters: unresolved again
        a3 = mk(); RU.evaluation(ddp, loader, t.model.device, tok, a3)
        assert "vtg_precise_table" in a3._eval_stats and "tvg_precise_table" in a3._eval_stats                       # ... and measured again
        t.model.engine.load_weight("final_norm", np.ones(H, np.float32))
        assert t.model.vtg_mode() == "auto"
    finally:
        t.model.engine.close()


def test_second_pass_auto_on_a_bf16_engine_is_measured_and_follows_the_weights(capsys):
    """`--second_pass auto` (bf16 engines, round 6): evaluation() measures whether the VTG calls' second walk over K may run on the e2m3 MFMA (the calibration pairs scored with the
    bf16 second pass and with the e2m3 one), switches the engine accordingly, reuses the answer while the weights stand and measures again after a change; either way every matrix
    stays within the bar of the run with the bf16 second pass."""
    t = _build("tiny", dtype="bf16")
    try:
        tok = types.SimpleNamespace(pad_token_id=synth.PAD_ID)
        mk = lambda: types.SimpleNamespace(topk=t.spec["topk"], batch_size_eval=t.spec["bs"], num_clips=t.dims.num_clips, cpn=True, resume="ckpt", eval=True, dataset="SYNTH",
                                           alpha=[0.4, 0.8], c=[0.3, 0.6, 0.9, 0.7], iv2_scores={"v2t": torch.from_numpy(t.prob.v2t_sims), "t2v": torch.from_numpy(t.prob.t2v_sims)})
        loader, ddp = _SynthLoader(t.prob, bs=4), DDPLike(t.model)
        assert t.model.vtg_precise == "full" and t.model.second_pass == "16bit" and not t.model.engine.lo6            # a bf16 engine's defaults: the parity form
        a0 = mk(); r0 = RU.evaluation(ddp, loader, t.model.device, tok, a0)
        assert "second_pass" not in a0._eval_stats
        t.model.second_pass = "auto"
        assert not t.model.second_pass_resolved() and not t.model.engine.lo6
        a1 = mk(); r1 = RU.evaluation(ddp, loader, t.model.device, tok, a1)
        st = a1._eval_stats
        assert st["second_pass"] in ("e2m3", "16bit") and "e2m3" in st["second_pass_table"] and t.model.second_pass_resolved()
        assert t.model.engine.lo6 == (st["second_pass"] == "e2m3") and t.model.second_pass == "auto"
        worst = 0.0
        for d0, d1 in zip(r0, r1):
            for k in d0:
                m = d0[k] != -100.0
                if k != "internvideo2" and m.any():
                    worst = max(worst, float(np.max(np.abs(d1[k][m] - d0[k][m]) / np.abs(d0[k][m]))))
        with capsys.disabled():
            e = st["second_pass_table"]["e2m3"]
            print(f"\n[tiny bf16] second_pass auto: e2m3 vs bf16 second pass on the calibration pairs max {e['max']:.1e} rms {e['rms']:.1e} -> {st['second_pass']}; "
                  f"whole evaluation vs the bf16 second pass: worst {worst:.1e}")
        assert worst < 1e-3
        a2 = mk(); RU.evaluation(ddp, loader, t.model.device, tok, a2)
        assert "second_pass_table" not in a2._eval_stats and a2._eval_stats["second_pass"] == st["second_pass"]        # same weights: the answer stands
        t.model.engine.load_weight("final_norm", np.ones(t.dims.hidden_size, np.float32))
        assert not t.model.second_pass_resolved()
        a3 = mk(); RU.evaluation(ddp, loader, t.model.device, tok, a3)
        assert "second_pass_table" in a3._eval_stats                                                                   # ... and measured again
    finally:
        t.model.engine.close()


# ----------------------------------------------------------------------------- fp8 mode (BASELINE config 5)
# Building blocks are checked exactly (the quantiser against torch's own e4m3 cast, the block-scaled MFMA GEMM on integer
# data); the end-to-end scores are compared with the same fp32 golden vectors and their deviation is REPORTED and bounded
# loosely ("for fp8 report deltas") -- fp8 is a separate mode, never the headline number.
F8_SCORE_RTOL = 5e-2


@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16])
def test_fp8_row_quantiser_matches_torch_e4m3(dt):
    g = torch.Generator(device="cpu").manual_seed(3)
    x = (torch.randn((37, 3584), generator=g) * torch.logspace(-3, 2, 37).unsqueeze(1)).to(dt)
    x[5] = 0                                                     # all-zero row: scale 1, zeros out
    q, sc = eng.quant_rows(x.cuda())
    xf = x.float()
    amax = xf.abs().amax(dim=1)
    sc_ref = torch.where(amax > 0, amax / 448.0, torch.ones_like(amax))
    assert torch.equal(sc.cpu(), sc_ref)
    q_ref = (xf * (1.0 / sc_ref).unsqueeze(1)).to(torch.float8_e4m3fn).view(torch.uint8)
    assert torch.equal(q.cpu(), q_ref)


@pytest.mark.parametrize("M,N,K", [(256, 256, 128), (300, 520, 384), (1024, 768, 3584)])
def test_fp8_gemm_is_exact_on_integer_data(M, N, K):
    g = torch.Generator(device="cpu").manual_seed(M + N + K)
    a = torch.randint(-4, 5, (M, K), generator=g).float()
    w = torch.randint(-3, 4, (N, K), generator=g).float()
    sa = torch.tensor([2.0 ** int(e) for e in torch.randint(-3, 3, (M,), generator=g)])
    sw = torch.tensor([2.0 ** int(e) for e in torch.randint(-6, -2, (N,), generator=g)])
    a8 = a.to(torch.float8_e4m3fn).view(torch.uint8).cuda()
    w8 = w.to(torch.float8_e4m3fn).view(torch.uint8).cuda()
    out = eng.gemm_f8(a8, sa.cuda(), w8, sw.cuda()).float().cpu()
    ref = (a.double() @ w.double().T) * sa.double().unsqueeze(1) * sw.double().unsqueeze(0)
    assert float(ref.abs().max()) < 60000                        # inside fp16 range, integers * powers of two: exactly representable?
    assert torch.equal(out.double(), ref.to(torch.float16).double())


# ----------------------------------------------------------------------------- the compensated GEMM's e2m3 second pass (option "precise_lo6", gemm.hip phase 2)
E2M3_VALUES = np.array([m / 8.0 for m in range(8)] + [(1 + m / 8.0) * 2.0 ** (e - 1) for e in (1, 2, 3) for m in range(8)])        # magnitude of code 0..31


def _e2m3_quant(x):
    """numpy statement of kernels.hip: e2m3_block -- x [..., K] (K % 32 == 0) -> (codes uint8 [..., K] incl. the sign bit, E8M0 bytes [..., K / 32], dequantised values)."""
    x = np.asarray(x, np.float64)
    b = x.reshape(x.shape[:-1] + (-1, 32))
    amax = np.abs(b).max(axis=-1, keepdims=True)
    mant, ex = np.frexp(amax / 7.5)                                   # amax / 7.5 = mant 2^ex, mant in [0.5, 1): smallest s with 2^s >= it
    s_ = np.where(mant == 0.5, ex - 1, ex)
    s_ = np.where(amax > 0, np.clip(s_, -127, 127), -127)
    a = np.minimum(np.abs(b) * np.where(amax > 0, 2.0 ** (-s_.astype(np.float64)), 0.0), 7.5)
    bin_ = np.where(a < 2, 0, np.where(a < 4, 1, 2))
    q = np.rint(a * np.choose(bin_, [8.0, 4.0, 2.0])).astype(np.int64)
    code = np.minimum(q + 8 * bin_, 31)
    val = np.sign(b) * E2M3_VALUES[code] * 2.0 ** s_.astype(np.float64)
    return (code | np.where(b < 0, 32, 0)).astype(np.uint8).reshape(x.shape), (s_[..., 0] + 127).astype(np.uint8), val.reshape(x.shape)


def _e2m3_tiles_decode(tiles, n_rows, K, w_side):
    """tiles uint8 [row tiles, K / 128, 25600] (the operand-tile image of csrc/gemm.hpp) -> (codes [n_rows, K], E8M0 bytes [n_rows, K / 32]); rows beyond n_rows must be zero."""
    nt, nk = tiles.shape[0], K // 128
    assert tiles.shape == (nt, nk, 25600) and nt == (n_rows + 255) // 256
    data = tiles[:, :, :24576].reshape(nt, nk, 16, 1536)                                       # [tile][step][fragment group][1024 B of 16-byte parts | 512 B of 8-byte parts]
    p16 = data[..., :1024].reshape(nt, nk, 16, 4, 16, 16)                                      # [..][g][row r][16 bytes]
    p8 = data[..., 1024:].reshape(nt, nk, 16, 4, 16, 8)
    packed = np.concatenate([p16, p8], axis=-1)                                                # 24 bytes = 32 x 6 bits, little endian
    bits = np.unpackbits(packed, axis=-1, bitorder="little").reshape(nt, nk, 16, 4, 16, 32, 6)
    codes = (bits * (1 << np.arange(6))).sum(-1).astype(np.uint8)                              # [tile][step][fb][g][r][j]
    codes = codes.transpose(0, 2, 4, 1, 3, 5).reshape(nt * 256, K)                             # row = tile * 256 + fb * 16 + r; k = step * 128 + g * 32 + j
    sc = tiles[:, :, 24576:]
    rl = np.arange(256)
    e8 = np.zeros((nt, 256, nk, 4), np.uint8)
    for g in range(4):
        idx = ((rl >> 6) * 4 + g) * 64 + (rl & 15) * 4 + ((rl >> 4) & 3) if w_side else ((rl >> 7) * 4 + g) * 128 + (rl & 15) * 8 + ((rl >> 4) & 7)
        e8[:, :, :, g] = sc[:, :, idx].transpose(0, 2, 1)
    e8 = e8.reshape(nt * 256, K // 32)
    assert not codes[n_rows:].any() and not e8[n_rows:].any()                                  # the padding rows of the last tile are zeros
    return codes[:n_rows], e8[:n_rows]


@pytest.mark.parametrize("M,N,K", [(256, 256, 128), (256, 512, 256), (300, 520, 384), (520, 300, 512), (256, 256, 640), (4096, 4608, 256), (700, 512, 3584), (512, 256, 18944)])   # 1 - 5 K-steps of the second pass (its loop is unrolled by two around a three-slot ring), more tiles than CUs, the decoder's depths
def test_lo6_gemm_quantiser_and_kernel_vs_numpy(M, N, K):
    """blim_gemm_f16_lo6 = the engine's compensated GEMM: C = hi . W^T on the fp16 MFMA + e2m3(lo) . e2m3(W)^T on the block-scaled MFMA, one kernel, one set of
    accumulators.  (a) the quantiser writes exactly the operand tiles numpy's statement of the rule gives (codes, scale bytes, lane order, zero padding rows), for
    both operand sides; (b) the kernel's result equals hi . W^T + dequantised(lo) . dequantised(W)^T in float64 to f32 summation accuracy -- a wrong K assignment, scale
    block, ring slot or pairing of the second pass shows at 1e-3 of the second term; (c) the second term is really there."""
    g = np.random.RandomState(M + N + K)
    hi = (g.randn(M, K) * 0.5).astype(np.float16)
    lo = (g.randn(M, K) * 2.0 ** -11 * np.exp(g.randn(M, 1)) * (1 + 50 * (g.rand(M, K) < 0.002))).astype(np.float16)        # ragged magnitudes, a few outliers per row
    lo[:, 32:64] = 0                                                                                                       # an all-zero block
    w = (g.randn(N, K) * 0.02 * (1 + 30 * (g.rand(N, K) < 0.001))).astype(np.float16)
    out, a6, w6 = eng.gemm_f16_lo6(torch.from_numpy(np.concatenate([hi, lo], axis=1)).cuda(), torch.from_numpy(w).cuda())
    out = out.cpu().numpy().astype(np.float64)
    c_lo, e_lo, v_lo = _e2m3_quant(lo.astype(np.float64))
    c_w, e_w, v_w = _e2m3_quant(w.astype(np.float64))
    got_c, got_e = _e2m3_tiles_decode(a6.cpu().numpy(), M, K, w_side=False)
    live = np.repeat(e_lo != 0, 32, axis=1)                                        # (the sign bit of a value in an all-zero block is free)
    assert np.array_equal(got_e, e_lo) and np.array_equal(got_c[live], c_lo[live]) and not (got_c[~live] & 31).any()
    got_c, got_e = _e2m3_tiles_decode(w6.cpu().numpy(), N, K, w_side=True)
    assert np.array_equal(got_e, e_w) and np.array_equal(got_c, c_w)
    first = hi.astype(np.float64) @ w.astype(np.float64).T
    second = v_lo @ v_w.T
    scale = np.abs(hi.astype(np.float64)) @ np.abs(w.astype(np.float64)).T
    assert np.abs(out - (first + second)).max() < 2e-6 * scale.max()
    assert np.abs(second).max() > 1e-4 * np.abs(first).max() and np.abs(out - first).max() > 0.5 * np.abs(second).max()


@pytest.fixture(scope="module")
def wide_f8():
    t = _build("wide", device_synth=True, dtype="f8")
    yield t
    t.model.engine.close()


def test_fp8_mode_scores_vs_fp32_golden(wide_f8, capsys):
    """7B-width golden case in fp8 mode: deviation of every pass from the reference's fp32 scores, reported and loosely bounded."""
    t = wide_f8
    g = np.load(os.path.join(GOLD, "wide.npz"))
    got = _six_passes(t, literal=False)
    worst = {}
    for name, S in got.items():
        G = g[f"S_{name}"]
        m = G != -100.0
        assert np.array_equal(S != -100.0, m), name
        assert np.isfinite(S[m]).all(), name
        worst[name] = float((np.abs(S[m] - G[m]) / np.abs(G[m])).max())
    with capsys.disabled():
        print("\nfp8 mode, 7B width (1 layer), worst relative deviation from the fp32 reference per pass:", {k: f"{v:.2e}" for k, v in worst.items()})
    assert max(worst.values()) < F8_SCORE_RTOL


def test_fp8_mode_full_size_7b_vs_fp16(capsys):
    """All 28 layers at 7B dims: fp8-mode scores against the fp16 engine on the same pairs (deviation reported, loosely bounded),
    order invariance bitwise (per-token quantisation is row-local), candidate ranking per query compared."""
    dims = synth.ModelDims()
    prob = synth.make_problem(21, 6, dims, tok_per_clip=24, text_len=(5, 32), reference_layout=True)
    tok = types.SimpleNamespace(pad_token_id=synth.PAD_ID)
    Tt = lambda rows: [torch.from_numpy(r) for r in rows]
    vtg = RU.padding_ids(Tt(prob.vtg_ids), Tt(prob.vtg_labels), Tt(prob.vtg_masks), tok)
    tvg = RU.padding_ids(Tt(prob.tvg_ids), Tt(prob.tvg_labels), Tt(prob.tvg_masks), tok)
    pairs = np.array([[j, i] for j in range(3) for i in range(6)])
    res = {}
    for dtype in ("f16", "f8"):
        model = BlimModel(dims, max_positions=1024, dtype=dtype)
        model.engine.init_synthetic_weights(0)
        model.set_tvg_prefix_length(prob.tvg_prefix_length)
        sc = RU.PairScorer(DDPLike(model), vtg[0], vtg[2], vtg[1], tvg[0], tvg[2], tvg[1], [torch.from_numpy(v) for v in prob.video],
                           torch.from_numpy(prob.video_vocab), torch.from_numpy(prob.tvg_video_labels), dims.num_clips)
        res[dtype] = (sc.vtg(pairs), sc.tvg(pairs), sc.vtg(pairs, cpn=True))
        if dtype == "f8":
            perm = np.random.RandomState(0).permutation(len(pairs))
            assert np.array_equal(sc.vtg(pairs[perm]), res[dtype][0][perm])
        model.engine.close()
    dev = [float((np.abs(a - b) / np.abs(b)).max()) for a, b in zip(res["f8"], res["f16"])]
    with capsys.disabled():
        print(f"\nfp8 vs fp16 engine, 28 layers: worst relative deviation VTG {dev[0]:.2e}, TVG {dev[1]:.2e}, VTG prior {dev[2]:.2e}")
    # 28 random-weight layers amplify the per-GEMM 1e-2 quantisation noise (tools/f8_ablation.py: every GEMM contributes alike);
    # the bound only guards against a broken path (a wrong scale or layout gives O(1) errors)
    assert all(np.isfinite(x).all() for x in res["f8"]) and max(dev) < 0.15


def test_c_abi_error_behaviour():
    """Every entry point returns a negative code + message instead of throwing or faulting (include/blim.h conventions)."""
    lib = eng.load_library()
    import ctypes as C
    # bad configuration: head_dim != 128
    cfg = eng.Config(1000, 192, 256, 1, 2, 1, 64, 4, 128, 1, 1e-6, 1e6)
    h = C.c_void_p()
    assert lib.blim_create(C.byref(cfg), C.byref(h)) == -1 and b"head_dim" in lib.blim_last_error()
    # scoring before the weights are loaded, unknown weight names, wrong sizes
    dims = synth.ModelDims(vocab_size=151700, hidden_size=256, intermediate_size=512, num_layers=1, num_heads=2, num_kv_heads=1, mm_hidden_size=64)
    e = eng.Engine(dims, max_positions=64, dtype="f16")
    batch = eng.PackedBatch(np.arange(4, dtype=np.int32), np.ones(4, np.uint8), np.array([0], np.int32), np.array([4], np.int32))
    emb = torch.zeros((4, 256), dtype=torch.float16, device="cuda")
    with pytest.raises(eng.BlimError, match="not loaded"):
        e.decode(batch, emb)
    z = np.zeros((4, 4), np.float32)
    assert lib.blim_load_weight(e.h, b"layers.0.nonsense", z.ctypes.data, 0, 0) == -1 and b"unknown weight" in lib.blim_last_error()
    with pytest.raises(eng.BlimError, match="RoPE table"):
        e.decode(eng.PackedBatch(np.arange(70, dtype=np.int32), np.ones(70, np.uint8), np.array([0], np.int32), np.array([70], np.int32)), torch.zeros((70, 256), dtype=torch.float16, device="cuda"))
    e.close()
    # plain GEMM: K must be a whole number of 128-byte steps
    a = torch.zeros((8, 72), dtype=torch.float16, device="cuda"); w = torch.zeros((8, 72), dtype=torch.float16, device="cuda")
    with pytest.raises(eng.BlimError, match="bad argument"):
        eng.gemm_bf16(a, w)
