"""The reference-side ctypes binding PRINTED in INTEGRATION.md section 2, extracted from the document and executed verbatim.

CPU part: the struct it declares has the layout of blim_amd/engine.py's (and so of include/blim.h's blim_config).
GPU part: create -> load the tiny synthetic weights -> blim_forward on the golden ragged batch -> compare with the
hidden states / scores the reference itself produced (tests/golden/tiny.npz)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from blim_amd import engine as eng
from blim_amd import synth
from oracle.gen_golden import CASES

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


def _stub_namespace():
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    sec = text[text.index("## 2. Binding the C ABI directly"):]
    block = re.search(r"```python\n(.*?)```", sec, flags=re.S).group(1)
    ns = {}
    exec(compile(block, "INTEGRATION.md#2", "exec"), ns)
    return ns


def test_documented_config_struct_matches_the_binding_and_the_header():
    ns = _stub_namespace()
    doc, ours = ns["Config"], eng.Config
    assert [(n, t) for n, t in doc._fields_] == [(n, t) for n, t in ours._fields_]
    assert C.sizeof(doc) == C.sizeof(ours) == 48
    hdr = re.sub(r"/\*.*?\*/", "", open(eng.HEADER_PATH).read(), flags=re.S)
    body = re.search(r"typedef struct blim_config \{(.*?)\} blim_config;", hdr, flags=re.S).group(1)
    names = [n.strip() for decl in body.split(";") if decl.strip() for n in decl.strip().split(None, 1)[1].split(",")]
    assert names == [n for n, _ in doc._fields_]


@pytest.mark.gpu
@pytest.mark.parametrize("compute_dtype", [1, 0], ids=["f16", "bf16"])
def test_documented_stub_runs_against_the_golden_vectors(compute_dtype):
    import torch
    from blim_amd import retrieval_utils as RU
    ns = _stub_namespace()
    spec = CASES["tiny"]
    d = spec["dims"]
    dims = synth.ModelDims(**d)
    state = synth.synthetic_weights(dims, spec["wseed"])
    cfg = dict(vocab_size=d["vocab_size"], hidden_size=d["hidden_size"], intermediate_size=d["intermediate_size"], num_hidden_layers=d["num_layers"],
               num_attention_heads=d["num_heads"], num_key_value_heads=d["num_kv_heads"], mm_hidden_size=d["mm_hidden_size"], rms_norm_eps=1e-6, rope_theta=1e6)
    fwd = ns["make_engine_forward"](eng.LIB_PATH, cfg, state, compute_dtype=compute_dtype, max_positions=512)
    g = np.load(os.path.join(GOLD, "tiny.npz"))
    try:
        for kind in ("vtg", "tvg"):
            emb = torch.from_numpy(g[f"prep_{kind}_embeds"]).cuda()
            valid = g[f"prep_{kind}_mask"].astype(bool)
            for tag, mk in (("", f"prep_{kind}_mask"), ("_cpn", f"prep_{kind}_cpn_mask")):
                out = fwd(inputs_embeds=emb, attention_mask=torch.from_numpy(g[mk]).cuda())
                got, want = out.hidden_states.cpu().numpy()[valid], g[f"fwd_{kind}{tag}_hidden"][valid]
                assert np.abs(got - want).max() / np.abs(want).max() < 1e-2
                if kind == "vtg":
                    sc = RU.vtg_criterion(out.logits, torch.from_numpy(g["prep_vtg_labels"]).cuda()).cpu().numpy()
                    np.testing.assert_allclose(sc, g[f"fwd_vtg{tag}_score"], rtol=1e-3)
    finally:
        fwd.close()


@pytest.mark.gpu
@pytest.mark.parametrize("compute_dtype", [1, 0], ids=["f16", "bf16"])
def test_documented_stub_with_lora_adapters_kept_apart(compute_dtype):
    """The same stub with its `adapters=` argument (blim_load_adapter): forward() of an engine holding the lora_tiny case's base weights + adapters apart against the
    numpy oracle on W + (alpha / r) B A merged in fp32 (which tests/golden/lora_tiny.npz pins to the reference's adapters-apart run)."""
    import torch
    import lora_fixture as LF
    from blim_amd import checkpoint as CK
    from oracle import blim_oracle as O
    ns = _stub_namespace()
    spec, g, dims, prob = LF.load_case("lora_tiny")
    d = spec["dims"]
    w = LF.base_weights_host(spec, dims)
    tr = LF.trainable_of(spec, dims)
    state = dict(w); state["visual_head"] = tr["visual_head"]
    adapters = {n: (tr[n + ":A"], tr[n + ":B"]) for n in CK.expected_adapters(dims)}
    cfg = dict(vocab_size=d["vocab_size"], hidden_size=d["hidden_size"], intermediate_size=d["intermediate_size"], num_hidden_layers=d["num_layers"],
               num_attention_heads=d["num_heads"], num_key_value_heads=d["num_kv_heads"], mm_hidden_size=d["mm_hidden_size"], rms_norm_eps=1e-6, rope_theta=1e6)
    fwd = ns["make_engine_forward"](eng.LIB_PATH, cfg, state, compute_dtype=compute_dtype, max_positions=512, adapters=adapters, lora_r=LF.R, lora_alpha=LF.ALPHA)
    om = O.OracleModel(O.OracleConfig(**d), LF.merged_fp32(w, tr)); om.set_tvg_prefix_length(prob.tvg_prefix_length)
    ov = O.padding_ids(prob.vtg_ids, prob.vtg_labels, prob.vtg_masks, synth.PAD_ID)
    sel = [0, 1, 2]
    mask, _, emb, lab = om.prepare_inputs_labels_for_multimodal(ov[0][sel], ov[2][sel], ov[1][sel], [prob.video[i] for i in sel])
    want_h = om.forward_hidden(emb, mask)
    want_s = om.label_logprobs(want_h, lab)
    base_s = O.OracleModel(O.OracleConfig(**d), w).label_logprobs(O.OracleModel(O.OracleConfig(**d), w).forward_hidden(emb, mask), lab)
    try:
        from blim_amd import retrieval_utils as RU
        out = fwd(inputs_embeds=torch.from_numpy(emb).cuda(), attention_mask=torch.from_numpy(mask).cuda())
        valid = mask.astype(bool)
        got = out.hidden_states.cpu().numpy()[valid]
        assert np.abs(got - want_h[valid]).max() / np.abs(want_h[valid]).max() < (1e-2 if compute_dtype == 1 else 3e-2)
        sc = RU.vtg_criterion(out.logits, torch.from_numpy(lab).cuda()).cpu().numpy()
        np.testing.assert_allclose(sc, want_s, rtol=1e-3 if compute_dtype == 1 else 3e-3)        # (the stub's plain bf16 forward is the non-parity bf16 mode)
        assert np.max(np.abs(want_s - base_s) / np.abs(base_s)) > 1e-4                            # the adapters are not a no-op
    finally:
        fwd.close()
