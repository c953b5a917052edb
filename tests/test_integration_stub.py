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
