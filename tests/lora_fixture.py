"""Shared by the LoRA-adapted scoring tests (tests/golden/lora_*.npz, oracle/gen_golden_lora.py): rebuilds a case's weights, adapters and
inputs from their seeds, and writes the two files the reference's `--eval --resume` flow reads -- a HF-layout base checkpoint (no tvg_mlp:
main.py:98 copies it from mlp) and a peft-layout resume file (util/misc.py:276-297)."""
import os

import numpy as np

from blim_amd import checkpoint as CK
from blim_amd import lora, synth

GOLD = os.path.join(os.path.dirname(__file__), "golden")
R, ALPHA, REL = 8, 32.0, 5e-2
# mirrors oracle/gen_golden_lora.py:CASES (the generator needs /root/reference and does not travel to the GPU box; the fixtures' meta_lora
# entry is checked against R / ALPHA / REL / aseed below)
CASES = {
    "lora_tiny": dict(dims=dict(vocab_size=151700, hidden_size=256, intermediate_size=512, num_layers=2, num_heads=2, num_kv_heads=1,
                                mm_hidden_size=64), wseed=11, pseed=5, aseed=41, n=6, tok_per_clip=8, text_len=(3, 9), topk=4, bs=3),
    "lora_deep": dict(dims=dict(vocab_size=152064, hidden_size=1024, intermediate_size=2816, num_layers=28, num_heads=8, num_kv_heads=2,
                                mm_hidden_size=256), wseed=13, pseed=7, aseed=43, n=8, tok_per_clip=16, text_len=(4, 24), topk=4, bs=3),
    "lora7b": dict(dims=dict(vocab_size=152064, hidden_size=3584, intermediate_size=18944, num_layers=28, num_heads=28, num_kv_heads=4,
                             mm_hidden_size=1024), wseed=0, pseed=9, aseed=47, n=6, tok_per_clip=6, text_len=(4, 10), topk=4, bs=3, queries=2),
}
PASSES = [("v2t_vtg", "v2t", "vtg", False), ("v2t_vtg_cpn", "v2t", "vtg", True), ("v2t_tvg", "v2t", "tvg", False),
          ("t2v_vtg", "t2v", "vtg", False), ("t2v_tvg", "t2v", "tvg", False), ("t2v_tvg_cpn", "t2v", "tvg", True)]


def load_case(name):
    spec = CASES[name]
    g = np.load(os.path.join(GOLD, f"{name}.npz"))
    assert list(g["meta_lora"]) == [R, ALPHA, REL, spec["aseed"]], "fixture was generated with other adapter settings"
    dims = synth.ModelDims(**spec["dims"])
    prob = synth.make_problem(spec["pseed"], spec["n"], dims, tok_per_clip=spec["tok_per_clip"], text_len=spec["text_len"])
    return spec, g, dims, prob


def trainable_of(spec, dims):
    return lora.synthetic_trainable(dims, R, spec["aseed"], rel=REL, alpha=ALPHA)


def base_weights_host(spec, dims):
    """Base weights as a base checkpoint + main.py:98 give them: tvg_mlp.* = copies of mlp.*."""
    w = synth.synthetic_weights(dims, spec["wseed"])
    for k in ("0.w", "0.b", "2.w", "2.b"):
        w["tvg_mlp." + k] = w["mlp." + k].copy()
    return w


def merged_fp32(w, trainable):
    """W + (alpha / r) B A in float32, visual_head replaced: what the reference's adapters-apart forward equals in exact arithmetic."""
    out = dict(w)
    for n in trainable:
        if n.endswith(":A"):
            base = n[:-2]
            out[base] = (w[base] + np.float32(ALPHA / R) * (trainable[base + ":B"] @ trainable[n])).astype(np.float32)
    out["visual_head"] = trainable["visual_head"].astype(np.float32)
    return out


def write_files(tmp, w, trainable, shards=2):
    """-> (base checkpoint dir, resume file)."""
    import torch
    base = os.path.join(str(tmp), "base")
    CK.save_hf_checkpoint({k: v for k, v in w.items() if k != "visual_head"}, base, shards=shards)      # the public base checkpoint predates visual_head
    resume = os.path.join(str(tmp), "resume.pth")
    st = lora.resume_state(trainable)
    st["epoch"] = 4
    torch.save(st, resume)
    return base, resume
