"""TEST INFRASTRUCTURE ONLY -- generates tests/golden/lora_*.npz by running the REFERENCE's own scoring loops on a model with LoRA adapters
loaded, i.e. the flow the reference actually ships (`main.py --eval --resume <fine-tuned checkpoint>`).

Run in the build container only (needs /root/reference):  python -m oracle.gen_golden_lora [--case lora_tiny|lora_deep|lora7b|all]

main.py:96-105 wraps the projector `mlp` Linear "0" / "2" (then `tvg_mlp = deepcopy(mlp)`), every q/k/v/o_proj and lm_head in peft LoRA and
main.py:125-128 loads the fine-tuned adapters + visual_head; the adapters stay APART at evaluation time: y = W x + b + (alpha / r) B (A x).
peft is not installed in this image, so -- as oracle/gen_golden_train.py does -- the wrapped modules are replaced by a LoRA Linear written
HERE from peft's published forward (eval mode: dropout is the identity); everything else is the reference's: the model, its
prepare_inputs_labels_for_multimodal, forward, criteria and the retrieval_utils.compute_*_scores_x loops, in fp32 on CPU.

The adapters are seeded and NON-ZERO (blim_amd/lora.py:synthetic_trainable: || (alpha / r) B A || = 5e-2 || W ||, the relative size this
repo's fine-tuning runs produce); `tvg_mlp`'s base weights are copies of `mlp`'s, as a base checkpoint + resume file gives them.  The tests
rebuild weights (seed), adapters (seed) and inputs (seed) themselves, write a HF-layout base checkpoint + a peft-layout resume file and
load them through blim_amd/checkpoint.py; the fixtures hold the reference's six score matrices only (no reference source text).
"""
from __future__ import annotations

import argparse
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)

from blim_amd import lora, synth  # noqa: E402
from oracle import ref_harness  # noqa: E402
from oracle.blim_oracle import OracleConfig  # noqa: E402
from oracle.gen_golden import PASS_KINDS, LazyWeights, run_passes  # noqa: E402

R, ALPHA, REL = 8, 32.0, 5e-2          # main.py:65-66 defaults; relative size of the update

CASES = {
    "lora_tiny": dict(dims=dict(vocab_size=151700, hidden_size=256, intermediate_size=512, num_layers=2, num_heads=2, num_kv_heads=1,
                                mm_hidden_size=64), wseed=11, pseed=5, aseed=41, n=6, tok_per_clip=8, text_len=(3, 9), topk=4, bs=3),
    # 28 layers at H = 1024 (the `deep` configuration and problem)
    "lora_deep": dict(dims=dict(vocab_size=152064, hidden_size=1024, intermediate_size=2816, num_layers=28, num_heads=8, num_kv_heads=2,
                                mm_hidden_size=256), wseed=13, pseed=7, aseed=43, n=8, tok_per_clip=16, text_len=(4, 24), topk=4, bs=3, lazy=True),
    # the real Qwen2-7B configuration, weight seed 0 = bench.py's weights; the `full7b` problem: 2 query rows x top-4 (bs 3) per pass kind.
    # (VTG passes of tests/golden/full7b.npz = the same problem on the base model: how far the adapters move the scores)
    "lora7b": dict(dims=dict(vocab_size=152064, hidden_size=3584, intermediate_size=18944, num_layers=28, num_heads=28, num_kv_heads=4,
                             mm_hidden_size=1024), wseed=0, pseed=9, aseed=47, n=6, tok_per_clip=6, text_len=(4, 10), topk=4, bs=3, queries=2, lazy=True),
}


class TvgCopy:
    """View of a weight dict / LazyWeights in which tvg_mlp.* are copies of mlp.* (main.py:98: tvg_mlp = deepcopy(mlp) of the base checkpoint)."""

    def __init__(self, inner, dims, seed):
        self.inner, self.dims, self.seed = inner, dims, seed

    def items(self):
        keep = {}
        for name, arr in self.inner.items():
            if name.startswith("tvg_mlp."):
                continue
            if name.startswith("mlp."):
                keep[name] = arr
            yield name, arr
        for name, arr in keep.items():
            yield "tvg_" + name, arr


def base_weights(spec, dims):
    inner = LazyWeights(dims, spec["wseed"]) if spec.get("lazy") else synth.synthetic_weights(dims, spec["wseed"])
    return TvgCopy(inner, dims, spec["wseed"])


def attach_adapters(model, dims, trainable, r=R, alpha=ALPHA):
    """Replaces the modules main.py:96-105 hands to peft by a LoRA Linear (peft.tuners.lora.Linear.forward, one adapter, eval mode) and
    loads visual_head (main.py:104-107, 125-128)."""
    import torch
    import torch.nn.functional as F

    class LoRALinear(torch.nn.Module):
        def __init__(self, base, A, B):
            super().__init__()
            self.base_layer = base
            self.lora_A = torch.nn.Parameter(torch.from_numpy(np.ascontiguousarray(A)), requires_grad=False)
            self.lora_B = torch.nn.Parameter(torch.from_numpy(np.ascontiguousarray(B)), requires_grad=False)
            self.scaling = alpha / r

        def forward(self, x):
            return self.base_layer(x) + F.linear(F.linear(x, self.lora_A), self.lora_B) * self.scaling

        @property
        def weight(self):
            return self.base_layer.weight

        @property
        def bias(self):
            return self.base_layer.bias

    def wrap(parent, attr, wname):
        base = parent[attr] if isinstance(attr, int) else getattr(parent, attr)
        m = LoRALinear(base, trainable[wname + ":A"], trainable[wname + ":B"])
        if isinstance(attr, int):
            parent[attr] = m
        else:
            setattr(parent, attr, m)

    proj = model.model.mm_projector
    for pname in ("mlp", "tvg_mlp"):
        for i in (0, 2):
            wrap(getattr(proj, pname), i, f"{pname}.{i}.w")
    wrap(model, "lm_head", "lm_head")
    for li, layer in enumerate(model.model.layers):
        for q in ("q_proj", "k_proj", "v_proj", "o_proj"):
            wrap(layer.self_attn, q, f"layers.{li}.{q}.w")
    with torch.no_grad():
        model.visual_head.weight.copy_(torch.from_numpy(trainable["visual_head"]))
    return model


def run_case(name: str, out_dir: str) -> None:
    import torch
    torch.set_num_threads(8)
    spec = CASES[name]
    dims = synth.ModelDims(**spec["dims"])
    ocfg = OracleConfig(**spec["dims"])
    t0 = time.time()
    prob = synth.make_problem(spec["pseed"], spec["n"], dims, tok_per_clip=spec["tok_per_clip"], text_len=spec["text_len"])
    ns = ref_harness.load()
    model = ref_harness.build_model(ocfg, base_weights(spec, dims))
    trainable = lora.synthetic_trainable(dims, R, spec["aseed"], rel=REL, alpha=ALPHA)
    attach_adapters(model, dims, trainable)
    model.eval()
    print(f"[{name}] reference model + adapters built in {time.time() - t0:.1f}s", flush=True)
    model.set_tvg_prefix_length(prob.tvg_prefix_length)
    ddp = ref_harness.DDPish(model)
    out = {}
    run_passes(out, "S_", ns.RU, ddp, torch.device("cpu"), prob, spec, dims, list(PASS_KINDS), name)
    out["meta_case"] = np.array(name)
    out["meta_lora"] = np.array([R, ALPHA, REL, spec["aseed"]], dtype=np.float64)
    path = os.path.join(out_dir, f"{name}.npz")
    np.savez_compressed(path, **out)
    print(f"[{name}] wrote {path} ({os.path.getsize(path) / 1e6:.3f} MB)", flush=True)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--case", default="lora_tiny")
    ap.add_argument("--out", default=os.path.join(ROOT, "tests", "golden"))
    a = ap.parse_args()
    if not ref_harness.available():
        sys.exit("reference not present; fixtures can only be generated in the build container")
    for c in (list(CASES) if a.case == "all" else a.case.split(",")):
        run_case(c, a.out)
