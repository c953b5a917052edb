"""TEST INFRASTRUCTURE ONLY -- generates tests/golden/train_*.npz by running the REFERENCE's model code under torch autograd.

Run in the build container only (needs /root/reference):  python -m oracle.gen_golden_train [--case train_tiny|train_wide|all]

SURVEY.md section 8f-4 (training_utils.py:39-104, main.py:96-150).  The reference's training loop cannot run here as a whole (it needs
`peft`, `timm.optim`, CUDA autocast and a DataLoader), so this script
  * builds the reference's own VideoChatFlashQwenForCausalLM (oracle/ref_harness.py) with this repo's seeded weights,
  * wraps the modules main.py:96-101 hands to peft -- projector mlp/tvg_mlp Linear "0"/"2", every q/k/v/o_proj, lm_head -- in a LoRA
    Linear written HERE from peft's published forward (result = base(x) + lora_B(lora_A(dropout(x))) * alpha / r; peft itself is not
    installed), with seeded non-zero A and B so that every gradient is exercised, dropout 0, and makes visual_head trainable,
  * runs the body of training_utils.py:57-83 line by line against the reference's modules (collate-style left padding,
    prepare_inputs_labels_for_multimodal(video_feature=True, cpn=True), forward, VTGCriterion (mean), the TVG gather / forward_visual /
    bmm / cross_entropy glue), loss = vtg_loss + tvg_loss, loss.backward() in fp32,
  * steps torch.optim.AdamW(betas=(0.9, 0.95)) with the weight-decay grouping of timm's param_groups_weight_decay (no decay on 1-D
    tensors; every trainable tensor here is 2-D) twice, on two different ragged batches,
and stores the losses, the step-1 gradients and the parameters after step 2.  Large tensors are stored as strided row samples plus
their full L2 norm.  Weights and inputs are regenerated from seeds by the tests; the fixtures hold no reference source text.
"""
from __future__ import annotations

import argparse
import math
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)

from blim_amd import lora, synth  # noqa: E402
from oracle import ref_harness  # noqa: E402
from oracle.blim_oracle import OracleConfig  # noqa: E402

CASES = {
    "train_tiny": dict(dims=dict(vocab_size=151700, hidden_size=256, intermediate_size=512, num_layers=2, num_heads=2, num_kv_heads=1,
                                 mm_hidden_size=64), wseed=11, pseed=21, aseed=31, n=5, tok_per_clip=8, text_len=(3, 9),
                       batches=((0, 1, 2), (1, 2, 3, 4)), r=8, alpha=32.0, lr=1e-2, wd=0.05),
    # 7B width (GQA 28/4, I = 18944, real vocabulary), one layer: the GEMM shapes of the real model
    "train_wide": dict(dims=dict(vocab_size=152064, hidden_size=3584, intermediate_size=18944, num_layers=1, num_heads=28, num_kv_heads=4,
                                 mm_hidden_size=1024), wseed=12, pseed=22, aseed=32, n=3, tok_per_clip=6, text_len=(4, 10),
                       batches=((0, 1), (1, 2)), r=8, alpha=32.0, lr=1e-2, wd=0.05),
    # depth: all 28 layers (H = 1024, 8 q / 2 kv heads, I = 2816, real vocabulary) -- gradients that crossed 28 decoder layers; a gentler
    # learning rate so that the second step stays comparable
    "train_deep": dict(dims=dict(vocab_size=152064, hidden_size=1024, intermediate_size=2816, num_layers=28, num_heads=8, num_kv_heads=2,
                                 mm_hidden_size=256), wseed=13, pseed=23, aseed=33, n=3, tok_per_clip=16, text_len=(4, 24),
                       batches=((0, 1), (1, 2)), r=8, alpha=32.0, lr=1e-3, wd=0.05, max_store=2048),
}
MAX_STORE = 1 << 16


def adapter_values(dims, r: int, seed: int):
    """Seeded trainable tensors: A ~ 0.05 * bell, B ~ 0.02 * bell (non-zero so that dA is exercised), visual_head from the weight seed."""
    out = {}
    for n, s in lora.trainable_shapes(dims, r).items():
        if n == "visual_head":
            continue
        out[n] = synth.tensor(seed, n, s, std=0.05 if n.endswith(":A") else 0.02)
    return out


def sample_rows(a: np.ndarray, max_store: int = MAX_STORE) -> np.ndarray:
    """Strided row sample keeping at most max_store elements (the tests apply the same rule; CASES[..]['max_store'] overrides)."""
    a = np.asarray(a)
    if a.size <= max_store:
        return a
    stride = int(math.ceil(a.size / max_store))
    return a.reshape(a.shape[0], -1)[::stride] if a.shape[0] >= stride else a.reshape(-1)[::stride]


def run_case(name: str, out_dir: str) -> None:
    import torch
    import torch.nn.functional as F
    torch.set_num_threads(8)
    spec = CASES[name]
    dims = synth.ModelDims(**spec["dims"])
    ocfg = OracleConfig(**spec["dims"])
    r, alpha = spec["r"], spec["alpha"]
    weights = synth.synthetic_weights(dims, spec["wseed"])
    prob = synth.make_problem(spec["pseed"], spec["n"], dims, tok_per_clip=spec["tok_per_clip"], text_len=spec["text_len"])
    ns = ref_harness.load()
    RU, TU = ns.RU, ns.TU
    model = ref_harness.build_model(ocfg, weights)
    model.set_tvg_prefix_length(prob.tvg_prefix_length)
    vocab = torch.from_numpy(prob.video_vocab)
    model.set_video_vocab(vocab)
    for p in model.parameters():
        p.requires_grad_(False)

    class LoRALinear(torch.nn.Module):          # peft.tuners.lora.Linear.forward with one adapter, dropout p = 0
        def __init__(self, base, A, B):
            super().__init__()
            self.base_layer = base
            self.lora_A = torch.nn.Parameter(torch.from_numpy(A.copy()))
            self.lora_B = torch.nn.Parameter(torch.from_numpy(B.copy()))
            self.scaling = alpha / r

        def forward(self, x):
            return self.base_layer(x) + F.linear(F.linear(x, self.lora_A), self.lora_B) * self.scaling

        @property
        def weight(self):
            return self.base_layer.weight

        @property
        def bias(self):
            return self.base_layer.bias

    ad = adapter_values(dims, r, spec["aseed"])
    params = {}

    def wrap(parent, attr, wname):
        base = parent[attr] if isinstance(attr, int) else getattr(parent, attr)
        m = LoRALinear(base, ad[wname + ":A"], ad[wname + ":B"])
        if isinstance(attr, int):
            parent[attr] = m
        else:
            setattr(parent, attr, m)
        params[wname + ":A"], params[wname + ":B"] = m.lora_A, m.lora_B

    proj = model.model.mm_projector
    for pname in ("mlp", "tvg_mlp"):
        for i in (0, 2):
            wrap(getattr(proj, pname), i, f"{pname}.{i}.w")
    wrap(model, "lm_head", "lm_head")
    for li, layer in enumerate(model.model.layers):
        for q in ("q_proj", "k_proj", "v_proj", "o_proj"):
            wrap(layer.self_attn, q, f"layers.{li}.{q}.w")
    model.visual_head.weight.requires_grad_(True)                       # main.py:104-107
    params["visual_head"] = model.visual_head.weight
    names = lora.trainable_names(dims)
    assert set(names) == set(params), set(names) ^ set(params)
    plist = [params[n] for n in names]
    opt = torch.optim.AdamW([{"params": plist, "weight_decay": spec["wd"]}], lr=spec["lr"], betas=(0.9, 0.95))   # main.py:146-147
    model.train(True)
    ddp = ref_harness.DDPish(model)
    crit = TU.VTGCriterion()
    T = lambda a: torch.from_numpy(np.asarray(a))
    tok = types.SimpleNamespace(pad_token_id=synth.PAD_ID)
    C = dims.num_clips
    IMAGE_TOKEN_ID = 151645
    out = {}
    for step, sel in enumerate(spec["batches"]):
        sel = list(sel)
        # dataloader/base_dataset.py:132-153: left padding to the batch maximum (same rule as retrieval_utils.padding_ids)
        vtg_in, vtg_lab, vtg_m = RU.padding_ids([T(prob.vtg_ids[i]) for i in sel], [T(prob.vtg_labels[i]) for i in sel], [T(prob.vtg_masks[i]) for i in sel], tok)
        tvg_in, tvg_lab, tvg_m = RU.padding_ids([T(prob.tvg_ids[i]) for i in sel], [T(prob.tvg_labels[i]) for i in sel], [T(prob.tvg_masks[i]) for i in sel], tok)
        video = [T(prob.video[i]) for i in sel]
        bs = len(sel)
        mod = ["video"] * bs
        sizes = [(448, 448)] * bs
        # training_utils.py:65-66
        (_, _, (vtg_masks, _), _, vtg_embeds, vtg_labels) = ddp.module.prepare_inputs_labels_for_multimodal(vtg_in, None, vtg_m, None, vtg_lab, video, mod, image_sizes=sizes, video_feature=True, cpn=True)
        vtg_out = ddp(inputs_embeds=vtg_embeds, attention_mask=vtg_masks)
        vtg_loss = crit(vtg_out.logits, vtg_labels)
        # training_utils.py:70-79
        (_, _, (tvg_masks, _), _, tvg_embeds, tvg_labels) = ddp.module.prepare_inputs_labels_for_multimodal(tvg_in, None, tvg_m, None, tvg_lab, video, mod, image_sizes=sizes, video_feature=True, tvg=True, cpn=True)
        idx = (tvg_labels == IMAGE_TOKEN_ID).nonzero()[:, 1][:, None].repeat(1, C) + (torch.arange(C) - (C + 1))
        tvl = T(prob.tvg_video_labels[sel])[:, None].repeat(1, C)
        tvg_out = ddp(inputs_embeds=tvg_embeds, attention_mask=tvg_masks)
        vte = torch.gather(tvg_out.hidden_states, 1, idx[..., None].repeat(1, 1, tvg_out.hidden_states.shape[-1]))
        vte = ddp.module.forward_visual(vte)
        tvg_logits = torch.bmm(vte.permute(1, 0, 2), vocab.permute(1, 2, 0)).transpose(0, 1) / math.sqrt(vocab.shape[-1])
        tvg_loss = F.cross_entropy(tvg_logits.reshape(-1, tvg_logits.shape[-1]), tvl.reshape(-1))
        loss = vtg_loss + tvg_loss
        opt.zero_grad()
        loss.backward()
        out[f"loss_vtg_{step}"] = np.float32(vtg_loss.item()); out[f"loss_tvg_{step}"] = np.float32(tvg_loss.item())
        print(f"[{name}] step {step}: vtg {vtg_loss.item():.6f} tvg {tvg_loss.item():.6f}", flush=True)
        for n in names:
            g = params[n].grad.detach().numpy()
            out[f"gnorm_{step}/{n}"] = np.float32(np.linalg.norm(g.astype(np.float64)))
            if step == 0:
                out[f"grad/{n}"] = sample_rows(g, spec.get("max_store", MAX_STORE)).copy()
        opt.step()
    for n in names:
        p = params[n].detach().numpy()
        out[f"param/{n}"] = sample_rows(p, spec.get("max_store", MAX_STORE)).copy()
        out[f"pnorm/{n}"] = np.float32(np.linalg.norm(p.astype(np.float64)))
    out["meta_case"] = np.array(name)
    path = os.path.join(out_dir, f"{name}.npz")
    np.savez_compressed(path, **out)
    print(f"[{name}] wrote {path} ({os.path.getsize(path) / 1e6:.2f} MB)", flush=True)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--case", default="all")
    ap.add_argument("--out", default=os.path.join(ROOT, "tests", "golden"))
    a = ap.parse_args()
    if not ref_harness.available():
        sys.exit("reference not present; fixtures can only be generated in the build container")
    for c in (list(CASES) if a.case == "all" else a.case.split(",")):
        run_case(c, a.out)
