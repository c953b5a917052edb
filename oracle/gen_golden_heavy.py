"""TEST INFRASTRUCTURE ONLY -- generates tests/golden/heavy.npz by running the REFERENCE's own code.

Run in the build container only (needs /root/reference):  python -m oracle.gen_golden_heavy

Every other fixture uses N(0, 0.02^2) weights (initializer_range).  Trained decoders do not look like that: norm weights are
heavy-tailed, Qwen2's q / k biases are of order one, and a few residual channels carry 'massive' activations from an early layer
on.  This case keeps the `deep` architecture and problem (28 layers, H = 1024, 8 / 2 heads, I = 2816, real vocabulary; six pass
kinds through the reference's scoring loops in fp32) and reshapes the seeded weights by `heavy_weights` below -- a pure numpy rule
the tests re-apply to the same seed, so the fixture holds outputs only.  It measures whether the engine's 16-bit operand formats
hold the score bar under those dynamic ranges, where the reference is fp32 here and fp16-autocast in production (main.py:97).
The fixture is data: no reference source text is stored."""
from __future__ import annotations

import argparse
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)

from blim_amd import synth  # noqa: E402

SPEC = dict(dims=dict(vocab_size=152064, hidden_size=1024, intermediate_size=2816, num_layers=28, num_heads=8, num_kv_heads=2, mm_hidden_size=256),
            wseed=14, pseed=15, n=8, tok_per_clip=16, text_len=(4, 24), topk=4, bs=3)
MASSIVE_LAYER, MASSIVE_CHANNELS, MASSIVE_GAIN = 2, (37, 611), 1000.0


def heavy_weights(dims, seed: int):
    """synth.synthetic_weights(dims, seed) reshaped towards a trained checkpoint's statistics (all values stay bf16-representable):
    * every RMSNorm weight: exp(N(0, 0.4^2)) per channel, three channels at 8 and three at 1/16;
    * q / k biases ~ N(0, 0.5^2) with four entries at +-6 per layer, q / k weights x 1.5 (sharper attention);
    * layer MASSIVE_LAYER's down_proj rows MASSIVE_CHANNELS x MASSIVE_GAIN: two residual channels carry activations one to two orders
      above the rest from that layer on."""
    w = synth.synthetic_weights(dims, seed)
    rs = np.random.RandomState(seed)
    H = dims.hidden_size
    for name in list(w):
        a = w[name]
        if name.endswith("_norm"):
            g = np.exp(rs.randn(H).astype(np.float32) * 0.4)
            idx = rs.choice(H, 6, replace=False)
            g[idx[:3]] = 8.0; g[idx[3:]] = 0.0625
            w[name] = synth.bf16_round(g.astype(np.float32))
        elif name.endswith("q_proj.b") or name.endswith("k_proj.b"):
            b = rs.randn(a.shape[0]).astype(np.float32) * 0.5
            idx = rs.choice(a.shape[0], 4, replace=False)
            b[idx] = np.array([6.0, -6.0, 6.0, -6.0], np.float32)
            w[name] = synth.bf16_round(b)
        elif name.endswith("q_proj.w") or name.endswith("k_proj.w"):
            w[name] = synth.bf16_round(a * np.float32(1.5))
        elif name == f"layers.{MASSIVE_LAYER}.down_proj.w":
            a = a.copy()
            a[list(MASSIVE_CHANNELS)] *= np.float32(MASSIVE_GAIN)
            w[name] = synth.bf16_round(a)
    return w


def main(out_dir: str) -> None:
    import torch
    from oracle import gen_golden as G
    from oracle import ref_harness
    from oracle.blim_oracle import OracleConfig
    torch.set_num_threads(8)
    dims = synth.ModelDims(**SPEC["dims"])
    t0 = time.time()
    weights = heavy_weights(dims, SPEC["wseed"])
    prob = synth.make_problem(SPEC["pseed"], SPEC["n"], dims, tok_per_clip=SPEC["tok_per_clip"], text_len=SPEC["text_len"])
    ns = ref_harness.load()
    model = ref_harness.build_model(OracleConfig(**SPEC["dims"]), weights)
    print(f"[heavy] weights + reference model built in {time.time() - t0:.1f}s", flush=True)
    model.set_tvg_prefix_length(prob.tvg_prefix_length)
    out = {}
    # residual-stream statistics of one VTG row, for the record (how 'massive' the massive channels are)
    T = lambda a: torch.from_numpy(np.asarray(a))
    import types
    tok = types.SimpleNamespace(pad_token_id=synth.PAD_ID)
    with torch.no_grad():
        ids, lab, msk = ns.RU.padding_ids([T(x) for x in prob.vtg_ids[:1]], [T(x) for x in prob.vtg_labels[:1]], [T(x) for x in prob.vtg_masks[:1]], tok)
        r = model.prepare_inputs_labels_for_multimodal(ids, None, msk, None, lab, [T(prob.video[0])], ["video"], image_sizes=None, video_feature=True, tvg=False, cpn=False)
        hs = []
        hooks = [l.register_forward_hook(lambda m, i, o: hs.append(o[0].detach())) for l in model.model.layers]
        model(inputs_embeds=r[4], attention_mask=r[2])
        for h in hooks:
            h.remove()
        out["resid_absmax_per_layer"] = np.array([float(h.abs().max()) for h in hs], np.float32)
        out["resid_rms_per_layer"] = np.array([float(h.pow(2).mean().sqrt()) for h in hs], np.float32)
    print("[heavy] residual |max| per layer:", np.round(out["resid_absmax_per_layer"], 1).tolist(), flush=True)
    print("[heavy] residual rms per layer:", np.round(out["resid_rms_per_layer"], 2).tolist(), flush=True)
    G.run_passes(out, "S_", ns.RU, ref_harness.DDPish(model), torch.device("cpu"), prob, SPEC, dims, list(G.PASS_KINDS), "heavy")
    out["meta_case"] = np.array("heavy")
    path = os.path.join(out_dir, "heavy.npz")
    np.savez_compressed(path, **out)
    print(f"[heavy] wrote {path} ({os.path.getsize(path) / 1e6:.2f} MB)", flush=True)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "tests", "golden"))
    a = ap.parse_args()
    from oracle import ref_harness
    if not ref_harness.available():
        sys.exit("reference not present; fixtures can only be generated in the build container")
    main(a.out)
