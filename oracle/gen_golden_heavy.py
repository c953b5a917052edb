"""TEST INFRASTRUCTURE ONLY -- generates tests/golden/heavy.npz by running the REFERENCE's own code.

Run in the build container only (needs /root/reference):  python -m oracle.gen_golden_heavy [--case heavy|heavy7b]

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
# the same reshaping on the REAL 7B configuration (the `full7b` problem: 2 query rows x top-4 per pass kind; fp32 reference = 30.5 GB of weights, streamed)
SPEC7B = dict(dims=dict(vocab_size=152064, hidden_size=3584, intermediate_size=18944, num_layers=28, num_heads=28, num_kv_heads=4, mm_hidden_size=1024),
              wseed=0, pseed=9, n=6, tok_per_clip=6, text_len=(4, 10), topk=4, bs=3, queries=2)
# `sink`: the heavy rule plus delimiter tokens whose EMBEDDINGS carry two channels at +-250 (<|im_start|>, which opens every row and every turn) / 120 ("\n"):
# massive activations on specific positions only, which the sharpened attention then uses as sinks -- the pattern trained decoders show on their first token
SPEC_SINK = dict(SPEC, sink=True)
CASES = {"heavy": SPEC, "heavy7b": SPEC7B, "sink": SPEC_SINK, "sink7b": dict(SPEC7B, sink=True)}
SINK_ROWS = ((151644, (5, 900), (250.0, -250.0)), (198, (5,), (120.0,)))
MASSIVE_LAYER, MASSIVE_CHANNELS, MASSIVE_GAIN = 2, (37, 611), 1000.0


def _changed(name: str, sink: bool = False) -> bool:
    return ((sink and name == "embed_tokens") or name.endswith("_norm") or name.endswith("q_proj.b") or name.endswith("k_proj.b") or name.endswith("q_proj.w") or name.endswith("k_proj.w")
            or name == f"layers.{MASSIVE_LAYER}.down_proj.w")


def _reshape(name: str, base, shape, rs, H: int, sink: bool = False):
    """One tensor of the rule below; `base` = the seeded tensor (a callable, evaluated only where the rule needs it).  Draws from `rs` in tensor order."""
    if sink and name == "embed_tokens":
        a = base().copy()
        for tok, chans, vals in SINK_ROWS:
            a[tok, [c % H for c in chans]] = np.array(vals, np.float32)
        return a
    if name.endswith("_norm"):
        g = np.exp(rs.randn(H).astype(np.float32) * 0.4)
        idx = rs.choice(H, 6, replace=False)
        g[idx[:3]] = 8.0; g[idx[3:]] = 0.0625
        return synth.bf16_round(g.astype(np.float32))
    if name.endswith("q_proj.b") or name.endswith("k_proj.b"):
        b = rs.randn(shape[0]).astype(np.float32) * 0.5
        idx = rs.choice(shape[0], 4, replace=False)
        b[idx] = np.array([6.0, -6.0, 6.0, -6.0], np.float32)
        return synth.bf16_round(b)
    if name.endswith("q_proj.w") or name.endswith("k_proj.w"):
        return synth.bf16_round(base() * np.float32(1.5))
    if name == f"layers.{MASSIVE_LAYER}.down_proj.w":
        a = base().copy()
        a[list(MASSIVE_CHANNELS)] *= np.float32(MASSIVE_GAIN)
        return synth.bf16_round(a)
    return base()


def heavy_items(dims, seed: int, only_changed: bool = False, sink: bool = False):
    """(name, tensor) of synth.synthetic_weights(dims, seed) reshaped towards a trained checkpoint's statistics, one tensor at a time (all values stay
    bf16-representable):
    * every RMSNorm weight: exp(N(0, 0.4^2)) per channel, three channels at 8 and three at 1/16;
    * q / k biases ~ N(0, 0.5^2) with four entries at +-6 per layer, q / k weights x 1.5 (sharper attention);
    * layer MASSIVE_LAYER's down_proj rows MASSIVE_CHANNELS x MASSIVE_GAIN: two residual channels carry activations far above the rest from that layer on.
    only_changed = True yields just the tensors the rule touches (a GPU test fills the rest on device from the same seed)."""
    from oracle.gen_golden import fast_tensor
    rs = np.random.RandomState(seed)
    for name, shape in synth.weight_shapes(dims).items():
        if only_changed and not _changed(name, sink):
            continue
        yield name, _reshape(name, lambda: fast_tensor(seed, name, shape, *synth.weight_dist(name)), shape, rs, dims.hidden_size, sink)


def heavy_weights(dims, seed: int, sink: bool = False):
    return dict(heavy_items(dims, seed, sink=sink))


class _Lazy:
    def __init__(self, dims, seed, sink=False):
        self.dims, self.seed, self.sink = dims, seed, sink

    def items(self):
        return heavy_items(self.dims, self.seed, sink=self.sink)


def main(out_dir: str, case: str = "heavy") -> None:
    import torch
    from oracle import gen_golden as G
    from oracle import ref_harness
    from oracle.blim_oracle import OracleConfig
    torch.set_num_threads(8)
    SPEC = CASES[case]
    dims = synth.ModelDims(**SPEC["dims"])
    t0 = time.time()
    weights = _Lazy(dims, SPEC["wseed"], SPEC.get("sink", False))
    prob = synth.make_problem(SPEC["pseed"], SPEC["n"], dims, tok_per_clip=SPEC["tok_per_clip"], text_len=SPEC["text_len"])
    ns = ref_harness.load()
    model = ref_harness.build_model(OracleConfig(**SPEC["dims"]), weights)
    del weights
    print(f"[{case}] weights + reference model built in {time.time() - t0:.1f}s", flush=True)
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
    print(f"[{case}] residual |max| per layer:", np.round(out["resid_absmax_per_layer"], 1).tolist(), flush=True)
    print(f"[{case}] residual rms per layer:", np.round(out["resid_rms_per_layer"], 2).tolist(), flush=True)
    G.run_passes(out, "S_", ns.RU, ref_harness.DDPish(model), torch.device("cpu"), prob, SPEC, dims, list(G.PASS_KINDS), case)
    if SPEC["dims"]["hidden_size"] <= 1024:
        # the reference's PRODUCTION numerics on the same weights: `.half()` (main.py:97), here literally on CPU -- how far its own fp16 run is from its fp32
        # run is the yardstick for the engine's plain fp16 mode on these statistics (keys H16_*; non-finite entries stored as 0 with a mask)
        model = model.half()
        ddp = ref_harness.DDPish(model)
        vtg = ns.RU.padding_ids([T(x) for x in prob.vtg_ids], [T(x) for x in prob.vtg_labels], [T(x) for x in prob.vtg_masks], tok)
        tvg = ns.RU.padding_ids([T(x) for x in prob.tvg_ids], [T(x) for x in prob.tvg_labels], [T(x) for x in prob.tvg_masks], tok)
        video = [T(v).half() for v in prob.video]
        vocab, vlab = T(prob.video_vocab).half(), T(prob.tvg_video_labels)
        args = types.SimpleNamespace(topk=SPEC["topk"], batch_size_eval=SPEC["bs"], num_clips=dims.num_clips)
        n = SPEC["n"]
        with torch.no_grad():
            for pname in G.PASS_KINDS:
                qv, ftype, cpn = G.PASS_KINDS[pname]
                fn = ns.RU.compute_v2t_scores_x if qv else ns.RU.compute_t2v_scores_x
                ids, lab, msk = vtg if ftype == "vtg" else tvg
                t1 = time.time()
                S = fn(torch.full((n, n), -100.0), T(prob.v2t_sims if qv else prob.t2v_sims), 0, ids, msk, lab, video, vocab, vlab, ddp, torch.device("cpu"), args,
                       forward_type=ftype, cpn=cpn).float().numpy()
                bad = ~np.isfinite(S)
                out[f"H16_{pname}"] = np.where(bad, np.float32(0), S); out[f"H16_{pname}_nonfinite"] = bad
                m = (out[f"S_{pname}"] != -100.0) & ~bad
                dev = float((np.abs(S[m] - out[f"S_{pname}"][m]) / np.abs(out[f"S_{pname}"][m])).max()) if m.any() else float("nan")
                print(f"[{case}] reference .half() pass {pname}: {time.time() - t1:.1f}s, non-finite {int(bad.sum())}, worst deviation from its own fp32 run {dev:.2e}", flush=True)
    out["meta_case"] = np.array(case)
    path = os.path.join(out_dir, f"{case}.npz")
    np.savez_compressed(path, **out)
    print(f"[{case}] wrote {path} ({os.path.getsize(path) / 1e6:.2f} MB)", flush=True)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "tests", "golden"))
    ap.add_argument("--case", default="heavy", choices=sorted(CASES))
    a = ap.parse_args()
    from oracle import ref_harness
    if not ref_harness.available():
        sys.exit("reference not present; fixtures can only be generated in the build container")
    main(a.out, a.case)
