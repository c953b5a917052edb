"""Per-pass throughput of a full six-pass BLiM evaluation ("per pass kind and the blended figure").

    python tools/pass_bench.py [--n 96] [--topk 16] [--shape syn|ref|both] [--dtype f16|bf16] [--reps 3]

Shapes:  syn = BASELINE.json headline (96 video tokens + 32 text tokens, no prompt);  ref = reference-shaped rows
(ChatML header + 256 video tokens + instruction + caption of 8..48 tokens, TVG rows with the 21-token prefix).
For every pass kind the fused PairScorer plans are built once (host time reported separately) and the device time of
running them is measured between synchronisations.  Prints a markdown table + one JSON object.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

H, I, V, LAYERS = 3584, 18944, 152064, 28
FLOP_TOKEN_LAYER = 2 * H * (H + 2 * 512) + 2 * H * H + 6 * H * I
FLOP_HEAD_ROW = 2 * H * V


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=96, help="videos = texts")
    ap.add_argument("--topk", type=int, default=16)
    ap.add_argument("--shape", default="both", choices=["syn", "ref", "both"])
    ap.add_argument("--dtype", default=None)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--max_tokens", type=int, default=32768)
    a = ap.parse_args()

    import torch
    from blim_amd import retrieval_utils as RU
    from blim_amd import synth
    from blim_amd.modeling import BlimModel, DDPLike

    dims = synth.ModelDims()
    model = BlimModel(dims, max_positions=1024, dtype=a.dtype)
    model.engine.init_synthetic_weights(0)
    model.engine.reserve(a.max_tokens + 1024, a.max_tokens + 1024)
    tok = type("T", (), {"pad_token_id": synth.PAD_ID})()
    T = lambda rows: [torch.from_numpy(r) for r in rows]
    result = {}
    for shape in (["syn", "ref"] if a.shape == "both" else [a.shape]):
        if shape == "syn":
            prob = synth.make_problem(1000, a.n, dims, tok_per_clip=24, text_len=(32, 32), reference_layout=False)
        else:
            prob = synth.make_problem(1000, a.n, dims, tok_per_clip=64, text_len=(8, 48), reference_layout=True)
        vtg = RU.padding_ids(T(prob.vtg_ids), T(prob.vtg_labels), T(prob.vtg_masks), tok)
        tvg = RU.padding_ids(T(prob.tvg_ids), T(prob.tvg_labels), T(prob.tvg_masks), tok)
        model.set_tvg_prefix_length(prob.tvg_prefix_length)
        scorer = RU.PairScorer(DDPLike(model), vtg[0], vtg[2], vtg[1], tvg[0], tvg[2], tvg[1], [torch.from_numpy(v) for v in prob.video],
                               torch.from_numpy(prob.video_vocab), torch.from_numpy(prob.tvg_video_labels), dims.num_clips, max_tokens=a.max_tokens)
        pv = RU._topk_pairs(torch.from_numpy(prob.v2t_sims), 0, a.topk, True)
        pt = RU._topk_pairs(torch.from_numpy(prob.t2v_sims), 0, a.topk, False)
        passes = [("v2t candidate_likelihood (VTG)", "vtg", pv, False), ("v2t candidate_prior (VTG, CPN)", "vtg", pv, True),
                  ("v2t query_likelihood (TVG)", "tvg", pv, False), ("t2v query_likelihood (VTG)", "vtg", pt, False),
                  ("t2v candidate_likelihood (TVG)", "tvg", pt, False), ("t2v candidate_prior (TVG, CPN)", "tvg", pt, True)]
        rows = []
        tot_pairs, tot_ms = 0, 0.0
        for name, kind, pairs, cpn in passes:
            t0 = time.perf_counter()
            try:
                plans = scorer.plan_vtg(pairs, cpn) if kind == "vtg" else scorer.plan_tvg(pairs, cpn)
            except ValueError as e:                      # SYN rows have no prompt: the VTG prior is undefined there
                rows.append({"pass": name + " -- n/a: " + str(e), "pairs": 0, "device_ms": 0.0, "pairs_per_s": 0.0})
                continue
            torch.cuda.synchronize()
            t_plan = time.perf_counter() - t0
            print(f"[{shape}] {name}: {len(plans)} plans, tokens {[p.n_tokens for p in plans]}, rows {[p.n_rows for p in plans]}, pairs {[p.n_pairs for p in plans]}", file=sys.stderr, flush=True)
            for p in plans:
                scorer.run(p)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(a.reps):
                outs = [scorer.run(p) for p in plans]
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / a.reps * 1e3
            assert all(torch.isfinite(o).all() for o in outs)
            model.engine.timing_enable(True)
            for p in plans:
                scorer.run(p)
            rep = model.engine.timing_report()
            model.engine.timing_enable(False)
            tot_cls = sum(v["ms"] for v in rep.values())
            shares = {k: round(100 * v["ms"] / tot_cls, 1) for k, v in rep.items() if v["ms"] > 0.005 * tot_cls}
            print(f"[{shape}] {name}: kernel-class shares % {shares}", file=sys.stderr, flush=True)
            n_tok = sum(p.n_tokens for p in plans)
            n_rows = sum(p.n_rows for p in plans) if kind == "vtg" else 0
            flops = LAYERS * FLOP_TOKEN_LAYER * n_tok + FLOP_HEAD_ROW * n_rows
            rows.append({"pass": name, "pairs": int(len(pairs)), "engine_calls": len(plans), "packed_tokens": int(n_tok), "tokens_per_pair": round(n_tok / len(pairs), 1),
                         "device_ms": round(ms, 2), "plan_ms": round(t_plan * 1e3, 1), "pairs_per_s": round(len(pairs) / ms * 1e3, 1),
                         "executed_tflops": round(flops / ms / 1e9, 1)})
            tot_pairs += len(pairs); tot_ms += ms
        rows.append({"pass": "six-pass blend", "pairs": tot_pairs, "device_ms": round(tot_ms, 2), "pairs_per_s": round(tot_pairs / tot_ms * 1e3, 1)})
        result[shape] = rows
        print(f"\n### {shape.upper()} shape, N = {a.n} videos = texts, top-{a.topk}, {model.engine.dtype}\n")
        print("| pass | pairs | engine calls | packed tokens | tokens/pair | device ms | host plan ms | pairs/s | executed TFLOP/s |")
        print("|---|---|---|---|---|---|---|---|---|")
        for r in rows:
            print(f"| {r['pass']} | {r['pairs']} | {r.get('engine_calls', '')} | {r.get('packed_tokens', '')} | {r.get('tokens_per_pair', '')} | {r['device_ms']} | "
                  f"{r.get('plan_ms', '')} | {r['pairs_per_s']} | {r.get('executed_tflops', '')} |")
    print(json.dumps(result))


if __name__ == "__main__":
    main()
