"""bf16 engines, `--second_pass auto` (round 6): is its decision right?  The calibration as evaluation() runs it (PairScorer.calibrate_second_pass: the e2m3 second pass measured
against the bf16 one on up to 256 / 2,048 of the evaluation's own pairs) next to the ground truth: the 16 N v2t VTG pairs of an N x top-16 evaluation (real 7B configuration,
reference-shaped rows) scored with BOTH second passes, entries over the 1e-3 bar counted, pairs/s of both.

    python tools/bf16_second_pass_population.py [--weights gaussian|sink7b|heavy7b] [--n 1000]
"""
import argparse, json, os, sys, time, types
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from blim_amd import retrieval_utils as RU, synth
from blim_amd.modeling import BlimModel, DDPLike

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=1000)
ap.add_argument("--topk", type=int, default=16)
ap.add_argument("--weights", default="gaussian", choices=["gaussian", "sink7b", "heavy7b"])
a = ap.parse_args()
dims = synth.ModelDims()
model = BlimModel(dims, max_positions=1024, dtype="bf16")
wseed = 0
if a.weights != "gaussian":
    from oracle.gen_golden_heavy import CASES, heavy_items          # (development aid: the reshaped tensors of the trained-like fixtures)
    spec = CASES[a.weights]; wseed = spec["wseed"]
model.engine.init_synthetic_weights(wseed)
if a.weights != "gaussian":
    for name, arr in heavy_items(dims, wseed, only_changed=True, sink=bool(spec.get("sink", False))):
        model.engine.load_weight(name, arr)
prob = synth.make_problem(1, a.n, dims, tok_per_clip=64, fast_video=True)
model.set_tvg_prefix_length(prob.tvg_prefix_length)
tok = types.SimpleNamespace(pad_token_id=synth.PAD_ID)
Tt = lambda rows: [torch.from_numpy(r) for r in rows]
vtg = RU.padding_ids(Tt(prob.vtg_ids), Tt(prob.vtg_labels), Tt(prob.vtg_masks), tok)
tvg = RU.padding_ids(Tt(prob.tvg_ids), Tt(prob.tvg_labels), Tt(prob.tvg_masks), tok)
video = [torch.from_numpy(v).to(torch.bfloat16) for v in prob.video]
model.vtg_precise = "full"
model.second_pass = "auto"
sc = RU.PairScorer(DDPLike(model), vtg[0], vtg[2], vtg[1], tvg[0], tvg[2], tvg[1], video, torch.from_numpy(prob.video_vocab), torch.from_numpy(prob.tvg_video_labels),
                   dims.num_clips, max_tokens=32768)
sims = torch.from_numpy(prob.v2t_sims)
k = min(a.topk, a.n)
n_eval = a.n * k * 2 + a.n * k
t0 = time.time()
chosen, table = sc.calibrate_second_pass(RU.calibration_pairs(sims, a.topk, n_queries=32, per_query=8), n_eval=n_eval,
                                         confirm_pairs=RU.calibration_pairs(sims, a.topk, n_queries=256, per_query=8))
torch.cuda.synchronize(); t_cal = time.time() - t0
pairs = RU._topk_pairs(sims, 0, a.topk, True)


def run(lo6):
    model.engine.set_option("precise_lo6", lo6)
    sc.set_vtg_mode("full")
    sc.vtg(pairs[:256])
    torch.cuda.synchronize(); t0 = time.time()
    out = sc.vtg(pairs).astype(np.float64)
    torch.cuda.synchronize()
    return out, time.time() - t0


ref, t16 = run(0)
got, t6 = run(1)
dev = np.abs(got - ref) / np.abs(ref)
q = np.quantile(dev, [0.5, 0.99, 0.999])
e = table["e2m3"]
row = {"weights": a.weights, "n": a.n, "population": int(len(pairs)), "e2m3_vs_bf16_second_pass": {"max": float(dev.max()), "rms": float(np.sqrt(np.mean(dev ** 2))), "p50": float(q[0]),
       "p99": float(q[1]), "p99.9": float(q[2]), "over_1e-3": int((dev > 1e-3).sum())}, "pairs_per_s": {"bf16_second_pass": round(len(pairs) / t16, 1), "e2m3": round(len(pairs) / t6, 1)},
       "auto": {"chosen": chosen, "sample": {k_: e[k_] for k_ in ("max", "rms", "pred", "n")}, "confirm": e.get("confirm"), "seconds": round(t_cal, 2)}}
print(f"[bf16 engine, {a.weights}, N = {a.n}] {len(pairs)} v2t VTG pairs, e2m3 second pass vs the bf16 one: max {dev.max():.2e} rms {row['e2m3_vs_bf16_second_pass']['rms']:.2e} 99.9 % {q[2]:.2e}, "
      f"{row['e2m3_vs_bf16_second_pass']['over_1e-3']} over 1e-3; {row['pairs_per_s']['bf16_second_pass']:.0f} / {row['pairs_per_s']['e2m3']:.0f} pairs/s | second_pass auto: sample max {e['max']:.1e} rms {e['rms']:.1e} "
      f"pred {e['pred']:.1e}" + (f", confirmation sample of {e['confirm']['n']}: max {e['confirm']['max']:.1e} pred {e['confirm']['pred']:.1e}" if "confirm" in e else "") + f" -> {chosen} ({t_cal:.1f} s)", flush=True)
print(json.dumps(row), flush=True)
model.engine.close()
