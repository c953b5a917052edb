"""Does `--vtg_precise auto` send an evaluation to the compensated mode (0.69x the rate) for nothing?  (/root/reference/retrieval_utils.py:218-250)

For one weight set and one evaluation size: (1) the calibration exactly as evaluation() runs it -- the 256-pair sample, and the 2,048-pair confirmation sample when the
first one's extrapolation alone misses the bar (blim_amd/calibration.py) -- with the decision of the round-5 rule (first sample only) next to the round-6 rule; (2) the
ground truth: `--limit` v2t VTG pairs of the evaluation itself (the top-k texts of the first queries) scored plain AND fully compensated, entries over 1e-3 counted,
pairs/s of both modes.  A false reject = the rule says `full` while no entry of the population is near the bar; a true reject must stay rejected.

    python tools/calibrator_false_rejects.py --weights gaussian --n 4917 --topk 32 [--limit 40000]
"""
import argparse, json, os, sys, time, types
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from blim_amd import retrieval_utils as RU, synth
from blim_amd.modeling import BlimModel, DDPLike

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=4917)
ap.add_argument("--topk", type=int, default=32)
ap.add_argument("--limit", type=int, default=40000)
ap.add_argument("--weights", default="gaussian", choices=["gaussian", "sink7b", "heavy7b"])
ap.add_argument("--cpn", type=int, default=1)
a = ap.parse_args()
dims = synth.ModelDims()
model = BlimModel(dims, max_positions=1024, dtype="f16")
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
video = [torch.from_numpy(v).half() for v in prob.video]
model.vtg_precise = "auto"
sc = RU.PairScorer(DDPLike(model), vtg[0], vtg[2], vtg[1], tvg[0], tvg[2], tvg[1], video, torch.from_numpy(prob.video_vocab), torch.from_numpy(prob.tvg_video_labels),
                   dims.num_clips, max_tokens=32768)
sims = torch.from_numpy(prob.v2t_sims)
k = min(a.topk, a.n)
n_eval = a.n * k * (2 if a.cpn else 1) + a.n * k                       # VTG-type entries of the whole evaluation, as evaluation() counts them
first = RU.calibration_pairs(sims, a.topk, n_queries=32, per_query=8)
confirm = RU.calibration_pairs(sims, a.topk, n_queries=256, per_query=8)

torch.cuda.synchronize(); t0 = time.time()
old_choice, old_table = sc.calibrate_vtg(first, n_eval=n_eval)                                     # round 5: the first sample decides
torch.cuda.synchronize(); t_old = time.time() - t0
torch.cuda.synchronize(); t0 = time.time()
new_choice, new_table = sc.calibrate_vtg(first, n_eval=n_eval, confirm_pairs=confirm)              # round 6
torch.cuda.synchronize(); t_new = time.time() - t0

pairs = RU._topk_pairs(sims, 0, a.topk, True)[: a.limit]


def run(mode):
    sc.set_vtg_mode(mode)
    sc.vtg(pairs[:256])                                            # feature rows of the mode's layout, warm
    torch.cuda.synchronize(); t0 = time.time()
    out = sc.vtg(pairs).astype(np.float64)
    torch.cuda.synchronize()
    return out, time.time() - t0


ref, t_full = run("full")
got, t_plain = run("none")
dev = np.abs(got - ref) / np.abs(ref)
q = np.quantile(dev, [0.5, 0.99, 0.999])
e = new_table["none"]
row = {"weights": a.weights, "n": a.n, "topk": a.topk, "n_eval_vtg": n_eval, "population": int(len(pairs)),
       "plain_vs_full": {"max": float(dev.max()), "rms": float(np.sqrt(np.mean(dev ** 2))), "p50": float(q[0]), "p99": float(q[1]), "p99.9": float(q[2]),
                         "over_1e-3": int((dev > 1e-3).sum()), "over_8e-4": int((dev > 8e-4).sum())},
       "pairs_per_s": {"plain": round(len(pairs) / t_plain, 1), "full": round(len(pairs) / t_full, 1)},
       "round5_rule": {"chosen": old_choice, **{k_: old_table["none"][k_] for k_ in ("max", "rms", "pred", "n")}, "seconds": round(t_old, 2)},
       "round6_rule": {"chosen": new_choice, "first": {k_: e[k_] for k_ in ("max", "rms", "pred", "n")}, "confirm": e.get("confirm"), "seconds": round(t_new, 2)}}
truth_ok = row["plain_vs_full"]["over_1e-3"] == 0
row["verdict"] = ("plain holds over the population; " if truth_ok else f"plain FAILS over the population ({row['plain_vs_full']['over_1e-3']} entries over 1e-3); ") + \
                 f"round-5 rule -> {old_choice}, round-6 rule -> {new_choice}"
print(f"[{a.weights}, N = {a.n}, top-{a.topk}: {n_eval} VTG-type entries] population of {len(pairs)} v2t VTG pairs, plain vs fully compensated: max {dev.max():.2e} rms {row['plain_vs_full']['rms']:.2e} "
      f"99.9 % {q[2]:.2e}, {row['plain_vs_full']['over_1e-3']} over 1e-3; plain {row['pairs_per_s']['plain']:.0f} / full {row['pairs_per_s']['full']:.0f} pairs/s | "
      f"round-5 rule: sample max {old_table['none']['max']:.1e} rms {old_table['none']['rms']:.1e} pred {old_table['none']['pred']:.1e} -> {old_choice} ({t_old:.1f} s) | "
      f"round-6 rule: " + (f"confirmation sample of {e['confirm']['n']}: max {e['confirm']['max']:.1e} rms {e['confirm']['rms']:.1e} pred {e['confirm']['pred']:.1e}" if "confirm" in e else "first sample decides")
      + f" -> {new_choice} ({t_new:.1f} s)", flush=True)
print(json.dumps(row), flush=True)
model.engine.close()
