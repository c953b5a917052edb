"""Experiment (round 4): how much of the compensated mode's accuracy is carried by a FEW outlier channels?  In the fully compensated VTG mode the lo halves of
the GEMM inputs are zeroed except in the k channels with the largest |hi| of the call (engine options oc_k_x / oc_k_attn / oc_k_act / oc_k_qkv; -1 = keep all,
0 = keep none) -- numerically what a plain 16-bit GEMM input with k K-augmented outlier channels would carry -- and the v2t VTG scores of a whole N x top-16
evaluation are compared with the fully compensated ones.

    python tools/oc_experiment.py [--weights heavy7b|sink7b|gaussian] [--n 1000]
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
ap.add_argument("--weights", default="heavy7b", choices=["gaussian", "sink7b", "heavy7b"])
ap.add_argument("--configs", default="")
a = ap.parse_args()
dims = synth.ModelDims()
model = BlimModel(dims, max_positions=1024, dtype="f16")
wseed = 0
if a.weights != "gaussian":
    from oracle.gen_golden_heavy import CASES, heavy_items
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
model.vtg_precise = "full"
sc = RU.PairScorer(DDPLike(model), vtg[0], vtg[2], vtg[1], tvg[0], tvg[2], tvg[1], video, torch.from_numpy(prob.video_vocab), torch.from_numpy(prob.tvg_video_labels),
                   dims.num_clips, max_tokens=32768)
pairs = RU._topk_pairs(torch.from_numpy(prob.v2t_sims), 0, 16, True)
E = model.engine

def run(mode, **oc):
    sc.set_vtg_mode(mode)
    for k in ("oc_k_x", "oc_k_attn", "oc_k_act", "oc_k_qkv"):
        E.set_option(k, int(oc.get(k, -1)))
    torch.cuda.synchronize(); t0 = time.time()
    out = sc.vtg(pairs).astype(np.float64)
    torch.cuda.synchronize()
    return out, time.time() - t0

ref, t_ref = run("full")
print(f"[{a.weights}] N = {a.n}: {len(pairs)} v2t VTG pairs; fully compensated {t_ref:.1f}s", flush=True)
K = lambda x, at, ac, q: dict(oc_k_x=x, oc_k_attn=at, oc_k_act=ac, oc_k_qkv=q)
configs = [("plain (none)", "none", {}), ("qk", "qk", {}), ("qkx", "qkx", {}), ("attn", "attn", {}),
           ("oc 64 on x / attn / act, q-k-v hi+lo", "full", K(64, 64, 64, -1)),
           ("oc 64 on x / attn / act, q-k-v plain", "full", K(64, 64, 64, 0)),
           ("oc 16 on x / attn / act, q-k-v hi+lo", "full", K(16, 16, 16, -1)),
           ("oc 64 on x only (attn, act plain), q-k-v hi+lo", "full", K(64, 0, 0, -1)),
           ("oc 64 on x and act (attn plain), q-k-v hi+lo", "full", K(64, 0, 64, -1)),
           ("oc 64 on x only, q-k-v plain", "full", K(64, 0, 0, 0)),
           ("no lo anywhere (= plain, through the compensated kernels)", "full", K(0, 0, 0, 0)),
           ("all lo on x, none elsewhere, q-k-v hi+lo", "full", K(-1, 0, 0, -1)),
           ("oc 256 on x / attn / act, q-k-v hi+lo", "full", K(256, 256, 256, -1))]
if a.configs:
    sel = [int(x) for x in a.configs.split(",")]
    configs = [configs[i] for i in sel]
rows = []
for name, mode, oc in configs:
    got, dt = run(mode, **oc)
    dev = np.abs(got - ref) / np.abs(ref)
    q = np.quantile(dev, [0.5, 0.99, 0.999])
    rows.append({"config": name, "max": float(dev.max()), "rms": float(np.sqrt(np.mean(dev ** 2))), "p50": float(q[0]), "p99": float(q[1]), "p99.9": float(q[2]), "over_1e-3": int((dev > 1e-3).sum()),
                 "seconds": round(dt, 1)})
    r = rows[-1]
    print(f"{name:62s} max {r['max']:.2e} rms {r['rms']:.2e} p50 {r['p50']:.2e} p99 {r['p99']:.2e} p99.9 {r['p99.9']:.2e} over {r['over_1e-3']:5d}  ({dt:.1f}s)", flush=True)
print(json.dumps({"weights": a.weights, "n": a.n, "pairs": len(pairs), "rows": rows}))
