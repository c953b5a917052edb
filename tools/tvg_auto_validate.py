"""Does the sample-based decision of `--tvg_precise auto` hold over a WHOLE evaluation?  Runs the fixed-size job of bench.py's strong-scaling leg
(N videos x N texts, top-16, six passes, real 7B configuration) twice -- TVG calls fully compensated, then in the mode `auto` picks from its 128-pair
sample -- and compares every computed entry of the three TVG-type matrices (2 x 16 N likelihood entries + 16 N prior entries) between the two runs.
The fully compensated mode sits at <= 1e-4 of the fp32 reference on every fixture (tests/test_gpu_parity.py), so the deviation from it is the quantity
the 1e-3 bar is about.  The VTG-type matrices must be bit-equal (the TVG mode does not touch them).

    python tools/tvg_auto_validate.py [--n 1000] [--weights gaussian|sink7b|heavy7b] [--dtype f16]
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
ap.add_argument("--dtype", default="f16")
ap.add_argument("--seed", type=int, default=1, help="seed of the synthetic problem (videos, texts, similarity matrices)")
ap.add_argument("--vtg", action="store_true", help="validate `--vtg_precise auto` the same way (the yardstick run then has its VTG calls fully compensated too: 2x the time)")
a = ap.parse_args()

dims = synth.ModelDims()
model = BlimModel(dims, max_positions=1024, dtype=a.dtype)
wseed = 0
if a.weights != "gaussian":
    from oracle.gen_golden_heavy import CASES, heavy_items          # (development aid: the reshaped tensors of the trained-like fixtures)
    spec = CASES[a.weights]
    wseed = spec["wseed"]
model.engine.init_synthetic_weights(wseed)
if a.weights != "gaussian":
    for name, arr in heavy_items(dims, wseed, only_changed=True, sink=bool(spec.get("sink", False))):
        model.engine.load_weight(name, arr)
prob = synth.make_problem(a.seed, a.n, dims, tok_per_clip=64, fast_video=True)
loader = synth.ProblemLoader(prob, 64, video_dtype=torch.float16)
tok = types.SimpleNamespace(pad_token_id=synth.PAD_ID)
nz = lambda x: np.where(x == 0, np.float32(1e-6), x)
dev = torch.device("cuda", 0)
out = {}
for mode in ("full", "auto"):
    args = types.SimpleNamespace(topk=a.topk, num_clips=dims.num_clips, cpn=True, resume="x", eval=True, dataset="MSRVTT", batch_size_eval=16,
                                 iv2_scores={"v2t": torch.from_numpy(nz(prob.v2t_sims)), "t2v": torch.from_numpy(nz(prob.t2v_sims))}, max_tokens=32768, dedup=True)
    model.clear_cache()
    model.tvg_precise = mode
    model.vtg_precise = (mode if a.vtg else None)
    t0 = time.time()
    t2v, v2t = RU.evaluation(DDPLike(model), loader, dev, tok, args)
    st = args._eval_stats
    out[mode] = (t2v, v2t, time.time() - t0, st.get("tvg_precise", mode), st.get("tvg_precise_table", {}), st.get("vtg_precise"), st.get("vtg_precise_table"))
(t2v_f, v2t_f, s_f, _, _, _, _), (t2v_a, v2t_a, s_a, chosen, table, vchosen, vtable) = out["full"], out["auto"]
rep = {"weights": a.weights, "dtype": a.dtype, "n": a.n, "seed": a.seed, "chosen": chosen, "sample_table": table, "vtg_chosen": vchosen, "vtg_sample_table": vtable,
       "seconds_full": round(s_f, 2), "seconds_auto": round(s_a, 2)}
for name, F, A in (("t2v candidate_likelihood (TVG)", t2v_f["candidate_likelihood"], t2v_a["candidate_likelihood"]),
                   ("t2v candidate_prior (TVG, CPN)", t2v_f["candidate_prior"], t2v_a["candidate_prior"]),
                   ("v2t query_likelihood (TVG)", v2t_f["query_likelihood"], v2t_a["query_likelihood"])):
    m = F != -100.0
    assert np.array_equal(m, A != -100.0)
    dev_ = np.abs(A[m].astype(np.float64) - F[m]) / np.abs(F[m])
    rep[name] = {"entries": int(m.sum()), "max": float(dev_.max()), "rms": float(np.sqrt(np.mean(dev_ ** 2))), "over_1e-3": int((dev_ > 1e-3).sum())}
for name, F, A in (("t2v query_likelihood (VTG)", t2v_f["query_likelihood"], t2v_a["query_likelihood"]), ("v2t candidate_likelihood (VTG)", v2t_f["candidate_likelihood"], v2t_a["candidate_likelihood"]),
                   ("v2t candidate_prior (VTG, CPN)", v2t_f["candidate_prior"], v2t_a["candidate_prior"])):
    rep[name + " bit-equal"] = bool(np.array_equal(F, A))
    if a.vtg:
        m = F != -100.0
        dev_ = np.abs(A[m].astype(np.float64) - F[m]) / np.abs(F[m])
        rep[name] = {"entries": int(m.sum()), "max": float(dev_.max()), "rms": float(np.sqrt(np.mean(dev_ ** 2))), "over_1e-3": int((dev_ > 1e-3).sum())}
if os.environ.get("DUMP"):
    F, A = v2t_f["candidate_likelihood"], v2t_a["candidate_likelihood"]
    m = F != -100.0
    jj, ii = np.nonzero(m)
    dev_ = np.abs(A[m].astype(np.float64) - F[m]) / np.abs(F[m])
    tl = np.array([int((np.asarray(l) != -100).sum()) for l in prob.vtg_labels])
    order = np.argsort(-dev_)[:40]
    print("top deviations v2t VTG: (video, text, dev, score_full, score_auto, text_len)")
    for k in order:
        print(int(jj[k]), int(ii[k]), f"{dev_[k]:.2e}", f"{F[m][k]:.4f}", f"{A[m][k]:.4f}", int(tl[ii[k]]))
    q = np.quantile(dev_, [0.5, 0.9, 0.99, 0.999, 1.0])
    print("quantiles 50/90/99/99.9/100:", [f"{x:.2e}" for x in q])
    bad = dev_ > 1e-3
    print("distinct videos among outliers:", len(set(jj[bad])), "distinct texts:", len(set(ii[bad])), "of", int(bad.sum()))
    for L in sorted(set(tl)):
        sel = tl[ii] == L
        if sel.sum() > 200: print("text_len", L, "n", int(sel.sum()), "rms", f"{np.sqrt(np.mean(dev_[sel]**2)):.2e}", "max", f"{dev_[sel].max():.2e}")
    np.savez_compressed(os.path.join(ROOT, "gpurun_out", f"dev_{a.weights}.npz"), dev=dev_.astype(np.float32), j=jj.astype(np.int32), i=ii.astype(np.int32), full=F[m], auto=A[m], tl=tl)
print(json.dumps(rep, indent=1))
