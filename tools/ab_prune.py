"""A/B in one process: engine option prune_last (rows nobody reads skip the last layer's o_proj / MLP) on the benched SYN step and on a
reference-shaped VTG / TVG plan (256 video tokens)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
from blim_amd import retrieval_utils as RU, synth
from blim_amd.modeling import BlimModel, DDPLike

dims = synth.ModelDims()
model = BlimModel(dims, max_positions=1024, dtype="f16")
model.engine.init_synthetic_weights(0)
((sc, plan, prob, pairs),) = bench.build_step_plans(model, 0, 1, 55, 16)
rprob = synth.make_problem(1, 96, dims, tok_per_clip=64, fast_video=True)
model.set_tvg_prefix_length(rprob.tvg_prefix_length)
tok = type("T", (), {"pad_token_id": synth.PAD_ID})()
Tt = lambda rows: [torch.from_numpy(r) for r in rows]
vtg = RU.padding_ids(Tt(rprob.vtg_ids), Tt(rprob.vtg_labels), Tt(rprob.vtg_masks), tok)
tvg = RU.padding_ids(Tt(rprob.tvg_ids), Tt(rprob.tvg_labels), Tt(rprob.tvg_masks), tok)
rsc = RU.PairScorer(DDPLike(model), vtg[0], vtg[2], vtg[1], tvg[0], tvg[2], tvg[1], [torch.from_numpy(v) for v in rprob.video], torch.from_numpy(rprob.video_vocab),
                    torch.from_numpy(rprob.tvg_video_labels), dims.num_clips, max_tokens=32768)
rpairs = RU._topk_pairs(torch.from_numpy(rprob.v2t_sims), 0, 28, True)
rplan = rsc.plan_vtg(rpairs)[0]
tplan = rsc.plan_tvg(rpairs)[0]
print(f"SYN plan: {plan.n_tokens} tokens, {plan.n_rows} rows; REF VTG plan: {rplan.n_tokens} tokens, {rplan.n_rows} rows, {rplan.n_pairs} pairs; REF TVG plan: {tplan.n_tokens} tokens, {tplan.n_rows} rows")


def t(scorer, pl, n=4):
    scorer.run(pl); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        scorer.run(pl)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for name, scorer, pl in (("SYN VTG", sc, plan), ("REF VTG", rsc, rplan), ("REF TVG", rsc, tplan)):
    res = {0: [], 1: []}
    for rep in range(3):
        for on in (0, 1):
            model.engine.set_option("prune_last", on)
            res[on].append(t(scorer, pl))
    a, b = np.mean(res[0]), np.mean(res[1])
    print(f"{name}: prune off {a:.2f} ms, on {b:.2f} ms  ({100 * (a - b) / a:+.2f} %)")

# per-class device time of one reference-shaped VTG plan and one TVG plan (HIP-event class timers)
for name, scorer, pl in (("REF VTG", rsc, rplan), ("REF TVG", rsc, tplan)):
    model.engine.timing_enable(True)
    scorer.run(pl); torch.cuda.synchronize()
    rep = model.engine.timing_report()
    model.engine.timing_enable(False)
    tot = sum(v["ms"] for v in rep.values())
    print(name, "classes (ms, %):", {k: (round(v["ms"], 2), round(100 * v["ms"] / tot, 1)) for k, v in rep.items() if v["calls"]})
