"""Development aid: where the time of PairScorer.calibrate_tvg goes on the strong-scaling problem (N = 1000, 7B, synthetic weights)."""
import os, sys, time, types
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from blim_amd import retrieval_utils as RU, synth
from blim_amd.modeling import BlimModel, DDPLike

dims = synth.ModelDims()
model = BlimModel(dims, max_positions=1024, dtype="f16")
model.engine.init_synthetic_weights(0)
prob = synth.make_problem(1, 1000, dims, tok_per_clip=64, fast_video=True)
tok = types.SimpleNamespace(pad_token_id=synth.PAD_ID)
Tt = lambda rows: [torch.from_numpy(r) for r in rows]
vtg = RU.padding_ids(Tt(prob.vtg_ids), Tt(prob.vtg_labels), Tt(prob.vtg_masks), tok)
tvg = RU.padding_ids(Tt(prob.tvg_ids), Tt(prob.tvg_labels), Tt(prob.tvg_masks), tok)
video = [torch.from_numpy(v).half() for v in prob.video]
for nq in [int(x) for x in os.environ.get("NQ", "16,8,4").split(",")]:
    for rep in range(2):
        model.clear_cache()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        sc = RU.PairScorer(DDPLike(model), vtg[0], vtg[2], vtg[1], tvg[0], tvg[2], tvg[1], video, torch.from_numpy(prob.video_vocab), torch.from_numpy(prob.tvg_video_labels),
                           dims.num_clips, max_tokens=32768)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        tp = RU.calibration_pairs(torch.from_numpy(prob.t2v_sims), 16, n_queries=nq)
        pairs = np.stack([tp[:, 1], tp[:, 0]], axis=1)
        marks = []
        for mode in ("full", "attn"):
            sc.set_tvg_mode(mode)
            for cpn in (False, True):
                torch.cuda.synchronize(); a = time.perf_counter()
                plans = list(sc.iter_tvg(pairs, cpn))
                b = time.perf_counter()
                sc.tvg(pairs, cpn)
                torch.cuda.synchronize(); c = time.perf_counter()
                marks.append((mode, cpn, round(b - a, 4), round(c - b, 4), sum(p.n_tokens for p in plans)))
        print(f"nq={nq} rep={rep} scorer {t1 - t0:.3f}s  calibrate {time.perf_counter() - t1:.3f}s", marks, flush=True)
