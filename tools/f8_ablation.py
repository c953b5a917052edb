"""fp8 mode: which GEMMs carry the deviation?  Full 28-layer 7B dims, synthetic weights; scores of 18 pairs in fp16 vs fp8 with
the fp8 set restricted by the engine option "f8_mask" (1 qkv, 2 o_proj, 4 gate|up, 8 down, 16 lm_head), plus step time."""
import os, sys, types, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from blim_amd import retrieval_utils as RU, synth
from blim_amd.modeling import BlimModel, DDPLike

dims = synth.ModelDims()
prob = synth.make_problem(21, 12, dims, tok_per_clip=24, text_len=(5, 32), reference_layout=True)
tok = types.SimpleNamespace(pad_token_id=synth.PAD_ID)
Tt = lambda rows: [torch.from_numpy(r) for r in rows]
vtg = RU.padding_ids(Tt(prob.vtg_ids), Tt(prob.vtg_labels), Tt(prob.vtg_masks), tok)
tvg = RU.padding_ids(Tt(prob.tvg_ids), Tt(prob.tvg_labels), Tt(prob.tvg_masks), tok)
pairs = np.array([[j, i] for j in range(12) for i in range(12)])


def scores(dtype, mask=None):
    model = BlimModel(dims, max_positions=1024, dtype=dtype)
    model.engine.init_synthetic_weights(0)
    if mask is not None:
        model.engine.set_option("f8_mask", mask)
    model.set_tvg_prefix_length(prob.tvg_prefix_length)
    sc = RU.PairScorer(DDPLike(model), vtg[0], vtg[2], vtg[1], tvg[0], tvg[2], tvg[1], [torch.from_numpy(v) for v in prob.video],
                       torch.from_numpy(prob.video_vocab), torch.from_numpy(prob.tvg_video_labels), dims.num_clips)
    out = (sc.vtg(pairs), sc.tvg(pairs))
    model.engine.close()
    return out


ref = scores("f16")
for mask, name in ((31, "all"), (12, "MLP only (gate|up, down)"), (28, "MLP + lm_head"), (3, "qkv + o only"), (4, "gate|up only"), (8, "down only"), (16, "lm_head only"), (1, "qkv only"), (2, "o only")):
    got = scores("f8", mask)
    dv = [np.abs(a - b) / np.abs(b) for a, b in zip(got, ref)]
    # rank agreement: per video query (rows of 12 texts), does the arg-max text agree?
    agree = [float((np.argmax(a.reshape(12, 12), 1) == np.argmax(b.reshape(12, 12), 1)).mean()) for a, b in zip(got, ref)]
    print(f"mask {mask:2d} {name:28s}: VTG max {dv[0].max():.2e} mean {dv[0].mean():.2e} top-1 agree {agree[0]:.2f} | TVG max {dv[1].max():.2e} mean {dv[1].mean():.2e} top-1 agree {agree[1]:.2f}", flush=True)
