"""TEST INFRASTRUCTURE ONLY -- generates tests/golden/*.npz by running the REFERENCE's own code.

Run in the build container only (needs /root/reference):  python -m oracle.gen_golden [--case tiny|wide|all]

For each case it builds the reference model (oracle/ref_harness.py) with this repo's seeded
synthetic weights, builds a seeded synthetic retrieval problem (blim_amd/synth.py), drives the
reference's retrieval_utils.compute_v2t_scores_x / compute_t2v_scores_x for all six pass kinds,
its criteria, padding_ids, prepare_inputs_labels_for_multimodal, forward and get_recall, and stores
inputs' seeds + the reference's outputs.  Weights are NOT stored (regenerated from the seed).
The fixtures are data: no reference source text is stored.
"""
from __future__ import annotations

import argparse
import os
import sys
import time
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)

from blim_amd import synth  # noqa: E402
from oracle import ref_harness  # noqa: E402
from oracle.blim_oracle import OracleConfig  # noqa: E402

CASES = {
    # head_dim 128 everywhere (the engine's attention kernel is specialised for it); vocab must exceed 151645.
    "tiny": dict(dims=dict(vocab_size=151700, hidden_size=256, intermediate_size=512, num_layers=2, num_heads=2,
                           num_kv_heads=1, mm_hidden_size=64), wseed=11, pseed=5, n=6, tok_per_clip=8, text_len=(3, 9),
                 topk=4, bs=3),
    # 7B width, one layer: GQA 28/4, I=18944, real vocab size.
    "wide": dict(dims=dict(vocab_size=152064, hidden_size=3584, intermediate_size=18944, num_layers=1, num_heads=28,
                           num_kv_heads=4, mm_hidden_size=1024), wseed=12, pseed=6, n=4, tok_per_clip=6, text_len=(4, 10),
                 topk=3, bs=2),
}


def run_case(name: str, out_dir: str) -> None:
    import torch
    torch.set_num_threads(8)
    spec = CASES[name]
    dims = synth.ModelDims(**spec["dims"])
    ocfg = OracleConfig(**spec["dims"])
    t0 = time.time()
    weights = synth.synthetic_weights(dims, spec["wseed"])
    prob = synth.make_problem(spec["pseed"], spec["n"], dims, tok_per_clip=spec["tok_per_clip"], text_len=spec["text_len"])
    print(f"[{name}] weights+problem built in {time.time() - t0:.1f}s", flush=True)
    ns = ref_harness.load()
    RU, TU = ns.RU, ns.TU
    model = ref_harness.build_model(ocfg, weights)
    del weights
    model.set_tvg_prefix_length(prob.tvg_prefix_length)
    ddp = ref_harness.DDPish(model)
    dev = torch.device("cpu")
    T = lambda a: torch.from_numpy(np.asarray(a))
    tok = types.SimpleNamespace(pad_token_id=synth.PAD_ID)
    out = {}

    # --- padding_ids (retrieval_utils.py:155-167)
    vtg = RU.padding_ids([T(x) for x in prob.vtg_ids], [T(x) for x in prob.vtg_labels], [T(x) for x in prob.vtg_masks], tok)
    tvg = RU.padding_ids([T(x) for x in prob.tvg_ids], [T(x) for x in prob.tvg_labels], [T(x) for x in prob.tvg_masks], tok)
    for k, t in zip(("ids", "labels", "masks"), vtg):
        out[f"pad_vtg_{k}"] = t.numpy()
    for k, t in zip(("ids", "labels", "masks"), tvg):
        out[f"pad_tvg_{k}"] = t.numpy()

    video = [T(v) for v in prob.video]
    vocab = T(prob.video_vocab)
    vlab = T(prob.tvg_video_labels)
    args = types.SimpleNamespace(topk=spec["topk"], batch_size_eval=spec["bs"], num_clips=dims.num_clips)
    n = spec["n"]

    # --- prepare_inputs_labels_for_multimodal + forward on one ragged batch (both layouts)
    with torch.no_grad():
        sel = list(range(min(n, 3)))
        for kind, (ids, lab, msk) in (("vtg", vtg), ("tvg", tvg)):
            r = model.prepare_inputs_labels_for_multimodal(ids[sel], None, msk[sel], None, lab[sel], [video[i] for i in sel],
                                                           ["video"] * len(sel), image_sizes=None, video_feature=True,
                                                           tvg=(kind == "tvg"), cpn=True)
            (_, _, (m, cm), _, emb, lab2) = r
            out[f"prep_{kind}_mask"] = m.numpy(); out[f"prep_{kind}_cpn_mask"] = cm.numpy()
            out[f"prep_{kind}_embeds"] = emb.numpy(); out[f"prep_{kind}_labels"] = lab2.numpy()
            for tag, mm in (("", m), ("_cpn", cm)):
                o = model(inputs_embeds=emb, attention_mask=mm)
                out[f"fwd_{kind}{tag}_hidden"] = o.hidden_states.numpy()
                if kind == "vtg":
                    out[f"fwd_{kind}{tag}_score"] = RU.vtg_criterion(o.logits, lab2).numpy()
                    if name == "tiny" and tag == "":
                        # logits only at the label positions (full [B,L,V] is too large to commit)
                        pos = (lab2[0, 1:] != -100).nonzero()[:, 0]
                        out["fwd_vtg_logits_row0_pos"] = pos.numpy()
                        out["fwd_vtg_logits_row0_sub"] = o.logits[0, pos][:, ::997].numpy()

        # --- six passes through the reference's scoring loops
        sims_v2t, sims_t2v = T(prob.v2t_sims), T(prob.t2v_sims)
        passes = [
            ("v2t_vtg", RU.compute_v2t_scores_x, sims_v2t, vtg, "vtg", False),
            ("v2t_vtg_cpn", RU.compute_v2t_scores_x, sims_v2t, vtg, "vtg", True),
            ("v2t_tvg", RU.compute_v2t_scores_x, sims_v2t, tvg, "tvg", False),
            ("t2v_vtg", RU.compute_t2v_scores_x, sims_t2v, vtg, "vtg", False),
            ("t2v_tvg", RU.compute_t2v_scores_x, sims_t2v, tvg, "tvg", False),
            ("t2v_tvg_cpn", RU.compute_t2v_scores_x, sims_t2v, tvg, "tvg", True),
        ]
        for pname, fn, sims, (ids, lab, msk), ftype, cpn in passes:
            t0 = time.time()
            S = torch.full((n, n), -100.0)
            S = fn(S, sims, 0, ids, msk, lab, video, vocab, vlab, ddp, dev, args, forward_type=ftype, cpn=cpn)
            out[f"S_{pname}"] = S.numpy()
            print(f"[{name}] pass {pname}: {time.time() - t0:.1f}s", flush=True)

    # --- criteria on random logits incl. ignored labels (retrieval_utils.py:18-43)
    g = torch.Generator().manual_seed(0)
    lg = torch.randn(3, 7, 50, generator=g)
    lb = torch.randint(0, 50, (3, 7), generator=g)
    lb[0, :3] = -100; lb[2, 5:] = -100
    out["crit_vtg_logits"] = lg.numpy(); out["crit_vtg_labels"] = lb.numpy()
    out["crit_vtg_out"] = RU.vtg_criterion(lg, lb).numpy()
    lg2 = torch.randn(5, 4, 9, generator=g); lb2 = torch.randint(0, 9, (5, 4), generator=g)
    out["crit_tvg_logits"] = lg2.numpy(); out["crit_tvg_labels"] = lb2.numpy()
    out["crit_tvg_out"] = RU.tvg_criterion(lg2, lb2).numpy()

    # --- recall + ensemble (training_utils.py:150-221)
    rs = np.random.RandomState(3)
    a, b = rs.randn(50, 50).astype(np.float32), rs.randn(50, 50).astype(np.float32)
    a += 2 * np.eye(50, dtype=np.float32)
    ids = {i: i for i in range(50)}
    rec = TU.get_recall(a, b, ids, ids)
    out["recall_t2v"] = a; out["recall_v2t"] = b
    out["recall_keys"] = np.array(sorted(rec.keys())); out["recall_vals"] = np.array([rec[k] for k in sorted(rec.keys())])
    bz = b.copy(); bz[3, 4] = 0.0
    rec0 = TU.get_recall(a, bz, ids, ids)
    out["recall_zero_vals"] = np.array([rec0[k] for k in sorted(rec0.keys())])

    if name != "tiny":   # keep wide fixtures small: every 16th column of the [B,L,H] tensors
        for k in list(out):
            if (k.startswith("prep_") and k.endswith("_embeds")) or k.endswith("_hidden"):
                out[k + "_sub16"] = out.pop(k)[..., ::16].copy()
    out["meta_case"] = np.array(name)
    path = os.path.join(out_dir, f"{name}.npz")
    np.savez_compressed(path, **out)
    print(f"[{name}] wrote {path} ({os.path.getsize(path) / 1e6:.2f} MB)", flush=True)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--case", default="all")
    ap.add_argument("--out", default=os.path.join(ROOT, "tests", "golden"))
    a = ap.parse_args()
    if not ref_harness.available():
        sys.exit("reference not present; fixtures can only be generated in the build container")
    os.makedirs(a.out, exist_ok=True)
    for c in (CASES if a.case == "all" else [a.case]):
        run_case(c, a.out)
