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
    # ---- depth: all 28 layers.
    # `deep`: 28 layers at H=1024 (8 q heads / 2 kv heads of 128, I=2816, real vocab); reference-shaped ragged rows, plus a SYN
    # sub-problem (BASELINE.json's headline row: 96 video + 32 text tokens, top-16) scored for `queries` query rows.
    "deep": dict(dims=dict(vocab_size=152064, hidden_size=1024, intermediate_size=2816, num_layers=28, num_heads=8,
                           num_kv_heads=2, mm_hidden_size=256), wseed=13, pseed=7, n=8, tok_per_clip=16, text_len=(4, 24),
                 topk=4, bs=3, sub16=True, lazy=True,
                 syn=dict(pseed=8, n=16, tok_per_clip=24, text_len=(32, 32), topk=16, bs=16, queries=2,
                          passes=("v2t_vtg", "t2v_vtg", "v2t_tvg", "t2v_tvg", "t2v_tvg_cpn"))),
    # `full7b`: the real Qwen2-7B configuration (28 layers, H=3584, 28/4 heads, I=18944, V=152064), weight seed 0 = the weights
    # bench.py runs on.  2 query rows x top-4 (bs 3: leftover batch) per pass kind on reference-shaped rows, and one SYN query
    # (96 + 32 tokens, top-16) per direction.  fp32 reference = 30.5 GB of weights, streamed tensor by tensor.
    "full7b": dict(dims=dict(vocab_size=152064, hidden_size=3584, intermediate_size=18944, num_layers=28, num_heads=28,
                             num_kv_heads=4, mm_hidden_size=1024), wseed=0, pseed=9, n=6, tok_per_clip=6, text_len=(4, 10),
                   topk=4, bs=3, queries=2, sub16=True, lazy=True,
                   syn=dict(pseed=10, n=16, tok_per_clip=24, text_len=(32, 32), topk=16, bs=16, queries=1,
                            passes=("v2t_vtg", "t2v_vtg", "v2t_tvg", "t2v_tvg", "t2v_tvg_cpn"))),
    # `full7b_ref`: the same 7B model on REFERENCE-SIZED rows: 4 x 64 = 256 video tokens (base_dataset.py:28), captions of 8-48 tokens
    # (VTG rows of ~290-330 tokens), one query row x top-4 per pass kind.
    "full7b_ref": dict(dims=dict(vocab_size=152064, hidden_size=3584, intermediate_size=18944, num_layers=28, num_heads=28,
                                 num_kv_heads=4, mm_hidden_size=1024), wseed=0, pseed=11, n=4, tok_per_clip=64, text_len=(8, 48),
                       topk=4, bs=4, queries=1, sub16=True, lazy=True),
    # `full7b_bench`: the real 7B model on the problem bench.py's first plan is built from (synth.make_problem(1000, 55, tok_per_clip=24,
    # text_len=(32, 32), reference_layout=False): 55 videos / texts, 96 video + 32 text tokens, top-16) -- the first 4 query rows of each
    # direction through the reference's own loops (64 + 64 VTG entries that sit INSIDE the benched 880-pair step).
    "full7b_bench": dict(dims=dict(vocab_size=152064, hidden_size=3584, intermediate_size=18944, num_layers=28, num_heads=28,
                                   num_kv_heads=4, mm_hidden_size=1024), wseed=0, pseed=1000, n=55, tok_per_clip=24, text_len=(32, 32),
                         topk=16, bs=16, queries=4, lazy=True, syn_only=True,
                         syn=dict(pseed=1000, n=55, tok_per_clip=24, text_len=(32, 32), topk=16, bs=16, queries=4,
                                  passes=("v2t_vtg", "t2v_vtg", "v2t_tvg", "t2v_tvg", "t2v_tvg_cpn"))),
}
PASS_KINDS = {  # name -> (query is video, forward_type, cpn)
    "v2t_vtg": (True, "vtg", False), "v2t_vtg_cpn": (True, "vtg", True), "v2t_tvg": (True, "tvg", False),
    "t2v_vtg": (False, "vtg", False), "t2v_tvg": (False, "tvg", False), "t2v_tvg_cpn": (False, "tvg", True),
}


def fast_tensor(seed: int, name: str, shape, std: float, mean: float = 0.0, threads: int = 8, chunk: int = 1 << 21) -> np.ndarray:
    """synth.tensor (same values, checked below on the first chunk) generated by a thread pool: 7.6 G elements in minutes."""
    from concurrent.futures import ThreadPoolExecutor
    n = int(np.prod(shape)); tid = synth.fnv1a64(name); out = np.empty(n, np.float32)
    sc, mu = synth.bell_scale(std), np.float32(mean)
    u = np.uint64

    def work(lo):
        hi = min(n, lo + chunk)
        z = synth._hash(seed, tid, np.arange(lo, hi, dtype=np.uint64))
        s = ((z & u(0xFFFF)) + ((z >> u(16)) & u(0xFFFF)) + ((z >> u(32)) & u(0xFFFF)) + (z >> u(48))).astype(np.int64)
        out[lo:hi] = synth.bf16_round((s - 131070).astype(np.float32) * sc + mu)

    with ThreadPoolExecutor(threads) as ex:
        list(ex.map(work, range(0, n, chunk)))
    m = min(n, 4096)
    assert np.array_equal(out[:m], synth.bf16_round(synth.bell_f32(seed, name, m, std, mean)))
    return out.reshape(shape)


class LazyWeights:
    """dict-like view of synth.synthetic_weights(dims, seed) that materialises one tensor at a time."""

    def __init__(self, dims, seed):
        self.dims, self.seed = dims, seed

    def items(self):
        for name, shape in synth.weight_shapes(self.dims).items():
            yield name, fast_tensor(self.seed, name, shape, *synth.weight_dist(name))


def problem_of(spec, dims, sub=None):
    s = spec if sub is None else sub
    return synth.make_problem(s["pseed"], s["n"], dims, tok_per_clip=s["tok_per_clip"], text_len=s["text_len"],
                              reference_layout=(sub is None))


def run_passes(out, prefix, RU, ddp, dev, prob, spec, dims, names, tag):
    """Drives the reference's compute_*_scores_x for the named pass kinds on `prob`; query rows = the first spec['queries']."""
    import torch
    T = lambda a: torch.from_numpy(np.asarray(a))
    tok = types.SimpleNamespace(pad_token_id=synth.PAD_ID)
    vtg = RU.padding_ids([T(x) for x in prob.vtg_ids], [T(x) for x in prob.vtg_labels], [T(x) for x in prob.vtg_masks], tok)
    tvg = RU.padding_ids([T(x) for x in prob.tvg_ids], [T(x) for x in prob.tvg_labels], [T(x) for x in prob.tvg_masks], tok)
    video = [T(v) for v in prob.video]
    vocab, vlab = T(prob.video_vocab), T(prob.tvg_video_labels)
    args = types.SimpleNamespace(topk=spec["topk"], batch_size_eval=spec["bs"], num_clips=dims.num_clips)
    n, q = spec["n"], spec.get("queries", spec["n"])
    with torch.no_grad():
        for pname in names:
            qv, ftype, cpn = PASS_KINDS[pname]
            fn = RU.compute_v2t_scores_x if qv else RU.compute_t2v_scores_x
            sims = T(prob.v2t_sims if qv else prob.t2v_sims)[:q]
            ids, lab, msk = vtg if ftype == "vtg" else tvg
            t0 = time.time()
            S = fn(torch.full((n, n), -100.0), sims, 0, ids, msk, lab, video, vocab, vlab, ddp, dev, args, forward_type=ftype, cpn=cpn)
            out[f"{prefix}{pname}"] = S.numpy()
            print(f"[{tag}] pass {prefix}{pname}: {time.time() - t0:.1f}s", flush=True)
    return vtg, tvg, video


def run_case(name: str, out_dir: str) -> None:
    import torch
    torch.set_num_threads(8)
    spec = CASES[name]
    dims = synth.ModelDims(**spec["dims"])
    ocfg = OracleConfig(**spec["dims"])
    t0 = time.time()
    weights = LazyWeights(dims, spec["wseed"]) if spec.get("lazy") else synth.synthetic_weights(dims, spec["wseed"])
    prob = problem_of(spec, dims, spec["syn"] if spec.get("syn_only") else None)
    print(f"[{name}] weights+problem built in {time.time() - t0:.1f}s", flush=True)
    ns = ref_harness.load()
    RU, TU = ns.RU, ns.TU
    t0 = time.time()
    model = ref_harness.build_model(ocfg, weights)
    del weights
    print(f"[{name}] reference model built in {time.time() - t0:.1f}s", flush=True)
    model.set_tvg_prefix_length(prob.tvg_prefix_length)
    ddp = ref_harness.DDPish(model)
    dev = torch.device("cpu")
    T = lambda a: torch.from_numpy(np.asarray(a))
    tok = types.SimpleNamespace(pad_token_id=synth.PAD_ID)
    out = {}
    if spec.get("syn_only"):             # headline rows only: the SYN passes through the reference's loops, nothing else
        run_passes(out, "SYN_", RU, ddp, dev, prob, spec["syn"], dims, spec["syn"]["passes"], name)
        out["meta_case"] = np.array(name)
        path = os.path.join(out_dir, f"{name}.npz")
        np.savez_compressed(path, **out)
        print(f"[{name}] wrote {path} ({os.path.getsize(path) / 1e6:.2f} MB)", flush=True)
        return

    # --- padding_ids (retrieval_utils.py:155-167)
    vtg = RU.padding_ids([T(x) for x in prob.vtg_ids], [T(x) for x in prob.vtg_labels], [T(x) for x in prob.vtg_masks], tok)
    tvg = RU.padding_ids([T(x) for x in prob.tvg_ids], [T(x) for x in prob.tvg_labels], [T(x) for x in prob.tvg_masks], tok)
    for k, t in zip(("ids", "labels", "masks"), vtg):
        out[f"pad_vtg_{k}"] = t.numpy()
    for k, t in zip(("ids", "labels", "masks"), tvg):
        out[f"pad_tvg_{k}"] = t.numpy()

    video = [T(v) for v in prob.video]
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
    run_passes(out, "S_", RU, ddp, dev, prob, spec, dims, list(PASS_KINDS), name)
    # --- SYN sub-problem: BASELINE.json's headline rows (96 video + 32 text tokens, top-16)
    if "syn" in spec:
        sprob = problem_of(spec, dims, spec["syn"])
        model.set_tvg_prefix_length(sprob.tvg_prefix_length)
        run_passes(out, "SYN_", RU, ddp, dev, sprob, spec["syn"], dims, spec["syn"]["passes"], name)

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
    if name == "full7b_ref":   # long rows: the spliced embeddings are not needed by any test of this case
        out = {k: v for k, v in out.items() if not k.endswith("_embeds_sub16")}
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
    for c in (["tiny", "wide"] if a.case == "all" else a.case.split(",")):
        run_case(c, a.out)
