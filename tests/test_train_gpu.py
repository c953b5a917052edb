"""HIP trainer (include/blim.h blim_train_*) against the fixtures recorded from the reference's model under autograd
(tests/golden/train_*.npz, oracle/gen_golden_train.py) -- SURVEY.md section 8f-4.  Calls go through the C ABI."""
import math
import os

import numpy as np
import pytest

from blim_amd import lora, synth

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(__file__), "golden")

# 16-bit operands everywhere (the reference itself trains under autocast(float16)): a gradient tensor is compared relative to its own
# largest element / its L2 norm against the fp32 autograd reference
GRAD_RTOL = {"f16": 1e-2, "bf16": 1e-1}        # worst element; measured 1.4e-3 / 9.0e-3 (tiny), 2.1e-3 (7B width), ~5e-3 at 28 layers.  Since round 3 every split
                                               # reduction is summed in a fixed order (no float atomics): the numbers no longer move from run to run
GRAD_NORM_RTOL = {"f16": 3e-3, "bf16": 3e-2}   # the tensor's L2 norm; measured <= 1.3e-3 (28 layers); bf16 (a reported, not a parity mode): 5.6e-3 tiny,
                                               # 5.9e-2 worst element / 1.6e-2 norm through 28 layers (8-bit mantissas in P and the activation gradients)
LOSS_RTOL = {"f16": 1e-3, "bf16": 1e-2}


def _case(name):
    from oracle.gen_golden_train import CASES, adapter_values
    spec = CASES[name]
    dims = synth.ModelDims(**spec["dims"])
    weights = synth.synthetic_weights(dims, spec["wseed"])
    prob = synth.make_problem(spec["pseed"], spec["n"], dims, tok_per_clip=spec["tok_per_clip"], text_len=spec["text_len"])
    tr = adapter_values(dims, spec["r"], spec["aseed"])
    tr["visual_head"] = weights["visual_head"].copy()
    return spec, dims, weights, prob, {n: tr[n] for n in lora.trainable_names(dims)}


def collate(prob, sel):
    """dataloader/base_dataset.py:132-153 (train split): left padding to the batch maximum."""
    import torch

    def pad(rows, fill):
        L = max(len(r) for r in rows)
        out = np.full((len(rows), L), fill, np.int64)
        for i, r in enumerate(rows):
            out[i, L - len(r):] = r
        return torch.from_numpy(out)
    sel = list(sel)
    return {"video": [torch.from_numpy(prob.video[i]) for i in sel],
            "vtg_ids": pad([prob.vtg_ids[i] for i in sel], synth.PAD_ID), "vtg_labels": pad([prob.vtg_labels[i] for i in sel], -100),
            "vtg_masks": pad([prob.vtg_masks[i] for i in sel], 0),
            "tvg_ids": pad([prob.tvg_ids[i] for i in sel], synth.PAD_ID), "tvg_labels": pad([prob.tvg_labels[i] for i in sel], -100),
            "tvg_masks": pad([prob.tvg_masks[i] for i in sel], 0),
            "tvg_video_labels": torch.from_numpy(prob.tvg_video_labels[sel])}


def _sample(a, spec=None):
    from oracle.gen_golden_train import MAX_STORE, sample_rows
    return sample_rows(a, (spec or {}).get("max_store", MAX_STORE))


@pytest.mark.parametrize("case,dtype", [("train_tiny", "f16"), ("train_tiny", "bf16"), ("train_wide", "f16"), ("train_deep", "f16"), ("train_deep", "bf16")])
def test_training_step_matches_reference_autograd(case, dtype):
    import torch
    from blim_amd.engine import Engine
    from blim_amd.training import Trainer
    g = np.load(os.path.join(GOLDEN, f"{case}.npz"))
    spec, dims, weights, prob, tr = _case(case)
    eng = Engine(dims, max_positions=1024, dtype=dtype)
    eng.load_weights(weights)
    t = Trainer(eng, lora_r=spec["r"], lora_alpha=spec["alpha"], lora_dropout=0.0, weight_decay=spec["wd"], trainable=tr)
    t.set_video_vocab(torch.from_numpy(prob.video_vocab))
    worst = {}
    for step, sel in enumerate(spec["batches"]):
        t.zero_grad()
        lv, lt = t.forward_backward(collate(prob, sel))
        rv, rt = float(g[f"loss_vtg_{step}"]), float(g[f"loss_tvg_{step}"])
        print(f"[{case}/{dtype}] step {step}: vtg {lv:.5f} (ref {rv:.5f})  tvg {lt:.5f} (ref {rt:.5f})")
        # step 1 runs on the parameters the first AdamW step produced; at lr 1e-2 that step is violent (the 7B-width TVG loss jumps from 1.2
        # to 4.7), so the second loss is a sanity check -- the parameters themselves are compared below
        tol = LOSS_RTOL[dtype] if step == 0 else 3e-2        # (elements with ~zero gradient take +-lr steps on rounding noise: measured up to 1.1e-2)
        assert abs(lv - rv) <= tol * abs(rv) and abs(lt - rt) <= tol * abs(rt)
        grads = t.state("grads")
        inv = 1.0 / t.scaler.scale
        for n in lora.trainable_names(dims):
            gr = grads[n] * inv
            ref_norm = float(g[f"gnorm_{step}/{n}"])
            if step == 0:
                ref = g[f"grad/{n}"]
                err = float(np.abs(_sample(gr, spec) - ref).max() / max(np.abs(ref).max(), 1e-30))
                nerr = abs(float(np.linalg.norm(gr.astype(np.float64))) - ref_norm) / max(ref_norm, 1e-30)
                worst[n] = (err, nerr)
        if step == 0:
            bad = {n: v for n, v in worst.items() if v[0] > GRAD_RTOL[dtype] or v[1] > GRAD_NORM_RTOL[dtype]}
            top = sorted(worst.items(), key=lambda kv: -kv[1][0])[:6]
            print(f"[{case}/{dtype}] worst gradient deviations (max-rel, norm-rel): " + ", ".join(f"{n} {a:.2e}/{b:.2e}" for n, (a, b) in top))
            assert not bad, bad
        st = t.optimizer_step(spec["lr"])
        assert st["skipped"] == 0.0
    # parameters after two AdamW steps: Adam normalises the update, so an element whose gradient is ~0 moves by O(lr) on noise
    params = t.state("params")
    for n in lora.trainable_names(dims):
        ref = g[f"param/{n}"]
        d = np.abs(_sample(params[n], spec) - ref)
        med, mx = (0.05, 5.0) if dtype == "f16" else (0.2, 5.0)     # max: an element with ~zero gradient can take opposite-sign steps twice
        assert np.median(d) <= med * spec["lr"] and d.max() <= mx * spec["lr"], (n, float(np.median(d)), float(d.max()))
    t.close(); eng.close()


@pytest.mark.parametrize("case,n_rows", [("train_tiny", 40), ("train_deep", 0)])
def test_gradients_are_reproducible_bit_for_bit(case, n_rows):
    """The reference's autograd is deterministic for these shapes (training_utils.py:81-91); so is the trainer: the time / column splits of
    the adapter-gradient, du and loss reductions meet through partials summed in a fixed order (csrc/train_kernels.hip: ordered_sum_kernel),
    not float atomics.  Two forward+backward passes over the same batch (dropout ON, same step seed) give identical bits -- on a batch long
    enough for several 1,024-token splits (40 samples of the tiny configuration) and through all 28 layers (train_deep)."""
    import torch
    from blim_amd.engine import Engine
    from blim_amd.training import Trainer
    spec, dims, weights, prob, tr = _case(case)
    if n_rows:
        prob = synth.make_problem(77, n_rows, dims, tok_per_clip=16, text_len=(10, 40))
    sel = list(range(n_rows)) if n_rows else list(spec["batches"][0])
    eng = Engine(dims, max_positions=1024, dtype="f16")
    eng.load_weights(weights)
    rng = np.random.default_rng(5)
    tr = {n: (v if not n.endswith(":B") else (0.02 * rng.standard_normal(v.shape)).astype(np.float32)) for n, v in tr.items()}    # B != 0: dA is not trivially zero
    t = Trainer(eng, lora_r=spec["r"], lora_alpha=spec["alpha"], lora_dropout=0.1, weight_decay=spec["wd"], trainable=tr, seed=3)
    t.set_video_vocab(torch.from_numpy(prob.video_vocab))
    batch = collate(prob, sel)
    runs = []
    for _ in range(3):
        t.zero_grad()
        lv, lt = t.forward_backward(batch, seed=11)         # same dropout seed for every repetition
        runs.append((np.float32(lv), np.float32(lt), {n: a.copy() for n, a in t.state("grads").items()}))
    n_tok = sum(int(np.count_nonzero(m)) for m in batch["vtg_masks"].numpy()) + sum(v.shape[0] * v.shape[1] for v in batch["video"])
    assert not n_rows or n_tok > 2048                       # several time splits
    for lv, lt, gr in runs[1:]:
        assert lv.tobytes() == runs[0][0].tobytes() and lt.tobytes() == runs[0][1].tobytes()
        for n in gr:
            assert np.array_equal(gr[n].view(np.uint32), runs[0][2][n].view(np.uint32)), n
    assert any(np.abs(a).max() > 0 for n, a in runs[0][2].items() if n.endswith(":A"))
    t.close(); eng.close()


def test_flat_layout_matches_library():
    import ctypes as C
    from blim_amd.engine import Engine
    from blim_amd.training import _lib
    dims = synth.ModelDims(vocab_size=151700, hidden_size=256, intermediate_size=512, num_layers=2, num_heads=2, num_kv_heads=1, mm_hidden_size=64)
    eng = Engine(dims, max_positions=256, dtype="f16")
    lib = _lib()
    lay, total = lora.flat_layout(dims, 8)
    assert lib.blim_train_flat_size(eng.h, 8) == total
    for n, (off, shape) in lay.items():
        o, r, c = C.c_int64(), C.c_int64(), C.c_int64()
        assert lib.blim_train_param_offset(eng.h, 8, n.encode(), C.byref(o), C.byref(r), C.byref(c)) == 0, n
        assert (o.value, (r.value, c.value)) == (off, tuple(shape)), n
    eng.close()


def test_merge_equals_adapter_forward():
    """blim_train_merge: the scoring engine with merged weights gives the score the adapter forward's loss implies."""
    import torch
    from blim_amd.engine import Engine
    from blim_amd.training import Trainer
    from blim_amd.modeling import BlimModel
    from blim_amd import retrieval_utils as RU
    spec, dims, weights, prob, tr = _case("train_tiny")
    model = BlimModel(dims, max_positions=1024, dtype="f16")
    eng = model.engine
    eng.load_weights(weights)
    t = Trainer(eng, lora_r=spec["r"], lora_alpha=spec["alpha"], lora_dropout=0.0, trainable=tr)
    t.set_video_vocab(torch.from_numpy(prob.video_vocab))
    lv, lt = t.forward_backward(collate(prob, [0, 1, 2]))
    t.merge_into_engine()
    # VTG loss of the batch = -(sum of per-row score * row tokens) / tokens, with the merged engine's literal forward
    model.set_tvg_prefix_length(prob.tvg_prefix_length)
    data = collate(prob, [0, 1, 2])
    dev = eng.device
    (_, _, (m, _), _, emb, lab) = model.prepare_inputs_labels_for_multimodal(data["vtg_ids"].to(dev), None, data["vtg_masks"].to(dev), None, data["vtg_labels"].to(dev),
                                                                             [v.to(dev) for v in data["video"]], ["video"] * 3, image_sizes=None, video_feature=True, cpn=True)
    out = model(inputs_embeds=emb, attention_mask=m)
    score = RU.vtg_criterion(out.logits, lab).float().cpu().numpy()
    ntok = (lab[:, 1:] != -100).sum(1).cpu().numpy()
    merged_loss = float(-(score * ntok).sum() / ntok.sum())
    assert abs(merged_loss - lv) <= 2e-3 * abs(lv), (merged_loss, lv)
    t.close(); eng.close()


def test_merged_and_apart_adapters_cannot_be_stacked():
    """The two ways a fine-tuned model reaches the scoring engine -- blim_train_merge (W + alpha/r B A written into the base weights) and blim_load_adapter (adapters apart) --
    exclude each other on one engine: both together would apply the update twice, silently.  Either order fails with BLIM_ERR_STATE; loading base weights again lifts the mark."""
    from blim_amd.engine import Engine, BlimError
    from blim_amd.training import Trainer
    spec, dims, weights, prob, tr = _case("train_tiny")
    eng = Engine(dims, max_positions=1024, dtype="f16")
    eng.load_weights(weights)
    t = Trainer(eng, lora_r=spec["r"], lora_alpha=spec["alpha"], lora_dropout=0.0, trainable=tr)
    try:
        t.adapters_into_engine()
        assert eng.num_adapters() > 0
        with pytest.raises(BlimError, match="twice"):
            t.merge_into_engine()
        eng.clear_adapters()
        t.merge_into_engine()
        with pytest.raises(BlimError, match="twice"):
            t.adapters_into_engine()
        assert eng.num_adapters() == 0
        t.merge_into_engine()                                  # repeated merges stay fine (always from the pristine base)
        eng.load_weights(weights)                              # base weights again: adapters apart are accepted
        t.adapters_into_engine()
        assert eng.num_adapters() > 0
    finally:
        t.close(); eng.close()


def test_lora_dropout_mask_is_consistent_forward_and_backward():
    """lora_drop > 0 (main.py --lora_drop 0.05): the oracle applies the engine's counter-based mask (restated in oracle/train_oracle.py:
    drop_mult) to the adapters' inputs; losses and gradients must agree, i.e. forward and backward use the same mask at every site."""
    import torch
    from blim_amd.engine import Engine
    from blim_amd.training import Trainer
    from oracle.blim_oracle import OracleConfig
    from oracle.train_oracle import TrainOracle
    spec, dims, weights, prob, tr = _case("train_tiny")
    p, seed, sel = 0.25, 4242, [0, 2, 3]
    eng = Engine(dims, max_positions=1024, dtype="f16")
    eng.load_weights(weights)
    t = Trainer(eng, lora_r=spec["r"], lora_alpha=spec["alpha"], lora_dropout=p, trainable=tr)
    t.set_video_vocab(torch.from_numpy(prob.video_vocab))
    lv, lt = t.forward_backward(collate(prob, sel), seed=seed)
    grads = t.state("grads")
    orc = TrainOracle(OracleConfig(**spec["dims"]), weights, tr, spec["r"], spec["alpha"], drop_p=p)
    orc.seed = seed
    ov, ot, og = orc.step_grads([prob.vtg_ids[i] for i in sel], [prob.vtg_labels[i] for i in sel], [prob.tvg_ids[i] for i in sel], [prob.tvg_labels[i] for i in sel],
                                [prob.video[i] for i in sel], prob.video_vocab, prob.tvg_video_labels[sel])
    print(f"[dropout] vtg {lv:.5f} (oracle {ov:.5f})  tvg {lt:.5f} (oracle {ot:.5f})")
    assert abs(lv - ov) <= 1e-3 * abs(ov) and abs(lt - ot) <= 1e-3 * abs(ot)
    worst = {}
    for n in lora.trainable_names(dims):
        gr = grads[n] / t.scaler.scale
        worst[n] = float(np.abs(gr - og[n]).max() / max(np.abs(og[n]).max(), 1e-30))
    print("[dropout] worst gradient deviations: " + ", ".join(f"{n} {v:.2e}" for n, v in sorted(worst.items(), key=lambda kv: -kv[1])[:5]))
    assert max(worst.values()) <= GRAD_RTOL["f16"], worst
    # and the mask matters: the same batch without dropout gives a different loss
    t0 = Trainer(eng, lora_r=spec["r"], lora_alpha=spec["alpha"], lora_dropout=0.0, trainable=tr)
    t0.set_video_vocab(torch.from_numpy(prob.video_vocab))
    lv0, _ = t0.forward_backward(collate(prob, sel), seed=seed)
    assert abs(lv0 - lv) > 1e-4
    t0.close(); t.close(); eng.close()


@pytest.mark.parametrize("r", [4, 16])
def test_other_ranks_match_the_oracle(r):
    """--lora_r other than 8 (the kernels have 8- and 16-wide register variants; r = 4 runs in the 8-wide one): against oracle/train_oracle.py."""
    import torch
    from blim_amd.engine import Engine
    from blim_amd.training import Trainer
    from oracle.blim_oracle import OracleConfig
    from oracle.gen_golden_train import adapter_values
    from oracle.train_oracle import TrainOracle
    spec, dims, weights, prob, _ = _case("train_tiny")
    tr = adapter_values(dims, r, 77)
    tr["visual_head"] = weights["visual_head"].copy()
    tr = {n: tr[n] for n in lora.trainable_names(dims)}
    sel = [0, 1, 4]
    eng = Engine(dims, max_positions=1024, dtype="f16")
    eng.load_weights(weights)
    t = Trainer(eng, lora_r=r, lora_alpha=16.0, lora_dropout=0.0, trainable=tr)
    t.set_video_vocab(torch.from_numpy(prob.video_vocab))
    lv, lt = t.forward_backward(collate(prob, sel))
    grads = t.state("grads")
    orc = TrainOracle(OracleConfig(**spec["dims"]), weights, tr, r, 16.0)
    ov, ot, og = orc.step_grads([prob.vtg_ids[i] for i in sel], [prob.vtg_labels[i] for i in sel], [prob.tvg_ids[i] for i in sel], [prob.tvg_labels[i] for i in sel],
                                [prob.video[i] for i in sel], prob.video_vocab, prob.tvg_video_labels[sel])
    assert abs(lv - ov) <= 1e-3 * abs(ov) and abs(lt - ot) <= 1e-3 * abs(ot)
    worst = max(float(np.abs(grads[n] / t.scaler.scale - og[n]).max() / max(np.abs(og[n]).max(), 1e-30)) for n in lora.trainable_names(dims))
    print(f"[r={r}] worst gradient deviation {worst:.2e}")
    assert worst <= GRAD_RTOL["f16"]
    t.close(); eng.close()


def test_gradient_accumulation_equals_the_mean_of_micro_batches():
    """--accum_iter 2 (training_utils.py:87-91): two micro-batches accumulated with loss / 2 give the mean of their separate gradients."""
    import torch
    from blim_amd.engine import Engine
    from blim_amd.training import Trainer
    spec, dims, weights, prob, tr = _case("train_tiny")
    eng = Engine(dims, max_positions=1024, dtype="f16")
    eng.load_weights(weights)
    t = Trainer(eng, lora_r=spec["r"], lora_alpha=spec["alpha"], lora_dropout=0.0, trainable=tr)
    t.set_video_vocab(torch.from_numpy(prob.video_vocab))
    a, b = collate(prob, [0, 1]), collate(prob, [2, 3, 4])
    sep = []
    for d in (a, b):
        t.zero_grad(); t.forward_backward(d); sep.append({n: g.copy() for n, g in t.state("grads").items()})
    t.zero_grad()
    t.forward_backward(a, accum_iter=2); t.forward_backward(b, accum_iter=2)
    acc = t.state("grads")
    for n in lora.trainable_names(dims):
        want = 0.5 * (sep[0][n] + sep[1][n])
        assert np.abs(acc[n] - want).max() <= 2e-3 * max(np.abs(want).max(), 1e-30), n      # 16-bit gradients at half the scale: rounding only
    st = t.optimizer_step(1e-3)
    assert st["skipped"] == 0.0 and np.isfinite(st["grad_norm"])
    t.close(); eng.close()


def test_full_size_training_properties_7b():
    """The real 7B configuration (28 layers, reference-shaped rows) through size-independent properties: with peft's initialisation
    (B = 0) the adapter forward is the base model, so (1) the VTG loss equals what the SCORING path gives for the same rows, (2) every
    dA is exactly zero while every dB is finite and non-zero; (3) AdamW steps on the batch lower both losses; (4) after the merge the
    scoring path reproduces the trainer's loss."""
    import torch
    from blim_amd import retrieval_utils as RU
    from blim_amd.modeling import BlimModel
    from blim_amd.training import Trainer
    dims = synth.ModelDims()
    model = BlimModel(dims, max_positions=1024, dtype="f16")
    eng = model.engine
    eng.init_synthetic_weights(0)
    prob = synth.make_problem(31, 4, dims, tok_per_clip=64, text_len=(8, 24), fast_video=True)
    model.set_tvg_prefix_length(prob.tvg_prefix_length)
    data = collate(prob, [0, 1, 2, 3])
    dev = eng.device

    def scoring_loss():
        model.clear_cache()
        (_, _, (m, _), _, emb, lab) = model.prepare_inputs_labels_for_multimodal(data["vtg_ids"].to(dev), None, data["vtg_masks"].to(dev), None, data["vtg_labels"].to(dev),
                                                                                 [v.to(dev) for v in data["video"]], ["video"] * 4, image_sizes=None, video_feature=True, cpn=True)
        score = RU.vtg_criterion(model(inputs_embeds=emb, attention_mask=m).logits, lab).float().cpu().numpy()
        ntok = (lab[:, 1:] != -100).sum(1).cpu().numpy()
        return float(-(score * ntok).sum() / ntok.sum())

    base = scoring_loss()
    t = Trainer(eng, lora_r=8, lora_alpha=32.0, lora_dropout=0.0, seed=5)                  # visual_head ~ N(0, 0.02): no head in the "checkpoint"
    t.set_video_vocab(torch.from_numpy(prob.video_vocab))
    lv0, lt0 = t.forward_backward(data)
    print(f"[7B] vtg loss {lv0:.5f} (scoring path {base:.5f}), tvg loss {lt0:.5f}")
    assert abs(lv0 - base) <= 1e-3 * abs(base)
    g = t.state("grads")
    for n, a in g.items():
        assert np.isfinite(a).all(), n
        if n.endswith(":A"):
            assert not a.any(), n                                   # dA = du^T x with du = s * dy . B = 0
        elif n.endswith(":B") and not n.startswith("tvg_mlp") and not n.startswith("mlp"):
            assert np.abs(a).max() > 0, n
    for _ in range(3):
        st = t.optimizer_step(2e-4)
        assert st["skipped"] == 0.0
        t.zero_grad()
        lv1, lt1 = t.forward_backward(data)
    print(f"[7B] after 3 AdamW steps: vtg {lv1:.5f}, tvg {lt1:.5f}")
    assert lv1 < lv0 and lt1 < lt0
    t.merge_into_engine()
    merged = scoring_loss()
    assert abs(merged - lv1) <= 2e-3 * abs(lv1), (merged, lv1)
    t.close(); eng.close()


@pytest.mark.parametrize("heads,kv,layers,tok,text", [(4, 2, 2, 16, (20, 70)), (4, 1, 1, 24, (1, 40)), (8, 8, 1, 8, (30, 100))])
def test_training_shapes_against_the_oracle(heads, kv, layers, tok, text):
    """Other head groupings (GQA 2:1, 4:1, MHA) and rows whose lengths straddle the 32 / 64 / 128-token tile edges of the attention
    kernels (32 - 96 video tokens + 1 - 100 caption tokens), ragged inside one batch: losses and every gradient against
    oracle/train_oracle.py (itself pinned by the reference's autograd fixtures)."""
    import torch
    from blim_amd.engine import Engine
    from blim_amd.training import Trainer
    from oracle.blim_oracle import OracleConfig
    from oracle.gen_golden_train import adapter_values
    from oracle.train_oracle import TrainOracle
    d = dict(vocab_size=151700, hidden_size=128 * heads, intermediate_size=256 * heads, num_layers=layers, num_heads=heads, num_kv_heads=kv, mm_hidden_size=64)
    dims = synth.ModelDims(**d)
    weights = synth.synthetic_weights(dims, 40 + heads)
    prob = synth.make_problem(50 + kv, 5, dims, tok_per_clip=tok, text_len=text)
    tr = adapter_values(dims, 8, 60 + layers)
    tr["visual_head"] = weights["visual_head"].copy()
    tr = {n: tr[n] for n in lora.trainable_names(dims)}
    sel = [0, 1, 2, 3, 4]
    eng = Engine(dims, max_positions=1024, dtype="f16")
    eng.load_weights(weights)
    t = Trainer(eng, lora_r=8, lora_alpha=32.0, lora_dropout=0.0, trainable=tr)
    t.set_video_vocab(torch.from_numpy(prob.video_vocab))
    lv, lt = t.forward_backward(collate(prob, sel))
    grads = t.state("grads")
    orc = TrainOracle(OracleConfig(**d), weights, tr, 8, 32.0)
    ov, ot, og = orc.step_grads([prob.vtg_ids[i] for i in sel], [prob.vtg_labels[i] for i in sel], [prob.tvg_ids[i] for i in sel], [prob.tvg_labels[i] for i in sel],
                                [prob.video[i] for i in sel], prob.video_vocab, prob.tvg_video_labels[sel])
    lens = sorted(len(prob.vtg_ids[i]) + dims.num_clips * tok - 1 for i in sel)
    worst = max(float(np.abs(grads[n] / t.scaler.scale - og[n]).max() / max(np.abs(og[n]).max(), 1e-30)) for n in lora.trainable_names(dims))
    print(f"[heads {heads}/{kv}, {layers} layer(s)] VTG row lengths {lens}: vtg {lv:.5f} ({ov:.5f}) tvg {lt:.5f} ({ot:.5f}), worst gradient deviation {worst:.2e}")
    assert abs(lv - ov) <= 1e-3 * abs(ov) and abs(lt - ot) <= 1e-3 * abs(ot)
    assert worst <= GRAD_RTOL["f16"]
    t.close(); eng.close()


def test_an_fp16_overflow_in_the_training_forward_skips_the_step():
    """the scoring path SATURATES its fp16 activation stores (finite scores where the reference's `.half()` forward returns NaN), but the trainer must
    not: under the reference's autocast + GradScaler an activation beyond 65504 becomes inf, reaches the loss and the gradients, and the scaler skips the optimizer
    step and halves the scale (util/misc.py:232-259).  `saturation.npz`'s weights (SwiGLU product ~7e6) make the training forward overflow: the step is reported as
    skipped, no parameter moves, the loss scale is halved; the same batch on a bf16 engine (f32's range) steps normally."""
    import torch
    from blim_amd.modeling import BlimModel
    from blim_amd.training import Trainer
    from oracle.gen_golden_saturation import DIMS, N, PSEED, TEXT, TOK, scaled_weights
    dims = synth.ModelDims(**DIMS)
    w = scaled_weights(dims)
    prob = synth.make_problem(PSEED, N, dims, tok_per_clip=TOK, text_len=TEXT)
    out = {}
    for dtype in ("f16", "bf16"):
        model = BlimModel(dims, max_positions=512, dtype=dtype)
        model.engine.load_weights(w)
        tr = Trainer(model.engine, lora_r=8, lora_alpha=32.0, lora_dropout=0.0, seed=3)
        try:
            tr.set_video_vocab(torch.from_numpy(prob.video_vocab))
            before = tr.params.clone()
            scale0 = tr.scaler.scale
            tr.zero_grad()
            losses = tr.forward_backward(collate(prob, range(min(N, 4))))
            st = tr.optimizer_step(1e-3)
            out[dtype] = (losses, st, bool(torch.equal(tr.params, before)), tr.scaler.scale / scale0)
        finally:
            tr.close(); model.engine.close()
    (l16, s16, same16, r16), (lb, sb, sameb, rb) = out["f16"], out["bf16"]
    assert s16["skipped"] == 1.0 and same16 and r16 == 0.5 and not all(math.isfinite(x) for x in l16), out["f16"]
    assert sb["skipped"] == 0.0 and not sameb and all(math.isfinite(x) for x in lb) and math.isfinite(sb["grad_norm"]), out["bf16"]
