"""The training oracle (oracle/train_oracle.py) against the fixtures recorded from the reference's model under autograd
(oracle/gen_golden_train.py) -- SURVEY.md section 8f-4."""
import math
import os

import numpy as np
import pytest

from blim_amd import lora, synth
from oracle.blim_oracle import OracleConfig
from oracle.gen_golden_train import CASES, MAX_STORE, adapter_values, sample_rows as _sample_rows
from oracle.train_oracle import AdamW, TrainOracle, cosine_lr

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def batch_of(prob, sel):
    sel = list(sel)
    return ([prob.vtg_ids[i] for i in sel], [prob.vtg_labels[i] for i in sel], [prob.tvg_ids[i] for i in sel], [prob.tvg_labels[i] for i in sel],
            [prob.video[i] for i in sel], prob.video_vocab, prob.tvg_video_labels[sel])


def build(case):
    spec = CASES[case]
    dims = synth.ModelDims(**spec["dims"])
    weights = synth.synthetic_weights(dims, spec["wseed"])
    prob = synth.make_problem(spec["pseed"], spec["n"], dims, tok_per_clip=spec["tok_per_clip"], text_len=spec["text_len"])
    tr = adapter_values(dims, spec["r"], spec["aseed"])
    tr["visual_head"] = weights["visual_head"].copy()
    tr = {n: tr[n] for n in lora.trainable_names(dims)}
    return spec, dims, weights, prob, tr


@pytest.mark.parametrize("case", ["train_tiny", "train_deep"])
def test_oracle_matches_reference_autograd(case):
    g = np.load(os.path.join(GOLDEN, f"{case}.npz"))
    spec, dims, weights, prob, tr = build(case)
    sample_rows = lambda a: _sample_rows(a, spec.get("max_store", MAX_STORE))
    orc = TrainOracle(OracleConfig(**spec["dims"]), weights, tr, spec["r"], spec["alpha"])
    params = {k: v.detach().numpy() for k, v in orc.p.items()}          # views: the optimizer updates the oracle's tensors in place
    opt = AdamW(params, spec["lr"], spec["wd"])
    for step, sel in enumerate(spec["batches"]):
        lv, lt, grads = orc.step_grads(*batch_of(prob, sel))
        assert abs(lv - float(g[f"loss_vtg_{step}"])) <= 2e-5 * abs(lv), (step, lv)
        assert abs(lt - float(g[f"loss_tvg_{step}"])) <= 2e-5 * abs(lt), (step, lt)
        for n, gr in grads.items():
            ref_norm = float(g[f"gnorm_{step}/{n}"])
            assert abs(np.linalg.norm(gr.astype(np.float64)) - ref_norm) <= 1e-4 * ref_norm + 1e-9, (step, n)
            if step == 0:
                ref = g[f"grad/{n}"]
                assert np.abs(sample_rows(gr) - ref).max() <= 2e-5 * np.abs(ref).max() + 1e-9, n
        with __import__("torch").no_grad():
            opt.step(params, grads)
    for n, p in params.items():
        ref = g[f"param/{n}"]
        # Adam normalises the update: an element whose gradient is ~0 moves by O(lr) on a relative gradient difference of 1e-5
        d = np.abs(sample_rows(p) - ref)
        assert np.median(d) <= 1e-3 * spec["lr"] + 2e-6 * np.abs(ref).max() and d.max() <= 3.0 * spec["lr"], (n, float(np.median(d)), float(d.max()))
        assert abs(np.linalg.norm(p.astype(np.float64)) - float(g[f"pnorm/{n}"])) <= 1e-5 * float(g[f"pnorm/{n}"]) + 1e-9


def test_cosine_schedule():
    assert cosine_lr(0.0, 1e-3, 0.0, 2, 10) == 0.0
    assert math.isclose(cosine_lr(1.0, 1e-3, 0.0, 2, 10), 5e-4)
    assert math.isclose(cosine_lr(2.0, 1e-3, 0.0, 2, 10), 1e-3)
    assert math.isclose(cosine_lr(6.0, 1e-3, 1e-5, 2, 10), 1e-5 + (1e-3 - 1e-5) * 0.5)


def test_resume_keys_round_trip():
    from blim_amd.checkpoint import parse_resume_key
    dims = synth.ModelDims(**CASES["train_tiny"]["dims"])
    for n in lora.trainable_names(dims):
        key = lora.resume_key(n)
        want = (n.split(":")[0], n.split(":")[1]) if ":" in n else ("visual_head", "full")
        assert parse_resume_key(key) == want, (n, key)
