"""Checkpoint loader (blim_amd/checkpoint.py): HF key mapping, sharded safetensors reading, LoRA merge W' = W + (alpha/r) B A,
tvg_mlp = copy of mlp, resume-file parsing (peft naming).  CPU part uses a recording stand-in engine; the GPU part loads a
synthetic checkpoint + adapters into the real engine (adapters apart / merged) and checks scores against the oracle run on weights merged in fp32."""
import os
import types

import numpy as np
import pytest
import torch

from blim_amd import checkpoint as CK
from blim_amd import synth

SMALL = dict(vocab_size=512, hidden_size=128, intermediate_size=256, num_layers=2, num_heads=1, num_kv_heads=1, mm_hidden_size=64)


class RecordingEngine:
    def __init__(self):
        self.w = {}

    def load_weight(self, name, arr):
        self.w[name] = np.array(arr, dtype=np.float32)


def _adapters(dims, seed, names, r=8):
    rs = np.random.RandomState(seed)
    shapes = synth.weight_shapes(dims)
    out = {}
    for n in names:
        o, i = shapes[n]
        out[n] = (rs.randn(r, i).astype(np.float32) * 0.05, rs.randn(o, r).astype(np.float32) * 0.05)
    return out


def _resume_state(adapters, visual_head=None):
    """A resume file in the reference's format: {'model': {peft-named trainable tensors}} (util/misc.py:276-297)."""
    sd = {}
    for n, (A, B) in adapters.items():
        hf = CK.canonical_to_hf(n)[: -len(".weight")]
        if n.startswith("mlp.") or n.startswith("tvg_mlp."):
            p, idx, _ = n.split(".")
            hf = f"model.mm_projector.{p}.base_model.model.{idx}"
        sd[f"base_model.model.{hf}.lora_A.default.weight"] = torch.from_numpy(A)
        sd[f"base_model.model.{hf}.lora_B.default.weight"] = torch.from_numpy(B)
    if visual_head is not None:
        sd["base_model.model.visual_head.weight"] = torch.from_numpy(visual_head)
    return {"model": sd, "epoch": 3}


def test_key_mapping_roundtrip():
    dims = synth.ModelDims(**SMALL)
    for n in synth.weight_shapes(dims):
        if n.startswith("tvg_mlp."):
            assert CK.hf_to_canonical(CK.canonical_to_hf(n)) == n
            continue
        assert CK.hf_to_canonical(CK.canonical_to_hf(n)) == n
    assert CK.hf_to_canonical("model.vision_tower.vision_tower.blocks.0.attn.qkv.weight") is None
    assert CK.hf_to_canonical("model.layers.1.self_attn.q_proj.base_layer.weight") == "layers.1.q_proj.w"
    assert CK.parse_resume_key("base_model.model.model.layers.7.self_attn.k_proj.lora_B.default.weight") == ("layers.7.k_proj.w", "B")
    assert CK.parse_resume_key("base_model.model.model.mm_projector.tvg_mlp.base_model.model.2.lora_A.default.weight") == ("tvg_mlp.2.w", "A")
    assert CK.parse_resume_key("base_model.model.lm_head.lora_A.default.weight") == ("lm_head", "A")
    assert CK.parse_resume_key("base_model.model.visual_head.weight") == ("visual_head", "full")
    assert CK.parse_resume_key("base_model.model.model.vision_tower.x.lora_A.default.weight") is None


def test_load_base_and_merge_lora(tmp_path):
    dims = synth.ModelDims(**SMALL)
    w = synth.synthetic_weights(dims, 5)
    CK.save_hf_checkpoint(w, str(tmp_path / "base"), shards=3)
    names = ["layers.0.q_proj.w", "layers.1.o_proj.w", "lm_head", "mlp.0.w", "mlp.2.w", "tvg_mlp.0.w"]
    ad = _adapters(dims, 1, names)
    vh = np.random.RandomState(2).randn(*synth.weight_shapes(dims)["visual_head"]).astype(np.float32)
    torch.save(_resume_state(ad, vh), tmp_path / "resume.pth")

    rec = RecordingEngine()
    with pytest.warns(UserWarning, match="expected LoRA adapter"):       # a partial adapter file loads only when asked to
        rep = CK.load_checkpoint(rec, dims, str(tmp_path / "base"), str(tmp_path / "resume.pth"), lora_r=8, lora_alpha=32.0, strict_resume=False)
    assert set(rec.w) == set(synth.weight_shapes(dims))
    for n in synth.weight_shapes(dims):
        base = w["mlp." + n[len("tvg_mlp."):]] if n.startswith("tvg_mlp.") else w[n]      # tvg_mlp starts as a copy of mlp
        want = base
        if n in ad:
            want = base + 4.0 * (ad[n][1] @ ad[n][0])
        if n == "visual_head":
            want = vh
        np.testing.assert_allclose(rec.w[n], want, rtol=1e-6, atol=1e-6, err_msg=n)
    assert "LoRA" in rep["lm_head"] and rep["tvg_mlp.2.w"].startswith("base (copy of mlp)") and rep["visual_head"] == "resume"

    # zero-shot: no resume file -> identity merge, visual_head absent -> zeros (TVG is not evaluated in that mode)
    rec0 = RecordingEngine()
    w_no_vh = {k: v for k, v in w.items() if k != "visual_head"}
    CK.save_hf_checkpoint(w_no_vh, str(tmp_path / "base0"), shards=1)
    CK.load_checkpoint(rec0, dims, str(tmp_path / "base0"))
    assert np.count_nonzero(rec0.w["visual_head"]) == 0
    np.testing.assert_array_equal(rec0.w["tvg_mlp.0.w"], w["mlp.0.w"])
    with pytest.raises(KeyError):
        CK.load_checkpoint(RecordingEngine(), dims, str(tmp_path / "base0"), allow_missing_visual_head=False)


def test_incomplete_adapter_is_an_error(tmp_path):
    dims = synth.ModelDims(**SMALL)
    w = synth.synthetic_weights(dims, 5)
    CK.save_hf_checkpoint(w, str(tmp_path / "base"), shards=1)
    ad = _adapters(dims, 1, ["layers.0.v_proj.w"])
    st = _resume_state(ad)
    del st["model"]["base_model.model.model.layers.0.self_attn.v_proj.lora_B.default.weight"]
    torch.save(st, tmp_path / "r.pth")
    with pytest.raises(KeyError), pytest.warns(UserWarning):
        CK.load_checkpoint(RecordingEngine(), dims, str(tmp_path / "base"), str(tmp_path / "r.pth"), strict_resume=False)


def test_resume_file_is_checked_like_the_reference_checks_it(tmp_path):
    """main.py:127 asserts #parameters(resume file) == #trainable parameters; a file whose keys drift (other adapter name, extra
    wrapper prefix, missing modules) must not silently evaluate the base model."""
    dims = synth.ModelDims(**SMALL)
    w = synth.synthetic_weights(dims, 5)
    CK.save_hf_checkpoint(w, str(tmp_path / "base"), shards=1)
    vh = np.random.RandomState(2).randn(*synth.weight_shapes(dims)["visual_head"]).astype(np.float32)
    full = _adapters(dims, 1, CK.expected_adapters(dims))
    assert len(full) == 4 + 1 + 4 * dims.num_layers
    torch.save(_resume_state(full, vh), tmp_path / "ok.pth")
    rep = CK.load_checkpoint(RecordingEngine(), dims, str(tmp_path / "base"), str(tmp_path / "ok.pth"))
    assert sum("+ LoRA" in p for p in rep.values()) == len(full) and "LoRA" in CK.summarize_report(rep)
    # (a) an adapter missing
    part = dict(full); del part["layers.1.k_proj.w"]
    torch.save(_resume_state(part, vh), tmp_path / "a.pth")
    with pytest.raises(ValueError, match="expected LoRA adapter"):
        CK.load_checkpoint(RecordingEngine(), dims, str(tmp_path / "base"), str(tmp_path / "a.pth"))
    # (b) naming drift: an extra wrapper level in front of every key -> nothing would be applied
    st = _resume_state(full, vh)
    st["model"] = {"module.wrapped." + k: v for k, v in st["model"].items()}
    torch.save(st, tmp_path / "b.pth")
    with pytest.raises(ValueError, match="map onto no engine tensor"):
        CK.load_checkpoint(RecordingEngine(), dims, str(tmp_path / "base"), str(tmp_path / "b.pth"))
    # (c) visual_head missing
    torch.save(_resume_state(full), tmp_path / "c.pth")
    with pytest.raises(ValueError, match="visual_head absent"):
        CK.load_checkpoint(RecordingEngine(), dims, str(tmp_path / "base"), str(tmp_path / "c.pth"))
    # (d) wrong rank -> parameter total differs from the trainable total
    st = _resume_state(_adapters(dims, 1, CK.expected_adapters(dims), r=4), vh)
    torch.save(st, tmp_path / "d.pth")
    with pytest.raises(ValueError, match="trainable parameters expected"):
        CK.load_checkpoint(RecordingEngine(), dims, str(tmp_path / "base"), str(tmp_path / "d.pth"))


@pytest.mark.gpu
@pytest.mark.parametrize("lora_mode", ["apart", "merge"])
def test_engine_scores_with_adapted_checkpoint(tmp_path, lora_mode):
    """Base checkpoint + resume file -> engine (adapters kept apart, the default, or merged on the host) against the numpy oracle on W + (alpha / r) B A merged in
    FLOAT32 -- what the reference's adapters-apart forward equals in exact arithmetic (pinned by tests/golden/lora_tiny.npz; the 28-layer and 7B-size cases
    against the reference's own run are tests/test_lora_gpu.py).  No rounding on the oracle's side: the reference never merges."""
    from blim_amd import retrieval_utils as RU
    from blim_amd.modeling import BlimModel, DDPLike
    from oracle import blim_oracle as O
    d = dict(vocab_size=151700, hidden_size=256, intermediate_size=512, num_layers=2, num_heads=2, num_kv_heads=1, mm_hidden_size=64)
    dims = synth.ModelDims(**d)
    w = synth.synthetic_weights(dims, 7)
    CK.save_hf_checkpoint(w, str(tmp_path / "base"), shards=2)
    names = [f"layers.{i}.{p}.w" for i in range(2) for p in ("q_proj", "k_proj", "v_proj", "o_proj")] + ["lm_head", "mlp.0.w", "mlp.2.w", "tvg_mlp.0.w", "tvg_mlp.2.w"]
    ad = _adapters(dims, 3, names)
    torch.save(_resume_state(ad, w["visual_head"]), tmp_path / "resume.pth")
    model = BlimModel(dims, max_positions=512)
    CK.load_checkpoint(model.engine, dims, str(tmp_path / "base"), str(tmp_path / "resume.pth"), lora_mode=lora_mode)
    assert model.engine.weights_ready() and model.engine.num_adapters() == (len(names) if lora_mode == "apart" else 0)
    merged = dict(w)
    for n in ("0.w", "0.b", "2.w", "2.b"):
        merged["tvg_mlp." + n] = w["mlp." + n].copy()
    for n, (A, B) in ad.items():
        merged[n] = (merged[n] + np.float32(4.0) * (B @ A)).astype(np.float32)
    prob = synth.make_problem(4, 4, dims, tok_per_clip=8, text_len=(3, 8))
    model.set_tvg_prefix_length(prob.tvg_prefix_length)
    tok = types.SimpleNamespace(pad_token_id=synth.PAD_ID)
    Tt = lambda rows: [torch.from_numpy(r) for r in rows]
    vtg = RU.padding_ids(Tt(prob.vtg_ids), Tt(prob.vtg_labels), Tt(prob.vtg_masks), tok)
    tvg = RU.padding_ids(Tt(prob.tvg_ids), Tt(prob.tvg_labels), Tt(prob.tvg_masks), tok)
    sc = RU.PairScorer(DDPLike(model), vtg[0], vtg[2], vtg[1], tvg[0], tvg[2], tvg[1], [torch.from_numpy(v) for v in prob.video],
                       torch.from_numpy(prob.video_vocab), torch.from_numpy(prob.tvg_video_labels), dims.num_clips)
    pairs = np.array([[0, 0], [1, 2], [3, 1]])
    got_v, got_t = sc.vtg(pairs), sc.tvg(pairs)
    om = O.OracleModel(O.OracleConfig(**d), merged); om.set_tvg_prefix_length(prob.tvg_prefix_length)
    ov = O.padding_ids(prob.vtg_ids, prob.vtg_labels, prob.vtg_masks, synth.PAD_ID)
    ot = O.padding_ids(prob.tvg_ids, prob.tvg_labels, prob.tvg_masks, synth.PAD_ID)
    for k, (j, i) in enumerate(pairs):
        mask, _, emb, lab = om.prepare_inputs_labels_for_multimodal(ov[0][[i]], ov[2][[i]], ov[1][[i]], [prob.video[j]])
        np.testing.assert_allclose(got_v[k], om.label_logprobs(om.forward_hidden(emb, mask), lab)[0], rtol=1e-3)
        mask, _, emb, lab = om.prepare_inputs_labels_for_multimodal(ot[0][[i]], ot[2][[i]], ot[1][[i]], [prob.video[j]], tvg=True)
        want = O._tvg_scores(om, om.forward_hidden(emb, mask), lab, prob.video_vocab, np.full((1, dims.num_clips), prob.tvg_video_labels[j]), dims.num_clips)[0]
        np.testing.assert_allclose(got_t[k], want, rtol=1e-3)
    model.engine.close()
