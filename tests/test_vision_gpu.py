"""GPU (-m gpu): offline feature extraction on the engine (blim_amd/vision.py -> blim_vision_* -> csrc/vision.hip) against the golden
vectors recorded from the reference's own vision tower + ToMe, and against the numpy oracle.

ToMe makes discrete choices (arg-max / arg-sort of cosine similarities); it runs in f32 and is checked (a) alone, on the reference's
own fp32 features: same merges, values to rounding; (b) inside the pipeline: the engine's merged tokens equal the oracle's ToMe
applied to the ENGINE's encoder output.  The encoder itself (16-bit MFMA operands, f32 residual stream) is compared with the
reference's fp32 features at 2e-2 of the tensor's max."""
import os
import time

import numpy as np
import pytest
import torch

from blim_amd import synth
from blim_amd import vision as V
from oracle import vision_oracle as VO
from oracle.gen_golden_vision import CASES

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def relmax(a, b):
    return float(np.abs(np.asarray(a, np.float64) - b).max() / (np.abs(b).max() + 1e-30))


def _encoder(case, dtype="f16"):
    spec = CASES[case]
    enc = V.VisionEncoder(V.VisionDims(image_size=spec["image_size"]), dtype=dtype)
    enc.init_synthetic_weights(spec["wseed"])
    S = spec["image_size"]
    frames = torch.from_numpy(synth.tensor(spec["fseed"], "frames", (16, 3, S, S), std=1.0))
    return enc, frames, np.load(os.path.join(GOLD, f"vision_{case}.npz"))


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16], ids=["f16", "bf16"])
@pytest.mark.parametrize("L", [37, 100, 64 * 17 + 12, 3136], ids=lambda v: f"L{v}")
def test_attention_kernel_alone(L, dtype):
    """blim_vit_attention against the oracle's statement of the encoder's attention (oracle/vision_oracle.py:157-162 = vision_tower_builder.py:100-128 'origin'
    branch) on the same 16-bit q / k / v: both workgroup shapes (four waves below 1,024 tokens, eight from there), sequence lengths that end inside a 64-key
    tile and inside a 32-query block, and the extraction's own 3,136 tokens."""
    clips, heads = 2, 16
    g = torch.Generator().manual_seed(L)
    x = (torch.randn((clips * L, 3 * heads * 64), generator=g) * 1.5).to(dtype).cuda()
    got = V.vit_attention(x, clips, heads).float().cpu().numpy()
    q, k, v = x.float().cpu().numpy().reshape(clips, L, 3, heads, 64).transpose(2, 0, 3, 1, 4)
    s_ = (q * np.float32(0.125)) @ k.transpose(0, 1, 3, 2)
    p = np.exp(s_ - s_.max(axis=-1, keepdims=True))
    want = ((p / p.sum(axis=-1, keepdims=True, dtype=np.float32)) @ v).transpose(0, 2, 1, 3).reshape(clips * L, heads * 64)
    # the probabilities enter the second product rounded to 16 bits (as in the reference's fp16 run): 2^-11 resp. 2^-8 of a value of order |v|
    assert np.abs(got - want).max() < (4e-3 if dtype == torch.float16 else 3e-2) * max(1.0, float(np.abs(want).max()) / 4)
    assert relmax(got, want) < (1.5e-3 if dtype == torch.float16 else 1e-2)


def test_tome_alone_on_the_reference_features():
    enc, _, g = _encoder("small")
    try:
        x = torch.from_numpy(g["feat_clip0"][None])
        out = enc.tome_merge(x, 64).cpu().numpy()
        np.testing.assert_allclose(out[0], g["tome"][0], rtol=0, atol=2e-6)
        # ragged schedule + several batch entries at once: 144 -> 50 tokens, 3 entries, against the oracle
        rs = np.random.RandomState(4)
        xb = rs.randn(3, 144, 1024).astype(np.float32)
        want = VO.merge_tokens(xb, 50, 16)
        np.testing.assert_allclose(enc.tome_merge(torch.from_numpy(xb), 50).cpu().numpy(), want, rtol=0, atol=2e-6)
    finally:
        enc.close()


@pytest.mark.parametrize("dtype", ["f16", "bf16"])
def test_encoder_and_pipeline_small(dtype):
    enc, frames, g = _encoder("small", dtype)
    try:
        tome, feat = enc.encode(frames, want_feat=True)
        feat_h, tome_h = feat.cpu().numpy(), tome.cpu().numpy()
        tol = 2e-2 if dtype == "f16" else 6e-2
        assert relmax(feat_h[..., ::16], g["feat_sub16"]) < tol
        assert relmax(feat_h[0], g["feat_clip0"]) < tol
        # ToMe inside the pipeline == the oracle's ToMe on the engine's own encoder output
        np.testing.assert_allclose(tome_h, VO.merge_tokens(feat_h, 64, 16), rtol=0, atol=2e-6)
        # and the file a user gets: fp16 [4, 64, 1024]
        f = enc.video_feature(frames)
        assert tuple(f.shape) == (4, 64, 1024) and f.dtype == torch.float16 and torch.isfinite(f.float()).all()
    finally:
        enc.close()


def test_full_size_448(capsys):
    """The size the reference extracts at: 4 clips x 3136 tokens, 23 blocks; encoder vs the reference's fp32 features, ToMe 3136 -> 64
    consistent with the oracle on the engine's features, merged tokens vs the reference's (reported: discrete merges can differ)."""
    enc, frames, g = _encoder("448")
    try:
        tome, feat = enc.encode(frames, want_feat=True)
        torch.cuda.synchronize()
        t0 = time.time()
        for _ in range(3):
            enc.encode(frames)
        torch.cuda.synchronize()
        dt = (time.time() - t0) / 3
        feat_h, tome_h = feat.cpu().numpy(), tome.cpu().numpy()
        e_feat = relmax(feat_h[:, ::8, ::16], g["feat_sub"])
        assert e_feat < 2e-2
        np.testing.assert_allclose(tome_h[:1], VO.merge_tokens(feat_h[:1], 64, 16), rtol=0, atol=2e-6)
        ref = g["tome"].astype(np.float32)
        # token-set agreement with the reference's merged tokens: every engine token has a reference token within 5e-2 (relative L2)
        d = np.linalg.norm(tome_h[:, :, None, :] - ref[:, None, :, :], axis=-1) / np.linalg.norm(ref, axis=-1)[:, None, :]
        frac = float((d.min(axis=2) < 5e-2).mean())
        with capsys.disabled():
            print(f"\n[vision 448] encoder max abs err / max {e_feat:.2e}; merged tokens matching a reference token (5e-2): {100 * frac:.1f} %; "
                  f"one video (16 frames -> [4, 64, 1024]) in {dt * 1e3:.1f} ms")
        # the reference's own production numerics (fp16 weights under autocast, extract.py:96-108; emulated on CPU by the fixture generator)
        # against its fp32 run: the same measure.  The engine's 16-bit encoder flips merges at the rate the reference flips its own.
        h = g["tome_fp16_autocast"].astype(np.float32)
        dh = np.linalg.norm(h[:, :, None, :] - ref[:, None, :, :], axis=-1) / np.linalg.norm(ref, axis=-1)[:, None, :]
        frac_ref = float((dh.min(axis=2) < 5e-2).mean())
        with capsys.disabled():
            print(f"[vision 448] the reference's fp16-autocast run vs its own fp32 run, same measure: {100 * frac_ref:.1f} %")
        assert frac >= 0.95 and frac >= frac_ref - 0.02
    finally:
        enc.close()


def test_scores_on_engine_extracted_vs_reference_extracted_features(capsys):
    """What the merge flips do downstream: one video's features extracted by the engine, by the reference in fp32 and by the reference in its
    production numerics (fp16 autocast) -- `vision_448.npz` -- are each scored against four captions by the full 28-layer 7B decoder (VTG and TVG,
    fused path).  The engine-extracted file must move the scores no more than the reference's own fp16 extraction does (x2), and < 1e-2."""
    from blim_amd import retrieval_utils as RU
    from blim_amd.modeling import BlimModel, DDPLike
    enc, frames, g = _encoder("448")
    try:
        f_eng = enc.video_feature(frames).float().numpy()                       # the file a user gets: fp16 [4, 64, 1024]
    finally:
        enc.close()
    f_ref, f_ref16 = g["tome"].astype(np.float32), g["tome_fp16_autocast"].astype(np.float32)
    dims = synth.ModelDims()
    prob = synth.make_problem(41, 4, dims, tok_per_clip=64, text_len=(8, 24))
    model = BlimModel(dims, max_positions=1024, dtype="f16")
    model.engine.init_synthetic_weights(0)
    model.set_tvg_prefix_length(prob.tvg_prefix_length)
    tok = type("T", (), {"pad_token_id": synth.PAD_ID})()
    Tt = lambda rows: [torch.from_numpy(r) for r in rows]
    vtg = RU.padding_ids(Tt(prob.vtg_ids), Tt(prob.vtg_labels), Tt(prob.vtg_masks), tok)
    tvg = RU.padding_ids(Tt(prob.tvg_ids), Tt(prob.tvg_labels), Tt(prob.tvg_masks), tok)
    pairs = np.array([[0, i] for i in range(4)])
    res = {}
    try:
        for tag, f in (("reference fp32", f_ref), ("reference fp16 autocast", f_ref16), ("engine", f_eng)):
            scale = np.float32(1.0 / np.abs(f_ref).max())                       # the synthetic tower's output scale -> O(1) features, like the synthetic videos
            video = [torch.from_numpy(f * scale)] + [torch.from_numpy(v) for v in prob.video[1:]]
            vocab = np.stack([v.numpy().mean(axis=1) for v in video]).astype(np.float32)
            vocab[0] = (f_ref * scale).mean(axis=1)                              # one video vocabulary (the reference's) for all three runs
            sc = RU.PairScorer(DDPLike(model), vtg[0], vtg[2], vtg[1], tvg[0], tvg[2], tvg[1], video, torch.from_numpy(vocab),
                               torch.from_numpy(prob.tvg_video_labels), dims.num_clips)
            res[tag] = np.concatenate([sc.vtg(pairs), sc.tvg(pairs)])
    finally:
        model.engine.close()
    rel = lambda a, b: float(np.max(np.abs(a - b) / np.abs(b)))
    d_eng, d_ref16 = rel(res["engine"], res["reference fp32"]), rel(res["reference fp16 autocast"], res["reference fp32"])
    with capsys.disabled():
        print(f"\n[vision -> scores] worst relative score change vs the reference's fp32-extracted features (4 VTG + 4 TVG scores, 7B, 28 layers): "
              f"engine-extracted {d_eng:.2e}, reference fp16-autocast-extracted {d_ref16:.2e}")
    assert np.isfinite(res["engine"]).all()
    assert d_eng < 1e-2 and d_eng <= 2.0 * d_ref16 + 1e-3


def test_error_behaviour():
    with pytest.raises(V.eng.BlimError, match="head_dim"):
        V.VisionEncoder(V.VisionDims(image_size=96, hidden_size=1024, num_heads=8))
    enc = V.VisionEncoder(V.VisionDims(image_size=96, depth=1))
    try:
        with pytest.raises(V.eng.BlimError, match="not loaded"):
            enc.encode(torch.zeros((4, 3, 96, 96)))
    finally:
        enc.close()


def test_extract_driver_end_to_end(tmp_path, monkeypatch):
    """python -m blim_amd.extract on a synthetic tree of pre-decoded frames: writes ./data/<DS>/features/<vid>.pth = fp16 [4, 64, 1024]
    (extract.py:107-110), which the dataset front end then serves (base_dataset.py:23-31)."""
    from blim_amd import extract as X
    monkeypatch.chdir(tmp_path)
    rs = np.random.RandomState(0)
    os.makedirs("data/MSRVTT/frames")
    vids = ["video1", "video2", "video3"]
    for v in vids:
        np.save(f"data/MSRVTT/frames/{v}.npy", rs.randint(0, 256, size=(16, 120, 160, 3), dtype=np.uint8))
    args = X.get_args_parser().parse_args(["--dataset", "MSRVTT", "--num_chunk", "1", "--chunk_idx", "0", "--batch_size", "2", "--synthetic", "21", "--clear"])
    assert X.main(args) == 3
    feats = {v: torch.load(f"data/MSRVTT/features/{v}.pth", weights_only=True) for v in vids}
    for f in feats.values():
        assert tuple(f.shape) == (4, 64, 1024) and f.dtype == torch.float16 and torch.isfinite(f.float()).all()
    assert not torch.equal(feats["video1"], feats["video2"])
    # a batch of one gives the same file as the batch of two did (videos are independent)
    args1 = X.get_args_parser().parse_args(["--dataset", "MSRVTT", "--num_chunk", "3", "--chunk_idx", "0", "--synthetic", "21"])
    os.rename("data/MSRVTT/features/video1.pth", "data/MSRVTT/features/keep.pth")
    assert X.main(args1) == 1
    assert torch.equal(torch.load("data/MSRVTT/features/video1.pth", weights_only=True), torch.load("data/MSRVTT/features/keep.pth", weights_only=True))
