"""CPU: the numpy restatement of the offline feature extraction (oracle/vision_oracle.py: UMT-L ViT + ToMe) against the golden
vectors recorded from the REFERENCE's own UMTVisionTower / ToMe16_mlp_hd64 (oracle/gen_golden_vision.py), and the host-side pieces
of blim_amd/vision.py (position table, weight naming, preprocessing)."""
import os

import numpy as np
import pytest

from blim_amd import synth
from blim_amd import vision as V
from oracle import vision_oracle as VO
from oracle.gen_golden_vision import CASES, ref_key, synthetic_weights

GOLD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def small():
    g = np.load(os.path.join(GOLD, "vision_small.npz"))
    spec = CASES["small"]
    cfg = VO.VisionConfig(image_size=spec["image_size"])
    w = synthetic_weights(cfg, spec["wseed"])
    frames = synth.tensor(spec["fseed"], "frames", (16, 3, cfg.image_size, cfg.image_size), std=1.0)
    return dict(g=g, cfg=cfg, w=w, frames=frames)


def test_vit_restatement_matches_the_reference(small):
    g, cfg, w, frames = small["g"], small["cfg"], small["w"], small["frames"]
    np.testing.assert_allclose(VO.pos_embed(cfg)[:, ::16], g["pos_embed_sub16"], atol=5e-6)
    parts = {}
    feat = VO.vit_forward(cfg, w, frames.reshape(4, 4, 3, cfg.image_size, cfg.image_size), parts)
    np.testing.assert_allclose(parts["embed"][..., ::16], g["embed_sub16"], atol=1e-5)
    np.testing.assert_allclose(parts["block0"][..., ::16], g["block0_sub16"], atol=2e-5)
    np.testing.assert_allclose(feat[..., ::16], g["feat_sub16"], atol=5e-5)
    np.testing.assert_allclose(feat[0], g["feat_clip0"], atol=5e-5)
    # end to end (same merges as the reference => the merged tokens agree to rounding)
    np.testing.assert_allclose(VO.merge_tokens(feat, 64, cfg.num_heads), g["tome"], atol=5e-5)


def test_tome_restatement_is_exact_on_the_reference_features(small):
    g = small["g"]
    out = VO.merge_tokens(g["feat_clip0"][None], 64, 16)
    assert np.array_equal(out[0], g["tome"][0])
    # the merge schedule of merge_tokens (mm_projector_builder.py:108-115): 3136 -> 64 at full size
    tmp, rs = 3136, []
    while tmp != 64:
        r = tmp - 64 if tmp - 64 <= tmp // 2 else tmp // 2
        rs.append(r); tmp -= r
    assert rs == [1568, 784, 392, 196, 98, 34]


def test_full_size_restatement_on_clip0_columns():
    """448 x 448 (28 x 28 patches x 4 frames = 3136 tokens per clip): the position table and the ToMe stage at the size the
    reference extracts at; the 23-block tower itself is checked at 96 x 96 above (the same code, 13 s per clip on this CPU)."""
    g = np.load(os.path.join(GOLD, "vision_448.npz"))
    cfg = VO.VisionConfig(image_size=448)
    np.testing.assert_allclose(VO.pos_embed(cfg)[::8, ::16], g["pos_embed_sub"], atol=5e-6)
    assert g["tome"].shape == (4, 64, 1024) and g["feat_sub"].shape == (4, 392, 64)


def test_product_position_table_and_bicubic_match_torch(small):
    import torch
    for S in (96, 224, 448):
        d = V.VisionDims(image_size=S)
        np.testing.assert_array_equal(V.pos_embed(d), VO.pos_embed(VO.VisionConfig(image_size=S)))
    x = np.random.RandomState(0).randn(2, 3, 14, 14).astype(np.float32)
    for size in (6, 28, 17):
        want = torch.nn.functional.interpolate(torch.from_numpy(x), size=(size, size), mode="bicubic", align_corners=False).numpy()
        np.testing.assert_allclose(V._bicubic_resize(x, size, size), want, atol=5e-6)
    np.testing.assert_allclose(V.pos_embed(V.VisionDims(image_size=96))[:, ::16], small["g"]["pos_embed_sub16"], atol=5e-6)


def test_weight_names_shapes_and_checkpoint_keys():
    d = V.VisionDims()
    shapes = V.vision_weight_shapes(d)
    assert shapes == VO.weight_shapes(VO.VisionConfig())
    assert len(shapes) == 4 + 13 * 23 and sum(int(np.prod(s)) for s in shapes.values()) == 290_479_104      # the reference tower's parameter count
    for n in shapes:
        assert V.checkpoint_key(n) == "model.vision_tower." + ref_key(n)
    assert V.vision_weight_dist("vit.blocks.3.norm2.w") == (0.1, 1.0) and V.vision_weight_dist("vit.blocks.3.norm2.b") == (0.02, 0.0)


def test_preprocess_and_frame_sampling():
    rs = np.random.RandomState(1)
    frames = rs.randint(0, 256, size=(3, 60, 80, 3), dtype=np.uint8)
    out = V.preprocess(frames, image_size=32)
    assert tuple(out.shape) == (3, 3, 32, 32) and str(out.dtype) == "torch.float16"
    same = V.preprocess(rs.randint(0, 256, size=(1, 32, 32, 3), dtype=np.uint8), image_size=32)       # no resize: pure rescale + normalise
    assert np.isfinite(same.float().numpy()).all() and abs(float(same.float().mean())) < 3.0
    assert list(V.sample_frame_indices(100, 16)) == list(np.linspace(0, 98, 16, dtype=int))
    # a DiDeMo video cut to 30 s at a fractional frame rate: vlen = 30 * fps is a float in the reference (extract.py:50-54)
    assert list(V.sample_frame_indices(30 * 23.976, 16)) == list(np.linspace(0, 30 * 23.976 - 2, 16, dtype=int))
    assert list(V.sample_frame_indices(30 * 23.976, 16)) != list(V.sample_frame_indices(int(30 * 23.976), 16))


def test_library_exports_the_vision_symbols():
    from blim_amd import engine as eng
    lib = eng.load_library()
    for s in ("blim_vision_create", "blim_vision_destroy", "blim_vision_load_weight", "blim_vision_init_synthetic_weights", "blim_vision_set_pos_embed",
              "blim_vision_ready", "blim_vision_encode", "blim_tome_merge"):
        assert hasattr(lib, s) and s in eng.declared_symbols()


def test_extract_driver_host_logic(tmp_path):
    """Chunking and video ids as the reference's extract.py computes them (extract.py:66-69, 83-90)."""
    from blim_amd import extract as X
    items = list(range(10))
    assert [X.chunk_of(items, 3, i) for i in range(3)] == [[0, 1, 2], [3, 4, 5], [6, 7, 8, 9]]
    assert X.chunk_of(items, 1, 0) == items
    assert X.video_id("./data/MSRVTT/videos/video7010.mp4", "MSRVTT") == "video7010"
    assert X.video_id("./data/LSMDC/videos/a/0001_American_Beauty_00.00.51.926-00.00.54.129.avi", "LSMDC") == "0001_American_Beauty_00.00.51.926-00.00.54.129"
    a = X.get_args_parser().parse_args(["--num_chunk", "2", "--chunk_idx", "1"])
    assert a.dataset == "DiDeMo" and a.num_frames == 16 and a.batch_size == 1 and a.model_path.endswith("VideoChat-Flash-Qwen2-7B_res448")


@pytest.mark.needs_reference
def test_preprocess_matches_the_reference_image_processor():
    """Build container only: blim_amd.vision.preprocess against the reference's own UMTImageProcessor (vision_tower_builder.py:441-475) on
    random uint8 frames, with and without a resize."""
    import torch
    from oracle import ref_harness
    ref_harness.load()
    from videochat_flash.vision_tower_builder import UMTImageProcessor
    rs = np.random.RandomState(5)
    for shape, S in (((3, 48, 64, 3), 32), ((2, 32, 32, 3), 32), ((2, 100, 75, 3), 64)):
        frames = rs.randint(0, 256, size=shape, dtype=np.uint8)
        want = UMTImageProcessor(size=(S, S)).preprocess(frames, return_tensors="pt")["pixel_values"].half()
        got = V.preprocess(frames, image_size=S)
        assert got.shape == want.shape
        # one fp16 ulp at most (the reference rescales and normalises in a different operation order)
        assert float((got.float() - want.float()).abs().max()) <= 2e-3
