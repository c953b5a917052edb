"""TEST INFRASTRUCTURE ONLY -- generates tests/golden/vision_*.npz by running the REFERENCE's own vision tower and ToMe merge
(videochat_flash/vision_tower_builder.py UMTVisionTower, mm_projector_builder.py ToMe16_mlp_hd64) on CPU in fp32.

Run in the build container only (needs /root/reference):  python -m oracle.gen_golden_vision [--case small|448|all]

The tower's attention is switched to its own 'origin' (eager) branch -- flash_attn is not installed, and the two branches
compute the same softmax(QK^T/sqrt(d))V (vision_tower_builder.py:100-128).  Weights are this repo's seeded synthetic ones
(regenerated from the seed by the tests); fixtures hold outputs only."""
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
from oracle.vision_oracle import VisionConfig, weight_shapes  # noqa: E402

CASES = {"small": dict(image_size=96, wseed=21, fseed=31), "448": dict(image_size=448, wseed=21, fseed=32)}


def weight_dist(name: str):
    if name.endswith("norm1.w") or name.endswith("norm2.w") or name == "vit.norm.w":
        return 0.1, 1.0
    return 0.02, 0.0


def synthetic_weights(cfg: VisionConfig, seed: int):
    from oracle.gen_golden import fast_tensor
    return {n: fast_tensor(seed, n, s, *weight_dist(n)) for n, s in weight_shapes(cfg).items()}


def ref_key(name: str) -> str:
    if name == "vit.patch.w": return "vision_tower.encoder.patch_embed.proj.weight"
    if name == "vit.patch.b": return "vision_tower.encoder.patch_embed.proj.bias"
    if name == "vit.norm.w": return "vision_tower.encoder.vision_layernorm.weight"
    if name == "vit.norm.b": return "vision_tower.encoder.vision_layernorm.bias"
    _, _, i, rest = name.split(".", 3)
    m = {"norm1.w": "norm1.weight", "norm1.b": "norm1.bias", "q_bias": "attn.q_bias", "v_bias": "attn.v_bias", "qkv.w": "attn.qkv.weight",
         "proj.w": "attn.proj.weight", "proj.b": "attn.proj.bias", "norm2.w": "norm2.weight", "norm2.b": "norm2.bias",
         "fc1.w": "mlp.fc1.weight", "fc1.b": "mlp.fc1.bias", "fc2.w": "mlp.fc2.weight", "fc2.b": "mlp.fc2.bias"}[rest]
    return f"vision_tower.encoder.blocks.{i}.{m}"


def build_reference(cfg: VisionConfig, weights):
    import torch
    ref_harness.load()
    from videochat_flash import mm_projector_builder as MP
    from videochat_flash import vision_tower_builder as VT
    tcfg = types.SimpleNamespace(mm_local_num_frames=cfg.num_frames, mm_vision_select_layer=-2)
    tower = VT.UMTVisionTower("umt-hd-golden", tcfg, delay_load=False, image_size=cfg.image_size).eval().float()
    for blk in tower.vision_tower.encoder.blocks:          # eager branch of the reference's own Attention.forward
        blk.attn.attn_type = "origin"
        blk.attn.attn_drop = torch.nn.Identity()
    sd = tower.state_dict()
    with torch.no_grad():
        for name, arr in weights.items():
            k = ref_key(name)
            sd[k].copy_(torch.from_numpy(arr).reshape(sd[k].shape))
    pcfg = types.SimpleNamespace(mm_hidden_size=cfg.hidden_size, hidden_size=64, mm_pos_num_frames=8)
    proj = MP.ToMe16_mlp_hd64(pcfg, tower.config).eval()
    return tower, proj


def run_case(name: str, out_dir: str):
    import torch
    torch.set_num_threads(8)
    spec = CASES[name]
    cfg = VisionConfig(image_size=spec["image_size"])
    t0 = time.time()
    w = synthetic_weights(cfg, spec["wseed"])
    tower, proj = build_reference(cfg, w)
    print(f"[{name}] reference tower built in {time.time() - t0:.1f}s", flush=True)
    S, T = cfg.image_size, cfg.num_frames
    frames = synth.tensor(spec["fseed"], "frames", (4 * T, 3, S, S), std=1.0)
    out = {}
    with torch.no_grad():
        t0 = time.time()
        x = torch.from_numpy(frames).reshape(4, T, 3, S, S)
        # intermediates from the reference's own modules: patch embedding + position table, and the first block
        enc = tower.vision_tower.encoder
        emb = enc.patch_embed(x.permute(0, 2, 1, 3, 4)) + enc.pos_embed
        if name == "small":
            out["embed_sub16"] = emb.numpy()[..., ::16].copy()
            out["block0_sub16"] = enc.blocks[0](emb).numpy()[..., ::16].copy()
            out["pos_embed_sub16"] = enc.pos_embed.numpy()[0, :, ::16].copy()
        else:                                              # full size: every 8th token x every 16th column keeps the fixture ~1 MB
            out["pos_embed_sub"] = enc.pos_embed.numpy()[0, ::8, ::16].copy()
        feat = tower(x)                                                           # [4, T*G*G, D]
        print(f"[{name}] tower forward: {time.time() - t0:.1f}s", flush=True)
        vf = [v.reshape(-1, v.shape[-2] // T, v.shape[-1]) for v in torch.split(feat, [4])]   # modeling_videochat_flash.py:153
        tome = proj(vf[0], compress=True, local_num_frames=T, return_video_feature=True)         # [4, 64, D]
    # The reference's PRODUCTION numerics (extract.py:96, 105-108: `.half()` weights under autocast(float16), features saved as fp16), emulated on CPU:
    # the same tower and ToMe with fp16 weights inside torch.autocast("cpu", float16).  ToMe makes discrete arg-max / arg-sort choices, so this run
    # merges a few tokens differently from the fp32 run -- the fixture records by how much the reference disagrees with ITSELF across precisions,
    # the yardstick for the engine's 16-bit encoder (tests/test_vision_gpu.py).
    with torch.no_grad():
        t0 = time.time()
        tower_h = tower.half()
        with torch.autocast(device_type="cpu", dtype=torch.float16):
            feat_h = tower_h(x.half())
            vf_h = [v.reshape(-1, v.shape[-2] // T, v.shape[-1]) for v in torch.split(feat_h, [4])]
            tome_h = proj(vf_h[0], compress=True, local_num_frames=T, return_video_feature=True)
        out["tome_fp16_autocast"] = tome_h.float().numpy().astype(np.float16)
        print(f"[{name}] fp16-autocast tower + ToMe: {time.time() - t0:.1f}s", flush=True)
    if name == "small":
        out["feat_sub16"] = feat.numpy()[..., ::16].copy()
        out["feat_clip0"] = feat.numpy()[0].copy()
        out["tome"] = tome.numpy()
    else:
        out["feat_sub"] = feat.numpy()[:, ::8, ::16].copy()
        out["tome"] = tome.numpy().astype(np.float16)
    out["meta_case"] = np.array(name)
    path = os.path.join(out_dir, f"vision_{name}.npz")
    np.savez_compressed(path, **out)
    print(f"[{name}] wrote {path} ({os.path.getsize(path) / 1e6:.2f} MB)", flush=True)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--case", default="all")
    ap.add_argument("--out", default=os.path.join(ROOT, "tests", "golden"))
    a = ap.parse_args()
    if not ref_harness.available():
        sys.exit("reference not present; fixtures can only be generated in the build container")
    for c in (CASES if a.case == "all" else a.case.split(",")):
        run_case(c, a.out)
