"""TEST INFRASTRUCTURE ONLY -- CPU (numpy, fp32) restatement of the reference's OFFLINE FEATURE EXTRACTION:
UMT-L vision encoder + ToMe token merging, i.e. what extract.py:96-110 computes for one video:

    frames [16, 3, S, S]  -> 4 clips of 4 frames -> ViT (23 of 24 blocks, vision_layernorm) -> [4, 4*(S/16)^2, 1024]
                          -> ToMe bipartite soft matching down to 64 tokens per clip          -> [4, 64, 1024]  (saved as fp16 .pth)

Only tests/ may import this file (it is the checker for blim_amd/vision.py + csrc/vision.hip).

Parity status: PINNED.  tests/golden/vision_small.npz and vision_448.npz were produced by oracle/gen_golden_vision.py, which runs
the reference's own UMTVisionTower / ToMe16_mlp_hd64 (videochat_flash/vision_tower_builder.py, mm_projector_builder.py) on seeded
frames with this repo's seeded synthetic weights; tests/test_vision_oracle.py checks every function below against those vectors.

Every function cites the reference lines it restates (paths relative to /root/reference/videochat_flash).
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, Tuple

import numpy as np


@dataclass
class VisionConfig:
    """build_vit's constants (vision_tower_builder.py:506-523) + UMTVisionConfig (:481-503)."""
    image_size: int = 448
    patch_size: int = 16
    num_frames: int = 4              # mm_local_num_frames: frames per clip
    hidden_size: int = 1024
    num_heads: int = 16
    mlp_hidden: int = 4096
    depth: int = 23                  # encoder_depth 24 + return_index (-2) + 1  (:293)
    tome_tokens_per_frame: int = 16  # mm_projector_builder.py:147: num_tome_tokens = 16 * num_frames
    ckpt_num_frame: int = 4

    @property
    def grid(self) -> int:
        return self.image_size // self.patch_size

    @property
    def tokens_per_clip(self) -> int:
        return self.num_frames * self.grid * self.grid


def weight_shapes(cfg: VisionConfig) -> Dict[str, Tuple[int, ...]]:
    D, Hm, P = cfg.hidden_size, cfg.mlp_hidden, cfg.patch_size
    s: Dict[str, Tuple[int, ...]] = {"vit.patch.w": (D, 3 * P * P), "vit.patch.b": (D,), "vit.norm.w": (D,), "vit.norm.b": (D,)}
    for i in range(cfg.depth):
        B = f"vit.blocks.{i}."
        s[B + "norm1.w"] = (D,); s[B + "norm1.b"] = (D,); s[B + "q_bias"] = (D,); s[B + "v_bias"] = (D,)
        s[B + "qkv.w"] = (3 * D, D); s[B + "proj.w"] = (D, D); s[B + "proj.b"] = (D,)
        s[B + "norm2.w"] = (D,); s[B + "norm2.b"] = (D,)
        s[B + "fc1.w"] = (Hm, D); s[B + "fc1.b"] = (Hm,); s[B + "fc2.w"] = (D, Hm); s[B + "fc2.b"] = (D,)
    return s


# ----------------------------------------------------------------------------- position table

def sinusoid_table(n_position: int, d_hid: int) -> np.ndarray:
    """vision_tower_builder.py:188-219 / 222-232: angle = pos / 10000^(2*(j//2)/d) in float64, sin on even, cos on odd columns,
    then torch.tensor(..., dtype=torch.float)."""
    j = np.arange(d_hid)
    ang = np.arange(n_position, dtype=np.float64)[:, None] / np.power(10000.0, 2 * (j // 2) / d_hid)[None, :]
    out = ang.copy()
    out[:, 0::2] = np.sin(ang[:, 0::2])
    out[:, 1::2] = np.cos(ang[:, 1::2])
    return out.astype(np.float32)


def _cubic_coeffs(t: np.ndarray, A: float = -0.75):
    """ATen UpSampleBicubic2d: cubic convolution weights for the 4 taps at offsets -1, 0, +1, +2."""
    def c1(x):  # |x| <= 1
        return ((A + 2) * x - (A + 3)) * x * x + 1
    def c2(x):  # 1 < |x| < 2
        return ((A * x - 5 * A) * x + 8 * A) * x - 4 * A
    return np.stack([c2(t + 1.0), c1(t), c1(1.0 - t), c2(2.0 - t)], axis=-1)


def bicubic_resize(x: np.ndarray, out_h: int, out_w: int) -> np.ndarray:
    """torch.nn.functional.interpolate(x [N,C,H,W], size, mode='bicubic', align_corners=False) in float32 (ATen semantics:
    source coordinate scale*(dst+0.5)-0.5, taps clamped to the border)."""
    n, c, h, w = x.shape

    def axis(in_n, out_n):
        scale = np.float32(in_n) / np.float32(out_n)
        src = (scale * (np.arange(out_n, dtype=np.float32) + np.float32(0.5)) - np.float32(0.5)).astype(np.float32)
        fl = np.floor(src)
        t = (src - fl).astype(np.float32)
        idx = np.clip(fl.astype(np.int64)[:, None] + np.arange(-1, 3)[None, :], 0, in_n - 1)
        return idx, _cubic_coeffs(t).astype(np.float32)

    iy, wy = axis(h, out_h)
    ix, wx = axis(w, out_w)
    # rows first (as ATen does: for every output pixel, 4 row taps each interpolated along x), all in float32
    tmp = np.zeros((n, c, h, out_w), dtype=np.float32)
    for k in range(4):
        tmp += x[:, :, :, ix[:, k]] * wx[None, None, None, :, k]
    out = np.zeros((n, c, out_h, out_w), dtype=np.float32)
    for k in range(4):
        out += tmp[:, :, iy[:, k], :] * wy[None, None, :, k, None]
    return out


def pos_embed(cfg: VisionConfig) -> np.ndarray:
    """[tokens_per_clip, D] table added after the patch embedding (vision_tower_builder.py:304-313, 353).
    image_size != 224: get_sinusoid_encoding_table2 (:222-269): 4 x 14 x 14 checkpoint table, bicubic to grid x grid.
    image_size == 224: get_sinusoid_encoding_table (:188-219).  Frames per clip == checkpoint frames (4): no temporal interpolation."""
    assert cfg.num_frames == cfg.ckpt_num_frame, "temporal interpolation of the position table is not on the extraction path"
    D, T, G = cfg.hidden_size, cfg.num_frames, cfg.grid
    if cfg.image_size == 224:
        return sinusoid_table(T * G * G, D)
    tab = sinusoid_table(T * 14 * 14, D)
    if G != 14:
        t4 = tab.reshape(T, 14, 14, D).transpose(0, 3, 1, 2)               # BT, C, H, W
        tab = bicubic_resize(np.ascontiguousarray(t4), G, G).transpose(0, 2, 3, 1).reshape(T * G * G, D)
    return np.ascontiguousarray(tab, dtype=np.float32)


# ----------------------------------------------------------------------------- ViT

def layer_norm(x: np.ndarray, w: np.ndarray, b: np.ndarray, eps: float) -> np.ndarray:
    mu = x.mean(axis=-1, keepdims=True, dtype=np.float32)
    var = ((x - mu) ** 2).mean(axis=-1, keepdims=True, dtype=np.float32)
    return ((x - mu) / np.sqrt(var + np.float32(eps)) * w + b).astype(np.float32)


def gelu(x: np.ndarray) -> np.ndarray:
    from scipy.special import erf
    return (np.float32(0.5) * x * (np.float32(1.0) + erf(x / np.float32(math.sqrt(2.0))))).astype(np.float32)


def patchify(frames: np.ndarray, P: int) -> np.ndarray:
    """frames [B, T, 3, S, S] -> [B, T*G*G, 3*P*P]: rows in (t, py, px) order (Conv3d kernel (1,P,P), stride the same, then
    flatten(2).transpose(1,2): vision_tower_builder.py:170-184), columns in (c, ky, kx) order = the Conv3d weight flattened."""
    B, T, C, S, _ = frames.shape
    G = S // P
    x = frames.reshape(B, T, C, G, P, G, P).transpose(0, 1, 3, 5, 2, 4, 6)   # B T Gy Gx C Ky Kx
    return np.ascontiguousarray(x.reshape(B, T * G * G, C * P * P))


def vit_forward(cfg: VisionConfig, w: Dict[str, np.ndarray], frames: np.ndarray, parts: dict = None) -> np.ndarray:
    """frames [B clips, T, 3, S, S] float32 -> [B, T*G*G, D] (UMTVisionTower.forward :558-571 -> PretrainVisionTransformerEncoder
    .forward_features :346-366; Block :130-159 with init_values 0 -> no gamma; Attention 'origin' branch :100-128)."""
    D, nh = cfg.hidden_size, cfg.num_heads
    hd = D // nh
    x = patchify(frames.astype(np.float32), cfg.patch_size) @ w["vit.patch.w"].T + w["vit.patch.b"]
    x = (x + pos_embed(cfg)[None]).astype(np.float32)
    if parts is not None:
        parts["embed"] = x.copy()
    B, L, _ = x.shape
    for i in range(cfg.depth):
        P = f"vit.blocks.{i}."
        h = layer_norm(x, w[P + "norm1.w"], w[P + "norm1.b"], 1e-6)
        bias = np.concatenate([w[P + "q_bias"], np.zeros(D, np.float32), w[P + "v_bias"]])
        qkv = (h @ w[P + "qkv.w"].T + bias).reshape(B, L, 3, nh, hd).transpose(2, 0, 3, 1, 4)
        q, k, v = qkv[0] * np.float32(hd ** -0.5), qkv[1], qkv[2]
        s = q @ k.transpose(0, 1, 3, 2)
        s = s - s.max(axis=-1, keepdims=True)
        p = np.exp(s)
        p = p / p.sum(axis=-1, keepdims=True, dtype=np.float32)
        a = (p @ v).transpose(0, 2, 1, 3).reshape(B, L, D)
        x = (x + a @ w[P + "proj.w"].T + w[P + "proj.b"]).astype(np.float32)
        h = layer_norm(x, w[P + "norm2.w"], w[P + "norm2.b"], 1e-6)
        x = (x + gelu(h @ w[P + "fc1.w"].T + w[P + "fc1.b"]) @ w[P + "fc2.w"].T + w[P + "fc2.b"]).astype(np.float32)
        if parts is not None and i == 0:
            parts["block0"] = x.copy()
    return layer_norm(x, w["vit.norm.w"], w["vit.norm.b"], 1e-12)


# ----------------------------------------------------------------------------- ToMe

def bipartite_soft_matching(metric: np.ndarray, r: int):
    """mm_projector_builder.py:6-55.  metric [b, t, c] -> (unm_idx [b, t1-r], src_idx [b, r], dst_idx [b, r]) over the even (a)
    / odd (b) token sets."""
    t = metric.shape[1]
    r = min(r, t // 2)
    assert r > 0
    m = metric / np.linalg.norm(metric, axis=-1, keepdims=True)
    a, b = m[:, ::2, :], m[:, 1::2, :]
    scores = a @ b.transpose(0, 2, 1)
    node_idx = scores.argmax(axis=-1)
    node_max = scores.max(axis=-1)
    edge_idx = np.argsort(-node_max, axis=-1, kind="stable")
    unm_idx, src_idx = edge_idx[:, r:], edge_idx[:, :r]
    dst_idx = np.take_along_axis(node_idx, src_idx, axis=1)
    return unm_idx, src_idx, dst_idx


def merge_sum(x: np.ndarray, unm_idx, src_idx, dst_idx) -> np.ndarray:
    """The `merge` closure (:35-42): unmerged even tokens, then the odd tokens with their merged sources added (scatter_add)."""
    src, dst = x[:, ::2, :], x[:, 1::2, :].copy()
    n = x.shape[0]
    unm = np.take_along_axis(src, unm_idx[:, :, None], axis=1)
    s = np.take_along_axis(src, src_idx[:, :, None], axis=1)
    for b in range(n):
        np.add.at(dst[b], dst_idx[b], s[b])                     # sequential, index order (what scatter_add does on CPU)
    return np.concatenate([unm, dst], axis=1)


def merge_tokens(x: np.ndarray, target: int, num_heads: int) -> np.ndarray:
    """ToMe16_mlp_hd64.merge_tokens (:100-130) with merge_wavg (:58-74): size-weighted averages, metric = mean over heads."""
    b, p, c = x.shape
    assert p > target
    r_list, tmp = [], p
    while tmp != target:
        if tmp - target <= tmp // 2:
            r_list.append(tmp - target)
            break
        r_list.append(tmp // 2)
        tmp -= tmp // 2
    size = None
    dim = c // num_heads
    x = x.astype(np.float32)
    for r in r_list:
        p = x.shape[1]
        metric = x.reshape(b, p, num_heads, dim).mean(axis=2, dtype=np.float32)
        idx = bipartite_soft_matching(metric, r)
        if size is None:
            size = np.ones((b, p, 1), dtype=np.float32)
        x = merge_sum(x * size, *idx)
        size = merge_sum(size, *idx)
        x = (x / size).astype(np.float32)
    return x


def encode_video(cfg: VisionConfig, w: Dict[str, np.ndarray], frames: np.ndarray) -> np.ndarray:
    """frames [n_clips * T, 3, S, S] -> [n_clips, 16 * T, D]: encode_video_image(..., return_video_feature=True)
    (modeling_videochat_flash.py:126-181) as extract.py:104 calls it."""
    T = cfg.num_frames
    n = frames.shape[0] // T
    feat = vit_forward(cfg, w, frames.reshape(n, T, *frames.shape[1:]))
    return merge_tokens(feat, cfg.tome_tokens_per_frame * T, cfg.num_heads)
