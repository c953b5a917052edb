"""Offline feature extraction on the MI355X engine: host side of blim_vision_* (include/blim.h).

Mirrors what the reference's extract.py does per video (extract.py:96-110):

    frames uint8 [16, H, W, 3] --UMTImageProcessor--> fp16 [16, 3, 448, 448]            (vision_tower_builder.py:441-475)
      --model.encode_video_image(video, ..., return_video_feature=True)-->  [4, 64, 1024] (modeling_videochat_flash.py:126-181:
        4 clips x 4 frames through UMTVisionTower, vision_tower_builder.py:525-571, then ToMe 3136 -> 64 tokens per clip,
        mm_projector_builder.py:100-154)
      --torch.save(feature.half(), ./data/<DS>/features/<vid>.pth)-->                    the files blim_amd.dataloader reads

All tensor work runs in HIP (csrc/vision.hip + the engine's GEMM); this module holds the configuration, the position table
(host numpy, computed once), the weight naming and the ctypes binding.  No CPU fallback.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import Dict, Optional, Tuple

import numpy as np

from . import engine as eng

IMAGE_MEAN, IMAGE_STD = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)      # UMTImageProcessor defaults, vision_tower_builder.py:442


@dataclass
class VisionDims:
    """build_vit's constants (vision_tower_builder.py:506-523) and the 'umt-hd' image size (:613-614)."""
    image_size: int = 448
    patch_size: int = 16
    num_frames: int = 4              # mm_local_num_frames
    hidden_size: int = 1024
    num_heads: int = 16
    mlp_hidden: int = 4096
    depth: int = 23                  # encoder_depth 24 + mm_vision_select_layer (-2) + 1
    tome_tokens: int = 64            # 16 * num_frames (mm_projector_builder.py:147)

    @property
    def grid(self) -> int:
        return self.image_size // self.patch_size

    @property
    def tokens_per_clip(self) -> int:
        return self.num_frames * self.grid * self.grid


def vision_weight_shapes(d: VisionDims) -> Dict[str, Tuple[int, ...]]:
    D, Hm, P = d.hidden_size, d.mlp_hidden, d.patch_size
    s: Dict[str, Tuple[int, ...]] = {"vit.patch.w": (D, 3 * P * P), "vit.patch.b": (D,), "vit.norm.w": (D,), "vit.norm.b": (D,)}
    for i in range(d.depth):
        B = f"vit.blocks.{i}."
        s[B + "norm1.w"] = (D,); s[B + "norm1.b"] = (D,); s[B + "q_bias"] = (D,); s[B + "v_bias"] = (D,)
        s[B + "qkv.w"] = (3 * D, D); s[B + "proj.w"] = (D, D); s[B + "proj.b"] = (D,)
        s[B + "norm2.w"] = (D,); s[B + "norm2.b"] = (D,)
        s[B + "fc1.w"] = (Hm, D); s[B + "fc1.b"] = (Hm,); s[B + "fc2.w"] = (D, Hm); s[B + "fc2.b"] = (D,)
    return s


def vision_weight_dist(name: str) -> Tuple[float, float]:
    """(std, mean) of a synthetic tensor: LayerNorm gains bell(1, 0.1), everything else bell(0, 0.02) -- the rule
    blim_vision_init_synthetic_weights applies on device."""
    return (0.1, 1.0) if (name.endswith("norm1.w") or name.endswith("norm2.w") or name == "vit.norm.w") else (0.02, 0.0)


def checkpoint_key(name: str) -> str:
    """Canonical tensor name -> key in the VideoChat-Flash checkpoint (model.vision_tower = UMTVisionTower, whose .vision_tower is
    the PretrainVisionTransformer with its .encoder: vision_tower_builder.py:272-433, 525-556)."""
    root = "model.vision_tower.vision_tower.encoder."
    if name == "vit.patch.w": return root + "patch_embed.proj.weight"          # [D, 3, 1, P, P]: flattened to [D, 3*P*P] on load
    if name == "vit.patch.b": return root + "patch_embed.proj.bias"
    if name == "vit.norm.w": return root + "vision_layernorm.weight"
    if name == "vit.norm.b": return root + "vision_layernorm.bias"
    _, _, i, rest = name.split(".", 3)
    m = {"norm1.w": "norm1.weight", "norm1.b": "norm1.bias", "q_bias": "attn.q_bias", "v_bias": "attn.v_bias", "qkv.w": "attn.qkv.weight",
         "proj.w": "attn.proj.weight", "proj.b": "attn.proj.bias", "norm2.w": "norm2.weight", "norm2.b": "norm2.bias",
         "fc1.w": "mlp.fc1.weight", "fc1.b": "mlp.fc1.bias", "fc2.w": "mlp.fc2.weight", "fc2.b": "mlp.fc2.bias"}[rest]
    return f"{root}blocks.{i}.{m}"


# ----------------------------------------------------------------------------- position table (host, once)

def _sinusoid_table(n_position: int, d_hid: int) -> np.ndarray:
    """vision_tower_builder.py:188-232: float64 angles pos / 10000^(2*(j//2)/d), sin on even and cos on odd columns, cast to f32."""
    j = np.arange(d_hid)
    ang = np.arange(n_position, dtype=np.float64)[:, None] / np.power(10000.0, 2 * (j // 2) / d_hid)[None, :]
    out = ang.copy()
    out[:, 0::2] = np.sin(ang[:, 0::2]); out[:, 1::2] = np.cos(ang[:, 1::2])
    return out.astype(np.float32)


def _bicubic_axis(in_n: int, out_n: int):
    """Source taps and cubic-convolution weights (A = -0.75) of one axis of F.interpolate(mode='bicubic', align_corners=False)."""
    A = np.float32(-0.75)
    scale = np.float32(in_n) / np.float32(out_n)
    src = (scale * (np.arange(out_n, dtype=np.float32) + np.float32(0.5)) - np.float32(0.5)).astype(np.float32)
    fl = np.floor(src)
    t = (src - fl).astype(np.float32)
    c1 = lambda x: ((A + 2) * x - (A + 3)) * x * x + 1
    c2 = lambda x: ((A * x - 5 * A) * x + 8 * A) * x - 4 * A
    w = np.stack([c2(t + 1), c1(t), c1(1 - t), c2(2 - t)], axis=-1).astype(np.float32)
    idx = np.clip(fl.astype(np.int64)[:, None] + np.arange(-1, 3)[None, :], 0, in_n - 1)
    return idx, w


def _bicubic_resize(x: np.ndarray, out_h: int, out_w: int) -> np.ndarray:
    """x [N, C, H, W] f32 -> [N, C, out_h, out_w]."""
    iy, wy = _bicubic_axis(x.shape[2], out_h)
    ix, wx = _bicubic_axis(x.shape[3], out_w)
    tmp = np.zeros(x.shape[:3] + (out_w,), dtype=np.float32)
    for k in range(4):
        tmp += x[:, :, :, ix[:, k]] * wx[None, None, None, :, k]
    out = np.zeros(x.shape[:2] + (out_h, out_w), dtype=np.float32)
    for k in range(4):
        out += tmp[:, :, iy[:, k], :] * wy[None, None, :, k, None]
    return out


def pos_embed(d: VisionDims) -> np.ndarray:
    """[tokens_per_clip, hidden] f32 table added to the patch embeddings (vision_tower_builder.py:304-313, 353): the 4 x 14 x 14
    checkpoint table, bicubically resized to the grid when image_size != 224 (get_sinusoid_encoding_table2, :222-269)."""
    if d.num_frames != 4:
        raise NotImplementedError("position table: clips of 4 frames (the checkpoint's frame count; no temporal interpolation on the extraction path)")
    D, T, G = d.hidden_size, d.num_frames, d.grid
    if d.image_size == 224:
        return _sinusoid_table(T * G * G, D)
    tab = _sinusoid_table(T * 14 * 14, D)
    if G != 14:
        t4 = np.ascontiguousarray(tab.reshape(T, 14, 14, D).transpose(0, 3, 1, 2))
        tab = _bicubic_resize(t4, G, G).transpose(0, 2, 3, 1).reshape(T * G * G, D)
    return np.ascontiguousarray(tab, dtype=np.float32)


# ----------------------------------------------------------------------------- binding

class VisionConfigC(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("image_size", "patch_size", "num_frames", "hidden_size", "num_heads", "mlp_hidden", "depth", "tome_tokens",
                                          "compute_dtype")]


_bound = False


def _lib():
    global _bound
    lib = eng.load_library()
    if not _bound:
        vp, i32, u64 = C.c_void_p, C.c_int32, C.c_uint64
        lib.blim_vision_create.argtypes = [C.POINTER(VisionConfigC), C.POINTER(vp)]; lib.blim_vision_create.restype = C.c_int
        lib.blim_vision_destroy.argtypes = [vp]; lib.blim_vision_destroy.restype = None
        lib.blim_vision_load_weight.argtypes = [vp, C.c_char_p, vp, i32, i32]; lib.blim_vision_load_weight.restype = C.c_int
        lib.blim_vision_init_synthetic_weights.argtypes = [vp, u64]; lib.blim_vision_init_synthetic_weights.restype = C.c_int
        lib.blim_vision_set_pos_embed.argtypes = [vp, vp]; lib.blim_vision_set_pos_embed.restype = C.c_int
        lib.blim_vision_ready.argtypes = [vp]; lib.blim_vision_ready.restype = C.c_int
        lib.blim_vision_encode.argtypes = [vp, vp, i32, vp, vp, vp]; lib.blim_vision_encode.restype = C.c_int
        lib.blim_tome_merge.argtypes = [vp, vp, i32, i32, i32, i32, i32, vp, vp]; lib.blim_tome_merge.restype = C.c_int
        _bound = True
    return lib


class VisionEncoder:
    """One vision encoder on the current HIP device."""

    def __init__(self, dims: Optional[VisionDims] = None, dtype: Optional[str] = None):
        import torch
        self.lib = _lib()
        if not torch.cuda.is_available():
            raise eng.BlimError("no HIP device visible: the vision encoder has no CPU fallback")
        self.dims = dims or VisionDims()
        self.dtype = dtype or "f16"                              # extract.py runs the tower under autocast(float16)
        if self.dtype not in ("f16", "bf16"):
            raise ValueError("vision encoder dtype: f16 or bf16")
        self.torch_dtype = eng.torch_dtype_of(self.dtype)
        d = self.dims
        cfg = VisionConfigC(d.image_size, d.patch_size, d.num_frames, d.hidden_size, d.num_heads, d.mlp_hidden, d.depth, d.tome_tokens,
                            eng.COMPUTE_DTYPES[self.dtype])
        h = C.c_void_p()
        eng._check(self.lib.blim_vision_create(C.byref(cfg), C.byref(h)), "blim_vision_create")
        self.h = h
        self.device = torch.device("cuda", torch.cuda.current_device())
        tab = pos_embed(d)
        eng._check(self.lib.blim_vision_set_pos_embed(self.h, tab.ctypes.data), "blim_vision_set_pos_embed")

    def close(self):
        if getattr(self, "h", None):
            self.lib.blim_vision_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- weights
    def load_weight(self, name: str, arr: np.ndarray):
        shape = vision_weight_shapes(self.dims)[name]
        a = np.ascontiguousarray(arr, dtype=np.float32).reshape(shape)      # the Conv3d kernel [D, 3, 1, P, P] flattens to [D, 3*P*P]
        eng._check(self.lib.blim_vision_load_weight(self.h, name.encode(), a.ctypes.data, eng.DTYPE_F32, 0), f"blim_vision_load_weight({name})")

    def load_weights(self, weights: Dict[str, np.ndarray]):
        for name, arr in weights.items():
            self.load_weight(name, arr)
        eng._check(self.lib.blim_vision_ready(self.h), "blim_vision_ready")

    def load_checkpoint(self, model_path: str):
        """The vision-tower tensors of the VideoChat-Flash checkpoint (sharded safetensors; main.py:96 loads the same files)."""
        from .checkpoint import open_base_checkpoint
        keys, get = open_base_checkpoint(model_path)
        have = set(keys)
        for name in vision_weight_shapes(self.dims):
            k = checkpoint_key(name)
            if k not in have:
                raise KeyError(f"vision tensor '{name}' ({k}) not found in {model_path}")
            self.load_weight(name, get(k))
        eng._check(self.lib.blim_vision_ready(self.h), "blim_vision_ready")

    def init_synthetic_weights(self, seed: int):
        eng._check(self.lib.blim_vision_init_synthetic_weights(self.h, seed), "blim_vision_init_synthetic_weights")

    # ---- compute
    def encode(self, frames, want_feat: bool = False):
        """frames: device tensor [n_clips * T, 3, S, S] (or [n_clips, T, 3, S, S]) of normalised pixels.
        Returns (tome f32 [n_clips, tome_tokens, D], feat f32 [n_clips, L, D] | None)."""
        import torch
        d = self.dims
        x = frames.to(device=self.device, dtype=self.torch_dtype).reshape(-1, d.num_frames, 3, d.image_size, d.image_size).contiguous()
        n = x.shape[0]
        tome = torch.empty((n, d.tome_tokens, d.hidden_size), dtype=torch.float32, device=self.device)
        feat = torch.empty((n, d.tokens_per_clip, d.hidden_size), dtype=torch.float32, device=self.device) if want_feat else None
        eng._check(self.lib.blim_vision_encode(self.h, eng._ptr(x), n, eng._ptr(feat), eng._ptr(tome), eng._stream()), "blim_vision_encode")
        return tome, feat

    def tome_merge(self, x, target: int):
        """ToMe alone on f32 tokens [b, p, c] -> [b, target, c] (bipartite soft matching, size-weighted averages)."""
        import torch
        b, p, c = x.shape
        x = x.to(device=self.device, dtype=torch.float32).contiguous()
        out = torch.empty((b, target, c), dtype=torch.float32, device=self.device)
        eng._check(self.lib.blim_tome_merge(self.h, eng._ptr(x), b, p, c, c // 64, target, eng._ptr(out), eng._stream()), "blim_tome_merge")
        return out

    def video_feature(self, frames):
        """One video's frames [n_clips * T, 3, S, S] -> the tensor extract.py:107-110 saves: fp16 [n_clips, tome_tokens, D] on host."""
        import torch
        tome, _ = self.encode(frames)
        return tome.to(torch.float16).cpu()


def vit_attention(qkv, n_clips: int, heads: int):
    """The encoder's attention alone (blim_vit_attention): qkv 16-bit device tensor [n_clips * L, 3 * heads * 64] -> [n_clips * L, heads * 64]."""
    import torch
    lib = eng.load_library()
    vp, i32 = C.c_void_p, C.c_int32
    lib.blim_vit_attention.argtypes = [vp, i32, i32, i32, i32, vp, vp]; lib.blim_vit_attention.restype = C.c_int
    assert qkv.dtype in (torch.float16, torch.bfloat16) and qkv.shape[1] == 3 * heads * 64 and qkv.shape[0] % n_clips == 0
    out = torch.empty((qkv.shape[0], heads * 64), dtype=qkv.dtype, device=qkv.device)
    eng._check(lib.blim_vit_attention(eng._ptr(qkv), n_clips, qkv.shape[0] // n_clips, heads, eng.COMPUTE_DTYPES["f16" if qkv.dtype == torch.float16 else "bf16"],
                                      eng._ptr(out), eng._stream()), "blim_vit_attention")
    return out


# ----------------------------------------------------------------------------- preprocessing (host)

def preprocess(frames_u8: np.ndarray, image_size: int = 448):
    """UMTImageProcessor.preprocess (vision_tower_builder.py:454-475): RGB uint8 [T, H, W, 3] -> bicubic resize to (S, S) (PIL, as
    transformers' `resize` does for uint8 images), x 1/255, normalise with the ImageNet mean / std, channels first; returns a
    torch half tensor [T, 3, S, S] (extract.py:58 .half())."""
    import torch
    from PIL import Image
    out = np.empty((len(frames_u8), 3, image_size, image_size), dtype=np.float32)
    mean = np.asarray(IMAGE_MEAN, np.float32)[:, None, None]; std = np.asarray(IMAGE_STD, np.float32)[:, None, None]
    for i, f in enumerate(frames_u8):
        img = Image.fromarray(np.asarray(f, dtype=np.uint8)).convert("RGB")
        if img.size != (image_size, image_size):
            img = img.resize((image_size, image_size), resample=Image.BICUBIC)
        a = np.asarray(img, dtype=np.float32).transpose(2, 0, 1) * np.float32(1.0 / 255.0)
        out[i] = (a - mean) / std
    return torch.from_numpy(out).half()


def sample_frame_indices(vlen, num_frames: int = 16) -> np.ndarray:
    """extract.py:54: np.linspace(0, vlen - 2, num_frames, dtype=int); vlen may be fractional (30 * fps of a cut DiDeMo video): only linspace truncates."""
    return np.linspace(0, vlen - 2, num_frames, dtype=int)
