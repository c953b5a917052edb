"""Synthetic weights / inputs for the scoring path (host side, numpy).

There is no network for checkpoints or datasets, so every benchmark and parity input is generated
from a counter-based rule that is bit-exact on any machine and that the HIP engine reproduces on
device (csrc/synth.hip: `blim_fill_bell_bf16`):

    x   = seed*K0 + tensor_id*K1 + index                (mod 2^64)
    z   = splitmix64_finalise(x + K0)
    s   = sum of the four 16-bit fields of z            (0 .. 262140)
    val = f32(s - 131070) * f32(std / SIGMA4) + f32(mean)      then RNE to bf16

which is an Irwin-Hall(4) bell curve of exactly the requested std, standing in for the reference's
N(0, initializer_range^2) init (videochat_flash/modeling_qwen2_flash.py:835-843).
tensor_id = FNV-1a-64 of the canonical tensor name (names: `weight_shapes`).

The token layouts follow the reference's dataset front end (dataloader/base_dataset.py:60-105):
VTG row  = [system+user header][<image>][instruction][assistant header][text][<|im_end|>][\\n]
TVG row  = [system+user header+instruction = tvg prefix]["\\nCaption: "+text][<|im_end|>\\n][assistant header][<image>][<|im_end|>][\\n]
with labels = -100 on the prompt and the response copied from the ids.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, List, Tuple

import numpy as np

IGNORE_INDEX = -100
IMAGE_TOKEN_INDEX = -200
IM_START, IM_END, NEWLINE, PAD_ID = 151644, 151645, 198, 151643

K0 = np.uint64(0x9E3779B97F4A7C15)
K1 = np.uint64(0xBF58476D1CE4E5B9)
K2 = np.uint64(0x94D049BB133111EB)
SIGMA4 = float(np.sqrt(4.0 * (65536.0 ** 2 - 1.0) / 12.0))


def fnv1a64(name: str) -> int:
    h = 0xCBF29CE484222325
    for b in name.encode("utf-8"):
        h = ((h ^ b) * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
    return h


def _hash(seed: int, tid: int, idx: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        z = np.uint64(seed) * K0 + np.uint64(tid) * K1 + idx.astype(np.uint64) + K0
        z = (z ^ (z >> np.uint64(30))) * K1
        z = (z ^ (z >> np.uint64(27))) * K2
        return z ^ (z >> np.uint64(31))


def bell_scale(std: float) -> np.float32:
    return np.float32(std / SIGMA4)


def bell_f32(seed: int, name: str, n: int, std: float, mean: float = 0.0) -> np.ndarray:
    tid = fnv1a64(name)
    out = np.empty(n, dtype=np.float32)
    sc, mu = bell_scale(std), np.float32(mean)
    step = 1 << 24
    for lo in range(0, n, step):
        z = _hash(seed, tid, np.arange(lo, min(n, lo + step), dtype=np.uint64))
        s = ((z & np.uint64(0xFFFF)) + ((z >> np.uint64(16)) & np.uint64(0xFFFF))
             + ((z >> np.uint64(32)) & np.uint64(0xFFFF)) + (z >> np.uint64(48))).astype(np.int64)
        out[lo:lo + len(s)] = (s - 131070).astype(np.float32) * sc + mu
    return out


def bf16_bits(x: np.ndarray) -> np.ndarray:
    u = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32)
    return ((u + np.uint32(0x7FFF) + ((u >> np.uint32(16)) & np.uint32(1))) >> np.uint32(16)).astype(np.uint16)


def bf16_round(x: np.ndarray) -> np.ndarray:
    return (bf16_bits(x).astype(np.uint32) << np.uint32(16)).view(np.float32)


def tensor(seed: int, name: str, shape, std: float, mean: float = 0.0) -> np.ndarray:
    """bf16-representable float32 array."""
    return bf16_round(bell_f32(seed, name, int(np.prod(shape)), std, mean)).reshape(shape)


def uniform_ids(seed: int, name: str, n: int, lo: int, hi: int) -> np.ndarray:
    z = _hash(seed, fnv1a64(name), np.arange(n, dtype=np.uint64))
    return np.int64(lo) + (z % np.uint64(hi - lo)).astype(np.int64)


# ----------------------------------------------------------------------------- weights

@dataclass
class ModelDims:
    """Qwen2-7B dims by default (checkpoint config.json values)."""
    vocab_size: int = 152064
    hidden_size: int = 3584
    intermediate_size: int = 18944
    num_layers: int = 28
    num_heads: int = 28
    num_kv_heads: int = 4
    rms_eps: float = 1e-6
    rope_theta: float = 1e6
    mm_hidden_size: int = 1024
    num_clips: int = 4

    @property
    def head_dim(self) -> int:
        return self.hidden_size // self.num_heads


def weight_shapes(d: ModelDims) -> Dict[str, Tuple[int, ...]]:
    H, I, V, M, hd = d.hidden_size, d.intermediate_size, d.vocab_size, d.mm_hidden_size, d.head_dim
    s: Dict[str, Tuple[int, ...]] = {"embed_tokens": (V, H), "final_norm": (H,), "lm_head": (V, H), "visual_head": (M, H)}
    for p in ("mlp", "tvg_mlp"):
        s[f"{p}.0.w"] = (H, M); s[f"{p}.0.b"] = (H,); s[f"{p}.2.w"] = (H, H); s[f"{p}.2.b"] = (H,)
    for i in range(d.num_layers):
        L = f"layers.{i}."
        s[L + "input_norm"] = (H,); s[L + "post_norm"] = (H,)
        s[L + "q_proj.w"] = (d.num_heads * hd, H); s[L + "q_proj.b"] = (d.num_heads * hd,)
        s[L + "k_proj.w"] = (d.num_kv_heads * hd, H); s[L + "k_proj.b"] = (d.num_kv_heads * hd,)
        s[L + "v_proj.w"] = (d.num_kv_heads * hd, H); s[L + "v_proj.b"] = (d.num_kv_heads * hd,)
        s[L + "o_proj.w"] = (H, H)
        s[L + "gate_proj.w"] = (I, H); s[L + "up_proj.w"] = (I, H); s[L + "down_proj.w"] = (H, I)
    return s


def weight_dist(name: str) -> Tuple[float, float]:
    """(std, mean) of a synthetic tensor: norm weights bell(1, 0.1), everything else bell(0, 0.02)."""
    return (0.1, 1.0) if name.endswith("norm") else (0.02, 0.0)


def synthetic_weights(d: ModelDims, seed: int) -> Dict[str, np.ndarray]:
    return {n: tensor(seed, n, s, *weight_dist(n)) for n, s in weight_shapes(d).items()}


# ----------------------------------------------------------------------------- inputs

@dataclass
class Problem:
    """One synthetic retrieval test set (N videos == N texts, ground truth = identity)."""
    video: List[np.ndarray]            # N x [clips, tok_per_clip, mm_hidden] bf16-representable f32
    video_vocab: np.ndarray            # [N, clips, mm_hidden]  (clip means, base_dataset.py:33-37)
    vtg_ids: List[np.ndarray]
    vtg_labels: List[np.ndarray]
    vtg_masks: List[np.ndarray]
    tvg_ids: List[np.ndarray]
    tvg_labels: List[np.ndarray]
    tvg_masks: List[np.ndarray]
    tvg_video_labels: np.ndarray       # [N]
    tvg_prefix_length: int
    v2t_sims: np.ndarray               # [N, N] first-stage scores (InternVideo2 stand-in)
    t2v_sims: np.ndarray


def make_problem(seed: int, n: int, dims: ModelDims, tok_per_clip: int = 64, text_len=(8, 48),
                 reference_layout: bool = True, fast_video: bool = False) -> Problem:
    """reference_layout=True: rows shaped like base_dataset.py:60-105 (system/user headers, instruction ...).
    reference_layout=False: BASELINE.json's headline shape -- VTG row = [<image>][text] only (the row is
    clips*tok_per_clip video tokens + len(text) label tokens), TVG row = [21-id prefix][text][<image>][tail]."""
    M, C = dims.mm_hidden_size, dims.num_clips
    word = lambda nm, k: uniform_ids(seed, nm, k, 1000, 150000)
    sys_hdr = np.concatenate([[IM_START], word("sys", 1), [NEWLINE], word("sys_text", 6), [IM_END, NEWLINE]])   # 11 ids
    usr_hdr = np.concatenate([[IM_START], word("usr", 1), [NEWLINE]])                                             # 3 ids
    asst_hdr = np.concatenate([[IM_START], word("asst", 1), [NEWLINE]])
    vtg_instr = np.concatenate([[NEWLINE], word("vtg_instr", 6)])
    tvg_instr = word("tvg_instr", 7)
    cap_hdr = np.concatenate([[NEWLINE], word("cap_hdr", 2)])
    lo, hi = text_len
    lens = lo + (uniform_ids(seed, "text_len", n, 0, max(1, hi - lo + 1)))
    video, vtg_ids, vtg_labels, tvg_ids, tvg_labels = [], [], [], [], []
    rng = np.random.default_rng(seed) if fast_video else None   # large dry runs (thousands of videos): numpy's generator instead of the
    for i in range(n):                                          # counter-based rule (not reproduced on device, not used by any fixture)
        video.append(bf16_round(rng.standard_normal((C, tok_per_clip, M), dtype=np.float32)) if fast_video
                     else tensor(seed, f"video.{i}", (C, tok_per_clip, M), std=1.0))
        text = word(f"text.{i}", int(lens[i]))
        resp = np.concatenate([text, [IM_END, NEWLINE]])
        if reference_layout:
            prompt = np.concatenate([sys_hdr, usr_hdr, [IMAGE_TOKEN_INDEX], vtg_instr, [IM_END, NEWLINE], asst_hdr])
        else:
            prompt = np.array([IMAGE_TOKEN_INDEX], dtype=np.int64)
            resp = text
        ids = np.concatenate([prompt, resp]).astype(np.int64)
        lab = ids.copy(); lab[: len(prompt)] = IGNORE_INDEX
        vtg_ids.append(ids); vtg_labels.append(lab)
        tprompt = np.concatenate([sys_hdr, usr_hdr, tvg_instr, cap_hdr, text, [IM_END, NEWLINE], asst_hdr]) if reference_layout else \
            np.concatenate([sys_hdr, usr_hdr, tvg_instr, text[:7]])
        tresp = np.array([IMAGE_TOKEN_INDEX, IM_END, NEWLINE], dtype=np.int64)
        tids = np.concatenate([tprompt, tresp]).astype(np.int64)
        tlab = tids.copy(); tlab[: len(tprompt)] = IGNORE_INDEX
        tvg_ids.append(tids); tvg_labels.append(tlab)
    vocab = np.stack([v.mean(axis=1, dtype=np.float32) for v in video]).astype(np.float32)
    sims = bell_f32(seed, "sims.v2t", n * n, 1.0).reshape(n, n) + 3.0 * np.eye(n, dtype=np.float32)
    sims_t = bell_f32(seed, "sims.t2v", n * n, 1.0).reshape(n, n) + 3.0 * np.eye(n, dtype=np.float32)
    ones = lambda rows: [np.ones(len(r), dtype=np.int64) for r in rows]
    return Problem(video=video, video_vocab=vocab, vtg_ids=vtg_ids, vtg_labels=vtg_labels, vtg_masks=ones(vtg_ids),
                   tvg_ids=tvg_ids, tvg_labels=tvg_labels, tvg_masks=ones(tvg_ids),
                   tvg_video_labels=np.arange(n, dtype=np.int64), tvg_prefix_length=len(sys_hdr) + len(usr_hdr) + len(tvg_instr),
                   v2t_sims=sims.astype(np.float32), t2v_sims=sims_t.astype(np.float32))


class ProblemLoader:
    """Iterates a synthetic Problem in the reference's eval collate format (dataloader/base_dataset.py:119-163: lists of 1-D
    tensors per batch) and carries the two dataset attributes evaluation() reads (video_vocab, tvg_prefix_length)."""

    class _Dataset:
        def __init__(self, prob):
            import torch
            self.video_vocab = torch.from_numpy(prob.video_vocab)
            self.tvg_prefix_length = prob.tvg_prefix_length
            self._n = len(prob.video)

        def __len__(self):
            return self._n

    def __init__(self, prob: Problem, batch_size: int, video_dtype=None):
        """video_dtype: dtype of the per-video feature tensors the loader hands out (converted once, here).  The reference's feature
        files hold fp16 (extract.py:108 `.cpu().half()`); None keeps the generator's fp32."""
        import torch
        self.prob, self.bs = prob, int(batch_size)
        self.dataset = ProblemLoader._Dataset(prob)
        self._video = [torch.from_numpy(v) if video_dtype is None else torch.from_numpy(v).to(video_dtype) for v in prob.video]

    def __len__(self):
        return (len(self.prob.video) + self.bs - 1) // self.bs

    def __iter__(self):
        import torch
        p, T = self.prob, torch.from_numpy
        for s in range(0, len(p.video), self.bs):
            e = min(len(p.video), s + self.bs)
            out = {"video": self._video[s:e], "tvg_video_labels": T(p.tvg_video_labels[s:e])}
            for k in ("vtg_ids", "vtg_labels", "vtg_masks", "tvg_ids", "tvg_labels", "tvg_masks"):
                out[k] = [T(x) for x in getattr(p, k)[s:e]]
            yield out
