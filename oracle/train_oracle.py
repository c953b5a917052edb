"""TEST INFRASTRUCTURE ONLY -- CPU restatement (torch fp32 + autograd) of the reference's training step, SURVEY.md section 8f-4.

Pinned by tests/golden/train_*.npz (oracle/gen_golden_train.py: the reference's own model under autograd).  Nothing under blim_amd/
imports this file; tests/ use it as the checker for the HIP trainer at sizes the fixtures do not cover.

Restates training_utils.py:57-83 on one batch:
    vtg_loss = CE(lm_head(decoder(vtg rows)), shifted labels), mean over all label tokens of the batch        (:23-32, :67-68)
    tvg_loss = CE(forward_visual(hidden at the 4 positions before <|im_end|>) . video_vocab / sqrt(M)), mean over B x 4  (:71-79)
    loss = vtg_loss + tvg_loss                                                                                (:81)
with LoRA (peft forward: base(x) + B A x * alpha / r, dropout 0) on the modules of main.py:96-101, and the AdamW update of
main.py:147 (decoupled weight decay, betas (0.9, 0.95), eps 1e-8, bias-corrected) written out in numpy.
Rows are processed one sequence at a time: with causal attention and right padding the padded batch of the reference computes
the same numbers for the real tokens (checked by the fixtures).
"""
from __future__ import annotations

import math
from typing import Dict, List, Sequence

import numpy as np

IMAGE_TOKEN_INDEX = -200
IGNORE_INDEX = -100
IM_END = 151645


def _rope(x, cos, sin):
    import torch
    h = x.shape[-1] // 2
    rot = torch.cat([-x[..., h:], x[..., :h]], dim=-1)                    # modeling_qwen2_flash.py:139-143
    return x * cos + rot * sin


def drop_mult(seed: int, site: int, n_rows: int, K: int, row0: int, p: float) -> np.ndarray:
    """The engine's counter-based LoRA-input dropout mask (csrc/train_kernels.hip: drop_hash / drop_mult4), restated: multiplier 1/(1-p) or 0
    for element (row0 + i, k) of adapter `site` under step seed `seed`.  One 64-bit hash per group of four consecutive elements, 16 bits each;
    an element is dropped when its 16 bits are below floor(65536 p)."""
    u64 = np.uint64
    idx = (np.arange(row0, row0 + n_rows, dtype=np.uint64)[:, None] * u64(K) + np.arange(K, dtype=np.uint64)[None, :])
    with np.errstate(over="ignore"):
        z = u64(seed & 0xFFFFFFFFFFFFFFFF) + u64(0x9E3779B97F4A7C15) * u64(site + 1) + (idx >> u64(2)) * u64(0xD1B54A32D192ED03)
        z = (z ^ (z >> u64(30))) * u64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> u64(27))) * u64(0x94D049BB133111EB)
        z = z ^ (z >> u64(31))
    bits = (z >> (u64(16) * (idx & u64(3)))) & u64(0xFFFF)
    thr = np.uint64(int(np.float32(p) * np.float32(65536.0)))
    return np.where(bits >= thr, np.float32(1.0) / (np.float32(1.0) - np.float32(p)), np.float32(0.0)).astype(np.float32)


class TrainOracle:
    def __init__(self, cfg, weights: Dict[str, np.ndarray], trainable: Dict[str, np.ndarray], r: int, alpha: float, drop_p: float = 0.0):
        import torch
        self.cfg, self.r, self.scaling = cfg, r, alpha / r
        self.drop_p, self.seed = float(drop_p), 0          # peft: dropout on the adapter's input only (lora_dropout)
        self.w = {k: torch.from_numpy(np.asarray(v, np.float32)) for k, v in weights.items() if k != "visual_head"}
        self.p = {k: torch.from_numpy(np.asarray(v, np.float32).copy()).requires_grad_(True) for k, v in trainable.items()}
        self.tvg_prefix_length = 0

    # LoRA Linear: peft forward with dropout 0
    def lin(self, x, wname: str, bias: str = None, site: int = None, row0: int = 0):
        import torch
        import torch.nn.functional as F
        y = F.linear(x, self.w[wname], self.w[bias] if bias else None)
        xd = x
        if self.drop_p > 0.0 and site is not None:         # the engine's mask for (adapter site, packed row, column)
            x2 = x.reshape(-1, x.shape[-1])
            xd = (x2 * torch.from_numpy(drop_mult(self.seed, site, x2.shape[0], x2.shape[1], row0, self.drop_p))).reshape(x.shape)
        return y + F.linear(F.linear(xd, self.p[wname + ":A"]), self.p[wname + ":B"]) * self.scaling

    def project(self, feat, tvg: bool, row0: int = 0):
        import torch.nn.functional as F
        p, w = ("tvg_mlp", 1) if tvg else ("mlp", 0)
        h = F.gelu(self.lin(feat, f"{p}.0.w", f"{p}.0.b", 1000 + 2 * w, row0))      # mm_projector_builder.py:88-93 (exact erf GELU)
        return self.lin(h, f"{p}.2.w", f"{p}.2.b", 1001 + 2 * w, row0)

    def embeds_of(self, ids: np.ndarray, video: np.ndarray, tvg: bool, feat_row0: int = 0):
        """One row: token embeddings with the projected video spliced in at <image> (modeling_videochat_flash.py:395-444)."""
        import torch
        feat = self.project(torch.from_numpy(np.asarray(video, np.float32)), tvg, feat_row0)
        feat = feat.mean(dim=1) if tvg else feat.reshape(-1, feat.shape[-1])     # :243
        where = int(np.nonzero(ids == IMAGE_TOKEN_INDEX)[0][0])
        E = self.w["embed_tokens"]
        return torch.cat([E[torch.from_numpy(ids[:where])], feat, E[torch.from_numpy(ids[where + 1:])]], dim=0), where, feat.shape[0]

    def decoder(self, x, row0: int = 0):
        """[L, H] -> final-norm hidden [L, H]; causal attention over the row (modeling_qwen2_flash.py:742-800, 247-326).
        row0: index of the row's first token in the packed batch (only the dropout mask depends on it)."""
        import torch
        import torch.nn.functional as F
        c = self.cfg
        L = x.shape[0]
        hd, nh, nkv = c.head_dim, c.num_heads, c.num_kv_heads
        inv = 1.0 / (c.rope_theta ** (torch.arange(0, hd, 2, dtype=torch.float32) / hd))
        ang = torch.arange(L, dtype=torch.float32)[:, None] * inv[None, :]
        emb = torch.cat([ang, ang], dim=-1)
        cos, sin = emb.cos()[:, None, :], emb.sin()[:, None, :]
        causal = torch.full((L, L), float("-inf")).triu(1)

        def norm(t, w):
            v = t.pow(2).mean(-1, keepdim=True)
            return w * (t * torch.rsqrt(v + c.rms_eps))
        for i in range(c.num_layers):
            P = f"layers.{i}."
            h = norm(x, self.w[P + "input_norm"])
            q = _rope(self.lin(h, P + "q_proj.w", P + "q_proj.b", 8 * i, row0).reshape(L, nh, hd), cos, sin)
            k = _rope(self.lin(h, P + "k_proj.w", P + "k_proj.b", 8 * i + 1, row0).reshape(L, nkv, hd), cos, sin)
            v = self.lin(h, P + "v_proj.w", P + "v_proj.b", 8 * i + 2, row0).reshape(L, nkv, hd)
            k = k.repeat_interleave(nh // nkv, dim=1); v = v.repeat_interleave(nh // nkv, dim=1)
            s = torch.einsum("qhd,khd->hqk", q, k) / math.sqrt(hd) + causal
            a = torch.einsum("hqk,khd->qhd", torch.softmax(s, dim=-1), v).reshape(L, nh * hd)
            x = x + self.lin(a, P + "o_proj.w", None, 8 * i + 3, row0)
            h = norm(x, self.w[P + "post_norm"])
            x = x + F.linear(F.silu(F.linear(h, self.w[P + "gate_proj.w"])) * F.linear(h, self.w[P + "up_proj.w"]), self.w[P + "down_proj.w"])
        return norm(x, self.w["final_norm"])

    def losses(self, vtg_ids: Sequence[np.ndarray], vtg_labels: Sequence[np.ndarray], tvg_ids: Sequence[np.ndarray], tvg_labels: Sequence[np.ndarray],
               videos: Sequence[np.ndarray], video_vocab: np.ndarray, tvg_video_labels: np.ndarray):
        import torch
        import torch.nn.functional as F
        C = self.cfg.num_clips
        nll, cnt = [], 0
        tok0 = frow0 = 0
        for ids, lab, vid in zip(vtg_ids, vtg_labels, videos):
            x, where, nv = self.embeds_of(np.asarray(ids), vid, False, frow0)
            lab = np.concatenate([lab[:where], np.full(nv, IGNORE_INDEX, np.int64), lab[where + 1:]])
            h = self.decoder(x, tok0)
            pos = np.nonzero(lab[1:] != IGNORE_INDEX)[0]
            logits = self.lin(h[torch.from_numpy(pos)], "lm_head", None, 2000, cnt)
            nll.append(F.cross_entropy(logits, torch.from_numpy(lab[1:][pos]), reduction="sum"))
            cnt += len(pos); tok0 += x.shape[0]; frow0 += int(np.prod(np.asarray(vid).shape[:2]))
        vtg_loss = torch.stack(nll).sum() / cnt
        vocab = torch.from_numpy(np.asarray(video_vocab, np.float32))
        rows = []
        frow0 = 0                                          # one packed batch: the TVG rows' tokens are numbered after the VTG rows' (tok0 continues)
        for ids, lab, vid, vl in zip(tvg_ids, tvg_labels, videos, tvg_video_labels):
            x, where, nv = self.embeds_of(np.asarray(ids), vid, True, frow0)
            lab = np.concatenate([lab[:where], np.full(nv, IGNORE_INDEX, np.int64), lab[where + 1:]])
            h = self.decoder(x, tok0)
            tok0 += x.shape[0]; frow0 += int(np.prod(np.asarray(vid).shape[:2]))
            p = int(np.nonzero(lab == IM_END)[0][0])
            hv = F.linear(h[p - (C + 1): p - 1], self.p["visual_head"])                 # [C, M]
            lg = torch.einsum("cm,ncm->cn", hv, vocab) / math.sqrt(vocab.shape[-1])       # training_utils.py:78
            rows.append(F.cross_entropy(lg, torch.full((C,), int(vl)), reduction="sum"))
        tvg_loss = torch.stack(rows).sum() / (C * len(rows))
        return vtg_loss, tvg_loss

    def step_grads(self, *batch):
        """(vtg_loss, tvg_loss, {name: grad}) of loss = vtg_loss + tvg_loss."""
        for p in self.p.values():
            p.grad = None
        a, b = self.losses(*batch)
        (a + b).backward()
        return float(a.detach()), float(b.detach()), {k: v.grad.detach().numpy().copy() for k, v in self.p.items()}


class AdamW:
    """torch.optim.AdamW written out (decoupled decay first, then the bias-corrected Adam update)."""

    def __init__(self, params: Dict[str, np.ndarray], lr: float, wd: float, betas=(0.9, 0.95), eps: float = 1e-8):
        self.lr, self.wd, self.b1, self.b2, self.eps, self.t = lr, wd, betas[0], betas[1], eps, 0
        self.m = {k: np.zeros_like(v, dtype=np.float32) for k, v in params.items()}
        self.v = {k: np.zeros_like(v, dtype=np.float32) for k, v in params.items()}

    def step(self, params: Dict[str, np.ndarray], grads: Dict[str, np.ndarray], lr: float = None) -> None:
        lr = self.lr if lr is None else lr
        self.t += 1
        c1, c2 = 1.0 - self.b1 ** self.t, 1.0 - self.b2 ** self.t
        for k, p in params.items():
            g = grads[k].astype(np.float32)
            wd = self.wd if p.ndim > 1 else 0.0                      # timm param_groups_weight_decay: no decay on 1-D tensors
            p *= np.float32(1.0 - lr * wd)
            self.m[k] = self.b1 * self.m[k] + (1 - self.b1) * g
            self.v[k] = self.b2 * self.v[k] + (1 - self.b2) * g * g
            p -= (lr / c1) * self.m[k] / (np.sqrt(self.v[k]) / math.sqrt(c2) + self.eps)


def cosine_lr(epoch: float, lr: float, min_lr: float, warmup_epochs: float, epochs: float) -> float:
    """util/lr_sched.py:9-21."""
    if epoch < warmup_epochs:
        return lr * epoch / warmup_epochs
    return min_lr + (lr - min_lr) * 0.5 * (1.0 + math.cos(math.pi * (epoch - warmup_epochs) / (epochs - warmup_epochs)))
