"""TEST INFRASTRUCTURE ONLY -- torch-CPU (fp32, threaded) statement of the per-layer arithmetic of oracle/blim_oracle.py.

Exists for ONE purpose: bench.py's `cpu_baseline` leg needs a CPU timing that uses the host's cores properly (numpy's BLAS
oversubscribes a 256-thread host; torch's intra-op pool is sized explicitly here).  It follows the same reference lines as
blim_oracle.OracleModel.decoder_layer (modeling_qwen2_flash.py:742-800, eager attention :247-326, RMSNorm :93-98, RoPE
:139-172, MLP :187-188) and is checked against it in tests/test_oracle_golden.py.  Never imported by blim_amd/.
"""
from __future__ import annotations

import math

import torch


def physical_cores() -> int:
    try:
        import psutil
        n = psutil.cpu_count(logical=False)
        if n:
            return int(n)
    except Exception:
        pass
    import os
    return os.cpu_count() or 1


def usable_cores() -> int:
    """Cores this PROCESS may use: physical cores, capped by the scheduler affinity mask and by the cgroup CPU quota (a container
    on a 256-thread host is often given far fewer; 128 threads on such a quota run slower than 8)."""
    import os
    n = physical_cores()
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except Exception:
        pass
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0])
                if q > 0:
                    n = min(n, max(1, q // int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())))
            break
        except Exception:
            continue
    return max(1, n)


def rms_norm(x, w, eps):
    var = x.pow(2).mean(-1, keepdim=True)
    return w * (x * torch.rsqrt(var + eps))


def rope_tables(head_dim, theta, n_pos):
    inv = 1.0 / (theta ** (torch.arange(0, head_dim, 2, dtype=torch.float32) / head_dim))
    fr = torch.outer(torch.arange(n_pos, dtype=torch.float32), inv)
    emb = torch.cat([fr, fr], dim=-1)
    return emb.cos(), emb.sin()


def apply_rope(x, cos, sin):
    d = x.shape[-1]
    rot = torch.cat([-x[..., d // 2:], x[..., : d // 2]], dim=-1)
    return x * cos[None, None] + rot * sin[None, None]


def additive_mask(key_mask, L):
    causal = torch.tril(torch.ones((L, L), dtype=torch.bool))
    vis = causal[None] & key_mask.bool()[:, None, :]
    return torch.where(vis, 0.0, torch.finfo(torch.float32).min)[:, None]


def decoder_layer(x, w, prefix, add_mask, cos, sin, num_heads, num_kv_heads, eps):
    """x [B, L, H] f32; w: {canonical name: tensor}; prefix 'layers.i.'."""
    B, L, H = x.shape
    hd = H // num_heads
    h = rms_norm(x, w[prefix + "input_norm"], eps)
    q = (h @ w[prefix + "q_proj.w"].T + w[prefix + "q_proj.b"]).view(B, L, num_heads, hd).transpose(1, 2)
    k = (h @ w[prefix + "k_proj.w"].T + w[prefix + "k_proj.b"]).view(B, L, num_kv_heads, hd).transpose(1, 2)
    v = (h @ w[prefix + "v_proj.w"].T + w[prefix + "v_proj.b"]).view(B, L, num_kv_heads, hd).transpose(1, 2)
    q, k = apply_rope(q, cos, sin), apply_rope(k, cos, sin)
    rep = num_heads // num_kv_heads
    k = k.repeat_interleave(rep, dim=1); v = v.repeat_interleave(rep, dim=1)
    s = q @ k.transpose(-1, -2) / math.sqrt(hd) + add_mask
    p = torch.softmax(s, dim=-1, dtype=torch.float32)
    a = (p @ v).transpose(1, 2).reshape(B, L, H)
    x = x + a @ w[prefix + "o_proj.w"].T
    h = rms_norm(x, w[prefix + "post_norm"], eps)
    g = torch.nn.functional.silu(h @ w[prefix + "gate_proj.w"].T) * (h @ w[prefix + "up_proj.w"].T)
    return x + g @ w[prefix + "down_proj.w"].T


def label_logprobs(hidden_rows, lm_head, labels):
    """log_softmax(hidden_rows @ lm_head^T)[labels] (modeling_qwen2_flash.py:1452-1453 + retrieval_utils.py:23-31)."""
    lg = hidden_rows @ lm_head.T
    return torch.log_softmax(lg, dim=-1).gather(1, labels[:, None])[:, 0]
