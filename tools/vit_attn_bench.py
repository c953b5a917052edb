"""The vision encoder's attention kernel alone (blim_vit_attention): time per call at the extraction's shape (8 videos = 32 clips of 3,136 tokens, 16 heads
of 64) and the worst deviation from torch's f32 softmax attention on a small case.   python tools/vit_attn_bench.py [clips] [L]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from blim_amd import vision as V

def ref(qkv, n_clips, heads):
    L = qkv.shape[0] // n_clips
    q, k, v = qkv.float().reshape(n_clips, L, 3, heads, 64).permute(2, 0, 3, 1, 4)
    p = torch.softmax(q @ k.transpose(-1, -2) * 0.125, dim=-1)
    return (p @ v).permute(0, 2, 1, 3).reshape(n_clips * L, heads * 64)

for dt in (torch.float16, torch.bfloat16):
    for L in (100, 3136 // 4 + 5):
        x = (torch.randn((2 * L, 3 * 16 * 64), device="cuda") * 1.5).to(dt)
        got = V.vit_attention(x, 2, 16).float(); want = ref(x, 2, 16)
        print(f"{dt} L={L}: max |diff| {float((got - want).abs().max()):.2e} (max |out| {float(want.abs().max()):.2f})", flush=True)
clips = int(sys.argv[1]) if len(sys.argv) > 1 else 32
L = int(sys.argv[2]) if len(sys.argv) > 2 else 3136
x = torch.randn((clips * L, 3 * 16 * 64), device="cuda").to(torch.float16)
for _ in range(3):
    V.vit_attention(x, clips, 16)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
reps = 20
e0.record()
for _ in range(reps):
    V.vit_attention(x, clips, 16)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / reps
fl = clips * 16 * 4 * L * L * 64
print(f"{clips} clips x {L} tokens x 16 heads: {ms:.3f} ms per call = {fl / ms / 1e9:.0f} TFLOP/s", flush=True)
