"""Throughput of the offline feature extraction (blim_amd/vision.py): V videos (16 frames of 448 x 448 each) per engine call.
    python tools/vision_bench.py [videos per call ...]        (under rocprofv3 --kernel-trace --stats for the per-kernel split)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from blim_amd import vision as V
enc = V.VisionEncoder(V.VisionDims(), dtype=os.environ.get("BLIM_DTYPE", "f16"))
enc.init_synthetic_weights(21)
d = enc.dims
flop_tok_layer = 2 * (3 * d.hidden_size ** 2 + d.hidden_size ** 2 + 2 * d.hidden_size * d.mlp_hidden)
for nv in [int(a) for a in sys.argv[1:]] or [1, 4, 8]:
    frames = (torch.randn((nv * 16, 3, d.image_size, d.image_size), device="cuda")).to(enc.torch_dtype)
    enc.encode(frames); torch.cuda.synchronize()
    t0 = time.time(); reps = 3
    for _ in range(reps):
        enc.encode(frames)
    torch.cuda.synchronize()
    dt = (time.time() - t0) / reps
    tok = nv * 4 * d.tokens_per_clip
    gemm = tok * d.depth * flop_tok_layer + tok * 2 * 768 * d.hidden_size
    attn = nv * 4 * d.depth * 4 * d.tokens_per_clip ** 2 * d.hidden_size
    print(f"{nv} video(s) per call: {dt * 1e3:.1f} ms = {nv / dt:.1f} videos/s; {tok} tokens, linear layers {gemm / 1e12:.2f} TFLOP + attention {attn / 1e12:.2f} TFLOP "
          f"-> {(gemm + attn) / dt / 1e12:.0f} TFLOP/s overall", flush=True)
enc.close()
