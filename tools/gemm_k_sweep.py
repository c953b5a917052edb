"""Per-tile fixed overhead of the GEMM kernel: time vs K at fixed M x N (tile time = t0 + (K/64) * ts)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from blim_amd import engine as eng
M, N = 32768, 37888
rounds = (M // 256) * (N // 256) / 256
res = []
for K in (64, 128, 256, 512, 1024, 2048, 3584, 7168):
    a = torch.empty((M, K), dtype=torch.bfloat16, device="cuda"); w = torch.empty((N, K), dtype=torch.bfloat16, device="cuda")
    eng.fill_bell_bf16(a, 1, "a", 1.0); eng.fill_bell_bf16(w, 1, "w", 0.02)
    eng.gemm_bf16(a, w); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        eng.gemm_bf16(a, w)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 3
    print(f"K={K:5d} steps={K//64:4d}: {ms:.3f} ms  tile {ms*1e3/rounds:.2f} us  {2.0*M*N*K/ms/1e9:.0f} TFLOP/s", flush=True)
    res.append((K // 64, ms * 1e3 / rounds))
    del a, w
(n1, t1), (n2, t2) = res[-2], res[-1]
ts = (t2 - t1) / (n2 - n1)
print(f"ts = {ts:.3f} us/step, t0 = {t1 - n1 * ts:.2f} us  (asymptote {2*256*256*64/ts*256/1e6:.0f} TFLOP/s)")
