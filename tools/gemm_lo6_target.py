"""Profiling target: the compensated fp16 GEMM (16-bit pass + e2m3 pass, gemm.hip) on the gate-up and down shapes, a few launches (tools/gemm_lo6_pmc.sh)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from blim_amd import engine as eng
for (M, N, K) in ((32768, 37888, 3584), (32768, 3584, 18944)):
    a = torch.empty((M, 2 * K), dtype=torch.float16, device="cuda"); w = torch.empty((N, K), dtype=torch.float16, device="cuda")
    a.normal_(); a[:, K:] *= 2.0 ** -11; w.normal_(std=0.02)
    for _ in range(3):
        eng.gemm_f16_lo6(a, w)
    torch.cuda.synchronize()
    del a, w
