"""Development aid: NaN / inf operands through the fp16 and bf16 GEMM (the finding behind csrc/common.hpp f16_saturate_on: with MODE.FP16_OVFL set during
the MFMAs a NaN operand reads as 0; the product sets the bit around the conversions only)."""
import sys, os
sys.path.insert(0, "/root/repo")
import torch, numpy as np
from blim_amd import engine as eng
for dt in (torch.float16, torch.bfloat16):
    M, N, K = 256, 256, 64
    a = torch.zeros((M, K), dtype=dt, device="cuda"); w = torch.zeros((N, K), dtype=dt, device="cuda")
    a[:, 0] = 1000.0
    w[0, 0] = 1000.0; w[4, 1] = 1.0; w[5, 0] = 1.0
    a[5, 1] = float("inf"); a[6, 1] = float("nan"); a[7, 0] = float("nan"); a[8, :] = float("nan")
    out = eng.gemm_bf16(a, w).float().cpu().numpy()
    print(dt, "rows 5..8, cols 0,4,5,6:\n", out[5:9][:, [0, 4, 5, 6]])
