"""Where a wave's time goes inside one K-step of the GEMM main loop (instrumented library: `make -C blim_amd/csrc waitprof`).

Group 0 (waves 0-3): [64 MFMA] [vmcnt(0): W(kt+1) landed] [barrier] [24 fragment reads + 8 LDS-DMA] [barrier]
Group 1 (waves 4-7): [24 fragment reads + 8 LDS-DMA] [vmcnt(8): A(kt+1) landed] [barrier] [64 MFMA] [barrier]
Numbers are s_memtime ticks summed over the K loop of each tile, averaged over tiles, shown as % of the loop.
"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from blim_amd import engine as eng
lib = eng.load_library(os.path.join(ROOT, "tools", "bin", "libblim_hip_waitprof.so"))
names = (("mfma", "vmcnt", "barrier1", "reads+dma", "barrier2"), ("reads+dma", "vmcnt", "barrier1", "mfma", "barrier2"))
for (M, N, K) in ((32768, 37888, 3584), (32768, 3584, 18944), (32768, 3584, 3584)):
    a = torch.empty((M, K), dtype=torch.bfloat16, device="cuda"); w = torch.empty((N, K), dtype=torch.bfloat16, device="cuda")
    eng.fill_bell_bf16(a, 1, "a", 1.0); eng.fill_bell_bf16(w, 1, "w", 0.02)
    nwg = (M // 256) * (N // 256)
    st = torch.zeros((nwg * 3, 8), dtype=torch.int64, device="cuda")
    eng.gemm_bf16(a, w); torch.cuda.synchronize()
    lib.blim_debug_gemm_stamps(st.data_ptr())
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record(); eng.gemm_bf16(a, w); e1.record(); torch.cuda.synchronize()
    lib.blim_debug_gemm_stamps(None)
    ms = e0.elapsed_time(e1)
    s = st.cpu().numpy().astype(np.float64)
    loop_us = (s[:nwg, 2] - s[:nwg, 1]).mean() * 0.01
    wt = s[nwg:].reshape(nwg, 2, 8)[:, :, :5]
    print(f"{M}x{N}x{K}: {ms:.3f} ms ({2.0 * M * N * K / ms / 1e9:.0f} TFLOP/s instrumented); main loop {loop_us:.1f} us per tile, {loop_us / (K // 64) * 1e3:.0f} ns per K-step")
    for g in range(2):
        tot = wt[:, g].sum(axis=1).mean()
        parts = wt[:, g].mean(axis=0)
        print(f"   group {g}: ticks/K-step {tot / (K // 64):.0f} (tick = {loop_us * 1e3 / tot:.2f} ns) | " + "  ".join(f"{n} {100 * v / tot:.1f}%" for n, v in zip(names[g], parts)))
    del a, w
