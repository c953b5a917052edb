"""Where a wave's time goes inside one K-step of the compensated GEMM's SECOND pass (gemm.hip phase 2; instrumented library: `make -C blim_amd/csrc waitprof`).

W group (waves 0-3): [26 fragment / scale reads] [barrier] [7 LDS-DMA + 32 e2m3 MFMA] [vmcnt] [barrier]      A group (waves 4-7): the same, half a step later.
Numbers are s_memtime ticks summed over the pass's K loop of each tile, averaged over tiles."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["BLIM_LIB_PATH"] = os.path.join(ROOT, "tools", "bin", "libblim_hip_waitprof.so")
import numpy as np, torch
from blim_amd import engine as eng
lib = eng.load_library()
names = ("reads", "barrier1", "dma+mfma", "vmcnt", "barrier2")
for (M, N, K) in ((32768, 37888, 3584), (32768, 3584, 18944), (32768, 3584, 3584)):
    a = torch.empty((M, 2 * K), dtype=torch.float16, device="cuda"); w = torch.empty((N, K), dtype=torch.float16, device="cuda")
    a.normal_(); a[:, K:] *= 2.0 ** -11; w.normal_(std=0.02)
    nwg = (M // 256) * (N // 256)
    st = torch.zeros((nwg * 3, 8), dtype=torch.int64, device="cuda")
    eng.gemm_f16_lo6(a, w); torch.cuda.synchronize()
    lib.blim_debug_gemm_stamps(st.data_ptr())
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record(); eng.gemm_f16_lo6(a, w); e1.record(); torch.cuda.synchronize()
    lib.blim_debug_gemm_stamps(None)
    s = st.cpu().numpy().astype(np.float64)
    loop_us = (s[:nwg, 2] - s[:nwg, 1]).mean() * 0.01
    wt = s[nwg:].reshape(nwg, 2, 8)[:, :, :5]
    print(f"{M}x{N}x{K}: both passes {loop_us:.1f} us per tile ({K // 64} 16-bit + {K // 128} e2m3 K-steps)")
    for g in range(2):
        tot = wt[:, g].sum(axis=1).mean()
        parts = wt[:, g].mean(axis=0)
        print(f"   group {g}: ticks per e2m3 K-step {tot / (K // 128):.0f} (x 10 ns = {tot / (K // 128) * 0.01:.2f} us) | " + "  ".join(f"{n} {100 * v / tot:.1f}% ({v / (K // 128) * 10:.0f} ns)" for n, v in zip(names, parts)))
    del a, w
