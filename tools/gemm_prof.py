"""Runs the plain bf16 GEMM on the decoder's shapes a few times (profiling target for rocprofv3)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from blim_amd import engine as eng  # noqa: E402

if os.environ.get("BLIM_LIB"):          # an instrumented / ablation build (tools/bin/*.so) instead of the product library
    eng.load_library(os.path.join(ROOT, os.environ["BLIM_LIB"]))

shapes = [(32768, 37888, 3584), (32768, 3584, 18944), (32768, 3584, 3584), (32768, 4608, 3584), (8192, 8192, 8192)]
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
for (M, N, K) in shapes:
    a = torch.empty((M, K), dtype=torch.bfloat16, device="cuda"); w = torch.empty((N, K), dtype=torch.bfloat16, device="cuda")
    eng.fill_bell_bf16(a, 1, "a", 1.0); eng.fill_bell_bf16(w, 1, "w", 0.02)
    mode = os.environ.get("BLIM_DTYPE", "f16")
    if mode in ("f16", "f8"):
        a = a.to(torch.float16); w = w.to(torch.float16)
    if mode == "f8":                                    # e4m3 operands, per-row scales, block-scaled MFMA (K-step 128)
        (a8, sa), (w8, sw) = eng.quant_rows(a), eng.quant_rows(w)
        del a, w
        run = lambda: eng.gemm_f8(a8, sa, w8, sw)
    else:
        run = lambda: eng.gemm_bf16(a, w)
    run()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        run()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print(f"gemm {M}x{N}x{K}: {ms:.3f} ms  {2.0 * M * N * K / ms / 1e9:.1f} TFLOP/s", flush=True)
    del run
