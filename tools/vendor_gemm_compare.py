"""Reference point, not a product path: the vendor library's GEMM (torch.matmul -> hipBLASLt) on the decoder's shapes next to this
repo's kernel (blim_gemm_f16, plain 16-bit epilogue).  python tools/vendor_gemm_compare.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from blim_amd import engine as E

lib = E.load_library()
M = 32768
shapes = [("qkv", 4608, 3584), ("o_proj", 3584, 3584), ("gate|up", 37888, 3584), ("down", 3584, 18944)]
for dt, fn in ((torch.float16, lib.blim_gemm_f16), (torch.bfloat16, lib.blim_gemm_bf16)):
    for name, N, K in shapes:
        a = torch.randn(M, K, device="cuda", dtype=dt) * 0.05
        w = torch.randn(N, K, device="cuda", dtype=dt) * 0.05
        c = torch.empty(M, N, device="cuda", dtype=dt)
        res = {}
        for who in ("vendor", "ours"):
            def run():
                if who == "vendor":
                    torch.matmul(a, w.t(), out=c)
                else:
                    rc = fn(a.data_ptr(), K, w.data_ptr(), M, N, K, c.data_ptr(), N, torch.cuda.current_stream().cuda_stream)
                    assert rc == 0
            for _ in range(3):
                run()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                run()
            e1.record(); torch.cuda.synchronize()
            res[who] = 2.0 * M * N * K / (e0.elapsed_time(e1) / 10 * 1e-3) / 1e12
        print(f"{str(dt).split('.')[-1]:9s} {name:8s} M={M} N={N} K={K}: vendor {res['vendor']:7.0f} TFLOP/s   ours {res['ours']:7.0f} TFLOP/s", flush=True)
