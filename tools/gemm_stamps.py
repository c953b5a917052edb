"""Per-workgroup phase times of the GEMM kernel from in-kernel s_memtime stamps (100 MHz constant clock? no: shader clock)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from blim_amd import engine as eng
lib = eng.load_library()
for (M, N, K) in ((32768, 37888, 3584), (32768, 3584, 18944), (32768, 3584, 3584)):
    a = torch.empty((M, K), dtype=torch.bfloat16, device="cuda"); w = torch.empty((N, K), dtype=torch.bfloat16, device="cuda")
    eng.fill_bell_bf16(a, 1, "a", 1.0); eng.fill_bell_bf16(w, 1, "w", 0.02)
    nwg = (M // 256) * (N // 256)
    st = torch.zeros((nwg, 8), dtype=torch.int64, device="cuda")
    eng.gemm_bf16(a, w); torch.cuda.synchronize()
    lib.blim_debug_gemm_stamps(st.data_ptr())
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record(); eng.gemm_bf16(a, w); e1.record(); torch.cuda.synchronize()
    lib.blim_debug_gemm_stamps(None)
    ms = e0.elapsed_time(e1)
    s = st.cpu().numpy().astype(np.float64)
    t_all = s[:, 3].max() - s[:, 0].min()
    tick_ns = 10.0                                 # s_memrealtime: constant 100 MHz
    print(f'   wall from stamps {t_all * tick_ns / 1e6:.3f} ms')
    d = np.diff(s[:, :4], axis=1) * tick_ns / 1e3         # us
    e_stage = (s[:, 4] - s[:, 2]) * tick_ns / 1e3; e_issue = (s[:, 5] - s[:, 4]) * tick_ns / 1e3; e_drain = (s[:, 3] - s[:, 5]) * tick_ns / 1e3
    print(f'   epilogue split: C->LDS {e_stage.mean():.2f} us, LDS->global issue {e_issue.mean():.2f} us, drain (vmcnt 0) {e_drain.mean():.2f} us')
    tot = (s[:, 3] - s[:, 0]) * tick_ns / 1e3
    # gaps between consecutive workgroups on the same CU cannot be seen directly; estimate turnover = wall - sum of in-kernel time per CU
    per_cu = tot.sum() / 256
    print(f"{M}x{N}x{K}: {ms:.3f} ms; tick {tick_ns:.3f} ns; per-WG us: setup {d[:,0].mean():.2f}  main loop {d[:,1].mean():.2f}  epilogue {d[:,2].mean():.2f}  total {tot.mean():.2f};"
          f" rounds {nwg/256:.1f}; sum in-kernel per CU {per_cu/1e3:.3f} ms -> turnover+idle {ms - per_cu/1e3:.3f} ms ({100*(ms - per_cu/1e3)/ms:.1f}%)")
    q = np.percentile(d[:, 1], [5, 50, 95]); print(f"    main loop p5/p50/p95 {q[0]:.1f}/{q[1]:.1f}/{q[2]:.1f} us; epilogue p5/p50/p95 {np.percentile(d[:,2],5):.1f}/{np.percentile(d[:,2],50):.1f}/{np.percentile(d[:,2],95):.1f}")
    del a, w
