"""Per-tile anatomy of the FUSED-epilogue GEMMs inside the decoder (in-kernel s_memrealtime stamps): one 7B-width layer over the
bench's token count; BLIM_GEMM_STAMP_EPI / BLIM_GEMM_STAMP_K select which launch writes the stamps (set by this script through
child processes).   python tools/epi_stamps.py            -> table for qkv, o_proj, gate|up, down"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CASES = [("gemm_qkv_rope", 3, 3584, 4608), ("gemm_o_resid", 2, 3584, 3584), ("gemm_gateup_swiglu", 4, 3584, 37888), ("gemm_down_resid", 2, 18944, 3584)]
if len(sys.argv) == 1:
    for name, epi, k, n in CASES:
        env = dict(os.environ, BLIM_GEMM_STAMP_EPI=str(epi), BLIM_GEMM_STAMP_K=str(k))
        subprocess.run([sys.executable, os.path.abspath(__file__), name, str(n), str(k)], env=env, check=True)
    sys.exit(0)
import numpy as np, torch
from blim_amd import engine as eng, synth
name, N, K = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
dims = synth.ModelDims(num_layers=1)
E = eng.Engine(dims, max_positions=1024, dtype=os.environ.get("BLIM_DTYPE", "f16"))
E.init_synthetic_weights(0)
T, Ls = 32560, 148
n_seq = T // Ls
batch = eng.PackedBatch(np.tile(np.arange(Ls, dtype=np.int32), n_seq), np.ones(T, np.uint8), np.arange(n_seq, dtype=np.int32) * Ls, np.full(n_seq, Ls, np.int32))
emb = (torch.randn((T, dims.hidden_size), device="cuda") * 0.02).to(E.torch_dtype)
E.decode(batch, emb); torch.cuda.synchronize()
nwg = ((T + 255) // 256) * ((N + 255) // 256)
st = torch.zeros((nwg, 8), dtype=torch.int64, device="cuda")
E.lib.blim_debug_gemm_stamps(st.data_ptr())
E.decode(batch, emb); torch.cuda.synchronize()
E.lib.blim_debug_gemm_stamps(None)
s = st.cpu().numpy().astype(np.float64)
s = s[s[:, 3] > 0]
us = lambda a: a * 10.0 / 1e3
tot = us(s[:, 3] - s[:, 0]); setup = us(s[:, 1] - s[:, 0]); main = us(s[:, 2] - s[:, 1]); epi = us(s[:, 3] - s[:, 2])
wall = us(s[:, 3].max() - s[:, 0].min())
stage = us(s[:, 4] - s[:, 2]); issue = us(s[:, 5] - s[:, 4]); drain = us(s[:, 3] - s[:, 5])
print(f"{name}: {len(s)} tiles ({len(s) / 256:.2f} rounds), wall {wall / 1e3:.3f} ms; per tile us: setup {setup.mean():.2f}  main loop {main.mean():.2f} "
      f"({main.mean() / (K * (1 if E.dtype == 'f8' else 2) / 128):.3f} per K-step)  epilogue {epi.mean():.2f} [p5 {np.percentile(epi, 5):.1f} p95 {np.percentile(epi, 95):.1f}]"
      + (f"  (C->LDS {stage.mean():.2f}, stores issued {issue.mean():.2f}, drain {drain.mean():.2f})" if (s[:, 4] > 0).all() else "")
      + f"  total {tot.mean():.2f}; sum per CU {tot.sum() / 256 / 1e3:.3f} ms", flush=True)
E.close()
