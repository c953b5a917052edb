"""Development aid: per-class kernel times (HIP events) of a decode over the bench's token count with 1 or 28 layers -- un-instrumented counterpart of
tools/epi_stamps.py (whose stamps slow the long gate|up / down launches by 12 - 17 %, the short ones not at all).   python tools/onelayer_classes.py [layers]"""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from blim_amd import engine as eng, synth
L = int(sys.argv[1]) if len(sys.argv) > 1 else 1
dims = synth.ModelDims(num_layers=L)
E = eng.Engine(dims, max_positions=1024, dtype="f16")
E.init_synthetic_weights(0)
T, Ls = 32560, 148
n_seq = T // Ls
batch = eng.PackedBatch(np.tile(np.arange(Ls, dtype=np.int32), n_seq), np.ones(T, np.uint8), np.arange(n_seq, dtype=np.int32) * Ls, np.full(n_seq, Ls, np.int32))
emb = (torch.randn((T, dims.hidden_size), device="cuda") * 0.02).to(E.torch_dtype)
for _ in range(3): E.decode(batch, emb)
torch.cuda.synchronize()
E.timing_enable(True)
for _ in range(3): E.decode(batch, emb)
torch.cuda.synchronize()
rep = E.timing_report()
print(L, "layers:", {k: round(v["ms"] / max(v["calls"], 1), 3) for k, v in rep.items() if v["calls"]})
E.close()
