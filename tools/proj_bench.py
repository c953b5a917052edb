"""Development aid: cost of projecting video features per video vs in chunks (plain and compensated mode), and of the upload."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from blim_amd import synth
from blim_amd.modeling import BlimModel

dims = synth.ModelDims()
m = BlimModel(dims, max_positions=1024, dtype="f16")
m.engine.init_synthetic_weights(0)
e = m.engine
C, T, M = 4, 64, 1024
vids = [torch.randn(C, T, M) for _ in range(128)]


def t(fn, n=3):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3


for precise in (False, True):
    for tvg in (False, True):
        e.set_precise(precise, embeds=precise)
        x1 = [v.to(e.device).to(m.dtype).reshape(C * T, M) for v in vids[:64]]
        xc = torch.cat(x1)
        per = t(lambda: [e.project_video(x, int(tvg)) for x in x1])
        chunk = t(lambda: e.project_video(xc, int(tvg)))
        y = e.project_video(xc, int(tvg))
        gm = t(lambda: e.group_mean(y, T))
        print(f"precise={precise} which={int(tvg)}: 64 per-video calls {per:.2f} ms, one 16K-row call {chunk:.2f} ms, group_mean of it {gm:.3f} ms", flush=True)
e.set_precise(False)
up1 = t(lambda: [v.to(e.device) for v in vids[:64]])
upc = t(lambda: torch.stack(vids[:64]).to(e.device))
st = t(lambda: torch.stack(vids[:64]))
print(f"upload 64 videos one by one {up1:.2f} ms; stacked {upc:.2f} ms (stack alone {st:.2f} ms)")
pin = torch.stack(vids[:64]).pin_memory()
print(f"pinned stacked upload {t(lambda: pin.to(e.device, non_blocking=True)):.2f} ms")
h = torch.stack(vids[:64]).half()
print(f"fp16 stacked upload {t(lambda: h.to(e.device)):.2f} ms; host fp32->fp16 {t(lambda: torch.stack(vids[:64]).half()):.2f} ms")
