"""Fine-tuning step throughput on the 7B configuration (blim_amd/training.py): B samples of reference shape per step
(VTG row = 26 prompt + 256 video + caption tokens, TVG row ~ 50 tokens), both losses forward + backward + AdamW.
    python tools/train_bench.py [B ...]          (under rocprofv3 --kernel-trace --stats for the per-kernel split)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from blim_amd import synth
from blim_amd.engine import Engine
from blim_amd.training import Trainer

small = os.environ.get("BLIM_TRAIN_BENCH_SMALL") == "1"
dims = synth.ModelDims(num_layers=4) if small else synth.ModelDims()
dtype = os.environ.get("BLIM_DTYPE", "f16")
eng = Engine(dims, max_positions=1024, dtype=dtype)
eng.init_synthetic_weights(0)
tr = Trainer(eng, lora_r=8, lora_alpha=32.0, lora_dropout=float(os.environ.get("BLIM_TRAIN_DROPOUT", "0.05")), seed=1)
n_vocab = int(os.environ.get("BLIM_TRAIN_VOCAB", "4096"))
steps = int(os.environ.get("BLIM_TRAIN_STEPS", "3"))
trace = os.environ.get("BLIM_TRAIN_TRACE") == "1"          # per-step losses (the same batch every step: the loss must fall)
lr = float(os.environ.get("BLIM_TRAIN_LR", "1e-4"))
H, I, L = dims.hidden_size, dims.intermediate_size, dims.num_layers
flop_tok = L * (2 * H * (H + 2 * dims.num_kv_heads * 128) + 2 * H * H + 6 * H * I)
for B in [int(a) for a in sys.argv[1:]] or [16]:
    prob = synth.make_problem(77, B, dims, tok_per_clip=64, text_len=(8, 48), fast_video=True)
    loader = synth.ProblemLoader(prob, B)
    vocab = torch.randn(n_vocab, dims.num_clips, dims.mm_hidden_size)
    vocab[:B] = torch.from_numpy(prob.video_vocab)
    tr.set_video_vocab(vocab)
    data = next(iter(loader))
    tok = sum(len(x) + 255 for x in prob.vtg_ids) + sum(len(x) + 3 for x in prob.tvg_ids)
    lab = sum(int((np.asarray(x)[1:] != -100).sum()) for x in prob.vtg_labels)
    nxt = tr.stage(data, 0)
    for it in range(steps + 1):                        # the loop of blim_amd.training.train_one_epoch: batch i + 1 is staged under batch i's kernels
        if it == 1:
            torch.cuda.synchronize(); t0 = time.time()
        tr.zero_grad()
        cur = nxt
        tr.launch(cur)
        nxt = tr.stage(data, it + 1) if it < steps else None
        lv, lt = tr.finish(cur)
        st = tr.optimizer_step(lr)
        if trace:
            print(f"  step {it}: vtg {lv:.4f} tvg {lt:.4f} grad norm {st['grad_norm']:.3g} scale {tr.scaler.scale:g} skipped {int(st['skipped'])}", flush=True)
    torch.cuda.synchronize()
    dt = (time.time() - t0) / steps
    fl = 2 * tok * flop_tok + 2 * lab * 2 * H * dims.vocab_size          # forward + input-gradient GEMMs (frozen weights: no weight-gradient GEMMs)
    print(f"B={B}: {dt * 1e3:.0f} ms/step = {B / dt:.1f} samples/s; {tok} tokens, {lab} label rows; loss {lv:.3f} + {lt:.3f}, grad norm {st['grad_norm']:.3g}; "
          f"{fl / 1e12:.0f} TFLOP of GEMMs -> {fl / dt / 1e12:.0f} TFLOP/s; HBM in use {torch.cuda.mem_get_info()[1] / 2**30 - torch.cuda.mem_get_info()[0] / 2**30:.0f} GiB", flush=True)
tr.close(); eng.close()
