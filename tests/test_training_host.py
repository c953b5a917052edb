"""Host side of the fine-tuning path (blim_amd/training.py, blim_amd/lora.py) on CPU: row packing against the oracle's per-row
construction, the AMP scaler rule, the LR schedule, checkpoint key naming, and the one-call gradient averaging over gloo (W = 2)."""
import math
import os
import socket
import types

import numpy as np
import torch

from blim_amd import lora, synth
from blim_amd.training import LossScaler, adjust_learning_rate, average_gradients, pack_tvg_rows, pack_vtg_rows
from oracle.train_oracle import IM_END, cosine_lr

DIMS = synth.ModelDims(vocab_size=151700, hidden_size=256, intermediate_size=512, num_layers=2, num_heads=2, num_kv_heads=1, mm_hidden_size=64)


def _left_pad(rows, fill):
    L = max(len(r) for r in rows)
    out = np.full((len(rows), L), fill, np.int64)
    for i, r in enumerate(rows):
        out[i, L - len(r):] = r
    return out


def test_pack_rows_match_the_reference_row_layout():
    prob = synth.make_problem(5, 4, DIMS, tok_per_clip=8, text_len=(3, 9))
    nv = DIMS.num_clips * 8
    ids, lab, msk = _left_pad(prob.vtg_ids, synth.PAD_ID), _left_pad(prob.vtg_labels, -100), _left_pad(prob.vtg_masks, 0)
    pk = pack_vtg_rows(list(ids), list(msk), list(lab), nv)
    base = 0
    rows, labels = [], []
    for b in range(4):          # restated from modeling_videochat_flash.py:395-444 + training_utils.py:24-26 (as oracle/train_oracle.py does per row)
        i, l = prob.vtg_ids[b], prob.vtg_labels[b]
        w = int(np.nonzero(i == -200)[0][0])
        full = np.concatenate([l[:w], np.full(nv, -100), l[w + 1:]])
        assert pk.seq_len[b] == len(full)
        seg = pk.src_index[base: base + len(full)]
        assert np.array_equal(seg[:w], i[:w]) and np.array_equal(seg[w + nv:], i[w + 1:])
        assert np.array_equal(seg[w: w + nv], -(np.arange(b * nv, (b + 1) * nv) + 1))
        pos = np.nonzero(full[1:] != -100)[0]
        rows.append(base + pos); labels.append(full[1:][pos])
        base += len(full)
    assert np.array_equal(pk.rows, np.concatenate(rows)) and np.array_equal(pk.labels, np.concatenate(labels))
    assert pk.n_feat_rows == 4 * nv

    C = DIMS.num_clips
    ids, lab, msk = _left_pad(prob.tvg_ids, synth.PAD_ID), _left_pad(prob.tvg_labels, -100), _left_pad(prob.tvg_masks, 0)
    pt = pack_tvg_rows(list(ids), list(msk), list(lab), C)
    base = 0
    for b in range(4):
        i, l = prob.tvg_ids[b], prob.tvg_labels[b]
        w = int(np.nonzero(i == -200)[0][0])
        full = np.concatenate([l[:w], np.full(C, -100), l[w + 1:]])
        p = int(np.nonzero(full == IM_END)[0][0])
        assert np.array_equal(pt.rows[b * C:(b + 1) * C], base + p + np.arange(C) - (C + 1))      # training_utils.py:73
        assert np.array_equal(pt.src_index[base + w: base + w + C], -(np.arange(b * C, (b + 1) * C) + 1))
        base += len(full)
    assert pt.n_feat_rows == 4 * C


def test_loss_scaler_follows_grad_scaler_rule():
    s = LossScaler(growth_interval=3)
    assert s.scale == 65536.0
    s.update(True); assert s.scale == 32768.0
    s.update(False); s.update(False); assert s.scale == 32768.0
    s.update(False); assert s.scale == 65536.0
    s.update(False); s.update(True); assert s.scale == 32768.0 and s._good == 0
    t = LossScaler(); t.load_state_dict(s.state_dict()); assert t.scale == s.scale
    off = LossScaler(enabled=False); off.update(True); assert off.scale == 1.0


def test_lr_schedule_is_the_reference_rule():
    args = types.SimpleNamespace(lr=2e-4, min_lr=1e-6, warmup_epochs=2, epochs=10)
    for e in (0.0, 0.5, 1.99, 2.0, 3.7, 9.99):
        assert math.isclose(adjust_learning_rate(e, args), cosine_lr(e, args.lr, args.min_lr, args.warmup_epochs, args.epochs), rel_tol=1e-12)


def test_flat_layout_and_init():
    lay, total = lora.flat_layout(DIMS, 8)
    offs = sorted((o, int(np.prod(s))) for o, s in lay.values())
    assert all(o % 64 == 0 for o, _ in offs)
    assert all(offs[i][0] + offs[i][1] <= offs[i + 1][0] for i in range(len(offs) - 1)) and offs[-1][0] + offs[-1][1] <= total
    init = lora.init_trainable(DIMS, 8, seed=3)
    assert all(np.all(v == 0) for k, v in init.items() if k.endswith(":B"))
    a = init["layers.0.q_proj.w:A"]
    assert a.shape == (8, 256) and np.abs(a).max() <= 1 / math.sqrt(256) and a.std() > 0.5 / math.sqrt(3 * 256)
    assert np.array_equal(init["tvg_mlp.0.w:A"], init["mlp.0.w:A"])                      # deepcopy, main.py:98
    assert abs(init["visual_head"].std() - 0.02) < 2e-3                                  # no head in the checkpoint: N(0, 0.02) like a missing nn.Linear
    assert np.array_equal(lora.init_trainable(DIMS, 8, seed=3, visual_head=np.ones((64, 256), np.float32))["visual_head"], np.ones((64, 256), np.float32))
    n_trainable = sum(int(np.prod(s)) for s in lora.trainable_shapes(synth.ModelDims(), 8).values())
    # Qwen2-7B, r = 8: 28 x (q 57,344 + k 32,768 + v 32,768 + o 57,344) + lm_head 1,245,184 + projectors 2 x (36,864 + 57,344) + visual_head 3,670,016
    assert n_trainable == 28 * 180_224 + 1_245_184 + 2 * 94_208 + 3_670_016


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _avg_worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = torch.arange(1000, dtype=torch.float32) * (rank + 1)
    average_gradients(g, world)
    q.put((rank, bool(torch.allclose(g, torch.arange(1000, dtype=torch.float32) * 1.5))))
    dist.barrier(); dist.destroy_process_group()


def test_gradient_averaging_world2_gloo():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q, port = ctx.Queue(), _free_port()
    procs = [ctx.Process(target=_avg_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60); assert p.exitcode == 0
    assert all(ok for _, ok in res)
    g = torch.ones(4); average_gradients(g, 1); assert torch.equal(g, torch.ones(4))


def test_resume_file_holding_parameters_like_the_reference_writes():
    """The reference's save_model stores named_parameters() objects with requires_grad=True (util/misc.py:282-285) and torch.load returns
    them unchanged: the trainer's resume path must detach (`.numpy()` on such a tensor raises)."""
    import io
    import torch
    from blim_amd import lora, synth
    from blim_amd.training import resume_tensors
    dims = synth.ModelDims(vocab_size=300, hidden_size=64, intermediate_size=128, num_layers=2, num_heads=2, num_kv_heads=1, mm_hidden_size=32)
    init = lora.init_trainable(dims, 8, seed=3)
    model = {lora.resume_key(n): torch.nn.Parameter(torch.from_numpy(a.copy()).half(), requires_grad=True) for n, a in init.items()}
    buf = io.BytesIO()
    torch.save({"model": model, "optimizer": {"state": {}, "param_groups": []}, "epoch": 4, "args": None}, buf)      # torch-format optimizer, as the reference
    buf.seek(0)
    ckpt = torch.load(buf, map_location="cpu", weights_only=False)
    assert all(v.requires_grad for v in ckpt["model"].values())
    got = resume_tensors(ckpt)
    assert set(got) == set(init)
    for n in init:
        assert got[n].dtype == np.float32 and np.array_equal(got[n], init[n].astype(np.float16).astype(np.float32)), n
