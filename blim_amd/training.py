"""Fine-tuning on the MI355X engine -- host side of include/blim.h's blim_train_* (SURVEY.md section 8f-4).

Mirrors the reference's training step (training_utils.py:39-104), optimizer set-up (main.py:146-148), AMP loss scaler
(util/misc.py:232-259), LR schedule (util/lr_sched.py) and checkpoint writer (util/misc.py:276-297):

    trainer = Trainer(model.engine, lora_r=8, lora_alpha=32, lora_dropout=0.05)         # main.py:96-111
    stats = train_one_epoch(trainer, data_loader_train, epoch, args)                   # training_utils.py:39
    trainer.adapters_into_engine()                                                     # then evaluation() scores the fine-tuned model (adapters apart; merge_into_engine() folds them in)
    save_model(args, epoch, trainer, name=f"epoch{epoch}")                             # util/misc.py:276

A batch is the reference's collate output (dataloader/base_dataset.py:119-163, train split: left-padded id / label / mask tensors, a
list of [4, 64, 1024] features, tvg_video_labels).  Rows are packed without padding; what the reference computes for pad positions
never reaches a loss.  Multi-GPU: one process per GPU, every rank steps on its own batches, the flat gradient buffer is averaged
with ONE all-reduce (RCCL) before the optimizer step -- what DistributedDataParallel does bucket by bucket for the reference
(main.py:141-143).  There is no CPU path: Trainer raises when the HIP library is missing.
"""
from __future__ import annotations

import ctypes as C
import math
import os
from typing import Dict, Optional

import numpy as np

from . import lora
from .engine import Batch, BlimError, Engine, PackedBatch, _check, _stream, load_library
from .synth import IGNORE_INDEX, IMAGE_TOKEN_INDEX

IM_END = 151645          # videochat_flash/conversation.py:13 IMAGE_TOKEN_ID: the label that follows the <image> placeholder in a TVG row


class TrainConfig(C.Structure):
    _fields_ = [("lora_r", C.c_int32), ("lora_alpha", C.c_float), ("lora_dropout", C.c_float)]


class TrainBatch(C.Structure):
    _fields_ = [("batch", C.POINTER(Batch)), ("src_index", C.c_void_p), ("feats", C.c_void_p), ("n_feat_rows", C.c_int64),
                ("tok_per_clip", C.c_int32), ("max_seq_len", C.c_int32), ("rows", C.c_void_p), ("labels", C.c_void_p), ("n_rows", C.c_int64),
                ("tvg_rows", C.c_void_p), ("tvg_labels", C.c_void_p), ("n_tvg_rows", C.c_int64),
                ("vocab", C.c_void_p), ("n_vocab", C.c_int32), ("grad_scale", C.c_float), ("dropout_seed", C.c_uint64)]


_bound = False


def _lib():
    global _bound
    lib = load_library()
    if not _bound:
        vp, i32, i64, f32 = C.c_void_p, C.c_int32, C.c_int64, C.c_float
        sig = {
            "blim_train_flat_size": ([vp, i32], i64),
            "blim_train_param_offset": ([vp, i32, C.c_char_p, C.POINTER(i64), C.POINTER(i64), C.POINTER(i64)], C.c_int),
            "blim_train_create": ([vp, C.POINTER(TrainConfig), vp, vp, C.POINTER(vp)], C.c_int),
            "blim_train_destroy": ([vp], None),
            "blim_train_sync_params": ([vp, vp], C.c_int),
            "blim_train_merge": ([vp, vp], C.c_int),
            "blim_train_step": ([vp, C.POINTER(TrainBatch), vp, vp], C.c_int),
            "blim_train_grad_stats": ([vp, f32, vp, vp], C.c_int),
            "blim_train_adamw": ([vp, vp, vp, f32, f32, f32, f32, f32, f32, i32, vp], C.c_int),
            "blim_train_debug_read": ([vp, C.c_char_p, vp, i64, vp], C.c_int),
        }
        for name, (args, res) in sig.items():
            fn = getattr(lib, name)
            fn.argtypes = args
            fn.restype = res
        _bound = True
    return lib


# ----------------------------------------------------------------------------- rows of one batch

class PackedRows:
    """Packed token rows of one loss (VTG or TVG) of one training batch."""

    def __init__(self, src_index, seq_len, rows, labels, n_feat_rows):
        self.src_index = np.asarray(src_index, np.int32)
        self.seq_len = np.asarray(seq_len, np.int32)
        self.rows = np.asarray(rows, np.int32)
        self.labels = np.asarray(labels, np.int32)
        self.n_feat_rows = int(n_feat_rows)


def _strip(ids, mask, labels):
    keep = np.asarray(mask).astype(bool)                      # modeling_videochat_flash.py:333-334: drop the left pad
    return np.asarray(ids)[keep], np.asarray(labels)[keep]


def pack_vtg_rows(ids_rows, mask_rows, label_rows, n_video_tokens: int) -> PackedRows:
    """VTG rows (training_utils.py:62-68): the <image> placeholder becomes the video's n_video_tokens projected tokens (labels -100,
    modeling_videochat_flash.py:421-433); scored rows are the positions whose NEXT label is a real token (training_utils.py:24-26)."""
    src, lens, rows, labs, base = [], [], [], [], 0
    for b, (ids, m, lab) in enumerate(zip(ids_rows, mask_rows, label_rows)):
        ids, lab = _strip(ids, m, lab)
        where = np.nonzero(ids == IMAGE_TOKEN_INDEX)[0]
        if len(where) != 1:
            raise ValueError(f"row {b}: expected exactly one <image> placeholder, found {len(where)}")
        w = int(where[0])
        f0 = b * n_video_tokens
        s = np.concatenate([ids[:w], -(np.arange(f0, f0 + n_video_tokens) + 1), ids[w + 1:]])
        l = np.concatenate([lab[:w], np.full(n_video_tokens, IGNORE_INDEX, np.int64), lab[w + 1:]])
        pos = np.nonzero(l[1:] != IGNORE_INDEX)[0]
        src.append(s); lens.append(len(s)); rows.append(base + pos); labs.append(l[1:][pos])
        base += len(s)
    return PackedRows(np.concatenate(src), lens, np.concatenate(rows), np.concatenate(labs), len(ids_rows) * n_video_tokens)


def pack_tvg_rows(ids_rows, mask_rows, label_rows, num_clips: int) -> PackedRows:
    """TVG rows (training_utils.py:70-79): the placeholder becomes num_clips clip-mean tokens; the scored rows are the num_clips
    positions ending two before the <|im_end|> label: p + (arange(C) - (C + 1))."""
    src, lens, rows, base = [], [], [], 0
    for b, (ids, m, lab) in enumerate(zip(ids_rows, mask_rows, label_rows)):
        ids, lab = _strip(ids, m, lab)
        where = np.nonzero(ids == IMAGE_TOKEN_INDEX)[0]
        if len(where) != 1:
            raise ValueError(f"row {b}: expected exactly one <image> placeholder, found {len(where)}")
        w = int(where[0])
        s = np.concatenate([ids[:w], -(np.arange(b * num_clips, (b + 1) * num_clips) + 1), ids[w + 1:]])
        l = np.concatenate([lab[:w], np.full(num_clips, IGNORE_INDEX, np.int64), lab[w + 1:]])
        p = np.nonzero(l == IM_END)[0]
        if len(p) != 1:
            raise ValueError(f"row {b}: expected exactly one <|im_end|> label, found {len(p)}")
        src.append(s); lens.append(len(s)); rows.append(base + int(p[0]) + np.arange(num_clips) - (num_clips + 1))
        base += len(s)
    return PackedRows(np.concatenate(src), lens, np.concatenate(rows), np.zeros(0, np.int32), len(ids_rows) * num_clips)


# ----------------------------------------------------------------------------- AMP loss scaler

class LossScaler:
    """torch.cuda.amp.GradScaler's rule (init 2^16, x2 after 2000 clean steps, x0.5 and skip on inf / nan), util/misc.py:232-259."""
    state_dict_key = "amp_scaler"

    def __init__(self, init_scale: float = 65536.0, growth_factor: float = 2.0, backoff_factor: float = 0.5, growth_interval: int = 2000, enabled: bool = True):
        self.scale = float(init_scale) if enabled else 1.0
        self.growth_factor, self.backoff_factor, self.growth_interval, self.enabled = growth_factor, backoff_factor, growth_interval, enabled
        self._good = 0

    def update(self, found_inf: bool) -> None:
        if not self.enabled:
            return
        if found_inf:
            self.scale *= self.backoff_factor; self._good = 0
        else:
            self._good += 1
            if self._good >= self.growth_interval:
                self.scale *= self.growth_factor; self._good = 0

    def state_dict(self):
        return {"scale": self.scale, "growth_factor": self.growth_factor, "backoff_factor": self.backoff_factor,
                "growth_interval": self.growth_interval, "_growth_tracker": self._good}

    def load_state_dict(self, sd):
        self.scale = float(sd["scale"]); self._good = int(sd.get("_growth_tracker", 0))


def average_gradients(flat_grads, world_size: int) -> None:
    """What DistributedDataParallel does for the reference (main.py:141-143), in one call: the whole trainable set (~20 M floats at 7B,
    r = 8) is ONE flat buffer, so there is one all-reduce (RCCL over xGMI: 80 MB, far below the per-link bandwidth-latency knee of a
    bucketed scheme) and a division by the world size."""
    if world_size > 1:
        import torch.distributed as dist
        if flat_grads.is_cuda and dist.get_backend() == "gloo":      # test set-up only (two ranks sharing one GPU): gloo reduces host tensors
            host = flat_grads.cpu()
            dist.all_reduce(host)
            flat_grads.copy_(host)
        else:
            dist.all_reduce(flat_grads)
        flat_grads.div_(world_size)


def adjust_learning_rate(epoch: float, args) -> float:
    """util/lr_sched.py:9-21: linear warm-up, then half-cycle cosine."""
    if epoch < args.warmup_epochs:
        return args.lr * epoch / args.warmup_epochs
    return args.min_lr + (args.lr - args.min_lr) * 0.5 * (1.0 + math.cos(math.pi * (epoch - args.warmup_epochs) / (args.epochs - args.warmup_epochs)))


# ----------------------------------------------------------------------------- trainer

class Trainer:
    """LoRA adapters + visual_head on a loaded Engine (main.py:96-111), AdamW state (main.py:147), loss scaler (main.py:149)."""

    def __init__(self, engine: Engine, lora_r: int = 8, lora_alpha: float = 32.0, lora_dropout: float = 0.05, seed: int = 0,
                 betas=(0.9, 0.95), eps: float = 1e-8, weight_decay: float = 0.05, amp: bool = True,
                 trainable: Optional[Dict[str, np.ndarray]] = None, visual_head: Optional[np.ndarray] = None):
        import torch
        self.lib = _lib()
        self.engine, self.dims = engine, engine.dims
        self.r, self.alpha, self.dropout = int(lora_r), float(lora_alpha), float(lora_dropout)
        self.betas, self.eps, self.weight_decay = betas, eps, weight_decay
        n = self.lib.blim_train_flat_size(engine.h, self.r)
        if n <= 0:
            raise BlimError("blim_train_flat_size failed")
        self.layout, total = lora.flat_layout(self.dims, self.r)
        if total != n:
            raise BlimError(f"flat layout mismatch between blim_amd/lora.py ({total}) and the library ({n})")
        dev = engine.device
        self.params = torch.zeros(n, dtype=torch.float32, device=dev)
        self.grads = torch.zeros(n, dtype=torch.float32, device=dev)
        self.exp_avg = torch.zeros(n, dtype=torch.float32, device=dev)
        self.exp_avg_sq = torch.zeros(n, dtype=torch.float32, device=dev)
        self._stats = torch.zeros(2, dtype=torch.float32, device=dev)
        self._loss = torch.zeros(2, dtype=torch.float32, device=dev)
        init = trainable if trainable is not None else lora.init_trainable(self.dims, self.r, seed, visual_head)
        missing = [n for n in self.layout if n not in init]
        if missing:
            raise BlimError(f"trainable tensors missing: {missing[:4]}{' ...' if len(missing) > 4 else ''} ({len(missing)} of {len(self.layout)})")
        for name, arr in init.items():
            off, shape = self.layout[name]
            assert tuple(arr.shape) == tuple(shape), (name, arr.shape, shape)
            self.params[off: off + arr.size] = torch.from_numpy(np.ascontiguousarray(arr, np.float32).reshape(-1)).to(dev)
        cfg = TrainConfig(self.r, self.alpha, self.dropout)
        h = C.c_void_p()
        _check(self.lib.blim_train_create(engine.h, C.byref(cfg), self.params.data_ptr(), self.grads.data_ptr(), C.byref(h)), "blim_train_create")
        self.h = h
        self.step_count = 0               # optimizer steps taken (AdamW bias correction)
        self.scaler = LossScaler(enabled=amp and engine.dtype == "f16")      # bf16 has fp32's exponent range: no scaling needed
        self._vocab_key, self._vocab = None, None
        self._copy_stream = None

    def close(self):
        if getattr(self, "h", None):
            self.lib.blim_train_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- parameters
    def tensor(self, name: str, which: str = "params"):
        off, shape = self.layout[name]
        return getattr(self, which)[off: off + int(np.prod(shape))].view(*shape)

    def state(self, which: str = "params") -> Dict[str, np.ndarray]:
        return {n: self.tensor(n, which).cpu().numpy() for n in self.layout}

    def load_trainable(self, tensors: Dict[str, np.ndarray]) -> None:
        import torch
        for n, a in tensors.items():
            self.tensor(n).copy_(torch.from_numpy(np.ascontiguousarray(a, np.float32)))
        _check(self.lib.blim_train_sync_params(self.h, _stream()), "blim_train_sync_params")

    def zero_grad(self) -> None:
        self.grads.zero_()

    def merge_into_engine(self) -> None:
        """Engine weights <- W + alpha/r * B A (+ visual_head): what evaluation() then scores (val_one_epoch after every epoch, main.py:166)."""
        _check(self.lib.blim_train_merge(self.h, _stream()), "blim_train_merge")
        self.engine.weights_version += 1          # (a numeric mode measured on the weights before the merge no longer stands: modeling.py)

    def adapters_into_engine(self) -> None:
        """The current adapters + visual_head handed to the scoring engine as SEPARATE matrices (blim_load_adapter): the in-training validation then scores
        y = W x + (alpha / r) B (A x) on the pristine base weights, as the reference's live peft model does (main.py:166 -> val_one_epoch)."""
        st = self.state()
        for name, arr in st.items():
            if name == "visual_head":
                self.engine.load_weight("visual_head", arr)
            elif name.endswith(":A"):
                w = name[:-2]
                self.engine.load_adapter(w, arr, st[w + ":B"], self.r, self.alpha)

    # ---- one batch
    def set_video_vocab(self, video_vocab) -> None:
        """[N, clips, mm_hidden] clip means of the training set (training_utils.py:50-51) -> clip-major 16-bit device tensor."""
        import torch
        key = (video_vocab.data_ptr() if hasattr(video_vocab, "data_ptr") else id(video_vocab), tuple(video_vocab.shape))
        if key != self._vocab_key:
            v = torch.as_tensor(video_vocab).to(self.engine.device, self.engine.torch_dtype)
            self._vocab = v.permute(1, 0, 2).contiguous()
            self._vocab_key = key

    def _train_batch(self, vtg: PackedRows, tvg: PackedRows, feats, tok_per_clip: int, seed: int, vocab, video_labels):
        """ONE packed batch: the VTG rows, then the TVG rows (a tenth of the tokens: as a decoder pass of their own they ran at a third of
        the VTG pass's efficiency).  VTG video tokens point at rows [0, F) of the `mlp` projection, TVG clip tokens at the clip means of the
        `tvg_mlp` projection, numbered from F."""
        import torch
        dev = self.engine.device
        F = vtg.n_feat_rows
        t_src = tvg.src_index.copy()
        neg = t_src < 0
        t_src[neg] -= F                                   # -(fm + 1) -> -(F + fm + 1)
        src = np.concatenate([vtg.src_index, t_src]).astype(np.int32)
        seq_len = np.concatenate([vtg.seq_len, tvg.seq_len]).astype(np.int32)
        seq_start = np.concatenate([[0], np.cumsum(seq_len)[:-1]]).astype(np.int32)
        positions = np.concatenate([np.arange(n, dtype=np.int32) for n in seq_len])
        n_vtg_tok = int(vtg.seq_len.sum())
        pb = PackedBatch(positions, np.ones(len(positions), np.uint8), seq_start, seq_len, device=dev)
        keep = [pb, torch.from_numpy(src).to(dev), torch.from_numpy(vtg.rows).to(dev), torch.from_numpy(vtg.labels).to(dev),
                torch.from_numpy((tvg.rows + n_vtg_tok).astype(np.int32)).to(dev), torch.from_numpy(np.asarray(video_labels, np.int32)).to(dev), feats]
        st = pb.struct(self.engine.max_positions)
        tb = TrainBatch()
        tb.batch = C.pointer(st)
        tb.src_index, tb.feats, tb.n_feat_rows = keep[1].data_ptr(), feats.data_ptr(), feats.shape[0]
        tb.tok_per_clip, tb.max_seq_len = int(tok_per_clip), int(seq_len.max())
        tb.rows, tb.labels, tb.n_rows = keep[2].data_ptr(), keep[3].data_ptr(), len(vtg.rows)
        tb.tvg_rows, tb.tvg_labels, tb.n_tvg_rows = keep[4].data_ptr(), keep[5].data_ptr(), len(tvg.rows)
        tb.vocab, tb.n_vocab = vocab.data_ptr(), vocab.shape[1]
        tb.dropout_seed = int(seed) & 0xFFFFFFFFFFFFFFFF
        return tb, keep + [st]

    # One batch in three phases, so that the host work of batch i + 1 (row packing, feature stacking, host-to-device copies on a side
    # stream) runs while the GPU is busy with batch i (train_one_epoch): stage() -> launch() -> finish().
    def stage(self, data: dict, seed: int = 0) -> dict:
        """Host side of one collated batch: packed VTG / TVG rows and everything the step reads, resident on the device."""
        import torch
        dims, dev, dt = self.dims, self.engine.device, self.engine.torch_dtype
        if self._vocab is None:
            raise BlimError("set_video_vocab() first (training_utils.py:50-51)")
        if self._copy_stream is None:
            self._copy_stream = torch.cuda.Stream(device=dev)
        video = [torch.as_tensor(v) for v in data["video"]]
        bs = len(video)
        C_, tok = video[0].shape[0], video[0].shape[1]
        if C_ != dims.num_clips:
            raise ValueError(f"features have {C_} clips, the model expects {dims.num_clips}")
        as_rows = lambda x: [np.asarray(r) for r in (x.cpu().numpy() if hasattr(x, "cpu") else x)]
        vtg = pack_vtg_rows(as_rows(data["vtg_ids"]), as_rows(data["vtg_masks"]), as_rows(data["vtg_labels"]), C_ * tok)
        tvg = pack_tvg_rows(as_rows(data["tvg_ids"]), as_rows(data["tvg_masks"]), as_rows(data["tvg_labels"]), C_)
        vl = np.asarray(data["tvg_video_labels"].cpu().numpy() if hasattr(data["tvg_video_labels"], "cpu") else data["tvg_video_labels"], np.int32)
        host = torch.stack(video)
        if not host.is_cuda and not host.is_pinned():
            host = host.pin_memory()                          # features may also arrive on the device already
        with torch.cuda.stream(self._copy_stream):
            feats = host.to(dev, non_blocking=True).to(dt).reshape(bs * C_ * tok, dims.mm_hidden_size).contiguous()
            tb, keep = self._train_batch(vtg, tvg, feats, tok, seed, self._vocab, vl)
            loss = torch.zeros(2, dtype=torch.float32, device=dev)
            ready = torch.cuda.Event()
            ready.record(self._copy_stream)
        return {"tb": tb, "keep": (keep, host), "loss": loss, "ready": ready, "n": (len(vtg.rows), len(tvg.rows))}

    def launch(self, staged: dict, accum_iter: int = 1) -> None:
        """training_utils.py:57-85: both losses forward + backward on the current stream, the two kinds of rows in one decoder pass;
        gradients accumulate, scaled by loss scale / accum_iter."""
        import torch
        torch.cuda.current_stream().wait_event(staged["ready"])
        tb = staged["tb"]
        tb.grad_scale = float(self.scaler.scale / accum_iter)
        _check(self.lib.blim_train_step(self.h, C.byref(tb), staged["loss"].data_ptr(), _stream()), "blim_train_step")

    def finish(self, staged: dict):
        """(vtg_loss, tvg_loss) of a launched batch; synchronises with the GPU (the reference's loss.item(), training_utils.py:83)."""
        sums = staged["loss"].cpu().numpy()
        staged["keep"] = None
        return float(sums[0]) / staged["n"][0], float(sums[1]) / staged["n"][1]

    def forward_backward(self, data: dict, accum_iter: int = 1, seed: int = 0):
        st = self.stage(data, seed)
        self.launch(st, accum_iter)
        return self.finish(st)

    def optimizer_step(self, lr: float, world_size: int = 1) -> Dict[str, float]:
        """loss_scaler(...)'s update branch (util/misc.py:240-249): [all-reduce], unscale, inf check, grad norm, AdamW, scaler update."""
        import torch
        average_gradients(self.grads, world_size)
        inv = 1.0 / self.scaler.scale
        self._stats.zero_()
        _check(self.lib.blim_train_grad_stats(self.h, inv, self._stats.data_ptr(), _stream()), "blim_train_grad_stats")
        st = self._stats.cpu().numpy()
        found_inf = bool(st[1] > 0) or not math.isfinite(float(st[0]))
        if not found_inf:
            self.step_count += 1
            _check(self.lib.blim_train_adamw(self.h, self.exp_avg.data_ptr(), self.exp_avg_sq.data_ptr(), lr, self.betas[0], self.betas[1], self.eps,
                                             self.weight_decay, inv, self.step_count, _stream()), "blim_train_adamw")
        self.scaler.update(found_inf)
        return {"grad_norm": float(math.sqrt(max(st[0], 0.0))) if not found_inf else float("inf"), "skipped": float(found_inf)}

    # ---- checkpoints (util/misc.py:276-297: only the tensors with requires_grad, under peft's key names)
    def checkpoint_state(self) -> dict:
        import torch
        model = {lora.resume_key(n): torch.from_numpy(a) for n, a in self.state().items()}
        opt = {"step": self.step_count, "exp_avg": self.exp_avg.cpu(), "exp_avg_sq": self.exp_avg_sq.cpu(), "betas": self.betas, "eps": self.eps,
               "weight_decay": self.weight_decay, "layout": {n: (o, tuple(s)) for n, (o, s) in self.layout.items()}}
        return {"model": model, "optimizer": opt, "scaler": self.scaler.state_dict()}

    def load_checkpoint_state(self, ckpt: dict, strict: bool = True) -> None:
        """`--resume` for training (util/misc.py:303-316).  strict: every trainable tensor must be in the file (the reference asserts the
        parameter total, main.py:127), so that a naming drift cannot silently continue from fresh adapters."""
        tensors = resume_tensors(ckpt)
        absent = [n for n in self.layout if n not in tensors]
        if strict and absent:
            raise BlimError(f"resume file lacks {len(absent)} of {len(self.layout)} trainable tensors (first: {absent[:3]})")
        self.load_trainable(tensors)
        opt = ckpt.get("optimizer")
        if isinstance(opt, dict) and "exp_avg" in opt:
            self.step_count = int(opt["step"])
            self.exp_avg.copy_(opt["exp_avg"]); self.exp_avg_sq.copy_(opt["exp_avg_sq"])
        if "scaler" in ckpt:
            self.scaler.load_state_dict(ckpt["scaler"])


def resume_tensors(ckpt: dict) -> Dict[str, np.ndarray]:
    """Trainable tensors of a resume file under the flat layout's names (`<weight>:A`, `<weight>:B`, `visual_head`).  The reference's
    save_model stores the nn.Parameter objects themselves (requires_grad=True, util/misc.py:282-285) and torch.load hands them back as such:
    detach before leaving torch."""
    from .checkpoint import parse_resume_key
    tensors = {}
    for k, v in ckpt["model"].items():
        parsed = parse_resume_key(k)
        if parsed is None:
            continue
        w, kind = parsed
        tensors["visual_head" if kind == "full" else f"{w}:{kind}"] = v.detach().float().cpu().numpy()
    return tensors


def train_one_epoch(trainer: Trainer, data_loader, epoch: int, args, world_size: int = 1, log=print) -> Dict[str, float]:
    """training_utils.py:39-104 on the engine.  `args`: accum_iter, lr, min_lr, warmup_epochs, epochs (the reference's flags)."""
    accum = max(1, int(getattr(args, "accum_iter", 1)))
    trainer.zero_grad()
    trainer.set_video_vocab(data_loader.dataset.video_vocab)                          # :50-51
    n_iter = len(data_loader)
    print_freq = max(1, int(n_iter / 4))                                              # :46
    sums = {"loss": 0.0, "vtg_loss": 0.0, "tvg_loss": 0.0}
    lr, seen = 0.0, 0
    rank = 0
    if world_size > 1:
        import torch.distributed as dist
        rank = dist.get_rank()
    seed_of = lambda i: ((epoch * n_iter + i) * 2 + 12345) * 1009 + rank          # dropout masks differ per step and per rank (main.py:87: seed + rank)
    batches = iter(data_loader)
    nxt = trainer.stage(next(batches), seed_of(0)) if n_iter else None
    for it in range(n_iter):
        if it % accum == 0:
            lr = adjust_learning_rate(it / n_iter + epoch, args)                     # :58-59
        cur = nxt
        trainer.launch(cur, accum_iter=accum)
        nxt = trainer.stage(next(batches), seed_of(it + 1)) if it + 1 < n_iter else None      # host work of the next batch under this batch's kernels
        vtg_loss, tvg_loss = trainer.finish(cur)
        loss = vtg_loss + tvg_loss
        if not math.isfinite(loss):                                                   # :83-85
            raise FloatingPointError(f"Loss is {loss}, stopping training")
        if (it + 1) % accum == 0:                                                     # :89-91
            trainer.optimizer_step(lr, world_size)
            trainer.zero_grad()
        sums["loss"] += loss; sums["vtg_loss"] += vtg_loss; sums["tvg_loss"] += tvg_loss; seen += 1
        if it % print_freq == 0 or it == n_iter - 1:
            log(f"Epoch: [{epoch}]  [{it}/{n_iter}]  lr: {lr:.6f}  loss: {loss:.4f}  vtg_loss: {vtg_loss:.4f}  tvg_loss: {tvg_loss:.4f}")
    out = {k: v / max(seen, 1) for k, v in sums.items()}
    out["lr"] = lr
    if world_size > 1:                                                                # metric_logger.synchronize_between_processes (:100)
        import torch
        import torch.distributed as dist
        t = torch.tensor([out["loss"], out["vtg_loss"], out["tvg_loss"]], dtype=torch.float64, device="cpu" if dist.get_backend() == "gloo" else trainer.engine.device)
        dist.all_reduce(t)
        out["loss"], out["vtg_loss"], out["tvg_loss"] = (t / world_size).tolist()
    log("Averaged stats: " + "  ".join(f"{k}: {v:.6f}" for k, v in out.items()))
    return out


def save_model(args, epoch: int, trainer: Trainer, name: str) -> str:
    """util/misc.py:276-297: <output_dir>/<name>.pth = {'model': trainable tensors, 'optimizer', 'epoch', 'scaler', 'args'}."""
    import torch
    os.makedirs(args.output_dir, exist_ok=True)
    path = os.path.join(args.output_dir, f"{name}.pth")
    st = trainer.checkpoint_state()
    st["epoch"] = epoch
    # the reference stores its argparse namespace; ours also carries tensors (iv2_scores: two N x N matrices) and private fields
    st["args"] = {k: v for k, v in vars(args).items() if not k.startswith("_") and isinstance(v, (int, float, str, bool, list, tuple, type(None)))}
    torch.save(st, path)
    return path
