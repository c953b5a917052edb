"""ctypes binding of libblim_hip.so (include/blim.h).

PyTorch is used only for device memory / streams: every tensor handed to the engine is a CUDA(HIP)
tensor whose data_ptr() goes through the C ABI.  There is NO CPU fallback: importing works without a
GPU (so the CPU test-suite can check symbols and host logic), but creating an engine or launching a
kernel without the library or without a device raises.
"""
from __future__ import annotations

import ctypes as C
import os
import re
from typing import Dict, Optional, Sequence

import numpy as np

from .synth import ModelDims, weight_shapes

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("BLIM_LIB_PATH") or os.path.join(_HERE, "libblim_hip.so")     # override: A/B of two builds (tools/)
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "blim.h")

DTYPE_F32, DTYPE_BF16 = 0, 1
COMPUTE_DTYPES = {"bf16": 0, "f16": 1, "f8": 2}     # "f8": e4m3 operands for the big GEMMs + lm_head, fp16 everywhere else
DEFAULT_COMPUTE_DTYPE = os.environ.get("BLIM_DTYPE", "f16")   # fp16 = the reference's own autocast dtype


def torch_dtype_of(name: str):
    import torch
    return {"bf16": torch.bfloat16, "f16": torch.float16, "f8": torch.float16}[name]


class BlimError(RuntimeError):
    pass


class Config(C.Structure):
    _fields_ = [("vocab_size", C.c_int32), ("hidden_size", C.c_int32), ("intermediate_size", C.c_int32),
                ("num_layers", C.c_int32), ("num_heads", C.c_int32), ("num_kv_heads", C.c_int32),
                ("mm_hidden_size", C.c_int32), ("num_clips", C.c_int32), ("max_positions", C.c_int32),
                ("compute_dtype", C.c_int32), ("rms_eps", C.c_float), ("rope_theta", C.c_float)]


class Batch(C.Structure):
    _fields_ = [("n_tokens", C.c_int64), ("n_seqs", C.c_int32), ("n_blocks", C.c_int32),
                ("positions", C.c_void_p), ("key_visible", C.c_void_p), ("seq_start", C.c_void_p),
                ("seq_len", C.c_void_p), ("pfx_start", C.c_void_p), ("pfx_len", C.c_void_p),
                ("blk_seq", C.c_void_p), ("blk_q0", C.c_void_p), ("own_start", C.c_void_p)]


def declared_symbols(header: str = HEADER_PATH) -> Sequence[str]:
    """Names of every function include/blim.h declares."""
    text = open(header).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(blim_[a-z0-9_]+)\s*\(", text)))


_lib = None


def load_library(path: str = LIB_PATH):
    """Loads the shared library (raises BlimError when it has not been built)."""
    global _lib
    if _lib is not None:
        return _lib
    import torch  # noqa: F401  -- FIRST: torch ships its own HIP runtime; loaded after ours, the process ends up with two and
    #                              the second one sees no device (build() followed by smoke() in one process did exactly that)
    if not os.path.exists(path):
        raise BlimError(f"{path} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                        f"(or `make -C blim_amd/csrc`). The BLiM engine has no CPU fallback.")
    lib = C.CDLL(path)
    vp, i32, i64, u64, f32 = C.c_void_p, C.c_int32, C.c_int64, C.c_uint64, C.c_float
    sig = {
        "blim_abi_version": ([], C.c_int),
        "blim_last_error": ([], C.c_char_p),
        "blim_create": ([C.POINTER(Config), C.POINTER(vp)], C.c_int),
        "blim_destroy": ([vp], None),
        "blim_load_weight": ([vp, C.c_char_p, vp, i32, i32], C.c_int),
        "blim_init_synthetic_weights": ([vp, u64], C.c_int),
        "blim_weights_ready": ([vp], C.c_int),
        "blim_load_adapter": ([vp, C.c_char_p, vp, vp, i32, f32], C.c_int),
        "blim_clear_adapters": ([vp], C.c_int),
        "blim_num_adapters": ([vp], C.c_int),
        "blim_reserve": ([vp, i64, i64], C.c_int),
        "blim_project_video": ([vp, vp, i64, i32, vp, vp], C.c_int),
        "blim_group_mean": ([vp, vp, i64, i32, vp, vp], C.c_int),
        "blim_assemble": ([vp, vp, i64, vp, vp, vp], C.c_int),
        "blim_decode": ([vp, C.POINTER(Batch), vp, vp, i64, vp, vp, vp], C.c_int),
        "blim_vtg_logprobs": ([vp, vp, vp, i64, vp, vp], C.c_int),
        "blim_segment_mean": ([vp, vp, vp, i32, i32, vp, vp], C.c_int),
        "blim_lm_head": ([vp, vp, i64, vp, vp], C.c_int),
        "blim_ce_rows": ([vp, vp, i64, i32, vp, i64, vp, vp], C.c_int),
        "blim_visual_head": ([vp, vp, i64, vp, vp], C.c_int),
        "blim_visual_head_f32": ([vp, vp, i64, vp, vp], C.c_int),
        "blim_tvg_scores": ([vp, vp, vp, i32, vp, i32, vp, vp], C.c_int),
        "blim_tvg_logits": ([vp, vp, vp, i32, i32, vp, vp], C.c_int),
        "blim_set_video_vocab": ([vp, vp, i32, vp], C.c_int),
        "blim_tvg_logits_f32": ([vp, vp, i32, vp, vp], C.c_int),
        "blim_score_vtg": ([vp, C.POINTER(Batch), vp, vp, vp, i64, vp, i32, vp, vp], C.c_int),
        "blim_score_tvg": ([vp, C.POINTER(Batch), vp, vp, vp, i32, vp, i32, vp, vp], C.c_int),
        "blim_forward": ([vp, vp, vp, i32, i32, vp, vp, vp], C.c_int),
        "blim_fill_bell_bf16": ([vp, i64, u64, C.c_char_p, f32, f32, vp], C.c_int),
        "blim_fill_bell_f32": ([vp, i64, u64, C.c_char_p, f32, f32, i32, vp], C.c_int),
        "blim_gemm_bf16": ([vp, i64, vp, i32, i32, i32, vp, i64, vp], C.c_int),
        "blim_gemm_f16": ([vp, i64, vp, i32, i32, i32, vp, i64, vp], C.c_int),
        "blim_f6_tiles_bytes": ([i64, i32], C.c_int64),
        "blim_gemm_f16_lo6": ([vp, vp, i32, i32, i32, vp, vp, vp, vp], C.c_int),
        "blim_quant_rows": ([vp, i64, i64, i32, i32, vp, vp, vp], C.c_int),
        "blim_gemm_f8": ([vp, i64, vp, vp, vp, i32, i32, i32, vp, i64, vp], C.c_int),
        "blim_timing_enable": ([vp, i32], C.c_int),
        "blim_timing_num_classes": ([], C.c_int),
        "blim_timing_class_name": ([i32], C.c_char_p),
        "blim_timing_report": ([vp, C.POINTER(C.c_double), C.POINTER(C.c_int64), C.POINTER(C.c_double)], C.c_int),
        "blim_set_option": ([vp, C.c_char_p, i32], C.c_int),
        "blim_debug_read": ([vp, C.c_char_p, vp, i64, vp], C.c_int),
        "blim_debug_gemm_stamps": ([vp], C.c_int),
    }
    for name, (args, res) in sig.items():
        fn = getattr(lib, name)
        fn.argtypes = args
        fn.restype = res
    _lib = lib
    return lib


def _check(rc: int, what: str):
    if rc != 0:
        raise BlimError(f"{what} failed (code {rc}): {load_library().blim_last_error().decode()}")


def _ptr(t) -> int:
    """data_ptr of a contiguous device tensor (or None)."""
    if t is None:
        return None
    assert t.is_cuda and t.is_contiguous(), "engine buffers must be contiguous device tensors"
    return t.data_ptr()


def _stream() -> int:
    import torch
    return torch.cuda.current_stream().cuda_stream


def vocab_key_of(video_vocab):
    """Identity of a video vocabulary tensor for "is this the one the engine holds?" (literal TVG path): address, shape, torch's in-place version counter AND a content
    fingerprint -- callers pass temporaries such as `vocab.to(dev)`, and the caching allocator hands the next temporary of the same shape the same address, so
    (address, shape) alone let another data set's vocabulary score against a stale registered copy.  The fingerprint is two sums over 4,096 strided values
    (one tiny device -> host copy per call of the literal, non-hot path)."""
    if not hasattr(video_vocab, "data_ptr"):
        return (id(video_vocab),)
    flat = video_vocab.reshape(-1)
    pick = flat[:: max(1, flat.numel() // 4096)][:4096].double()
    fp = (float(pick.sum()), float((pick * pick).sum()))
    return (video_vocab.data_ptr(), tuple(video_vocab.shape), int(getattr(video_vocab, "_version", 0)), fp)


class PackedBatch:
    """Device-side description of packed sequences (blim_batch).  Built from host numpy arrays."""

    def __init__(self, positions: np.ndarray, key_visible: np.ndarray, seq_start: np.ndarray, seq_len: np.ndarray,
                 pfx_start: Optional[np.ndarray] = None, pfx_len: Optional[np.ndarray] = None, device="cuda", own_start: Optional[np.ndarray] = None):
        """own_start (optional, [n_tokens]): first own-segment index (inside its sequence) each token attends to -- segmented sequences
        (blim.h: blim_batch.own_start); None or all zeros = plain causal sequences."""
        import torch
        n_seqs = len(seq_start)
        if pfx_start is None:
            pfx_start = np.zeros(n_seqs, dtype=np.int32)
            pfx_len = np.zeros(n_seqs, dtype=np.int32)
        seq_len = np.asarray(seq_len, dtype=np.int32)
        nblk = (seq_len + 31) // 32
        blk_seq = np.repeat(np.arange(n_seqs, dtype=np.int32), nblk)
        blk_q0 = (np.concatenate([np.arange(n, dtype=np.int32) for n in nblk]) * 32).astype(np.int32) if n_seqs else np.zeros(0, np.int32)
        self.n_tokens = int(len(positions))
        self.max_position = int(np.max(positions)) if len(positions) else 0     # checked against the engine's RoPE table
        self.n_seqs = int(n_seqs)
        self.n_blocks = int(len(blk_seq))
        # one host->device copy for all index arrays
        parts = [np.asarray(positions, np.int32), np.asarray(seq_start, np.int32), seq_len, np.asarray(pfx_start, np.int32),
                 np.asarray(pfx_len, np.int32), blk_seq, blk_q0]
        offs = np.cumsum([0] + [len(p) for p in parts])
        flat = torch.from_numpy(np.concatenate(parts)).to(device)
        self._flat = flat
        self.positions, self.seq_start, self.seq_len, self.pfx_start, self.pfx_len, self.blk_seq, self.blk_q0 = \
            [flat[offs[i]:offs[i + 1]] for i in range(7)]
        self.key_visible = torch.from_numpy(np.ascontiguousarray(key_visible, dtype=np.uint8)).to(device)
        self.own_start = None
        if own_start is not None and np.any(np.asarray(own_start) != 0):
            own_start = np.asarray(own_start, dtype=np.int32)
            assert len(own_start) == self.n_tokens
            self.own_start = torch.from_numpy(np.ascontiguousarray(own_start)).to(device)

    def struct(self, max_positions: Optional[int] = None) -> Batch:
        if max_positions is not None and self.max_position >= max_positions:
            raise BlimError(f"a sequence reaches position {self.max_position} but the engine's RoPE table holds {max_positions} positions: "
                            f"create the engine / BlimModel with a larger max_positions")
        b = Batch()
        b.n_tokens, b.n_seqs, b.n_blocks = self.n_tokens, self.n_seqs, self.n_blocks
        b.positions = self.positions.data_ptr(); b.key_visible = self.key_visible.data_ptr()
        b.seq_start = self.seq_start.data_ptr(); b.seq_len = self.seq_len.data_ptr()
        b.pfx_start = self.pfx_start.data_ptr(); b.pfx_len = self.pfx_len.data_ptr()
        b.blk_seq = self.blk_seq.data_ptr(); b.blk_q0 = self.blk_q0.data_ptr()
        b.own_start = self.own_start.data_ptr() if self.own_start is not None else None
        return b


class Engine:
    """One scoring engine on the current HIP device."""

    def __init__(self, dims: ModelDims, max_positions: int = 4096, dtype: Optional[str] = None):
        import torch
        self.lib = load_library()
        if not torch.cuda.is_available():
            raise BlimError("no HIP device visible: the BLiM engine has no CPU fallback")
        self.dims = dims
        self.max_positions = int(max_positions)
        self.dtype = dtype or DEFAULT_COMPUTE_DTYPE          # "f16" | "bf16": 16-bit format of activations / weights / MFMA operands
        self.torch_dtype = torch_dtype_of(self.dtype)
        cfg = Config(dims.vocab_size, dims.hidden_size, dims.intermediate_size, dims.num_layers, dims.num_heads, dims.num_kv_heads,
                     dims.mm_hidden_size, dims.num_clips, max_positions, COMPUTE_DTYPES[self.dtype], dims.rms_eps, dims.rope_theta)
        h = C.c_void_p()
        _check(self.lib.blim_create(C.byref(cfg), C.byref(h)), "blim_create")
        self.h = h
        self.device = torch.device("cuda", torch.cuda.current_device())
        # blim_create honours BLIM_PRECISE_MLP (A/B runs): the Python-side cache of that option starts from the same value, so that set_precise() neither clobbers
        # an override nor believes in a default the engine does not have
        # mirrors the engine's default of option "precise_lo6" (include/blim.h): the compensated modes' second pass over K in e2m3 on fp16 engines
        # (bf16 engines: off by default -- their parity mode runs the second pass in bf16 -- on with set_option("precise_lo6", 1) / BLIM_PRECISE_LO6=1: round 6)
        lo6_ok = dims.hidden_size % 128 == 0 and dims.intermediate_size % 128 == 0 and max(dims.hidden_size, dims.intermediate_size) <= 20480
        self.lo6 = lo6_ok and ((dtype == "f16" and os.environ.get("BLIM_PRECISE_LO6", "1") != "0") or (dtype == "bf16" and os.environ.get("BLIM_PRECISE_LO6", "0") == "1"))
        self._precise_mlp = os.environ.get("BLIM_PRECISE_MLP", "1") != "0"
        self.weights_version = 0          # bumped by every weight / adapter change: what a measured numeric mode was measured on (modeling.py: resolve_*)

    def close(self):
        if getattr(self, "h", None):
            self.lib.blim_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- weights
    def load_weight(self, name: str, arr: np.ndarray):
        """One tensor: canonical name, float32 numpy array in the checkpoint's natural [out, in] layout."""
        shape = weight_shapes(self.dims)[name]
        assert tuple(arr.shape) == tuple(shape), (name, arr.shape, shape)
        a = np.ascontiguousarray(arr, dtype=np.float32)
        _check(self.lib.blim_load_weight(self.h, name.encode(), a.ctypes.data, DTYPE_F32, 0), f"blim_load_weight({name})")
        self.weights_version += 1

    def load_weights(self, weights: Dict[str, np.ndarray]):
        """weights: canonical name -> float32 numpy array; every tensor of the model must be present."""
        for name, arr in weights.items():
            self.load_weight(name, arr)
        _check(self.lib.blim_weights_ready(self.h), "blim_weights_ready")

    def weights_ready(self) -> bool:
        return self.lib.blim_weights_ready(self.h) == 0

    def init_synthetic_weights(self, seed: int):
        _check(self.lib.blim_init_synthetic_weights(self.h, seed), "blim_init_synthetic_weights")
        self.weights_version += 1

    # ---- LoRA adapters kept apart (blim.h: blim_load_adapter; the reference's --resume flow, main.py:96-105, 125-128)
    def load_adapter(self, weight_name: str, A: np.ndarray, B: np.ndarray, lora_r: int, lora_alpha: float):
        """A [r, in], B [out, r] float32 (peft's lora_A.weight / lora_B.weight) of the adapter on canonical weight `weight_name`."""
        n_out, n_in = weight_shapes(self.dims)[weight_name]
        assert tuple(A.shape) == (lora_r, n_in) and tuple(B.shape) == (n_out, lora_r), (weight_name, A.shape, B.shape, (n_out, n_in), lora_r)
        a = np.ascontiguousarray(A, dtype=np.float32); b = np.ascontiguousarray(B, dtype=np.float32)
        _check(self.lib.blim_load_adapter(self.h, weight_name.encode(), a.ctypes.data, b.ctypes.data, int(lora_r), float(lora_alpha)), f"blim_load_adapter({weight_name})")
        self.weights_version += 1

    def clear_adapters(self):
        _check(self.lib.blim_clear_adapters(self.h), "blim_clear_adapters")
        self.weights_version += 1

    def num_adapters(self) -> int:
        return int(self.lib.blim_num_adapters(self.h))

    def reserve(self, max_tokens: int, max_rows: int, compensated: bool = False):
        """Pre-size the workspaces.  compensated=True (16-bit engines): size them for compensated calls ([hi | lo] rows) and, on a "precise_lo6" engine, build the weights'
        e2m3 images and tile workspaces NOW -- a shortage of device memory is this call's error, not a scoring call's (blim.h: blim_reserve looks at the engine's
        compensated state, which the scoring paths only switch on around each call)."""
        if compensated and self.can_precise:
            self.set_precise(True, embeds=True, mlp=True)
            try:
                _check(self.lib.blim_reserve(self.h, max_tokens, max_rows), "blim_reserve")
            finally:
                self.set_precise(False)
        else:
            _check(self.lib.blim_reserve(self.h, max_tokens, max_rows), "blim_reserve")

    def set_option(self, key: str, value: int):
        _check(self.lib.blim_set_option(self.h, key.encode(), value), "blim_set_option")
        if key == "precise_lo6":
            self.lo6 = self._lo6_live = bool(value)

    @property
    def can_precise(self) -> bool:
        return self.dtype in ("f16", "bf16")

    def set_precise(self, on: bool, embeds: bool = False, mlp: bool = True, tvg: bool = False):
        """Compensated mode for the following calls (16-bit engines; a no-op request on others): every 16-bit activation travels as hi + lo and the GEMMs take
        both parts (fp16 engines: the lo part on the e2m3 MFMA, option "precise_lo6").  The host turns it on for the TVG calls, whose scores are ~10x smaller in
        magnitude than the VTG ones, and for the VTG calls of checkpoints that need it (`--vtg_precise`).
        embeds=True: the input embeddings (assemble -> decode / score_*) are [hi | lo] rows of width 2H as well -- the fused path, whose projected video features
        are produced in this mode; the literal forward() keeps [B, L, H] embeddings.  mlp=False: only the attention branch (QKV, attention, o_proj) and the
        scored rows are compensated -- the TVG calls' "attn" mode.  tvg=True: the call is a TVG call -- on a bf16 engine that was asked for the e2m3 second pass
        (`second_pass = "e2m3"`, round 6) it still takes the bf16 second pass: TVG scores are ~10x smaller in magnitude, read 1.7 - 2.6e-4 with the e2m3 pass against
        1 - 2.4e-5 with the bf16 one at 7B depth, and cost a few percent of an evaluation either way."""
        on = bool(on) and self.can_precise
        embeds = bool(embeds) and on
        want6 = bool(self.lo6) and not (bool(tvg) and on and self.dtype == "bf16")
        if want6 != getattr(self, "_lo6_live", bool(self.lo6)):
            _check(self.lib.blim_set_option(self.h, b"precise_lo6", int(want6)), "blim_set_option")
            self._lo6_live = want6
        if on != getattr(self, "_precise", False):
            self.set_option("precise", int(on))
            self._precise = on
        if embeds != getattr(self, "_precise_embeds", False):
            self.set_option("precise_embeds", int(embeds))
            self._precise_embeds = embeds
        mlp = bool(mlp) and os.environ.get("BLIM_PRECISE_MLP", "1") != "0"
        if on and mlp != getattr(self, "_precise_mlp", True):
            self.set_option("precise_mlp", int(mlp))
            self._precise_mlp = mlp

    # ---- component ops (torch device tensors in/out)
    def project_video(self, feats, which: int):
        import torch
        n = feats.shape[0]
        out = torch.empty((n, self.dims.hidden_size * (2 if getattr(self, "_precise", False) else 1)), dtype=self.torch_dtype, device=self.device)
        _check(self.lib.blim_project_video(self.h, _ptr(feats), n, which, _ptr(out), _stream()), "blim_project_video")
        return out

    def group_mean(self, x, group: int):
        import torch
        n_out = x.shape[0] // group
        out = torch.empty((n_out, x.shape[1]), dtype=self.torch_dtype, device=self.device)
        _check(self.lib.blim_group_mean(self.h, _ptr(x), n_out, group, _ptr(out), _stream()), "blim_group_mean")
        return out

    def assemble(self, src_index, feats=None):
        import torch
        n = src_index.shape[0]
        out = torch.empty((n, self.dims.hidden_size * (2 if getattr(self, "_precise_embeds", False) else 1)), dtype=self.torch_dtype, device=self.device)
        _check(self.lib.blim_assemble(self.h, _ptr(src_index), n, _ptr(feats), _ptr(out), _stream()), "blim_assemble")
        return out

    def decode(self, batch: PackedBatch, embeds, out_rows=None, want_f32=False, want_bf16=True):
        import torch
        n = batch.n_tokens if out_rows is None else out_rows.shape[0]
        H = self.dims.hidden_size
        ob = torch.empty((n, H), dtype=self.torch_dtype, device=self.device) if want_bf16 else None
        of = torch.empty((n, H), dtype=torch.float32, device=self.device) if want_f32 else None
        bs = batch.struct(self.max_positions)
        _check(self.lib.blim_decode(self.h, C.byref(bs), _ptr(embeds), _ptr(out_rows), n, _ptr(ob), _ptr(of), _stream()), "blim_decode")
        return ob, of

    def vtg_logprobs(self, hidden_bf16, labels):
        import torch
        n = hidden_bf16.shape[0]
        out = torch.empty(n, dtype=torch.float32, device=self.device)
        _check(self.lib.blim_vtg_logprobs(self.h, _ptr(hidden_bf16), _ptr(labels), n, _ptr(out), _stream()), "blim_vtg_logprobs")
        return out

    def tvg_logits(self, vh_bf16, vocab_clip_major, n_pairs: int):
        import torch
        out = torch.empty((n_pairs, self.dims.num_clips, vocab_clip_major.shape[1]), dtype=torch.float32, device=self.device)
        _check(self.lib.blim_tvg_logits(self.h, _ptr(vh_bf16), _ptr(vocab_clip_major), vocab_clip_major.shape[1], n_pairs, _ptr(out), _stream()),
               "blim_tvg_logits")
        return out

    def set_video_vocab(self, video_vocab):
        """video_vocab [N, clips, M] (any float dtype, host or device): registered with the engine as hi + lo 16-bit operands (blim_set_video_vocab);
        score_tvg / tvg_scores / tvg_logits called without a vocabulary then use it."""
        import torch
        v = torch.as_tensor(video_vocab).to(device=self.device, dtype=torch.float32).permute(1, 0, 2).contiguous()      # clip-major [clips, N, M]
        assert v.shape[0] == self.dims.num_clips and v.shape[2] == self.dims.mm_hidden_size, tuple(v.shape)
        _check(self.lib.blim_set_video_vocab(self.h, _ptr(v), int(v.shape[1]), _stream()), "blim_set_video_vocab")
        self.n_vocab = int(v.shape[1])
        self._vocab_key = vocab_key_of(video_vocab)

    def tvg_logits_f32(self, vh_f32, n_pairs: int):
        """vh_f32 [n_pairs * clips, M] float32 -> logits [n_pairs, clips, n_vocab] against the registered vocabulary (three-term compensated product)."""
        import torch
        out = torch.empty((n_pairs, self.dims.num_clips, self.n_vocab), dtype=torch.float32, device=self.device)
        _check(self.lib.blim_tvg_logits_f32(self.h, _ptr(vh_f32), n_pairs, _ptr(out), _stream()), "blim_tvg_logits_f32")
        return out

    def lm_head(self, hidden_bf16):
        import torch
        n = hidden_bf16.shape[0]
        out = torch.empty((n, self.dims.vocab_size), dtype=torch.float32, device=self.device)
        _check(self.lib.blim_lm_head(self.h, _ptr(hidden_bf16), n, _ptr(out), _stream()), "blim_lm_head")
        return out

    def visual_head(self, hidden_bf16):
        import torch
        n = hidden_bf16.shape[0]
        out = torch.empty((n, self.dims.mm_hidden_size), dtype=self.torch_dtype, device=self.device)
        _check(self.lib.blim_visual_head(self.h, _ptr(hidden_bf16), n, _ptr(out), _stream()), "blim_visual_head")
        return out

    def visual_head_f32(self, hidden_f32):
        """float32 [n, H] -> float32 [n, M]: head and hidden rows as hi + lo 16-bit operands (blim_visual_head_f32)."""
        import torch
        n = hidden_f32.shape[0]
        out = torch.empty((n, self.dims.mm_hidden_size), dtype=torch.float32, device=self.device)
        _check(self.lib.blim_visual_head_f32(self.h, _ptr(hidden_f32), n, _ptr(out), _stream()), "blim_visual_head_f32")
        return out

    def tvg_scores(self, vh_bf16, vocab_clip_major, labels):
        """vh [n_pairs*clips, M] bf16; vocab [clips, n_vocab, M] bf16; labels [n_pairs] int32."""
        import torch
        n_pairs = labels.shape[0]
        out = torch.empty(n_pairs, dtype=torch.float32, device=self.device)
        _check(self.lib.blim_tvg_scores(self.h, _ptr(vh_bf16), _ptr(vocab_clip_major), vocab_clip_major.shape[1], _ptr(labels), n_pairs,
                                        _ptr(out), _stream()), "blim_tvg_scores")
        return out

    def score_vtg(self, batch: PackedBatch, embeds, rows, labels, row_start):
        import torch
        n_pairs = row_start.shape[0] - 1
        out = torch.empty(n_pairs, dtype=torch.float32, device=self.device)
        bs = batch.struct(self.max_positions)
        _check(self.lib.blim_score_vtg(self.h, C.byref(bs), _ptr(embeds), _ptr(rows), _ptr(labels), rows.shape[0], _ptr(row_start), n_pairs,
                                       _ptr(out), _stream()), "blim_score_vtg")
        return out

    def score_tvg(self, batch: PackedBatch, embeds, rows, vocab_clip_major, labels):
        """vocab_clip_major None: the vocabulary registered with set_video_vocab()."""
        import torch
        n_pairs = labels.shape[0]
        out = torch.empty(n_pairs, dtype=torch.float32, device=self.device)
        bs = batch.struct(self.max_positions)
        n_vocab = self.n_vocab if vocab_clip_major is None else vocab_clip_major.shape[1]
        _check(self.lib.blim_score_tvg(self.h, C.byref(bs), _ptr(embeds), _ptr(rows), _ptr(vocab_clip_major), n_vocab,
                                       _ptr(labels), n_pairs, _ptr(out), _stream()), "blim_score_tvg")
        return out

    def forward(self, embeds, mask, want_logits=True, want_hidden=True):
        """Literal forward: embeds [B,L,H] bf16, mask [B,L] uint8 -> (logits f32 [B,L,V] | None, hidden f32 [B,L,H] | None)."""
        import torch
        B, L, H = embeds.shape
        if getattr(self, "_precise_embeds", False):        # [hi | lo] rows of width 2 * hidden (compensated mode with split embeddings)
            H //= 2
        lg = torch.empty((B, L, self.dims.vocab_size), dtype=torch.float32, device=self.device) if want_logits else None
        hd = torch.empty((B, L, H), dtype=torch.float32, device=self.device) if want_hidden else None
        _check(self.lib.blim_forward(self.h, _ptr(embeds), _ptr(mask), B, L, _ptr(lg), _ptr(hd), _stream()), "blim_forward")
        return lg, hd

    def debug_read(self, which: str, shape, dtype):
        """Copy of an internal workspace as the last decode left it (bring-up aid)."""
        import torch
        out = torch.empty(shape, dtype=dtype, device=self.device)
        _check(self.lib.blim_debug_read(self.h, which.encode(), _ptr(out), out.numel() * out.element_size(), _stream()), "blim_debug_read")
        return out

    # ---- timing
    def timing_enable(self, on: bool):
        _check(self.lib.blim_timing_enable(self.h, int(on)), "blim_timing_enable")

    def timing_report(self) -> Dict[str, Dict[str, float]]:
        n = self.lib.blim_timing_num_classes()
        ms = (C.c_double * n)(); calls = (C.c_int64 * n)(); fl = (C.c_double * n)()
        _check(self.lib.blim_timing_report(self.h, ms, calls, fl), "blim_timing_report")
        return {self.lib.blim_timing_class_name(i).decode(): {"ms": ms[i], "calls": int(calls[i]), "flops": fl[i]} for i in range(n)}


def ce_rows(logits, labels):
    """logprob[r] = log_softmax(logits[r])[labels[r]] (0 where labels[r] < 0); logits f32 [n, V], labels int32 [n]."""
    import torch
    lib = load_library()
    n, v = logits.shape
    out = torch.empty(n, dtype=torch.float32, device=logits.device)
    _check(lib.blim_ce_rows(None, _ptr(logits), v, v, _ptr(labels), n, _ptr(out), _stream()), "blim_ce_rows")
    return out


def segment_mean(logprob, row_start, mode: int = 0):
    """mode 0: sum / count_nonzero (VTG), mode 1: plain mean (TVG)."""
    import torch
    lib = load_library()
    n = row_start.shape[0] - 1
    out = torch.empty(n, dtype=torch.float32, device=logprob.device)
    _check(lib.blim_segment_mean(None, _ptr(logprob), _ptr(row_start), n, mode, _ptr(out), _stream()), "blim_segment_mean")
    return out


def fill_bell_bf16(out, seed: int, name: str, std: float, mean: float = 0.0):
    lib = load_library()
    _check(lib.blim_fill_bell_bf16(_ptr(out), out.numel(), seed, name.encode(), std, mean, _stream()), "blim_fill_bell_bf16")
    return out


def quant_rows(x):
    """x [M,K] bf16/f16 -> (e4m3 bytes as uint8 [M,K], f32 scale [M] = absmax / 448)."""
    import torch
    lib = load_library()
    M, K = x.shape
    q = torch.empty((M, K), dtype=torch.uint8, device=x.device)
    sc = torch.empty(M, dtype=torch.float32, device=x.device)
    _check(lib.blim_quant_rows(_ptr(x), x.stride(0), M, K, 0 if x.dtype == torch.bfloat16 else 1, _ptr(q), _ptr(sc), _stream()), "blim_quant_rows")
    return q, sc


def gemm_f8(a8, a_scale, w8, w_scale):
    """(a8 [M,K] . w8 [N,K]^T) * a_scale[m] * w_scale[n] -> f16 [M,N]; a8 / w8 hold e4m3 bytes (uint8)."""
    import torch
    lib = load_library()
    M, K = a8.shape
    N = w8.shape[0]
    out = torch.empty((M, N), dtype=torch.float16, device=a8.device)
    _check(lib.blim_gemm_f8(_ptr(a8), K, _ptr(a_scale), _ptr(w8), _ptr(w_scale), M, N, K, _ptr(out), N, _stream()), "blim_gemm_f8")
    return out


def gemm_f16_lo6(a_hilo, w):
    """The compensated GEMM of fp16 engines (option "precise_lo6"): a_hilo [M, 2K] f16 rows [hi | lo], w [N, K] f16 -> (C f32 [M, N] = hi . w^T + e2m3(lo) . e2m3(w)^T,
    the e2m3 operand tiles of lo and of w as uint8 [row tiles, K / 128, 25600] -- layout: csrc/gemm.hpp)."""
    import torch
    lib = load_library()
    M, K2 = a_hilo.shape
    N, K = w.shape
    assert K2 == 2 * K and a_hilo.dtype == torch.float16 and w.dtype == torch.float16 and a_hilo.is_contiguous() and w.is_contiguous()
    a6 = torch.empty(lib.blim_f6_tiles_bytes(M, K), dtype=torch.uint8, device=w.device)
    w6 = torch.empty(lib.blim_f6_tiles_bytes(N, K), dtype=torch.uint8, device=w.device)
    out = torch.empty((M, N), dtype=torch.float32, device=w.device)
    _check(lib.blim_gemm_f16_lo6(_ptr(a_hilo), _ptr(w), M, N, K, _ptr(a6), _ptr(w6), _ptr(out), _stream()), "blim_gemm_f16_lo6")
    return out, a6.view(-1, K // 128, 25600), w6.view(-1, K // 128, 25600)


def gemm_bf16(a, w):
    """a [M,K], w [N,K] (both bf16 or both f16) -> [M,N] of the same dtype (plain epilogue)."""
    import torch
    lib = load_library()
    M, K = a.shape
    N = w.shape[0]
    assert a.dtype == w.dtype and a.dtype in (torch.bfloat16, torch.float16)
    out = torch.empty((M, N), dtype=a.dtype, device=a.device)
    fn = lib.blim_gemm_bf16 if a.dtype == torch.bfloat16 else lib.blim_gemm_f16
    _check(fn(_ptr(a), K, _ptr(w), M, N, K, _ptr(out), N, _stream()), "blim_gemm")
    return out
