"""TEST INFRASTRUCTURE ONLY (oracle side) -- counter-based synthetic tensor generator, numpy.

This is the oracle's own statement of the synthetic-data rule that the HIP engine
implements on device (blim_amd/csrc/synth.hip) and the host package mirrors
(blim_amd/synth.py).  Keeping a private copy here lets tests check the three against
each other; nothing under blim_amd/ imports this file.

Rule (bit-exact on any machine: integer hash, one int->f32 convert, one f32 multiply,
round-to-nearest-even to bf16):

    x   = seed*K0 + tensor_id*K1 + index                (mod 2^64)
    z   = splitmix64_finalise(x + K0)
    s   = sum of the four 16-bit fields of z            (0 .. 262140)
    val = f32(s - 131070) * f32(std / SIGMA4) + f32(mean)
    SIGMA4 = sqrt(4 * (65536^2 - 1) / 12)               (std of s)

`val` is an Irwin-Hall(4) bell curve with exactly the requested std; it stands in for the
reference's N(0, initializer_range^2) init (modeling_qwen2_flash.py:835-843).
tensor_id = FNV-1a-64 of the canonical tensor name.
"""
import numpy as np

K0 = np.uint64(0x9E3779B97F4A7C15)
K1 = np.uint64(0xBF58476D1CE4E5B9)
K2 = np.uint64(0x94D049BB133111EB)
SIGMA4 = float(np.sqrt(4.0 * (65536.0 ** 2 - 1.0) / 12.0))


def fnv1a64(name: str) -> int:
    h = 0xCBF29CE484222325
    for b in name.encode("utf-8"):
        h ^= b
        h = (h * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
    return h


def hash_u64(seed: int, tensor_id: int, index: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        x = np.uint64(seed) * K0 + np.uint64(tensor_id) * K1 + index.astype(np.uint64)
        z = x + K0
        z = (z ^ (z >> np.uint64(30))) * K1
        z = (z ^ (z >> np.uint64(27))) * K2
        z = z ^ (z >> np.uint64(31))
    return z


def bell_f32(seed: int, tensor_id: int, n: int, std: float, mean: float = 0.0, start: int = 0) -> np.ndarray:
    """n values of the rule above as float32 (not yet rounded to bf16)."""
    out = np.empty(n, dtype=np.float32)
    scale = np.float32(std / SIGMA4)
    m = np.float32(mean)
    chunk = 1 << 24
    for lo in range(0, n, chunk):
        hi = min(n, lo + chunk)
        z = hash_u64(seed, tensor_id, np.arange(start + lo, start + hi, dtype=np.uint64))
        s = ((z & np.uint64(0xFFFF)) + ((z >> np.uint64(16)) & np.uint64(0xFFFF))
             + ((z >> np.uint64(32)) & np.uint64(0xFFFF)) + (z >> np.uint64(48))).astype(np.int64)
        out[lo:hi] = (s - 131070).astype(np.float32) * scale + m
    return out


def round_bf16(x: np.ndarray) -> np.ndarray:
    """Round float32 to the nearest bf16 (ties to even), returned as float32."""
    u = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32)
    r = (u + np.uint32(0x7FFF) + ((u >> np.uint32(16)) & np.uint32(1))) & np.uint32(0xFFFF0000)
    return r.view(np.float32)


def to_bf16_bits(x: np.ndarray) -> np.ndarray:
    """float32 -> uint16 bf16 bit patterns (RNE)."""
    u = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32)
    return ((u + np.uint32(0x7FFF) + ((u >> np.uint32(16)) & np.uint32(1))) >> np.uint32(16)).astype(np.uint16)


def tensor(seed: int, name: str, shape, std: float, mean: float = 0.0, bf16: bool = True) -> np.ndarray:
    n = int(np.prod(shape))
    v = bell_f32(seed, fnv1a64(name), n, std, mean)
    if bf16:
        v = round_bf16(v)
    return v.reshape(shape)


def token_ids(seed: int, name: str, n: int, lo: int, hi: int) -> np.ndarray:
    """n integers uniform in [lo, hi) (modulo bias ignored; deterministic)."""
    z = hash_u64(seed, fnv1a64(name), np.arange(n, dtype=np.uint64))
    return (np.int64(lo) + (z % np.uint64(hi - lo)).astype(np.int64))
