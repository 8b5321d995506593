"""Host restatement of the library's counter RNG (Philox4x32-10 keyed by the 64-bit seed, counter
words {call lo, call hi, state counter lo, state counter hi}) and of its two token-noise transforms, so
GPU tests can check in-kernel draws value by value rather than only by their moments."""
import numpy as np

M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
W0, W1 = 0x9E3779B9, 0xBB67AE85
MASK = np.uint64(0xFFFFFFFF)


def philox4x32_10(seed: int, call: np.ndarray, counter: int) -> np.ndarray:
    """(len(call), 4) uint32 words of draw `call` of stream (seed, counter)."""
    call = call.astype(np.uint64)
    c = [call & MASK, call >> np.uint64(32),
         np.full_like(call, counter & 0xFFFFFFFF), np.full_like(call, (counter >> 32) & 0xFFFFFFFF)]
    k0, k1 = seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF
    for _ in range(10):
        p0, p1 = M0 * c[0], M1 * c[2]
        hi0, lo0, hi1, lo1 = p0 >> np.uint64(32), p0 & MASK, p1 >> np.uint64(32), p1 & MASK
        c = [hi1 ^ c[1] ^ np.uint64(k0), lo1, hi0 ^ c[3] ^ np.uint64(k1), lo0]
        k0, k1 = (k0 + W0) & 0xFFFFFFFF, (k1 + W1) & 0xFFFFFFFF
    return np.stack(c, axis=1).astype(np.uint32)


def u16(half: np.ndarray) -> np.ndarray:
    """(h + 0.5) / 65536 for a 16-bit half word (exact in float32 and float64 alike)."""
    return (half.astype(np.float64) + 0.5) / 65536.0


def _halves(seed: int, counter: int, n_elements: int):
    """Token noise: draw e >> 3 serves elements e .. e+7; word i of the draw serves elements 2i, 2i+1
    through its low (u) and high (v) 16 bits."""
    calls = np.arange((n_elements + 7) // 8, dtype=np.uint64)
    w = philox4x32_10(seed, calls, counter)           # (calls, 4)
    return w & np.uint32(0xFFFF), w >> np.uint32(16)


def normal(seed: int, counter: int, n_elements: int) -> np.ndarray:
    """Exact (float64) Box-Muller values the Gaussian token-noise kernels approximate."""
    lo, hi = _halves(seed, counter, n_elements)
    u, v = u16(lo), u16(hi)
    r = np.sqrt(-2.0 * np.log(u))
    out = np.stack([r * np.cos(2 * np.pi * v), r * np.sin(2 * np.pi * v)], axis=2)  # (calls, 4, 2)
    return out.reshape(-1)[:n_elements]


def uniform(seed: int, counter: int, n_elements: int) -> np.ndarray:
    lo, hi = _halves(seed, counter, n_elements)
    out = np.stack([2.0 * u16(lo) - 1.0, 2.0 * u16(hi) - 1.0], axis=2)
    return out.reshape(-1)[:n_elements]
