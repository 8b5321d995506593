"""Host restatement of the library's counter RNG (Philox4x32-10 keyed by the 64-bit seed, counter
words {call lo, call hi, state counter lo, state counter hi}) and of its two noise transforms, so
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


def u01(words: np.ndarray) -> np.ndarray:
    """(r + 0.5) * 2^-32, evaluated as the kernels do it: the word converted to float32 first."""
    return (words.astype(np.float32) + np.float32(0.5)).astype(np.float64) * 2.0 ** -32


def normal(seed: int, counter: int, n_elements: int) -> np.ndarray:
    """Exact (float64) Box-Muller values the Gaussian noise kernels approximate: element e takes
    value e & 3 of draw e >> 2; values 0,1 come from words 0,1 and values 2,3 from words 2,3."""
    calls = np.arange((n_elements + 3) // 4, dtype=np.uint64)
    w = philox4x32_10(seed, calls, counter)
    out = np.empty((len(calls), 4))
    for a, b, o in ((0, 1, 0), (2, 3, 2)):
        u, v = u01(w[:, a]), u01(w[:, b])
        r = np.sqrt(-2.0 * np.log(u))
        out[:, o] = r * np.cos(2 * np.pi * v)
        out[:, o + 1] = r * np.sin(2 * np.pi * v)
    return out.reshape(-1)[:n_elements]


def uniform(seed: int, counter: int, n_elements: int) -> np.ndarray:
    calls = np.arange((n_elements + 3) // 4, dtype=np.uint64)
    w = philox4x32_10(seed, calls, counter)
    return (2.0 * u01(w) - 1.0).reshape(-1)[:n_elements]
