"""Renyi-DP accountant for the Poisson-subsampled Gaussian mechanism.

Serves `privacy_engine.get_epsilon(delta)` (train.py:209, 244 of the reference) and the sigma
search behind `make_private_with_epsilon` (train.py:144-153).  The reference obtains both from
Opacus 1.4.0; this is a restatement of the published analysis (Mironov, Talwar, Zhang 2019,
"Renyi Differential Privacy of the Sampled Gaussian Mechanism", Sec. 3.3) with the
RDP -> (eps, delta) conversion of Balle et al. 2020 (Thm. 21), in log-space double arithmetic.
Host-side only: a few hundred scalar evaluations per query.
"""
from __future__ import annotations

import math
from typing import List, Sequence, Tuple

DEFAULT_ALPHAS: List[float] = [1 + x / 10.0 for x in range(1, 100)] + [float(x) for x in range(12, 64)]


def _log_add(a: float, b: float) -> float:
    if a == -math.inf:
        return b
    if b == -math.inf:
        return a
    hi, lo = (a, b) if a > b else (b, a)
    return hi + math.log1p(math.exp(lo - hi))


def _log_sub(a: float, b: float) -> float:
    if b == -math.inf:
        return a
    if a == b:
        return -math.inf
    if a < b:
        raise ValueError("log-space subtraction would be negative")
    return a + math.log1p(-math.exp(b - a))


def _log_erfc(x: float) -> float:
    r = math.erfc(x)
    if r > 0.0:
        return math.log(r)
    # asymptotic expansion for large x where erfc underflows
    return (-math.log(math.pi) / 2 - math.log(x) - x * x - 0.5 * x ** -2 + 0.625 * x ** -4
            - 37.0 / 24.0 * x ** -6 + 353.0 / 64.0 * x ** -8)


def _log_binom(n: float, k: int) -> Tuple[float, float]:
    """(log |C(n,k)|, sign) for real n, integer k >= 0."""
    sign, logv = 1.0, 0.0
    for j in range(k):
        term = (n - j) / (j + 1.0)
        if term == 0.0:
            return -math.inf, 0.0
        if term < 0:
            sign = -sign
        logv += math.log(abs(term))
    return logv, sign


def _log_a_int(q: float, sigma: float, alpha: int) -> float:
    log_a = -math.inf
    for i in range(alpha + 1):
        log_c, _ = _log_binom(alpha, i)
        term = log_c + i * math.log(q) + (alpha - i) * math.log1p(-q) + (i * i - i) / (2.0 * sigma ** 2)
        log_a = _log_add(log_a, term)
    return log_a


def _log_a_frac(q: float, sigma: float, alpha: float) -> float:
    log_a0, log_a1 = -math.inf, -math.inf
    z0 = sigma ** 2 * math.log(1.0 / q - 1.0) + 0.5
    i = 0
    while True:
        log_c, sign = _log_binom(alpha, i)
        j = alpha - i
        log_t0 = log_c + i * math.log(q) + j * math.log1p(-q)
        log_t1 = log_c + j * math.log(q) + i * math.log1p(-q)
        log_e0 = math.log(0.5) + _log_erfc((i - z0) / (math.sqrt(2.0) * sigma))
        log_e1 = math.log(0.5) + _log_erfc((z0 - j) / (math.sqrt(2.0) * sigma))
        log_s0 = log_t0 + (i * i - i) / (2.0 * sigma ** 2) + log_e0
        log_s1 = log_t1 + (j * j - j) / (2.0 * sigma ** 2) + log_e1
        if sign > 0:
            log_a0, log_a1 = _log_add(log_a0, log_s0), _log_add(log_a1, log_s1)
        elif sign < 0:
            log_a0, log_a1 = _log_sub(log_a0, log_s0), _log_sub(log_a1, log_s1)
        i += 1
        if max(log_s0, log_s1) < -30.0 or sign == 0.0:
            break
    return _log_add(log_a0, log_a1)


def compute_rdp(q: float, noise_multiplier: float, steps: int, alphas: Sequence[float] = DEFAULT_ALPHAS
                ) -> List[float]:
    """RDP epsilon at each order after `steps` compositions of the sampled Gaussian."""
    out = []
    for alpha in alphas:
        if q == 0.0:
            r = 0.0
        elif noise_multiplier == 0.0:
            r = math.inf
        elif q == 1.0:
            r = alpha / (2.0 * noise_multiplier ** 2)
        elif math.isinf(alpha):
            r = math.inf
        else:
            la = _log_a_int(q, noise_multiplier, int(alpha)) if float(alpha).is_integer() \
                else _log_a_frac(q, noise_multiplier, alpha)
            r = la / (alpha - 1.0)
        out.append(r * steps)
    return out


def rdp_to_epsilon(rdp: Sequence[float], alphas: Sequence[float], delta: float) -> Tuple[float, float]:
    """min over orders of  rdp - (log delta + log alpha)/(alpha-1) + log((alpha-1)/alpha)."""
    best, best_alpha = math.inf, float("nan")
    for r, a in zip(rdp, alphas):
        if a <= 1.0 or math.isinf(r) or math.isnan(r):
            continue
        eps = r - (math.log(delta) + math.log(a)) / (a - 1.0) + math.log((a - 1.0) / a)
        if eps < best:
            best, best_alpha = eps, a
    return max(best, 0.0), best_alpha


class RDPAccountant:
    """Accumulates (noise_multiplier, sample_rate, steps) runs; `get_epsilon(delta)` composes them."""

    def __init__(self):
        self.history: List[Tuple[float, float, int]] = []

    @classmethod
    def mechanism(cls) -> str:
        return "rdp"

    def step(self, *, noise_multiplier: float, sample_rate: float) -> None:
        if self.history and self.history[-1][0] == noise_multiplier and self.history[-1][1] == sample_rate:
            nm, sr, n = self.history[-1]
            self.history[-1] = (nm, sr, n + 1)
        else:
            self.history.append((noise_multiplier, sample_rate, 1))

    def __len__(self):
        return sum(n for _, _, n in self.history)

    def get_epsilon(self, delta: float, alphas: Sequence[float] = DEFAULT_ALPHAS) -> float:
        if not self.history:
            return 0.0
        total = [0.0] * len(alphas)
        for nm, sr, n in self.history:
            for k, r in enumerate(compute_rdp(sr, nm, n, alphas)):
                total[k] += r
        return rdp_to_epsilon(total, alphas, delta)[0]


def get_noise_multiplier(*, target_epsilon: float, target_delta: float, sample_rate: float, steps: int,
                         epsilon_tolerance: float = 0.01, accountant: str = "rdp") -> float:
    """Noise multiplier whose epsilon after `steps` steps lands within `epsilon_tolerance` below
    target_epsilon.  The bracketing and bisection follow Opacus 1.4.0's search step for step
    (upper bracket 10 doubled BEFORE its first evaluation, loop on the epsilon gap only, upper
    end returned), so the same target yields the same sigma; parity unpinned (Opacus absent)."""
    if accountant == "prv":
        from .prv import PRVAccountant

        def eps_of(sigma: float) -> float:
            acc = PRVAccountant()
            acc.history = [(sigma, sample_rate, steps)]
            return acc.get_epsilon(target_delta)
    elif accountant == "rdp":
        def eps_of(sigma: float) -> float:
            return rdp_to_epsilon(compute_rdp(sample_rate, sigma, steps), DEFAULT_ALPHAS, target_delta)[0]
    else:
        raise ValueError(f"unknown accountant {accountant!r}")

    eps_high = math.inf
    sigma_low, sigma_high = 0.0, 10.0
    while eps_high > target_epsilon:
        sigma_high *= 2.0
        eps_high = eps_of(sigma_high)
        if sigma_high > 1e6:
            raise ValueError("the privacy budget is too low")
    while target_epsilon - eps_high > epsilon_tolerance:
        sigma = (sigma_low + sigma_high) / 2.0
        eps = eps_of(sigma)
        if eps < target_epsilon:
            sigma_high, eps_high = sigma, eps
        else:
            sigma_low = sigma
    return sigma_high
