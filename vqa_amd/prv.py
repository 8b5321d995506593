"""PRV (privacy loss random variable) accountant for the Poisson-subsampled Gaussian mechanism.

`PrivacyEngine()` in the reference (train.py:127) is Opacus 1.4.0's, whose default accountant is
"prv": the epsilon that train.py:209 logs and the sigma that make_private_with_epsilon
(train.py:144-153) searches for both come from it.  Opacus is absent here, so this restates the
published algorithm (Gopi, Lee, Wutschitz, "Numerical Composition of Differential Privacy",
NeurIPS 2021, Alg. 1-2): discretise the privacy loss distribution of one step on a uniform mesh
(mean-preserving), compose by FFT powers, read epsilon off the composed distribution with the
paper's error terms.  Parity with Opacus' numbers is unpinned; what is checked is the mathematics
(closed form for the unsubsampled Gaussian, dominance by the RDP bound, error interval).

Host-side numpy / scipy only.
"""
from __future__ import annotations

import math
from typing import List, Sequence, Tuple

import numpy as np
from scipy import integrate
from scipy.special import ndtr

from .accountant import RDPAccountant


class SubsampledGaussianPRV:
    """Privacy loss Y = log(P/Q)(X), X ~ P, of the dominating pair P = (1-q) N(0, s^2) + q N(1, s^2),
    Q = N(0, s^2) (s = noise multiplier, q = sample rate)."""

    def __init__(self, sample_rate: float, noise_multiplier: float):
        self.q, self.s = float(sample_rate), float(noise_multiplier)

    def cdf(self, t):
        """P(Y <= t).  Y <= t  <=>  x <= s^2 z + 1/2 with z = log((e^t + q - 1) / q); the loss is
        bounded below by log(1 - q)."""
        t = np.asarray(t, dtype=np.float64)
        q, s = self.q, self.s
        if q >= 1.0:
            return ndtr(s * t + 0.5 / s - 1.0 / s)  # x ~ N(1, s^2): Y = (2x - 1) / (2 s^2)
        lo = math.log1p(-q)
        inside = t > lo
        tt = np.where(inside, t, 0.0)
        # log((e^t + q - 1) / q), written so that large t does not overflow
        z = np.where(tt > 30.0, tt - math.log(q) + np.log1p((q - 1.0) * np.exp(-np.minimum(tt, 700.0))),
                     np.log(np.maximum((np.exp(np.minimum(tt, 30.0)) + q - 1.0) / q, 1e-300)))
        val = (1.0 - q) * ndtr(s * z + 0.5 / s) + q * ndtr(s * z - 0.5 / s)
        return np.where(inside, val, 0.0)


class _Truncated:
    """The loss conditioned on [t_min, t_max]."""

    def __init__(self, prv: SubsampledGaussianPRV, t_min: float, t_max: float):
        self.prv, self.t_min, self.t_max = prv, t_min, t_max
        self.c_lo = float(prv.cdf(t_min))
        self.mass = float(prv.cdf(t_max)) - self.c_lo

    def cdf(self, t):
        t = np.asarray(t, dtype=np.float64)
        v = (self.prv.cdf(np.clip(t, self.t_min, self.t_max)) - self.c_lo) / self.mass
        return np.where(t < self.t_min, 0.0, np.where(t >= self.t_max, 1.0, v))

    def mean(self) -> float:
        """E[Y] = [t F(t)] - integral of F, split around zero where the cdf bends most."""
        pts = np.concatenate([[self.t_min], -np.logspace(-5, -1, 5)[::-1], np.logspace(-5, -1, 5), [self.t_max]])
        pts = np.unique(np.clip(pts, self.t_min, self.t_max))
        m = 0.0
        for a, b in zip(pts[:-1], pts[1:]):
            area, _ = integrate.quad(lambda x: float(self.cdf(x)), a, b, limit=500)
            m += b * float(self.cdf(b)) - a * float(self.cdf(a)) - area
        return m


class _Discrete:
    """pmf on the mesh t_i = t0 + i dt + shift, i in [0, size); size even, index size/2 is t = shift."""

    def __init__(self, pmf: np.ndarray, t0: float, dt: float, shift: float):
        self.pmf, self.t0, self.dt, self.shift = pmf, t0, dt, shift

    def ts(self) -> np.ndarray:
        return self.t0 + self.dt * np.arange(len(self.pmf)) + self.shift

    def self_compose(self, n: int) -> "_Discrete":
        """n-fold convolution by FFT power.  The n-fold sum of mesh points t0 + i dt is
        n t0 + (sum i) dt; on the same mesh that is index sum i + (n - 1) t0 / dt = sum i - (n - 1) size / 2,
        and the circular transform already reduces indices mod size: a half turn is left when n - 1 is odd.
        Mass that wraps around is the tail mass the mesh was sized to make negligible."""
        size = len(self.pmf)
        out = np.fft.irfft(np.fft.rfft(self.pmf) ** n, n=size)
        if (n - 1) % 2 == 1:
            out = np.roll(out, size // 2)
        return _Discrete(out, self.t0, self.dt, self.shift * n)

    def compose(self, other: "_Discrete") -> "_Discrete":
        size = len(self.pmf)
        assert len(other.pmf) == size and other.dt == self.dt and other.t0 == self.t0
        out = np.fft.irfft(np.fft.rfft(self.pmf) * np.fft.rfft(other.pmf), n=size)
        out = np.roll(out, size // 2)
        return _Discrete(out, self.t0, self.dt, self.shift + other.shift)

    def epsilon(self, delta: float, delta_error: float, eps_error: float) -> Tuple[float, float, float]:
        """(lower, estimate, upper): delta(eps) = sum_{t_j > eps} pmf_j (1 - e^{eps - t_j}), inverted on the mesh."""
        if np.finfo(np.float64).eps * len(self.pmf) > delta - delta_error:
            raise ValueError("floating-point error would dominate: raise delta_error or lower the mesh size")
        t = self.ts()
        pmf = np.maximum(self.pmf, 0.0)
        tail = np.flip(np.cumsum(np.flip(pmf)))
        wtail = np.flip(np.cumsum(np.flip(pmf * np.exp(-t))))
        deltas = tail - np.exp(t) * wtail

        def find(target: float) -> float:
            i = int(np.searchsorted(-deltas, -target, side="left"))
            if i <= 0:
                raise ValueError("epsilon lies outside the mesh")
            if i >= len(t):
                return float(t[-1])
            return float(math.log((tail[i] - target) / wtail[i]))

        return find(delta + delta_error) - eps_error, find(delta), find(delta - delta_error) + eps_error


def _rdp_eps(history: Sequence[Tuple[float, float, int]], delta: float) -> float:
    acc = RDPAccountant()
    acc.history = list(history)
    return acc.get_epsilon(delta)


class PRVAccountant:
    """Same surface as RDPAccountant: `step(noise_multiplier=, sample_rate=)`, `history`, `get_epsilon(delta)`.
    `get_epsilon` returns the UPPER end of the error interval (estimate + eps_error, delta - delta_error), as
    Opacus' accountant does: a valid bound, at most ~2 eps_error above the true value."""

    def __init__(self):
        self.history: List[Tuple[float, float, int]] = []

    @classmethod
    def mechanism(cls) -> str:
        return "prv"

    def step(self, *, noise_multiplier: float, sample_rate: float) -> None:
        if self.history and self.history[-1][0] == noise_multiplier and self.history[-1][1] == sample_rate:
            nm, sr, n = self.history[-1]
            self.history[-1] = (nm, sr, n + 1)
        else:
            self.history.append((noise_multiplier, sample_rate, 1))

    def __len__(self):
        return sum(n for _, _, n in self.history)

    def _mesh(self, eps_error: float, delta_error: float) -> Tuple[float, float, int]:
        """Half width L and mesh size dt (Gopi et al., Thm. 5.5): the composed loss leaves [-L, L] with
        probability below delta_error (bounded through RDP), and dt keeps the discretisation error of
        epsilon below eps_error after all compositions."""
        total = sum(n for _, _, n in self.history)
        L = eps_error + _rdp_eps(self.history, delta_error / 4.0)
        for nm, sr, _n in self.history:
            L = max(L, _rdp_eps([(nm, sr, 1)], delta_error / (8.0 * total)))
        L += 3.0
        dt = 2.0 * eps_error / math.sqrt(2.0 * total * math.log(12.0 / delta_error))
        half = int(math.ceil(L / dt))
        return half * dt, dt, 2 * half

    def composed(self, eps_error: float, delta_error: float) -> _Discrete:
        L, dt, size = self._mesh(eps_error, delta_error)
        t0 = -L
        centres = t0 + dt * np.arange(size)
        total = None
        for nm, sr, n in self.history:
            tr = _Truncated(SubsampledGaussianPRV(sr, nm), -L, L)
            pmf = tr.cdf(centres + dt / 2.0) - tr.cdf(centres - dt / 2.0)
            shift = tr.mean() - float(np.dot(centres, pmf))  # mean-preserving placement of the mesh
            if abs(shift) >= dt / 2.0:
                raise ValueError("discretised mean differs from the continuous one by more than half a mesh step")
            part = _Discrete(pmf, t0, dt, shift).self_compose(n)
            total = part if total is None else total.compose(part)
        return total

    def get_epsilon(self, delta: float, *, eps_error: float = 0.01, delta_error: float = None) -> float:
        if not self.history:
            return 0.0
        if delta_error is None:
            delta_error = delta / 1000.0
        return self.composed(eps_error, delta_error).epsilon(delta, delta_error, eps_error)[2]

    def get_epsilon_interval(self, delta: float, *, eps_error: float = 0.01, delta_error: float = None
                             ) -> Tuple[float, float, float]:
        if delta_error is None:
            delta_error = delta / 1000.0
        return self.composed(eps_error, delta_error).epsilon(delta, delta_error, eps_error)
