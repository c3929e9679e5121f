"""Excitation and read-out helpers for identification runs (the reference's ``src/utils/signal.py``): multisine inputs for
``FlowSolver.step`` / ``FlowSolver.run`` / ``BatchedFlowSolver.step`` and the dominant frequency of a measured series.  Same
function names and argument meaning; no plotting (matplotlib is not part of this image: ``plot=True`` is accepted and ignored).
"""

from __future__ import annotations

import logging

import numpy as np

logger = logging.getLogger(__name__)


def compute_signal_frequency(sig, Tf: float, dt: float, nzp: int = 10) -> float:
    """Frequency of the largest spectral peak of ``sig`` after dropping the first half of the record (the transient) and the
    mean; the FFT is zero-padded to ``nzp`` times the kept length (signal.py:16-44)."""
    tail = np.asarray(sig, dtype=float)[int((Tf / 2) / dt):]
    tail = tail - tail.mean()
    n_fft = tail.size * nzp
    spectrum = np.abs(np.fft.fft(tail, n_fft))[: n_fft // 2]
    return float(np.argmax(spectrum) / (n_fft * dt))


def sample_lco(Tlco: float, Tstartlco: float, nsim: int) -> np.ndarray:
    """``nsim`` instants spread evenly over one period of a limit cycle that starts at ``Tstartlco`` (signal.py:47-64)."""
    return Tstartlco + (Tlco / nsim) * np.arange(nsim)


def pad_upto(L, N: int, v=0):
    """``L`` (list or 1-D array) extended with ``v`` to ``N`` entries (signal.py:67-75)."""
    if isinstance(L, list):
        return L + [v] * (N - len(L))
    if isinstance(L, np.ndarray):
        return np.concatenate([L, np.full(N - L.shape[0], v, dtype=L.dtype)])
    raise TypeError("Type not supported for padding")


def saturate(x, xmin, xmax):
    """``x`` clipped to [xmin, xmax] (signal.py:78-80)."""
    return min(max(x, xmin), xmax)


def crest_factor(y) -> float:
    """max |y| / rms(y) (signal.py:189-191)."""
    y = np.asarray(y, dtype=float)
    return float(np.abs(y).max() / np.sqrt(np.mean(y * y)))


def multisine(N: int, Fs: float, fmin: float, fmax: float, skip_even: bool = False, opt_cf: int = 0, plot: bool = False,
              include_fbounds: bool = True) -> np.ndarray:
    """One period (``N`` samples at rate ``Fs``) of a multisine: unit-amplitude sines with random phases on the harmonics of
    ``Fs / N`` that fall in [fmin, fmax] · Fs / 2 (open interval when ``include_fbounds`` is False; odd harmonics only with
    ``skip_even``), divided by the square root of their number.  ``opt_cf`` further phase draws are tried and the realisation with
    the smallest crest factor is kept (signal.py:92-160).  Phases come from ``np.random.rand`` (seed with ``np.random.seed``)."""
    f_lo, f_hi = max(fmin, 0.0) * Fs / 2, min(fmax, 1.0) * Fs / 2
    step = 2 if skip_even else 1
    harmonics = np.arange(1 if skip_even else 0, N + (1 if skip_even else 0), step) * Fs / N
    inside = (harmonics >= f_lo) & (harmonics <= f_hi) if include_fbounds else (harmonics > f_lo) & (harmonics < f_hi)
    freqs = harmonics[inside][:, None]
    t = np.linspace(0.0, (N - 1) / Fs, N)

    def draw():
        phases = 2.0 * np.pi * np.random.rand(*freqs.shape)
        return np.sin(2.0 * np.pi * freqs * t + phases).sum(axis=0) / np.sqrt(freqs.shape[0])

    y = draw()
    best = crest_factor(y) if opt_cf else None
    for _ in range(int(opt_cf)):
        cand = draw()
        cf = crest_factor(cand)
        if cf < best:
            y, best = cand, cf
    if plot:
        logger.info("multisine(plot=True): plotting is not available in this package")
    return y


def multisine_MP(M: int, P: int, unwrap: bool = True, **kwargs) -> np.ndarray:
    """``M`` independent multisine realisations (``kwargs`` go to :func:`multisine` and must hold ``N``), each repeated over ``P``
    periods: shape (M, N·P), or flattened when ``unwrap`` (signal.py:163-186)."""
    rows = np.stack([multisine(**kwargs) for _ in range(M)])
    tiled = np.tile(rows, (1, P))
    return tiled.ravel() if unwrap else tiled


__all__ = ["compute_signal_frequency", "sample_lco", "pad_upto", "saturate", "crest_factor", "multisine", "multisine_MP"]
