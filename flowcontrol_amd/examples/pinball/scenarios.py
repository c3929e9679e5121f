"""Scenarios of BASELINE config 5 (fluidic pinball, Re = 100, ROTATION) shared by the golden-fixture generator, the tests and
``bench.py``: the open-loop Gaussian bumps of the reference's run script and a synthetic closed-loop controller."""
import numpy as np


def pinball_bumps(t: float) -> np.ndarray:
    """Gaussian-bump rotation of the three cylinders, run_pinball_rotation_example.py:101-112 with the
    peaks at 0.05 / 0.10 / 0.15 s and width 0.03 s so that all three act within 50 steps of 0.005 s."""
    tlen, tpeak, u0peak = 0.03, (0.05, 0.10, 0.15), (+2.0, -1.5, -2.0)
    return np.array([u0 * np.exp(-0.5 * (t - tp) ** 2 / tlen**2) for u0, tp in zip(u0peak, tpeak)])


# synthetic stable 3-in / 3-out controller of the closed-loop leg: u = C x, x' = A x + B y
PINBALL_K = dict(
    A=np.array([[-20.0, 2.0, 0.0], [-2.0, -30.0, 1.0], [0.0, -1.0, -40.0]]),
    B=np.eye(3),
    C=2.0e4 * np.array([[40.0, -10.0, 0.0], [5.0, 30.0, -5.0], [0.0, 10.0, -35.0]]),  # probes at x = 8..12 read O(1e-4) in the first 0.25 s
    D=np.zeros((3, 3)),
)
