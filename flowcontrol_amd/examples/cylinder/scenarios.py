"""Benchmark / regression scenarios of the cylinder case that are not part of the reference: BASELINE config 4 (O1 red-refined
once) is driven by this fixed two-actuator signal in the golden fixture, the tests and ``bench.py``."""
import numpy as np


def config4_actuation(n: int) -> np.ndarray:
    k = np.arange(n)
    return np.stack([0.05 * np.sin(0.15 * k), -0.03 * np.cos(0.11 * k)], axis=1)
