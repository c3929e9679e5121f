"""Scenario of BASELINE config 3 (open cavity, Re = 7500, closed loop): the reference ships no cavity controller, so the golden
fixture, the tests and ``bench.py`` share this documented synthetic stable SISO controller (first-order low-pass with gain)."""

CAVITY_K = dict(A=[[-100.0]], B=[[1.0]], C=[[50.0]], D=[[0.0]])
