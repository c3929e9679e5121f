"""Small helpers user scripts import from the reference's ``utils`` package
(``utils/utils_flowsolver.py`` aggregator: ``flu.apply_fun``, ``flu.MpiUtils``, ``flu.summarize_timings``,
``flu.read_xdmf`` / ``flu.write_xdmf`` as the lid-cavity scripts use them, ``flu.export_subdomains``, ``flu.export_square_operators``,
the ``*_cpp`` predicate builders, ``flu.boundary_force``).  The control-design, frequency-response and eigenvalue helpers the
reference aggregates under the same name need python-control / PETSc / SLEPc and are not part of this package."""

from __future__ import annotations

import logging
import time
from typing import Any, Callable

import numpy as np

from .fem.forces import boundary_force, force_coefficients  # noqa: F401  (flu.* names)
from .dolfin_compat import and_cpp, between_cpp, near_cpp, on_boundary_cpp, or_cpp  # noqa: F401  (C-string predicates of the case files)
from .io import export_sparse_matrix, export_square_operators, export_subdomains, read_xdmf, write_xdmf  # noqa: F401

logger = logging.getLogger(__name__)


def apply_fun(u, fun: Callable[[np.ndarray], Any]) -> Any:
    """Apply a numpy reduction to all DoF values of ``u`` (``utils/fem.py:20-28``)."""
    return fun(u.vector().get_local())


def get_rank() -> int:
    try:
        import torch.distributed as dist

        if dist.is_available() and dist.is_initialized():
            return dist.get_rank()
    except Exception:  # pragma: no cover
        pass
    return 0


def print0(*args: Any, **kwargs: Any) -> None:
    """Log on rank 0 only (``utils/fem.py:30-33``)."""
    if get_rank() == 0:
        logger.info(*args, **kwargs)


def get_subspace_dofs(W) -> dict[str, np.ndarray]:
    """Dof indices of the u, v and p sub-spaces of the mixed space (``utils/fem.py:76-86``); in this package's W layout they are
    the three consecutive blocks [ux | uy | p]."""
    th = W.th
    return {"u": np.arange(th.nn), "v": th.nn + np.arange(th.nn), "p": 2 * th.nn + np.arange(th.nv)}


def mpi_broadcast(x):
    """Identity on one rank; every rank of a multi-GPU run already holds identical ``y_meas``."""
    return x


def peval(f, x):
    """Point evaluation (``utils/mpi.py:22-37``) — one process owns the whole host mirror."""
    return f(x)


class MpiUtils:
    get_rank = staticmethod(get_rank)
    peval = staticmethod(peval)
    mpi_broadcast = staticmethod(mpi_broadcast)


def summarize_timings(fs: Any, t0: float | None = None) -> None:
    """Iteration-1/2/mean step times and time/iter/dof (``utils/fem.py:89-102``)."""
    if fs.iter > 3:
        ts = fs.timeseries
        if t0 is not None:
            logger.info("Total time is: %f", time.time() - t0)
        logger.info("Iteration 1 time     --- %E", ts.loc[1, "runtime"])
        logger.info("Iteration 2 time     --- %E", ts.loc[2, "runtime"])
        logger.info("Mean iteration time  --- %E", np.mean(ts.loc[3:, "runtime"]))
        logger.info("Time/iter/dof        --- %E", np.mean(ts.loc[3:, "runtime"]) / fs.W.dim())
