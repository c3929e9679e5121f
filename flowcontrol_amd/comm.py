"""Who the ranks of a multi-GPU run are and how their host sides talk.

The reference runs SPMD under ``mpirun`` and leaves the partition to dolfin / PETSc (``src/flowcontrol/flowsolver.py:236-238``);
its own collectives are a MIN-allreduce for point probes (``src/utils/mpi.py:22-37``) and rank-0-only file writes
(``src/flowcontrol/exporter.py:260,266``).  Here the data path's exchanges live inside ``libfc_hip.so`` (RCCL on the solver's
stream, or a host callback); what the Python side needs is small: rank / world, an object broadcast for the RCCL unique id,
a float64 all-reduce for merged field reads, a barrier.

* :class:`TorchComm` — one process per GPU under ``torchrun`` (``torch.distributed``: backend ``nccl`` = RCCL, or ``gloo``).
* :class:`ThreadComm` — the ranks are THREADS of one process, every rank with its own solver handle on the same GPU and the
  exchanges staged through the host.  This is how a ``world = 8`` partition is exercised on a one-GPU box (a box admits only
  a handful of processes on its card): the launch sequence, partition tables and exchange points are those of the 8-GPU run,
  only the all-reduce itself is a sum over the threads' buffers.  ``ctypes`` releases the GIL inside every library call, so
  the ranks' kernels do overlap on the device.
"""
from __future__ import annotations

import threading
from typing import Any, Callable

import numpy as np


class Comm:
    """rank / world and the four host-side collectives the solver uses."""

    rank: int = 0
    world: int = 1
    #: True when the library's exchanges run in-stream over RCCL; False: they are staged through :meth:`allreduce`
    in_stream: bool = False

    def allreduce(self, a: np.ndarray) -> None:  # in-place sum of a float64 array over the ranks
        raise NotImplementedError

    def allreduce_max(self, x: float) -> float:
        raise NotImplementedError

    def bcast(self, obj: Any, src: int = 0) -> Any:
        raise NotImplementedError

    def barrier(self) -> None:
        raise NotImplementedError

    def gather_rows(self, row: np.ndarray) -> np.ndarray:
        """Every rank's float64 vector (same length everywhere) as a [world, n] array, on every rank."""
        row = np.ascontiguousarray(row, dtype=np.float64).reshape(-1)
        out = np.zeros((self.world, row.size))
        out[self.rank] = row
        flat = out.reshape(-1)
        self.allreduce(flat)
        return flat.reshape(self.world, row.size)


class SingleComm(Comm):
    """One rank: every collective is the identity (lets drivers be written once for N = 1 and N > 1)."""

    def allreduce(self, a: np.ndarray) -> None:
        return None

    def allreduce_max(self, x: float) -> float:
        return float(x)

    def bcast(self, obj: Any, src: int = 0) -> Any:
        return obj

    def barrier(self) -> None:
        return None


class TorchComm(Comm):
    """``torch.distributed`` (one process per GPU; ``nccl`` is RCCL on ROCm)."""

    def __init__(self, device_index: int | None = None):
        import torch.distributed as dist

        self._dist = dist
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        self.in_stream = dist.get_backend() == "nccl"
        self._device_index = device_index

    def _tensor(self, a: np.ndarray):
        import torch

        t = torch.from_numpy(a)
        if self.in_stream:
            dev = torch.cuda.current_device() if self._device_index is None else self._device_index
            t = t.to(torch.device("cuda", dev))
        return t

    def allreduce(self, a: np.ndarray) -> None:
        t = self._tensor(a)
        self._dist.all_reduce(t)
        if self.in_stream:
            a[...] = t.cpu().numpy()

    def allreduce_max(self, x: float) -> float:
        a = np.array([float(x)])
        t = self._tensor(a)
        self._dist.all_reduce(t, op=self._dist.ReduceOp.MAX)
        return float(t.cpu().numpy()[0])

    def bcast(self, obj: Any, src: int = 0) -> Any:
        box = [obj]
        self._dist.broadcast_object_list(box, src=src)
        return box[0]

    def barrier(self) -> None:
        self._dist.barrier()


class _ThreadGroup:
    def __init__(self, world: int, timeout: float):
        self.world = world
        self.timeout = timeout
        self.barrier = threading.Barrier(world)
        self.slots: list[Any] = [None] * world


class ThreadComm(Comm):
    """Rank ``rank`` of ``world`` ranks that are threads of this process (see the module docstring).  Sums run in rank order
    on every rank, so all ranks hold bit-identical results."""

    def __init__(self, group: _ThreadGroup, rank: int):
        self._g = group
        self.rank, self.world = rank, group.world

    @staticmethod
    def group(world: int, timeout: float = 600.0) -> list["ThreadComm"]:
        g = _ThreadGroup(world, timeout)
        return [ThreadComm(g, r) for r in range(world)]

    def _wait(self) -> None:
        self._g.barrier.wait(self._g.timeout)

    def _exchange(self, mine: Any) -> list[Any]:
        g = self._g
        g.slots[self.rank] = mine
        self._wait()
        seen = list(g.slots)
        self._wait()  # nobody overwrites a slot before every rank has read it
        return seen

    def allreduce(self, a: np.ndarray) -> None:
        parts = self._exchange(np.array(a, dtype=np.float64, copy=True))
        total = parts[0].copy()
        for p in parts[1:]:
            total += p
        a[...] = total

    def allreduce_max(self, x: float) -> float:
        return max(self._exchange(float(x)))

    def bcast(self, obj: Any, src: int = 0) -> Any:
        return self._exchange(obj if self.rank == src else None)[src]

    def barrier(self) -> None:
        self._wait()

    def abort(self) -> None:
        self._g.barrier.abort()


def run_threaded(world: int, fn: Callable[..., Any], *args, timeout: float = 600.0) -> list[Any]:
    """``fn(comm, *args)`` on ``world`` threads, one :class:`ThreadComm` each; returns the ranks' results in rank order.  An
    exception on one rank breaks the group's barrier (the other ranks then fail on their next collective instead of waiting
    for ever) and is re-raised here."""
    comms = ThreadComm.group(world, timeout)
    results: list[Any] = [None] * world
    errors: list[BaseException | None] = [None] * world

    def body(r: int) -> None:
        try:
            results[r] = fn(comms[r], *args)
        except BaseException as err:  # noqa: BLE001
            errors[r] = err
            comms[r].abort()

    threads = [threading.Thread(target=body, args=(r,), name=f"fc-rank-{r}") for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    first = next((e for e in errors if e is not None and not isinstance(e, threading.BrokenBarrierError)), None)
    first = first or next((e for e in errors if e is not None), None)
    if first is not None:
        raise first
    return results


def default_comm() -> Comm | None:
    """The process group of this process, if ``torch.distributed`` is initialised with more than one rank."""
    import sys

    if "torch" not in sys.modules:
        return None  # nobody in this process can have initialised torch.distributed: do not pay the import
    try:
        import torch.distributed as dist
    except Exception:  # pragma: no cover
        return None
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return None
    return TorchComm()


__all__ = ["Comm", "SingleComm", "TorchComm", "ThreadComm", "run_threaded", "default_comm"]
