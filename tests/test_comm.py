"""flowcontrol_amd.comm on the CPU: ranks as threads of one process (ThreadComm) — the transport the world = 8 GPU tests and
bench.py's one-GPU rehearsal use — and its error path."""
import threading

import numpy as np
import pytest

from flowcontrol_amd.comm import SingleComm, run_threaded


def _body(comm, n):
    a = np.arange(n, dtype=np.float64) * (comm.rank + 1)
    comm.allreduce(a)
    rows = comm.gather_rows(np.array([comm.rank, 10.0 * comm.rank]))
    word = comm.bcast({"from": comm.rank} if comm.rank == 2 else None, src=2)
    comm.barrier()
    return a, rows, word, comm.allreduce_max(float(-comm.rank))


@pytest.mark.parametrize("world", [2, 8])
def test_thread_ranks_allreduce_gather_bcast(world):
    outs = run_threaded(max(world, 3), _body, 5)
    w = len(outs)
    for r, (a, rows, word, mx) in enumerate(outs):
        assert np.array_equal(a, np.arange(5.0) * w * (w + 1) / 2)
        assert np.array_equal(rows, np.array([[k, 10.0 * k] for k in range(w)]))
        assert word == {"from": 2} and mx == 0.0
    assert all(np.array_equal(o[0], outs[0][0]) for o in outs)  # bit-identical on every rank (fixed summation order)


def test_a_failing_rank_does_not_leave_the_others_waiting():
    def body(comm):
        if comm.rank == 1:
            raise ValueError("rank 1 gives up")
        comm.barrier()  # would wait for ever without the abort
        return "done"

    with pytest.raises(ValueError, match="rank 1 gives up"):
        run_threaded(4, body, timeout=30.0)


def test_single_comm_is_the_identity():
    c = SingleComm()
    a = np.array([1.0, 2.0])
    c.allreduce(a)
    assert c.world == 1 and c.rank == 0 and np.array_equal(a, [1.0, 2.0]) and c.bcast("x") == "x" and c.allreduce_max(3.0) == 3.0
    assert np.array_equal(c.gather_rows(a), a[None])
    assert threading.active_count() >= 1
