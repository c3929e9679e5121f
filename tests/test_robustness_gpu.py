"""Failure paths that must be loud: the side stream's gate giving up, a rank that cannot load RCCL."""
import tempfile
from pathlib import Path

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parents[1]


def _cylinder(tmp, nsteps, comm=None):
    from flowcontrol_amd.examples.cylinder.cylinderflowsolver import CylinderFlowSolver
    from flowcontrol_amd.fem.spaces import Function
    from flowcontrol_amd.flowsolverparameters import ParamIC

    g = np.load(ROOT / "tests" / "golden" / "cylinder_O1.npz")
    fs = CylinderFlowSolver.make_default(Re=100, path_out=tmp, num_steps=nsteps)
    fs.comm = comm
    fs.params_ic = ParamIC(xloc=2.0, yloc=0.0, radius=0.5, amplitude=1.0)
    U0, P0 = Function(fs.W, g["UP0"]).split()
    fs._assign_steady_state(U0, P0)
    fs.initialize_time_stepping(ic=None)
    return fs, g


def test_a_side_stream_that_stops_waiting_poisons_its_late_record(monkeypatch, tmp_path):
    """FC_GATE_SPIN=0: the gate of the overlapped tail gives up at once, i.e. residual monitor and energy of the step may have
    read buffers the main stream was still writing.  The late record then carries the flag (inside its checksum) and
    ``fc_step_collect`` reports FC_ERR_HIP instead of handing out the numbers; y of the step is not affected."""
    from flowcontrol_amd import _lib

    monkeypatch.setenv("FC_GATE_SPIN", "0")
    fs, g = _cylinder(tmp_path, 4)
    y = fs.step([0.0, 0.0])
    assert np.allclose(y, g["ol_y"][1], rtol=1e-8, atol=0.0)
    with pytest.raises(_lib.FcError, match="stopped waiting"):
        fs.solve_info
    fs.th.release_device()
    monkeypatch.delenv("FC_GATE_SPIN")
    fs, g = _cylinder(tmp_path, 4)
    fs.step([0.0, 0.0])
    assert fs.solve_info[1] < 1e-12  # the default gate waits: a valid record
    fs.th.release_device()


def test_one_rank_without_rccl_takes_every_rank_to_the_host_exchange(monkeypatch, capfd):
    """FC_RCCL_FAIL_RANK=1: rank 1 cannot load RCCL, rank 0 can.  ncclCommInitRank is a collective, so the ranks settle BEFORE it
    (fc_comm_probe + the process group's own max-reduce) that none of them creates a communicator; all go on over the host exchange."""
    from flowcontrol_amd.comm import ThreadComm, run_threaded

    monkeypatch.setenv("FC_RCCL_FAIL_RANK", "1")
    monkeypatch.setattr(ThreadComm, "in_stream", True, raising=False)  # the ranks believe they are an RCCL process group
    g = np.load(ROOT / "tests" / "golden" / "cylinder_O1.npz")

    def rank_body(comm, nsteps):
        fs, _ = _cylinder(tempfile.mkdtemp(), nsteps, comm)
        for _ in range(nsteps):
            fs.step([0.0, 0.0])
        ts = fs.timeseries
        out = (fs.exchange_fallback, fs.th.device().comm_info()["transport"], ts[[c for c in ts.columns if c.startswith("y_meas")]].to_numpy())
        fs.th.release_device()
        return out

    outs = run_threaded(2, rank_body, 6)
    for why, transport, y in outs:
        assert why is not None and "RCCL" in why and transport == "host", why
        assert np.linalg.norm(y - g["ol_y"][:7]) < 1e-8 * np.linalg.norm(g["ol_y"][:7])
    assert "exchanges go through the host" in capfd.readouterr().err
