"""bench.py's N > 1 code path (argument plumbing, barriers, max-over-ranks timing, part_info, the strong-scaling legs of configs
4 / 5 / 3 with their per-rank phase split, the JSON line) under two gloo ranks on the CPU, with the device layer replaced by a stand-in: every rank runs the host
emulation of its device program (``tests/support/nd_numeric.solve_partitioned_reference``: its own sub-tree, the two root
exchanges) plus the third, small all-reduce of a step.  No GPU is involved; what an 8-GPU driver run would exercise besides
is ncclAllReduce itself and the kernels, which the ``-m gpu`` tests cover."""
import contextlib
import io
import json
import os
import socket
import sys
from pathlib import Path
from types import SimpleNamespace

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = Path(__file__).resolve().parents[1]


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


class _FakeSolver:
    """FlowSolver-shaped stand-in: step() = one partitioned solve of a fixed Oseen-like system through gloo."""

    def __init__(self, n, distributed):
        from tests.support import ndsolver
        from flowcontrol_amd.fem.mesh import Mesh
        from flowcontrol_amd.fem.spaces import TaylorHood
        from oracle import ns_oracle as O
        from tests.support import nd_numeric

        self.nd = nd_numeric
        th = TaylorHood(Mesh.unit_square(n, n))
        d = O.Disc.from_taylor_hood(th)
        x = th.node_coords
        U = np.r_[1 + 0.3 * np.sin(x[:, 0]), 0.2 * np.cos(x[:, 1])]
        m = th.mesh
        be = m.boundary_edges()
        be = be[m.edge_midpoints()[be, 0] < 1 - 1e-9]
        nodes = np.unique(np.r_[m.edges[be].reshape(-1), th.nv + be])
        dofs = np.sort(np.r_[nodes, nodes + th.nn])
        self.A, _ = O.apply_bc_symmetric(O.assemble_matrix(d, mass=300.0, nu=0.01, adv=U, lin=U), None, dofs, np.zeros(len(dofs)))
        skip = np.zeros(th.N, bool)
        skip[dofs] = True
        self.world = dist.get_world_size() if distributed else 1
        self.rank = dist.get_rank() if distributed else 0
        p = int(np.log2(self.world))
        self.tree = ndsolver.build_tree(th.cell_dofs, m.cell_centroids(), th.N, 4, skip, merge=2, top_bits=p)
        self.fac = nd_numeric.factorize_blocks(self.A, self.tree)
        self.part = ndsolver.partition(self.fac, self.rank, self.world)
        self.b = np.random.default_rng(1).standard_normal(th.N)
        self.b[dofs] = 0.0
        self.exchanges = 0
        part = self.part
        phases = ("rhs", "up_sweeps", "exchange1", "root", "exchange2", "down_sweeps", "tail", "exchange3", "publish")
        dev = SimpleNamespace(part=part, local_factor_nnz=int(part.seg_len.sum()), _n_factor_values=int(self.fac.vals.size), depth=self.tree.depth,
                              PHASES=phases, refactor_ms={0: 1.0, 1: 1.0}, set_phase_timing=lambda on: None,
                              get_phase_timing=lambda: {**{k: 1.0 + self.rank for k in phases}, "steps": 5},
                              comm_info=lambda: {"nranks": self.world, "rank": self.rank, "transport": "host" if self.world > 1 else "none"},
                              partition_info=lambda: {"rhs_cells": int(part.local_cells.size), "matrix_cells": int(part.local_cells.size), "lead": self.rank == 0,
                                                      "ranks": self.world})
        self.th = SimpleNamespace(N=th.N, nc=th.nc, device=lambda *_: dev, release_device=lambda: None)
        self.y_meas = np.zeros(2)
        self.solve_info = np.array([0.0, 0.0, 0.0, 0.0])
        self.refine_steps = 0
        self.comm, self.distributed = None, distributed
        self.params_control = SimpleNamespace(actuator_number=2)
        self.params_time = SimpleNamespace(dt=0.005)

    def _allreduce(self, a):
        if self.world > 1:
            dist.all_reduce(torch.from_numpy(a))
        self.exchanges += 1

    def step(self, u):
        kind = self.part.rowkind
        mine = (kind == 1) | ((kind == 2) & (self.rank == 0))
        b_local = np.where(mine, self.b, 0.0)  # a rank's share of the right-hand side (root rows: lead rank)
        xp = self.nd.solve_partitioned_reference(self.fac, self.part, b_local[self.tree.perm], self._allreduce)
        x = np.zeros(self.th.N)
        own = (kind[self.tree.perm] == 1) | ((kind[self.tree.perm] == 2) & (self.rank == 0))
        x[self.tree.perm[own]] = xp[own]
        tail = np.r_[x[3], x[11], np.sum(x * x)]  # the third exchange of a step: sensors and norms
        self._allreduce(tail)
        full = x.copy()
        self._allreduce(full)  # test-only: assemble the solution to check it
        self.solve_info[1] = np.linalg.norm(self.A @ full - self.b) / np.linalg.norm(self.b)
        self.y_meas = tail[:2].copy()
        return self.y_meas


def _worker(rank, world, port, out):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                      FC_BENCH_SAME_DEVICE="1")
    sys.path.insert(0, str(ROOT))
    import bench

    torch.cuda.is_available = lambda: True  # bench refuses to run without a GPU; the stand-in below needs none
    torch.cuda.set_device = lambda *_: None
    torch.cuda.synchronize = lambda *_: None
    built = []

    def fake_build(device, distributed=False, refine=None):
        built.append((distributed, refine))
        return _FakeSolver(12 if refine else 8, distributed)

    bench.build_solver = fake_build
    for key in ("config5", "config3"):  # the pinball and cavity legs: the same stand-in behind the real leg driver (run_case)
        bench.CASES[key] = bench.Case(key, f"stand-in for {key}", lambda device, shared, key=key: fake_build(device, True, key),
                                      prepare=(lambda comm, device: comm.bcast("base flow" if comm.rank == 0 else None)) if key == "config3" else None)
    sys.argv = ["bench.py", "--gpus", str(world), "--steps", "3", "--warmup", "2", "--no-extras"]
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        bench.main()
    if rank == 0:
        out["line"] = buf.getvalue().strip()
        out["built"] = built


def test_bench_n2_code_path_under_gloo():
    world = 2
    with mp.Manager() as mgr:
        out = mgr.dict()
        mp.spawn(_worker, args=(world, _free_port(), out), nprocs=world, join=True)
        res = json.loads(out["line"])
        assert res["n_gpus"] == world and res["steps"] == 3 and res["warmup"] == 2
        assert res["scaling"] == "strong" and res["higher_is_better"] is True and res["value"] > 0
        assert abs(res["ms_per_step"] * res["value"] - 1e3) < 1e-6 * 1e3  # strong scaling: value = steps / max time
        part = res["config"]["partition"]
        assert part["exchanges_per_step"] == 3 and part["local_cells"] > 0 and part["root_dofs"] > 0
        assert part["exchange_transport"] == "host" and part["rccl_ranks"] is None  # no RCCL on the CPU: reported as such
        assert part["local_factor_nnz"] < part["stored_factor_nnz"]
        for key in ("strong_scaling_config4", "strong_scaling_config5", "strong_scaling_config3"):
            leg = res[key]
            assert leg is not None and "error" not in leg, leg
            assert leg["n_gpus"] == world and leg["steps_per_s"] > 0 and leg["worst_relative_residual"] < 1e-10 and leg["scaling"] == "strong"
            ph = leg["phase_us"]
            assert ph["phases"][:3] == ["rhs", "up_sweeps", "exchange1"] and len(ph["per_rank"]) == world
            assert ph["per_rank"][0][0] == 1.0 and ph["per_rank"][1][0] == 2.0  # every rank's own figures, gathered
            lp = leg["partition"]
            assert lp["exchanges_per_step"] == 3 and len(lp["local_cells"]) == world and lp["exchange_transport"] == "host"
        assert len(res["phase_us"]["per_rank"]) == world
        assert res["solve_rel_residual_pre_refine"] < 1e-10  # the partitioned emulation really solved the system
        built = [tuple(b) for b in out["built"]]
        assert (True, None) in built and (True, 1) in built and (True, "config5") in built and (True, "config3") in built
