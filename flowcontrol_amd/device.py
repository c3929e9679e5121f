"""Host-side driver of one ``fc_handle``: numpy in, numpy out, everything else on the MI355X.

This is the only module that talks to ``libfc_hip.so``.  ``FlowSolver`` (the mirror of the
reference's ``src/flowcontrol/flowsolver.py``) owns one :class:`DeviceSolver`.
"""

from __future__ import annotations

import ctypes as C
import logging
import os

import numpy as np
import scipy.sparse as sp

from . import _lib
from ._lib import SLOT_BDF1, SLOT_BDF2, SLOT_MASS, SLOT_SCRATCH, check, ptr
from .fem.spaces import TaylorHood


def _f64(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.float64)


def _i32(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.int32)


logger = logging.getLogger(__name__)


class DeviceSolver:
    def __init__(self, th: TaylorHood, device: int = 0):
        self.lib = _lib.load()
        if _lib.device_count() <= 0:
            raise _lib.FcError(_lib.FC_ERR_HIP, "no HIP device visible: the MI355X path has no CPU fallback")
        self.th = th
        m = th.mesh
        self._h = C.c_void_p()
        check(
            self.lib.fc_create(
                C.byref(self._h), device, m.num_vertices, m.num_edges, m.num_cells, _f64(m.coords), _i32(m.cells), _i32(m.cell_edges)
            )
        )
        N, nnz, nn = C.c_int64(), C.c_int64(), C.c_int64()
        check(self.lib.fc_get_sizes(self._h, C.byref(N), C.byref(nnz), C.byref(nn)))
        self.N, self.nnz, self.nn = N.value, nnz.value, nn.value
        assert self.N == th.N and self.nn == th.nn
        self.rowptr = np.empty(self.N + 1, dtype=np.int32)
        self.colidx = np.empty(self.nnz, dtype=np.int32)
        check(self.lib.fc_get_pattern(self._h, self.rowptr, self.colidx))
        self.n_act = 0
        self.n_sens = 0
        self.perm: np.ndarray | None = None  # elimination ordering (new position -> W dof)
        self.factor_nnz: dict[int, int] = {}
        self.rank, self.world = 0, 1
        self.part = None  # multi-GPU: this rank's share (rowkind, local cells, exchange stages) as the library laid it out
        self._sensor_rows: list | None = None
        # the whole solver setup -- symbolic phase (tree, factor layout, elimination plan, sweep tables: csrc/fc_symbolic.hpp) and numeric
        # factorisation -- runs inside the library (fc_setup_solver).  The numpy specification the symbolic phase is compared with lives
        # with the tests (tests/support/ndsolver.py, tests/test_symbolic_cabi.py).
        self._tree_args = None
        self._structured: set[int] = set()
        self.refactor_ms: dict[int, float] = {}
        #: slot -> the factors missed the acceptance residual by the direct apply and serve as GMRES preconditioner (fc_accept_factors)
        self.factors_inexact: dict[int, bool] = {}
        self._probe: np.ndarray | None = None
        self._pin_shift = 1.0
        self._step_bufs = None
        self._pin: int | None = None  # pressure dof of an enclosed flow whose level is fixed (diagonal shift in the factors)
        self._truncate = 0  # > 0: only the tree levels >= this are factorised (memory-lean preconditioner)
        self.device_index = device
        self.batch_k = 0
        self._batch_bufs = None
        self._xchg_error: BaseException | None = None

    # ── multi-GPU ────────────────────────────────────────────────────────────
    def join(self, rank: int, world: int, broadcast_bytes, host_allreduce=None, agree=None) -> None:
        """Make this handle rank ``rank`` of a ``world``-GPU run (one process per GPU).

        ``broadcast_bytes(b | None) -> bytes`` ships the 128-byte RCCL unique id from rank 0 to every
        rank (e.g. ``torch.distributed.broadcast_object_list``); the communicator then lives inside
        the library and its all-reduces run on the solver's own HIP stream.

        ``agree(flag: float) -> float`` (the max of ``flag`` over the ranks, through the caller's own process group): with it the
        ranks settle TOGETHER whether the RCCL communicator exists — every rank probes RCCL before the collective
        ``ncclCommInitRank`` and reports after it; on any failure every rank raises :class:`FcCommInitError` (a rank that did get a
        communicator gives it back first), so that all of them can take the host exchange instead of some waiting in a collective
        for a rank that never comes.
        """
        if world & (world - 1):
            raise ValueError("world size must be a power of two (one elimination sub-tree per GPU)")
        self.rank, self.world = int(rank), int(world)
        self._host_allreduce = host_allreduce
        # FC_FORCE_COMM=1 (test aid): build a 1-rank RCCL communicator and run the partitioned code
        # path (cell list, row kinds, in-stream all-reduces) on a single GPU
        self._force_comm = world == 1 and os.environ.get("FC_FORCE_COMM", "0") == "1"
        if world == 1 and not self._force_comm:
            return
        # another root for the elimination tree: whatever was set up for the single-GPU role is void
        self.perm, self.part = None, None
        self._structured.clear()
        if host_allreduce is not None:
            # exchange through the host (fc_set_host_exchange): no RCCL communicator; ``host_allreduce(array)`` sums a
            # float64 array over the ranks in place.  The launch sequence is the one of the RCCL path.
            def _cb(ptr, n, _user):
                # ctypes swallows an exception raised in a callback: park it, poison the buffer (the residual / divergence
                # checks of the step then trip instead of continuing on un-reduced sums) and re-raise after the C call
                arr = np.ctypeslib.as_array(ptr, shape=(int(n),))
                try:
                    host_allreduce(arr)
                except BaseException as err:  # noqa: BLE001
                    self._xchg_error = err
                    arr[:] = np.nan

            self._xchg_cb = _lib.EXCHANGE_FN(_cb)  # keep the callback object alive as long as the handle
            check(self.lib.fc_set_host_exchange(self._h, world, rank, C.cast(self._xchg_cb, C.c_void_p), None))
            self.comm_selftest()
            return
        import sys

        torch = sys.modules.get("torch")
        if world > 1 and torch is not None and torch.cuda.is_available() and torch.cuda.is_initialized() \
                and torch.cuda.current_device() != self.device_index:
            # one process per GPU: RCCL would see duplicate devices if every rank's handle sat on GPU 0
            raise _lib.FcError(_lib.FC_ERR_INVALID, f"rank {rank}: the solver handle lives on GPU {self.device_index} but this process's "
                               f"current GPU is {torch.cuda.current_device()} (create the solver after torch.cuda.set_device(LOCAL_RANK))")
        # creating the communicator may fail for reasons of the machine (RCCL not loadable, ncclCommInitRank refused): that is
        # FcCommInitError, which a caller may answer with the host exchange; rank 0 ships an empty id instead of leaving the
        # others in the broadcast.  A communicator that exists but does not sum (comm_selftest) is a plain FcError: never retried
        if agree is not None:
            # before anything of RCCL is started (ncclGetUniqueId opens rank 0's bootstrap listener): can EVERY rank load RCCL?
            mine = None
            try:
                check(self.lib.fc_comm_probe(rank))
            except _lib.FcError as err:
                mine = err
            if agree(0.0 if mine is None else 1.0) != 0.0:
                raise _lib.FcCommInitError(mine if mine is not None else "another rank cannot load RCCL")
        buf = C.create_string_buffer(128)
        payload, first = None, None
        if rank == 0:
            try:
                check(self.lib.fc_comm_unique_id(buf))
                payload = buf.raw
            except _lib.FcError as err:
                payload, first = b"", err
        uid = broadcast_bytes(payload)
        if not uid:
            raise _lib.FcCommInitError(first if first is not None else "rank 0 could not create the RCCL unique id")
        mine = None
        try:
            check(self.lib.fc_comm_init(self._h, world, rank, C.create_string_buffer(uid, 128)))
        except _lib.FcError as err:
            mine = err
        if agree is not None and agree(0.0 if mine is None else 1.0) != 0.0:
            if mine is None:
                check(self.lib.fc_comm_destroy(self._h))  # mixed outcome: give the communicator back, everybody falls back
            raise _lib.FcCommInitError(mine if mine is not None else "another rank could not create its RCCL communicator")
        if mine is not None:
            raise _lib.FcCommInitError(mine) from mine
        self.comm_selftest()

    def comm_selftest(self) -> float:
        """Pre-flight of the exchange (``fc_comm_selftest``, every rank): one all-reduce of a known vector through the path
        the time steps use, checked on this rank; raises :class:`FcError` naming the first wrong entry."""
        err = C.c_double()
        code = self.lib.fc_comm_selftest(self._h, C.byref(err))
        self._raise_exchange_error()
        check(code)
        return err.value

    def comm_info(self) -> dict:
        """Ranks / rank / transport of the handle's exchange as the library sees it (RCCL: read back from the communicator)."""
        n, r, t = C.c_int32(), C.c_int32(), C.c_int32()
        check(self.lib.fc_comm_info(self._h, C.byref(n), C.byref(r), C.byref(t)))
        return {"nranks": n.value, "rank": r.value, "transport": {0: "none", 1: "rccl", 2: "host"}[t.value]}

    def _raise_exchange_error(self) -> None:
        """An exception inside the host exchange callback (gloo timeout, dead peer) surfaces here, after the C call returned."""
        err, self._xchg_error = self._xchg_error, None
        if err is not None:
            raise _lib.FcError(_lib.FC_ERR_HIP, f"host exchange failed: {err!r}") from err

    # ── lifetime ─────────────────────────────────────────────────────────────
    def close(self) -> None:
        if getattr(self, "_h", None) is not None and self._h:
            self.lib.fc_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ── matrices ─────────────────────────────────────────────────────────────
    def assemble_matrix(self, slot, mass=0.0, nu=0.0, adv=None, lin=None, adv_scale=1.0, lin_scale=1.0, pressure=-1.0, divergence=-1.0):
        adv = None if adv is None else _f64(adv)
        lin = None if lin is None else _f64(lin)
        check(self.lib.fc_assemble_matrix(self._h, slot, mass, nu, ptr(adv), adv_scale, ptr(lin), lin_scale, pressure, divergence))

    def matrix(self, slot) -> sp.csr_matrix:
        vals = np.empty(self.nnz)
        check(self.lib.fc_get_matrix_values(self._h, slot, vals))
        return sp.csr_matrix((vals, self.colidx.copy(), self.rowptr.copy()), shape=(self.N, self.N))

    def set_matrix_values(self, slot, vals) -> None:
        check(self.lib.fc_set_matrix_values(self._h, slot, _f64(vals)))

    def spmv(self, slot, x) -> np.ndarray:
        y = np.empty(self.N)
        check(self.lib.fc_spmv(self._h, slot, _f64(x), y))
        return y

    def bench_spmv(self, slot, reps=1000) -> float:
        ms = C.c_double()
        check(self.lib.fc_bench_spmv(self._h, slot, reps, C.byref(ms)))
        return ms.value

    # ── problem data ─────────────────────────────────────────────────────────
    def set_bc(self, bc_dofs, profiles) -> None:
        bc_dofs = _i32(bc_dofs)
        profiles = _f64(profiles).reshape(len(bc_dofs), -1) if len(bc_dofs) else np.zeros((0, np.shape(profiles)[-1] if np.ndim(profiles) > 1 else 0))
        self.n_act = profiles.shape[1]
        if self.perm is not None and not np.array_equal(np.sort(bc_dofs), np.sort(getattr(self, "bc_dofs", bc_dofs))):
            # the elimination tree parks the Dirichlet dofs in the leaves: a different set needs a new tree
            self.perm = None
            self._structured.clear()
        self.bc_dofs = bc_dofs
        check(self.lib.fc_set_bc(self._h, len(bc_dofs), ptr(bc_dofs), self.n_act, ptr(np.ascontiguousarray(profiles))))

    def set_force(self, profiles) -> None:
        if profiles is None:
            check(self.lib.fc_set_force(self._h, 0, None))
            return
        profiles = _f64(profiles).reshape(self.n_act, 2 * self.nn)
        check(self.lib.fc_set_force(self._h, self.n_act, ptr(profiles)))

    def set_sensors(self, rows: list[tuple[np.ndarray, np.ndarray]]) -> None:
        self._sensor_rows = rows  # (partitioned handles: the library restricts every row to the dofs this rank accounts for)
        self.n_sens = len(rows)
        rp = np.zeros(len(rows) + 1, dtype=np.int32)
        for i, (idx, _) in enumerate(rows):
            rp[i + 1] = rp[i] + len(idx)
        idx = _i32(np.concatenate([r[0] for r in rows])) if rows else np.zeros(0, np.int32)
        w = _f64(np.concatenate([r[1] for r in rows])) if rows else np.zeros(0)
        check(self.lib.fc_set_sensors(self._h, len(rows), ptr(rp), ptr(idx), ptr(w)))

    def set_time_scheme(self, dt: float, nonlinear: bool = True) -> None:
        check(self.lib.fc_set_time_scheme(self._h, float(dt), int(bool(nonlinear))))

    def apply_bc(self, slot) -> None:
        check(self.lib.fc_apply_bc(self._h, slot))

    # ── solver setup (analysis + factorisation inside the library) ───────────
    def setup_solver(self, slot: int, depth: int | None = None, refine: int = 0, check_residual: bool | int = True, merge: int = 2,
                     restructure: bool = False, truncate: int = 0) -> None:
        """Factorise the (BC-eliminated) matrix of ``slot`` on the device (``fc_setup_solver``).

        ``depth`` binary bisections (default: leaves of ≈ 12 cells), fused ``merge`` at a time into a
        2**merge-ary elimination tree (on ``world`` GPUs the root is ``world``-ary first: one sub-tree
        per rank); ``refine`` iterative-refinement sweeps per solve (the fp64 selected inverse is
        accurate to round-off on its own, so 0 + residual monitoring is the default).

        Tree, permutation, factor layout, elimination plan and sweep tables are laid out once per tree by the library's own
        symbolic phase (``csrc/fc_symbolic.hpp``); a later call for the same slot is just the numeric phase on the device
        (on a partitioned handle every rank factorises its own sub-tree and the root).

        ``truncate = d > 0`` (memory-lean preconditioner): only the tree levels ≥ d are factorised and stored (the
        sub-domain solves and their couplings to the separators above — memory shrinks towards O(nnz) as d grows); the
        Schur complement on the top d levels is replaced by a diagonal estimate.  The slot is then a PRECONDITIONER:
        solves and time steps go through GMRES / BiCGStab (``set_solver_options(method=...)``)."""
        if os.environ.get("FC_ND_DEPTH"):  # tuning aids: tree shape (binary bisections, levels fused per tree level)
            depth = int(os.environ["FC_ND_DEPTH"])
        if os.environ.get("FC_ND_MERGE"):
            merge = int(os.environ["FC_ND_MERGE"])
        self._setup_solver_native(slot, depth, refine, check_residual, merge, truncate)

    def _setup_solver_native(self, slot, depth, refine, check_residual, merge, truncate) -> None:
        """``fc_setup_solver``: tree, factor layout, elimination plan, sweep tables, numeric factorisation and its
        acceptance solve all happen inside the library; only sizes come back."""
        code = self.lib.fc_setup_solver(self._h, slot, int(depth or 0), int(merge), int(truncate), int(refine), int(check_residual))
        self._raise_exchange_error()
        check(code)
        info = np.zeros(10, dtype=np.int64)
        check(self.lib.fc_get_solver_info(self._h, slot, info))
        self._n_factor_values, self.total_factor_nnz, self.local_factor_nnz = int(info[0]), int(info[1]), int(info[2])
        self.factor_nnz[slot] = int(info[0])
        self.n_stages, self.depth = int(info[3]), int(info[4])
        self._truncate = int(truncate)
        if self.perm is None:
            self.perm = np.empty(self.N, dtype=np.int32)
            check(self.lib.fc_get_permutation(self._h, self.perm))
            self._tree_args = (int(depth or 0), int(merge))
        if (self.world > 1 or getattr(self, "_force_comm", False)) and self.part is None:
            from types import SimpleNamespace

            kind = np.empty(self.N, dtype=np.uint8)
            check(self.lib.fc_get_rowkind(self._h, kind))
            cells = np.empty(int(info[8]), dtype=np.int32)
            check(self.lib.fc_get_local_cells(self._h, cells))
            self.part = SimpleNamespace(rowkind=kind, local_cells=cells, ar_n=int(info[5]), ar_stage=int(info[6]), ar2_stage=int(info[7]))
        ms = C.c_double()
        check(self.lib.fc_get_refactor_ms(self._h, slot, C.byref(ms)))
        self.refactor_ms[slot] = ms.value
        flag = C.c_int32()
        check(self.lib.fc_get_factors_inexact(self._h, slot, C.byref(flag)))
        self.factors_inexact[slot] = bool(flag.value)
        if flag.value:
            logger.warning("slot %d: the direct factor apply misses 1e-10 on this operator; its solves run GMRES preconditioned by the factors", slot)
        self._structured.add(slot)
        self._solver_opts = (int(refine), int(check_residual), "refine", 1e-10)

    def set_factor_precision(self, bits: int) -> None:
        """Storage width of the factor values of every slot set up from now on: 64 = exact selected inverse (default), 32 / 16
        = compressed factors (fp32 / bfloat16, 50 % / 25 % of the memory; a preconditioner for ``method="gmres" | "bicgstab"``)."""
        if int(bits) != getattr(self, "_factor_bits", 64):
            self._structured.clear()
        check(self.lib.fc_set_factor_precision(self._h, int(bits)))
        self._factor_bits = int(bits)

    def factor_storage(self, slot: int) -> tuple[int, int]:
        """(bits per factor value, bytes of factor values held) of ``slot``."""
        bits, nbytes = C.c_int32(), C.c_int64()
        check(self.lib.fc_get_factor_storage(self._h, slot, C.byref(bits), C.byref(nbytes)))
        return bits.value, nbytes.value

    def set_pressure_pin(self, dof: int | None, shift: float = 1.0) -> None:
        """Enclosed flows (velocity prescribed on the whole boundary): the monolithic matrix is singular, the
        pressure being defined up to a constant.  A positive shift on the diagonal of ONE pressure dof inside
        the factorisation (the matrix itself has no pressure-pressure entries) selects, for every compatible
        right-hand side, exactly the solution with that pressure = 0 — nothing else changes."""
        dof = None if dof is None else int(dof)
        if dof is not None and not 2 * self.nn <= dof < self.N:
            raise ValueError("the pinned dof must be a pressure dof")
        if dof != self._pin:
            self._pin, self._pin_shift = dof, float(shift)
            self._probe = None
            check(self.lib.fc_set_pressure_pin(self._h, -1 if dof is None else dof, float(shift)))

    def refactor(self, slot: int) -> float:
        """Numeric factorisation of the slot's current matrix on the device (the structure of the first
        ``setup_solver`` is reused): what ``solver.set_operator(A)`` costs.  Returns device milliseconds."""
        if slot not in self._structured:
            raise RuntimeError("setup_solver(slot) must run once before refactor(slot)")
        if getattr(self, "_truncate", 0):
            # truncated factors (a preconditioner): the library redoes the numeric phase AND the diagonal stand-in of the dropped
            # levels' Schur complement in one call; the caller's Krylov options stay
            opts = self._solver_opts
            depth, merge = self._tree_args
            self._setup_solver_native(slot, depth, 0, opts[1], merge, self._truncate)
            self.set_solver_options(*opts)
            return self.refactor_ms[slot]
        ms = C.c_double()
        code = self.lib.fc_refactor(self._h, slot, C.byref(ms))
        self._raise_exchange_error()
        check(code)
        self.refactor_ms[slot] = ms.value
        # end-to-end acceptance of the new factors: one solve with a fixed right-hand side, residual
        # against the matrix itself (block-local pivoting in fc_fe_pivot is not trusted blindly)
        if self._probe is None:
            self._probe = np.cos(0.37 * np.arange(self.N) + 0.1)
            if self._pin is not None:
                self._probe[2 * self.nn :] = 0.0  # compatible with the constant-pressure null space
        # (on a partitioned handle the probe is a collective: every rank refactorises, every rank probes)
        res, inexact = C.c_double(), C.c_int32()
        code = self.lib.fc_accept_factors(self._h, slot, C.byref(res), C.byref(inexact))
        self._raise_exchange_error()
        check(code)
        self.factors_inexact[slot] = bool(inexact.value)
        if inexact.value:
            logger.warning("slot %d: the direct factor apply reaches only %.1e on this operator; its solves run GMRES preconditioned by the factors", slot, res.value)
        return ms.value

    def refactor_flops(self) -> tuple[float, float]:
        """Trailing-update flops of the last numeric factorisation: (as run, with every row of a multi-GPU root front swept)."""
        a, b = C.c_double(), C.c_double()
        check(self.lib.fc_get_refactor_flops(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def factor_values(self, slot: int) -> np.ndarray:
        """Factor values of ``slot`` as they sit on the device (the layout ``fcsym::layout_factors`` / ``tests/support/ndsolver.BlockFactors`` describe)."""
        n = int(self._n_factor_values)
        out = np.empty(n)
        check(self.lib.fc_get_factor_values(self._h, slot, n, out))
        return out

    def set_solver_options(self, refine: int = 0, check_residual: bool | int = True, method: str = "refine", rtol: float = 1e-10) -> None:
        """``check_residual``: 0 / False = no residual monitor, n >= 1 = the time steps form ``|b - A x| / |b|`` on every n-th step
        (``info[1]``; NaN on the steps in between).
        ``method="refine"``: factor sweeps (+ ``refine`` iterative-refinement sweeps) — what the time steps use.
        ``method="bicgstab"`` / ``"gmres"``: right-preconditioned BiCGStab / restarted GMRES(30), device-resident
        (``refine`` = iteration cap, ``rtol`` = relative residual target), with the slot's current factors as
        preconditioner; for :meth:`solve` only."""
        self._solver_opts = (int(refine), int(check_residual), method, float(rtol))
        m = {"refine": _lib.METHOD_REFINE, "bicgstab": _lib.METHOD_BICGSTAB, "gmres": _lib.METHOD_GMRES}[method]
        check(self.lib.fc_set_solver_options(self._h, m, int(refine), float(rtol), int(check_residual)))

    def setup_krylov(self, slot: int, sweeps: int = 2, method: str = "gmres", max_iter: int = 200, rtol: float = 1e-10,
                     check_residual: bool | int = True) -> dict:
        """Factorisation-free solver setup of ``slot`` (``fc_setup_krylov``): nothing is factorised; solves and time steps run
        the device GMRES / BiCGStab right-preconditioned by the SIMPLE / AMG block preconditioner (``sweeps`` damped-Jacobi
        sweeps on the velocity block, one smoothed-aggregation V-cycle on the pressure Schur complement ``B diag(F)^-1 Bt``).
        Memory O(nnz).  Returns the sizes of what was built (``krylov_info``)."""
        m = {"bicgstab": _lib.METHOD_BICGSTAB, "gmres": _lib.METHOD_GMRES}[method]
        check(self.lib.fc_setup_krylov(self._h, slot, int(sweeps), m, int(max_iter), float(rtol), int(check_residual)))
        self._structured.discard(slot)
        self._krylov_slots = getattr(self, "_krylov_slots", set()) | {slot}
        self._solver_opts = (int(max_iter), int(check_residual), method, float(rtol))
        if self.perm is None:
            self.perm = np.empty(self.N, dtype=np.int32)
            check(self.lib.fc_get_permutation(self._h, self.perm))
        self.factor_nnz[slot] = 0
        return self.krylov_info(slot)

    def tree_info(self, min_tree: bool = False) -> dict:
        """Bisections fused per level of the elimination tree (root first) and, on request, the factor values of the
        all-binary-pairs tree of the same depth (``fc_get_tree_info``)."""
        bits, n = np.zeros(16, dtype=np.int32), C.c_int32()
        nnz = C.c_int64(-1)
        check(self.lib.fc_get_tree_info(self._h, bits, C.byref(n), C.cast(C.byref(nnz), C.c_void_p) if min_tree else None))
        return {"bits": [int(b) for b in bits[: n.value]], "nnz_min_tree": int(nnz.value) if min_tree else None}

    def krylov_info(self, slot: int) -> dict:
        """What ``setup_krylov`` holds on the device for ``slot``."""
        info, om = np.zeros(8, dtype=np.int64), C.c_double()
        check(self.lib.fc_get_krylov_info(self._h, slot, info, C.byref(om)))
        return {"bytes": int(info[0]), "velocity_dofs": int(info[1]), "pressure_dofs": int(info[2]), "amg_levels": int(info[3]),
                "coarsest_rows": int(info[4]), "launches_per_apply": int(info[5]), "jacobi_sweeps": int(info[6]),
                "setup_ms": int(info[7]), "jacobi_omega": om.value}

    def update_operator(self, slot: int) -> None:
        """The slot's matrix changed (assemble + apply_bc) but its factors are kept: refresh the permuted copy
        only.  ``solve`` with ``method="bicgstab"`` then uses the old factors as preconditioner."""
        if slot not in self._structured and slot not in getattr(self, "_krylov_slots", ()):
            raise RuntimeError("setup_solver(slot) must run once before update_operator(slot)")
        check(self.lib.fc_update_operator(self._h, slot))

    def _upload_energy_matrix(self) -> None:
        """(u, v) mass matrix in the permuted numbering for the fused energy evaluation."""
        self.assemble_matrix(SLOT_MASS, mass=1.0, nu=0.0, pressure=0.0, divergence=0.0)
        M = self.matrix(SLOT_MASS)
        nn2 = 2 * self.nn
        M = sp.block_diag([M[:nn2, :nn2], sp.csr_matrix((self.N - nn2, self.N - nn2))]).tocsr()
        M.eliminate_zeros()
        p = self.perm
        Mp = M[p][:, p].tocsr()
        Mp.sort_indices()
        check(self.lib.fc_set_energy_matrix(self._h, _i32(Mp.indptr), _i32(Mp.indices), _f64(Mp.data)))

    # ── state ────────────────────────────────────────────────────────────────
    def set_state(self, u_n, u_nn, p_n=None) -> None:
        p = None if p_n is None else _f64(p_n)
        check(self.lib.fc_set_state(self._h, _f64(u_n), _f64(u_nn), ptr(p)))

    def partition_info(self) -> dict:
        """Cells this rank computes the right-hand side for / assembles element matrices of, lead flag, ranks."""
        out = np.zeros(4, dtype=np.int32)
        check(self.lib.fc_get_partition_info(self._h, out))
        return {"rhs_cells": int(out[0]), "matrix_cells": int(out[1]), "lead": bool(out[2]), "ranks": int(out[3])}

    def undo_step(self) -> None:
        """``fc_undo_step``: (u_n, u_nn, p_n) as they were before the last single step (after FC_ERR_DIVERGED: the reference's
        state is untouched by a failed step, flowsolver.py:727-751)."""
        check(self.lib.fc_undo_step(self._h))

    def get_state(self):
        u_n, u_nn, p_n = np.empty(2 * self.nn), np.empty(2 * self.nn), np.empty(self.th.nv)
        check(self.lib.fc_get_state(self._h, ptr(u_n), ptr(u_nn), ptr(p_n)))
        return u_n, u_nn, p_n

    def owned_mask(self) -> np.ndarray:
        """W-layout mask of the dofs whose values this rank is responsible for when fields are merged."""
        if self.part is None:
            return np.ones(self.N, dtype=bool)
        k = self.part.rowkind
        return (k == 1) | ((k == 2) & (self.rank == 0))

    def get_solution(self) -> np.ndarray:
        up = np.empty(self.N)
        check(self.lib.fc_get_solution(self._h, up))
        return up

    # ── hot path ─────────────────────────────────────────────────────────────
    def set_rhs_operator(self, slot: int, Cmat: sp.csr_matrix | None) -> None:
        """b -= C u_n with C given in W numbering (N × 2nn); rows are permuted here."""
        if Cmat is None:
            check(self.lib.fc_set_rhs_operator(self._h, slot, None, None, None))
            return
        Cp = Cmat.tocsr()[self.perm].tocsr()
        Cp.eliminate_zeros()
        Cp.sort_indices()
        check(self.lib.fc_set_rhs_operator(self._h, slot, ptr(_i32(Cp.indptr)), ptr(_i32(Cp.indices)), ptr(_f64(Cp.data))))

    def _step_buffers(self):
        # persistent argument buffers and their ctypes pointers: the call is on the critical path of
        # every synchronous step (a fresh ndarray + ctypes cast per argument costs ~1 us each)
        b = self._step_bufs
        if b is None or b[0].size != max(self.n_act, 1) or b[2].size != max(self.n_sens, 1):
            arrs = (np.zeros(max(self.n_act, 1)), np.zeros(max(self.n_act, 1)), np.empty(max(self.n_sens, 1)), np.empty(4))
            dE = C.c_double()
            b = self._step_bufs = arrs + (dE, tuple(ptr(a) for a in arrs), C.byref(dE))
        return b

    def step_begin(self, order_slot: int, u_ctrl, compute_energy: bool = True, u_force=None) -> None:
        """First half of :meth:`step`: hand the controls over and enqueue the step; the GPU works from here on."""
        u, uf, y, info, dE, (pu, puf, py, pinfo), pdE = self._step_buffers()
        if self.n_act:
            u[:] = u_ctrl
            if u_force is not None:
                uf[:] = u_force
        code = self.lib.fc_step_begin(self._h, order_slot, pu if self.n_act else None, puf if (self.n_act and u_force is not None) else None,
                                      1 if compute_energy else 0)
        if code:
            check(code)

    def step_end(self, early: bool = False):
        """Second half: wait for the step's record; returns (y, dE, info) like :meth:`step`.

        ``early=True``: return as soon as the measurements (and the non-finite flag) are there — ``(y, None, None)`` — and leave
        energy / solve info to :meth:`step_collect`; on a single-GPU handle they are computed on a second stream while the host and
        the next step go on (``fc_step_collect`` in fc_hip.h)."""
        u, uf, y, info, dE, (pu, puf, py, pinfo), pdE = self._step_bufs
        code = self.lib.fc_step_end(self._h, py, None if early else pdE, None if early else pinfo)
        if self._xchg_error is not None:
            self._raise_exchange_error()
        if code:
            check(code)
        if early:
            return y[: self.n_sens].copy(), None, None
        return y[: self.n_sens].copy(), dE.value, info

    def step_collect(self):
        """(dE, info) of the last step collected with ``step_end(early=True)``; blocks until they exist."""
        u, uf, y, info, dE, (pu, puf, py, pinfo), pdE = self._step_bufs
        check(self.lib.fc_step_collect(self._h, pdE, pinfo))
        return dE.value, info

    def step(self, order_slot: int, u_ctrl, compute_energy: bool = True, u_force=None):
        """One blocking ``fc_step``: (y, dE, info) of this step, all waited for."""
        u, uf, y, info, dE, (pu, puf, py, pinfo), pdE = self._step_buffers()
        if self.n_act:
            u[:] = u_ctrl
            if u_force is not None:
                uf[:] = u_force
        code = self.lib.fc_step(self._h, order_slot, pu if self.n_act else None, puf if (self.n_act and u_force is not None) else None, py, pdE,
                                1 if compute_energy else 0, pinfo)
        if self._xchg_error is not None:
            self._raise_exchange_error()
        if code:
            check(code)
        return y[: self.n_sens].copy(), dE.value, info

    def run(self, first_order_slot: int, n_steps: int, u_ctrl, compute_energy: bool = True):
        u = _f64(u_ctrl)
        is_seq = int(u.ndim == 2)
        y = np.empty((n_steps, max(self.n_sens, 1)))
        yv = np.empty((n_steps, self.n_sens))
        dE = np.empty(n_steps)
        check(self.lib.fc_run(self._h, first_order_slot, n_steps, ptr(u), is_seq, ptr(yv), ptr(dE), int(compute_energy)))
        del y
        return yv, dE

    # ── shared-operator batched stepping (k lock-step simulations on this handle) ────────────
    def set_batch(self, k: int) -> None:
        """Allocate (k > 0) or free (k = 0) the state of k lock-step simulations that share this handle's operators and
        factors (``fc_set_batch``): IC sweeps / controller sweeps of the reference run k FlowSolver instances instead."""
        check(self.lib.fc_set_batch(self._h, int(k)))
        self.batch_k = int(k)
        self._batch_bufs = None

    def set_state_batch(self, u_n, u_nn, p_n=None) -> None:
        k = self.batch_k
        u_n, u_nn = _f64(u_n).reshape(k, 2 * self.nn), _f64(u_nn).reshape(k, 2 * self.nn)
        p = None if p_n is None else _f64(p_n).reshape(k, self.th.nv)
        check(self.lib.fc_set_state_batch(self._h, k, u_n, u_nn, ptr(p)))

    def get_state_batch(self):
        k = self.batch_k
        u_n, u_nn, p_n = np.empty((k, 2 * self.nn)), np.empty((k, 2 * self.nn)), np.empty((k, self.th.nv))
        check(self.lib.fc_get_state_batch(self._h, k, ptr(u_n), ptr(u_nn), ptr(p_n)))
        return u_n, u_nn, p_n

    def get_solution_batch(self) -> np.ndarray:
        up = np.empty((self.batch_k, self.N))
        check(self.lib.fc_get_solution_batch(self._h, self.batch_k, up))
        return up

    def _batch_buffers(self):
        b = self._batch_bufs
        if b is None:
            k = self.batch_k
            arrs = (np.zeros((k, max(self.n_act, 1))), np.zeros((k, max(self.n_act, 1))), np.empty((k, max(self.n_sens, 1))), np.empty(k),
                    np.empty((k, 4)))
            b = self._batch_bufs = arrs + (tuple(ptr(a) for a in arrs),)
        return b

    def step_batch_begin(self, order_slot: int, u_ctrl, compute_energy: bool = True, u_force=None) -> None:
        u, uf, y, dE, info, (pu, puf, py, pdE, pinfo) = self._batch_buffers()
        if self.n_act:
            u[:, : self.n_act] = np.asarray(u_ctrl, dtype=np.float64).reshape(self.batch_k, self.n_act)
            if u_force is not None:
                uf[:, : self.n_act] = np.asarray(u_force, dtype=np.float64).reshape(self.batch_k, self.n_act)
        code = self.lib.fc_step_batch_begin(self._h, order_slot, self.batch_k, pu if self.n_act else None,
                                            puf if (self.n_act and u_force is not None) else None, 1 if compute_energy else 0)
        if code:
            check(code)

    def step_batch_end(self):
        """(y [k, n_sens], dE [k], info [k, 4]) of the step in flight; raises :class:`FcDiverged` if any simulation
        produced a non-finite velocity (``info[:, 3]`` of ``self._batch_bufs`` marks which)."""
        u, uf, y, dE, info, (pu, puf, py, pdE, pinfo) = self._batch_bufs
        code = self.lib.fc_step_batch_end(self._h, self.batch_k, py, pdE, pinfo)
        if code:
            check(code)
        return y[:, : self.n_sens].copy(), dE.copy(), info

    def step_batch_end_early(self):
        """(y [k, n_sens], flags [k]) as soon as the batched solve is done; energy and solve info follow (:meth:`step_batch_collect`).
        Raises :class:`FcDiverged` if any simulation produced a non-finite velocity (``self._batch_flags`` marks which)."""
        u, uf, y, dE, info, (pu, puf, py, pdE, pinfo) = self._batch_bufs
        if getattr(self, "_batch_flags", None) is None or self._batch_flags.size != self.batch_k:
            self._batch_flags = np.zeros(self.batch_k, dtype=np.int32)
        code = self.lib.fc_step_batch_end_early(self._h, self.batch_k, py, ptr(self._batch_flags))
        if code:
            check(code)
        return y[:, : self.n_sens].copy(), self._batch_flags

    def step_batch_collect(self):
        """(dE [k], info [k, 4]) of the last batched step that ended early; blocks until they exist."""
        u, uf, y, dE, info, (pu, puf, py, pdE, pinfo) = self._batch_bufs
        check(self.lib.fc_step_batch_collect(self._h, self.batch_k, pdE, pinfo))
        return dE.copy(), info

    def step_batch(self, order_slot: int, u_ctrl, compute_energy: bool = True, u_force=None):
        """One blocking batched step: (y, dE, info), all waited for."""
        self.step_batch_begin(order_slot, u_ctrl, compute_energy, u_force)
        return self.step_batch_end()

    def reset_sim_batch(self, s: int) -> None:
        """``fc_reset_sim_batch``: take simulation ``s`` (e.g. one that diverged) out of the batch's dynamics — zero state, flag cleared."""
        check(self.lib.fc_reset_sim_batch(self._h, int(s)))

    def batch_info(self) -> dict:
        a = np.zeros(8)
        check(self.lib.fc_get_batch_info(self._h, a))
        return {"k": int(a[0]), "KB": int(a[1]), "scratch_rows": int(a[2]), "block_launches": int(a[3]), "fold_launches": int(a[4]),
                "factor_bytes": a[5], "vector_bytes": a[6], "tasks": int(a[7])}

    def bench_batch_apply(self, slot: int, reps: int = 200) -> float:
        ms = C.c_double()
        check(self.lib.fc_bench_batch_apply(self._h, slot, reps, C.byref(ms)))
        return ms.value

    def solve_batch(self, slot: int, b) -> np.ndarray:
        b = _f64(b).reshape(self.batch_k, self.N)
        x = np.empty_like(b)
        check(self.lib.fc_solve_batch(self._h, slot, self.batch_k, b, x))
        return x

    # ── parity hooks ─────────────────────────────────────────────────────────
    def assemble_rhs(self, order_slot: int, u_ctrl) -> np.ndarray:
        u = _f64(np.atleast_1d(u_ctrl)) if self.n_act else None
        b = np.empty(self.N)
        check(self.lib.fc_assemble_rhs(self._h, order_slot, ptr(u), b))
        return b

    def solve(self, slot: int, b):
        x = np.empty(self.N)
        info = np.empty(4)
        code = self.lib.fc_solve(self._h, slot, _f64(b), x, ptr(info))
        self._raise_exchange_error()
        check(code)
        return x, info

    def energy(self, u) -> float:
        E = C.c_double()
        check(self.lib.fc_energy(self._h, _f64(u), C.byref(E)))
        return E.value

    def measure(self, up) -> np.ndarray:
        y = np.empty(max(self.n_sens, 1))
        check(self.lib.fc_measure(self._h, _f64(up), ptr(y)))
        return y[: self.n_sens]

    # ── measurement ──────────────────────────────────────────────────────────
    def bench_sweeps(self, slot: int, reps: int = 200):
        ms, nl = C.c_double(), C.c_int32()
        check(self.lib.fc_bench_sweeps(self._h, slot, reps, C.byref(ms), C.byref(nl)))
        return ms.value, nl.value

    PHASES = ("rhs", "up_sweeps", "exchange1", "root", "exchange2", "down_sweeps", "tail", "exchange3", "publish")

    def set_phase_timing(self, on: bool) -> None:
        """HIP-event marks at the phase boundaries of every ``fc_step`` from now on (an instrumented replay, see fc_hip_internal.h)."""
        check(self.lib.fc_set_phase_timing(self._h, int(bool(on))))

    def get_phase_timing(self) -> dict:
        """Mean microseconds per step of every phase since :meth:`set_phase_timing`, keyed by :attr:`PHASES`, plus ``steps``."""
        us, n = np.zeros(len(self.PHASES)), C.c_int64()
        check(self.lib.fc_get_phase_timing(self._h, us, C.byref(n)))
        d = {k: float(v) / max(n.value, 1) for k, v in zip(self.PHASES, us)}
        d["steps"] = int(n.value)
        return d

    def set_timing(self, on: bool) -> None:
        check(self.lib.fc_set_timing(self._h, int(bool(on))))

    def get_timing(self) -> dict:
        a, b = C.c_double(), C.c_double()
        na, nb = C.c_int64(), C.c_int64()
        check(self.lib.fc_get_timing(self._h, C.byref(a), C.byref(na), C.byref(b), C.byref(nb)))
        return {"sweep_ms": a.value, "sweep_launches": na.value, "spmv_ms": b.value, "spmv_launches": nb.value}

    def algorithmic_bytes(self, slot: int):
        a, b = C.c_double(), C.c_double()
        check(self.lib.fc_algorithmic_bytes(self._h, slot, C.byref(a), C.byref(b)))
        return a.value, b.value


__all__ = ["DeviceSolver", "SLOT_BDF1", "SLOT_BDF2", "SLOT_MASS", "SLOT_SCRATCH"]
