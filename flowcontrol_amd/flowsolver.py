"""``FlowSolver`` — the reference's public API (``src/flowcontrol/flowsolver.py``) on an MI355X.

Same constructor, method names, attributes and error behaviour as the reference class so that
case files and closed-loop scripts port unchanged; what differs is where the work happens:

* ``_prepare_systems`` (reference ``:665-701``): both LHS matrices are assembled by the HIP
  element loop, Dirichlet rows/columns are eliminated on the device, and the time-invariant
  operator is factorised once (``fc_setup_solver``) into level-wise sparse factors that live in HBM.
* ``step`` (reference ``:703-799``) crosses the C ABI **once** (``fc_step``): RHS element loop with
  BC lifting → factor sweeps + refinement → split/shift → sensors → energy, all on one HIP stream;
  only ``u_ctrl`` goes in and ``(y_meas, dE, info)`` comes out.
* fields (``fs.fields.u_`` …) are downloaded lazily, only when user code reads them.

There is no CPU fallback: without ``libfc_hip.so`` and a visible GPU the first device operation
raises :class:`flowcontrol_amd._lib.FcError`.
"""

from __future__ import annotations

import json
import logging
import time
from abc import ABC, abstractmethod
from pathlib import Path
from typing import Any, Iterable, Sequence

import numpy as np
import pandas as pd
from numpy.typing import NDArray

from . import flowsolverparameters
from ._lib import SLOT_BDF1, SLOT_BDF2, SLOT_MASS, FcDiverged
from .actuator import ACTUATOR_TYPE
from .exporter import FlowExporter, read_frame, write_frame
from .fem.boundary import Constant, DirichletBC, combine_bcs, pressure_pin
from .fem.mesh import Mesh, read_xdmf_mesh
from .fem.spaces import Function, TaylorHood
from .flowfield import BoundaryConditions, FlowField, FlowFieldCollection, SimPaths
from .nsforms import NSForms
from .steadystate import SteadyStateSolver

logger = logging.getLogger(__name__)


class EnergyField:
    """u'·u' on the P4 Lagrange nodes of the mesh (:meth:`FlowSolver.compute_energy_field`): a minimal stand-in for the dolfin P4 Function."""

    def __init__(self, coords: np.ndarray, values: np.ndarray):
        self.coords, self.values = coords, values

    def vector(self):
        return self

    def get_local(self) -> np.ndarray:
        return self.values.copy()

    def __call__(self, x):  # nearest node (enough for a look-up at a node; the field is piecewise P4 in between)
        return float(self.values[np.argmin(np.sum((self.coords - np.asarray(x, dtype=float)) ** 2, axis=1))])


class FlowSolver(ABC):
    """Abstract base class for flow simulation and control.

    Subclasses implement ``_make_boundaries() -> DataFrame`` (index = boundary name, column
    ``subdomain``), ``_make_bcs() -> BoundaryConditions`` (``bcu[0]`` must be the inlet) and the
    ``make_default`` factory.
    """

    def __init__(
        self,
        params_flow: flowsolverparameters.ParamFlow,
        params_time: flowsolverparameters.ParamTime,
        params_save: flowsolverparameters.ParamSave,
        params_solver: flowsolverparameters.ParamSolver,
        params_mesh: flowsolverparameters.ParamMesh,
        params_control: flowsolverparameters.ParamControl,
        params_ic: flowsolverparameters.ParamIC,
        params_restart: flowsolverparameters.ParamRestart | None = None,
        verbose: int = 1,
    ) -> None:
        self._validate_params(params_flow, params_time, params_save, params_solver, params_mesh, params_control, params_ic, params_restart)
        self.params_flow = params_flow
        self.params_time = params_time
        self.params_save = params_save
        self.params_solver = params_solver
        self.params_mesh = params_mesh
        self.params_restart = params_restart
        self.params_control = params_control
        self.params_ic = params_ic
        self.verbose = verbose
        #: device solver knobs (not in the reference): ND tree depth (None = automatic) and the
        #: number of iterative-refinement sweeps per solve (0 = factor apply + residual monitoring)
        self.nd_depth: int | None = None
        self.refine_steps: int = 0
        # every step's solve is monitored: relative residual of the linear system above this → error (the
        # factor apply is exact to round-off, ~1e-16; a larger value means broken factors or a singular system)
        self.residual_tol: float = 1e-8
        #: ... on every n-th step (1: every step; 0: never).  The reference forms no such residual (flowsolver.py:728-737 tests finiteness
        #: only): the monitor is this solver's own check and costs one pass over the system matrix per monitored step.  On the steps
        #: in between ``solve_info[1]`` is NaN; the non-finite test runs on every step regardless.
        #: None = auto: every step while the factors stay in the Infinity Cache (the pass hides beside the next step), every 8th step on
        #: meshes whose factors stream from HBM (refined cylinder, pinball, cavity meshes: there it costs ~10 % of a step)
        self.check_residual_every: int | None = None
        #: memory-lean mode: factorise only the tree levels >= nd_truncate and solve every step with a Krylov method
        #: preconditioned by those truncated factors (0 = full selected inverse, applied directly)
        self.nd_truncate: int = 0
        #: compressed factors: 64 = exact selected inverse applied directly (default); 32 / 16 = the factor values are stored in
        #: fp32 / bfloat16 (50 % / 25 % of the memory) and every step is solved by the device GMRES / BiCGStab preconditioned
        #: with them (a few iterations)
        self.factor_bits: int = 64
        self.krylov_method: str = "gmres"
        self.krylov_max_iter: int = 500
        self.krylov_rtol: float = 1e-12
        #: factorisation-free mode: ``"schur_amg"`` = nothing is factorised; every step is solved by the device GMRES / BiCGStab
        #: right-preconditioned by the SIMPLE / AMG block preconditioner (``krylov_sweeps`` damped-Jacobi sweeps on the velocity
        #: block + one algebraic-multigrid V-cycle on the pressure Schur complement); memory O(nnz).  None = factors (default)
        self.krylov_precond: str | None = None
        self.krylov_sweeps: int = 2
        #: multi-GPU: who the ranks are (flowcontrol_amd.comm.Comm); None = the torch.distributed process group of this process, if any
        self.comm = None
        #: set when the in-library RCCL communicator could not be created and the exchanges were staged through the host instead
        self.exchange_fallback: str | None = None
        #: largest relative residual the monitor has seen on this solver's steps (collected with every step's log row, i.e. without
        #: making the host wait for the residual of the step that just ended, as reading ``solve_info`` right after ``step`` does)
        self.residual_max = 0.0
        self._setup()

    _pending_log = None        # (iter, t, u_ctrl, y, dE | None, runtime) of the last step, not yet in the exporter
    _late_pending = False      # the last device step's (dE, info) have not been fetched yet
    _solve_info = None
    _last_dE = float("nan")
    _residual_breach = None

    # ── the exporter, with the last step's log row possibly still pending (step() books a step's row while the NEXT step runs
    #    on the GPU: the host work of a step is hidden behind the device's) ──────────────────────────────────────────────
    @property
    def exporter(self) -> "FlowExporter":
        self._flush_log()
        return self._exporter

    def _report_breach(self) -> bool:
        """The residual monitor's verdict on the last step(s), if one is pending: logged, and raised when ``throw_error`` is set.
        ``step()`` consumes it itself; the readers of a finished run (``timeseries``, ``write_timeseries``, a checkpoint) call this so
        that a breach on the LAST step of a run is not lost (returns True when there was one)."""
        breach = getattr(self, "_residual_breach", None)
        if breach is None:
            return False
        self._residual_breach = None
        msg = f"linear solve residual {breach[0]:.2e} exceeds residual_tol = {self.residual_tol:.1e} at iteration {breach[1]}"
        logger.critical(msg)
        if self.params_solver.throw_error:
            raise RuntimeError(msg)
        return True

    def _before_release(self) -> None:
        """The device handle is about to go (``th.release_device()``): book the deferred log row while its energy / residual can
        still be fetched."""
        try:
            self._flush_log()
            self._collect()
        except Exception as err:  # the handle may be broken already: the row keeps NaN
            logger.warning("could not collect the last step's energy / residual before the device was released: %s", err)
            self._late_pending = False

    @exporter.setter
    def exporter(self, value) -> None:
        self._pending_log = None
        self._exporter = value

    def _collect(self) -> None:
        """Energy and solve info of the last step, which the device computes on a second stream while the host and the next step go on
        (``fc_step_collect``): fetched once, when somebody needs them — the log row, ``solve_info``, the residual check."""
        if getattr(self, "_late_pending", False):
            self._late_pending = False
            if getattr(self.th, "_device", None) is None:  # the handle was released without the hook (th._device = None by hand)
                return
            dE, info = self.th.device().step_collect()
            self._last_dE = dE
            self._solve_info = info
            if info[1] > self.residual_max:  # (NaN compares False)
                self.residual_max = float(info[1])
            if info[1] > self.residual_tol:  # NaN (monitor off / not this step) compares False
                self._residual_breach = (float(info[1]), self.iter)

    @property
    def solve_info(self):
        """(refinement sweeps / Krylov iterations, relative residual, |b|, flag) of the last step (the device's buffer, reused by the next)."""
        self._collect()
        return self._solve_info

    @solve_info.setter
    def solve_info(self, value) -> None:
        self._late_pending = False
        self._solve_info = value

    def _flush_log(self) -> None:
        row = getattr(self, "_pending_log", None)
        if row is not None:
            self._pending_log = None
            it, t, u_ctrl, y, dE, runtime = row
            self._collect()  # energy / residual of that step (an overlapped step computed them behind the host's back): here by now
            if dE is None:
                dE = self._last_dE
            if self._niter_multiple_of(it, self.verbose):
                self._exporter.log_progress(it, self.params_time.num_steps, t, self.params_time.Tfinal + self.params_time.Tstart, runtime)
            self._exporter.log(u_ctrl=u_ctrl, y_meas=y, dE=dE, t=t, runtime=runtime)

    # ── validation (reference :108-165) ──────────────────────────────────────
    @staticmethod
    def _validate_params(params_flow, params_time, params_save, params_solver, params_mesh, params_control, params_ic, params_restart=None) -> None:
        if params_time.dt <= 0:
            raise ValueError(f"dt must be positive, got {params_time.dt}")
        if params_time.num_steps < 0:
            raise ValueError(f"num_steps must be non-negative, got {params_time.num_steps}")
        if params_flow.Re <= 0:
            raise ValueError(f"Re must be positive, got {params_flow.Re}")
        if params_save.save_every < 0:
            raise ValueError(f"save_every must be non-negative, got {params_save.save_every}")
        if params_save.energy_every < 0:
            raise ValueError(f"energy_every must be non-negative, got {params_save.energy_every}")
        if params_control.actuator_number < 0:
            raise ValueError(f"actuator_number must be non-negative, got {params_control.actuator_number}")
        if params_control.sensor_number < 0:
            raise ValueError(f"sensor_number must be non-negative, got {params_control.sensor_number}")
        if len(params_control.actuator_list) != params_control.actuator_number:
            raise ValueError(
                f"actuator_list length ({len(params_control.actuator_list)}) does not match actuator_number ({params_control.actuator_number})"
            )
        if len(params_control.sensor_list) != params_control.sensor_number:
            raise ValueError(
                f"sensor_list length ({len(params_control.sensor_list)}) does not match sensor_number ({params_control.sensor_number})"
            )
        if not Path(params_mesh.meshpath).exists():
            raise FileNotFoundError(f"Mesh file not found at {params_mesh.meshpath}")
        if params_restart is not None and params_restart.Trestartfrom < 0:
            raise ValueError(f"Trestartfrom must be non-negative, got {params_restart.Trestartfrom}")

    # ── setup (reference :169-263) ───────────────────────────────────────────
    def _setup(self) -> None:
        self.fields = FlowFieldCollection()
        self.E0: float = 0.0
        self.paths = self._define_paths()
        self.mesh = self._make_mesh()
        self.V, self.P, self.W = self._make_function_spaces()
        self.boundaries = self._make_boundaries()
        self._mark_boundaries()
        self._load_actuators()
        self._load_sensors()
        self.bc = self._make_bcs()
        self.forms = NSForms(
            W=self.W,
            Re=self.params_flow.Re,
            dt=self.params_time.dt,
            is_nonlinear=self.params_solver.is_eq_nonlinear,
            shift=self.params_solver.shift,
        )
        self.exporter = FlowExporter(
            paths=self.paths,
            fields=self.fields,
            V=self.V,
            P=self.P,
            Tstart=self.params_time.Tstart,
            dt=self.params_time.dt,
            save_every=self.params_save.save_every,
        )
        self.first_step = True
        self._systems_ready = False
        hooks = getattr(self.th, "_release_hooks", None)
        if hooks is None:
            hooks = self.th._release_hooks = []
        hooks.append(self._before_release)

    def _define_paths(self) -> SimPaths:
        def ext(T: float) -> str:
            return f"_restart{T:.3f}".replace(".", ",")

        Tstart = self.params_time.Tstart
        Trestartfrom = self.params_restart.Trestartfrom if self.params_restart else 0.0
        path_out = Path(self.params_save.path_out)
        return SimPaths(
            U0=path_out / "steady" / "U0.xdmf",
            P0=path_out / "steady" / "P0.xdmf",
            steady_meta=path_out / "steady" / "meta.json",
            U=path_out / ("U" + ext(Trestartfrom) + ".xdmf"),
            P=path_out / ("P" + ext(Trestartfrom) + ".xdmf"),
            Uprev=path_out / ("Uprev" + ext(Trestartfrom) + ".xdmf"),
            U_restart=path_out / ("U" + ext(Tstart) + ".xdmf"),
            Uprev_restart=path_out / ("Uprev" + ext(Tstart) + ".xdmf"),
            P_restart=path_out / ("P" + ext(Tstart) + ".xdmf"),
            timeseries=path_out / ("timeseries1D" + ext(Tstart) + ".csv"),
            metadata=path_out / ("meta" + ext(Tstart) + ".json"),
            mesh=Path(self.params_mesh.meshpath),
        )

    def _make_mesh(self) -> Mesh:
        logger.info(f"Mesh @ {self.params_mesh.meshpath}")
        mesh = read_xdmf_mesh(self.params_mesh.meshpath)
        logger.info(f"Mesh has {mesh.num_cells} cells")
        return mesh

    def _make_function_spaces(self):
        self.th = TaylorHood(self.mesh)
        logger.debug(f"DOFs: {self.th.N} ({2 * self.th.nn} velocity + {self.th.nv} pressure)")
        return self.th.V, self.th.P, self.th.W

    def _mark_boundaries(self) -> None:
        """Facet markers per boundary (later boundaries overwrite earlier ones, as repeated
        ``SubDomain.mark`` calls do in the reference, ``flowsolver.py:252-263``)."""
        self.bnd_markers = np.full(self.mesh.num_edges, np.iinfo(np.int64).max, dtype=np.int64)
        indices = []
        for i, row in enumerate(self.boundaries.itertuples()):
            row.subdomain.mark(self.bnd_markers, i, self.mesh)
            indices.append(i)
        self.boundaries["idx"] = indices

    # ── actuators / sensors (reference :267-325) ─────────────────────────────
    def _load_actuators(self) -> None:
        for actuator in self.params_control.actuator_list:
            actuator.load_expression(self)

    def _load_sensors(self) -> None:
        for sensor in self.params_control.sensor_list:
            if sensor.require_loading:
                sensor.load(self)

    def set_actuators_u_ctrl(self, u_ctrl: Iterable) -> None:
        u_ctrl = list(u_ctrl)
        if len(u_ctrl) != self.params_control.actuator_number:
            raise ValueError(f"Expected {self.params_control.actuator_number} control inputs, got {len(u_ctrl)}")
        for actuator, val in zip(self.params_control.actuator_list, u_ctrl):
            actuator.expression.u_ctrl = val

    def flush_actuators_u_ctrl(self) -> None:
        self.set_actuators_u_ctrl([0] * self.params_control.actuator_number)

    def get_actuators_u_ctrl(self) -> list:
        return [a.expression.u_ctrl for a in self.params_control.actuator_list]

    def _normalise_force_actuators(self) -> None:
        """Give every FORCE actuator its unit-L2-norm scaling η (reference ``actuator.py:308-312``)."""
        for a in self.params_control.actuator_list:
            if a.actuator_type is ACTUATOR_TYPE.FORCE and not getattr(a, "_normalised", True):
                a.expression.eta = 1.0
                nodal = a.expression.profile(self.th.node_coords)
                a.normalise(self._velocity_l2_norm(np.r_[nodal[:, 0], nodal[:, 1]]))

    def _gather_actuators_expressions(self):
        """Sum of FORCE-type expressions as a callable x → (n,2), or None when there is none."""
        self._normalise_force_actuators()
        forces = [a.expression for a in self.params_control.actuator_list if a.actuator_type is ACTUATOR_TYPE.FORCE]
        if not forces:
            return None
        return lambda x: sum(f(x) for f in forces)

    def make_measurement(self, up: Function) -> NDArray[np.float64]:
        return np.array([sensor.eval(up=up) for sensor in self.params_control.sensor_list])

    # ── boundary conditions ──────────────────────────────────────────────────
    def _make_BCs(self) -> BoundaryConditions:
        """Full-field BCs: uniform inlet profile + the perturbation BCs of the other boundaries
        (reference ``:329-337``)."""
        bcu_inlet = DirichletBC(self.W.sub(0), Constant((self.params_flow.uinf, 0)), self.boundaries.loc["inlet"].subdomain)
        bcs = self._make_bcs()
        return BoundaryConditions(bcu=[bcu_inlet] + bcs.bcu[1:], bcp=[])

    def _bc_tables(self) -> tuple[np.ndarray, np.ndarray]:
        """Dirichlet dofs and the per-actuator value profiles: g = profiles @ u_ctrl."""
        acts = self.params_control.actuator_list
        saved = self.get_actuators_u_ctrl()
        n_act = len(acts)
        try:
            self.flush_actuators_u_ctrl()
            dofs, base = combine_bcs(self.bc.bcu, self.th.N)
            if np.any(base != 0.0):
                raise NotImplementedError("perturbation BCs with a non-zero constant part are not supported")
            prof = np.zeros((len(dofs), max(n_act, 0)))
            for k in range(n_act):
                u = [0.0] * n_act
                u[k] = 1.0
                self.set_actuators_u_ctrl(u)
                d2, v = combine_bcs(self.bc.bcu, self.th.N)
                assert np.array_equal(d2, dofs)
                prof[:, k] = v
        finally:
            self.set_actuators_u_ctrl(saved)
        return dofs, prof

    def _force_tables(self) -> np.ndarray | None:
        acts = self.params_control.actuator_list
        if not any(a.actuator_type is ACTUATOR_TYPE.FORCE for a in acts):
            return None
        self._normalise_force_actuators()
        out = np.zeros((len(acts), 2 * self.th.nn))
        for k, a in enumerate(acts):
            if a.actuator_type is ACTUATOR_TYPE.FORCE:
                v = a.expression.profile(self.th.node_coords)
                out[k, : self.th.nn], out[k, self.th.nn :] = v[:, 0], v[:, 1]
        return out

    def _velocity_l2_norm(self, u: np.ndarray) -> float:
        """‖u‖_L2 of a nodal velocity field via the device mass matrix (``dolfin.norm``)."""
        dev = self.th.device()
        self._ensure_mass()
        return float(np.sqrt(2.0 * dev.energy(u)))

    def _ensure_mass(self) -> None:
        if not getattr(self, "_mass_ready", False):
            self.th.device().assemble_matrix(SLOT_MASS, mass=1.0, nu=0.0, pressure=0.0, divergence=0.0)
            self._mass_ready = True

    # ── steady state (reference :341-460) ────────────────────────────────────
    def compute_steady_state(self, u_ctrl: list, method: str = "newton", initial_guess: Function | None = None, max_iter: int = 10, **kwargs) -> None:
        self.set_actuators_u_ctrl(u_ctrl)
        f = self._gather_actuators_expressions()
        UP0 = self._define_initial_guess(initial_guess)
        ss = SteadyStateSolver(W=self.W, bcu=self._make_BCs().bcu, forms=self.forms, verbose=bool(self.verbose))
        self._mass_ready = False  # the steady solver may reuse the mass slot
        self._systems_ready = False  # ... and the BDF1 slot / boundary tables of the device handle
        if method == "newton":
            UP0 = ss.newton(UP0, f=f, max_iter=max_iter, **kwargs)
        elif method == "picard":
            UP0 = ss.picard(UP0, f=f, max_iter=max_iter, **kwargs)
        else:
            raise ValueError(f"method must be 'newton' or 'picard', got {method!r}")
        self._mass_ready = False
        U0, P0 = UP0.split(deepcopy=True)
        if self.params_save.save_every and self._is_writer():
            write_frame(self.paths.U0, U0, 0, name="U0")
            write_frame(self.paths.P0, P0, 0, name="P0")
            self.paths.steady_meta.parent.mkdir(parents=True, exist_ok=True)
            self.paths.steady_meta.write_text(json.dumps({"mesh_cells": self.mesh.num_cells}, indent=2))
        self._assign_steady_state(U0, P0)

    def load_steady_state(self, path_u_p: Sequence[Path] | None = None) -> None:
        paths = path_u_p or (self.paths.U0, self.paths.P0)
        self._check_steady_state_compatible(Path(paths[0]))
        U0, P0 = Function(self.V), Function(self.P)
        read_frame(paths[0], U0, 0, name="U0")
        read_frame(paths[1], P0, 0, name="P0")
        self._assign_steady_state(U0, P0)

    def _check_steady_state_compatible(self, u0_path: Path) -> None:
        meta_path = u0_path.parent / "meta.json"
        try:
            meta = json.loads(meta_path.read_text())
        except FileNotFoundError:
            meta = {}
        stored = meta.get("mesh_cells")
        current = self.mesh.num_cells
        if stored is not None and stored != current:
            raise ValueError(
                f"Steady-state checkpoint at {u0_path.parent} was written with {stored} mesh cells, but the current "
                f"mesh has {current}. Load a checkpoint from the same mesh, or recompute the steady state."
            )

    def _assign_steady_state(self, U0: Function, P0: Function) -> None:
        self.fields.U0 = U0
        self.fields.P0 = P0
        self.fields.UP0 = self.merge(U0, P0)
        self.E0 = 0.5 * self._velocity_l2_norm(U0.vector().array()) ** 2
        self._systems_ready = False

    def _define_initial_guess(self, initial_guess: Function | None = None) -> Function:
        if initial_guess is None:
            logger.info("Steady-state solver — no initial guess provided, using default")
            UP0 = Function(self.W)
            UP0.interpolate(self._default_steady_state_initial_guess())
        else:
            logger.info("Steady-state solver — using provided initial guess")
            UP0 = initial_guess
        return UP0

    # ── time stepping (reference :464-663) ───────────────────────────────────
    def initialize_time_stepping(self, Tstart: float = 0.0, ic: Function | None = None) -> None:
        restart_order = self.params_restart.restart_order if self.params_restart else "n/a"
        logger.info(f"Initialising from t={Tstart}, restart_order={restart_order}")
        self.fields._set_sync(None)
        if Tstart == 0.0:
            u_, p_, u_n, u_nn, p_n = self._initialize_with_ic(ic)
        else:
            u_, p_, u_n, u_nn, p_n = self._initialize_at_time(Tstart)
        self.fields.u_ = u_
        self.fields.p_ = p_
        self.fields.u_n = u_n
        self.fields.u_nn = u_nn
        self.fields.p_n = p_n
        self.fields.up_ = self.merge(u_, p_)
        self.first_step = True
        self._state_uploaded = False
        self.exporter.reset()
        self.y_meas = self.make_measurement(up=self.fields.ic.up)
        self._dE_host = 0.5 * self._velocity_l2_norm(u_.vector().array()) ** 2
        self.exporter.log_ic(t=self.params_time.Tstart, y_meas=self.y_meas, dE=self._dE_host)

    def _initialize_with_ic(self, ic: Function | None = None):
        self.order = "cn" if self.params_solver.time_scheme == "cn" else 1
        self._u_ctrl_prev = None
        self.iter = 0
        self.t = self.params_time.Tstart
        self.fields.ic = FlowField(up=Function(self.W) if ic is None else ic)
        if self.params_ic.amplitude:
            ic_pert = self._default_initial_perturbation(xloc=self.params_ic.xloc, yloc=self.params_ic.yloc, radius=self.params_ic.radius)
            self.fields.ic.up.vector()[:] = self.fields.ic.up.vector().array() + self.params_ic.amplitude * ic_pert.vector().array()
            self.fields.ic = FlowField(self.fields.ic.up)
        # projectm(ic.u, V, bcs=bc.bcu): L2 projection of a P2 field onto P2 is the identity, and
        # the BCs (built on W.sub(0)) do not apply to the separately constructed V — pinned by the
        # reference's cylinder regression constants.
        u_n = self.fields.ic.u.copy(deepcopy=True)
        u_nn = u_n.copy(deepcopy=True)
        p_n = self.fields.ic.p.copy(deepcopy=True)
        u_ = u_n.copy(deepcopy=True)
        p_ = p_n.copy(deepcopy=True)
        if self.params_save.save_every and self._is_writer():
            self.exporter.export_xdmf(u_n, u_nn, p_n, time=0.0, append=False, write_mesh=True, adjust_baseflow=1.0)
        return u_, p_, u_n, u_nn, p_n

    def _find_restart_source(self, Tstart: float):
        result = self._find_restart_from_json(Tstart)
        if result is not None:
            return result
        return self._find_restart_from_params(Tstart)

    def _find_restart_from_json(self, Tstart: float):
        """The run whose checkpoints cover ``Tstart``, from the ``meta_restart*.json`` sidecars an earlier run left in ``path_out``
        (reference flowsolver.py:564-577): (its metadata, index of the checkpoint written at Tstart, the directory) or None."""
        out_dir = Path(self.params_save.path_out)
        tol = 1e-10
        for sidecar in sorted(out_dir.glob("meta_restart*.json")):
            meta = json.loads(sidecar.read_text())
            written = int(meta["checkpoints_written"])
            if written == 0:
                continue
            first, spacing = meta["Tstart"], meta["dt"] * meta["save_every"]
            if not (first - tol <= Tstart <= first + spacing * written + tol):
                continue
            index = round((Tstart - first) / spacing)
            logger.info("Restart: found JSON sidecar %s, counter=%d", sidecar.name, index)
            return meta, index, out_dir
        return None

    def _find_restart_from_params(self, Tstart: float):
        if self.params_restart is None:
            raise FileNotFoundError(
                f"No JSON metadata sidecar found covering Tstart={Tstart} in {self.params_save.path_out}, and no ParamRestart was provided."
            )
        pr = self.params_restart
        step = pr.dt_old * pr.save_every_old
        counter = round((Tstart - pr.Trestartfrom) / step)
        meta = {"restart_order": pr.restart_order, "files": {"U": self.paths.U.name, "Uprev": self.paths.Uprev.name, "P": self.paths.P.name}}
        return meta, counter, Path(self.params_save.path_out)

    def _initialize_at_time(self, Tstart: float):
        meta, counter, base_dir = self._find_restart_source(Tstart)
        self.order = meta["restart_order"]
        self._u_ctrl_prev = None
        self.iter = 0
        self.t = Tstart
        U_path, Uprev_path, P_path = (base_dir / meta["files"][k] for k in ("U", "Uprev", "P"))
        U_, U_nn, P_ = Function(self.V), Function(self.V), Function(self.P)
        read_frame(U_path, U_, counter, name="U")
        read_frame(Uprev_path, U_nn, counter, name="U_n")
        read_frame(P_path, P_, counter, name="P")
        U_n, P_n = U_.copy(deepcopy=True), P_.copy(deepcopy=True)
        if self.fields.U0 is None:
            raise RuntimeError("no base flow: call load_steady_state() before restarting")
        if self.params_save.save_every and self._is_writer():
            # full fields were read: no base-flow adjustment (reference :633-643)
            self.exporter.export_xdmf(U_n, U_nn, P_n, time=Tstart, append=False, write_mesh=True, adjust_baseflow=0.0)
        U0v, P0v = self.fields.U0.vector().array(), self.fields.P0.vector().array()
        u_ = Function(self.V, U_.vector().array() - U0v)
        u_n = Function(self.V, U_n.vector().array() - U0v)
        u_nn = Function(self.V, U_nn.vector().array() - U0v)
        p_ = Function(self.P, P_.vector().array() - P0v)
        p_n = Function(self.P, P_n.vector().array() - P0v)
        self.fields.ic = FlowField(up=self.merge(u_, p_))
        return u_, p_, u_n, u_nn, p_n

    def _prepare_systems(self, u_n: Function | None = None, u_nn: Function | None = None) -> None:
        """Assemble both LHS operators on the device, eliminate Dirichlet dofs, factorise, and ship
        BC / force / sensor tables — the one-time work of reference ``:665-701``."""
        if self.fields.U0 is None:
            raise RuntimeError("no base flow: call compute_steady_state() or load_steady_state() first")
        dev = self.th.device()
        self._join_process_group(dev)
        U0 = self.fields.U0
        dofs, prof = self._bc_tables()
        dev.set_bc(dofs, prof)
        dev.set_pressure_pin(pressure_pin(self.th, dofs))  # enclosed flows only (lid-driven cavity)
        dev.set_factor_precision(self.factor_bits)
        dev.set_force(self._force_tables())
        dev.set_sensors([s.row(self) for s in self.params_control.sensor_list])
        dev.set_time_scheme(self.params_time.dt, self.params_solver.is_eq_nonlinear)
        self._ensure_mass()
        self.solvers: dict[int | str, Any] = {}
        scheme = self.params_solver.time_scheme
        orders = (("cn", SLOT_BDF1),) if scheme == "cn" else ((1, SLOT_BDF1), (2, SLOT_BDF2))
        for order, slot in orders:
            F = self.forms.transient(order=order, U0=U0, u_n=u_n, u_nn=u_nn, f=None, f_n=0.0)
            a = F.a
            dev.assemble_matrix(slot, mass=a.mass, nu=a.nu, adv=a.adv, lin=a.lin, adv_scale=a.adv_scale, lin_scale=a.lin_scale,
                                pressure=a.pressure, divergence=a.divergence)
            dev.apply_bc(slot)
            solver = self._make_solver(order=order)
            solver.set_operator(DeviceOperator(dev, slot))  # reference: solver.set_operator(A) (flowsolver.py:697)
            self.solvers[order] = solver
            if F.explicit is not None:
                from ._lib import SLOT_SCRATCH

                c = F.explicit
                dev.assemble_matrix(SLOT_SCRATCH, mass=c.mass, nu=c.nu, adv=c.adv, lin=c.lin, adv_scale=c.adv_scale,
                                    lin_scale=c.lin_scale, pressure=c.pressure, divergence=c.divergence)
                dev.set_rhs_operator(slot, dev.matrix(SLOT_SCRATCH)[:, : 2 * self.th.nn])
            else:
                dev.set_rhs_operator(slot, None)
        self._systems_ready = True

    def _join_process_group(self, dev) -> None:
        """One process per GPU: when ``torch.distributed`` is initialised with more than one rank the
        solver partitions the elimination tree over the ranks (RCCL inside the library).  ``self.comm`` (a
        :class:`flowcontrol_amd.comm.Comm`) overrides the detection — e.g. ranks that are threads of one process."""
        if getattr(self, "_joined", False):
            return
        self._joined = True
        from .comm import default_comm

        comm = self.comm if self.comm is not None else default_comm()
        if comm is None or comm.world == 1 or not getattr(self, "distributed", True):
            return
        self.comm = comm
        # no RCCL process group (CPU collectives, or several ranks sharing one GPU; FC_EXCHANGE=host asks for it): the exchange steps
        # of a time step are staged through the host; the arithmetic stays on the GPU
        import os

        from ._lib import FcCommInitError

        in_stream = comm.in_stream and os.environ.get("FC_EXCHANGE", "rccl") != "host"
        if not in_stream:
            dev.join(comm.rank, comm.world, comm.bcast, comm.allreduce)
            return
        why = None
        try:
            # the ranks settle inside join() -- over the process group's own collective, before and after the collective
            # ncclCommInitRank -- whether ALL of them have a communicator: either every rank returns, or every rank raises
            # FcCommInitError (a rank that did get one has given it back)
            dev.join(comm.rank, comm.world, comm.bcast, None, agree=comm.allreduce_max)
            return
        except FcCommInitError as err:
            why = err
        if os.environ.get("FC_EXCHANGE_FALLBACK", "1") == "0":
            raise why
        # the library's RCCL communicator could not be created although the process group works: same partition, same launch
        # sequence, exchanges staged through the host over the process group -- slower, and said so (DeviceSolver.comm_info()
        # reports transport "host", bench.py prints exchange_fallback)
        import sys

        self.exchange_fallback = str(why)
        print(f"[flowcontrol_amd] rank {comm.rank}: in-library RCCL communicator failed ({why}); exchanges go through the host "
              f"over the process group (FC_EXCHANGE_FALLBACK=0 makes this fatal)", file=sys.stderr, flush=True)
        dev.join(comm.rank, comm.world, comm.bcast, comm.allreduce)

    def _upload_state(self) -> None:
        f = self.fields
        self.th.device().set_state(f._store["u_n"].vector().array(), f._store["u_nn"].vector().array(), f._store["p_n"].vector().array())
        self._state_uploaded = True
        f._dirty = False
        f._stale = False
        self.fields._set_sync(self._download_fields)

    def _begin_stepping(self) -> None:
        """Operators on the device, state in HBM — and again whenever the host replaced a field (``fields.push()``)."""
        if self.first_step:
            if not self._systems_ready:
                self._prepare_systems(self.fields._store["u_n"], self.fields._store["u_nn"])
            if not self._state_uploaded:
                self._upload_state()
            self.first_step = False
        if self.fields._dirty:
            self._upload_state()  # fc_set_state also drops the speculated right-hand side and the divergence flag

    def _is_writer(self) -> bool:
        """Files (checkpoints, sidecars, CSV) are written by rank 0 only, as the reference does (exporter.py:260,266)."""
        if self.comm is not None:
            return self.comm.rank == 0
        from .utils import get_rank

        return get_rank() == 0

    def _download_fields(self) -> None:
        dev = self.th.device()
        u_n, u_nn, p_n = dev.get_state()
        if dev.world > 1:
            # every rank holds its own dofs (+ the replicated root): merge by masked sum
            m = dev.owned_mask()
            nn2 = 2 * self.th.nn
            flat = np.concatenate([np.where(m[:nn2], u_n, 0.0), np.where(m[:nn2], u_nn, 0.0), np.where(m[nn2:], p_n, 0.0)])
            self.comm.allreduce(flat)
            u_n, u_nn, p_n = flat[:nn2], flat[nn2 : 2 * nn2], flat[2 * nn2 :]
        st = self.fields._store
        st["u_n"].vector().set_local(u_n)
        st["u_nn"].vector().set_local(u_nn)
        st["p_n"].vector().set_local(p_n)
        st["u_"] = Function(self.V, u_n)
        st["p_"] = Function(self.P, p_n)
        st["up_"] = Function(self.W, np.r_[u_n, p_n])

    def step(self, u_ctrl: NDArray[np.float64]) -> NDArray[np.float64] | None:
        """Advance by one Δt; returns the measurement vector, or ``None`` if the solver diverged and
        ``params_solver.throw_error`` is False (reference ``:703-799``).  A failed step leaves ``u_n``, ``u_nn``, ``p_n`` as they
        were, as in the reference (which detects the non-finite velocity before it shifts its fields, ``:727-751``): the step's
        solution sits in a buffer of its own and is simply not adopted (``fc_undo_step``)."""
        self._begin_stepping()
        t0 = time.time()
        u_ctrl = np.atleast_1d(np.asarray(u_ctrl, dtype=np.float64))
        next_iter = self.iter + 1
        want_energy = self._niter_multiple_of(next_iter, self.params_save.energy_every)
        u_force = None
        if self.order == "cn":
            # Crank–Nicolson averages the body force: ½(f^{n+1} + f^n), f^0 = 0 (flowsolver.py:681-686,755-758)
            prev = np.zeros_like(u_ctrl) if self._u_ctrl_prev is None else self._u_ctrl_prev
            u_force = 0.5 * (u_ctrl + prev)
        try:
            solver = self.solvers[self.order]
            slot = SLOT_BDF2 if self.order == 2 else SLOT_BDF1
            if isinstance(solver, _DeviceNDSolver):
                dev = self.th.device()
                dev.step_begin(slot, u_ctrl, compute_energy=want_energy, u_force=u_force)  # the GPU works from here on ...
                # ... while the host does what does not depend on this step's result: the previous step's log row and progress
                # line, the actuators' bookkeeping (the reference does all of it inside the step, flowsolver.py:721-799)
                try:
                    self._flush_log()
                    self.set_actuators_u_ctrl(u_ctrl)
                except BaseException:
                    # the host's bookkeeping failed (exporter I/O ...) with a step in flight: end it and take it back, so that
                    # the handle is not left refusing every later fc_step_begin and the device state matches self.iter
                    try:
                        dev.step_end(early=True)
                        dev.undo_step()
                    except Exception:  # noqa: BLE001 -- the original error is the one to report
                        pass
                    raise
                # back as soon as the measurements are: energy and residual of this step follow (self._collect)
                y, dE, info = dev.step_end(early=True)
                self._late_pending = True
            else:
                self._flush_log()
                self.set_actuators_u_ctrl(u_ctrl)
                y, dE, info = self._step_with_plugin_solver(solver, slot, u_ctrl, want_energy)
                self.solve_info = info
        except FcDiverged:
            logger.critical("Solver diverged (Inf detected)")
            # the reference detects the non-finite velocity BEFORE it shifts the fields (flowsolver.py:727-751): u_n, u_nn, p_n stay
            # what they were.  On the device the step's solution sits in a buffer of its own: it is simply not adopted.
            if isinstance(solver, _DeviceNDSolver):
                self.th.device().undo_step()
            self.fields._mark_stale()  # the host mirrors are re-read from the device
            if not self.params_solver.throw_error:
                return None
            raise RuntimeError("Failed solving: Inf found in solution")
        if info is not None and info[1] > self.residual_tol:  # (plug-in solvers report none: NaN compares False)
            self._residual_breach = (float(info[1]), next_iter)
        self.iter = next_iter
        self.t = self.params_time.Tstart + self.iter * self.params_time.dt
        self._u_ctrl_prev = u_ctrl.copy()
        if self.params_solver.time_scheme != "cn":
            self.order = 2
        self.fields._mark_stale()
        self.y_meas = y
        # this step's log row is booked while the next step runs (or as soon as anybody looks at the exporter); dE = None: to be collected
        self._pending_log = (self.iter, self.t, self._u_ctrl_prev, y, dE if want_energy else np.nan, time.time() - t0)
        checkpoint = self._niter_multiple_of(self.iter, self.params_save.save_every)
        if checkpoint:
            self._collect()  # a checkpoint is written only once THIS step's residual verdict exists
        # the residual monitor is this solver's own check (the reference makes none): its verdict on a step arrives with the NEXT step
        # at the latest (with the step itself before a checkpoint) -- the factors (or the system) are broken, do not keep stepping silently
        if self._report_breach():
            return None
        if checkpoint:
            self._checkpoint()
        return self.y_meas

    def _step_with_plugin_solver(self, solver, slot: int, u_ctrl, want_energy: bool):
        """A ``_make_solver`` override that returns its own solver object (the reference's documented plug-in point,
        ``docs/numerical-details.md:44-48``, ``flowsolver.py:812-814``) is honoured as the reference honours it: the
        right-hand side is assembled on the device, ``solver.solve(x, b)`` runs wherever the plug-in runs, and the result
        becomes the new state (``flowsolver.py:728-751``).  Vectors cross PCIe every step: this is the compatibility path."""
        if self.order == "cn":
            raise NotImplementedError("plug-in solvers with the Crank-Nicolson scheme")
        dev = self.th.device()
        if dev.perm is None:
            # no device factorisation was set up (the plug-in solves): the assembly kernels still want a row ordering
            from ._lib import check

            dev.perm = np.arange(dev.N, dtype=np.int32)
            check(dev.lib.fc_set_permutation(dev._h, dev.perm))
        b = dev.assemble_rhs(slot, u_ctrl)
        x = np.zeros(dev.N)
        solver.solve(x, b)
        nn2 = 2 * self.th.nn
        if not np.all(np.isfinite(x[:nn2])):
            from ._lib import FC_ERR_DIVERGED

            raise FcDiverged(FC_ERR_DIVERGED, "non-finite velocity after solve")
        u_prev, _, _ = dev.get_state()
        dev.set_state(x[:nn2], u_prev, x[nn2:])
        y = dev.measure(x)
        dE = dev.energy(x[:nn2]) if want_energy else float("nan")
        return y, dE, np.array([0.0, float("nan"), float(np.linalg.norm(b)), 0.0])

    def _checkpoint(self) -> None:
        u_n, u_nn, p_n = self.fields.u_n, self.fields.u_nn, self.fields.p_n  # collective on several ranks
        if self._is_writer():
            self.exporter.export_xdmf(u_n, u_nn, p_n, time=self.t, adjust_baseflow=1.0)
            self.exporter.write_metadata(restart_order="cn" if self.params_solver.time_scheme == "cn" else 2)
            self.exporter.write_timeseries()

    def run(self, n_steps: int, u_ctrl) -> tuple[np.ndarray, np.ndarray] | None:
        """Open-loop batch of ``n_steps`` steps with no host synchronisation in between (device
        extension; not in the reference).  ``u_ctrl``: (n_act,) constant or (n_steps, n_act).

        Same log as ``n_steps`` calls of :meth:`step`: dE is NaN off the ``energy_every`` multiples, checkpoints are
        written at the ``save_every`` multiples (the batch is cut there).  On a divergence the device state has
        advanced to the non-finite step; ``None`` is returned when ``throw_error`` is False."""
        self._begin_stepping()
        u = np.asarray(u_ctrl, dtype=np.float64)
        if self.order == "cn":
            # Crank-Nicolson needs the previous control for its averaged forcing: the steps go one by one (same log)
            ys, dEs = [], []
            for k in range(n_steps):
                y = self.step(u[k] if u.ndim == 2 else u)
                if y is None:
                    return None
                ys.append(np.asarray(y, dtype=np.float64).copy())
                dEs.append(float(self.exporter._records[-1]["dE"]))
            return np.vstack(ys), np.asarray(dEs)
        every = self.params_save.energy_every
        ys, dEs = [], []
        done = 0
        while done < n_steps:
            n = n_steps - done
            if self.params_save.save_every:  # stop at the next checkpoint
                n = min(n, self.params_save.save_every - self.iter % self.params_save.save_every)
            t0 = time.time()
            try:
                y, dE = self.th.device().run(SLOT_BDF1 if self.order == 1 else SLOT_BDF2, n, u[done : done + n] if u.ndim == 2 else u,
                                             compute_energy=bool(every))
            except FcDiverged:
                self.fields._mark_stale()
                if not self.params_solver.throw_error:
                    return None
                raise RuntimeError("Failed solving: Inf found in solution")
            runtime = (time.time() - t0) / n
            for s in range(n):
                self.iter += 1
                self.t = self.params_time.Tstart + self.iter * self.params_time.dt
                us = u[done + s] if u.ndim == 2 else np.atleast_1d(u)
                if not self._niter_multiple_of(self.iter, every):
                    dE[s] = np.nan
                self.exporter.log(u_ctrl=us, y_meas=y[s], dE=dE[s], t=self.t, runtime=runtime)
            self.order = 2
            self.y_meas = y[-1].copy()
            self.fields._mark_stale()
            ys.append(y), dEs.append(dE)
            done += n
            if self._niter_multiple_of(self.iter, self.params_save.save_every):
                self._checkpoint()
        return np.vstack(ys), np.concatenate(dEs)

    def write_timeseries(self) -> None:
        self._flush_log()
        self._report_breach()  # a breach on the last step of a run surfaces here at the latest
        if self._is_writer():
            self.exporter.write_timeseries()

    @property
    def timeseries(self) -> pd.DataFrame:
        self._flush_log()
        self._report_breach()
        return self.exporter.to_dataframe()

    # ── solver plug-in point (reference :812-819; docs/numerical-details.md:44-48) ──
    def _make_solver(self, order: int | str) -> Any:
        """Return the linear solver of ``order``: an object with ``set_operator(slot)``.  The default
        factorises the device matrix with the nested-dissection selected inverse."""
        return _DeviceNDSolver(self)

    def _solver_diverged(self, field: Function) -> bool:
        return not bool(np.all(np.isfinite(field.vector().array())))

    def _niter_multiple_of(self, iter: int, divider: int) -> bool:
        return bool(divider and not iter % divider)

    # ── energy ───────────────────────────────────────────────────────────────
    def compute_perturbation_energy(self) -> float:
        """½‖u'‖²_L2 of the current perturbation (reference ``:827-829``)."""
        return 0.5 * self._velocity_l2_norm(self.fields.u_.vector().array()) ** 2

    def compute_energy_field(self, export: bool = False, filename=None):
        """u'·u' as a field on the P4 Lagrange nodes of the mesh (reference ``:831-841``: ``projectm(dot(u_, u_), P4)``).  The product of two
        P2 fields is piecewise P4 and continuous, so its L2 projection onto the P4 space IS its nodal interpolant: the values below are exact.
        Returns an :class:`EnergyField` (``coords`` (n, 2), ``values`` (n,), ``vector().get_local()``); nodes: vertices, three per edge (from the
        lower to the higher vertex id), three per cell.  ``export=True`` writes ``coords`` / ``values`` to ``filename`` (``.npz``)."""
        th, m = self.th, self.th.mesh
        u = self.fields.u_.vector().array()
        nv, ne, nc = th.nv, th.ne, th.nc
        coords = np.empty((nv + 3 * ne + 3 * nc, 2))
        values = np.empty(nv + 3 * ne + 3 * nc)
        # the 15 P4 nodes of the reference cell in barycentric coordinates: vertices, edge k (opposite vertex k) from local vertex k + 1 to k + 2, interior
        lam = [np.eye(3)[k] for k in range(3)]
        for k in range(3):
            a, b = (k + 1) % 3, (k + 2) % 3
            for j in (1, 2, 3):
                w = np.zeros(3)
                w[a], w[b] = 1.0 - j / 4.0, j / 4.0
                lam.append(w)
        for k in range(3):
            w = np.full(3, 0.25)
            w[k] = 0.5
            lam.append(w)
        lam = np.array(lam)  # (15, 3)
        phi = np.empty((15, 6))  # P2 basis (vertex nodes, then the midpoint of the edge opposite vertex k) at those points
        for k in range(3):
            phi[:, k] = lam[:, k] * (2.0 * lam[:, k] - 1.0)
            phi[:, 3 + k] = 4.0 * lam[:, (k + 1) % 3] * lam[:, (k + 2) % 3]
        ux, uy = u[: th.nn][th.cell_nodes], u[th.nn :][th.cell_nodes]  # (nc, 6)
        e = (ux @ phi.T) ** 2 + (uy @ phi.T) ** 2  # (nc, 15)
        xy = np.einsum("pk,ckd->cpd", lam, m.coords[m.cells])  # (nc, 15, 2)
        # global ids: an edge's three nodes run from its lower to its higher vertex id
        ids = np.empty((nc, 15), dtype=np.int64)
        ids[:, :3] = m.cells
        for k in range(3):
            a, b = m.cells[:, (k + 1) % 3], m.cells[:, (k + 2) % 3]
            fwd = a < b
            for j in range(3):
                ids[:, 3 + 3 * k + j] = nv + 3 * m.cell_edges[:, k] + np.where(fwd, j, 2 - j)
        ids[:, 12:] = nv + 3 * ne + 3 * np.arange(nc)[:, None] + np.arange(3)[None, :]
        values[ids.reshape(-1)] = e.reshape(-1)
        coords[ids.reshape(-1)] = xy.reshape(-1, 2)
        field = EnergyField(coords, values)
        if export:
            np.savez(filename, coords=coords, values=values)
        return field

    # ── utilities ────────────────────────────────────────────────────────────
    def merge(self, u: Function, p: Function) -> Function:
        return Function(self.W, np.r_[u.vector().array(), p.vector().array()])

    def get_subdomain(self, name: str):
        return self.boundaries.loc[name].subdomain

    # ── default IC / perturbation (reference :887-912) ───────────────────────
    def _default_steady_state_initial_guess(self):
        uinf = self.params_flow.uinf

        def uniform(x):
            out = np.zeros((x.shape[0], 3))
            out[:, 0] = uinf
            return out

        return uniform

    def _default_initial_perturbation(self, xloc: float = 0.0, yloc: float = 0.0, radius: float = 1.0) -> Function:
        return self._perturbation_div0(xloc, yloc, radius)

    def _perturbation_div0(self, xloc: float = 0.0, yloc: float = 0.0, radius: float = 1.0) -> Function:
        """Divergence-free Gaussian vortex ψ = ¼ exp(−r²/2s²), u = (∂ψ/∂y, −∂ψ/∂x), interpolated at
        the P2 nodes (``utils/physics.py:32-56``: the L2 projection of a P2 interpolant onto P2 is
        the identity), merged with the base-flow pressure (``flowsolver.py:908-912``)."""
        u = Function(self.V)
        if radius > 0:
            x = self.th.node_coords
            dx, dy = x[:, 0] - xloc, x[:, 1] - yloc
            psi = 0.25 * np.exp(-0.5 * (dx * dx + dy * dy) / radius**2)
            u.vector()[:] = np.r_[-dy / radius**2 * psi, dx / radius**2 * psi]
        else:
            logger.warning(f"_perturbation_div0: radius={radius} <= 0, returning zero field")
        p = self.fields.P0.copy(deepcopy=True) if self.fields.P0 is not None else Function(self.P)
        return self.merge(u=u, p=p)

    # ── abstract ─────────────────────────────────────────────────────────────
    @abstractmethod
    def _make_boundaries(self) -> pd.DataFrame:
        ...

    @abstractmethod
    def _make_bcs(self) -> BoundaryConditions:
        """Perturbation-field BCs; ``bcu[0]`` MUST be the inlet BC (``_make_BCs`` replaces it)."""

    @classmethod
    @abstractmethod
    def make_default(cls, **kwargs) -> "FlowSolver":
        ...


class DeviceOperator:
    """The assembled, BC-eliminated system matrix of one time order as ``_make_solver`` products receive it in
    ``set_operator(A)`` (reference ``flowsolver.py:697``: a ``dolfin.PETScMatrix``).  The values live in a matrix slot of the
    device handle; a solver written against the reference contract reads them as CSR — ``A.mat().getValuesCSR()`` is what
    ``as_backend_type(A).mat().getValuesCSR()`` gives in the reference — or as a scipy matrix (``A.tocsr()``)."""

    def __init__(self, dev, slot: int):
        self.dev, self.slot = dev, int(slot)
        self.shape = (dev.N, dev.N)

    def tocsr(self):
        return self.dev.matrix(self.slot)

    def mat(self):
        return self

    def getValuesCSR(self):
        A = self.tocsr()
        return A.indptr, A.indices, A.data

    def array(self) -> np.ndarray:
        return self.tocsr().toarray()


class _DeviceNDSolver:
    """Default ``_make_solver`` product: factorise-once / apply-many on the device.

    ``set_operator(A)`` follows the reference contract (``flowsolver.py:697``): ``A`` is the operator — a
    :class:`DeviceOperator` (what ``FlowSolver`` passes: the values are on the device already, only the numeric
    factorisation runs), any scipy sparse matrix on (a subset of) the Taylor–Hood pattern (its values are uploaded into the
    order's slot first), or, as before, a bare slot id."""

    def __init__(self, fs: FlowSolver, slot: int | None = None):
        self.fs = fs
        self.slot: int | None = slot

    def set_operator(self, A) -> None:
        fs = self.fs
        dev = fs.th.device()
        if isinstance(A, DeviceOperator):
            self.slot = A.slot
        elif isinstance(A, (int, np.integer)):
            self.slot = int(A)
        else:
            import scipy.sparse as sp

            if self.slot is None:
                raise ValueError("set_operator(matrix): construct the solver with the slot that takes the values")
            M = sp.csr_matrix(A)
            if M.shape != (dev.N, dev.N):
                raise ValueError(f"operator has shape {M.shape}, expected {(dev.N, dev.N)}")
            # values onto the handle's pattern (an entry outside it cannot be represented: refuse rather than drop it)
            M.sort_indices()
            N = dev.N
            key = np.repeat(np.arange(N, dtype=np.int64), np.diff(M.indptr)) * N + M.indices
            pkey = np.repeat(np.arange(N, dtype=np.int64), np.diff(dev.rowptr)) * N + dev.colidx
            pos = np.searchsorted(pkey, key)
            bad = (pos >= pkey.size) | (pkey[np.minimum(pos, pkey.size - 1)] != key)
            if np.any(bad & (M.data != 0.0)):
                raise ValueError("the operator has entries outside the Taylor–Hood pattern of this mesh")
            vals = np.zeros(dev.nnz)
            vals[pos[~bad]] = M.data[~bad]
            dev.set_matrix_values(self.slot, vals)
        if fs.krylov_precond is not None:
            if fs.krylov_precond != "schur_amg":
                raise ValueError("krylov_precond: None (factors) or 'schur_amg'")
            dev.setup_krylov(self.slot, sweeps=fs.krylov_sweeps, method=fs.krylov_method, max_iter=fs.krylov_max_iter, rtol=fs.krylov_rtol,
                             check_residual=-1 if fs.check_residual_every is None else fs.check_residual_every)
            return
        dev.setup_solver(self.slot, depth=fs.nd_depth, refine=fs.refine_steps, truncate=fs.nd_truncate, check_residual=-1 if fs.check_residual_every is None else fs.check_residual_every)
        if fs.nd_truncate or fs.factor_bits != 64:
            dev.set_solver_options(refine=fs.krylov_max_iter, check_residual=-1 if fs.check_residual_every is None else fs.check_residual_every,
                                   method=fs.krylov_method, rtol=fs.krylov_rtol)

    def solve(self, x: np.ndarray, b: np.ndarray) -> None:
        sol, _ = self.fs.th.device().solve(self.slot, b)
        x[:] = sol
