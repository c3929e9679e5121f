/*
 * fc_hip.h — C ABI of libfc_hip.so: the MI355X (gfx950) implementation of FlowControl's
 * per-timestep hot path (FEM assembly of the semi-implicit Navier–Stokes forms + the linear
 * solve that advances (v, p) by one Δt).
 *
 * The reference (williamjussiau/FlowControl) has no FFI of its own: its hot path is
 * `FlowSolver.step` (src/flowcontrol/flowsolver.py:703-799) driving two third-party dolfin
 * objects, `dolfin.SystemAssembler` (:693-696, :728) and `dolfin.LUSolver("mumps")`
 * (:697, :729, :812-814).  Every entry point below names the reference interface it replaces.
 * The binding a reference maintainer would add is the ctypes stub in INTEGRATION.md.
 *
 * Conventions
 *   - plain C, no torch / C++ types in any signature; caller owns all host buffers, the library
 *     owns all device memory; one opaque handle per solver, one HIP stream per handle,
 *     not re-entrant per handle.
 *   - every function returns an int status: 0 = ok, <0 = error class (FC_ERR_*); nothing throws
 *     across the boundary.  fc_last_error() returns a message for the calling thread.
 *   - all floating point data is IEEE fp64 (the reference computes in fp64 throughout);
 *     all index data is int32 unless stated.
 *   - mixed-space vector layout "W": [ux(nn) | uy(nn) | p(nv)], nn = nv + ne P2 scalar nodes
 *     (vertex v → v, edge e → nv + e); N = 2 nn + nv.
 */
#ifndef FC_HIP_H
#define FC_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct fc_ctx* fc_handle;

enum {
  FC_OK = 0,
  FC_ERR_INVALID = -1,     /* bad argument / call order */
  FC_ERR_HIP = -2,         /* HIP runtime error (no device, OOM, launch failure) */
  FC_ERR_DIVERGED = -3,    /* non-finite velocity after the solve (flowsolver.py:731,816-819) */
  FC_ERR_NOT_CONVERGED = -4, /* Krylov / refinement did not reach the requested tolerance */
  FC_ERR_NOT_READY = -5    /* a required setup call is missing */
};

/* matrix slots: independent value arrays on the shared Taylor–Hood CSR pattern */
enum { FC_SLOT_BDF1 = 0, FC_SLOT_BDF2 = 1, FC_SLOT_MASS = 2, FC_SLOT_SCRATCH = 3, FC_NUM_SLOTS = 4 };

/* Krylov / refinement method of fc_solve / fc_step */
enum { FC_METHOD_REFINE = 0, FC_METHOD_BICGSTAB = 1, FC_METHOD_GMRES = 2 };

const char* fc_last_error(void);
int fc_device_count(int* count);

/* ── construction: replaces FlowSolver._make_mesh / _make_function_spaces
 *    (flowsolver.py:233-250) on the device side.  cells are CCW vertex triples,
 *    cell_edges[c][k] is the edge opposite local vertex k. ------------------------------------ */
int fc_create(fc_handle* out, int device, int32_t nv, int32_t ne, int32_t nc,
              const double* coords /* [nv][2] */, const int32_t* cells /* [nc][3] */,
              const int32_t* cell_edges /* [nc][3] */);
int fc_destroy(fc_handle h);

/* sizes of the mixed space and of the CSR pattern (full 15x15 element coupling minus the
 * pressure-pressure block) */
int fc_get_sizes(fc_handle h, int64_t* N, int64_t* nnz, int64_t* nn);
int fc_get_pattern(fc_handle h, int32_t* rowptr /* [N+1] */, int32_t* colidx /* [nnz] */);

/* ── bilinear-form assembly: replaces dolfin.SystemAssembler.assemble(A) (flowsolver.py:693-696),
 *    dolfin.assemble(a) in SteadyStateSolver.picard (steadystate.py:137) and the Jacobian
 *    assembly of OperatorGetter.get_A (operatorgetter.py:79-80).  Element loop on the device:
 *      mass (u,v) + adv_scale ((adv.grad)u, v) + lin_scale ((u.grad)lin, v) + nu (grad u, grad v)
 *      + pressure (p, div v) + divergence (q, div u)
 *    adv / lin are host velocity fields [2 nn] or NULL.  No boundary conditions. ------------- */
int fc_assemble_matrix(fc_handle h, int slot, double mass, double nu, const double* adv,
                       double adv_scale, const double* lin, double lin_scale, double pressure,
                       double divergence);
int fc_get_matrix_values(fc_handle h, int slot, double* vals /* [nnz] */);
int fc_set_matrix_values(fc_handle h, int slot, const double* vals /* [nnz] */);

/* y = A_slot x on the device (original numbering); parity hook + SpMV roofline probe */
int fc_spmv(fc_handle h, int slot, const double* x /* [N] */, double* y /* [N] */);

/* ── Dirichlet data: replaces the DirichletBC list handed to SystemAssembler
 *    (flowsolver.py:693; case files' _make_bcs).  Every actuator expression is linear in u_ctrl
 *    (actuator.py:190-199,241-251,269-276), so the value on BC dof i is
 *    sum_k profiles[i][k] * u_ctrl[k]. -------------------------------------------------------- */
int fc_set_bc(fc_handle h, int32_t n_bc, const int32_t* bc_dofs /* [n_bc] W indices */,
              int32_t n_act /* <= 32 for fc_step */, const double* profiles /* [n_bc][n_act] */);
/* body force of FORCE-type actuators (actuator.py:297-313): nodal P2 values per unit u_ctrl */
int fc_set_force(fc_handle h, int32_t n_act, const double* profiles /* [n_act][2 nn] or NULL */);
/* sensors as sparse rows of the functional up -> y (sensor.py:96-98,166-197; utils/mpi.py:22-37) */
int fc_set_sensors(fc_handle h, int32_t n_sens, const int32_t* rowptr /* [n_sens+1] */,
                   const int32_t* idx, const double* w);
/* time scheme constants (flowsolverparameters.py ParamTime.dt, ParamSolver.is_eq_nonlinear) */
int fc_set_time_scheme(fc_handle h, double dt, int nonlinear);

/* SystemAssembler's symmetric Dirichlet elimination on a slot: stores the lifting vectors
 * A[:, D] * profile_k for `order_slot`, then zeroes BC rows and columns and puts 1 on the
 * diagonal. */
int fc_apply_bc(fc_handle h, int slot);

/* COMPRESSED factors, the memory-lean Krylov mode (reference plug-in point flowsolver.py:812-814; north_star: "HIP
 * BiCGStab/GMRES ... preconditioning"): the selected inverse is computed in fp64 front by front as always, but its values are
 * STORED rounded to fp32 (bits = 32: 50 % of the memory) or bfloat16 (bits = 16: 25 %) — the fp64 array never exists — and
 * applied with fp64 accumulation as right preconditioner of the device GMRES / BiCGStab (cylinder operator: 2 / ~9 GMRES
 * iterations to 1e-12).  Call before fc_setup_solver; a change lays the slots out anew.  fc_setup_solver then accepts the
 * factors through a GMRES probe (<= 40 iterations); fc_solve / fc_step need FC_METHOD_GMRES or FC_METHOD_BICGSTAB. */
int fc_set_factor_precision(fc_handle h, int bits /* 64 (default, exact), 32, 16 */);
int fc_get_factor_storage(fc_handle h, int slot, int32_t* bits, int64_t* bytes /* bytes of factor values held for the slot */);
/* THE solver setup in one call (what LUSolver.set_operator + the first solve cost the reference, flowsolver.py:697,729):
 * symbolic analysis inside the library from the mesh the handle holds (element-based nested-dissection tree of
 * `depth` bisections — 0: leaves of ~12 cells — fused `merge` at a time; on a handle with a communicator / host exchange
 * the root is nranks-ary and this rank lays out its own sub-tree and the root), permutation, segment tables, workgroup
 * tiles, factorisation plan, task dependencies, then the numeric factorisation of the slot's current matrix on the
 * device and the acceptance solve of fc_accept_factors (below).  truncate = d > 0: only the tree levels >= d
 * are factorised (memory-lean preconditioner; use FC_METHOD_GMRES / FC_METHOD_BICGSTAB afterwards).  A later call for the
 * same slot redoes only the numeric phase.  Needs: fc_set_bc, fc_assemble_matrix(slot), fc_apply_bc(slot).  (The
 * array-level entry points a caller with an analysis of its own would use -- fc_set_permutation, fc_solver_setup,
 * fc_factor_plan ... -- are declared in fc_hip_internal.h with the bench / debug hooks; the boundary is this file.) */
int fc_setup_solver(fc_handle h, int slot, int32_t depth, int32_t merge, int32_t truncate, int32_t refine, int32_t check_residual);
int fc_get_permutation(fc_handle h, int32_t* perm /* [N] new -> old */);
/* info[10]: factor values stored on this rank, of the whole tree, swept by this rank per solve; stages; tree depth;
 * exchanged rows, the two exchange stages; cells assembled by this rank; truncate */
int fc_get_solver_info(fc_handle h, int slot, int64_t* info /* [10] */);
/* enclosed flows (velocity prescribed on the whole boundary: pressure defined up to a constant): a positive shift on the
 * diagonal of ONE pressure dof inside the factorisation of fc_setup_solver (dof = -1: none); cf. fc_set_front_shifts */
int fc_set_pressure_pin(fc_handle h, int32_t dof, double shift);
/* device milliseconds of the slot's last numeric factorisation (fc_refactor, also inside fc_setup_solver) */
int fc_get_refactor_ms(fc_handle h, int slot, double* ms);
int fc_get_local_cells(fc_handle h, int32_t* cells /* [info[8] of fc_get_solver_info] */);
/* out[4]: cells of this rank (right-hand side, energy), cells whose element matrices fc_assemble_matrix computes on this handle (its own
 * and the other ranks' cells that touch a root dof: the rows a rank owns and the root's rows are complete, the other ranks' rows are
 * never read), lead rank (0 / 1), ranks.  A single-GPU handle: all cells, all cells, 1, 1. */
int fc_get_partition_info(fc_handle h, int32_t* out);
/* (On a partitioned handle fc_get_matrix_values returns this rank's view -- complete rows for the dofs it owns and the root's, partial
 * rows elsewhere -- and fc_spmv is a collective: every rank calls it with the same x and receives the whole product.) */
int fc_get_rowkind(fc_handle h, uint8_t* rowkind /* [N]: 0 other rank, 1 owned, 2 root (all 1 on a single-GPU handle) */);
/* Numeric factorisation ON THE DEVICE of the slot's current matrix values (after fc_assemble_matrix + fc_apply_bc) with the
 * structure of the first fc_setup_solver: what `solver.set_operator(A)` costs in the reference (flowsolver.py:697,812-814 ->
 * PETSc/MUMPS numeric phase; every Newton / Picard iteration of steadystate.py:60-159).  Multifrontal, all fronts of a tree
 * level together, blocked Gauss-Jordan steps with threshold partial pivoting, trailing updates on the fp64 matrix cores
 * (csrc/fc_front.hip.h); no vendor BLAS / LAPACK.  ms_out (optional): device time.  On a partitioned handle every rank
 * factorises its own sub-tree; the root front is summed over the ranks once (exchange). */
int fc_refactor(fc_handle h, int slot, double* ms_out);
/* Acceptance solve of a slot's freshly computed factors (fc_setup_solver runs it itself; call it after a bare fc_refactor): a fixed
 * right-hand side, residual against the matrix.  < 1e-10 by the direct apply: exact factors (time-step operators give <= 1e-12).
 * Between 1e-10 and 1e-2 (pivoting confined to the pivot blocks lost digits on an ill-conditioned operator, e.g. a steady Oseen
 * operator far beyond the Reynolds number the mesh resolves, where sparse LU with partial pivoting still reaches 1e-12): the factors
 * are kept as a PRECONDITIONER if GMRES reaches 1e-8 -- or, on a single-device handle, a normwise backward error
 * |r| / (|A|_F |x| + |b|) < 1e-13 -- within 40 iterations; fc_solve / fc_step on this slot then run GMRES(60, 1e-12, restarted on
 * the true residual until it stagnates) by themselves whenever FC_METHOD_REFINE is selected (*inexact_out = 1; the batched calls
 * refuse such a slot).  Anything else is FC_ERR_HIP.  residual_out: the direct apply's relative residual.  A collective on a
 * partitioned handle.  fc_refactor clears the flag. */
int fc_accept_factors(fc_handle h, int slot, double* residual_out, int32_t* inexact_out);
int fc_get_factors_inexact(fc_handle h, int slot, int32_t* inexact /* 1: the slot's factors serve as GMRES preconditioner */);
/* optional explicit operator C of the right-hand side, b -= C u_n (rows in the solver's permuted
 * numbering, columns = velocity dofs in W numbering): the explicit half of the linear terms of the
 * Crank-Nicolson form (NSForms._cn, nsforms.py:191-236).  rowptr == NULL removes it. */
int fc_set_rhs_operator(fc_handle h, int slot, const int32_t* rowptr, const int32_t* col, const double* val);
/* the slot's matrix changed (fc_assemble_matrix + fc_apply_bc) but its factors are kept as they are: only the
 * permuted copy that SpMV / residuals use is refreshed.  With FC_METHOD_BICGSTAB the old factors then act as
 * preconditioner of the new operator (Picard / Newton iterations of steadystate.py:60-159 between two
 * refactorisations; the reference re-factorises with MUMPS at every iteration). */
int fc_update_operator(fc_handle h, int slot);
/* method: FC_METHOD_*; max_iter: refinement sweeps (REFINE) or the Krylov iteration cap; rtol: Krylov target;
 * check_residual: 0 = no residual monitor, n >= 1 = fc_step / fc_run form |b - A x| / |b| on every n-th step of the handle (info[1];
 * NaN on the steps in between) -- the reference never forms it (flowsolver.py:728-737 tests finiteness only), n > 1 amortises the
 * pass over the system matrix it costs; -1 = auto: every step while the slot's factors stay in the 256 MiB Infinity Cache (the pass
 * hides beside the next step), every 8th step where they stream from HBM (there it costs ~10 % of a step).  The non-finite test
 * runs on every step regardless. */
int fc_set_solver_options(fc_handle h, int method, int max_iter, double rtol, int check_residual);

/* FACTORISATION-FREE Krylov mode (reference plug-in point FlowSolver._make_solver, flowsolver.py:812-814: "any object with
 * set_operator / solve"; BASELINE.json north_star: "HIP BiCGStab/GMRES with CSR SpMV and block-Jacobi/ILU(0) preconditioning").
 * Instead of fc_setup_solver: NOTHING is factorised.  The slot's solves and time steps run the device GMRES / BiCGStab on the
 * permuted system matrix, right-preconditioned by a SIMPLE-type block preconditioner built from the assembled values alone:
 *     u  = `sweeps` damped-Jacobi sweeps on the velocity block F         (the time-step operators are mass dominated)
 *     zp = one smoothed-aggregation AMG V(1,1)-cycle on S zp = B u - r_p,  S = B diag(F)^-1 Bt  (pressure Schur complement)
 *     zu = u - diag(F)^-1 Bt zp
 * Memory is O(nnz) (matrix blocks + an AMG hierarchy of ~1.3 nnz(S)), nothing grows like the fill of a factorisation.
 * Cylinder O1 BDF2 operator, sweeps = 3: ~20 GMRES iterations to 1e-10 from a zero guess, fewer inside time steps (which start
 * from the previous solution).  method: FC_METHOD_GMRES or FC_METHOD_BICGSTAB; max_iter / rtol / check_residual as in
 * fc_set_solver_options.  Needs fc_set_bc, fc_assemble_matrix(slot), fc_apply_bc(slot); single-GPU handles; after the
 * slot's matrix changed call it again (fc_update_operator alone keeps the old preconditioner for the new operator).
 * fc_setup_solver on the same slot later replaces the mode.  Batched stepping needs factors. */
int fc_setup_krylov(fc_handle h, int slot, int32_t sweeps, int method, int32_t max_iter, double rtol, int32_t check_residual);
/* info[8]: device bytes held for the slot's Krylov mode (permuted matrix + blocks + AMG hierarchy), velocity dofs, pressure dofs, AMG
 * levels (the dense coarsest one included), rows of the coarsest level, kernel launches per preconditioner apply, Jacobi sweeps, host
 * milliseconds of the setup; omega_out (optional): the Jacobi damping chosen from the spectral radius of diag(F)^-1 F */
int fc_get_krylov_info(fc_handle h, int slot, int64_t* info /* [8] */, double* omega_out);

/* ── state: FlowFieldCollection u_n, u_nn, p_n (flowfield.py:67-105; flowsolver.py:487-491) ─ */
int fc_set_state(fc_handle h, const double* u_n /* [2 nn] */, const double* u_nn /* [2 nn] */,
                 const double* p_n /* [nv] */);
/* Withdraw the last fc_step: (u_n, u_nn, p_n) as they were before it (the step's tail keeps what its shift overwrites).
 * The reference leaves its state untouched when a step fails (a non-finite velocity is detected BEFORE the fields are
 * shifted, flowsolver.py:727-751); a host program gets the same by calling this after FC_ERR_DIVERGED -- FlowSolver.step does.
 * Valid once after a single fc_step / fc_step_end (not after fc_run, fc_set_state or a batched step). */
int fc_undo_step(fc_handle h);
int fc_get_state(fc_handle h, double* u_n, double* u_nn, double* p_n);
int fc_get_solution(fc_handle h, double* up /* [N] last solve, W layout */);

/* ── THE per-step crossing: replaces assemblers[order].assemble(rhs); solvers[order].solve(...);
 *    split; _solver_diverged; field shift; make_measurement; compute_perturbation_energy
 *    (flowsolver.py:724-779).  order_slot is FC_SLOT_BDF1 or FC_SLOT_BDF2.
 *    y_out[n_sens], dE_out (1/2 |u|^2_L2, NaN if compute_energy == 0), info_out[4] =
 *    {iterations, relative residual of the first refinement residual, |b|, flags}. ------------- */
int fc_step(fc_handle h, int order_slot, const double* u_ctrl /* [n_act] */,
            const double* u_force /* [n_act] body-force amplitudes, NULL = u_ctrl (CN passes the
                                     mean of the new and the previous control, nsforms.py:224-226) */,
            double* y_out, double* dE_out, int compute_energy, double* info_out);
/* the same step in two halves: fc_step_begin writes the controls and enqueues the launches (the GPU works from here),
 * fc_step_end waits for the record -- the caller's own per-step bookkeeping fits in between (flowsolver.py:775-799: the
 * reference's exporter.log / progress / checkpoint work of the previous step).  fc_step = begin + end. */
int fc_step_begin(fc_handle h, int order_slot, const double* u_ctrl, const double* u_force, int compute_energy);
int fc_step_end(fc_handle h, double* y_out, double* dE_out, double* info_out);
/* What the controller of a closed loop waits for is y: on a single-GPU handle with the direct factor apply the step publishes the sensors
 * and the non-finite flag right behind the last sweep launch and computes residual monitor and energy on a second stream, overlapped with
 * the host's work and the next step's sweeps.  fc_step_end with dE_out == NULL and info_out == NULL returns as soon as y (and the flag:
 * FC_ERR_DIVERGED) is there; fc_step_collect hands over (dE, info) of that step later -- it blocks until they exist; the next
 * fc_step_begin keeps them if nobody asked.  With non-NULL dE_out / info_out fc_step_end (and fc_step) wait for everything, as before.
 * The reference's loop wants exactly this order: y_meas back to the controller at once, dE into the log (flowsolver.py:760-799). */
int fc_step_collect(fc_handle h, double* dE_out, double* info_out);
/* n_steps open-loop steps without host synchronisation in between (u_ctrl constant or a
 * sequence [n_steps][n_act]); y_seq [n_steps][n_sens], dE_seq [n_steps] (may be NULL). */
int fc_run(fc_handle h, int first_order_slot, int32_t n_steps, const double* u_ctrl,
           int u_ctrl_is_sequence, double* y_seq, double* dE_seq, int compute_energy);

/* ── base-flow (steady-state) iterations: replace SteadyStateSolver.picard / .newton (steadystate.py:60-159:
 *    dolfin.solve(F == 0, ...) :95 and the assemble / bc.apply / LUSolver.solve loop :137-145).  A host program drives the
 *    loop and its stopping rule (picard: relative change < tol, steadystate.py:150-156; newton: dolfin's residual criterion
 *    rel 1e-9 / abs 1e-10); every iteration's assembly, Dirichlet elimination, factorisation and solve run on the device.
 *    fc_set_baseflow_bc: the FULL-field Dirichlet data (FlowSolver._make_BCs, flowsolver.py:329-337) — it replaces the
 *    fc_set_bc tables (call fc_set_bc again before time stepping).  up: mixed vector [N], W layout, in = iterate, out = next
 *    iterate; load: (f, v) of a body force or NULL; nu = 1 / Re.  Enclosed flows: fc_set_pressure_pin first.
 *    fc_newton_step returns |F(up_in)| over the free rows in res_norm; update = 0 evaluates the residual only. */
int fc_set_baseflow_bc(fc_handle h, int32_t n_bc, const int32_t* bc_dofs, const double* bc_values);
int fc_picard_step(fc_handle h, double nu, double* up, const double* load, double* rel_change);
int fc_newton_step(fc_handle h, double nu, double* up, const double* load, double* res_norm, int update);

/* ── shared-operator batched stepping: k <= 32 lock-step simulations on ONE handle ────────────
 *    Replaces k independent FlowSolver instances that step the SAME operator with different initial
 *    conditions / controls / controllers — the reference's outer workloads: IC sweeps
 *    (examples/lidcavity/batch_run_lidcavity.py:197-215), controller optimisation (utils/optim.py:95-102),
 *    each of which runs FlowSolver.step (flowsolver.py:703-799) once per simulation and time step.
 *    All k simulations share the handle's operators, factors, BC / force / sensor tables and time scheme and are
 *    advanced together: every vector is a matrix [row][KB] on the device (KB = 4, 8, 16 or 32 >= k), every level
 *    of the factor sweep a dense block product on the fp64 matrix cores, so the factors are read once per
 *    step for all of them.  Needs fc_setup_solver (full factors, single GPU, no refinement sweeps).
 *    fc_set_batch(h, k) allocates the batched state (all zero; k = 0 frees it).  Host arrays are [k][...]
 *    (simulation-major).  The single-simulation state of the handle is independent of the batched one. */
/*    (fc_set_solver_options' check_residual = n applies to the batched steps too: the residual monitor runs on every n-th batched step of
 *    the handle, info_out[s][1], [2] are NaN in between; the non-finite test runs on every step.) */
int fc_set_batch(fc_handle h, int32_t k);
int fc_set_state_batch(fc_handle h, int32_t k, const double* u_n /* [k][2 nn] */, const double* u_nn /* [k][2 nn] */,
                       const double* p_n /* [k][nv] or NULL */);
int fc_get_state_batch(fc_handle h, int32_t k, double* u_n, double* u_nn, double* p_n /* any may be NULL */);
int fc_get_solution_batch(fc_handle h, int32_t k, double* up /* [k][N] last solve, W layout */);
/* fc_step for k simulations: u_ctrl [k][n_act], u_force [k][n_act] or NULL (= u_ctrl), y_out [k][n_sens], dE_out [k],
 * info_out [k][4] = {0, relative residual, |b|, flags}.  FC_ERR_DIVERGED when any simulation produced a non-finite
 * velocity (info_out[s][3] marks which; the others are unaffected: the columns are independent). */
int fc_step_batch(fc_handle h, int order_slot, int32_t k, const double* u_ctrl, const double* u_force, double* y_out,
                  double* dE_out, int compute_energy, double* info_out);
/* (the flags are per step: after FC_ERR_DIVERGED the other simulations simply go on; fc_reset_sim_batch(h, s) takes a diverged
 * run out of the dynamics -- its state becomes zero, so that its column stops producing non-finite values -- the host then ignores
 * its outputs.  The reference's per-run equivalent: FlowSolver.step returns None / raises and that run ends, flowsolver.py:727-737.) */
int fc_reset_sim_batch(fc_handle h, int32_t s);
int fc_step_batch_begin(fc_handle h, int order_slot, int32_t k, const double* u_ctrl, const double* u_force, int compute_energy);
int fc_step_batch_end(fc_handle h, int32_t k, double* y_out, double* dE_out, double* info_out);
/* fc_step_end(early) / fc_step_collect for k simulations (cache-resident factors: residual monitor and energy of a batched step run on a
 * second stream while the host and the next step go on): the early end hands over y [k][n_sens] and flags [k] (1: that simulation's
 * velocity became non-finite; FC_ERR_DIVERGED if any) as soon as the solve is done; fc_step_batch_collect returns (dE [k], info [k][4]) of
 * that step and blocks until they exist.  fc_step_batch_end with dE_out / info_out, and fc_step_batch, wait for everything as before. */
int fc_step_batch_end_early(fc_handle h, int32_t k, double* y_out, int32_t* flags_out);
int fc_step_batch_collect(fc_handle h, int32_t k, double* dE_out, double* info_out);
/* parity hook: X = A_bc^{-1} B for k right-hand sides through the batched factor apply; b, x: [k][N] */
int fc_solve_batch(fc_handle h, int slot, int32_t k, const double* b, double* x);

/* ── parity hooks (tests) ------------------------------------------------------------------- */
/* RHS of `order_slot` for the current state and u_ctrl, in W layout, BCs lifted and imposed:
 * what SystemAssembler.assemble(rhs) returns (flowsolver.py:728) */
int fc_assemble_rhs(fc_handle h, int order_slot, const double* u_ctrl, double* b_out /* [N] */);
/* x = A_bc^{-1} b with the device solver of `slot` */
int fc_solve(fc_handle h, int slot, const double* b /* [N] */, double* x /* [N] */,
             double* info_out /* [4] */);
/* 1/2 u^T M u for a host velocity field (flowsolver.py:827-829); needs FC_SLOT_MASS assembled */
int fc_energy(fc_handle h, const double* u /* [2 nn] */, double* E);
int fc_measure(fc_handle h, const double* up /* [N] */, double* y /* [n_sens] */);

/* ── multi-GPU (one process per GPU): replaces dolfin's MPI mesh partitioning
 *    (flowsolver.py:236-238) and PETSc/MUMPS' internal MPI.  Each rank holds the whole (small)
 *    discretisation but assembles only its cells and sweeps only its sub-tree of the elimination
 *    tree; rowkind[N] (W numbering): 0 = other rank's dof, 1 = owned, 2 = root separator (replicated).
 *    Exchange steps per step (all-reduces over RCCL/xGMI, or fc_set_host_exchange): the root right-hand side and the
 *    root solution (every rank applies its block of the root's rows) inside the solve, and the 80-double step
 *    tail (sensor partials, energy, residual norms, divergence flag). */
int fc_comm_unique_id(char* out128 /* ncclUniqueId bytes, made on rank 0 and broadcast by the host */);
int fc_comm_init(fc_handle h, int nranks, int rank, const char* id128);
/* fc_comm_probe: can this rank load RCCL at all?  Every rank calls it BEFORE fc_comm_init and the ranks agree on the outcome over
 * their own process group: ncclCommInitRank is a collective, so a rank that cannot load RCCL must not leave the others waiting in it.
 * fc_comm_destroy: give the communicator back (mixed outcome of fc_comm_init: everybody then uses fc_set_host_exchange). */
int fc_comm_probe(int rank);
int fc_comm_destroy(fc_handle h);
/* Exchange through the host for a partitioned handle that has NO RCCL communicator (several ranks on one GPU,
 * CPU-only collectives such as gloo): `fn(buf, n, user)` must sum the n doubles of `buf` over the ranks, in place.
 * The launch sequence of a step is the one of the RCCL path; only the exchange itself differs (device -> pinned
 * host buffer -> fn -> device instead of an in-stream ncclAllReduce).  Three exchanges per step: the root
 * right-hand side, the root solution (row blocks), the 80-double step record. */
/* what the handle's exchange is: transport 0 = none, 1 = RCCL (nranks / rank READ BACK from the communicator with
 * ncclCommCount / ncclCommUserRank), 2 = host callback */
int fc_comm_info(fc_handle h, int32_t* nranks, int32_t* rank, int32_t* transport);
typedef void (*fc_exchange_fn)(double* buf, int64_t n, void* user);
int fc_set_host_exchange(fc_handle h, int nranks, int rank, fc_exchange_fn fn, void* user);
/* Pre-flight of the handle's exchange (call on EVERY rank after fc_comm_init / fc_set_host_exchange, before any setup): one
 * all-reduce of a known vector (entry i of rank r = (r + 1) * (i + 1), 256 doubles) through the very path the time steps use
 * (in-stream ncclAllReduce, or the host callback), checked entry by entry against nranks (nranks + 1) / 2 * (i + 1).
 * FC_ERR_HIP with the first wrong entry in fc_last_error() on a mismatch: a communicator that connects the wrong ranks, a
 * callback that does not sum, a second HIP runtime in the process.  max_err_out (optional): largest deviation seen.
 * Replaces nothing in the reference (MPI_Init's own checks, src/utils/mpi.py:22-37). */
int fc_comm_selftest(fc_handle h, double* max_err_out);


#ifdef __cplusplus
}
#endif
#endif /* FC_HIP_H */
