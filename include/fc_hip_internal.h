/*
 * fc_hip_internal.h — entry points of libfc_hip.so that are NOT the drop-in boundary (include/fc_hip.h is):
 *   - array-level setup for callers that bring their own symbolic analysis (flowcontrol_amd/ndsolver.py is the readable
 *     specification of the in-library one; tests compare the two table by table),
 *   - the symbolic phase on its own (fc_sym_*),
 *   - bench / profiling / debug hooks used by bench.py, scripts/ and the tests.
 * Same conventions as fc_hip.h (plain C, int status, opaque handle).  Nothing here is needed to drive a simulation.
 */
#ifndef FC_HIP_INTERNAL_H
#define FC_HIP_INTERNAL_H

#include "fc_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* time `reps` back-to-back launches of the CSR SpMV kernel with HIP events on the handle's
 * stream; returns the mean milliseconds per launch */
int fc_bench_spmv(fc_handle h, int slot, int reps, double* ms_per_launch);
/* ── solver setup: replaces LUSolver.set_operator(A) + the factorisation MUMPS performs at the
 *    first solve (flowsolver.py:697,729).  The host side (flowcontrol_amd/ndsolver.py) supplies
 *    the nested-dissection permutation and the level-wise selected-inverse factors. ---------- */
int fc_set_permutation(fc_handle h, const int32_t* perm /* [N] new -> old */);
int fc_solver_setup(fc_handle h, int slot, const int32_t* Ap_rowptr, const int32_t* Ap_col,
                    const double* Ap_val, /* permuted system matrix (N rows) */
                    int32_t n_stages, const int64_t* stage_begin /* [n_stages] first row in seg_ptr */,
                    const int32_t* stage_row0 /* [n_stages] first destination row */,
                    const int32_t* stage_nrows /* [n_stages] */,
                    const int32_t* stage_kind /* [n_stages] 0 = up (y += ..), 1 = down (x = ..), 2 = diagonal (x = dscale y) */,
                    const int64_t* seg_ptr /* [total_rows + 1] */, int64_t n_seg,
                    const int64_t* seg_val /* [n_seg] offset into vals */,
                    const int32_t* seg_col /* [n_seg] >=0: first buffer index; <0: -(offset into idx)-1 */,
                    const int32_t* seg_len /* [n_seg] */, int64_t n_idx, const int32_t* idx,
                    int64_t n_val, const double* vals,
                    /* multi-GPU: after stage `ar_stage` (-1: none) the rows [ar_row0, ar_row0+ar_n) of the
                     * work buffer are summed over the ranks (RCCL all-reduce) */
                    int32_t ar_stage, int32_t ar_row0, int32_t ar_n,
                    /* and the rows x[ar_row0 .. +ar_n) after stage `ar2_stage` (-1: none): the root's down stage, in which
                     * every rank fills its own block of the root's rows (the others are zeroed before the launch) */
                    int32_t ar2_stage);
/* Truncated factors (memory-lean preconditioner): stages of kind 2 stand for tree levels whose pivot blocks are NOT
 * stored; on their rows x = dscale * y (dscale [N], permuted numbering: a diagonal stand-in for the Schur complement).
 * Such a slot is a preconditioner only: fc_solve / fc_step need FC_METHOD_GMRES or FC_METHOD_BICGSTAB. */
int fc_set_stage_diag(fc_handle h, int slot, const double* dscale /* [N] */);
/* optional: hand the down-sweep stages to the LDS-tiled block kernel.  Every block is up to 32
 * consecutive rows of ONE tree node, whose rows all read the same operand
 * [ y[i0..i0+ni) | x[idx[idx_off..+nb)] ] and whose values lie row-major (stride ni+nb) at blk_val.
 * stage_* arrays have one entry per stage of fc_solver_setup (count 0 = keep the segment kernel). */
int fc_solver_set_blocks(fc_handle h, int slot, int32_t n_stages, const int64_t* stage_blk_begin,
                         const int32_t* stage_blk_count, const int32_t* stage_lpr, int64_t n_blk,
                         const int64_t* blk_val, const int32_t* blk_row0, const int32_t* blk_nrows,
                         const int32_t* blk_i0, const int32_t* blk_ni, const int32_t* blk_idx,
                         const int32_t* blk_nb, int64_t n_idx, int64_t n_val);
/* the symbolic phase on its own, no device involved: every table fc_setup_solver derives, by name, widened to int64
 * (tests compare them with flowcontrol_amd/ndsolver.py entry by entry) */
int fc_sym_build(int32_t nv, int32_t ne, int32_t nc, const double* coords, const int32_t* cells, const int32_t* cell_edges,
                 int32_t n_bc, const int32_t* bc_dofs, int32_t depth, int32_t merge, int32_t world, int32_t rank, int32_t truncate,
                 void** out);
int fc_sym_size(void* sym, const char* name, int64_t* n);
int fc_sym_get(void* sym, const char* name, int64_t* out);
int fc_sym_free(void* sym);
/* Numeric factorisation ON THE DEVICE (what `solver.set_operator(A)` costs in the reference,
 * flowsolver.py:697,812-814 -> PETSc/MUMPS numeric phase; also every Newton/Picard iteration of
 * steadystate.py:60-159).  fc_factor_plan uploads the symbolic side once per (tree, pattern): all
 * fronts of the elimination tree live in one row-major buffer; `nodes` has 7 int64 per tree node in
 * elimination order (level, front offset, front order nf, pivot order ni, offset of the node's
 * [D^-1 | -U] rows in the factor values or -1, parent row or -1, child slot), `level_ptr`/`a_ptr` give
 * node and matrix-entry ranges per level (deepest first), (a_src, a_dst) scatter the CSR values of a slot
 * into the fronts, ext_p[ext_off[g] ...] are the positions of child g's update block in its parent's
 * front, ap_src maps the permuted matrix of the residual monitor to CSR value indices.
 * fc_refactor(slot) then recomputes the factor values and the permuted matrix of `slot` from the
 * slot's current CSR values (after fc_assemble_matrix + fc_apply_bc): scatter, per level extend-add of
 * the children's Schur complements, then the in-place elimination of all fronts of the level together by blocked
 * Gauss-Jordan steps of 32 pivot columns (pivot block inverted in LDS with partial pivoting inside the block, panels,
 * trailing update on the fp64 matrix cores, v_mfma_f64_16x16x4_f64: csrc/fc_front.hip.h), exported straight into the
 * layout the sweeps read.  No vendor BLAS / LAPACK is involved.
 * The structure (fc_solver_setup / fc_solver_set_blocks) must have been uploaded before, with any
 * values.  ms_out (optional): device time of the numeric phase.  On a partitioned handle every rank factorises its own
 * sub-tree and the root (plan built with keep=); the root front is summed over the ranks once (exchange). */
int fc_factor_plan(fc_handle h, int32_t n_nodes, const int64_t* nodes, int32_t n_levels, const int64_t* level_ptr,
                   int64_t front_size, int64_t n_a, const int64_t* a_src, const int64_t* a_dst,
                   const int64_t* a_ptr, const int64_t* ext_off, int64_t n_ext, const int32_t* ext_p,
                   int64_t n_ap, const int64_t* ap_src, int32_t max_slots);
/* Multi-GPU layouts made outside fc_setup_solver: of the ROOT's pivot-block inverse (the last plan node; every rank
 * eliminates the whole root front) this handle stores only the pivot rows [first, first + count) (0-based inside the
 * root's block) -- the rows it applies in the root's down stage -- at the root's value offset, row `first` first.
 * first = -1: all rows (single GPU).  Call before fc_refactor; fc_setup_solver does it itself. */
int fc_set_root_rows(fc_handle h, int32_t first, int32_t count);
/* values added to front entries (offsets into the front buffer of fc_factor_plan) after the matrix has
 * been scattered, in every later fc_refactor: a positive shift on ONE pressure diagonal selects the
 * solution with that pressure = 0 of an enclosed flow's singular system (lid-driven cavity; the reference
 * leaves that system to MUMPS, examples/lidcavity/lidcavityflowsolver.py:57-72).  n = 0 clears. */
int fc_set_front_shifts(fc_handle h, int32_t n, const int64_t* slots, const double* values);
/* download the factor values of a slot (n = the n_val given to fc_solver_setup): parity checks */
int fc_get_factor_values(fc_handle h, int slot, int64_t n, double* out);
/* velocity mass matrix (u,v) in the solver's permuted numbering, CSR with N rows (pressure rows
 * empty): the matrix behind compute_perturbation_energy (flowsolver.py:827-829), used by the
 * fused step tail */
int fc_set_energy_matrix(fc_handle h, const int32_t* rowptr, const int32_t* col, const double* val);
/* info[8]: k, KB, scratch rows of the up-sweep, block launches and fold launches per apply, factor bytes of one
 * batched apply (they serve KB simulated steps), vector (operand / result / fold) bytes of one batched apply, tasks */
int fc_get_batch_info(fc_handle h, double* info /* [8] */);
/* HIP-event timing of `reps` back-to-back batched factor applies; mean milliseconds per apply */
int fc_bench_batch_apply(fc_handle h, int slot, int reps, double* ms_per_apply);
/* time `reps` back-to-back factor applies (all sweep launches of one M^-1 application) with HIP
 * events on the handle's stream; mean milliseconds per apply and launches per apply */
int fc_bench_sweeps(fc_handle h, int slot, int reps, double* ms_per_apply, int32_t* launches_per_apply);
/* HIP-event timing inside fc_step / fc_run: when on, the back-to-back factor-sweep launches of
 * every apply are bracketed by ONE event pair on the handle's stream (sweep_ms / sweep_launches =
 * mean launch duration including the inter-launch gap) and every in-step CSR SpMV launch by its own
 * pair; totals are accumulated after the step's synchronisation.  fc_set_timing resets them. */
int fc_set_timing(fc_handle h, int on);
int fc_get_timing(fc_handle h, double* sweep_ms, int64_t* sweep_launches, double* spmv_ms,
                  int64_t* spmv_launches);
/* algorithmic bytes of one factor apply (sum over sweep launches) and of one CSR SpMV */
int fc_algorithmic_bytes(fc_handle h, int slot, double* sweep_bytes, double* spmv_bytes);

/* shape of the handle's elimination tree: bits_out[<= 16] = bisections fused per tree level, root first (fc_setup_solver's tree, or the
 * default shape of this mesh before it ran); nnz_min_tree (optional, one host symbolic pass): factor values of the all-binary-pairs tree
 * [2, 2, ...] of the same depth -- the fixed denominator of bench.py's roofline.frac_min_tree */
int fc_get_tree_info(fc_handle h, int32_t* bits_out /* [16] */, int32_t* n_bits, int64_t* nnz_min_tree);

/* the multi-GPU partition as arrays (fc_setup_solver derives and applies it itself on a handle with an exchange): this rank's
 * cells and rowkind[N] (W numbering): 0 = other rank's dof, 1 = owned, 2 = root separator (replicated) */
int fc_set_partition(fc_handle h, int32_t n_local_cells, const int32_t* local_cells,
                     const uint8_t* rowkind /* [N] */, int lead);

/* trailing-update flops of the handle's last fc_refactor (each level priced by its widest front): as run, and what they would be if
 * every row of a multi-GPU root front were swept on every rank (the scheme up to round 3; FC_ROOT_SKIP=0 runs it): the eliminated
 * rows of the root front that a rank does not export are dead and skipped -- (1 + 1 / world) n^3 instead of 2 n^3 per rank */
int fc_get_refactor_flops(fc_handle h, double* run, double* full);

/* Per-phase HIP-event timing of fc_step on the handle's stream (an instrumented replay: the marks cost ~1-2 us each and the
 * host polls less eagerly, so use it for the SPLIT of a step, not for its total).  When on, every fc_step records event marks at
 * its phase boundaries; fc_get_phase_timing returns the accumulated microseconds per phase and the number of steps since the
 * last fc_set_phase_timing.  Phases: 0 rhs (element loop + gather), 1 up-sweeps (local, to the first exchange), 2 exchange 1
 * (root right-hand side), 3 root stage, 4 exchange 2 (root solution), 5 down-sweeps, 6 tail kernels (residual rows, shift,
 * energy, sensors), 7 exchange 3 (80-double record), 8 publish.  On a single-GPU handle phases 2, 4, 7 stay zero and the whole
 * apply is reported under 1 (up to the root) and 5.  With a host exchange the exchange phases include the stream
 * synchronisation, the callback and the copies. */
int fc_set_phase_timing(fc_handle h, int on);
int fc_get_phase_timing(fc_handle h, double* us /* [9] */, int64_t* steps);

#ifdef __cplusplus
}
#endif
#endif /* FC_HIP_INTERNAL_H */
