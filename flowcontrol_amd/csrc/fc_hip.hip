// libfc_hip.so — host side of the C ABI declared in include/fc_hip.h.
//
// One fc_ctx per solver: owns the device copy of the discretisation (SoA cell tables, CSR pattern,
// inverted scatter indices), the matrix slots, the per-order solver data (permuted system matrix +
// nested-dissection selected-inverse factors) and all work vectors.  All work is enqueued on one
// HIP stream per handle; the only host synchronisation in the per-step path is the final copy of
// (y, dE, info) to pinned memory.
#include "../../include/fc_hip_internal.h"

#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <limits>
#include <map>
#include <string>
#include <vector>

#include "fc_kernels.hip.h"
#include "fc_front.hip.h"
#include "fc_batch.hip.h"
#include "fc_precond.hip.h"
#include "fc_symbolic.hpp"
#include "fc_precond.hpp"

namespace {

thread_local std::string g_err;

int fail(int code, const std::string& msg) {
  g_err = msg;
  return code;
}

#define HIPCHK(expr)                                                                         \
  do {                                                                                       \
    hipError_t _e = (expr);                                                                  \
    if (_e != hipSuccess)                                                                    \
      return fail(FC_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e) + " (" + __FILE__ + \
                                  ":" + std::to_string(__LINE__) + ")");                     \
  } while (0)

#define FCCHK(expr)          \
  do {                       \
    int _s = (expr);         \
    if (_s != FC_OK) return _s; \
  } while (0)

template <typename T>
struct DevBuf {
  T* p = nullptr;
  size_t n = 0;
  DevBuf() = default;
  DevBuf(const DevBuf&) = delete;
  DevBuf& operator=(const DevBuf&) = delete;
  DevBuf(DevBuf&& o) noexcept : p(o.p), n(o.n) { o.p = nullptr, o.n = 0; }
  DevBuf& operator=(DevBuf&& o) noexcept {
    if (this != &o) {
      release();
      p = o.p, n = o.n;
      o.p = nullptr, o.n = 0;
    }
    return *this;
  }
  int alloc(size_t count) {
    release();
    n = count;
    if (count == 0) return FC_OK;
    HIPCHK(hipMalloc((void**)&p, count * sizeof(T)));
    return FC_OK;
  }
  int upload(const T* src, size_t count, hipStream_t s) {
    if (count != n || (!p && count)) FCCHK(alloc(count));
    if (count) HIPCHK(hipMemcpyAsync(p, src, count * sizeof(T), hipMemcpyHostToDevice, s));
    return FC_OK;
  }
  int upload(const std::vector<T>& v, hipStream_t s) { return upload(v.data(), v.size(), s); }
  int zero(hipStream_t s) {
    if (n) HIPCHK(hipMemsetAsync(p, 0, n * sizeof(T), s));
    return FC_OK;
  }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    n = 0;
  }
  ~DevBuf() { release(); }
};

struct Stage {
  int64_t rp_begin;  // offset of the stage's first row in seg_ptr
  int row0;          // first destination row (permuted numbering)
  int nrows;
  int kind;  // 0 up, 1 down, 2 diagonal (truncated factors: x = dscale * y on the rows whose pivot blocks are not kept)
  int lanes, sub;  // lanes per row, lanes per segment slot
  int64_t blk_begin = 0;  // down stages: LDS-tiled block kernel (fc_nd_down_block) when blk_count > 0
  int blk_count = 0, blk_lpr = 64, blk_rps = 1;
  int blk_flat = 0;  // > 0: small nodes -- the stage runs fc_nd_flat_block with this many loads per thread (tiles of <= 256 x blk_flat values)
  double bytes;  // algorithmic bytes of this launch
  int64_t wg_begin = -1;  // offset of this launch's workgroup order (OrderSys::wg_order), -1: launch order = row order
  bool nt = false;  // its values are streamed with nontemporal loads (OrderSys::nt, minus the stages kept cache-resident)
};

// Device copy of the factorisation-free preconditioner of a slot (fc_setup_krylov; fc_precond.hpp / fc_precond.hip.h): the blocks of the
// saddle-point operator in compact velocity / pressure numberings and the AMG hierarchy of the pressure Schur complement.
struct PcMat {
  DevBuf<int> rp, ci;
  DevBuf<double> v;
  int nrows = 0, lanes = 8;
  int64_t nnz = 0;
  int upload(const fcpc::Csr& M, hipStream_t s) {
    nrows = M.nrows;
    nnz = M.nnz();
    const double mean = nrows ? (double)nnz / nrows : 0.0;
    lanes = mean <= 6 ? 4 : (mean <= 40 ? 8 : (mean <= 96 ? 16 : 32));
    FCCHK(rp.upload(M.rp, s));
    if (nnz == 0) {
      FCCHK(ci.alloc(1));
      FCCHK(v.alloc(1));
      return FC_OK;
    }
    FCCHK(ci.upload(M.ci, s));
    FCCHK(v.upload(M.v, s));
    return FC_OK;
  }
  int64_t bytes() const { return 12 * nnz + 4 * ((int64_t)nrows + 1); }
};
struct PcLevel {
  PcMat G, U;            // folded transfer operators of one AMG level (fc_precond.hpp: fold_down / fold_up)
  DevBuf<double> cat;    // [r (n) | z_c (n_next)]: the level's right-hand side next to the coarse correction U reads with it
  int n = 0, n_next = 0;
};
struct Precond {
  bool ready = false;
  int sweeps = 2;       // damped-Jacobi sweeps on the velocity block (the first two are one folded product)
  double omega = 1.0;   // their damping
  int nu = 0, np = 0;
  DevBuf<int> vpos, ppos;  // compact velocity / pressure index -> position in the permuted Krylov vector
  PcMat KF;                // sweeps >= 2: Wd (2 I - F Wd), columns = positions in the Krylov vector
  PcMat F, B, Bt;          // compact numbering (F only for sweeps >= 3)
  DevBuf<double> dinvF, wdinvF;  // 1 / diag(F), omega / diag(F)
  DevBuf<double> u0, u1, zp;     // velocity ping-pong, pressure correction
  std::vector<PcLevel> lv;
  int n_coarse = 0;
  DevBuf<double> cinv, rc;       // dense inverse of the coarsest operator, its right-hand side
  int64_t bytes = 0;             // device bytes held by the preconditioner
  int launches = 0;              // kernel launches per apply
  void release() {
    ready = false;
    for (PcMat* M : {&KF, &F, &B, &Bt}) M->rp.release(), M->ci.release(), M->v.release(), M->nnz = 0;
    vpos.release(), ppos.release(), dinvF.release(), wdinvF.release(), u0.release(), u1.release(), zp.release();
    lv.clear();
    cinv.release(), rc.release();
    bytes = 0;
  }
  double setup_ms = 0.0;
  std::vector<int> level_rows;
};

struct OrderSys {
  bool have_lift = false, ready = false;
  bool factor_free = false;  // no factors: the slot's Krylov solves are preconditioned by `pc` (fc_setup_krylov)
  Precond pc;
  DevBuf<int64_t> ap_src;    // factor-free slots: permuted CSR entry -> entry of the handle's CSR (fc_update_operator)
  bool structured = false;  // segment lists / stage table uploaded (fc_solver_setup); values may be stale
  DevBuf<double> lift;    // [n_act][N] original numbering
  DevBuf<double> lift_p;  // [n_act][N] permuted
  DevBuf<int> Ap_rowptr, Ap_col;
  DevBuf<double> Ap_val;
  int64_t Ap_nnz = 0;
  DevBuf<int64_t> seg_ptr;  // per-row segment lists of all stages, concatenated
  DevBuf<FcSeg> seg;
  DevBuf<FcBlk> blk;
  DevBuf<int> wg_order;  // per sweep launch: row groups by decreasing work
  DevBuf<int> f_idx;
  DevBuf<double> f_val;
  DevBuf<float> f_val32;    // compressed factors (fc_set_factor_precision): the values rounded once to fp32 ...
  DevBuf<FcBf16> f_val16;   // ... or to bfloat16; f_val is then not allocated
  int bits = 64;            // storage width of this slot's factor values
  int64_t f_nnz = 0;
  DevBuf<double> dscale;  // kind-2 stages (fc_set_stage_diag), permuted numbering
  bool truncated = false;  // the factors are a preconditioner only (some pivot blocks replaced by a diagonal)
  bool nt = false;         // fp64 factors larger than the Infinity Cache: the sweeps stream them with nontemporal loads
  bool inexact = false;    // full fp64 factors whose direct apply missed the acceptance residual (an ill-conditioned operator): they precondition GMRES
  std::vector<Stage> stages;
  double sweep_bytes = 0.0;
  int ar_stage = -1, ar_row0 = 0, ar_n = 0;  // all-reduce buf[ar_row0 .. +ar_n) after this stage
  int ar2_stage = -1;  // the root's down stage (this rank's block of rows): x[ar_row0 .. +ar_n) is zeroed before, summed after
  // optional explicit operator of the rhs (Crank-Nicolson): rows permuted, columns index u_n (W layout)
  DevBuf<int> c_rowptr, c_col;
  DevBuf<double> c_val;
  bool have_c = false;
};

constexpr int kPinDoubles = 8192;  // the host-mapped page: 32 simulation records of kRecStride doubles, the sequence slots and the late records behind them
constexpr int FC_N_PHASES = 9;  // fc_get_phase_timing
enum { PH_RHS = 0, PH_UP, PH_X1, PH_ROOT, PH_X2, PH_DOWN, PH_TAIL, PH_X3, PH_PUBLISH };
#ifndef FC_DOWN_DEPTH
#define FC_DOWN_DEPTH 2  // down-sweep rows: lanes ~ mean segment length / this
#define FC_RESIDENT_BYTES 0.0  // ... except the first stages up to this many bytes (experiment: kept in the Infinity Cache)
#define FC_NT_BYTES 268435456.0  // fp64 factors beyond this (the Infinity Cache) are streamed with nontemporal loads
#define FC_UP_THREADS 1048576.0  // up-sweep rows: segments of a row run side by side only while rows x lanes stays below this
#endif

}  // namespace

struct fc_ctx {
  int device = 0;
  int n_cu = 256;  // compute units of the device (persistent launches are sized by it)
  hipStream_t stream = nullptr;
  int nv = 0, ne = 0, nc = 0, nn = 0, N = 0;
  int64_t nnz = 0;
  // discretisation
  DevBuf<int> cn;       // [6][nc]
  DevBuf<double> geom;  // [5][nc]
  std::vector<int> h_rowptr, h_col;
  DevBuf<int> rowptr, col;
  DevBuf<int> mptr, midx;  // matrix scatter (per CSR slot)
  std::vector<int> h_gptr, h_gidx;  // vector scatter per W row (original numbering)
  std::vector<int> h_cell_dofs;     // [nc][15] W dofs of every cell (fc_setup_solver: elimination tree)
  std::vector<double> h_cent;       // [nc][2] cell centroids
  // symbolic phase done inside the library (fc_setup_solver): kept for the second slot and for the getters
  bool sym_ready = false;
  int sym_truncate = 0;
  std::vector<int> sym_bits;  // bisections fused per level of sym_tree, root first
  fcsym::Tree sym_tree;
  fcsym::Factors sym_fac;
  fcsym::Plan sym_plan;
  int64_t sym_total_nnz = 0;  // factor values of the WHOLE tree (all ranks, no truncation)
  double refactor_ms[2] = {0.0, 0.0};  // device time of the last fc_refactor per slot
  double refactor_flops = 0.0, refactor_flops_full = 0.0;  // trailing-update flops of the last fc_refactor: as run / if no dead rows were skipped
  bool step_pending = false;  // fc_step_begin without its fc_step_end
  int pend_slot = 0, pend_energy = 0;
  bool pend_checked = true;
  double pend_seq = 0.0;
  int factor_bits = 64;       // fc_set_factor_precision: storage width of the factor values laid out from now on (64 exact, 32 / 16 compressed)
  int pin_dof = -1;           // fc_set_pressure_pin: pressure dof whose diagonal is shifted inside the factorisation
  double pin_shift = 1.0;
  int64_t sym_local_values[2] = {0, 0};  // factor values this rank sweeps per solve, per slot
  std::vector<int> h_s_rowptr, h_s_idx;  // sensors as given by the caller (restricted to owned dofs on a partitioned handle)
  std::vector<double> h_s_w;
  DevBuf<int> gptr_p, gidx_p;       // permuted row order
  DevBuf<double> em, ev;
  DevBuf<double> vals[FC_NUM_SLOTS];
  bool slot_ok[FC_NUM_SLOTS] = {false, false, false, false};
  // BC / force / sensors
  int n_bc = 0, n_act = 0, n_sens = 0;
  std::vector<int> h_bc_dofs;
  std::vector<double> h_bcprof;
  DevBuf<unsigned char> isbc;
  DevBuf<int> bcslot_p;
  DevBuf<double> bcprof, fprof;
  bool have_force = false;
  DevBuf<double> fvec;   // [n_act][N] permuted: assembled load vector of every body-force profile (build_force_vectors)
  bool fvec_ok = false;  // ... valid for the present profiles, permutation, Dirichlet rows and partition
  DevBuf<int> s_rowptr, s_idx;
  DevBuf<int> s_idxp;        // sensor dofs as positions in the permuted solution (what fc_final reads: the state ring holds no W-layout copy)
  bool have_sidxp = false;
  DevBuf<unsigned> fin_cnt;  // arrival counters of the fused final (self-resetting)
  DevBuf<double> s_w;
  // time scheme
  double dt = 0.0;
  int nonlinear = 1;
  // permutation
  bool have_perm = false;
  std::vector<int> h_perm;
  DevBuf<int> perm, iperm;  // permuted row -> W dof, and its inverse
  OrderSys sys[2];
  DevBuf<int> mp_rowptr, mp_col;  // velocity mass matrix in permuted numbering (energy)
  DevBuf<double> mp_val;
  bool have_mp = false;
  // solver options
  int method = FC_METHOD_REFINE, max_iter = 1, check_residual = 1;
  bool floor_ok = false;  // GMRES may return at a stagnated true residual (set while inexact factors precondition it: KrylovOverride)
  DevBuf<double> kry;  // Krylov work vectors (BiCGStab: 8 N; GMRES(m): (m + 4) N), allocated on first use
  DevBuf<double> ks;   // device-resident scalars of the Krylov recurrences (KS_* in fc_kernels.hip.h)
  DevBuf<double> gm, mdot;  // GMRES: Hessenberg / rotations / small vectors; multi-dot partials
  int gmres_m = 30;    // restart length
  // Arnoldi steps of the factorisation-free GMRES as HIP graphs (each is ~30 launch-bound launches: preconditioner apply, mat-vec,
  // two Gram-Schmidt passes, rotation): one graph per (slot, basis column j), captured on first use; FC_KRYLOV_GRAPH=0: plain launches
  struct KryGraphs {
    std::vector<hipGraphExec_t> step;
    uint64_t sig = 0;
  } kgraph[2];
  bool pc_warm_start = true;  // factor-free slots: GMRES inside a time step starts from the previous solution (FC_PC_WARM_START=0: from zero)
  int last_krylov_iters = 0;
  double rtol = 1e-10;
  // state + work
  // THE STATE RING: four work buffers [y | x] of 2 N doubles in the solver's permuted numbering.  A solve writes its solution into the
  // x half of the buffer it works in; for a time step that half IS the new (v, p): the step then just moves `cur` on, nothing is
  // copied, scattered or shifted.  Slot cur: (u_n, p_n); cur - 1: u_nn; cur - 2: what fc_undo_step restores; cur + 1 (= h->buf): the
  // work buffer of the next solve.  Host-facing vectors (fc_set_state / fc_get_state / fc_get_solution) are in the W layout and are
  // permuted on the way in and out.
  DevBuf<double> ring;
  int cur = 0;
  bool state_live = false;                     // the ring holds a state (fc_set_state with a permutation, or a step)
  std::vector<double> hs_n, hs_nn;             // W-layout state handed over before any permutation existed (N doubles each: [u | p])
  DevBuf<int> cnp;                             // [12][nc] permuted position of the x- / y-velocity dof of every cell node
  DevBuf<unsigned char> velrow_p;              // 1 on the velocity rows (permuted numbering): the non-finite test
  DevBuf<int> asm_cells;  // multi-GPU: cells whose element matrices this rank assembles (own + touching a root dof)
  int n_asm = 0;
  // OVERLAPPED TAIL (single GPU, direct factor apply): the host waits only for the sensors + non-finite flag, published right behind
  // the last sweep launch (fc_early; the down-sweep launches test finiteness as they write).  Residual monitor, energy and the next
  // step's element loop run on a second stream while the host and the next step's sweeps go on; their results (dE, |r|, |b|) land in
  // a late record that fc_step_end collects only if the caller asks for them, else fc_step_collect / the next step does.
  hipStream_t stream2 = nullptr;
  DevBuf<fc_u64> solved;    // sequence number of the last step whose solve has finished (fc_early -> fc_wait_solved)
  DevBuf<int> side_err;     // fc_wait_solved gave up (rewritten by every gate; fc_final_late publishes it inside the late record)
  long gate_spin = 2000000L;  // polls of the side stream's gate before it gives up (~50 ms; FC_GATE_SPIN: test aid, 0 = give up at once)
  bool side_busy = false;   // stream2 has work that has not been synchronised with
  bool overlap = true;      // FC_OVERLAP_TAIL=0: everything on one stream, one record
  bool sweep_check = false; // the down-sweep launches of the apply being enqueued test the solution for finiteness
  // late records, by step parity: two steps' tails may be in flight (the host makes sure that of step n - 2 has arrived before it enqueues step n,
  // which is what frees b(n - 2) and, two steps later, the ring slot)
  struct Late {
    bool pending = false;
    double seq = 0.0;
    int energy = 0;
    bool checked = false;
  } late[2];
  int last_par = 0, pend_par = 0;  // parity of the last step that ended / of the step in flight
  bool pend_overlapped = false;
  bool want_all = false;    // fc_step with dE_out / info_out: the caller waits for everything anyway -- one stream, one record
  double last_dE = 0.0, last_info[4] = {0.0, 0.0, 0.0, 0.0};
  bool undo_ok = false;
  uint64_t step_count = 0;   // steps enqueued on this handle (the residual monitor's cadence counts them)
  bool last_checked = true;  // the last enqueued step formed its residual
  struct BufView {
    double* p = nullptr;
    size_t n = 0;
  } buf;  // = ring slot cur + 1: [y | x] (2N)
  DevBuf<double> bstore;  // two right-hand sides: the late tail of step n reads b(n) while step n + 1 assembles b(n + 1)
  BufView b;
  DevBuf<double> xsol, tmpN, tmpN2;
  DevBuf<double> partial, scal;               // reductions; scal: [0]=E [1]=r2 [2]=b2
  DevBuf<double> uctrl, ydev, yseq, Eseq, useq;
  DevBuf<int> flag;
  DevBuf<int> flag2;  // the late tail's row workgroups test finiteness too: into a word nobody reads
  DevBuf<int> flag2x; // ... one per simulation (32) for the batched late tail
  double* pin = nullptr;    // pinned, device-mapped host record: [0..63] u_ctrl in, [64..] outputs
  double* pin_dev = nullptr;  // device address of the same memory
  uint64_t seq = 0;           // step sequence number published by the last kernel of a step
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  int nblk_N = 0;
  // multi-GPU partition (fc_set_partition / fc_comm_init): this rank's cells and rows
  bool partitioned = false, lead = true;
  int ncl = 0;
  std::vector<int> h_cell_list;
  std::vector<unsigned char> h_rowkind;  // original numbering: 0 other rank, 1 owned, 2 shared root
  DevBuf<int> cell_list;
  DevBuf<unsigned char> rowkind_p;  // permuted numbering
  DevBuf<unsigned char> rowkind_w;  // W numbering (columns of an explicit rhs operator; row mask of the collective fc_spmv)
  DevBuf<unsigned char> rootmask_p;  // 1 on the root's rows (permuted numbering): row mask of the partitioned Krylov mat-vec
  void* comm = nullptr;             // ncclComm_t
  int nranks = 1, rank = 0;
  // exchange staged through the host when the ranks have no RCCL communicator (CPU collectives, ranks sharing a GPU)
  fc_exchange_fn host_xchg = nullptr;
  void* host_xchg_user = nullptr;
  double* xstage = nullptr;  // pinned
  size_t xstage_n = 0;
  DevBuf<double> tail;  // [y(64) | E | r2 | b2 | .. | flag@72] partial sums of a step, all-reduced
  // per-launch HIP-event timing (fc_set_timing): pairs recorded around every sweep / SpMV launch
  bool timing = false;
  std::vector<hipEvent_t> tev;   // pool, 2 per launch
  // element vectors of the NEXT step's right-hand side, enqueued behind a synchronous step while the host
  // is busy (they depend on the state only): slot whose coefficients they were computed with, or -1
  int pre_slot = -1;
  // column form of the up-sweep (full fp64 factors, in-library symbolic phase; single-GPU and partitioned handles): per tree level one LDS-tiled block launch over
  // the nodes' dense -L blocks (fc_nd_down_block with `out`: the node's y rows staged once per workgroup, whole rows streamed) + one fold
  // launch (fc_nd_fold1) instead of one segment-row launch.  Two launches per level instead of one, but the blocks stream at the down-sweep's
  // rate (~5 TB/s) where the segment rows reach 2.7-3.9: it pays where the factors stream from HBM (refined cylinder + 3 %, pinball + 3 %,
  // cavity_fine + 4.5 % steps/s) and costs 6 % on O1, whose launches sit on their floor.  up_form: 0 = auto (column form for slots with
  // OrderSys::nt), 1 = row form, 2 = column form (FC_UP_FORM=auto|row|column, read by fc_create)
  int up_form = 0;
  int elem_reg_min = 40000;  // FC_ELEM_REG_MIN (read by fc_create): meshes / cell lists of at least this many cells run the element loop on a thread per cell
  struct UpCol {
    bool ready = false, tried = false;
    DevBuf<FcBlk> blk;
    DevBuf<int> fptr, fsrc;
    DevBuf<double> scratch;
    struct Lvl {
      int64_t begin;
      int count, lpr, rps, fold_row0, fold_nrows;
      int flat = 0;  // as Stage::blk_flat
    };
    std::vector<Lvl> lv;  // deepest level first
  } upc;
  // device-side numeric factorisation (fc_factor_plan / fc_refactor)
  struct PlanNode {
    int64_t front, voff;
    int level, nf, ni, parent;
  };
  bool have_plan = false;
  bool huge_lds_ok = false;  // fc_fe_pivot_huge was granted its dynamic LDS
  bool huge_refused = false;  // ... or was refused it: no 128-column steps on this handle
  int root_x0 = -1, root_xn = 0;  // multi-GPU: of the root's pivot rows (0-based inside its block) this handle stores [root_x0, root_x0 + root_xn) only (fc_set_root_rows); -1: all
  std::vector<PlanNode> pnodes;
  std::vector<int64_t> plevel_ptr, pa_ptr;
  std::vector<std::vector<std::pair<int64_t, int>>> pext_groups;  // per (level, slot): (first FcExt, count)
  std::vector<std::vector<int>> pext_maxnb;
  int pmax_slots = 1, pmax_ni = 0;
  DevBuf<double> fronts;
  DevBuf<int64_t> pa_src, pa_dst, pap_src;
  DevBuf<FcExpItem> pexp;  // work list of fc_fe_export
  int64_t pexp_n = 0, pfront_total = 0;
  DevBuf<FcExt> pext;
  DevBuf<FcExt> pext2;                                 // the same descriptors grouped by parent (children in slot order): fc_extend_add_parents
  DevBuf<FcExtPar> pextpar;
  std::vector<std::pair<int64_t, int>> pextpar_groups;  // per level of the children: (first parent entry, parents)
  std::vector<int> pextpar_maxnf;                       // ... and the largest parent front
  DevBuf<FcFront> pfront;                              // fronts with a pivot block, grouped per level
  std::vector<std::pair<int64_t, int>> pfront_groups;  // per level: (first, count)
  std::vector<int> plevel_max_ni, plevel_max_nf;       // per level: block steps / tile grid of the elimination kernels
  std::vector<int64_t> plevel_fsize;                   // per level: doubles in its fronts with a pivot block
  DevBuf<double> pscratch;                             // per front: inverse of the current pivot block + copied column panel
  DevBuf<int> pext_p;
  DevBuf<int64_t> pshift_slot;  // fc_set_front_shifts: added to the fronts after the scatter
  DevBuf<double> pshift_val;
  int pn_shift = 0;
  // per-phase timing of fc_step (fc_set_phase_timing): event marks at the phase boundaries, folded after the step's synchronisation
  bool phase_timing = false;
  std::vector<hipEvent_t> pev;   // pool
  std::vector<int> pid;          // phase that ENDS at mark i (-1: start of a step)
  size_t pused = 0;
  double ph_us[FC_N_PHASES] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  int64_t ph_steps = 0;
  std::vector<int> tkind;        // 0 sweep, 1 spmv (per recorded pair)
  std::vector<int> tcount;       // kernel launches bracketed by the pair
  size_t tused = 0;
  double t_ms[2] = {0.0, 0.0};
  int64_t t_cnt[2] = {0, 0};
  // base-flow iterations inside the library (fc_set_baseflow_bc / fc_picard_step / fc_newton_step): full-field Dirichlet data
  std::vector<int> ss_dofs;
  std::vector<double> ss_vals;
  // batched stepping (fc_set_batch): k lock-step simulations that share the operators and the factors of this handle;
  // every vector is a matrix [row][KB] (fc_batch.hip.h)
  struct BLaunch {
    int kind;  // 0: fc_nd_block_b over tasks [first, first + count) with cg column-group waves per row tile; 1: fc_nd_fold_b
    int first, count, cg;
    int row0, nrows, dst_off, accumulate;
    double mean_chunks = 0.0;  // block launches: 32-column chunks per tile row, weighted by the values behind them (cg is chosen from it per batch width)
  };
  struct Batch {
    int k = 0, KB = 0;
    bool tables = false;
    // the batched state ring: four work buffers [y (N) | x (N) | scratch | zero row] of KB columns, permuted numbering; the x half a
    // batched step writes IS the new state of all KB simulations (slot cur: (u_n, p_n); cur - 1: u_nn; cur + 1 = buf: the next solve)
    DevBuf<double> ring;
    int cur = 0;
    size_t slot_doubles = 0;
    fc_ctx::BufView buf;
    DevBuf<double> bstore;        // two right-hand sides (overlapped tail: the late tail of step n reads b(n) while step n + 1 assembles its own)
    fc_ctx::BufView b;
    DevBuf<double> ev, partial;
    // overlapped tail of the batched step (as fc_ctx::stream2): late records per simulation and step parity
    struct Late {
      bool pending = false;
      double seq = 0.0;
      int energy = 0;
      bool checked = true;
    } late[2];
    int last_par = 0, pend_par = 0;
    bool pend_overlapped = false, side_busy = false;
    std::vector<double> last_dE, last_r, last_b;   // per simulation: what fc_step_batch_collect hands out
    hipGraphExec_t gside[4][4] = {};               // side-stream graph per (ring phase, energy flag + 2 x residual monitor)
    uint64_t gside_sig[4][4] = {};
    DevBuf<int> flag;                                        // [KB] non-finite velocity seen, per simulation
    DevBuf<FcBTask> tasks;
    DevBuf<double> part;          // split tiles (FcBTask::split): a partial slot of 16 x 32 doubles per part ...
    DevBuf<unsigned> ticket;      // ... and the arrival counter of every split tile (at its first slot; self-resetting)
    int64_t pslots = 0;
    DevBuf<int> trowd;            // tail row blocks: one int4 per row, in the cell order of build_tail_blocks (FcTBlock)
    DevBuf<int> fptr, fsrc;       // up-sweep fold lists: permuted row -> scratch rows (absolute buffer rows) of its descendants
    DevBuf<int> olist;            // per tree node: the buffer row of every operand column ([y rows of the node | x rows of its boundary])
    DevBuf<double> ftile[2];      // per slot: the factor values in the tiled layout the batched block kernel streams (fc_b_repack)
    bool ftile_ok[2] = {false, false};
    int64_t tiled_values = 0;
    DevBuf<FcTBlock> tblocks;     // row blocks of the batched tail (fc_tail_b): <= 16 consecutive permuted rows, <= tb_cols distinct columns
    int tb_cols = FC_TB_COLS;     // (FC_TB_COLS in the environment: tuning aid)
    bool tb_built = false;
    DevBuf<int> tcols;            // their distinct columns
    DevBuf<unsigned short> tlidx; // per matrix entry (order of the permuted CSR): position of its column in its block's list
    int n_tblocks = 0;
    std::vector<BLaunch> launches;
    int64_t scratch_rows = 0, factor_values = 0;
    double vec_rows = 0.0;        // operand / result rows moved per apply (x KB x 8 B)
    bool pending = false;
    int pend_slot = 0, pend_energy = 0;
    double pend_seq = 0.0;
    // the launches of one batched step as a HIP graph per (order slot, energy flag): captured on first use, replayed as long as
    // no buffer or parameter that a kernel argument was taken from has changed (gsig: hash of all of them)
    // (last index: 2 = the graph starts with the element loop, 1 = with the full gather (the previous step's graph ran the element loop:
    //  pre_slot), 0 = with the control rows only (it gathered the control-independent right-hand side too: pre_gather))
    // (first index: the ring phase `cur` the step starts from -- the buffers a step reads and writes rotate with period four)
    hipGraphExec_t gexec[4][2][4][3] = {};   // [ring phase][slot][energy + 2 x residual monitor][lead]
    uint64_t gsig[4][2][4][3] = {};
    int pre_slot = -1;  // order slot whose element vectors (ev) the LAST launch of the previous step left behind, -1: none
    uint64_t step_count = 0;   // batched steps enqueued (the residual monitor's cadence counts them: fc_set_solver_options check_residual = n)
    bool pend_checked = true;  // the step in flight forms its residual
    bool pre_gather = false;  // ... and whose control-independent right-hand side it has gathered as well -- into pre_b / pre_y
    const double* pre_b = nullptr;
    const double* pre_y = nullptr;
    DevBuf<int> ctrl_rows[2];  // per slot: the rows whose right-hand side depends on u_ctrl (Dirichlet rows + rows with a lifting entry)
    int n_ctrl_rows[2] = {0, 0};
    bool ctrl_ok[2] = {false, false};
  } bat;
};

extern "C" int collect_late(fc_ctx* h, int par);        // late records of overlapped steps (defined with the step functions)
extern "C" int collect_late_batch(fc_ctx* h, int par);


namespace {

// RCCL is resolved at run time (dlopen): single-GPU use has no dependency on it, and inside a
// torch process the copy torch already loaded is reused.
struct FcNcclId {
  char internal[128];
};
struct Rccl {
  void* lib = nullptr;
  int (*GetUniqueId)(FcNcclId*) = nullptr;
  int (*CommInitRank)(void**, int, FcNcclId, int) = nullptr;
  int (*AllReduce)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
  int (*CommDestroy)(void*) = nullptr;
  int (*CommCount)(void*, int*) = nullptr;
  int (*CommUserRank)(void*, int*) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
};
Rccl g_rccl;

int rccl_load() {
  if (const char* e = std::getenv("FC_RCCL_DISABLE"); e && e[0] == '1')  // test aid: a machine whose RCCL cannot be loaded
    return fail(FC_ERR_HIP, "cannot load RCCL: disabled by FC_RCCL_DISABLE");
  if (g_rccl.lib) return FC_OK;
  // first the copy that sits next to the HIP runtime this library is bound to (PyTorch ships its own set and
  // has it loaded already), then by soname, then the system one
  std::string dir;
  Dl_info di;
  if (dladdr((void*)&hipStreamSynchronize, &di) && di.dli_fname) {
    dir = di.dli_fname;
    const size_t cut = dir.rfind('/');
    dir = cut == std::string::npos ? std::string() : dir.substr(0, cut + 1);
  }
  const std::string names[] = {dir + "librccl.so", dir + "librccl.so.1", "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  void* lib = nullptr;
  for (const std::string& n : names) {
    lib = dlopen(n.c_str(), RTLD_NOW | RTLD_GLOBAL);
    if (lib) break;
  }
  if (!lib) return fail(FC_ERR_HIP, std::string("cannot load RCCL: ") + dlerror());
  g_rccl.GetUniqueId = (int (*)(FcNcclId*))dlsym(lib, "ncclGetUniqueId");
  g_rccl.CommInitRank = (int (*)(void**, int, FcNcclId, int))dlsym(lib, "ncclCommInitRank");
  g_rccl.AllReduce = (int (*)(const void*, void*, size_t, int, int, void*, hipStream_t))dlsym(lib, "ncclAllReduce");
  g_rccl.CommDestroy = (int (*)(void*))dlsym(lib, "ncclCommDestroy");
  g_rccl.GetErrorString = (const char* (*)(int))dlsym(lib, "ncclGetErrorString");
  g_rccl.CommCount = (int (*)(void*, int*))dlsym(lib, "ncclCommCount");
  g_rccl.CommUserRank = (int (*)(void*, int*))dlsym(lib, "ncclCommUserRank");
  if (!g_rccl.GetUniqueId || !g_rccl.CommInitRank || !g_rccl.AllReduce || !g_rccl.CommDestroy)
    return fail(FC_ERR_HIP, "RCCL symbols missing");
  g_rccl.lib = lib;
  return FC_OK;
}

#define NCCLCHK(expr)                                                                             \
  do {                                                                                            \
    int _r = (expr);                                                                              \
    if (_r != 0)                                                                                  \
      return fail(FC_ERR_HIP, std::string(#expr) + ": " +                                         \
                                  (g_rccl.GetErrorString ? g_rccl.GetErrorString(_r) : "rccl error")); \
  } while (0)

constexpr int kNcclDouble = 8, kNcclSum = 0;

void tabulate(double* phi2, double* dphi2, double* phi1, double* qw) {
  const double s15 = std::sqrt(15.0);
  const double a1 = (6.0 - s15) / 21.0, a2 = (6.0 + s15) / 21.0;
  const double w0 = 9.0 / 40.0, w1 = (155.0 - s15) / 1200.0, w2 = (155.0 + s15) / 1200.0;
  double lam[7][3] = {{1.0 / 3, 1.0 / 3, 1.0 / 3}, {1 - 2 * a1, a1, a1}, {a1, 1 - 2 * a1, a1}, {a1, a1, 1 - 2 * a1},
                      {1 - 2 * a2, a2, a2},        {a2, 1 - 2 * a2, a2}, {a2, a2, 1 - 2 * a2}};
  const double w[7] = {w0, w1, w1, w1, w2, w2, w2};
  const double dl[3][2] = {{-1, -1}, {1, 0}, {0, 1}};
  const int ev[3][2] = {{1, 2}, {2, 0}, {0, 1}};
  for (int q = 0; q < 7; ++q) {
    qw[q] = w[q];
    for (int i = 0; i < 3; ++i) {
      phi1[q * 3 + i] = lam[q][i];
      phi2[q * 6 + i] = lam[q][i] * (2 * lam[q][i] - 1);
      for (int d = 0; d < 2; ++d) dphi2[(q * 6 + i) * 2 + d] = (4 * lam[q][i] - 1) * dl[i][d];
    }
    for (int k = 0; k < 3; ++k) {
      const int i = ev[k][0], j = ev[k][1];
      phi2[q * 6 + 3 + k] = 4 * lam[q][i] * lam[q][j];
      for (int d = 0; d < 2; ++d) dphi2[(q * 6 + 3 + k) * 2 + d] = 4 * (lam[q][i] * dl[j][d] + lam[q][j] * dl[i][d]);
    }
  }
}

inline int nblocks(int64_t n, int per) { return (int)((n + per - 1) / per); }

// cadence of the residual monitor for a slot: fc_set_solver_options' check_residual, -1 = auto -- every step while the factors stay in the
// Infinity Cache (the pass over the system matrix hides beside the next step), every 8th step where they stream from HBM (there the
// pass costs ~10 % of the step: cavity_fine 150 of 1 098 us).  The reference forms no residual at all (flowsolver.py:728-737).
// (partitioned handles keep every step: their residual rides in the tail record's all-reduce at no extra exchange)
// ... and so do Krylov solves (the monitor is what checks their result against the fp64 operator, at a fraction of their cost)
inline int residual_every(const fc_ctx* h, const OrderSys& S) {
  if (h->check_residual >= 0) return h->check_residual;
  return (S.nt && !h->partitioned && h->method == FC_METHOD_REFINE && !S.inexact) ? 8 : 1;
}


int pick_lanes(double mean_nnz) {
  if (const char* e = std::getenv("FC_SPMV_LANES")) {  // tuning aid
    const int v = std::atoi(e);
    if (v == 4 || v == 8 || v == 16 || v == 32 || v == 64) return v;
  }
  // the CSR kernel issues 4 predicated loads per lane and trip: lanes ~ row length / 4
  if (mean_nnz <= 16) return 4;
  if (mean_nnz <= 40) return 8;
  if (mean_nnz <= 96) return 16;
  if (mean_nnz <= 256) return 32;
  return 64;
}

int time_begin(fc_ctx* h, int kind, int launches = 1) {
  if (!h->timing) return FC_OK;
  if (h->tused + 2 > h->tev.size()) {
    for (int i = 0; i < 64; ++i) {
      hipEvent_t e;
      HIPCHK(hipEventCreate(&e));
      h->tev.push_back(e);
    }
  }
  h->tkind.push_back(kind);
  h->tcount.push_back(launches);
  HIPCHK(hipEventRecord(h->tev[h->tused], h->stream));
  return FC_OK;
}
int time_end(fc_ctx* h) {
  if (!h->timing) return FC_OK;
  HIPCHK(hipEventRecord(h->tev[h->tused + 1], h->stream));
  h->tused += 2;
  return FC_OK;
}
// after a stream synchronisation: fold the recorded pairs into the accumulators
int time_collect(fc_ctx* h) {
  if (!h->timing) return FC_OK;
  for (size_t i = 0; i < h->tused; i += 2) {
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, h->tev[i], h->tev[i + 1]));
    const int k = h->tkind[i / 2];
    h->t_ms[k] += ms;
    h->t_cnt[k] += h->tcount[i / 2];
  }
  h->tused = 0;
  h->tkind.clear();
  h->tcount.clear();
  return FC_OK;
}

// phase marks of fc_step (fc_set_phase_timing): `phase` is the phase that ends here, -1 opens a step
int phase_mark(fc_ctx* h, int phase) {
  if (!h->phase_timing) return FC_OK;
  if (h->pused >= h->pev.size()) {
    for (int i = 0; i < 32; ++i) {
      hipEvent_t e;
      HIPCHK(hipEventCreate(&e));
      h->pev.push_back(e);
    }
  }
  HIPCHK(hipEventRecord(h->pev[h->pused], h->stream));
  if (h->pid.size() <= h->pused) h->pid.resize(h->pused + 1);
  h->pid[h->pused++] = phase;
  return FC_OK;
}
int phase_collect(fc_ctx* h) {
  if (!h->phase_timing) return FC_OK;
  for (size_t i = 1; i < h->pused; ++i) {
    if (h->pid[i] < 0) continue;
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, h->pev[i - 1], h->pev[i]));
    h->ph_us[h->pid[i]] += 1e3 * ms;
  }
  for (size_t i = 0; i < h->pused; ++i) h->ph_steps += h->pid[i] < 0 ? 1 : 0;
  h->pused = 0;
  return FC_OK;
}

template <int MODE>
int launch_spmv(fc_ctx* h, int nrows, double mean, const int* rp, const int* col, const double* val, const double* x,
                const double* b, double* y, double* xsave, double* partial, const unsigned char* rowmask = nullptr) {
  const int lanes = pick_lanes(mean);
  const int rpb = 256 / lanes;
  dim3 grid(nblocks(nrows, rpb)), block(256);
  FCCHK(time_begin(h, 1));
  switch (lanes) {
    case 4: hipLaunchKernelGGL((fc_spmv_csr<4, MODE>), grid, block, 0, h->stream, nrows, rp, col, val, x, b, y, xsave, partial, rowmask); break;
    case 8: hipLaunchKernelGGL((fc_spmv_csr<8, MODE>), grid, block, 0, h->stream, nrows, rp, col, val, x, b, y, xsave, partial, rowmask); break;
    case 16: hipLaunchKernelGGL((fc_spmv_csr<16, MODE>), grid, block, 0, h->stream, nrows, rp, col, val, x, b, y, xsave, partial, rowmask); break;
    case 32: hipLaunchKernelGGL((fc_spmv_csr<32, MODE>), grid, block, 0, h->stream, nrows, rp, col, val, x, b, y, xsave, partial, rowmask); break;
    default: hipLaunchKernelGGL((fc_spmv_csr<64, MODE>), grid, block, 0, h->stream, nrows, rp, col, val, x, b, y, xsave, partial, rowmask); break;
  }
  FCCHK(time_end(h));
  HIPCHK(hipGetLastError());
  return grid.x;
}

// one instance of the fused CSR kernel of the factorisation-free preconditioner (fc_precond.hip.h):
// out[opos(i)] = base[i] + scale * dinv[i] * (rhs[rpos(i)] - (M x)[i])
int pc_launch(fc_ctx* h, const PcMat* M, int n, const double* x, const double* rhs, const int* rpos, const double* dinv, const double* base,
              double scale, double* out, const int* opos) {
  if (n <= 0) return FC_OK;
  const FcPcArgs a{n, M ? M->rp.p : nullptr, M ? M->ci.p : nullptr, M ? M->v.p : nullptr, x, rhs, rpos, dinv, base, scale, out, opos};
  const int lanes = M ? M->lanes : 1;
  const dim3 grid(nblocks(n, 256 / lanes)), block(256);
  switch (lanes) {
    case 1: hipLaunchKernelGGL(fc_pc_csr<1>, grid, block, 0, h->stream, a); break;
    case 4: hipLaunchKernelGGL(fc_pc_csr<4>, grid, block, 0, h->stream, a); break;
    case 8: hipLaunchKernelGGL(fc_pc_csr<8>, grid, block, 0, h->stream, a); break;
    case 16: hipLaunchKernelGGL(fc_pc_csr<16>, grid, block, 0, h->stream, a); break;
    default: hipLaunchKernelGGL(fc_pc_csr<32>, grid, block, 0, h->stream, a); break;
  }
  return FC_OK;
}

// out = M^-1 in with the factorisation-free preconditioner of the slot (fc_setup_krylov), both in the permuted numbering:
//   u  = k damped-Jacobi sweeps on F u = in_u                    (velocity block: mass dominated; the first two are one product)
//   zp = AMG V(1,1)-cycle on S zp = B u - in_p, S = B diag(F)^-1 Bt  (pressure Schur complement: Poisson-like; 2 launches per level)
//   zu = u - diag(F)^-1 Bt zp
// max(1, sweeps - 1) + 3 + 2 (sparse AMG levels) launches; `in` is only read, `out` only written (they may not alias).
int apply_pc(fc_ctx* h, OrderSys& S, const double* in, double* out) {
  Precond& P = S.pc;
  if (!P.ready) return fail(FC_ERR_NOT_READY, "fc_setup_krylov not called for this slot");
  double *u = P.u0.p, *un = P.u1.p;
  if (P.sweeps >= 2)
    FCCHK(pc_launch(h, &P.KF, P.nu, in, nullptr, nullptr, nullptr, nullptr, -1.0, u, nullptr));  // u = K_F in (columns address `in`)
  else
    FCCHK(pc_launch(h, nullptr, P.nu, nullptr, in, P.vpos.p, P.dinvF.p, nullptr, 1.0, u, nullptr));
  for (int k = 2; k < P.sweeps; ++k) {
    FCCHK(pc_launch(h, &P.F, P.nu, u, in, P.vpos.p, P.wdinvF.p, u, 1.0, un, nullptr));
    std::swap(u, un);
  }
  const int L = (int)P.lv.size();
  FCCHK(pc_launch(h, &P.B, P.np, u, in, P.ppos.p, nullptr, nullptr, -1.0, L ? P.lv[0].cat.p : P.rc.p, nullptr));  // B u - in_p
  for (int l = 0; l < L; ++l) {
    PcLevel& V = P.lv[(size_t)l];
    FCCHK(pc_launch(h, &V.G, V.n_next, V.cat.p, nullptr, nullptr, nullptr, nullptr, -1.0, l + 1 < L ? P.lv[(size_t)l + 1].cat.p : P.rc.p, nullptr));
  }
  hipLaunchKernelGGL(fc_pc_dense, dim3(nblocks(P.n_coarse, 4)), dim3(256), 0, h->stream, P.n_coarse, P.cinv.p, P.rc.p,
                     L ? P.lv[(size_t)L - 1].cat.p + P.lv[(size_t)L - 1].n : P.zp.p);
  for (int l = L - 1; l >= 0; --l) {
    PcLevel& V = P.lv[(size_t)l];
    FCCHK(pc_launch(h, &V.U, V.n, V.cat.p, nullptr, nullptr, nullptr, nullptr, -1.0, l > 0 ? P.lv[(size_t)l - 1].cat.p + P.lv[(size_t)l - 1].n : P.zp.p, nullptr));
  }
  hipLaunchKernelGGL(fc_pc_final, dim3(nblocks(P.nu + P.np, 64)), dim3(256), 0, h->stream, P.nu, P.np, P.Bt.rp.p, P.Bt.ci.p, P.Bt.v.p, P.zp.p,
                     P.dinvF.p, u, P.vpos.p, P.ppos.p, out);
  HIPCHK(hipGetLastError());
  return FC_OK;
}

// levels of small nodes go through the flat block kernel (fc_nd_flat_block): rows up to FC_FLAT_ROW values wide (environment, default 256;
// 0: never), tiles of at most FC_FLAT_CAP values.  Returns the loads per thread (4, 8, 12, 16) or 0.
int flat_loads(int max_row, int64_t max_tile) {
  const char* e = std::getenv("FC_FLAT_ROW");  // (read per setup, not cached: tests lay out handles with different settings in one process)
  const int flat_row = e ? std::atoi(e) : 256;
  if (flat_row <= 0 || max_row > flat_row || max_row > FC_FLAT_WD || max_tile > FC_FLAT_CAP || max_tile <= 0) return 0;
  return (int)(4 * ((max_tile + 1023) / 1024));
}

// values per tile the flat levels aim at (FC_FLAT_TILE; 2048 = 8 loads per thread measured best: refined O1 apply 241.9 -> 234.1 us,
// pinball 308.3 -> 302.8, cavity_fine 813 -> 806; 4096: 238.2 / 307.8 / 825; 1024: 240.8 / 306.6 / 823)
int flat_tile_values() {
  const char* e = std::getenv("FC_FLAT_TILE");
  return e ? std::min(FC_FLAT_CAP, std::max(256, std::atoi(e))) : 2048;
}

// down stages that will run the flat kernel: their tiles (<= 32 rows of a node, fcsym::down_blocks) are cut into equal runs of rows of at
// most flat_tile_values() values
void retile_flat(fcsym::Blocks& B) {
  fcsym::Blocks R;
  const size_t nst = B.begin.size();
  R.begin.assign(nst, 0), R.count.assign(nst, 0), R.lpr = B.lpr;
  for (size_t s = 0; s < nst; ++s) {
    R.begin[s] = (int64_t)R.val.size();
    int max_row = 0;
    for (int64_t q = B.begin[s]; q < B.begin[s] + B.count[s]; ++q) max_row = std::max(max_row, B.ni[(size_t)q] + B.nb[(size_t)q]);
    const bool flat = B.count[s] > 0 && flat_loads(max_row, 1) > 0;
    for (int64_t q = B.begin[s]; q < B.begin[s] + B.count[s]; ++q) {
      const size_t u = (size_t)q;
      const int64_t wd = (int64_t)B.ni[u] + B.nb[u];
      int parts = 1;
      if (flat && (int64_t)B.nrows[u] * wd > flat_tile_values()) {
        const int64_t fit = std::max<int64_t>(1, flat_tile_values() / wd);
        parts = (int)((B.nrows[u] + fit - 1) / fit);
      }
      const int rc = (B.nrows[u] + parts - 1) / parts;
      for (int r0 = 0; r0 < B.nrows[u]; r0 += rc) {
        R.val.push_back(B.val[u] + (int64_t)r0 * wd);
        R.row0.push_back(B.row0[u] + r0);
        R.nrows.push_back(std::min(rc, B.nrows[u] - r0));
        R.i0.push_back(B.i0[u]), R.ni.push_back(B.ni[u]), R.idx.push_back(B.idx[u]), R.nb.push_back(B.nb[u]);
      }
    }
    R.count[s] = (int)((int64_t)R.val.size() - R.begin[s]);
  }
  B = std::move(R);
}

int launch_flat(fc_ctx* h, const OrderSys& S, const FcBlk* bp, int count, int loads, bool nt, const unsigned char* vr, double* out) {
#define FC_FLAT(UU)                                                                                                                          \
  do {                                                                                                                                       \
    if (nt)                                                                                                                                  \
      hipLaunchKernelGGL((fc_nd_flat_block<UU, true>), dim3(count), dim3(256), 0, h->stream, bp, S.f_idx.p, S.f_val.p, h->buf.p, h->N, vr,  \
                         h->flag.p, out);                                                                                                    \
    else                                                                                                                                     \
      hipLaunchKernelGGL((fc_nd_flat_block<UU, false>), dim3(count), dim3(256), 0, h->stream, bp, S.f_idx.p, S.f_val.p, h->buf.p, h->N, vr, \
                         h->flag.p, out);                                                                                                    \
  } while (0)
  switch (loads) {
    case 4: FC_FLAT(4); break;
    case 8: FC_FLAT(8); break;
    case 12: FC_FLAT(12); break;
    case 16: FC_FLAT(16); break;
    default: return fail(FC_ERR_INVALID, "launch_flat: unsupported tile size");
  }
#undef FC_FLAT
  HIPCHK(hipGetLastError());
  return FC_OK;
}

int launch_sweep(fc_ctx* h, const OrderSys& S, const Stage& st) {
  if (st.kind == 2) {
    if (S.dscale.n != (size_t)h->N) return fail(FC_ERR_NOT_READY, "fc_set_stage_diag not called for a truncated factorisation");
    hipLaunchKernelGGL(fc_diag_stage, dim3(nblocks(st.nrows, 256)), dim3(256), 0, h->stream, st.nrows, S.dscale.p + st.row0, h->buf.p + st.row0,
                       h->buf.p + h->N + st.row0);
    HIPCHK(hipGetLastError());
    return FC_OK;
  }
  if (S.bits != 64) {
    // compressed factors: a reduced set of launch geometries (any geometry is correct for any row; this is the memory-lean
    // path, not the fast one), values widened to fp64 as they are loaded
    const bool f32 = S.bits == 32;
    if (st.kind == 1 && st.blk_count > 0) {
      const FcBlk* bp = S.blk.p + st.blk_begin;
      const int lpr = st.blk_lpr <= 16 ? 16 : (st.blk_lpr <= 32 ? 32 : 64);
#define FC_BLOCK_LP(L, R)                                                                                                                    \
  do {                                                                                                                                         \
    if (f32)                                                                                                                                   \
      hipLaunchKernelGGL((fc_nd_down_block<L, R, float>), dim3(st.blk_count), dim3(256), 0, h->stream, bp, S.f_idx.p, S.f_val32.p, h->buf.p, h->N); \
    else                                                                                                                                       \
      hipLaunchKernelGGL((fc_nd_down_block<L, R, FcBf16>), dim3(st.blk_count), dim3(256), 0, h->stream, bp, S.f_idx.p, S.f_val16.p, h->buf.p, h->N); \
  } while (0)
      if (lpr == 16) FC_BLOCK_LP(16, 2);
      else if (lpr == 32) FC_BLOCK_LP(32, 4);
      else FC_BLOCK_LP(64, 8);
#undef FC_BLOCK_LP
      HIPCHK(hipGetLastError());
      return FC_OK;
    }
    const int lanes = st.lanes <= 16 ? 16 : (st.lanes <= 64 ? 64 : 256), sub = lanes == 16 ? 4 : (lanes == 64 ? 16 : 64);
    dim3 grid(nblocks(st.nrows, 256 / lanes)), block(256);
    const int64_t* rp = S.seg_ptr.p + st.rp_begin;
    const int dest0 = st.kind == 0 ? st.row0 : h->N + st.row0, acc = st.kind == 0 ? 1 : 0;
#define FC_SWEEP_LP(L, SB)                                                                                                                  \
  do {                                                                                                                                        \
    if (f32)                                                                                                                                  \
      hipLaunchKernelGGL((fc_nd_sweep<L, SB, float>), grid, block, 0, h->stream, st.nrows, rp, S.seg.p, S.f_idx.p, S.f_val32.p, h->buf.p, dest0, acc); \
    else                                                                                                                                      \
      hipLaunchKernelGGL((fc_nd_sweep<L, SB, FcBf16>), grid, block, 0, h->stream, st.nrows, rp, S.seg.p, S.f_idx.p, S.f_val16.p, h->buf.p, dest0, acc); \
  } while (0)
    if (lanes == 16) FC_SWEEP_LP(16, 4);
    else if (lanes == 64) FC_SWEEP_LP(64, 16);
    else FC_SWEEP_LP(256, 64);
#undef FC_SWEEP_LP
    HIPCHK(hipGetLastError());
    return FC_OK;
  }
  if (st.kind == 1 && st.blk_count > 0 && st.blk_flat > 0)
    return launch_flat(h, S, S.blk.p + st.blk_begin, st.blk_count, st.blk_flat, st.nt, h->sweep_check ? h->velrow_p.p : nullptr, nullptr);
  if (st.kind == 1 && st.blk_count > 0) {
    const FcBlk* bp = S.blk.p + st.blk_begin;
    const unsigned char* vr = h->sweep_check ? h->velrow_p.p : nullptr;  // (overlapped tail: finiteness tested as the solution is written)
#define FC_BLOCK(L, R)                                                                                                                   \
  do {                                                                                                                                   \
    if (st.nt)                                                                                                                           \
      hipLaunchKernelGGL((fc_nd_down_block<L, R, double, true>), dim3(st.blk_count), dim3(256), 0, h->stream, bp, S.f_idx.p, S.f_val.p, \
                         h->buf.p, h->N, vr, h->flag.p);                                                                                 \
    else                                                                                                                                 \
      hipLaunchKernelGGL((fc_nd_down_block<L, R>), dim3(st.blk_count), dim3(256), 0, h->stream, bp, S.f_idx.p, S.f_val.p, h->buf.p,     \
                         h->N, vr, h->flag.p);                                                                                           \
  } while (0)
    switch (st.blk_lpr * 100 + st.blk_rps) {
      case 1601: FC_BLOCK(16, 1); break;
      case 1602: FC_BLOCK(16, 2); break;
      case 3201: FC_BLOCK(32, 1); break;
      case 3202: FC_BLOCK(32, 2); break;
      case 3204: FC_BLOCK(32, 4); break;
      case 6401: FC_BLOCK(64, 1); break;
      case 6402: FC_BLOCK(64, 2); break;
      case 6404: FC_BLOCK(64, 4); break;
      case 6408: FC_BLOCK(64, 8); break;
      default: return fail(FC_ERR_INVALID, "launch_sweep: unsupported block geometry");
    }
#undef FC_BLOCK
    HIPCHK(hipGetLastError());
    return FC_OK;
  }
  const int rpb = 256 / st.lanes;
  dim3 grid(nblocks(st.nrows, rpb)), block(256);
  const int64_t* rp = S.seg_ptr.p + st.rp_begin;
  double* buf = h->buf.p;
  const int dest0 = st.kind == 0 ? st.row0 : h->N + st.row0;
  const int acc = st.kind == 0 ? 1 : 0;
  const int* wgo = st.wg_begin >= 0 ? S.wg_order.p + st.wg_begin : nullptr;
  const unsigned char* vr = (h->sweep_check && st.kind == 1) ? h->velrow_p.p + st.row0 : nullptr;
#define FC_SWEEP(L, SB)                                                                                                          \
  do {                                                                                                                           \
    if (st.nt)                                                                                                                   \
      hipLaunchKernelGGL((fc_nd_sweep<L, SB, double, true>), grid, block, 0, h->stream, st.nrows, rp, S.seg.p, S.f_idx.p,       \
                         S.f_val.p, buf, dest0, acc, wgo, vr, h->flag.p);                                                        \
    else                                                                                                                         \
      hipLaunchKernelGGL((fc_nd_sweep<L, SB>), grid, block, 0, h->stream, st.nrows, rp, S.seg.p, S.f_idx.p, S.f_val.p, buf,     \
                         dest0, acc, wgo, vr, h->flag.p);                                                                        \
  } while (0)
  const int key = st.lanes * 1000 + st.sub;
  switch (key) {
    case 8004: FC_SWEEP(8, 4); break;
    case 8008: FC_SWEEP(8, 8); break;
    case 16004: FC_SWEEP(16, 4); break;
    case 16008: FC_SWEEP(16, 8); break;
    case 16016: FC_SWEEP(16, 16); break;
    case 32004: FC_SWEEP(32, 4); break;
    case 32008: FC_SWEEP(32, 8); break;
    case 32016: FC_SWEEP(32, 16); break;
    case 32032: FC_SWEEP(32, 32); break;
    case 64004: FC_SWEEP(64, 4); break;
    case 64008: FC_SWEEP(64, 8); break;
    case 64016: FC_SWEEP(64, 16); break;
    case 64032: FC_SWEEP(64, 32); break;
    case 64064: FC_SWEEP(64, 64); break;
    case 256016: FC_SWEEP(256, 16); break;
    case 256032: FC_SWEEP(256, 32); break;
    case 256064: FC_SWEEP(256, 64); break;
    case 256256: FC_SWEEP(256, 256); break;
    default: return fail(FC_ERR_INVALID, "launch_sweep: unsupported (lanes, sub) combination");
  }
#undef FC_SWEEP
  HIPCHK(hipGetLastError());
  return FC_OK;
}

// x_p (in buf[N..2N)) = M^-1 rhs_p, rhs_p must already be in buf[0..N)
// The exchange step of the partitioned path: sum buf[0..n) over the ranks, in place, in stream order.
// RCCL communicator: in-stream ncclAllReduce over xGMI.  Host exchange (fc_set_host_exchange): device -> pinned host,
// the caller's all-reduce (e.g. torch.distributed / gloo), host -> device.  Same launch sequence either way.
int exchange(fc_ctx* h, double* dptr, size_t n) {
  if (n == 0) return FC_OK;
  if (h->comm) {
    NCCLCHK(g_rccl.AllReduce(dptr, dptr, n, kNcclDouble, kNcclSum, h->comm, h->stream));
    return FC_OK;
  }
  if (!h->host_xchg) return fail(FC_ERR_NOT_READY, "partitioned handle without communicator: call fc_comm_init or fc_set_host_exchange");
  if (h->xstage_n < n) {
    if (h->xstage) (void)hipHostFree(h->xstage);
    h->xstage = nullptr;
    HIPCHK(hipHostMalloc((void**)&h->xstage, n * sizeof(double), hipHostMallocDefault));
    h->xstage_n = n;
  }
  HIPCHK(hipMemcpyAsync(h->xstage, dptr, n * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  h->host_xchg(h->xstage, (int64_t)n, h->host_xchg_user);
  HIPCHK(hipMemcpyAsync(dptr, h->xstage, n * sizeof(double), hipMemcpyHostToDevice, h->stream));
  return FC_OK;
}
bool exchanges(const fc_ctx* h) { return h->comm != nullptr || h->host_xchg != nullptr; }

// the state ring (fc_ctx::ring)
inline double* ring_slot(const fc_ctx* h, int k) { return h->ring.p + (size_t)(((k % 4) + 4) % 4) * 2 * (size_t)h->N; }
inline double* st_n(const fc_ctx* h) { return ring_slot(h, h->cur) + h->N; }       // (u_n, p_n), permuted numbering
inline double* st_nn(const fc_ctx* h) { return ring_slot(h, h->cur + 3) + h->N; }  // u_nn
inline void ring_point(fc_ctx* h) { h->buf.p = ring_slot(h, h->cur + 1); }
inline void ring_advance(fc_ctx* h) {  // the solution in h->buf becomes the state
  h->cur = (h->cur + 1) % 4;
  ring_point(h);
}
// the second stream (overlapped tail) idle: before anything but a time step touches the buffers its kernels read or write
int quiesce(fc_ctx* h) {
  if (h->side_busy || h->bat.side_busy) {
    HIPCHK(hipStreamSynchronize(h->stream));   // (the side stream's gate waits for the main stream's last solve)
    HIPCHK(hipStreamSynchronize(h->stream2));
    h->side_busy = false;
    // whatever late records are outstanding are complete now: take them before another call reuses the mapped page
    FCCHK(collect_late(h, 0));
    FCCHK(collect_late(h, 1));
    if (h->bat.side_busy) {
      FCCHK(collect_late_batch(h, 0));
      FCCHK(collect_late_batch(h, 1));
      h->bat.side_busy = false;
    }
  }
  return FC_OK;
}
// W-layout host vectors [u (2 nn) | p (nv)] <-> ring slots (needs a permutation)
int state_upload(fc_ctx* h, const double* wn, const double* wnn) {
  const int N = h->N, g = nblocks(N, 256);
  const double* src[2] = {wn, wnn};
  double* dst[2] = {st_n(h), st_nn(h)};
  for (int k = 0; k < 2; ++k) {
    HIPCHK(hipMemcpyAsync(h->tmpN.p, src[k], (size_t)N * sizeof(double), hipMemcpyHostToDevice, h->stream));
    hipLaunchKernelGGL(fc_gather_perm, dim3(g), dim3(256), 0, h->stream, N, h->perm.p, h->tmpN.p, dst[k]);
    HIPCHK(hipStreamSynchronize(h->stream));
  }
  h->state_live = true;
  return FC_OK;
}
int state_download(fc_ctx* h, double* wn, double* wnn) {
  const int N = h->N, g = nblocks(N, 256);
  double* dst[2] = {wn, wnn};
  const double* src[2] = {st_n(h), st_nn(h)};
  for (int k = 0; k < 2; ++k) {
    if (!dst[k]) continue;
    hipLaunchKernelGGL(fc_scatter_perm, dim3(g), dim3(256), 0, h->stream, N, h->perm.p, src[k], (const double*)nullptr, h->tmpN.p);
    HIPCHK(hipMemcpyAsync(dst[k], h->tmpN.p, (size_t)N * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
  }
  return FC_OK;
}

// Workgroups a level of the LDS-tiled block kernel should at least have (rows per workgroup are halved from 32 until it does): enough to
// keep every CU busy through the tail of the launch, and more of them -- shorter tiles -- the more rows a level has.  Measured on the
// three streaming meshes (steps/s with 1024 / 2048 / 4096 / 8192 for the down stages, profiles/r05_block_target.txt): refined O1
// (223 k dofs) 3 534 / 3 627 / 3 612 / 3 444, pinball (302 k) 2 830 / 2 899 / 2 885 / 2 797, cavity_fine (877 k) 1 037 / 1 059 / 1 088 / 1 102.
inline int block_target(int N) {
  int t = 1024;
  while (t < 8192 && (double)N / 128.0 > 1.4142 * t) t *= 2;  // nearest power of two to N / 128, within [1024, 8192]
  return t;
}

// tables of the column-form up-sweep from the in-library symbolic phase (the values are the ones the row form reads: a node's -L block
// is stored row-major, nb x ni, behind its [D^-1 | -U] rows).  Partitioned handles: the rank's own nodes only (that is what its factor
// layout holds); their slots for root rows fold into this rank's share of the root right-hand side, which the first exchange sums
int build_up_column(fc_ctx* h) {
  fc_ctx::UpCol& U = h->upc;
  U.ready = false;
  U.tried = true;
  U.lv.clear();
  if (!h->sym_ready || h->sym_truncate > 0) return FC_OK;
  const fcsym::Factors& fac = h->sym_fac;
  const fcsym::Tree& t = h->sym_tree;
  const int N = h->N;
  const size_t G = fac.nodes.size() / 7;
  auto nd = [&](size_t g, int f) { return fac.nodes[g * 7 + (size_t)f]; };  // level, n, i0, ni, nb, voff, ioff
  std::vector<int64_t> soff(G, 0);
  int64_t S = 0;
  for (size_t g = 0; g < G; ++g) {
    soff[g] = S;
    S += nd(g, 4);
  }
  std::vector<int> fptr((size_t)N + 1, 0);
  for (size_t g = 0; g < G; ++g)
    for (int64_t j = 0; j < nd(g, 4); ++j) fptr[(size_t)(fac.idx[(size_t)(nd(g, 6) + j)] - N) + 1]++;
  for (int i = 0; i < N; ++i) fptr[(size_t)i + 1] += fptr[(size_t)i];
  std::vector<int> fsrc((size_t)std::max(1, fptr[(size_t)N]));
  {
    std::vector<int> fill(fptr.begin(), fptr.end() - 1);
    for (size_t g = 0; g < G; ++g)  // nodes in elimination order (deepest first): the order of the row form's segment lists
      for (int64_t j = 0; j < nd(g, 4); ++j) fsrc[(size_t)fill[(size_t)(fac.idx[(size_t)(nd(g, 6) + j)] - N)]++] = (int)(soff[g] + j);
  }
  std::vector<FcBlk> blk;
  for (int k = t.depth; k >= 1; --k) {
    std::vector<size_t> sel;
    double values = 0.0;
    int64_t rows = 0;
    for (size_t g = 0; g < G; ++g)
      if (nd(g, 0) == k && nd(g, 3) > 0 && nd(g, 4) > 0) {
        sel.push_back(g);
        values += (double)nd(g, 3) * (double)nd(g, 4);
        rows += nd(g, 4);
      }
    fc_ctx::UpCol::Lvl L{(int64_t)blk.size(), 0, 16, 1, 0, 0};
    if (!sel.empty()) {
      const double wd = values / (double)std::max<int64_t>(rows, 1);
      static const int lpr_shift = [] { const char* e = std::getenv("FC_UPC_LPR_SHIFT"); return e ? std::atoi(e) : 0; }();  // tuning aids
      static const int rc_max = [] { const char* e = std::getenv("FC_UPC_RC"); return e ? std::max(8, std::atoi(e)) : 32; }();
      L.lpr = wd <= 32 ? 8 : (wd <= 64 ? 16 : (wd <= 128 ? 32 : 64));
      for (int q = 0; q < lpr_shift && L.lpr < 64; ++q) L.lpr *= 2;
      for (int q = 0; q > lpr_shift && L.lpr > 8; --q) L.lpr /= 2;
      const int slots = 256 / L.lpr;
      int rc = rc_max;
      static const int upc_env = [] { const char* e = std::getenv("FC_UPC_TARGET"); return e ? std::max(1, std::atoi(e)) : 0; }();
      const int upc_target = upc_env ? upc_env : std::max(2048, block_target(N) / 2);
      while (rc > slots && rows / rc < upc_target) rc /= 2;
      rc = std::max(rc, 1);
      // small nodes (rows of a few dozen values): the flat kernel, on tiles of as many whole rows as FC_FLAT_CAP values hold
      int max_ni = 0;
      for (size_t g : sel) max_ni = std::max(max_ni, (int)nd(g, 3));
      const bool flat = flat_loads(max_ni, 1) > 0;
      int64_t max_tile = 0;
      int maxr = 1;
      for (size_t g : sel) {
        const int64_t i0 = nd(g, 2), ni = nd(g, 3), nb = nd(g, 4), voff = nd(g, 5), nf = ni + nb;
        if (ni > FC_BLK_TILE * 64) return fail(FC_ERR_INVALID, "build_up_column: node too large");
        if (flat) {  // (the node's rows in equal tiles)
          const int64_t fit = std::max<int64_t>(1, flat_tile_values() / ni), parts = (nb + fit - 1) / fit;
          rc = (int)((nb + parts - 1) / parts);
        }
        for (int64_t r0 = 0; r0 < nb; r0 += rc) {
          const int nr = (int)std::min<int64_t>(rc, nb - r0);
          max_tile = std::max(max_tile, (int64_t)nr * ni);
          blk.push_back(FcBlk{(long long)(voff + ni * nf + r0 * ni), (int)(soff[g] + r0), nr, (int)i0, (int)ni, 0, 0});
          maxr = std::max(maxr, nr);
        }
      }
      std::stable_sort(blk.begin() + L.begin, blk.end(), [](const FcBlk& a, const FcBlk& b) { return (int64_t)a.nrows * a.ni > (int64_t)b.nrows * b.ni; });
      L.count = (int)(blk.size() - (size_t)L.begin);
      int rps = 1;
      while (rps * slots < maxr) rps *= 2;
      L.rps = rps;
      L.flat = flat ? flat_loads(max_ni, max_tile) : 0;
    }
    const int64_t r0 = t.node_ptr[(size_t)k - 1].front(), r1 = t.node_ptr[(size_t)k - 1].back();
    L.fold_row0 = (int)r0;
    L.fold_nrows = (int)(r1 - r0);
    U.lv.push_back(L);
  }
  if (blk.empty()) return FC_OK;
  FCCHK(U.blk.upload(blk, h->stream));
  FCCHK(U.fptr.upload(fptr, h->stream));
  FCCHK(U.fsrc.upload(fsrc, h->stream));
  FCCHK(U.scratch.alloc((size_t)std::max<int64_t>(1, S)));
  FCCHK(U.scratch.zero(h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  U.ready = true;
  return FC_OK;
}

static bool up_column_wanted(const fc_ctx* h, const OrderSys& S) { return h->up_form == 2 || (h->up_form == 0 && S.nt); }

// the up-sweep of one apply in column form: per level (deepest first) the nodes' -L blocks times their y rows -> scratch, then the fold
// of the next level's rows
int launch_up_column(fc_ctx* h, OrderSys& S) {
  fc_ctx::UpCol& U = h->upc;
  for (const fc_ctx::UpCol::Lvl& L : U.lv) {
    if (L.count > 0 && L.flat > 0) {
      FCCHK(launch_flat(h, S, U.blk.p + L.begin, L.count, L.flat, S.nt, nullptr, U.scratch.p));
    } else if (L.count > 0) {
      const FcBlk* bp = U.blk.p + L.begin;
#define FC_UPB(LP, R)                                                                                                                      \
  do {                                                                                                                                     \
    if (S.nt)                                                                                                                              \
      hipLaunchKernelGGL((fc_nd_down_block<LP, R, double, true>), dim3(L.count), dim3(256), 0, h->stream, bp, S.f_idx.p, S.f_val.p, h->buf.p, \
                         h->N, (const unsigned char*)nullptr, (int*)nullptr, U.scratch.p);                                                 \
    else                                                                                                                                   \
      hipLaunchKernelGGL((fc_nd_down_block<LP, R>), dim3(L.count), dim3(256), 0, h->stream, bp, S.f_idx.p, S.f_val.p, h->buf.p, h->N,     \
                         (const unsigned char*)nullptr, (int*)nullptr, U.scratch.p);                                                       \
  } while (0)
      switch (L.lpr * 100 + L.rps) {
        case 801: FC_UPB(8, 1); break;
        case 802: FC_UPB(8, 2); break;
        case 1604: FC_UPB(16, 4); break;
        case 1601: FC_UPB(16, 1); break;
        case 1602: FC_UPB(16, 2); break;
        case 3201: FC_UPB(32, 1); break;
        case 3202: FC_UPB(32, 2); break;
        case 3204: FC_UPB(32, 4); break;
        case 6401: FC_UPB(64, 1); break;
        case 6402: FC_UPB(64, 2); break;
        case 6404: FC_UPB(64, 4); break;
        case 6408: FC_UPB(64, 8); break;
        default: return fail(FC_ERR_INVALID, "launch_up_column: unsupported block geometry");
      }
#undef FC_UPB
    }
    if (L.fold_nrows > 0)
      hipLaunchKernelGGL(fc_nd_fold1, dim3(nblocks(L.fold_nrows, 256)), dim3(256), 0, h->stream, L.fold_nrows, L.fold_row0, U.fptr.p, U.fsrc.p, U.scratch.p, h->buf.p);
  }
  HIPCHK(hipGetLastError());
  return FC_OK;
}

int apply_factors(fc_ctx* h, OrderSys& S, int first = 0, int last = -1) {
  if (last < 0) last = (int)S.stages.size() - 1;
  // column form: all up stages of a whole apply as block + fold launches per level (fc_ctx::upc)
  bool upc = false;
  if (up_column_wanted(h, S) && first == 0 && last == (int)S.stages.size() - 1 && !S.truncated && S.bits == 64) {
    if (!h->upc.ready && !h->upc.tried && h->sym_ready) FCCHK(build_up_column(h));
    upc = h->upc.ready;
  }
  // timing: ONE event pair around the back-to-back sweep launches of this apply (a pair per launch
  // would serialise the short kernels and read ~2 us high); the launch count is recorded with it
  int nlaunch = 0;
  for (size_t i = (size_t)first; i < S.stages.size() && (int)i <= last; ++i) nlaunch += (S.stages[i].nrows > 0 && !(upc && S.stages[i].kind == 0)) ? 1 : 0;
  if (upc)
    for (const fc_ctx::UpCol::Lvl& L : h->upc.lv) nlaunch += (L.count > 0 ? 1 : 0) + (L.fold_nrows > 0 ? 1 : 0);
  FCCHK(time_begin(h, 0, nlaunch));
  if (upc) FCCHK(launch_up_column(h, S));
  for (size_t i = (size_t)first; i < S.stages.size() && (int)i <= last; ++i) {
    const Stage& st = S.stages[i];
    const bool ex = exchanges(h) && S.ar_n > 0;
    if (ex && (int)i == S.ar2_stage)  // root down-sweep: this rank fills its block of rows, the others' stay zero
      HIPCHK(hipMemsetAsync(h->buf.p + h->N + S.ar_row0, 0, (size_t)S.ar_n * sizeof(double), h->stream));
    if (st.nrows > 0 && !(upc && st.kind == 0)) FCCHK(launch_sweep(h, S, st));
    // multi-GPU: (1) the root separator's right-hand side is the sum of every rank's element and sub-tree
    // contributions; (2) the root solution is assembled from the ranks' row blocks — the two exchange steps of a solve
    if (ex && (int)i == S.ar_stage) {
      FCCHK(phase_mark(h, PH_UP));
      FCCHK(exchange(h, h->buf.p + S.ar_row0, (size_t)S.ar_n));
      FCCHK(phase_mark(h, PH_X1));
    }
    if (ex && (int)i == S.ar2_stage) {
      FCCHK(phase_mark(h, PH_ROOT));
      FCCHK(exchange(h, h->buf.p + h->N + S.ar_row0, (size_t)S.ar_n));
      FCCHK(phase_mark(h, PH_X2));
    }
    if (!ex && st.kind == 1 && i > 0 && S.stages[i - 1].kind == 0) FCCHK(phase_mark(h, PH_UP));  // single GPU: up-sweeps | root + down-sweeps
  }
  FCCHK(time_end(h));
  FCCHK(phase_mark(h, PH_DOWN));
  return FC_OK;
}

// permuted-ordering solve of A_p x = b_p.  On entry b_p is in h->b AND in the y-half of h->buf.
// max_iter = number of iterative-refinement sweeps (0: factor apply only).  With check_residual the
// first residual r0 = b - A x0 is always formed (one SpMV) and |r0|^2, |b|^2 partials are left in
// h->partial (n_rpartial blocks) for fc_final.  Returns x (and an optional correction dx to add).
int solve_permuted(fc_ctx* h, OrderSys& S, const double** x_out, const double** dx_out, int* n_rpartial) {
  const int N = h->N;
  const int g = nblocks(N, 256);
  FCCHK(apply_factors(h, S));
  const double* x = h->buf.p + N;
  const double* dx = nullptr;
  const double mean = (double)S.Ap_nnz / std::max(1, N);
  *n_rpartial = 0;
  const int iters = h->max_iter;
  const bool dist = h->partitioned && exchanges(h);
  if (h->partitioned && !dist && iters > 0) return fail(FC_ERR_INVALID, "iterative refinement on a partitioned handle needs its exchange");
  if (iters == 0 && h->check_residual) {
    // monitor only: r0 -> tmpN (x stays in the x-half of buf); partitioned: owned rows only
    const int nb = launch_spmv<1>(h, N, mean, S.Ap_rowptr.p, S.Ap_col.p, S.Ap_val.p, x, h->b.p, h->tmpN.p, nullptr,
                                  h->partial.p, h->partitioned ? h->rowkind_p.p : nullptr);
    if (nb < 0) return nb;
    *n_rpartial = nb;
  }
  // partitioned refinement: the residual is formed in the form the apply consumes -- a rank's own rows in full, the root's
  // rows as this rank's share (its columns; the right-hand side's root rows are shares already), summed by the apply's
  // first exchange.  h->b holds that right-hand side (fc_rhs_gather / fc_solve's masking).
  auto dist_residual = [&](const double* xin, bool want_partials) -> int {
    const int nb = launch_spmv<1>(h, N, mean, S.Ap_rowptr.p, S.Ap_col.p, S.Ap_val.p, xin, h->b.p, h->buf.p, nullptr,
                                  want_partials ? h->partial.p : nullptr, h->rowkind_p.p);
    if (nb < 0) return nb;
    hipLaunchKernelGGL(fc_copy, dim3(g), dim3(256), 0, h->stream, N, xin, h->tmpN.p);
    hipLaunchKernelGGL(fc_mask_rows, dim3(g), dim3(256), 0, h->stream, N, h->rowkind_p.p, h->lead ? 1 : 0, h->tmpN.p);
    const int nb2 = launch_spmv<1>(h, N, mean, S.Ap_rowptr.p, S.Ap_col.p, S.Ap_val.p, h->tmpN.p, h->b.p, h->tmpN2.p, nullptr, nullptr,
                                   h->rootmask_p.p);
    if (nb2 < 0) return nb2;
    if (S.ar_n > 0)
      hipLaunchKernelGGL(fc_copy, dim3(nblocks(S.ar_n, 256)), dim3(256), 0, h->stream, S.ar_n, h->tmpN2.p + S.ar_row0, h->buf.p + S.ar_row0);
    return nb;
  };
  for (int it = 0; it < iters; ++it) {
    if (dist) {
      if (it == 0)
        hipLaunchKernelGGL(fc_copy, dim3(g), dim3(256), 0, h->stream, N, h->buf.p + N, h->xsol.p);
      else
        hipLaunchKernelGGL(fc_axpy, dim3(g), dim3(256), 0, h->stream, N, 1.0, h->buf.p + N, h->xsol.p);
      const int nb = dist_residual(h->xsol.p, it == 0 && h->check_residual);
      if (nb < 0) return nb;
      if (it == 0 && h->check_residual) *n_rpartial = nb;
    } else if (it == 0) {
      const int nb = launch_spmv<1>(h, N, mean, S.Ap_rowptr.p, S.Ap_col.p, S.Ap_val.p, x, h->b.p, h->buf.p, h->xsol.p,
                                    h->check_residual ? h->partial.p : nullptr);
      if (nb < 0) return nb;
      if (h->check_residual) *n_rpartial = nb;
    } else {
      hipLaunchKernelGGL(fc_axpy, dim3(g), dim3(256), 0, h->stream, N, 1.0, h->buf.p + N, h->xsol.p);
      const int nb = launch_spmv<1>(h, N, mean, S.Ap_rowptr.p, S.Ap_col.p, S.Ap_val.p, h->xsol.p, h->b.p, h->buf.p,
                                    nullptr, nullptr);
      if (nb < 0) return nb;
    }
    x = h->xsol.p;
    FCCHK(apply_factors(h, S));
    dx = h->buf.p + N;
  }
  HIPCHK(hipGetLastError());
  *x_out = x;
  *dx_out = dx;
  return FC_OK;
}

int refresh_permuted(fc_ctx* h) {
  h->fvec_ok = false;  // load vectors of the body-force profiles: permuted rows, Dirichlet rows
  if (!h->have_perm) return FC_OK;
  const int N = h->N;
  // vector-scatter lists and BC slots in permuted row order
  std::vector<int> gp(N + 1, 0), gi;
  gi.reserve(h->h_gidx.size());
  for (int i = 0; i < N; ++i) {
    const int r = h->h_perm[i];
    for (int k = h->h_gptr[r]; k < h->h_gptr[r + 1]; ++k) gi.push_back(h->h_gidx[k]);
    gp[i + 1] = (int)gi.size();
  }
  FCCHK(h->gptr_p.upload(gp, h->stream));
  FCCHK(h->gidx_p.upload(gi, h->stream));
  std::vector<int> slot_of(N, -1), bs(N);
  for (int k = 0; k < h->n_bc; ++k) slot_of[h->h_bc_dofs[k]] = k;
  for (int i = 0; i < N; ++i) bs[i] = slot_of[h->h_perm[i]];
  FCCHK(h->bcslot_p.upload(bs, h->stream));
  {
    // the cells' node table and the velocity-row mask in the permuted numbering (the state vectors live there)
    std::vector<int> ip((size_t)N);
    for (int i = 0; i < N; ++i) ip[(size_t)h->h_perm[i]] = i;
    std::vector<int> cp((size_t)12 * h->nc);
    for (int c = 0; c < h->nc; ++c)
      for (int k = 0; k < 12; ++k) cp[(size_t)k * h->nc + c] = ip[(size_t)h->h_cell_dofs[(size_t)c * 15 + k]];
    FCCHK(h->cnp.upload(cp, h->stream));
    std::vector<unsigned char> vr((size_t)N);
    for (int i = 0; i < N; ++i) vr[(size_t)i] = h->h_perm[i] < 2 * h->nn ? 1 : 0;
    FCCHK(h->velrow_p.upload(vr.data(), vr.size(), h->stream));
  }
  if (h->partitioned) {
    std::vector<unsigned char> rk(N);
    for (int i = 0; i < N; ++i) rk[i] = h->h_rowkind[h->h_perm[i]];
    FCCHK(h->rowkind_p.upload(rk.data(), rk.size(), h->stream));
    FCCHK(h->rowkind_w.upload(h->h_rowkind.data(), h->h_rowkind.size(), h->stream));
    for (int i = 0; i < N; ++i) rk[i] = rk[i] == 2 ? 1 : 0;
    FCCHK(h->rootmask_p.upload(rk.data(), rk.size(), h->stream));
  }
  for (int o = 0; o < 2; ++o) {
    OrderSys& S = h->sys[o];
    if (!S.have_lift) continue;
    FCCHK(S.lift_p.alloc((size_t)std::max(1, h->n_act) * N));
    for (int k = 0; k < h->n_act; ++k)
      hipLaunchKernelGGL(fc_gather_perm, dim3(nblocks(N, 256)), dim3(256), 0, h->stream, N, h->perm.p,
                         S.lift.p + (size_t)k * N, S.lift_p.p + (size_t)k * N);
  }
  HIPCHK(hipStreamSynchronize(h->stream));
  return FC_OK;
}

struct StepCoeffs {
  double cm_n, cm_nn, cc_n, cc_nn;
};
StepCoeffs coeffs_for(const fc_ctx* h, int order_slot) {
  const double nl = h->nonlinear ? 1.0 : 0.0;
  if (order_slot == FC_SLOT_BDF1) return {1.0 / h->dt, 0.0, -nl, 0.0};
  return {2.0 / h->dt, -0.5 / h->dt, -2.0 * nl, nl};
}

int build_force_vectors(fc_ctx* h);
int check_step_ready(fc_ctx* h, int order_slot) {
  if (!h) return fail(FC_ERR_INVALID, "null handle");
  if (order_slot != FC_SLOT_BDF1 && order_slot != FC_SLOT_BDF2) return fail(FC_ERR_INVALID, "order_slot must be BDF1/BDF2");
  if (h->dt <= 0) return fail(FC_ERR_NOT_READY, "fc_set_time_scheme not called");
  if (!h->have_perm) return fail(FC_ERR_NOT_READY, "fc_set_permutation not called");
  if (!h->sys[order_slot].have_lift) return fail(FC_ERR_NOT_READY, "fc_apply_bc not called for this order");
  return build_force_vectors(h);  // (no-op unless body-force profiles changed: outside any graph capture, before the step is enqueued)
}

// enqueue RHS assembly for the current state into h->b (permuted numbering)
// Body-force actuators enter the right-hand side linearly: b += sum_k u_k F_k with F_k the load vector of profile k.  F_k is
// assembled once per profile (same element kernel, state terms off, unit amplitude on actuator k), so that the element loop of a
// time step depends on the state only -- it runs ahead of u_ctrl like that of a boundary-actuated flow (speculate_next_rhs).
int build_force_vectors(fc_ctx* h) {
  if (!h->have_force || h->fvec_ok) return FC_OK;
  HIPCHK(hipSetDevice(h->device));
  const int N = h->N, ncl = h->partitioned ? h->ncl : h->nc;
  FCCHK(h->fvec.alloc((size_t)std::max(1, h->n_act) * N));
  std::vector<double> unit((size_t)std::max(1, h->n_act), 0.0);
  DevBuf<double> amp;
  FCCHK(amp.alloc(unit.size()));
  FCCHK(h->tmpN2.zero(h->stream));  // a zero "state" (its terms are switched off; 0 x a non-finite state entry would still poison the load vector)
  for (int k = 0; k < h->n_act; ++k) {
    std::fill(unit.begin(), unit.end(), 0.0);
    unit[(size_t)k] = 1.0;
    HIPCHK(hipMemcpyAsync(amp.p, unit.data(), unit.size() * sizeof(double), hipMemcpyHostToDevice, h->stream));
    if (ncl > 0)
      hipLaunchKernelGGL(fc_rhs_elem, dim3(nblocks((int64_t)ncl * 8, 256)), dim3(256), 0, h->stream, h->nc, h->nn, h->cn.p, h->cnp.p, h->geom.p,
                         h->tmpN2.p, h->tmpN2.p, h->fprof.p, h->n_act, amp.p, 0.0, 0.0, 0.0, 0.0, h->ev.p,
                         h->partitioned ? h->cell_list.p : nullptr, ncl);
    hipLaunchKernelGGL(fc_force_rows, dim3(nblocks(N, 256)), dim3(256), 0, h->stream, N, h->gptr_p.p, h->gidx_p.p, h->ev.p,
                       h->bcslot_p.p, h->partitioned ? h->rowkind_p.p : nullptr, h->fvec.p + (size_t)k * N);
    HIPCHK(hipStreamSynchronize(h->stream));  // `unit` is rewritten for the next actuator
  }
  HIPCHK(hipGetLastError());
  h->pre_slot = -1;  // ev was used as scratch
  h->fvec_ok = true;
  return FC_OK;
}

// the element loop of a time step (no body-force profiles: pre-assembled load vectors): eight lanes per cell with the nodal values shared
// through LDS on small meshes, a thread per cell (fc_rhs_elem_reg) where the cells alone fill the SIMDs (fc_ctx::elem_reg_min cells; 0: never)
void launch_step_elem(fc_ctx* h, hipStream_t stream, const StepCoeffs& c, const double* ucoef, int ncl) {
  const int reg_min = h->elem_reg_min;
  const int* cells = h->partitioned ? h->cell_list.p : nullptr;
  if (reg_min > 0 && ncl >= reg_min)
    hipLaunchKernelGGL(fc_rhs_elem_reg, dim3(nblocks(ncl, 256)), dim3(256), 0, stream, h->nc, h->cnp.p, h->geom.p, st_n(h), st_nn(h), c.cm_n, c.cm_nn,
                       c.cc_n, c.cc_nn, h->ev.p, cells, ncl);
  else
    hipLaunchKernelGGL(fc_rhs_elem, dim3(nblocks((int64_t)ncl * 8, 256)), dim3(256), 0, stream, h->nc, h->nn, h->cn.p, h->cnp.p, h->geom.p, st_n(h),
                       st_nn(h), (const double*)nullptr, 0, ucoef, c.cm_n, c.cm_nn, c.cc_n, c.cc_nn, h->ev.p, cells, ncl);
}

int enqueue_rhs(fc_ctx* h, int order_slot, const double* d_uctrl, const double* d_uforce = nullptr) {
  const StepCoeffs c = coeffs_for(h, order_slot);
  OrderSys& S = h->sys[order_slot];
  if (!d_uforce) d_uforce = d_uctrl;
  const int ncl = h->partitioned ? h->ncl : h->nc;
  if (h->have_force && !h->fvec_ok) return fail(FC_ERR_NOT_READY, "enqueue_rhs: force vectors not built");
  const bool have_ev = h->pre_slot == order_slot;
  h->pre_slot = -1;
  if (ncl > 0 && !have_ev) launch_step_elem(h, h->stream, c, d_uforce, ncl);
  hipLaunchKernelGGL(fc_rhs_gather, dim3(nblocks(h->N, 256)), dim3(256), 0, h->stream, h->N, h->gptr_p.p, h->gidx_p.p,
                     h->ev.p, h->bcslot_p.p, h->bcprof.p, S.lift_p.p, h->n_act, d_uctrl, h->b.p, h->buf.p,
                     h->partitioned ? h->rowkind_p.p : nullptr, h->lead ? 1 : 0, S.have_c ? S.c_rowptr.p : nullptr,
                     S.c_col.p, S.c_val.p, st_n(h), h->partitioned ? h->rowkind_p.p : nullptr,
                     h->have_force ? h->fvec.p : nullptr, d_uforce);
  HIPCHK(hipGetLastError());
  return FC_OK;
}

int enqueue_energy(fc_ctx* h, const double* d_u, double* d_out) {
  const int nrows = 2 * h->nn;
  const int nb = nblocks(nrows, 32);
  hipLaunchKernelGGL(fc_energy_partial, dim3(nb), dim3(256), 0, h->stream, nrows, h->rowptr.p, h->col.p,
                     h->vals[FC_SLOT_MASS].p, d_u, h->partial.p);
  hipLaunchKernelGGL(fc_reduce_final, dim3(1), dim3(256), 0, h->stream, nb, h->partial.p, 0.5, d_out);
  HIPCHK(hipGetLastError());
  return FC_OK;
}

// single-GPU step without refinement: residual monitor + state shift + energy are ONE launch (then fc_final)
bool use_fused_tail(const fc_ctx* h) {
  static const bool enabled = [] {
    const char* e = std::getenv("FC_FUSED_TAIL");  // 0: residual SpMV and fc_finish as separate launches
    return !(e && e[0] == '0');
  }();
  return enabled && h->max_iter == 0;
}

// the solution of this step sits in the x half of h->buf (the ring slot that becomes the state): residual monitor, non-finite
// test, energy, then fc_final (sensors, record)
int launch_tail(fc_ctx* h, OrderSys& S, int compute_energy, double* d_y, double* d_E, double* d_r, double* d_flag_out,
                double* d_seq, double seq) {
  // check_residual = n >= 1: the monitor runs on every n-th step of the handle (the reference never forms this residual,
  // flowsolver.py:728-737 tests finiteness only; n > 1 amortises the matrix pass); the steps in between report NaN
  const int every = residual_every(h, S);
  const bool res = every != 0 && (h->step_count % (uint64_t)every) == 0;
  h->last_checked = res;
  const bool part = h->partitioned;
  const int ncl = part ? h->ncl : h->nc;
  const double* x = h->buf.p + h->N;
  static const int reps_env = [] { const char* e = std::getenv("FC_TAIL_REPS"); return e ? std::max(1, std::atoi(e)) : 0; }();  // tuning aid
  const int reps = reps_env ? reps_env : std::max(1, nblocks(h->N, 32 * 2048));  // <= ~2048 + ~500 partials per array for fc_final
  const int g_rows = res ? nblocks(h->N, 32 * reps) : 0, g_cells = (compute_energy && ncl > 0) ? nblocks(ncl, 32 * reps) : 0;
  const int g_check = res ? 0 : nblocks(h->N, 256 * FC_TAIL_CHECK);  // the row workgroups test finiteness themselves
  const int g = g_rows + g_check + g_cells;
  if (g > h->nblk_N) return fail(FC_ERR_INVALID, "launch_tail: partial buffer too small");
  // FC_FUSED_FINAL=1 (single GPU): the last workgroup to arrive does fc_final's work inside this launch.  Measured equal
  // to the separate launch (24.7 vs 24.2 us for tail + final on O1, identical results): the last arriver's serial chain
  // (sc1 loads of the partials, sensor rows, PCIe publish) is what fc_final costs, the boundary itself is ~1.5 us
  static const bool fuse_final = [] {
    const char* e = std::getenv("FC_FUSED_FINAL");
    return e && e[0] == '1';
  }();
  FcFin fin = {};
  const bool fused = fuse_final && !part;
  if (fused) {
    constexpr int kGroup = 32;
    const int n_groups = nblocks(g, kGroup);
    const size_t need = 32 * ((size_t)n_groups + 1);
    if (h->fin_cnt.n < need) {
      FCCHK(h->fin_cnt.alloc(std::max(need, (size_t)32 * 4096)));
      FCCHK(h->fin_cnt.zero(h->stream));
    }
    fin = FcFin{h->fin_cnt.p, kGroup, n_groups, h->n_sens, h->s_rowptr.p, h->s_idxp.p, h->s_w.p, d_y, d_E, d_r, d_flag_out, d_seq, seq};
  }
#define FC_TAIL_ARGS h->N, h->velrow_p.p, x, h->b.p, res ? S.Ap_rowptr.p : nullptr, S.Ap_col.p, S.Ap_val.p, g_rows, g_check, reps, h->nc, g_cells > 0 ? h->cnp.p : nullptr, h->geom.p, part ? h->rowkind_p.p : nullptr, part ? h->cell_list.p : nullptr, ncl, h->flag.p, h->partial.p, fin
  if (fused)
    hipLaunchKernelGGL(fc_tail<true>, dim3(g), dim3(256), 0, h->stream, FC_TAIL_ARGS);
  else
    hipLaunchKernelGGL(fc_tail<false>, dim3(g), dim3(256), 0, h->stream, FC_TAIL_ARGS);
#undef FC_TAIL_ARGS
  const double* e_part = g_cells > 0 ? h->partial.p + 2 * (size_t)g : nullptr;
  if (fused) {
    FCCHK(phase_mark(h, PH_TAIL));
  } else if (!part) {
    hipLaunchKernelGGL(fc_final, dim3(1), dim3(256), 0, h->stream, g, e_part, d_E, res ? g : 0, res ? h->partial.p : nullptr, d_r,
                       h->n_sens, h->s_rowptr.p, h->s_idxp.p, h->s_w.p, x, d_y, h->flag.p, d_flag_out, d_seq, seq);
    FCCHK(phase_mark(h, PH_TAIL));
  } else {
    // partitioned: this rank's share (owned rows, its cells, its part of every sensor row) goes to the 80-double
    // tail record, ONE all-reduce sums the ranks' records, the result is published (fc_final rewrites every
    // used word of the record each step)
    hipLaunchKernelGGL(fc_final, dim3(1), dim3(256), 0, h->stream, g, e_part, h->tail.p + 64, res ? g : 0,
                       res ? h->partial.p : nullptr, h->tail.p + 65, h->n_sens, h->s_rowptr.p, h->s_idxp.p, h->s_w.p, x,
                       h->tail.p, h->flag.p, h->tail.p + 72, (double*)nullptr, 0.0);
    FCCHK(phase_mark(h, PH_TAIL));
    FCCHK(exchange(h, h->tail.p, 80));  // the third exchange of a step: 80 doubles
    FCCHK(phase_mark(h, PH_X3));
    hipLaunchKernelGGL(fc_publish_tail, dim3(1), dim3(64), 0, h->stream, h->tail.p, d_y, h->n_sens, d_E, d_r, d_flag_out, d_seq, seq);
    FCCHK(phase_mark(h, PH_PUBLISH));
  }
  HIPCHK(hipGetLastError());
  return FC_OK;
}

// Right-preconditioned BiCGStab on the permuted system A_p x = b_p, M^-1 = the factor sweeps of the slot
// (exact factors: one iteration; factors of an EARLIER operator — fc_update_operator without
// fc_refactor — : a few; truncated factors: the memory-lean preconditioner).  b_p in h->b on entry, x_p in
// kry[0..N) on exit.  DEVICE-RESIDENT: rho, alpha, omega, the update coefficients and the convergence state live in
// h->ks; the vector kernels read them there and become no-ops once the state says "done".  The host enqueues
// FC_KRYLOV_CHECK iterations at a time and reads the state word once per batch — no synchronisation per dot product.
// Fixed reduction order.  iters / relres report what happened.
constexpr int kKrylovCheck = 4;

int krylov_state(fc_ctx* h, double* ks_host) {
  HIPCHK(hipMemcpyAsync(ks_host, h->ks.p, KS_SIZE * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  return FC_OK;
}

int bicgstab_permuted(fc_ctx* h, OrderSys& S, int* iters, double* relres) {
  const int N = h->N, g = nblocks(N, 256), gd = std::min(g, 512);
  if (h->kry.n != 8 * (size_t)N) FCCHK(h->kry.alloc(8 * (size_t)N));
  if (h->ks.n != KS_SIZE) FCCHK(h->ks.alloc(KS_SIZE));
  double *x = h->kry.p, *r = x + N, *rh = r + N, *p = rh + N, *v = p + N, *s = v + N, *t = s + N, *ph = t + N;
  double* sh = h->tmpN2.p;
  double* ks = h->ks.p;
  const double mean = (double)S.Ap_nnz / std::max(1, N);
  // Partitioned handles: every vector lives in the "distributed" form -- this rank's rows valid, the root's rows replicated
  // on all ranks, other ranks' rows zero.  Dot products run over the rows a rank accounts for and are summed over the ranks;
  // a mat-vec evaluates the rank's rows in full and the root's rows over the columns the rank accounts for, then sums the
  // root block over the ranks; the preconditioner is the partitioned factor apply (its own two exchanges).
  const bool dist = h->partitioned && exchanges(h);
  const unsigned char* kinds = dist ? h->rowkind_p.p : nullptr;
  const int lead = h->lead ? 1 : 0;
  auto dots = [&](int phase, const double* a, const double* b_, const double* c, const double* d) -> int {
    hipLaunchKernelGGL(fc_dots2, dim3(gd), dim3(256), 0, h->stream, N, a, b_, c, d, h->partial.p, kinds, lead);
    if (!dist) {
      hipLaunchKernelGGL(fc_bicg_phase, dim3(1), dim3(256), 0, h->stream, phase, gd, h->partial.p, ks, h->rtol, 1, 1);
      return FC_OK;
    }
    hipLaunchKernelGGL(fc_bicg_phase, dim3(1), dim3(256), 0, h->stream, phase, gd, h->partial.p, ks, h->rtol, 1, 0);
    FCCHK(exchange(h, ks + KS_D0, 2));
    hipLaunchKernelGGL(fc_bicg_phase, dim3(1), dim3(256), 0, h->stream, phase, gd, h->partial.p, ks, h->rtol, 0, 1);
    return FC_OK;
  };
  auto lin3 = [&](double* out, int coef, const double* v0, const double* v1, const double* v2) {
    hipLaunchKernelGGL(fc_lin3_dev, dim3(g), dim3(256), 0, h->stream, N, out, ks + KS_COEF + 3 * coef, v0, v1, v2, ks);
  };
  auto precond = [&](const double* in, double* out) -> int {  // out = M^-1 in
    if (S.factor_free) return apply_pc(h, S, in, out);
    hipLaunchKernelGGL(fc_copy, dim3(g), dim3(256), 0, h->stream, N, in, h->buf.p);
    if (dist) hipLaunchKernelGGL(fc_mask_rows, dim3(g), dim3(256), 0, h->stream, N, kinds, lead, h->buf.p);  // root rows: summed by the apply
    FCCHK(apply_factors(h, S));
    hipLaunchKernelGGL(fc_copy, dim3(g), dim3(256), 0, h->stream, N, h->buf.p + N, out);
    if (dist) hipLaunchKernelGGL(fc_mask_rows, dim3(g), dim3(256), 0, h->stream, N, kinds, 1, out);  // other ranks' rows: zero
    return FC_OK;
  };
  auto matvec = [&](const double* in, double* out) -> int {
    const int nb = launch_spmv<0>(h, N, mean, S.Ap_rowptr.p, S.Ap_col.p, S.Ap_val.p, in, nullptr, out, nullptr, nullptr, kinds);
    if (nb < 0) return nb;
    if (dist) {
      hipLaunchKernelGGL(fc_copy, dim3(g), dim3(256), 0, h->stream, N, in, h->tmpN.p);
      hipLaunchKernelGGL(fc_mask_rows, dim3(g), dim3(256), 0, h->stream, N, kinds, lead, h->tmpN.p);
      const int nb2 = launch_spmv<0>(h, N, mean, S.Ap_rowptr.p, S.Ap_col.p, S.Ap_val.p, h->tmpN.p, nullptr, h->xsol.p, nullptr, nullptr,
                                     h->rootmask_p.p);
      if (nb2 < 0) return nb2;
      if (S.ar_n > 0) {
        hipLaunchKernelGGL(fc_copy, dim3(nblocks(S.ar_n, 256)), dim3(256), 0, h->stream, S.ar_n, h->xsol.p + S.ar_row0, out + S.ar_row0);
        FCCHK(exchange(h, out + S.ar_row0, (size_t)S.ar_n));
      }
    }
    return FC_OK;
  };
  HIPCHK(hipMemsetAsync(ks, 0, KS_SIZE * sizeof(double), h->stream));
  // all work vectors start at 0: the first p = 1 r + 0 p + 0 v must not meet NaN bit patterns in fresh memory
  HIPCHK(hipMemsetAsync(x, 0, 8 * (size_t)N * sizeof(double), h->stream));
  hipLaunchKernelGGL(fc_copy, dim3(g), dim3(256), 0, h->stream, N, h->b.p, r);
  hipLaunchKernelGGL(fc_copy, dim3(g), dim3(256), 0, h->stream, N, h->b.p, rh);
  FCCHK(dots(0, r, r, rh, r));
  double kh[KS_SIZE];
  *iters = 0;
  *relres = 0.0;
  constexpr int kBicgRestarts = 8;
  int restarts = 0;
  for (int it = 1; it <= h->max_iter; ++it) {
    lin3(p, 3, r, p, v);  // p = r + beta (p - omega v)   (first iteration: p = r)
    FCCHK(precond(p, ph));
    FCCHK(matvec(ph, v));
    FCCHK(dots(1, rh, v, rh, v));
    lin3(s, 0, r, v, nullptr);  // s = r - alpha v
    FCCHK(dots(2, s, s, s, s));
    FCCHK(precond(s, sh));
    FCCHK(matvec(sh, t));
    FCCHK(dots(3, t, s, t, t));
    lin3(x, 1, x, ph, sh);      // x += alpha ph + omega sh
    lin3(r, 2, s, t, nullptr);  // r = s - omega t
    FCCHK(dots(4, r, r, rh, r));
    if (it % kKrylovCheck == 0 || it == h->max_iter) {
      FCCHK(krylov_state(h, kh));
      if (kh[KS_STATE] < 0.0 && restarts < kBicgRestarts && it < h->max_iter) {
        // breakdown (rh.v or t.t vanished): x is the last complete iterate (the vector kernels are no-ops once the state
        // is negative) — restart the recurrences from it with a new shadow residual rh = r = b - A x
        ++restarts;
        FCCHK(matvec(x, t));
        hipLaunchKernelGGL(fc_lin3, dim3(g), dim3(256), 0, h->stream, N, r, 1.0, h->b.p, -1.0, t, 0.0, (const double*)nullptr);
        hipLaunchKernelGGL(fc_copy, dim3(g), dim3(256), 0, h->stream, N, r, rh);
        FCCHK(dots(5, r, r, rh, r));
        continue;
      }
      if (kh[KS_STATE] != 0.0) break;
    }
  }
  FCCHK(krylov_state(h, kh));
  *iters = (int)kh[KS_ITERS];
  const double bnorm = std::sqrt(kh[KS_BNORM2]);
  if (kh[KS_STATE] < 0.0) return fail(FC_ERR_NOT_CONVERGED, "BiCGStab breakdown (code " + std::to_string((int)kh[KS_STATE]) + ")");
  if (!(bnorm > 0.0)) return FC_OK;  // b = 0 -> x = 0
  if (kh[KS_STATE] != 1.0) {
    *relres = std::sqrt(kh[KS_RNORM2]) / bnorm;
    return fail(FC_ERR_NOT_CONVERGED, "BiCGStab: residual " + std::to_string(*relres) + " after " + std::to_string(*iters) +
                                          " iterations (rtol " + std::to_string(h->rtol) + ")");
  }
  // report the TRUE residual of the returned x
  FCCHK(matvec(x, t));
  hipLaunchKernelGGL(fc_lin3, dim3(g), dim3(256), 0, h->stream, N, t, 1.0, h->b.p, -1.0, t, 0.0, (const double*)nullptr);
  // the same masked dot (rows this rank accounts for) and exchange as inside the loop: on a partitioned handle every rank
  // must report the SAME residual (callers branch on it)
  FCCHK(dots(-1, t, t, t, t));
  FCCHK(krylov_state(h, kh));
  *relres = std::sqrt(kh[KS_D0]) / bnorm;
  HIPCHK(hipGetLastError());
  return FC_OK;
}

int capture_graph(fc_ctx* h, hipStream_t stream, hipGraphExec_t* gx, const std::function<int()>& launches) {
  if (*gx) (void)hipGraphExecDestroy(*gx);
  *gx = nullptr;
  hipGraph_t graph = nullptr;
  HIPCHK(hipStreamBeginCapture(stream, hipStreamCaptureModeThreadLocal));
  const int code = launches();
  const hipError_t e = hipStreamEndCapture(stream, &graph);
  if (code != FC_OK) {
    if (graph) (void)hipGraphDestroy(graph);
    return code;
  }
  if (e != hipSuccess) return fail(FC_ERR_HIP, std::string("hipStreamEndCapture: ") + hipGetErrorString(e));
  const hipError_t e2 = hipGraphInstantiate(gx, graph, nullptr, nullptr, 0);
  (void)hipGraphDestroy(graph);
  if (e2 != hipSuccess) return fail(FC_ERR_HIP, std::string("hipGraphInstantiate: ") + hipGetErrorString(e2));
  return FC_OK;
}

void krylov_drop_graphs(fc_ctx* h, int slot) {
  for (hipGraphExec_t& g : h->kgraph[slot].step)
    if (g) (void)hipGraphExecDestroy(g);
  h->kgraph[slot].step.clear();
  h->kgraph[slot].sig = 0;
}

// Restarted GMRES(m), right-preconditioned by the slot's factor sweeps: x = M^-1 (V y).  Classical Gram-Schmidt with
// one re-orthogonalisation (two multi-dot launches per Arnoldi step instead of j sequential dots), Givens rotations and
// the back substitution in a one-thread kernel, everything on the device; the host reads the state word once per
// kKrylovCheck Arnoldi steps.  b_p in h->b on entry, x_p in kry[0..N) on exit.
int gmres_permuted(fc_ctx* h, OrderSys& S, int* iters, double* relres, const double* x0 = nullptr) {
  const int N = h->N, g = nblocks(N, 256), gd = std::min(g, 256);
  const int m = std::max(1, std::min(h->gmres_m, h->max_iter));
  const size_t need = (size_t)(m + 4) * N;
  if (h->kry.n < need) {
    FCCHK(h->kry.alloc(need));
    FCCHK(h->kry.zero(h->stream));  // basis columns not reached yet meet zero coefficients: keep NaN bit patterns out
  }
  if (h->ks.n != KS_SIZE) FCCHK(h->ks.alloc(KS_SIZE));
  const size_t gm_n = (size_t)(m + 1) * m + 2 * m + (m + 1) + m + (m + 2) + 2 + (m + 2);  // ... | norm2 | used | hcol2
  if (h->gm.n < gm_n) FCCHK(h->gm.alloc(gm_n));
  if (h->mdot.n < (size_t)(m + 1) * gd) FCCHK(h->mdot.alloc((size_t)(m + 1) * gd));
  double *x = h->kry.p, *r = x + N, *w = r + N, *z = w + N, *V = z + N;
  double *ks = h->ks.p, *gm = h->gm.p;
  double* hcol = gm + (size_t)(m + 1) * m + 2 * m + (m + 1) + m;
  double* norm2 = hcol + m + 2;
  double* used_p = norm2 + 1;
  double* hcol2 = used_p + 1;
  double* yv = gm + (size_t)(m + 1) * m + 2 * m + (m + 1);
  const double mean = (double)S.Ap_nnz / std::max(1, N);
  // Partitioned handles: the vectors live in the distributed form of bicgstab_permuted (a rank's own rows, the root's rows
  // replicated, zeros elsewhere); every inner product runs over the rows a rank accounts for and is summed over the ranks,
  // so the Hessenberg matrix, the rotations and the convergence state are the same on all ranks.
  const bool dist = h->partitioned && exchanges(h);
  const unsigned char* kinds = dist ? h->rowkind_p.p : nullptr;
  const int lead = h->lead ? 1 : 0;
  auto precond = [&](const double* in, double* out) -> int {
    if (S.factor_free) return apply_pc(h, S, in, out);
    hipLaunchKernelGGL(fc_copy, dim3(g), dim3(256), 0, h->stream, N, in, h->buf.p);
    if (dist) hipLaunchKernelGGL(fc_mask_rows, dim3(g), dim3(256), 0, h->stream, N, kinds, lead, h->buf.p);  // root rows: summed by the apply
    FCCHK(apply_factors(h, S));
    hipLaunchKernelGGL(fc_copy, dim3(g), dim3(256), 0, h->stream, N, h->buf.p + N, out);
    if (dist) hipLaunchKernelGGL(fc_mask_rows, dim3(g), dim3(256), 0, h->stream, N, kinds, 1, out);  // other ranks' rows: zero
    return FC_OK;
  };
  auto matvec = [&](const double* in, double* out) -> int {
    const int nb = launch_spmv<0>(h, N, mean, S.Ap_rowptr.p, S.Ap_col.p, S.Ap_val.p, in, nullptr, out, nullptr, nullptr, kinds);
    if (nb < 0) return nb;
    if (dist) {  // the root's rows: every rank's share over the columns it accounts for, summed over the ranks
      hipLaunchKernelGGL(fc_copy, dim3(g), dim3(256), 0, h->stream, N, in, h->tmpN.p);
      hipLaunchKernelGGL(fc_mask_rows, dim3(g), dim3(256), 0, h->stream, N, kinds, lead, h->tmpN.p);
      const int nb2 = launch_spmv<0>(h, N, mean, S.Ap_rowptr.p, S.Ap_col.p, S.Ap_val.p, h->tmpN.p, nullptr, h->xsol.p, nullptr, nullptr,
                                     h->rootmask_p.p);
      if (nb2 < 0) return nb2;
      if (S.ar_n > 0) {
        hipLaunchKernelGGL(fc_copy, dim3(nblocks(S.ar_n, 256)), dim3(256), 0, h->stream, S.ar_n, h->xsol.p + S.ar_row0, out + S.ar_row0);
        FCCHK(exchange(h, out + S.ar_row0, (size_t)S.ar_n));
      }
    }
    return FC_OK;
  };
  // |r|^2 -> ks[KS_D0], |b|^2 -> ks[KS_D1]
  auto norms = [&]() -> int {
    hipLaunchKernelGGL(fc_dots2, dim3(gd), dim3(256), 0, h->stream, N, r, r, h->b.p, h->b.p, h->partial.p, kinds, lead);
    hipLaunchKernelGGL(fc_bicg_phase, dim3(1), dim3(256), 0, h->stream, -1, gd, h->partial.p, ks, h->rtol, 1, 0);
    if (dist) FCCHK(exchange(h, ks + KS_D0, 2));
    return FC_OK;
  };
  // out[i] = V_i . w for i < nv (summed over the ranks)
  auto multidot = [&](int nv, const double* Vv, const double* ww, double* out) -> int {
    hipLaunchKernelGGL(fc_multidot, dim3(gd, nv), dim3(256), 0, h->stream, N, nv, Vv, ww, h->mdot.p, ks, kinds, lead);
    hipLaunchKernelGGL(fc_multidot_reduce, dim3(nv), dim3(64), 0, h->stream, nv, gd, h->mdot.p, out, 0, ks);
    if (dist) FCCHK(exchange(h, out, (size_t)nv));
    return FC_OK;
  };
  auto begin_cycle = [&](int first) -> int {
    FCCHK(norms());
    hipLaunchKernelGGL(fc_gmres_begin, dim3(1), dim3(1), 0, h->stream, m, gm, ks, h->rtol, first);
    hipLaunchKernelGGL(fc_scale_by_norm, dim3(g), dim3(256), 0, h->stream, N, r, norm2, V, ks);
    return FC_OK;
  };
  // one Arnoldi step: column j of the basis -> column j + 1, Hessenberg column, rotation, convergence state (all on the device)
  auto arnoldi = [&](int j) -> int {
    double* vj = V + (size_t)j * N;
    FCCHK(precond(vj, z));
    FCCHK(matvec(z, w));
    // classical Gram-Schmidt, twice: h = V^T w, w -= V h; h2 = V^T w, w -= V h2; Hessenberg column = h + h2
    FCCHK(multidot(j + 1, V, w, hcol));
    hipLaunchKernelGGL(fc_gmres_project, dim3(g), dim3(256), 0, h->stream, N, j + 1, V, hcol, w, ks);
    FCCHK(multidot(j + 1, V, w, hcol2));
    if (dist) {
      hipLaunchKernelGGL(fc_small_add, dim3(nblocks(j + 1, 64)), dim3(64), 0, h->stream, j + 1, hcol2, hcol, ks);
      hipLaunchKernelGGL(fc_gmres_project, dim3(g), dim3(256), 0, h->stream, N, j + 1, V, hcol2, w, ks);
      FCCHK(multidot(1, w, w, norm2));
      hipLaunchKernelGGL(fc_gmres_givens, dim3(1), dim3(1), 0, h->stream, j, m, gm, ks, h->rtol, (const double*)nullptr, (const double*)nullptr, 0);
    } else {
      // single GPU: the column's second-pass coefficients and the fold of |w|^2 ride in the rotation kernel (same sums, same order)
      hipLaunchKernelGGL(fc_gmres_project, dim3(g), dim3(256), 0, h->stream, N, j + 1, V, hcol2, w, ks);
      hipLaunchKernelGGL(fc_multidot, dim3(gd, 1), dim3(256), 0, h->stream, N, 1, w, w, h->mdot.p, ks, kinds, lead);
      hipLaunchKernelGGL(fc_gmres_givens, dim3(1), dim3(64), 0, h->stream, j, m, gm, ks, h->rtol, (const double*)hcol2, (const double*)h->mdot.p, gd);
    }
    hipLaunchKernelGGL(fc_scale_by_norm, dim3(g), dim3(256), 0, h->stream, N, w, norm2, V + (size_t)(j + 1) * N, ks);
    return FC_OK;
  };
  // factor-free slots: the step is ~30 launches of a few microseconds each -- replayed as one graph per column
  static const bool graphs_on = [] { const char* e = std::getenv("FC_KRYLOV_GRAPH"); return !(e && e[0] == '0'); }();
  const bool use_graph = graphs_on && S.factor_free && !dist && !h->timing;
  const int slot_id = (int)(&S - h->sys);
  fc_ctx::KryGraphs& KG = h->kgraph[slot_id];
  if (use_graph) {
    uint64_t sig = 1469598103934665603ull;
    auto mix = [&sig](uint64_t v) { sig = (sig ^ v) * 1099511628211ull; };
    uint64_t rbits;
    std::memcpy(&rbits, &h->rtol, sizeof rbits);
    for (uint64_t v : {(uint64_t)(uintptr_t)h->kry.p, (uint64_t)(uintptr_t)gm, (uint64_t)(uintptr_t)h->mdot.p, (uint64_t)(uintptr_t)ks, (uint64_t)(uintptr_t)S.pc.u0.p, (uint64_t)(uintptr_t)S.pc.zp.p,
                       (uint64_t)(uintptr_t)S.Ap_val.p, (uint64_t)(uintptr_t)S.pc.cinv.p, (uint64_t)m, (uint64_t)N, (uint64_t)S.pc.sweeps, (uint64_t)S.pc.lv.size(), rbits})
      mix(v);
    if (KG.sig != sig || KG.step.size() != (size_t)m) {
      krylov_drop_graphs(h, slot_id);
      KG.step.assign((size_t)m, nullptr);
      KG.sig = sig;
    }
  }
  HIPCHK(hipMemsetAsync(ks, 0, KS_SIZE * sizeof(double), h->stream));
  HIPCHK(hipMemsetAsync(gm, 0, gm_n * sizeof(double), h->stream));
  if (x0) {  // warm start (time steps of a factor-free slot: the previous solution): r = b - A x0
    hipLaunchKernelGGL(fc_copy, dim3(g), dim3(256), 0, h->stream, N, x0, x);
    FCCHK(matvec(x, w));
    hipLaunchKernelGGL(fc_lin3, dim3(g), dim3(256), 0, h->stream, N, r, 1.0, h->b.p, -1.0, w, 0.0, (const double*)nullptr);
  } else {
    HIPCHK(hipMemsetAsync(x, 0, (size_t)N * sizeof(double), h->stream));
    hipLaunchKernelGGL(fc_copy, dim3(g), dim3(256), 0, h->stream, N, h->b.p, r);
  }
  FCCHK(begin_cycle(1));
  double kh[KS_SIZE];
  *iters = 0;
  *relres = 0.0;
  int total = 0;
  bool done = false, stagnated = false;
  double last_true = -1.0;
  while (!done) {
    int j = 0;
    double state = 0.0;
    for (; j < m && total < h->max_iter; ++j, ++total) {
      if (use_graph) {
        hipGraphExec_t& gx = KG.step[(size_t)j];
        if (!gx) FCCHK(capture_graph(h, h->stream, &gx, [&]() { return arnoldi(j); }));
        HIPCHK(hipGraphLaunch(gx, h->stream));
      } else {
        FCCHK(arnoldi(j));
      }
      if ((j + 1) % kKrylovCheck == 0 || j + 1 == m || total + 1 == h->max_iter) {
        FCCHK(krylov_state(h, kh));
        state = kh[KS_STATE];
        if (state != 0.0) {
          ++j, ++total;
          break;
        }
      }
    }
    FCCHK(krylov_state(h, kh));
    state = kh[KS_STATE];
    if (state < 0.0) return fail(FC_ERR_NOT_CONVERGED, "GMRES breakdown (code " + std::to_string((int)state) + ")");
    if (state == 1.0) break;  // converged before the cycle started (or b = 0)
    HIPCHK(hipMemcpyAsync(kh, used_p, sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    const int used = (int)kh[0];
    if (state == 0.0) {
      // iteration cap inside a cycle: close it (back substitution for the columns built so far)
      return fail(FC_ERR_NOT_CONVERGED, "GMRES: iteration cap reached inside a cycle");
    }
    // x += M^-1 (V y)
    hipLaunchKernelGGL(fc_gmres_combine, dim3(g), dim3(256), 0, h->stream, N, used, V, yv, w);
    FCCHK(precond(w, z));
    hipLaunchKernelGGL(fc_axpy, dim3(g), dim3(256), 0, h->stream, N, 1.0, z, x);
    // true residual
    FCCHK(matvec(x, w));
    hipLaunchKernelGGL(fc_lin3, dim3(g), dim3(256), 0, h->stream, N, r, 1.0, h->b.p, -1.0, w, 0.0, (const double*)nullptr);
    if (state == 3.0 && total < h->max_iter) {
      // converged by the Arnoldi residual: let the TRUE residual decide (the two part ways by eps * cond on operators next to
      // singular), and go on with a fresh cycle -- iterative refinement in effect -- as long as that still buys a factor of two
      FCCHK(begin_cycle(0));
      FCCHK(krylov_state(h, kh));
      if (kh[KS_STATE] == 1.0) {
        done = true;
      } else {
        if (last_true >= 0.0 && !(kh[KS_RNORM2] < 0.25 * last_true)) done = stagnated = true;
        last_true = kh[KS_RNORM2];
      }
    } else if (state == 3.0) {
      done = true;
    } else if (total >= h->max_iter) {
      done = true;
    } else {
      FCCHK(begin_cycle(0));
    }
  }
  FCCHK(norms());  // the same masked sums and exchange on every rank: all ranks report the same residual
  FCCHK(krylov_state(h, kh));
  *iters = (int)kh[KS_ITERS];
  const double bnorm = std::sqrt(kh[KS_D1]);
  *relres = bnorm > 0.0 ? std::sqrt(kh[KS_D0]) / bnorm : 0.0;
  HIPCHK(hipGetLastError());
  if (bnorm > 0.0 && !(*relres <= 10.0 * h->rtol) && !(stagnated && h->floor_ok && *relres < 1e-4))
    return fail(FC_ERR_NOT_CONVERGED, "GMRES: residual " + std::to_string(*relres) + " after " + std::to_string(*iters) + " iterations" +
                                          (stagnated ? " (stagnated: the operator's conditioning sets this floor)" : ""));
  return FC_OK;
}

// Inexact factors (S.inexact): a slot whose direct apply is not accurate enough still preconditions GMRES to round-off in a few
// iterations, so REFINE turns into GMRES(<= 60, 1e-12) for it -- transparently, in fc_solve and inside the time steps.
struct KrylovOverride {
  fc_ctx* h;
  int method, max_iter;
  double rtol;
  bool on;
  KrylovOverride(fc_ctx* h_, const OrderSys& S) : h(h_), method(h_->method), max_iter(h_->max_iter), rtol(h_->rtol), on(h_->method == FC_METHOD_REFINE && S.inexact) {
    if (on) {
      h->method = FC_METHOD_GMRES;
      h->max_iter = 60;
      h->rtol = 1e-12;
      h->floor_ok = true;
    }
  }
  ~KrylovOverride() {
    if (on) {
      h->floor_ok = false;
      h->method = method;
      h->max_iter = max_iter;
      h->rtol = rtol;
    }
  }
};

// enqueue one full step; y -> d_y, E -> d_E; residual norms -> scal[1], scal[2]
int enqueue_step_launches(fc_ctx* h, int order_slot, const double* d_uctrl, double* d_y, double* d_E, double* d_r,
                          double* d_flag_out, int compute_energy, const double* d_uforce, double* d_seq, double seq) {
  OrderSys& S = h->sys[order_slot];
  if (!S.ready) return fail(FC_ERR_NOT_READY, "fc_solver_setup not called for this order");
  if (S.truncated && h->method == FC_METHOD_REFINE)
    return fail(FC_ERR_INVALID, "truncated factors are a preconditioner: set FC_METHOD_GMRES or FC_METHOD_BICGSTAB");
  if (S.factor_free && h->method == FC_METHOD_REFINE)
    return fail(FC_ERR_INVALID, "this slot has no factors (fc_setup_krylov): set FC_METHOD_GMRES or FC_METHOD_BICGSTAB");
  if (compute_energy && !h->partitioned && !h->have_mp) return fail(FC_ERR_NOT_READY, "fc_set_energy_matrix not called");
  FCCHK(phase_mark(h, -1));
  FCCHK(enqueue_rhs(h, order_slot, d_uctrl, d_uforce));
  FCCHK(phase_mark(h, PH_RHS));
  const double *x = nullptr, *dx = nullptr;
  int nrp = 0;
  KrylovOverride inexact_factors(h, S);
  if (h->method != FC_METHOD_REFINE) {
    // Krylov solve inside the step (the memory-lean path: truncated factors as preconditioner; or lagged factors):
    // the drivers synchronise with the host every few iterations; the residual monitor of the tail checks the result
    if (h->partitioned) {
      // the assembled right-hand side holds this rank's rows in full and its PARTIAL sums of the root's rows: the Krylov
      // vectors want the root's rows complete on every rank and zeros on the other ranks' rows
      if (!exchanges(h)) return fail(FC_ERR_INVALID, "partitioned handle without an exchange");
      if (S.ar_n > 0) FCCHK(exchange(h, h->b.p + S.ar_row0, (size_t)S.ar_n));
      hipLaunchKernelGGL(fc_mask_rows, dim3(nblocks(h->N, 256)), dim3(256), 0, h->stream, h->N, h->rowkind_p.p, 1, h->b.p);
    }
    int iters = 0;
    double relres = 0.0;
    // factor-free slots start GMRES from the previous solution (u_n, p_n): |b - A x_n| / |b| is O(dt) already
    const double* x0 = (S.factor_free && h->state_live && h->pc_warm_start) ? st_n(h) : nullptr;
    FCCHK(h->method == FC_METHOD_GMRES ? gmres_permuted(h, S, &iters, &relres, x0) : bicgstab_permuted(h, S, &iters, &relres));
    h->last_krylov_iters = iters;
    hipLaunchKernelGGL(fc_copy, dim3(nblocks(h->N, 256)), dim3(256), 0, h->stream, h->N, h->kry.p, h->buf.p + h->N);
    return launch_tail(h, S, compute_energy, d_y, d_E, d_r, d_flag_out, d_seq, seq);
  }
  if (use_fused_tail(h)) {
    FCCHK(apply_factors(h, S));
    return launch_tail(h, S, compute_energy, d_y, d_E, d_r, d_flag_out, d_seq, seq);
  }
  FCCHK(solve_permuted(h, S, &x, &dx, &nrp));
  h->last_checked = nrp > 0;
  // the new state = the x half of the work buffer: after refinement sweeps that half holds the last correction, x the sum so far
  if (dx) hipLaunchKernelGGL(fc_axpy, dim3(nblocks(h->N, 256)), dim3(256), 0, h->stream, h->N, 1.0, x, h->buf.p + h->N);
  const double* xn = h->buf.p + h->N;
  const int g = nblocks(h->N, 32);  // fc_finish: 8 lanes per row, 32 rows per workgroup
  double* e_partial = h->partial.p + 2 * (size_t)h->nblk_N;  // energy partials live after the residual ones
  if (!h->partitioned) {
    hipLaunchKernelGGL(fc_finish, dim3(g), dim3(256), 0, h->stream, h->N, h->velrow_p.p, xn, h->flag.p, compute_energy ? h->mp_rowptr.p : nullptr,
                       h->mp_col.p, h->mp_val.p, compute_energy ? e_partial : nullptr, (const unsigned char*)nullptr);
    hipLaunchKernelGGL(fc_final, dim3(1), dim3(256), 0, h->stream, g, compute_energy ? e_partial : nullptr,
                       d_E, nrp, nrp > 0 ? h->partial.p : nullptr, d_r, h->n_sens, h->s_rowptr.p, h->s_idxp.p,
                       h->s_w.p, xn, d_y, h->flag.p, d_flag_out, d_seq, seq);
  } else {
    // partitioned: owned + root rows, energy from this rank's cells, sensor rows restricted to
    // owned dofs; the partial tail is summed over the ranks with one small all-reduce
    hipLaunchKernelGGL(fc_finish, dim3(g), dim3(256), 0, h->stream, h->N, h->velrow_p.p, xn, h->flag.p, (const int*)nullptr, (const int*)nullptr,
                       (const double*)nullptr, (double*)nullptr, h->rowkind_p.p);
    int ne = 0;
    if (compute_energy && h->ncl > 0) {
      ne = nblocks(h->ncl, 256);
      hipLaunchKernelGGL(fc_energy_elem, dim3(ne), dim3(256), 0, h->stream, h->nc, h->cnp.p, h->geom.p, xn, h->cell_list.p, h->ncl, e_partial);
    }
    HIPCHK(hipMemsetAsync(h->tail.p, 0, 128 * sizeof(double), h->stream));
    hipLaunchKernelGGL(fc_final, dim3(1), dim3(256), 0, h->stream, ne, ne > 0 ? e_partial : nullptr,
                       h->tail.p + 64, nrp, nrp > 0 ? h->partial.p : nullptr, h->tail.p + 65, h->n_sens, h->s_rowptr.p,
                       h->s_idxp.p, h->s_w.p, xn, h->tail.p, h->flag.p, h->tail.p + 72, (double*)nullptr, 0.0);
    FCCHK(phase_mark(h, PH_TAIL));
    FCCHK(exchange(h, h->tail.p, 80));
    FCCHK(phase_mark(h, PH_X3));
    hipLaunchKernelGGL(fc_publish_tail, dim3(1), dim3(64), 0, h->stream, h->tail.p, d_y, h->n_sens, d_E, d_r, d_flag_out,
                       d_seq, seq);
    FCCHK(phase_mark(h, PH_PUBLISH));
  }
  if (!h->partitioned) FCCHK(phase_mark(h, PH_TAIL));
  HIPCHK(hipGetLastError());
  return FC_OK;
}

// enqueue one full step; y -> d_y, E -> d_E; residual norms -> scal[1], scal[2].  On return the state ring has moved on: the
// solution the launches write IS the new (u_n, p_n), the old u_n is u_nn (nothing is copied)
int enqueue_step(fc_ctx* h, int order_slot, const double* d_uctrl, double* d_y, double* d_E, double* d_r,
                 double* d_flag_out, int compute_energy, const double* d_uforce = nullptr, double* d_seq = nullptr,
                 double seq = 0.0) {
  FCCHK(enqueue_step_launches(h, order_slot, d_uctrl, d_y, d_E, d_r, d_flag_out, compute_energy, d_uforce, d_seq, seq));
  ring_advance(h);
  h->state_live = true;
  ++h->step_count;
  return FC_OK;
}

}  // namespace

extern "C" {

const char* fc_last_error(void) { return g_err.c_str(); }

int fc_device_count(int* count) {
  if (!count) return fail(FC_ERR_INVALID, "null count");
  hipError_t e = hipGetDeviceCount(count);
  if (e != hipSuccess) {
    *count = 0;
    return fail(FC_ERR_HIP, std::string("hipGetDeviceCount: ") + hipGetErrorString(e));
  }
  return FC_OK;
}

int fc_create(fc_handle* out, int device, int32_t nv, int32_t ne, int32_t nc, const double* coords, const int32_t* cells,
              const int32_t* cell_edges) {
  if (!out || !coords || !cells || !cell_edges || nv <= 0 || ne <= 0 || nc <= 0)
    return fail(FC_ERR_INVALID, "fc_create: bad arguments");
  if ((int64_t)nc * 225 > std::numeric_limits<int>::max())
    return fail(FC_ERR_INVALID, "fc_create: mesh too large for int32 element-matrix indexing");
  *out = nullptr;
  HIPCHK(hipSetDevice(device));
  fc_ctx* h = new fc_ctx();
  h->device = device;
  h->nv = nv;
  h->ne = ne;
  h->nc = nc;
  h->nn = nv + ne;
  h->N = 2 * h->nn + nv;
  const int nn = h->nn, N = h->N;
  auto bail = [&](int code) {
    fc_destroy(h);
    return code;
  };
#define TRY(expr)                     \
  do {                                \
    int _s = (expr);                  \
    if (_s != FC_OK) return bail(_s); \
  } while (0)
#define TRYHIP(expr)                                                                               \
  do {                                                                                             \
    hipError_t _e = (expr);                                                                        \
    if (_e != hipSuccess) return bail(fail(FC_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e))); \
  } while (0)
  {
    hipDeviceProp_t prop;
    TRYHIP(hipGetDeviceProperties(&prop, device));
    h->n_cu = std::max(1, prop.multiProcessorCount);
  }
  TRYHIP(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
  TRYHIP(hipStreamCreateWithFlags(&h->stream2, hipStreamNonBlocking));
  if (const char* e = std::getenv("FC_OVERLAP_TAIL")) h->overlap = e[0] != '0';
  if (const char* e = std::getenv("FC_GATE_SPIN")) h->gate_spin = std::max(0L, std::atol(e));
  if (const char* e = std::getenv("FC_ELEM_REG_MIN")) h->elem_reg_min = std::atoi(e);
  if (const char* e = std::getenv("FC_UP_FORM")) h->up_form = std::string(e) == "row" ? 1 : (std::string(e) == "column" ? 2 : 0);
  TRYHIP(hipEventCreate(&h->ev0));
  TRYHIP(hipEventCreate(&h->ev1));
  TRYHIP(hipHostMalloc((void**)&h->pin, kPinDoubles * sizeof(double), hipHostMallocMapped | hipHostMallocCoherent));
  TRYHIP(hipHostGetDevicePointer((void**)&h->pin_dev, h->pin, 0));
  std::memset(h->pin, 0, kPinDoubles * sizeof(double));
  // element tables -> constant memory
  double phi2[42], dphi2[84], phi1[21], qw[7];
  tabulate(phi2, dphi2, phi1, qw);
  TRYHIP(hipMemcpyToSymbol(HIP_SYMBOL(c_phi2), phi2, sizeof(phi2)));
  TRYHIP(hipMemcpyToSymbol(HIP_SYMBOL(c_dphi2), dphi2, sizeof(dphi2)));
  TRYHIP(hipMemcpyToSymbol(HIP_SYMBOL(c_phi1), phi1, sizeof(phi1)));
  TRYHIP(hipMemcpyToSymbol(HIP_SYMBOL(c_qw), qw, sizeof(qw)));
  // SoA cell tables and geometry
  std::vector<int> cn((size_t)6 * nc);
  std::vector<double> geom((size_t)5 * nc);
  std::vector<int> cd((size_t)nc * 15);
  for (int c = 0; c < nc; ++c) {
    int v[3], e[3];
    for (int k = 0; k < 3; ++k) {
      v[k] = cells[3 * c + k];
      e[k] = cell_edges[3 * c + k];
      if (v[k] < 0 || v[k] >= nv || e[k] < 0 || e[k] >= ne) {
        return bail(fail(FC_ERR_INVALID, "fc_create: cell index out of range"));
      }
      cn[(size_t)k * nc + c] = v[k];
      cn[(size_t)(3 + k) * nc + c] = nv + e[k];
    }
    const double x0 = coords[2 * v[0]], y0 = coords[2 * v[0] + 1];
    const double a = coords[2 * v[1]] - x0, b = coords[2 * v[2]] - x0;       // J = [[a b],[c d]]
    const double cc = coords[2 * v[1] + 1] - y0, d = coords[2 * v[2] + 1] - y0;
    const double det = a * d - b * cc;
    if (!(det > 0.0)) return bail(fail(FC_ERR_INVALID, "fc_create: cells must be counter-clockwise and non-degenerate"));
    geom[c] = d / det;                    // Jinv[0][0]
    geom[(size_t)nc + c] = -b / det;      // Jinv[0][1]
    geom[(size_t)2 * nc + c] = -cc / det; // Jinv[1][0]
    geom[(size_t)3 * nc + c] = a / det;   // Jinv[1][1]
    geom[(size_t)4 * nc + c] = det;
    for (int k = 0; k < 6; ++k) {
      const int node = cn[(size_t)k * nc + c];
      cd[(size_t)c * 15 + k] = node;
      cd[(size_t)c * 15 + 6 + k] = nn + node;
    }
    for (int k = 0; k < 3; ++k) cd[(size_t)c * 15 + 12 + k] = 2 * nn + v[k];
  }
  h->h_cell_dofs = cd;
  h->h_cent.resize((size_t)nc * 2);
  for (int c = 0; c < nc; ++c)
    for (int d = 0; d < 2; ++d)
      h->h_cent[2 * (size_t)c + d] = (coords[2 * cells[3 * c] + d] + coords[2 * cells[3 * c + 1] + d] + coords[2 * cells[3 * c + 2] + d]) / 3.0;
  TRY(h->cn.upload(cn, h->stream));
  TRY(h->geom.upload(geom, h->stream));
  // CSR pattern (no pressure-pressure coupling)
  {
    std::vector<uint64_t> keys;
    keys.reserve((size_t)nc * 216);
    for (int c = 0; c < nc; ++c)
      for (int i = 0; i < 15; ++i)
        for (int j = 0; j < 15; ++j) {
          if (i >= 12 && j >= 12) continue;
          keys.push_back(((uint64_t)cd[(size_t)c * 15 + i] << 32) | (uint32_t)cd[(size_t)c * 15 + j]);
        }
    std::sort(keys.begin(), keys.end());
    keys.erase(std::unique(keys.begin(), keys.end()), keys.end());
    if (keys.size() > (size_t)std::numeric_limits<int>::max())
      return bail(fail(FC_ERR_INVALID, "fc_create: pattern exceeds int32 nnz"));
    h->nnz = (int64_t)keys.size();
    h->h_rowptr.assign(N + 1, 0);
    h->h_col.resize(keys.size());
    for (size_t k = 0; k < keys.size(); ++k) {
      h->h_rowptr[(keys[k] >> 32) + 1]++;
      h->h_col[k] = (int)(keys[k] & 0xFFFFFFFFu);
    }
    for (int r = 0; r < N; ++r) h->h_rowptr[r + 1] += h->h_rowptr[r];
  }
  TRY(h->rowptr.upload(h->h_rowptr, h->stream));
  TRY(h->col.upload(h->h_col, h->stream));
  // inverted index of matrix contributions: slot -> list of em entries
  {
    std::vector<int> cnt(h->nnz + 1, 0);
    std::vector<int> slot_of((size_t)nc * 225, -1);
    for (int c = 0; c < nc; ++c)
      for (int i = 0; i < 15; ++i) {
        const int r = cd[(size_t)c * 15 + i];
        const int* b0 = h->h_col.data() + h->h_rowptr[r];
        const int* b1 = h->h_col.data() + h->h_rowptr[r + 1];
        for (int j = 0; j < 15; ++j) {
          if (i >= 12 && j >= 12) continue;
          const int cc2 = cd[(size_t)c * 15 + j];
          const int s = (int)(std::lower_bound(b0, b1, cc2) - h->h_col.data());
          slot_of[(size_t)c * 225 + i * 15 + j] = s;
          cnt[s + 1]++;
        }
      }
    for (int64_t s = 0; s < h->nnz; ++s) cnt[s + 1] += cnt[s];
    std::vector<int> midx(cnt[h->nnz]);
    std::vector<int> fill(cnt.begin(), cnt.end() - 1);
    for (int c = 0; c < nc; ++c)  // cell-major order => fixed summation order per slot
      for (int ij = 0; ij < 225; ++ij) {
        const int s = slot_of[(size_t)c * 225 + ij];
        if (s >= 0) midx[fill[s]++] = ij * nc + c;
      }
    TRY(h->mptr.upload(cnt, h->stream));
    TRY(h->midx.upload(midx, h->stream));
  }
  // inverted index of vector contributions: W row -> list of ev entries (velocity rows only)
  {
    h->h_gptr.assign(N + 1, 0);
    for (int c = 0; c < nc; ++c)
      for (int k = 0; k < 12; ++k) h->h_gptr[cd[(size_t)c * 15 + k] + 1]++;
    for (int r = 0; r < N; ++r) h->h_gptr[r + 1] += h->h_gptr[r];
    h->h_gidx.resize(h->h_gptr[N]);
    std::vector<int> fill(h->h_gptr.begin(), h->h_gptr.end() - 1);
    for (int c = 0; c < nc; ++c)
      for (int k = 0; k < 12; ++k) h->h_gidx[fill[cd[(size_t)c * 15 + k]]++] = k * nc + c;
  }
  TRY(h->em.alloc((size_t)225 * nc));
  TRY(h->ev.alloc((size_t)12 * nc));
  for (int s = 0; s < FC_NUM_SLOTS; ++s) TRY(h->vals[s].alloc((size_t)h->nnz));
  TRY(h->ring.alloc(8 * (size_t)N));  // the state ring: four work buffers [y | x]
  TRY(h->ring.zero(h->stream));
  h->cur = 0;
  h->buf.n = 2 * (size_t)N;
  ring_point(h);
  TRY(h->bstore.alloc(2 * (size_t)N));
  TRY(h->bstore.zero(h->stream));
  h->b.p = h->bstore.p;
  h->b.n = (size_t)N;
  TRY(h->xsol.alloc(N));
  TRY(h->tmpN.alloc(N));
  TRY(h->tmpN2.alloc(N));
  h->nblk_N = nblocks(N, 4) + 16;  // upper bound on blocks of any row-wise reduction
  TRY(h->partial.alloc(3 * (size_t)h->nblk_N));
  TRY(h->scal.alloc(8));
  TRY(h->flag.alloc(1));
  TRY(h->flag2.alloc(1));
  TRY(h->flag2.zero(h->stream));
  TRY(h->flag2x.alloc(32));
  TRY(h->flag2x.zero(h->stream));
  TRY(h->solved.alloc(1));
  TRY(h->solved.zero(h->stream));
  TRY(h->side_err.alloc(1));
  TRY(h->side_err.zero(h->stream));
  TRY(h->isbc.alloc(N));
  TRY(h->scal.zero(h->stream));
  TRY(h->flag.zero(h->stream));
  TRY(h->isbc.zero(h->stream));
  TRY(h->uctrl.alloc(64));
  TRY(h->uctrl.zero(h->stream));
  TRY(h->ydev.alloc(64));
  TRY(h->tail.alloc(128));
  TRY(h->tail.zero(h->stream));
  TRYHIP(hipStreamSynchronize(h->stream));
#undef TRY
#undef TRYHIP
  *out = h;
  return FC_OK;
}

static void batch_drop_graphs(fc_ctx* h);
static int batch_repack(fc_ctx* h, int slot);

int fc_destroy(fc_handle h) {
  if (!h) return FC_OK;
  (void)hipSetDevice(h->device);
  if (h->stream2) (void)hipStreamSynchronize(h->stream2);
  if (h->stream) (void)hipStreamSynchronize(h->stream);
  batch_drop_graphs(h);
  krylov_drop_graphs(h, 0), krylov_drop_graphs(h, 1);
  if (h->stream2) (void)hipStreamDestroy(h->stream2);
  if (h->comm && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(h->comm);
  if (h->pin) (void)hipHostFree(h->pin);
  if (h->xstage) (void)hipHostFree(h->xstage);
  if (h->ev0) (void)hipEventDestroy(h->ev0);
  if (h->ev1) (void)hipEventDestroy(h->ev1);
  for (hipEvent_t e : h->tev) (void)hipEventDestroy(e);
  for (hipEvent_t e : h->pev) (void)hipEventDestroy(e);
  hipStream_t s = h->stream;
  delete h;
  if (s) (void)hipStreamDestroy(s);
  return FC_OK;
}

int fc_get_sizes(fc_handle h, int64_t* N, int64_t* nnz, int64_t* nn) {
  if (!h) return fail(FC_ERR_INVALID, "null handle");
  if (N) *N = h->N;
  if (nnz) *nnz = h->nnz;
  if (nn) *nn = h->nn;
  return FC_OK;
}

int fc_get_pattern(fc_handle h, int32_t* rowptr, int32_t* colidx) {
  if (!h || !rowptr || !colidx) return fail(FC_ERR_INVALID, "fc_get_pattern: null argument");
  std::memcpy(rowptr, h->h_rowptr.data(), (size_t)(h->N + 1) * sizeof(int));
  std::memcpy(colidx, h->h_col.data(), (size_t)h->nnz * sizeof(int));
  return FC_OK;
}

int fc_assemble_matrix(fc_handle h, int slot, double mass, double nu, const double* adv, double adv_scale,
                       const double* lin, double lin_scale, double pressure, double divergence) {
  if (!h || slot < 0 || slot >= FC_NUM_SLOTS) return fail(FC_ERR_INVALID, "fc_assemble_matrix: bad slot");
  HIPCHK(hipSetDevice(h->device));
  FCCHK(quiesce(h));
  const size_t nv2 = 2 * (size_t)h->nn;
  double *d_adv = nullptr, *d_lin = nullptr;
  if (adv) {
    HIPCHK(hipMemcpyAsync(h->tmpN.p, adv, nv2 * sizeof(double), hipMemcpyHostToDevice, h->stream));
    d_adv = h->tmpN.p;
  }
  if (lin) {
    HIPCHK(hipMemcpyAsync(h->tmpN2.p, lin, nv2 * sizeof(double), hipMemcpyHostToDevice, h->stream));
    d_lin = h->tmpN2.p;
  }
  // (a partitioned handle assembles the element matrices of its own cells and of the cells touching a root dof: the rows it owns and
  //  the root's rows come out complete, the other ranks' rows -- never read on this rank -- partial)
  const bool sub = h->partitioned && h->n_asm > 0;
  hipLaunchKernelGGL(fc_mat_elem, dim3(nblocks(sub ? h->n_asm : h->nc, 256), 6), dim3(256), 0, h->stream, h->nc, h->nn, h->cn.p,
                     h->geom.p, mass, nu, d_adv, adv_scale, d_lin, lin_scale, pressure, divergence, h->em.p,
                     sub ? h->asm_cells.p : nullptr, sub ? h->n_asm : 0);
  hipLaunchKernelGGL(fc_mat_gather, dim3(nblocks(h->nnz, 256)), dim3(256), 0, h->stream, h->nnz, h->mptr.p, h->midx.p,
                     h->em.p, h->vals[slot].p);
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(h->stream));
  h->slot_ok[slot] = true;
  if (slot < 2) {
    h->sys[slot].have_lift = false;
    h->sys[slot].ready = false;
  }
  return FC_OK;
}

int fc_get_matrix_values(fc_handle h, int slot, double* vals) {
  if (!h || slot < 0 || slot >= FC_NUM_SLOTS || !vals) return fail(FC_ERR_INVALID, "fc_get_matrix_values: bad argument");
  if (!h->slot_ok[slot]) return fail(FC_ERR_NOT_READY, "slot not assembled");
  HIPCHK(hipSetDevice(h->device));
  HIPCHK(hipMemcpyAsync(vals, h->vals[slot].p, (size_t)h->nnz * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  return FC_OK;
}

int fc_set_matrix_values(fc_handle h, int slot, const double* vals) {
  if (!h || slot < 0 || slot >= FC_NUM_SLOTS || !vals) return fail(FC_ERR_INVALID, "fc_set_matrix_values: bad argument");
  HIPCHK(hipSetDevice(h->device));
  HIPCHK(hipMemcpyAsync(h->vals[slot].p, vals, (size_t)h->nnz * sizeof(double), hipMemcpyHostToDevice, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  h->slot_ok[slot] = true;
  return FC_OK;
}

int fc_spmv(fc_handle h, int slot, const double* x, double* y) {
  if (!h || slot < 0 || slot >= FC_NUM_SLOTS || !x || !y) return fail(FC_ERR_INVALID, "fc_spmv: bad argument");
  if (!h->slot_ok[slot]) return fail(FC_ERR_NOT_READY, "slot not assembled");
  HIPCHK(hipSetDevice(h->device));
  FCCHK(quiesce(h));
  HIPCHK(hipMemcpyAsync(h->tmpN.p, x, (size_t)h->N * sizeof(double), hipMemcpyHostToDevice, h->stream));
  const int nb = launch_spmv<0>(h, h->N, (double)h->nnz / h->N, h->rowptr.p, h->col.p, h->vals[slot].p, h->tmpN.p, nullptr,
                                h->tmpN2.p, nullptr, nullptr);
  if (nb < 0) return nb;
  if (h->partitioned && exchanges(h) && h->rowkind_w.n == (size_t)h->N) {
    // a partitioned handle holds complete rows for the dofs it owns and for the root's only (fc_assemble_matrix): the product is a
    // collective -- every rank keeps the rows it accounts for (its own; the root's on the lead rank), one all-reduce assembles it
    hipLaunchKernelGGL(fc_mask_rows, dim3(nblocks(h->N, 256)), dim3(256), 0, h->stream, h->N, h->rowkind_w.p, h->lead ? 1 : 0, h->tmpN2.p);
    FCCHK(exchange(h, h->tmpN2.p, (size_t)h->N));
  }
  HIPCHK(hipMemcpyAsync(y, h->tmpN2.p, (size_t)h->N * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  return FC_OK;
}

int fc_bench_spmv(fc_handle h, int slot, int reps, double* ms_per_launch) {
  if (!h || slot < 0 || slot >= FC_NUM_SLOTS || reps <= 0 || !ms_per_launch)
    return fail(FC_ERR_INVALID, "fc_bench_spmv: bad argument");
  if (!h->slot_ok[slot]) return fail(FC_ERR_NOT_READY, "slot not assembled");
  HIPCHK(hipSetDevice(h->device));
  FCCHK(quiesce(h));
  for (int i = 0; i < 3; ++i) {
    const int nb = launch_spmv<0>(h, h->N, (double)h->nnz / h->N, h->rowptr.p, h->col.p, h->vals[slot].p, h->tmpN.p,
                                  nullptr, h->tmpN2.p, nullptr, nullptr);
    if (nb < 0) return nb;
  }
  HIPCHK(hipEventRecord(h->ev0, h->stream));
  for (int i = 0; i < reps; ++i) {
    const int nb = launch_spmv<0>(h, h->N, (double)h->nnz / h->N, h->rowptr.p, h->col.p, h->vals[slot].p, h->tmpN.p,
                                  nullptr, h->tmpN2.p, nullptr, nullptr);
    if (nb < 0) return nb;
  }
  HIPCHK(hipEventRecord(h->ev1, h->stream));
  HIPCHK(hipEventSynchronize(h->ev1));
  float ms = 0.f;
  HIPCHK(hipEventElapsedTime(&ms, h->ev0, h->ev1));
  *ms_per_launch = (double)ms / reps;
  return FC_OK;
}

int fc_set_bc(fc_handle h, int32_t n_bc, const int32_t* bc_dofs, int32_t n_act, const double* profiles) {
  if (!h || n_bc < 0 || n_act < 0 || n_act > 64 || (n_bc > 0 && !bc_dofs) || (n_bc > 0 && n_act > 0 && !profiles))
    return fail(FC_ERR_INVALID, "fc_set_bc: bad argument");
  HIPCHK(hipSetDevice(h->device));
  FCCHK(quiesce(h));
  std::vector<unsigned char> isbc(h->N, 0);
  for (int k = 0; k < n_bc; ++k) {
    if (bc_dofs[k] < 0 || bc_dofs[k] >= 2 * h->nn) return fail(FC_ERR_INVALID, "fc_set_bc: dof is not a velocity dof");
    if (isbc[bc_dofs[k]]) return fail(FC_ERR_INVALID, "fc_set_bc: duplicate dof");
    isbc[bc_dofs[k]] = 1;
  }
  {
    // the elimination tree parks the Dirichlet dofs in the leaves: a different SET needs a new tree
    std::vector<int> a(h->h_bc_dofs), b(bc_dofs, bc_dofs + n_bc);
    std::sort(a.begin(), a.end());
    std::sort(b.begin(), b.end());
    if (a != b && h->sym_ready) {
      h->sym_ready = false;
      h->upc.ready = h->upc.tried = false;
      h->have_plan = false;
      h->bat.tables = h->bat.tb_built = false;  // and the batched launch tables with it (fc_set_batch rebuilds them)
      h->bat.k = h->bat.KB = 0;
      for (int o = 0; o < 2; ++o) h->sys[o].structured = false;
    }
  }
  h->n_bc = n_bc;
  h->n_act = n_act;
  h->bat.ctrl_ok[0] = h->bat.ctrl_ok[1] = false;
  h->bat.pre_slot = -1;
  h->h_bc_dofs.assign(bc_dofs, bc_dofs + n_bc);
  h->h_bcprof.assign(profiles, profiles + (size_t)n_bc * n_act);
  FCCHK(h->isbc.upload(isbc.data(), isbc.size(), h->stream));
  std::vector<double> prof(std::max<size_t>(1, (size_t)n_bc * n_act), 0.0);
  if (n_bc * n_act) std::copy(profiles, profiles + (size_t)n_bc * n_act, prof.begin());
  FCCHK(h->bcprof.upload(prof, h->stream));
  for (int o = 0; o < 2; ++o) h->sys[o].have_lift = h->sys[o].ready = false;
  HIPCHK(hipStreamSynchronize(h->stream));
  if (h->have_perm) FCCHK(refresh_permuted(h));
  return FC_OK;
}

int fc_set_force(fc_handle h, int32_t n_act, const double* profiles) {
  if (!h) return fail(FC_ERR_INVALID, "null handle");
  HIPCHK(hipSetDevice(h->device));
  FCCHK(quiesce(h));
  h->fvec_ok = false;
  if (!profiles || n_act == 0) {
    h->have_force = false;
    return FC_OK;
  }
  if (n_act != h->n_act) return fail(FC_ERR_INVALID, "fc_set_force: n_act differs from fc_set_bc");
  FCCHK(h->fprof.upload(profiles, (size_t)n_act * 2 * h->nn, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  h->have_force = true;
  return FC_OK;
}

static int upload_sensors(fc_ctx* h) {
  // partitioned: every rank evaluates the part of each sensor row that lives on dofs it owns (root dofs: lead rank
  // only); the partial readings are summed by the step's last exchange
  const int n_sens = h->n_sens;
  if (n_sens == 0) return FC_OK;
  std::vector<int> rp(1, 0), idx;
  std::vector<double> w;
  for (int s = 0; s < n_sens; ++s) {
    for (int k = h->h_s_rowptr[s]; k < h->h_s_rowptr[s + 1]; ++k) {
      const int d = h->h_s_idx[k];
      if (h->partitioned && !h->h_rowkind.empty()) {
        const unsigned char kind = h->h_rowkind[(size_t)d];
        if (!(kind == 1 || (kind == 2 && h->lead))) continue;
      }
      idx.push_back(d);
      w.push_back(h->h_s_w[k]);
    }
    rp.push_back((int)idx.size());
  }
  if (idx.empty()) {
    idx.push_back(0);
    w.push_back(0.0);
  }
  FCCHK(h->s_rowptr.upload(rp, h->stream));
  FCCHK(h->s_idx.upload(idx, h->stream));
  FCCHK(h->s_w.upload(w, h->stream));
  h->have_sidxp = false;
  if (h->have_perm) {
    std::vector<int> ip((size_t)h->N), idxp(idx.size());
    for (int i = 0; i < h->N; ++i) ip[(size_t)h->h_perm[i]] = i;
    for (size_t k = 0; k < idx.size(); ++k) idxp[k] = ip[(size_t)idx[k]];
    FCCHK(h->s_idxp.upload(idxp, h->stream));
    h->have_sidxp = true;
  }
  HIPCHK(hipStreamSynchronize(h->stream));
  return FC_OK;
}

int fc_set_sensors(fc_handle h, int32_t n_sens, const int32_t* rowptr, const int32_t* idx, const double* w) {
  if (!h || n_sens < 0 || n_sens > 64 || (n_sens > 0 && (!rowptr || !idx || !w)))
    return fail(FC_ERR_INVALID, "fc_set_sensors: bad argument");
  HIPCHK(hipSetDevice(h->device));
  FCCHK(quiesce(h));
  const int nz = n_sens ? rowptr[n_sens] : 0;
  for (int k = 0; k < nz; ++k)
    if (idx[k] < 0 || idx[k] >= h->N) return fail(FC_ERR_INVALID, "fc_set_sensors: index out of range");
  h->n_sens = n_sens;
  h->h_s_rowptr.assign(rowptr, rowptr + (n_sens ? n_sens + 1 : 0));
  h->h_s_idx.assign(idx, idx + nz);
  h->h_s_w.assign(w, w + nz);
  return upload_sensors(h);
}

int fc_set_time_scheme(fc_handle h, double dt, int nonlinear) {
  if (!h || !(dt > 0.0)) return fail(FC_ERR_INVALID, "fc_set_time_scheme: dt must be positive");
  h->dt = dt;
  h->nonlinear = nonlinear ? 1 : 0;
  h->pre_slot = -1;
  h->bat.pre_slot = -1;
  return FC_OK;
}

int fc_apply_bc(fc_handle h, int slot) {
  if (!h || slot < 0 || slot >= FC_NUM_SLOTS) return fail(FC_ERR_INVALID, "fc_apply_bc: bad slot");
  if (!h->slot_ok[slot]) return fail(FC_ERR_NOT_READY, "slot not assembled");
  HIPCHK(hipSetDevice(h->device));
  FCCHK(quiesce(h));
  const int N = h->N;
  if (slot < 2) {
    OrderSys& S = h->sys[slot];
    FCCHK(S.lift.alloc((size_t)std::max(1, h->n_act) * N));
    FCCHK(S.lift.zero(h->stream));
    std::vector<double> g(N);
    for (int k = 0; k < h->n_act; ++k) {
      std::fill(g.begin(), g.end(), 0.0);
      for (int i = 0; i < h->n_bc; ++i) g[h->h_bc_dofs[i]] = h->h_bcprof[(size_t)i * h->n_act + k];
      HIPCHK(hipMemcpyAsync(h->tmpN.p, g.data(), (size_t)N * sizeof(double), hipMemcpyHostToDevice, h->stream));
      const int nb = launch_spmv<0>(h, N, (double)h->nnz / N, h->rowptr.p, h->col.p, h->vals[slot].p, h->tmpN.p, nullptr,
                                    S.lift.p + (size_t)k * N, nullptr, nullptr);
      if (nb < 0) return nb;
      HIPCHK(hipStreamSynchronize(h->stream));
    }
    S.have_lift = true;
    S.ready = false;
    if (slot < 2) h->bat.ctrl_ok[slot] = false;  // the rows with a lifting entry are listed from these vectors (build_ctrl_rows)
    h->bat.pre_slot = -1;
  }
  hipLaunchKernelGGL(fc_apply_bc_rows, dim3(nblocks(N, 256)), dim3(256), 0, h->stream, N, h->rowptr.p, h->col.p,
                     h->isbc.p, h->vals[slot].p);
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(h->stream));
  if (h->have_perm) FCCHK(refresh_permuted(h));
  return FC_OK;
}

int fc_set_permutation(fc_handle h, const int32_t* perm) {
  if (!h || !perm) return fail(FC_ERR_INVALID, "fc_set_permutation: null argument");
  HIPCHK(hipSetDevice(h->device));
  FCCHK(quiesce(h));
  std::vector<unsigned char> seen(h->N, 0);
  for (int i = 0; i < h->N; ++i) {
    if (perm[i] < 0 || perm[i] >= h->N || seen[perm[i]]) return fail(FC_ERR_INVALID, "fc_set_permutation: not a permutation");
    seen[perm[i]] = 1;
  }
  // the state vectors live in the permuted numbering: carry them over (W layout on the host in between)
  std::vector<double> wn, wnn;
  if (h->have_perm && h->state_live) {
    wn.resize((size_t)h->N), wnn.resize((size_t)h->N);
    FCCHK(state_download(h, wn.data(), wnn.data()));
  } else if (!h->hs_n.empty()) {
    wn.swap(h->hs_n), wnn.swap(h->hs_nn);
  }
  h->h_perm.assign(perm, perm + h->N);
  FCCHK(h->perm.upload(h->h_perm, h->stream));
  {
    std::vector<int> ip((size_t)h->N);
    for (int i = 0; i < h->N; ++i) ip[(size_t)h->h_perm[i]] = i;
    FCCHK(h->iperm.upload(ip, h->stream));
  }
  h->have_perm = true;
  h->have_mp = false;
  h->pre_slot = -1;
  h->bat.ctrl_ok[0] = h->bat.ctrl_ok[1] = false;
  h->undo_ok = false;
  for (int o = 0; o < 2; ++o) h->sys[o].ready = h->sys[o].structured = false;
  FCCHK(upload_sensors(h));  // sensor positions in the permuted numbering
  FCCHK(refresh_permuted(h));
  if (!wn.empty()) FCCHK(state_upload(h, wn.data(), wnn.data()));
  return FC_OK;
}

int fc_solver_setup(fc_handle h, int slot, const int32_t* Ap_rowptr, const int32_t* Ap_col, const double* Ap_val,
                    int32_t n_stages, const int64_t* stage_begin, const int32_t* stage_row0,
                    const int32_t* stage_nrows, const int32_t* stage_kind, const int64_t* seg_ptr, int64_t n_seg,
                    const int64_t* seg_val, const int32_t* seg_col, const int32_t* seg_len, int64_t n_idx,
                    const int32_t* idx, int64_t n_val, const double* vals, int32_t ar_stage, int32_t ar_row0,
                    int32_t ar_n, int32_t ar2_stage) {
  if (!h || slot < 0 || slot > 1 || !Ap_rowptr || !Ap_col || n_stages <= 0 || !stage_begin || !stage_row0 ||
      !stage_nrows || !stage_kind || !seg_ptr || !seg_val || !seg_col || !seg_len || n_seg < 0 || n_idx < 0 ||
      n_val <= 0 || (n_idx > 0 && !idx))
    return fail(FC_ERR_INVALID, "fc_solver_setup: bad argument");
  if (!h->have_perm) return fail(FC_ERR_NOT_READY, "fc_set_permutation not called");
  HIPCHK(hipSetDevice(h->device));
  FCCHK(quiesce(h));
  const int N = h->N;
  OrderSys& S = h->sys[slot];
  S.ready = false;
  if (S.factor_free) {
    S.factor_free = false;
    krylov_drop_graphs(h, slot);
    S.pc.release();
    S.ap_src.release();
  }
  S.Ap_nnz = Ap_rowptr[N];
  for (int64_t k = 0; k < S.Ap_nnz; ++k)
    if (Ap_col[k] < 0 || Ap_col[k] >= N) return fail(FC_ERR_INVALID, "fc_solver_setup: system column out of range");
  // validate every segment against the buffers it will index (a bad table would fault the GPU)
  for (int64_t q = 0; q < n_seg; ++q) {
    const int64_t len = seg_len[q];
    if (len < 0 || seg_val[q] < 0 || seg_val[q] + len > n_val) return fail(FC_ERR_INVALID, "fc_solver_setup: segment values out of range");
    if (seg_col[q] >= 0) {
      if ((int64_t)seg_col[q] + len > 2 * (int64_t)N) return fail(FC_ERR_INVALID, "fc_solver_setup: segment columns out of range");
    } else {
      const int64_t o = -((int64_t)seg_col[q] + 1);
      if (o + len > n_idx) return fail(FC_ERR_INVALID, "fc_solver_setup: segment index list out of range");
    }
  }
  for (int64_t k = 0; k < n_idx; ++k)
    if (idx[k] < 0 || idx[k] >= 2 * N) return fail(FC_ERR_INVALID, "fc_solver_setup: shared index out of range");
  int64_t total_rows = 0;
  S.stages.clear();
  S.sweep_bytes = 0.0;
  std::vector<int> wg_order;  // workgroup order of every sweep launch, one after the other
  for (int s = 0; s < n_stages; ++s) {
    Stage st;
    st.rp_begin = stage_begin[s];
    st.row0 = stage_row0[s];
    st.nrows = stage_nrows[s];
    st.kind = stage_kind[s];
    if (st.nrows < 0 || st.row0 < 0 || st.row0 + st.nrows > N || st.kind < 0 || st.kind > 2 || st.rp_begin != total_rows)
      return fail(FC_ERR_INVALID, "fc_solver_setup: inconsistent stage table");
    const int64_t q0 = seg_ptr[total_rows], q1 = seg_ptr[total_rows + st.nrows];
    if (q0 < 0 || q1 < q0 || q1 > n_seg) return fail(FC_ERR_INVALID, "fc_solver_setup: segment pointers out of range");
    int64_t nz = 0, nz_idx = 0;
    for (int64_t q = q0; q < q1; ++q) {
      nz += seg_len[q];
      if (seg_col[q] < 0) nz_idx += seg_len[q];
      // an up stage must only read the y-half; a down stage reads y of its own level and x above
      if (st.kind == 0 && (seg_col[q] < 0 || (int64_t)seg_col[q] + seg_len[q] > N))
        return fail(FC_ERR_INVALID, "fc_solver_setup: up-sweep segment reads outside y");
    }
    const double mean_seg = (q1 > q0) ? (double)nz / (double)(q1 - q0) : 0.0;
    const double mean_row = st.nrows ? (double)nz / st.nrows : 0.0;
    // geometry of the launch (see fc_nd_sweep): SUB lanes cover one segment with a 4-deep issue,
    // G = LANES / SUB segments of a row are in flight together
    auto pow2_ceil = [](double v) { int p = 1; while (p < v) p <<= 1; return p; };
    auto pow2_floor = [](double v) { int p = 1; while (2 * p <= v) p <<= 1; return p; };
    const double segs_per_row = st.nrows ? (double)(q1 - q0) / st.nrows : 0.0;
    int sub, lanes;
    const bool few_long_rows = (int64_t)st.nrows < 4096 && mean_row >= 1024;
    if (st.kind == 0) {
      // up: every segment is a contiguous slice -> run G of them side by side
      sub = std::min(64, std::max(4, pow2_ceil(mean_seg / 4.0)));
      // ... as long as the launch is short of threads: with >= FC_UP_THREADS lanes from the rows alone, rows in flight hide the
      // latency better than segments of one row in flight (cavity_fine, the three deepest up stages: 64 lanes x 4 segments ->
      // 16 lanes x 1 segment took 45 us off the 1 000 us apply)
      static const double up_threads = [] { const char* e = std::getenv("FC_UP_THREADS"); return e ? std::atof(e) : FC_UP_THREADS; }();
      const int grp_fill = std::max(1, pow2_floor(up_threads / std::max(1.0, (double)st.nrows * sub)));
      const int grp = std::min(grp_fill, std::min(16, std::max(1, pow2_floor(segs_per_row))));
      lanes = sub * grp;
      if (lanes > 64) {
        if (few_long_rows) {
          lanes = 256;  // a whole workgroup per row keeps >= ~4 waves on every SIMD near the root
          sub = std::max(sub, 16);
        } else {
          lanes = 64;
        }
      }
    } else {
      // down: one contiguous (D^-1) and one indexed (-U) segment per row; running them side by
      // side would diverge inside the wave, so all lanes of the row walk them one after the other
      double dd = FC_DOWN_DEPTH;
      if (const char* e = std::getenv("FC_DOWN_DEPTH")) dd = std::max(1.0, std::atof(e));  // tuning aid
      lanes = few_long_rows ? 256 : std::min(64, std::max(8, pow2_ceil(mean_seg / dd)));
      sub = lanes;
    }
    if (lanes < 8) lanes = 8;
    if (sub > lanes) sub = lanes;
    if (const char* e = std::getenv("FC_SWEEP_GEOM")) {  // tuning aid: "lanes:sub,lanes:sub,..." per stage, 0 = keep
      int idx = 0, l = 0, sb = 0;
      const char* p = e;
      while (*p && idx <= s) {
        l = std::atoi(p);
        const char* c = std::strchr(p, ':');
        sb = c ? std::atoi(c + 1) : 0;
        if (idx == s && l > 0 && sb > 0) {
          lanes = l;
          sub = sb;
        }
        const char* nx = std::strchr(p, ',');
        if (!nx) break;
        p = nx + 1;
        ++idx;
      }
    }
    st.lanes = lanes;
    st.sub = sub;
    // algorithmic bytes: values 8 B, x/y operand 8 B per value is served on-chip (vectors are < 1 MB),
    // segment descriptors 16 B, shared index lists 4 B per indexed value (re-used by the rows of a node:
    // counted once per row as an upper bound), row pointers 8 B, destination 8 B (+8 B read when accumulating)
    st.bytes = 8.0 * (double)nz + 16.0 * (double)(q1 - q0) + 4.0 * (double)nz_idx +
               (double)st.nrows * (8.0 + 8.0 + (st.kind == 0 ? 8.0 : 0.0));
    S.sweep_bytes += st.bytes;
    // row groups of this launch by decreasing work (values behind their rows); FC_WG_SORT=0: row order
    static const bool wg_sort = [] { const char* e = std::getenv("FC_WG_SORT"); return !(e && e[0] == '0'); }();
    if (wg_sort && st.kind != 2 && st.nrows > 0) {
      const int rpb = 256 / st.lanes, ng = (st.nrows + rpb - 1) / rpb;
      std::vector<int64_t> work((size_t)ng, 0);
      for (int r = 0; r < st.nrows; ++r)
        for (int64_t q = seg_ptr[total_rows + r]; q < seg_ptr[total_rows + r + 1]; ++q) work[(size_t)(r / rpb)] += seg_len[q];
      std::vector<int> order((size_t)ng);
      for (int g = 0; g < ng; ++g) order[(size_t)g] = g;
      std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return work[(size_t)a] > work[(size_t)b]; });
      st.wg_begin = (int64_t)wg_order.size();
      wg_order.insert(wg_order.end(), order.begin(), order.end());
    }
    S.stages.push_back(st);
    total_rows += st.nrows;
  }
  if (wg_order.empty()) wg_order.push_back(0);
  FCCHK(S.wg_order.upload(wg_order, h->stream));
  if (ar_stage >= n_stages || ar_row0 < 0 || ar_n < 0 || (int64_t)ar_row0 + ar_n > N)
    return fail(FC_ERR_INVALID, "fc_solver_setup: bad all-reduce range");
  if (ar2_stage >= n_stages || (ar2_stage >= 0 && (ar2_stage <= ar_stage || S.stages[ar2_stage].kind != 1 || S.stages[ar2_stage].row0 < ar_row0 ||
                                                   S.stages[ar2_stage].row0 + S.stages[ar2_stage].nrows > ar_row0 + ar_n)))
    return fail(FC_ERR_INVALID, "fc_solver_setup: the second exchange stage must be a down stage inside the exchanged rows");
  S.ar2_stage = ar2_stage;
  S.ar_stage = ar_stage;
  S.ar_row0 = ar_row0;
  S.ar_n = ar_n;
  S.f_nnz = n_val;
  {
    static const double nt_bytes = [] { const char* e = std::getenv("FC_NT_BYTES"); return e ? std::atof(e) : FC_NT_BYTES; }();
    S.nt = 8.0 * (double)n_val > nt_bytes;
    // ... except for the stages that fit a resident budget, taken in launch order (the deepest up levels first: short segments,
    // latency-bound -- the ones a cache hit helps most)
    static const double resident = [] { const char* e = std::getenv("FC_RESIDENT_BYTES"); return e ? std::atof(e) : FC_RESIDENT_BYTES; }();
    double kept = 0.0;
    for (Stage& st : S.stages) {
      st.nt = S.nt;
      if (S.nt && kept + st.bytes <= resident) {
        st.nt = false;
        kept += st.bytes;
      }
    }
  }
  FCCHK(S.Ap_rowptr.upload(Ap_rowptr, N + 1, h->stream));
  FCCHK(S.Ap_col.upload(Ap_col, (size_t)S.Ap_nnz, h->stream));
  if (Ap_val) {
    FCCHK(S.Ap_val.upload(Ap_val, (size_t)S.Ap_nnz, h->stream));
  } else {  // structure only: fc_refactor / fc_update_operator fill the values
    FCCHK(S.Ap_val.alloc((size_t)S.Ap_nnz));
    FCCHK(S.Ap_val.zero(h->stream));
  }
  FCCHK(S.seg_ptr.upload(seg_ptr, (size_t)total_rows + 1, h->stream));
  {
    std::vector<FcSeg> packed((size_t)std::max<int64_t>(1, n_seg));
    for (int64_t q = 0; q < n_seg; ++q) packed[q] = FcSeg{(long long)seg_val[q], seg_col[q], seg_len[q]};
    FCCHK(S.seg.upload(packed, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
  }
  if (n_idx > 0) FCCHK(S.f_idx.upload(idx, (size_t)n_idx, h->stream));
  else FCCHK(S.f_idx.alloc(1));
  // 64 zero values behind the last block: the batched block kernel (fc_nd_block_b) reads value pairs in groups of 8
  // columns and may touch up to 7 values past the end of a row (they meet zero operand rows)
  S.bits = h->factor_bits;
  S.f_val.release(), S.f_val32.release(), S.f_val16.release();
  if (S.bits == 64) {
    FCCHK(S.f_val.alloc((size_t)n_val + 64));
    FCCHK(S.f_val.zero(h->stream));
    if (vals) HIPCHK(hipMemcpyAsync(S.f_val.p, vals, (size_t)n_val * sizeof(double), hipMemcpyHostToDevice, h->stream));
  } else {
    // compressed factors: only the rounded values are ever stored (fc_fe_export writes them straight from the fronts)
    if (vals) return fail(FC_ERR_INVALID, "fc_solver_setup: host-supplied factor values need 64-bit storage (fc_set_factor_precision)");
    if (S.bits == 32) {
      FCCHK(S.f_val32.alloc((size_t)n_val + 64));
      HIPCHK(hipMemsetAsync(S.f_val32.p, 0, S.f_val32.n * sizeof(float), h->stream));
    } else {
      FCCHK(S.f_val16.alloc((size_t)n_val + 64));
      HIPCHK(hipMemsetAsync(S.f_val16.p, 0, S.f_val16.n * sizeof(FcBf16), h->stream));
    }
  }
  HIPCHK(hipStreamSynchronize(h->stream));
  S.truncated = S.bits != 64;  // compressed factors are a preconditioner, like truncated ones
  for (const Stage& st : S.stages) S.truncated = S.truncated || st.kind == 2;
  S.ready = true;
  S.structured = true;
  return FC_OK;
}

int fc_set_stage_diag(fc_handle h, int slot, const double* dscale) {
  if (!h || slot < 0 || slot > 1 || !dscale) return fail(FC_ERR_INVALID, "fc_set_stage_diag: bad argument");
  HIPCHK(hipSetDevice(h->device));
  FCCHK(quiesce(h));
  FCCHK(h->sys[slot].dscale.upload(dscale, (size_t)h->N, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  return FC_OK;
}

int fc_solver_set_blocks(fc_handle h, int slot, int32_t n_stages, const int64_t* stage_blk_begin, const int32_t* stage_blk_count,
                         const int32_t* stage_lpr, int64_t n_blk, const int64_t* blk_val, const int32_t* blk_row0,
                         const int32_t* blk_nrows, const int32_t* blk_i0, const int32_t* blk_ni, const int32_t* blk_idx,
                         const int32_t* blk_nb, int64_t n_idx, int64_t n_val) {
  if (!h || slot < 0 || slot > 1 || !stage_blk_begin || !stage_blk_count || !stage_lpr || n_blk < 0 ||
      (n_blk > 0 && (!blk_val || !blk_row0 || !blk_nrows || !blk_i0 || !blk_ni || !blk_idx || !blk_nb)))
    return fail(FC_ERR_INVALID, "fc_solver_set_blocks: bad argument");
  OrderSys& S = h->sys[slot];
  if (!S.ready) return fail(FC_ERR_NOT_READY, "fc_solver_setup must be called first");
  if (n_stages != (int)S.stages.size()) return fail(FC_ERR_INVALID, "fc_solver_set_blocks: stage count differs from fc_solver_setup");
  if (n_val != S.f_nnz) return fail(FC_ERR_INVALID, "fc_solver_set_blocks: value count differs from fc_solver_setup");
  HIPCHK(hipSetDevice(h->device));
  FCCHK(quiesce(h));
  const int N = h->N;
  std::vector<FcBlk> packed((size_t)std::max<int64_t>(1, n_blk));
  for (int64_t q = 0; q < n_blk; ++q) {
    const int64_t wd = (int64_t)blk_ni[q] + blk_nb[q];
    if (blk_nrows[q] <= 0 || blk_nrows[q] > 32 || blk_row0[q] < 0 || (int64_t)blk_row0[q] + blk_nrows[q] > N || blk_ni[q] < 0 ||
        blk_nb[q] < 0 || blk_i0[q] < 0 || (int64_t)blk_i0[q] + blk_ni[q] > N || blk_idx[q] < 0 ||
        (blk_nb[q] > 0 && (int64_t)blk_idx[q] + blk_nb[q] > n_idx) || blk_val[q] < 0 ||
        blk_val[q] + (int64_t)blk_nrows[q] * wd > n_val)
      return fail(FC_ERR_INVALID, "fc_solver_set_blocks: block descriptor out of range");
    packed[q] = FcBlk{(long long)blk_val[q], blk_row0[q], blk_nrows[q], blk_i0[q], blk_ni[q], blk_idx[q], blk_nb[q]};
  }
  for (int s = 0; s < n_stages; ++s) {
    Stage& st = S.stages[s];
    if (stage_blk_count[s] < 0 || stage_blk_begin[s] < 0 || stage_blk_begin[s] + stage_blk_count[s] > n_blk)
      return fail(FC_ERR_INVALID, "fc_solver_set_blocks: stage block range out of bounds");
    if (stage_blk_count[s] > 0) {
      if (st.kind != 1) return fail(FC_ERR_INVALID, "fc_solver_set_blocks: blocks are for down stages only");
      if (stage_lpr[s] != 16 && stage_lpr[s] != 32 && stage_lpr[s] != 64) return fail(FC_ERR_INVALID, "fc_solver_set_blocks: lanes per row must be 16, 32 or 64");
      // the blocks of a stage must tile exactly the stage's destination rows
      int64_t rows = 0;
      for (int64_t q = stage_blk_begin[s]; q < stage_blk_begin[s] + stage_blk_count[s]; ++q) {
        if (blk_row0[q] < st.row0 || blk_row0[q] + blk_nrows[q] > st.row0 + st.nrows) return fail(FC_ERR_INVALID, "fc_solver_set_blocks: block outside its stage");
        rows += blk_nrows[q];
      }
      if (rows != st.nrows) return fail(FC_ERR_INVALID, "fc_solver_set_blocks: blocks do not cover the stage rows");
    }
    // launch order = decreasing work: the workgroups of the widest blocks start first instead of forming the launch's tail
    // (blocks are independent of one another; FC_BLK_SORT=0: tree order)
    static const bool blk_sort = [] { const char* e = std::getenv("FC_BLK_SORT"); return !(e && e[0] == '0'); }();
    if (blk_sort && stage_blk_count[s] > 1)
      std::stable_sort(packed.begin() + stage_blk_begin[s], packed.begin() + stage_blk_begin[s] + stage_blk_count[s],
                       [](const FcBlk& a, const FcBlk& b) { return (int64_t)a.nrows * (a.ni + a.nb) > (int64_t)b.nrows * (b.ni + b.nb); });
    st.blk_begin = stage_blk_begin[s];
    st.blk_count = stage_blk_count[s];
    st.blk_lpr = stage_lpr[s];
    // rows per row slot: the smallest power of two that covers the stage's tallest block
    int maxr = 1;
    for (int64_t q = stage_blk_begin[s]; q < stage_blk_begin[s] + stage_blk_count[s]; ++q) maxr = std::max(maxr, (int)blk_nrows[q]);
    const int slots = 256 / std::max(16, st.blk_lpr);
    int rps = 1;
    while (rps * slots < maxr) rps *= 2;
    st.blk_rps = rps;
    int max_row = 0;
    int64_t max_tile = 0;
    for (int64_t q = stage_blk_begin[s]; q < stage_blk_begin[s] + stage_blk_count[s]; ++q) {
      max_row = std::max(max_row, blk_ni[q] + blk_nb[q]);
      max_tile = std::max(max_tile, (int64_t)blk_nrows[q] * (blk_ni[q] + blk_nb[q]));
    }
    st.blk_flat = (stage_blk_count[s] > 0 && S.bits == 64) ? flat_loads(max_row, max_tile) : 0;
  }
  FCCHK(S.blk.upload(packed, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  return FC_OK;
}

int fc_factor_plan(fc_handle h, int32_t n_nodes, const int64_t* nodes, int32_t n_levels, const int64_t* level_ptr,
                   int64_t front_size, int64_t n_a, const int64_t* a_src, const int64_t* a_dst, const int64_t* a_ptr,
                   const int64_t* ext_off, int64_t n_ext, const int32_t* ext_p, int64_t n_ap, const int64_t* ap_src,
                   int32_t max_slots) {
  if (!h || n_nodes <= 0 || !nodes || n_levels <= 0 || !level_ptr || front_size <= 0 || n_a < 0 || !a_src || !a_dst || !a_ptr ||
      !ext_off || n_ext < 0 || !ext_p || n_ap <= 0 || !ap_src || max_slots < 1 || max_slots > 64)
    return fail(FC_ERR_INVALID, "fc_factor_plan: bad argument");
  HIPCHK(hipSetDevice(h->device));
  FCCHK(quiesce(h));
  if (level_ptr[0] != 0 || level_ptr[n_levels] != n_nodes || a_ptr[0] != 0 || a_ptr[n_levels] != n_a)
    return fail(FC_ERR_INVALID, "fc_factor_plan: level pointers do not cover the nodes / entries");
  h->have_plan = false;
  h->pnodes.assign((size_t)n_nodes, {});
  h->pmax_ni = 0;
  for (int g = 0; g < n_nodes; ++g) {
    const int64_t* r = nodes + (size_t)g * 7;  // level, front offset, nf, ni, value offset, parent, child slot
    fc_ctx::PlanNode& nd = h->pnodes[g];
    nd.level = (int)r[0];
    nd.front = r[1];
    nd.nf = (int)r[2];
    nd.ni = (int)r[3];
    nd.voff = r[4];
    nd.parent = (int)r[5];
    if (nd.nf < 0 || nd.ni < 0 || nd.ni > nd.nf || nd.front < 0 || nd.front + (int64_t)nd.nf * nd.nf > front_size ||
        nd.parent >= n_nodes || (nd.parent >= 0 && nd.parent <= g) || r[6] < 0 || r[6] >= max_slots || (nd.ni > 0 && nd.voff < 0))
      return fail(FC_ERR_INVALID, "fc_factor_plan: node table out of range");
    h->pmax_ni = std::max(h->pmax_ni, nd.ni);
  }
  for (int64_t k = 0; k < n_a; ++k)
    if (a_src[k] < 0 || a_src[k] >= h->nnz || a_dst[k] < 0 || a_dst[k] >= front_size)
      return fail(FC_ERR_INVALID, "fc_factor_plan: matrix scatter map out of range");
  for (int64_t k = 0; k < n_ap; ++k)
    if (ap_src[k] < 0 || ap_src[k] >= h->nnz) return fail(FC_ERR_INVALID, "fc_factor_plan: permuted-matrix map out of range");
  h->plevel_ptr.assign(level_ptr, level_ptr + n_levels + 1);
  h->pa_ptr.assign(a_ptr, a_ptr + n_levels + 1);
  h->pmax_slots = max_slots;
  // extend-add descriptors grouped by (level of the children, child slot)
  std::vector<FcExt> ext;
  h->pext_groups.assign((size_t)n_levels, std::vector<std::pair<int64_t, int>>((size_t)max_slots, {0, 0}));
  h->pext_maxnb.assign((size_t)n_levels, std::vector<int>((size_t)max_slots, 0));
  for (int li = 0; li < n_levels; ++li) {
    for (int sl = 0; sl < max_slots; ++sl) {
      const int64_t first = (int64_t)ext.size();
      int mx = 0;
      for (int64_t g = level_ptr[li]; g < level_ptr[li + 1]; ++g) {
        const fc_ctx::PlanNode& c = h->pnodes[(size_t)g];
        if (ext_off[g] < 0 || nodes[(size_t)g * 7 + 6] != sl) continue;
        const int nbc = c.nf - c.ni;
        if (c.parent < 0 || nbc <= 0 || ext_off[g] + nbc > n_ext) return fail(FC_ERR_INVALID, "fc_factor_plan: bad extend-add entry");
        const fc_ctx::PlanNode& par = h->pnodes[(size_t)c.parent];
        if (par.level != c.level - 1) return fail(FC_ERR_INVALID, "fc_factor_plan: parent is not one level up");
        for (int i = 0; i < nbc; ++i)
          if (ext_p[ext_off[g] + i] < 0 || ext_p[ext_off[g] + i] >= par.nf) return fail(FC_ERR_INVALID, "fc_factor_plan: extend-add position out of range");
        ext.push_back(FcExt{(long long)(c.front + (int64_t)c.ni * c.nf + c.ni), (long long)par.front, c.nf, par.nf, nbc, (int)ext_off[g]});
        mx = std::max(mx, nbc);
      }
      h->pext_groups[li][sl] = {first, (int)((int64_t)ext.size() - first)};
      h->pext_maxnb[li][sl] = mx;
    }
  }
  if (ext.empty()) ext.push_back(FcExt{0, 0, 0, 0, 0, 0});
  // ... and grouped by parent, children in slot order (one launch per level: fc_extend_add_parents)
  std::vector<FcExt> ext2;
  std::vector<FcExtPar> extpar;
  h->pextpar_groups.assign((size_t)n_levels, {0, 0});
  h->pextpar_maxnf.assign((size_t)n_levels, 0);
  for (int li = 0; li < n_levels; ++li) {
    std::map<int, std::vector<std::pair<int, FcExt>>> by_parent;  // parent node -> (slot, descriptor)
    for (int64_t g = level_ptr[li]; g < level_ptr[li + 1]; ++g) {
      const fc_ctx::PlanNode& c = h->pnodes[(size_t)g];
      const int nbc = c.nf - c.ni;
      if (ext_off[g] < 0 || c.parent < 0 || nbc <= 0) continue;
      const fc_ctx::PlanNode& par = h->pnodes[(size_t)c.parent];
      by_parent[c.parent].push_back({(int)nodes[(size_t)g * 7 + 6],
                                     FcExt{(long long)(c.front + (int64_t)c.ni * c.nf + c.ni), (long long)par.front, c.nf, par.nf, nbc, (int)ext_off[g]}});
    }
    const int64_t first = (int64_t)extpar.size();
    for (auto& kv : by_parent) {
      std::sort(kv.second.begin(), kv.second.end(), [](const auto& a, const auto& b) { return a.first < b.first; });
      extpar.push_back(FcExtPar{(int)ext2.size(), (int)kv.second.size()});
      for (const auto& e : kv.second) ext2.push_back(e.second);
      h->pextpar_maxnf[li] = std::max(h->pextpar_maxnf[li], h->pnodes[(size_t)kv.first].nf);
    }
    h->pextpar_groups[li] = {first, (int)((int64_t)extpar.size() - first)};
  }
  if (ext2.empty()) ext2.push_back(FcExt{0, 0, 0, 0, 0, 0});
  if (extpar.empty()) extpar.push_back(FcExtPar{0, 0});
  // fronts with a pivot block, level by level, with their scratch (fc_front.hip.h)
  std::vector<FcFront> fr;
  h->pfront_groups.assign((size_t)n_levels, {0, 0});
  h->plevel_max_ni.assign((size_t)n_levels, 0);
  h->plevel_max_nf.assign((size_t)n_levels, 0);
  h->plevel_fsize.assign((size_t)n_levels, 0);
  int64_t scratch_max = 1;
  for (int li = 0; li < n_levels; ++li) {
    const int64_t first = (int64_t)fr.size();
    int64_t off = 0;
    for (int64_t g = level_ptr[li]; g < level_ptr[li + 1]; ++g) {
      const fc_ctx::PlanNode& nd = h->pnodes[(size_t)g];
      if (nd.ni == 0) continue;
      h->plevel_max_ni[li] = std::max(h->plevel_max_ni[li], nd.ni);
      h->plevel_max_nf[li] = std::max(h->plevel_max_nf[li], nd.nf);
      h->plevel_fsize[li] += (int64_t)nd.nf * nd.nf;
    }
    // scratch of a front: W (KB x KB) then Cs (nf x KB) for the widest block step its level may take (128 columns on levels
    // whose largest front has order >= FC_FE_HUGE_MIN_NF, 64 elsewhere)
    const int64_t kbm = h->plevel_max_nf[li] >= FC_FE_HUGE_MIN_NF ? FC_FE_KH : FC_FE_KB_MAX;
    for (int64_t g = level_ptr[li]; g < level_ptr[li + 1]; ++g) {
      const fc_ctx::PlanNode& nd = h->pnodes[(size_t)g];
      if (nd.ni == 0) continue;
      fr.push_back(FcFront{(long long)nd.front, (long long)nd.voff, nd.nf, nd.ni, (long long)off});
      off += kbm * kbm + (int64_t)nd.nf * kbm;
    }
    scratch_max = std::max(scratch_max, off);
    h->pfront_groups[li] = {first, (int)((int64_t)fr.size() - first)};
    if (h->pfront_groups[li].second > 65535) return fail(FC_ERR_INVALID, "fc_factor_plan: more than 65535 fronts in one level");
  }
  h->pfront_total = (int64_t)fr.size();
  std::vector<FcExpItem> items;  // work list of the export: 16 rows of a front per workgroup
  for (size_t f = 0; f < fr.size(); ++f)
    for (int i0 = 0; i0 < fr[f].nf; i0 += 16) items.push_back(FcExpItem{(int)f, i0});
  h->pexp_n = (int64_t)items.size();
  if (items.empty()) items.push_back(FcExpItem{0, 0});
  FCCHK(h->pexp.upload(items, h->stream));
  if (fr.empty()) fr.push_back(FcFront{0, 0, 0, 0, 0});
  FCCHK(h->pscratch.alloc((size_t)scratch_max));
  FCCHK(h->pfront.upload(fr, h->stream));
  FCCHK(h->fronts.alloc((size_t)front_size));
  FCCHK(h->pa_src.upload(a_src, (size_t)std::max<int64_t>(1, n_a), h->stream));
  FCCHK(h->pa_dst.upload(a_dst, (size_t)std::max<int64_t>(1, n_a), h->stream));
  FCCHK(h->pap_src.upload(ap_src, (size_t)n_ap, h->stream));
  FCCHK(h->pext.upload(ext, h->stream));
  FCCHK(h->pext2.upload(ext2, h->stream));
  FCCHK(h->pextpar.upload(extpar, h->stream));
  FCCHK(h->pext_p.upload(ext_p, (size_t)std::max<int64_t>(1, n_ext), h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  h->pn_shift = 0;
  h->have_plan = true;
  return FC_OK;
}

int fc_set_front_shifts(fc_handle h, int32_t n, const int64_t* slots, const double* values) {
  if (!h || n < 0 || (n > 0 && (!slots || !values))) return fail(FC_ERR_INVALID, "fc_set_front_shifts: bad argument");
  if (!h->have_plan) return fail(FC_ERR_NOT_READY, "fc_factor_plan not called");
  HIPCHK(hipSetDevice(h->device));
  for (int k = 0; k < n; ++k)
    if (slots[k] < 0 || slots[k] >= (int64_t)h->fronts.n) return fail(FC_ERR_INVALID, "fc_set_front_shifts: slot outside the fronts");
  const int64_t no_slot = 0;
  const double no_value = 0.0;
  FCCHK(h->pshift_slot.upload(n > 0 ? slots : &no_slot, (size_t)std::max(1, n), h->stream));
  FCCHK(h->pshift_val.upload(n > 0 ? values : &no_value, (size_t)std::max(1, n), h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  h->pn_shift = n;
  return FC_OK;
}

int fc_set_root_rows(fc_handle h, int32_t first, int32_t count) {
  if (!h || first < -1 || (first >= 0 && count < 0)) return fail(FC_ERR_INVALID, "fc_set_root_rows: bad argument");
  h->root_x0 = first;
  h->root_xn = first >= 0 ? count : 0;
  return FC_OK;
}

int fc_refactor(fc_handle h, int slot, double* ms_out) {
  if (!h || slot < 0 || slot > 1) return fail(FC_ERR_INVALID, "fc_refactor: bad argument");
  if (!h->have_plan) return fail(FC_ERR_NOT_READY, "fc_factor_plan not called");
  OrderSys& S = h->sys[slot];
  if (!S.structured) return fail(FC_ERR_NOT_READY, "fc_solver_setup (structure) must be called first");
  if ((int64_t)h->pap_src.n != S.Ap_nnz) return fail(FC_ERR_INVALID, "fc_refactor: plan and solver structure disagree (matrix)");
  HIPCHK(hipSetDevice(h->device));
  FCCHK(quiesce(h));
  h->refactor_flops = h->refactor_flops_full = 0.0;
  for (size_t g = 0; g < h->pnodes.size(); ++g) {
    const fc_ctx::PlanNode& nd = h->pnodes[g];
    const int nb = nd.nf - nd.ni;
    const bool root_block = h->root_x0 >= 0 && g + 1 == h->pnodes.size();
    if (root_block && (nd.parent >= 0 || nb != 0 || h->root_x0 + h->root_xn > nd.ni))
      return fail(FC_ERR_INVALID, "fc_refactor: fc_set_root_rows needs a root without boundary and rows inside its pivot block");
    const int64_t rows = root_block ? h->root_xn : nd.ni;
    if (nd.ni > 0 && nd.voff + rows * nd.nf + (int64_t)nb * nd.ni > S.f_nnz)
      return fail(FC_ERR_INVALID, "fc_refactor: plan and solver structure disagree (factor values)");
  }
  HIPCHK(hipEventRecord(h->ev0, h->stream));
  const double* av = h->vals[slot].p;
  double* F = h->fronts.p;
  double* fv = S.f_val.p;
  HIPCHK(hipMemsetAsync(F, 0, h->fronts.n * sizeof(double), h->stream));
  const int n_levels = (int)h->plevel_ptr.size() - 1;
  // multi-GPU: this rank's plan holds its own sub-tree and the root (fc_factor_plan built with keep=): the root front is
  // the sum over the ranks of the sub-trees' Schur complements plus the matrix entries, which only the lead rank scatters
  const bool dist = h->partitioned && exchanges(h) && h->nranks > 1;
  int64_t n_a = h->pa_ptr.back();
  if (dist && !h->lead) n_a = h->pa_ptr[(size_t)n_levels - 1];  // entries below the root level
  // matrix entries -> fronts, and (same launch) the permuted copy of the matrix for the residual monitor
  hipLaunchKernelGGL(fc_front_scatter, dim3(nblocks(n_a + S.Ap_nnz, 256)), dim3(256), 0, h->stream, n_a, h->pa_src.p, h->pa_dst.p, av, F,
                     (int64_t)S.Ap_nnz, h->pap_src.p, S.Ap_val.p);
  if (h->pn_shift > 0) {
    int64_t skip0 = 0, skip1 = 0;  // the root front is summed over the ranks: only the lead rank shifts there
    if (dist && !h->lead) {
      const fc_ctx::PlanNode& root = h->pnodes[(size_t)h->plevel_ptr[(size_t)n_levels - 1]];
      skip0 = root.front;
      skip1 = root.front + (int64_t)root.nf * root.nf;
    }
    hipLaunchKernelGGL(fc_front_shift, dim3(nblocks(h->pn_shift, 64)), dim3(64), 0, h->stream, h->pn_shift, h->pshift_slot.p,
                       h->pshift_val.p, F, skip0, skip1);
  }
  for (int li = 0; li < n_levels; ++li) {
    static const bool ext_by_slot = [] { const char* e = std::getenv("FC_EXTEND_BY_SLOT"); return e && e[0] == '1'; }();  // A/B reference
    if (li > 0 && !ext_by_slot) {
      // update blocks of the level below: one launch, a workgroup per 16 rows of a parent front, children in slot order
      const auto pg = h->pextpar_groups[li - 1];
      int rows = FC_EXTP_ROWS;  // fewer rows per workgroup on the levels of few, large parents: at least ~2 000 workgroups
      while (rows > 4 && (int64_t)pg.second * ((h->pextpar_maxnf[li - 1] + rows - 1) / rows) < 2048) rows /= 2;
      const int gx = (h->pextpar_maxnf[li - 1] + rows - 1) / rows;
      for (int c0 = 0; c0 < pg.second; c0 += 65535) {  // grid.y limit
        const int nc_ = std::min(65535, pg.second - c0);
        hipLaunchKernelGGL(fc_extend_add_parents, dim3(gx, nc_), dim3(256), 0, h->stream, h->pextpar.p + pg.first + c0, h->pext2.p, h->pext_p.p, F, rows);
      }
    } else if (li > 0) {
      // update blocks of the level below, one launch per child slot (deterministic, conflict-free)
      for (int sl = 0; sl < h->pmax_slots; ++sl) {
        const auto grp = h->pext_groups[li - 1][sl];
        if (grp.second == 0) continue;
        const int gx = (h->pext_maxnb[li - 1][sl] + FC_EXT_ROWS - 1) / FC_EXT_ROWS;
        for (int c0 = 0; c0 < grp.second; c0 += 65535) {  // grid.y limit
          const int nc_ = std::min(65535, grp.second - c0);
          hipLaunchKernelGGL(fc_extend_add, dim3(gx, nc_), dim3(256), 0, h->stream, h->pext.p + grp.first + c0, h->pext_p.p, F);
        }
      }
    }
    if (dist && li == n_levels - 1) {
      // the one exchange of a numeric factorisation: sum the root front over the ranks
      if (h->plevel_ptr[li + 1] - h->plevel_ptr[li] != 1) return fail(FC_ERR_INVALID, "fc_refactor: a partitioned plan needs a single root node");
      const fc_ctx::PlanNode& root = h->pnodes[(size_t)h->plevel_ptr[li]];
      FCCHK(exchange(h, F + root.front, (size_t)root.nf * root.nf));
    }
    // all fronts of the level together: block steps of FC_FE_KB pivot columns (fc_front.hip.h)
    const auto grp = h->pfront_groups[li];
    if (grp.second > 0) {
      const FcFront* fp = h->pfront.p + grp.first;
      const int nfmax = h->plevel_max_nf[li];
      const int ct = (nfmax + 63) / 64;  // 64-wide tiles per side of the widest front
      // block step of the level: wide fronts take 64 pivot columns at a time, small ones 32 (fc_front.hip.h)
      // FC_FE_WIDE_NF (environment, read per factorisation): tests run the 64-column kernels on small meshes with it
      int wide_nf = FC_FE_WIDE_NF;
      if (const char* e = std::getenv("FC_FE_WIDE_NF")) wide_nf = std::max(1, std::atoi(e));
      // 128-column steps where the update is bound by the traffic of the fronts: very wide fronts, or a level whose fronts together
      // exceed what the caches hold (many mid-size fronts: the dependent chain of a step is hidden behind the other fronts)
      int huge_nf = FC_FE_HUGE_NF;
      double huge_mb = FC_FE_HUGE_MB;
      if (const char* e = std::getenv("FC_FE_HUGE_NF")) huge_nf = std::atoi(e);
      if (const char* e = std::getenv("FC_FE_HUGE_MB")) huge_mb = std::atof(e);
      bool huge = !h->huge_refused && h->plevel_max_nf[li] >= FC_FE_HUGE_MIN_NF &&  // (the scratch of smaller levels is not sized for it)
                  (h->plevel_max_nf[li] >= huge_nf || (double)h->plevel_fsize[li] * 8e-6 >= huge_mb);
      if (huge && !h->huge_lds_ok) {  // 148 KB of LDS per workgroup: above the 64 KB a kernel gets without asking
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(fc_fe_pivot_huge), hipFuncAttributeMaxDynamicSharedMemorySize, FC_FE_KH_LDS_BYTES) == hipSuccess) {
          h->huge_lds_ok = true;
        } else {  // a runtime that does not grant it: the level takes 64- / 32-column steps (slower, same result)
          (void)hipGetLastError();
          h->huge_refused = true;
          huge = false;
        }
      }
      const bool wide = !huge && h->plevel_max_nf[li] >= wide_nf;
      const int kbs = huge ? FC_FE_KH : (wide ? FC_FE_KB_WIDE : FC_FE_KB);
      const int steps = (h->plevel_max_ni[li] + kbs - 1) / kbs;
      // the root front of a multi-GPU layout: this handle exports the pivot rows [keep0, keep1) only; the eliminated rows outside them are dead
      // (fc_fe_dead_rows).  FC_ROOT_SKIP=0: the whole front is swept, as every rank did up to round 3 (A/B, and the flop count's reference)
      static const bool root_skip = [] { const char* e = std::getenv("FC_ROOT_SKIP"); return !(e && e[0] == '0'); }();
      const bool root_level = dist && li == n_levels - 1 && h->root_x0 >= 0;
      const int keep0 = (root_level && root_skip) ? h->root_x0 : 0, keep1 = (root_level && root_skip) ? h->root_x0 + h->root_xn : INT_MAX;
      // bookkeeping: flops of the trailing updates (2 kb rows nf per block step and front; the widest front stands for the level) with and
      // without the dead-row skip -- what a rank pays for the replicated root against what it paid
      {
        const int nfm = h->plevel_max_nf[li];
        for (int k = 0; k < steps; ++k) {
          const int k0 = k * kbs;
          int alive = 0;
          for (int i0 = 0; i0 < nfm; i0 += 64) alive += (i0 + 64 <= k0 && (i0 + 64 <= keep0 || i0 >= keep1)) ? 0 : std::min(64, nfm - i0);
          const double per_row = 2.0 * (double)std::min(kbs, std::max(0, h->plevel_max_ni[li] - k0)) * (double)nfm * (double)grp.second;
          h->refactor_flops += per_row * alive;
          h->refactor_flops_full += per_row * nfm;
        }
      }
      for (int k = 0; k < steps; ++k) {
        if (huge) {
          // (a two-stream look-ahead -- the next pivot block's four tiles first, then its inversion beside the rest of the update -- was
          // measured: the streams do not overlap on this runtime, 70.5 -> 69.5 ms on cavity_fine, slower on O1; not kept)
          if (k == 0) hipLaunchKernelGGL(fc_fe_pivot_huge, dim3(grp.second), dim3(256), FC_FE_KH_LDS_BYTES, h->stream, fp, F, h->pscratch.p, k);
          hipLaunchKernelGGL(fc_fe_panels_huge, dim3(2 * ct, grp.second), dim3(256), FC_FE_KH_PANEL_LDS_BYTES, h->stream, fp, F, h->pscratch.p, k, ct, keep0, keep1);
          hipLaunchKernelGGL(fc_fe_update_huge, dim3(ct * ct, grp.second), dim3(256), 0, h->stream, fp, F, h->pscratch.p, k, ct, keep0, keep1);
          if (k + 1 < steps) hipLaunchKernelGGL(fc_fe_pivot_huge, dim3(grp.second), dim3(256), FC_FE_KH_LDS_BYTES, h->stream, fp, F, h->pscratch.p, k + 1);
        } else if (wide) {
          if (k == 0) hipLaunchKernelGGL(fc_fe_pivot<FC_FE_KB_WIDE>, dim3(grp.second), dim3(256), 0, h->stream, fp, F, h->pscratch.p, k);
          hipLaunchKernelGGL(fc_fe_panels<FC_FE_KB_WIDE>, dim3(2 * ct, grp.second), dim3(256), 0, h->stream, fp, F, h->pscratch.p, k, ct, keep0, keep1);
          hipLaunchKernelGGL(fc_fe_update<FC_FE_KB_WIDE>, dim3(ct * ct + 1, grp.second), dim3(256), 0, h->stream, fp, F, h->pscratch.p, k, ct, keep0, keep1);
        } else {
          if (k == 0) hipLaunchKernelGGL(fc_fe_pivot<FC_FE_KB>, dim3(grp.second), dim3(256), 0, h->stream, fp, F, h->pscratch.p, k);
          hipLaunchKernelGGL(fc_fe_panels<FC_FE_KB>, dim3(2 * ct, grp.second), dim3(256), 0, h->stream, fp, F, h->pscratch.p, k, ct, keep0, keep1);
          hipLaunchKernelGGL(fc_fe_update<FC_FE_KB>, dim3(ct * ct + 1, grp.second), dim3(256), 0, h->stream, fp, F, h->pscratch.p, k, ct, keep0, keep1);
        }
      }
      HIPCHK(hipGetLastError());
    }
  }
  {
    // all fronts -> the layout the sweeps read, in the storage type of the slot (the root of a multi-GPU layout: only this handle's
    // block of pivot rows has storage)
    const int n_items = (int)h->pexp_n;
    const int root_front = h->root_x0 >= 0 ? (int)h->pfront_total - 1 : -1;
    const int xr0 = h->root_x0 >= 0 ? h->root_x0 : 0, xr1 = h->root_x0 >= 0 ? h->root_x0 + h->root_xn : INT_MAX;
    if (n_items > 0) {
      if (S.bits == 64)
        hipLaunchKernelGGL(fc_fe_export<double>, dim3(n_items), dim3(256), 0, h->stream, h->pfront.p, h->pexp.p, F, fv, root_front, xr0, xr1);
      else if (S.bits == 32)
        hipLaunchKernelGGL(fc_fe_export<float>, dim3(n_items), dim3(256), 0, h->stream, h->pfront.p, h->pexp.p, F, S.f_val32.p, root_front, xr0, xr1);
      else
        hipLaunchKernelGGL(fc_fe_export<FcBf16>, dim3(n_items), dim3(256), 0, h->stream, h->pfront.p, h->pexp.p, F, S.f_val16.p, root_front, xr0, xr1);
      HIPCHK(hipGetLastError());
    }
  }
  HIPCHK(hipEventRecord(h->ev1, h->stream));
  HIPCHK(hipEventSynchronize(h->ev1));
  float ms = 0.f;
  HIPCHK(hipEventElapsedTime(&ms, h->ev0, h->ev1));
  if (ms_out) *ms_out = (double)ms;
  h->refactor_ms[slot] = (double)ms;
  S.ready = true;
  S.inexact = false;  // (decided by the acceptance solve of the caller: fc_setup_solver, fc_accept_factors)
  return batch_repack(h, slot);  // the batched block kernel streams its own (tiled) copy of the values
}

// trailing-update flops of the handle's last fc_refactor (a level is priced by its widest front): as run, and what they would be with
// every row of the multi-GPU root front swept (FC_ROOT_SKIP=0, the scheme up to round 3)
int fc_get_refactor_flops(fc_handle h, double* run, double* full) {
  if (!h) return fail(FC_ERR_INVALID, "null handle");
  if (run) *run = h->refactor_flops;
  if (full) *full = h->refactor_flops_full;
  return FC_OK;
}

int fc_get_refactor_ms(fc_handle h, int slot, double* ms) {
  if (!h || slot < 0 || slot > 1 || !ms) return fail(FC_ERR_INVALID, "fc_get_refactor_ms: bad argument");
  *ms = h->refactor_ms[slot];
  return FC_OK;
}

// ── the whole solver setup behind the C ABI: symbolic analysis (fc_symbolic.hpp) + uploads + numeric factorisation ──
static int upload_energy_matrix(fc_ctx* h) {
  // (u, v) mass matrix in the permuted numbering for the un-fused energy evaluation (fc_finish)
  FCCHK(fc_assemble_matrix(h, FC_SLOT_MASS, 1.0, 0.0, nullptr, 1.0, nullptr, 1.0, 0.0, 0.0));
  std::vector<double> mv((size_t)h->nnz);
  HIPCHK(hipMemcpy(mv.data(), h->vals[FC_SLOT_MASS].p, mv.size() * sizeof(double), hipMemcpyDeviceToHost));
  const int N = h->N, nn2 = 2 * h->nn;
  const std::vector<int>& perm = h->h_perm;  // (the permutation the handle holds: fc_setup_solver's tree or fc_setup_krylov's)
  std::vector<int> iperm((size_t)N);
  for (int i = 0; i < N; ++i) iperm[(size_t)perm[(size_t)i]] = i;
  std::vector<int> rp((size_t)N + 1, 0), col;
  std::vector<double> val;
  std::vector<std::pair<int, double>> row;
  for (int i = 0; i < N; ++i) {
    const int r = perm[i];
    row.clear();
    if (r < nn2)
      for (int k = h->h_rowptr[r]; k < h->h_rowptr[r + 1]; ++k)
        if (h->h_col[k] < nn2 && mv[(size_t)k] != 0.0) row.emplace_back(iperm[h->h_col[k]], mv[(size_t)k]);
    std::sort(row.begin(), row.end());
    for (auto& e : row) {
      col.push_back(e.first);
      val.push_back(e.second);
    }
    rp[(size_t)i + 1] = (int)col.size();
  }
  if (col.empty()) {
    col.push_back(0);
    val.push_back(0.0);
  }
  return fc_set_energy_matrix(h, rp.data(), col.data(), val.data());
}

// ndsolver.schur_diagonal_scaling in the permuted numbering (truncated factors)
static int upload_stage_diag(fc_ctx* h, int slot) {
  std::vector<double> av((size_t)h->nnz);
  HIPCHK(hipMemcpy(av.data(), h->vals[slot].p, av.size() * sizeof(double), hipMemcpyDeviceToHost));
  const int N = h->N, nn2 = 2 * h->nn;
  std::vector<double> diag((size_t)N, 0.0), out((size_t)N, 1.0);
  auto entry = [&](int r, int c) -> double {
    const int* b0 = h->h_col.data() + h->h_rowptr[r];
    const int* b1 = h->h_col.data() + h->h_rowptr[r + 1];
    const int* it = std::lower_bound(b0, b1, c);
    return (it != b1 && *it == c) ? av[(size_t)(it - h->h_col.data())] : 0.0;
  };
  for (int r = 0; r < N; ++r) diag[r] = entry(r, r);
  for (int r = 0; r < nn2; ++r) out[r] = 1.0 / (diag[r] != 0.0 ? diag[r] : 1.0);
  for (int r = nn2; r < N; ++r) {
    double sacc = 0.0;
    for (int k = h->h_rowptr[r]; k < h->h_rowptr[r + 1]; ++k) {
      const int j = h->h_col[k];
      if (j < nn2) sacc += av[(size_t)k] * entry(j, r) / (diag[j] != 0.0 ? diag[j] : 1.0);
    }
    sacc = -sacc;
    out[r] = sacc != 0.0 ? 1.0 / sacc : 1.0;
  }
  std::vector<double> dp((size_t)N);
  for (int i = 0; i < N; ++i) dp[i] = out[(size_t)h->sym_tree.perm[i]];
  return fc_set_stage_diag(h, slot, dp.data());
}

// diagonal entry of the pinned dof in the front of the plan node that eliminates it (ndsolver.front_diagonal_slot)
static int apply_pressure_pin(fc_ctx* h) {
  if (!h->sym_ready || !h->have_plan) return FC_OK;
  if (h->pin_dof < 0) return fc_set_front_shifts(h, 0, nullptr, nullptr);
  const int ip = h->sym_tree.iperm[(size_t)h->pin_dof];
  const fcsym::Plan& pl = h->sym_plan;
  for (size_t g = 0; g < pl.node_i0.size(); ++g) {
    const int64_t i0 = pl.node_i0[g], nf = pl.nodes[g * 7 + 2], ni = pl.nodes[g * 7 + 3];
    if (ip >= i0 && ip < i0 + ni) {
      const int64_t slot = pl.nodes[g * 7 + 1] + (ip - i0) * (nf + 1);
      return fc_set_front_shifts(h, 1, &slot, &h->pin_shift);
    }
  }
  return fc_set_front_shifts(h, 0, nullptr, nullptr);  // eliminated by another rank
}

int fc_set_pressure_pin(fc_handle h, int32_t dof, double shift) {
  if (!h || dof >= h->N || (dof >= 0 && dof < 2 * h->nn)) return fail(FC_ERR_INVALID, "fc_set_pressure_pin: the pinned dof must be a pressure dof (or -1)");
  h->pin_dof = dof;
  h->pin_shift = shift;
  return apply_pressure_pin(h);
}

int fc_get_partition_info(fc_handle h, int32_t* out) {
  if (!h || !out) return fail(FC_ERR_INVALID, "fc_get_partition_info: null argument");
  out[0] = h->partitioned ? h->ncl : h->nc;                      // cells whose right-hand side / energy this rank computes
  out[1] = h->partitioned && h->n_asm > 0 ? h->n_asm : h->nc;    // cells whose element matrices it assembles (own + along the separator)
  out[2] = h->partitioned ? (h->lead ? 1 : 0) : 1;
  out[3] = (h->comm || h->host_xchg) ? h->nranks : 1;
  return FC_OK;
}

int fc_get_local_cells(fc_handle h, int32_t* cells) {
  if (!h || !cells) return fail(FC_ERR_INVALID, "fc_get_local_cells: null argument");
  if (h->partitioned)
    std::copy(h->h_cell_list.begin(), h->h_cell_list.end(), cells);
  else
    for (int c = 0; c < h->nc; ++c) cells[c] = c;
  return FC_OK;
}

int fc_accept_factors(fc_handle h, int slot, double* residual_out, int32_t* inexact_out);

// (shape of the default elimination tree: fcsym::default_bits, mirrored by flowcontrol_amd/ndsolver.py::default_bits)

int fc_setup_solver(fc_handle h, int slot, int32_t depth, int32_t merge, int32_t truncate, int32_t refine, int32_t check_residual) {
  if (!h || slot < 0 || slot > 1 || merge < 1 || merge > 4 || depth < 0 || truncate < 0 || refine < 0 || check_residual < -1)
    return fail(FC_ERR_INVALID, "fc_setup_solver: bad argument");
  if (!h->slot_ok[slot] || !h->sys[slot].have_lift) return fail(FC_ERR_NOT_READY, "fc_setup_solver: assemble the slot and call fc_apply_bc first");
  HIPCHK(hipSetDevice(h->device));
  const int N = h->N;
  const int world = (h->comm || h->host_xchg) ? h->nranks : 1, rank = world > 1 ? h->rank : 0;
  int top = 0;
  while ((1 << top) < world) ++top;
  if ((1 << top) != world) return fail(FC_ERR_INVALID, "fc_setup_solver: the number of ranks must be a power of two");
  try {
    fcsym::Keep keep;
    if (world > 1) {
      const std::vector<int>* cum = &h->sym_tree.cum;
      keep = [cum, rank, top](int k, int n) { return k == 0 || (n >> ((*cum)[k] - top)) == rank; };
    }
    if (truncate > 0) {
      if (world > 1) return fail(FC_ERR_INVALID, "fc_setup_solver: truncated factors need a single-GPU handle");
      keep = [truncate](int k, int) { return k >= truncate; };
    }
    if (!h->sym_ready) {
      const std::vector<int> bits = depth == 0 ? fcsym::default_bits(h->nc, merge, top) : fcsym::uniform_bits(depth, merge, top);
      h->sym_bits = bits;
      std::vector<unsigned char> skip((size_t)N, 0);
      for (int k = 0; k < h->n_bc; ++k) skip[(size_t)h->h_bc_dofs[k]] = 1;
      h->sym_tree = fcsym::build_tree(h->h_cell_dofs, 15, h->h_cent, h->nc, N, bits, &skip, top);
      if (truncate > h->sym_tree.depth) return fail(FC_ERR_INVALID, "fc_setup_solver: truncate exceeds the tree depth");
      FCCHK(fc_set_permutation(h, h->sym_tree.perm.data()));
      FCCHK(upload_energy_matrix(h));
      if (world > 1) {  // of the root's pivot-block inverse a rank stores the rows it applies
        const auto rr = fcsym::root_row_block(h->sym_tree, rank, world);
        h->sym_fac = fcsym::layout_factors(h->sym_tree, keep, rr.first, rr.second);
        FCCHK(fc_set_root_rows(h, (int)(rr.first - h->sym_tree.node_ptr[0].front()), (int)(rr.second - rr.first)));
      } else {
        h->sym_fac = fcsym::layout_factors(h->sym_tree, keep);
        FCCHK(fc_set_root_rows(h, -1, 0));
      }
      if (truncate > 0)
        for (int k = 0; k < truncate; ++k) h->sym_fac.stage_kind[(size_t)h->sym_tree.depth + k] = 2;
      h->sym_plan = fcsym::factor_plan(h->sym_tree, h->sym_fac, h->h_rowptr, h->h_col, &skip, keep);
      h->sym_total_nnz = (world > 1 || truncate > 0) ? fcsym::layout_factors(h->sym_tree, nullptr).nnz : h->sym_fac.nnz;
      const fcsym::Plan& pl = h->sym_plan;
      const int64_t zero64 = 0;
      FCCHK(fc_factor_plan(h, (int)(pl.nodes.size() / 7), pl.nodes.data(), (int)pl.level_ptr.size() - 1, pl.level_ptr.data(), pl.front_size,
                           (int64_t)pl.a_src.size(), pl.a_src.empty() ? &zero64 : pl.a_src.data(), pl.a_dst.empty() ? &zero64 : pl.a_dst.data(),
                           pl.a_ptr.data(), pl.ext_off.data(), (int64_t)pl.ext_p.size(), pl.ext_p.data(), (int64_t)pl.ap_src.size(),
                           pl.ap_src.data(), pl.max_slots));
      h->sym_truncate = truncate;
      h->sym_ready = true;
      h->upc.ready = h->upc.tried = false;
      FCCHK(apply_pressure_pin(h));
    } else if (truncate != h->sym_truncate) {
      return fail(FC_ERR_INVALID, "fc_setup_solver: both slots of a handle share one tree: same truncate");
    }
    OrderSys& S = h->sys[slot];
    if (!S.structured) {
      const fcsym::Tree& t = h->sym_tree;
      const fcsym::Factors& fac = h->sym_fac;
      fcsym::Partition part = fcsym::partition(t, fac, rank, world);
      // a ONE-rank communicator (test aid, FC_FORCE_COMM=1 on the Python side: fc_comm_init with nranks = 1): everything is owned and the
      // root's rows are "shared" with nobody, but the partitioned code path -- cell list, row kinds, both in-stream all-reduces of the
      // apply, the tail record's -- runs as it does on several GPUs
      const bool one_rank_comm = world == 1 && h->comm != nullptr && truncate == 0;
      if (one_rank_comm) {
        const int64_t root0 = t.node_ptr[0].front(), root1 = t.node_ptr[0].back();
        for (int64_t i = root0; i < root1; ++i) part.rowkind[(size_t)t.perm[(size_t)i]] = 2;
        part.ar_stage = t.depth - 1, part.ar_row0 = (int)root0, part.ar_n = (int)(root1 - root0), part.ar2_stage = t.depth;
      }
      if ((world > 1 || one_rank_comm) && !h->partitioned)
        FCCHK(fc_set_partition(h, (int)part.local_cells.size(), part.local_cells.data(), part.rowkind.data(), rank == 0 ? 1 : 0));
      const int zero32 = 0;
      const int64_t zero64 = 0;
      FCCHK(fc_solver_setup(h, slot, h->sym_plan.Ap_rowptr.data(), h->sym_plan.Ap_col.data(), nullptr, (int)part.stage_kind.size(),
                            part.stage_begin.data(), part.stage_row0.data(), part.stage_nrows.data(), part.stage_kind.data(),
                            part.seg_ptr.data(), (int64_t)part.seg_val.size(), part.seg_val.empty() ? &zero64 : part.seg_val.data(),
                            part.seg_col.empty() ? &zero32 : part.seg_col.data(), part.seg_len.empty() ? &zero32 : part.seg_len.data(),
                            (int64_t)fac.idx.size(), fac.idx.empty() ? &zero32 : fac.idx.data(), std::max<int64_t>(1, fac.n_val), nullptr,
                            part.ar_stage, part.ar_row0, part.ar_n, part.ar2_stage));
      int64_t local = 0;
      for (int v : part.seg_len) local += v;
      h->sym_local_values[slot] = local;
      // (tuning / test aids: FC_BLOCK_KERNEL=0 keeps every down stage on the segment kernel, FC_BLOCK_TARGET / FC_BLOCK_MIN move the
      //  rows-per-workgroup choice of the LDS-tiled block kernel through all of its instantiations)
      const char* eb = std::getenv("FC_BLOCK_KERNEL");
      const char* et = std::getenv("FC_BLOCK_TARGET");
      const char* em = std::getenv("FC_BLOCK_MIN");
      fcsym::Blocks B = fcsym::down_blocks(t, fac, rank, world, 32, et ? std::max(1, std::atoi(et)) : block_target(N), em ? std::max(1, std::atoi(em)) : 512);
      if (eb && eb[0] == '0') std::fill(B.count.begin(), B.count.end(), 0);
      retile_flat(B);
      FCCHK(fc_solver_set_blocks(h, slot, (int)B.begin.size(), B.begin.data(), B.count.data(), B.lpr.data(), (int64_t)B.val.size(),
                                 B.val.empty() ? &zero64 : B.val.data(), B.row0.empty() ? &zero32 : B.row0.data(),
                                 B.nrows.empty() ? &zero32 : B.nrows.data(), B.i0.empty() ? &zero32 : B.i0.data(),
                                 B.ni.empty() ? &zero32 : B.ni.data(), B.idx.empty() ? &zero32 : B.idx.data(),
                                 B.nb.empty() ? &zero32 : B.nb.data(), (int64_t)fac.idx.size(), std::max<int64_t>(1, fac.n_val)));
    }
    if (truncate > 0) FCCHK(upload_stage_diag(h, slot));
  } catch (const std::exception& e) {
    return fail(FC_ERR_INVALID, std::string("fc_setup_solver: ") + e.what());
  }
  FCCHK(fc_refactor(h, slot, nullptr));
  // end-to-end acceptance of the new factors: one solve with a fixed right-hand side, residual against the matrix itself
  // (partitioned: the probe is a collective, every rank calls fc_setup_solver; truncated factors are a preconditioner:
  // nothing to probe)
  if (truncate == 0 && h->sys[slot].bits != 64) {
    // compressed factors: the acceptance solve goes through GMRES (they are a preconditioner): it must converge in a
    // handful of iterations, or the rounded factors are no good for this operator
    if (world > 1 && !exchanges(h)) return fail(FC_ERR_INVALID, "fc_setup_solver: compressed factors on a partitioned handle need its exchange first (the acceptance solve is a collective GMRES)");
    std::vector<double> b((size_t)N), x((size_t)N);
    for (int i = 0; i < N; ++i) b[i] = std::cos(0.37 * i + 0.1);
    if (h->pin_dof >= 0)
      for (int i = 2 * h->nn; i < N; ++i) b[i] = 0.0;
    double info[4];
    FCCHK(fc_set_solver_options(h, FC_METHOD_GMRES, 60, 1e-10, 1));
    FCCHK(fc_solve(h, slot, b.data(), x.data(), info));
    if (!(info[1] < 1e-8) || info[0] > 40)
      return fail(FC_ERR_HIP, "fc_setup_solver: GMRES on the compressed factors needed " + std::to_string((int)info[0]) + " iterations (residual " + std::to_string(info[1]) + ")");
    return FC_OK;  // the caller chooses the Krylov method and its tolerances (fc_set_solver_options)
  }
  if (truncate == 0 && (world == 1 ? (!h->partitioned || h->comm != nullptr) : exchanges(h))) {  // (a one-rank communicator probes like a single GPU)
    std::vector<double> b((size_t)N), x((size_t)N);
    for (int i = 0; i < N; ++i) b[i] = std::cos(0.37 * i + 0.1);
    // enclosed flow: a right-hand side compatible with the constant-pressure null space.  Decided by the GLOBAL pin
    // (h->pin_dof), not by pn_shift: on a partitioned handle only the rank that eliminates the pinned dof carries a
    // shift, and the ranks of a collective fc_solve must all build the same right-hand side.
    if (h->pin_dof >= 0)
      for (int i = 2 * h->nn; i < N; ++i) b[i] = 0.0;
    (void)b, (void)x;
    FCCHK(fc_accept_factors(h, slot, nullptr, nullptr));
  }
  return fc_set_solver_options(h, FC_METHOD_REFINE, refine, 1e-10, check_residual);
}

// Factorisation-free solver setup of a slot (include/fc_hip.h): permutation from the nested-dissection tree alone (no factor layout,
// no elimination plan, no fronts), permuted system matrix for the Krylov mat-vec, and the SIMPLE / AMG preconditioner of fc_precond.hpp
// built on the host from the slot's assembled values.  Device memory: the matrix twice (W order + permuted), its velocity block and
// B / Bt once more in compact numbering, an AMG hierarchy of ~1.3 x nnz(S) -- O(nnz), nothing that grows like the fill.
int fc_setup_krylov(fc_handle h, int slot, int32_t sweeps, int method, int32_t max_iter, double rtol, int32_t check_residual) {
  if (!h || slot < 0 || slot > 1 || sweeps < 1 || sweeps > 16) return fail(FC_ERR_INVALID, "fc_setup_krylov: bad argument");
  if (method != FC_METHOD_GMRES && method != FC_METHOD_BICGSTAB) return fail(FC_ERR_INVALID, "fc_setup_krylov: method must be FC_METHOD_GMRES or FC_METHOD_BICGSTAB");
  if (!h->slot_ok[slot] || !h->sys[slot].have_lift) return fail(FC_ERR_NOT_READY, "fc_setup_krylov: assemble the slot and call fc_apply_bc first");
  if (h->partitioned || h->comm || h->host_xchg) return fail(FC_ERR_INVALID, "fc_setup_krylov: the factorisation-free preconditioner runs on a single-GPU handle");
  HIPCHK(hipSetDevice(h->device));
  const auto t0 = std::chrono::steady_clock::now();
  const int N = h->N;
  OrderSys& S = h->sys[slot];
  try {
    if (!h->have_perm) {
      // the tree is used for its ordering only (sub-domain by sub-domain: rows that share columns sit next to each other)
      std::vector<unsigned char> skip((size_t)N, 0);
      for (int k = 0; k < h->n_bc; ++k) skip[(size_t)h->h_bc_dofs[k]] = 1;
      const fcsym::Tree t = fcsym::build_tree(h->h_cell_dofs, 15, h->h_cent, h->nc, N, fcsym::default_bits(h->nc, 2, 0), &skip, 0);
      FCCHK(fc_set_permutation(h, t.perm.data()));
      FCCHK(upload_energy_matrix(h));
    }
    FCCHK(quiesce(h));
    // permuted pattern + where each of its entries comes from
    if (!S.factor_free || S.ap_src.n != (size_t)h->nnz) {
      std::vector<int> ip((size_t)N);
      for (int i = 0; i < N; ++i) ip[(size_t)h->h_perm[(size_t)i]] = i;
      std::vector<int> rp((size_t)N + 1, 0), ci((size_t)h->nnz);
      std::vector<int64_t> src((size_t)h->nnz);
      std::vector<std::pair<int, int>> row;
      for (int i = 0; i < N; ++i) {
        const int w = h->h_perm[(size_t)i];
        row.clear();
        for (int k = h->h_rowptr[(size_t)w]; k < h->h_rowptr[(size_t)w + 1]; ++k) row.emplace_back(ip[(size_t)h->h_col[(size_t)k]], k);
        std::sort(row.begin(), row.end());
        int q = rp[(size_t)i];
        for (const auto& e : row) ci[(size_t)q] = e.first, src[(size_t)q] = e.second, ++q;
        rp[(size_t)i + 1] = q;
      }
      S.Ap_nnz = h->nnz;
      FCCHK(S.Ap_rowptr.upload(rp, h->stream));
      FCCHK(S.Ap_col.upload(ci, h->stream));
      FCCHK(S.Ap_val.alloc((size_t)h->nnz));
      FCCHK(S.ap_src.upload(src, h->stream));
    }
    hipLaunchKernelGGL(fc_gather64, dim3(nblocks(S.Ap_nnz, 256)), dim3(256), 0, h->stream, S.Ap_nnz, S.ap_src.p, h->vals[slot].p, S.Ap_val.p);
    std::vector<double> vals((size_t)h->nnz);
    HIPCHK(hipMemcpyAsync(vals.data(), h->vals[slot].p, (size_t)h->nnz * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    // host: blocks, Schur complement, AMG hierarchy
    fcpc::Blocks X = fcpc::split_blocks(N, 2 * h->nn, h->h_rowptr, h->h_col, vals.data(), h->h_perm);
    std::vector<double> dinv((size_t)X.nu), wdinv((size_t)X.nu);
    for (int i = 0; i < X.nu; ++i) {
      if (!(std::fabs(X.dF[(size_t)i]) > 0.0)) return fail(FC_ERR_INVALID, "fc_setup_krylov: zero diagonal in the velocity block");
      dinv[(size_t)i] = 1.0 / X.dF[(size_t)i];
    }
    const double rhoF = fcpc::rho_dinv(X.F, X.dF);
    const double omega = sweeps > 1 ? std::min(1.0, 1.4 / rhoF) : 1.0;
    for (int i = 0; i < X.nu; ++i) wdinv[(size_t)i] = omega * dinv[(size_t)i];
    fcpc::Csr Sh = fcpc::spgemm(X.B, X.Bt, dinv.data());
    if (h->pin_dof >= 0) {  // enclosed flow: the Schur complement has the constant in its kernel; pin it where the factorisation does
      int ipin = -1;
      for (int k = 0; k < X.np; ++k)
        if (h->h_perm[(size_t)X.ppos[(size_t)k]] == h->pin_dof) ipin = k;
      if (ipin >= 0)
        for (int q = Sh.rp[(size_t)ipin]; q < Sh.rp[(size_t)ipin + 1]; ++q)
          if (Sh.ci[(size_t)q] == ipin) Sh.v[(size_t)q] += h->pin_shift;
    }
    const int64_t s_nnz = Sh.nnz();
    fcpc::Amg H = fcpc::build_amg(std::move(Sh));
    // smoothing sweeps of the V-cycle before and after the coarse correction (folded into the transfer operators either way: the launch
    // count does not change, the products get denser): FC_PC_AMG_SWEEPS=1|2
    // (O1, time steps: 18.6 -> 15.6 iterations per step, 520 -> 583 steps/s with two sweeps; cavity_fine: 39 -> 35 iterations but 453 ->
    //  510 us per iteration and twice the setup time, pinball 41 -> 36 at 237 -> 269 us: a wash -- the denser products stop being free once
    //  they leave the caches: one sweep beyond 16 384 pressure dofs)
    static const int amg_env = [] { const char* e = std::getenv("FC_PC_AMG_SWEEPS"); return e ? std::max(1, std::min(2, std::atoi(e))) : 0; }();
    const int amg_sweeps = amg_env ? amg_env : (X.np <= 16384 ? 2 : 1);
    // device copy (graphs captured over the old one hold its sizes and pointers: gone with it)
    krylov_drop_graphs(h, slot);
    Precond& P = S.pc;
    P.release();
    P.sweeps = sweeps, P.omega = omega, P.nu = X.nu, P.np = X.np;
    FCCHK(P.vpos.upload(X.vpos, h->stream));
    FCCHK(P.ppos.upload(X.ppos, h->stream));
    if (sweeps >= 2) {
      fcpc::Csr KF = fcpc::fold_jacobi2(X.F, wdinv);
      for (int& c : KF.ci) c = X.vpos[(size_t)c];  // the product reads the Krylov vector itself
      KF.ncols = N;
      FCCHK(P.KF.upload(KF, h->stream));
    }
    if (sweeps >= 3) FCCHK(P.F.upload(X.F, h->stream));
    FCCHK(P.B.upload(X.B, h->stream));
    FCCHK(P.Bt.upload(X.Bt, h->stream));
    FCCHK(P.dinvF.upload(dinv, h->stream));
    FCCHK(P.wdinvF.upload(wdinv, h->stream));
    FCCHK(P.u0.alloc((size_t)X.nu));
    FCCHK(P.u1.alloc((size_t)X.nu));
    FCCHK(P.zp.alloc((size_t)std::max(1, X.np)));
    P.bytes = P.KF.bytes() + P.F.bytes() + P.B.bytes() + P.Bt.bytes() + 4 * ((int64_t)X.nu + X.np) + 8 * (4 * (int64_t)X.nu + X.np);
    P.level_rows.clear();
    P.lv.resize(H.levels.size());
    for (size_t l = 0; l < H.levels.size(); ++l) {
      PcLevel& V = P.lv[l];
      const fcpc::Level& G = H.levels[l];
      V.n = G.A.nrows;
      V.n_next = G.P.ncols;
      if (amg_sweeps >= 2) {  // V(2,2): still two products per level, denser ones
        fcpc::Csr Gd, Uu;
        fcpc::fold_v22(G, Gd, Uu);
        FCCHK(V.G.upload(Gd, h->stream));
        FCCHK(V.U.upload(Uu, h->stream));
      } else {
        FCCHK(V.G.upload(fcpc::fold_down(G), h->stream));
        FCCHK(V.U.upload(fcpc::fold_up(G), h->stream));
      }
      FCCHK(V.cat.alloc((size_t)V.n + V.n_next));
      P.bytes += V.G.bytes() + V.U.bytes() + 8 * ((int64_t)V.n + V.n_next);
      P.level_rows.push_back(V.n);
    }
    P.n_coarse = H.n_coarse;
    P.level_rows.push_back(H.n_coarse);
    FCCHK(P.cinv.upload(H.coarse_inv, h->stream));
    FCCHK(P.rc.alloc((size_t)std::max(1, H.n_coarse)));
    P.bytes += 8 * ((int64_t)H.n_coarse * H.n_coarse + H.n_coarse);
    P.launches = std::max(1, sweeps - 1) + 3 + 2 * (int)H.levels.size();
    HIPCHK(hipStreamSynchronize(h->stream));  // the host vectors above go out of scope
    (void)s_nnz;
    P.ready = true;
  } catch (const std::exception& e) {
    return fail(FC_ERR_INVALID, std::string("fc_setup_krylov: ") + e.what());
  }
  // a slot is either factorised or factor-free: drop what fc_setup_solver may have left
  S.stages.clear();
  S.seg_ptr.release(), S.seg.release(), S.blk.release(), S.f_idx.release(), S.f_val.release(), S.f_val32.release(), S.f_val16.release();
  S.f_nnz = 0, S.sweep_bytes = 0.0, S.bits = 64;
  S.structured = S.truncated = S.inexact = S.nt = false;
  S.ar_stage = S.ar2_stage = -1, S.ar_n = 0;
  S.factor_free = true;
  S.ready = true;
  S.pc.setup_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  h->gmres_m = std::max(30, std::min(60, (int)max_iter));
  if (const char* e = std::getenv("FC_PC_WARM_START")) h->pc_warm_start = e[0] != '0';
  return fc_set_solver_options(h, method, max_iter, rtol, check_residual);
}

// info[8]: device bytes held by the preconditioner, velocity dofs, pressure dofs, AMG levels (sparse + the dense coarsest), rows of the
// coarsest level, kernel launches per apply, Jacobi sweeps, host milliseconds of the setup; omega_out: the Jacobi damping
int fc_get_krylov_info(fc_handle h, int slot, int64_t* info, double* omega_out) {
  if (!h || slot < 0 || slot > 1 || !info) return fail(FC_ERR_INVALID, "fc_get_krylov_info: bad argument");
  const OrderSys& S = h->sys[slot];
  if (!S.factor_free || !S.pc.ready) return fail(FC_ERR_NOT_READY, "fc_setup_krylov not called for this slot");
  const Precond& P = S.pc;
  info[0] = P.bytes + 12 * S.Ap_nnz + 8 * S.Ap_nnz;  // + the permuted system matrix and its source map
  info[1] = P.nu, info[2] = P.np, info[3] = (int64_t)P.lv.size() + 1, info[4] = P.n_coarse, info[5] = P.launches, info[6] = P.sweeps;
  info[7] = (int64_t)std::llround(P.setup_ms);
  if (omega_out) *omega_out = P.omega;
  return FC_OK;
}

static std::string sci(double v) {
  char buf[32];
  std::snprintf(buf, sizeof buf, "%.3e", v);
  return buf;
}

// Direct-apply residual of the acceptance solve below which factors count as exact.  Time-step operators sit at 1e-12 and below;
// sparse LU with partial pivoting reaches 1e-12 on the steady Re ~ 10^4 operators where block-local pivoting gives 1e-9 .. 1e-5.
static constexpr double kExactFactors = 1e-10;

int fc_get_factors_inexact(fc_handle h, int slot, int32_t* inexact) {
  if (!h || slot < 0 || slot > 1 || !inexact) return fail(FC_ERR_INVALID, "fc_get_factors_inexact: bad argument");
  *inexact = h->sys[slot].inexact ? 1 : 0;
  return FC_OK;
}

int fc_accept_factors(fc_handle h, int slot, double* residual_out, int32_t* inexact_out) {
  if (!h || slot < 0 || slot > 1) return fail(FC_ERR_INVALID, "fc_accept_factors: bad argument");
  OrderSys& S = h->sys[slot];
  if (!S.ready) return fail(FC_ERR_NOT_READY, "fc_accept_factors: no factors (fc_refactor)");
  if (S.truncated || S.bits != 64) return fail(FC_ERR_INVALID, "fc_accept_factors: truncated / compressed factors are preconditioners by construction");
  const int N = h->N;
  std::vector<double> b((size_t)N), x((size_t)N);
  for (int i = 0; i < N; ++i) b[i] = std::cos(0.37 * i + 0.1);
  // enclosed flow: a right-hand side compatible with the constant-pressure null space.  Decided by the GLOBAL pin
  // (h->pin_dof), not by pn_shift: on a partitioned handle only the rank that eliminates the pinned dof carries a
  // shift, and the ranks of a collective fc_solve must all build the same right-hand side.
  if (h->pin_dof >= 0)
    for (int i = 2 * h->nn; i < N; ++i) b[i] = 0.0;
  // the probe runs with options of its own; the caller's come back on EVERY exit (an early HIPCHK return kept the probe's)
  struct OptionsGuard {
    fc_ctx* h;
    int method, max_iter, check;
    double rtol;
    explicit OptionsGuard(fc_ctx* h_) : h(h_), method(h_->method), max_iter(h_->max_iter), check(h_->check_residual), rtol(h_->rtol) {}
    ~OptionsGuard() { h->method = method, h->max_iter = max_iter, h->rtol = rtol, h->check_residual = check; }
  } restore(h);
  double info[4] = {0, 0, 0, 0};
  S.inexact = false;
  h->method = FC_METHOD_REFINE, h->max_iter = 0, h->rtol = 1e-10, h->check_residual = 1;
  int code = fc_solve(h, slot, b.data(), x.data(), info);
  if (code == FC_OK && !(info[1] < kExactFactors) && info[1] < 1e-2) {
    // block-local pivoting lost digits on this (ill-conditioned) operator: the factors still are an excellent preconditioner --
    // accept them as such when GMRES reaches round-off in a handful of iterations; solves and steps on this slot then run GMRES
    const double direct = info[1];
    S.inexact = true;
    code = fc_solve(h, slot, b.data(), x.data(), info);  // (KrylovOverride: GMRES(60), 1e-12)
    // an operator this close to singular has |x| >> |b| / |A|, and eps |A| |x| / |b| is the floor of ANY solver's residual: what
    // GMRES must reach then is a normwise backward error |r| / (|A|_F |x| + |b|) at working precision.  (Single-device handles
    // only: the ranks of a partitioned handle hold different parts of A and must all take the same decision.)
    bool backward_ok = false;
    double eta = -1.0;
    if (code == FC_OK && !(info[1] < 1e-8) && !h->partitioned) {
      std::vector<double> av((size_t)h->nnz);
      HIPCHK(hipMemcpyAsync(av.data(), h->vals[slot].p, av.size() * sizeof(double), hipMemcpyDeviceToHost, h->stream));
      HIPCHK(hipStreamSynchronize(h->stream));
      double a2 = 0, x2 = 0, b2 = 0;
      for (double v : av) a2 += v * v;
      for (int i = 0; i < N; ++i) x2 += x[i] * x[i], b2 += b[i] * b[i];
      eta = info[1] * std::sqrt(b2) / (std::sqrt(a2) * std::sqrt(x2) + std::sqrt(b2));
      backward_ok = eta < 1e-13;
    }
    if (code != FC_OK || !(info[1] < 1e-8 || backward_ok) || info[0] > 40) {
      S.inexact = false;
      return fail(FC_ERR_HIP, "the factorisation failed its residual check (" + sci(direct) + " by the direct apply, " + sci(info[1]) + " after " + std::to_string((int)info[0]) + " GMRES iterations, backward error " + sci(eta) + ")");
    }
    info[1] = direct;
  } else if (code == FC_OK && !(info[1] < kExactFactors)) {
    code = fail(FC_ERR_HIP, "the factorisation failed its residual check (" + sci(info[1]) + ")");
  }
  if (code != FC_OK) return code;
  if (residual_out) *residual_out = info[1];
  if (inexact_out) *inexact_out = S.inexact ? 1 : 0;
  return FC_OK;
}

// ── the symbolic phase on its own, no device involved (tests compare it with flowcontrol_amd/ndsolver.py) ──
struct fc_sym {
  std::map<std::string, std::vector<int64_t>> v;
};

int fc_sym_build(int32_t nv, int32_t ne, int32_t nc, const double* coords, const int32_t* cells, const int32_t* cell_edges, int32_t n_bc,
                 const int32_t* bc_dofs, int32_t depth, int32_t merge, int32_t world, int32_t rank, int32_t truncate, void** out) {
  if (!out || !coords || !cells || !cell_edges || nv <= 0 || ne <= 0 || nc <= 0 || merge < 1 || world < 1 || rank < 0 || rank >= world)
    return fail(FC_ERR_INVALID, "fc_sym_build: bad argument");
  *out = nullptr;
  try {
    const int nn = nv + ne, N = 2 * nn + nv;
    std::vector<int> cd((size_t)nc * 15);
    std::vector<double> cent((size_t)nc * 2);
    for (int c = 0; c < nc; ++c) {
      for (int k = 0; k < 3; ++k) {
        const int v = cells[3 * c + k], e = cell_edges[3 * c + k];
        cd[(size_t)c * 15 + k] = v;
        cd[(size_t)c * 15 + 3 + k] = nv + e;
        cd[(size_t)c * 15 + 6 + k] = nn + v;
        cd[(size_t)c * 15 + 9 + k] = nn + nv + e;
        cd[(size_t)c * 15 + 12 + k] = 2 * nn + v;
      }
      for (int d = 0; d < 2; ++d)
        cent[2 * (size_t)c + d] = (coords[2 * cells[3 * c] + d] + coords[2 * cells[3 * c + 1] + d] + coords[2 * cells[3 * c + 2] + d]) / 3.0;
    }
    // CSR pattern (no pressure-pressure coupling), as fc_create builds it
    std::vector<uint64_t> keys;
    keys.reserve((size_t)nc * 216);
    for (int c = 0; c < nc; ++c)
      for (int i = 0; i < 15; ++i)
        for (int j = 0; j < 15; ++j) {
          if (i >= 12 && j >= 12) continue;
          keys.push_back(((uint64_t)cd[(size_t)c * 15 + i] << 32) | (uint32_t)cd[(size_t)c * 15 + j]);
        }
    std::sort(keys.begin(), keys.end());
    keys.erase(std::unique(keys.begin(), keys.end()), keys.end());
    std::vector<int> rowptr((size_t)N + 1, 0), col(keys.size());
    for (size_t k = 0; k < keys.size(); ++k) {
      rowptr[(keys[k] >> 32) + 1]++;
      col[k] = (int)(keys[k] & 0xFFFFFFFFu);
    }
    for (int r = 0; r < N; ++r) rowptr[r + 1] += rowptr[r];
    std::vector<unsigned char> skip((size_t)N, 0);
    for (int k = 0; k < n_bc; ++k) skip[(size_t)bc_dofs[k]] = 1;
    int top = 0;
    while ((1 << top) < world) ++top;
    const std::vector<int> bits = depth == 0 ? fcsym::default_bits(nc, merge, top) : fcsym::uniform_bits(depth, merge, top);
    const bool timing = getenv("FC_SYM_TIMING") != nullptr;
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto lap = [&](const char* what, std::chrono::steady_clock::time_point& t0) {
      if (timing) fprintf(stderr, "[fc_sym] %-18s %.3f s\n", what, std::chrono::duration<double>(now() - t0).count());
      t0 = now();
    };
    auto t0 = now();
    fcsym::Tree t = fcsym::build_tree(cd, 15, cent, nc, N, bits, &skip, top);
    lap("build_tree", t0);
    fcsym::Keep keep;
    if (world > 1) keep = [&t, rank, top](int k, int n) { return k == 0 || (n >> (t.cum[k] - top)) == rank; };
    if (truncate > 0) keep = [truncate](int k, int) { return k >= truncate; };
    const auto rr = fcsym::root_row_block(t, rank, world);
    fcsym::Factors fac = world > 1 && truncate == 0 ? fcsym::layout_factors(t, keep, rr.first, rr.second) : fcsym::layout_factors(t, keep);
    lap("layout_factors", t0);
    if (truncate > 0)
      for (int k = 0; k < truncate; ++k) fac.stage_kind[(size_t)t.depth + k] = 2;
    fcsym::Plan pl = fcsym::factor_plan(t, fac, rowptr, col, &skip, keep);
    lap("factor_plan", t0);
    fcsym::Partition part = fcsym::partition(t, fac, rank, world);
    lap("partition", t0);
    fcsym::Blocks B = fcsym::down_blocks(t, fac, rank, world);
    lap("down_blocks", t0);
    fc_sym* sy = new fc_sym();
    auto put = [&](const char* name, auto const& vec) { sy->v[name].assign(vec.begin(), vec.end()); };
    put("perm", t.perm);
    put("cum", t.cum);
    put("leaf_of_cell", t.leaf_of_cell);
    put("idx", fac.idx);
    put("seg_ptr", fac.seg_ptr);
    put("seg_val", fac.seg_val);
    put("seg_col", fac.seg_col);
    put("seg_len", fac.seg_len);
    put("stage_begin", fac.stage_begin);
    put("stage_row0", fac.stage_row0);
    put("stage_nrows", fac.stage_nrows);
    put("stage_kind", fac.stage_kind);
    put("nodes", fac.nodes);
    sy->v["n_val"] = {fac.n_val};
    sy->v["nnz"] = {fac.nnz};
    put("plan_nodes", pl.nodes);
    put("level_ptr", pl.level_ptr);
    put("a_src", pl.a_src);
    put("a_dst", pl.a_dst);
    put("a_ptr", pl.a_ptr);
    put("ext_off", pl.ext_off);
    put("ext_p", pl.ext_p);
    put("ap_src", pl.ap_src);
    put("Ap_rowptr", pl.Ap_rowptr);
    put("Ap_col", pl.Ap_col);
    sy->v["front_size"] = {pl.front_size};
    sy->v["max_slots"] = {pl.max_slots};
    put("part_rowkind", part.rowkind);
    put("part_local_cells", part.local_cells);
    put("part_seg_ptr", part.seg_ptr);
    put("part_seg_val", part.seg_val);
    put("part_seg_col", part.seg_col);
    put("part_seg_len", part.seg_len);
    put("part_stage_begin", part.stage_begin);
    put("part_stage_row0", part.stage_row0);
    put("part_stage_nrows", part.stage_nrows);
    put("part_stage_kind", part.stage_kind);
    sy->v["part_ar"] = {part.ar_stage, part.ar_row0, part.ar_n, part.ar2_stage, part.root_row0, part.root_nrows};
    put("blk_begin", B.begin);
    put("blk_count", B.count);
    put("blk_lpr", B.lpr);
    put("blk_val", B.val);
    put("blk_row0", B.row0);
    put("blk_nrows", B.nrows);
    put("blk_i0", B.i0);
    put("blk_ni", B.ni);
    put("blk_idx", B.idx);
    put("blk_nb", B.nb);
    *out = sy;
  } catch (const std::exception& e) {
    return fail(FC_ERR_INVALID, std::string("fc_sym_build: ") + e.what());
  }
  return FC_OK;
}

int fc_sym_size(void* sym, const char* name, int64_t* n) {
  if (!sym || !name || !n) return fail(FC_ERR_INVALID, "fc_sym_size: null argument");
  const auto& m = static_cast<fc_sym*>(sym)->v;
  const auto it = m.find(name);
  if (it == m.end()) return fail(FC_ERR_INVALID, std::string("fc_sym_size: no table named ") + name);
  *n = (int64_t)it->second.size();
  return FC_OK;
}

int fc_sym_get(void* sym, const char* name, int64_t* out) {
  if (!sym || !name || !out) return fail(FC_ERR_INVALID, "fc_sym_get: null argument");
  const auto& m = static_cast<fc_sym*>(sym)->v;
  const auto it = m.find(name);
  if (it == m.end()) return fail(FC_ERR_INVALID, std::string("fc_sym_get: no table named ") + name);
  std::copy(it->second.begin(), it->second.end(), out);
  return FC_OK;
}

int fc_sym_free(void* sym) {
  delete static_cast<fc_sym*>(sym);
  return FC_OK;
}

int fc_get_permutation(fc_handle h, int32_t* perm) {
  if (!h || !perm) return fail(FC_ERR_INVALID, "fc_get_permutation: null argument");
  if (!h->have_perm) return fail(FC_ERR_NOT_READY, "no permutation yet (fc_setup_solver / fc_set_permutation)");
  std::copy(h->h_perm.begin(), h->h_perm.end(), perm);
  return FC_OK;
}

int fc_get_solver_info(fc_handle h, int slot, int64_t* info) {
  if (!h || slot < 0 || slot > 1 || !info) return fail(FC_ERR_INVALID, "fc_get_solver_info: bad argument");
  const OrderSys& S = h->sys[slot];
  info[0] = S.f_nnz;                   // factor values stored on this rank
  info[1] = h->sym_ready ? h->sym_total_nnz : S.f_nnz;  // ... of the whole tree (all ranks, no truncation)
  info[2] = h->sym_ready ? h->sym_local_values[slot] : 0;  // factor values this rank sweeps per solve
  info[3] = (int64_t)S.stages.size();
  info[4] = h->sym_ready ? h->sym_tree.depth : 0;
  info[5] = S.ar_n;
  info[6] = S.ar_stage;
  info[7] = S.ar2_stage;
  info[8] = h->partitioned ? h->ncl : h->nc;
  info[9] = h->sym_ready ? h->sym_truncate : 0;
  return FC_OK;
}

// Shape of the handle's elimination tree (bench / documentation): bits_out[<= 16] = bisections fused per level, root first (the tree of
// fc_setup_solver if it ran, else the default shape for this mesh); nnz_min_tree (optional) = factor values of the all-binary-pairs
// tree [2, 2, ...] with the same number of bisections -- the layout with the least fill among the shapes fc_setup_solver chooses from,
// the fixed denominator of bench.py's roofline.frac_min_tree (one symbolic pass on the host, ~0.1 s on O1).
int fc_get_tree_info(fc_handle h, int32_t* bits_out, int32_t* n_bits, int64_t* nnz_min_tree) {
  if (!h || !bits_out || !n_bits) return fail(FC_ERR_INVALID, "fc_get_tree_info: bad argument");
  const std::vector<int> bits = h->sym_ready && !h->sym_bits.empty() ? h->sym_bits : fcsym::default_bits(h->nc, 2, 0);
  if (bits.size() > 16) return fail(FC_ERR_INVALID, "fc_get_tree_info: more than 16 tree levels");
  *n_bits = (int32_t)bits.size();
  for (size_t i = 0; i < bits.size(); ++i) bits_out[i] = bits[i];
  if (nnz_min_tree) {
    try {
      int sum = 0;
      for (int b : bits) sum += b;
      std::vector<unsigned char> skip((size_t)h->N, 0);
      for (int k = 0; k < h->n_bc; ++k) skip[(size_t)h->h_bc_dofs[k]] = 1;
      const fcsym::Tree t = fcsym::build_tree(h->h_cell_dofs, 15, h->h_cent, h->nc, h->N, fcsym::uniform_bits(sum, 2, 0), &skip, 0);
      *nnz_min_tree = fcsym::layout_factors(t, nullptr).nnz;
    } catch (const std::exception& e) {
      return fail(FC_ERR_INVALID, std::string("fc_get_tree_info: ") + e.what());
    }
  }
  return FC_OK;
}

int fc_get_rowkind(fc_handle h, uint8_t* rowkind) {
  if (!h || !rowkind) return fail(FC_ERR_INVALID, "fc_get_rowkind: null argument");
  if (h->h_rowkind.empty())
    std::fill(rowkind, rowkind + h->N, (uint8_t)1);
  else
    std::copy(h->h_rowkind.begin(), h->h_rowkind.end(), rowkind);
  return FC_OK;
}

int fc_update_operator(fc_handle h, int slot) {
  if (!h || slot < 0 || slot > 1) return fail(FC_ERR_INVALID, "fc_update_operator: bad argument");
  if (h->sys[slot].factor_free) {  // the preconditioner of the earlier operator stays (call fc_setup_krylov again for a new one)
    OrderSys& S = h->sys[slot];
    HIPCHK(hipSetDevice(h->device));
    FCCHK(quiesce(h));
    hipLaunchKernelGGL(fc_gather64, dim3(nblocks(S.Ap_nnz, 256)), dim3(256), 0, h->stream, S.Ap_nnz, S.ap_src.p, h->vals[slot].p, S.Ap_val.p);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(h->stream));
    return FC_OK;
  }
  if (!h->have_plan) return fail(FC_ERR_NOT_READY, "fc_factor_plan not called");
  OrderSys& S = h->sys[slot];
  if (!S.structured) return fail(FC_ERR_NOT_READY, "fc_solver_setup (structure) must be called first");
  if ((int64_t)h->pap_src.n != S.Ap_nnz) return fail(FC_ERR_INVALID, "fc_update_operator: plan and solver structure disagree");
  HIPCHK(hipSetDevice(h->device));
  FCCHK(quiesce(h));
  hipLaunchKernelGGL(fc_gather64, dim3(nblocks(S.Ap_nnz, 256)), dim3(256), 0, h->stream, S.Ap_nnz, h->pap_src.p, h->vals[slot].p, S.Ap_val.p);
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(h->stream));
  S.ready = true;  // the factors of the earlier operator stay: a preconditioner for FC_METHOD_BICGSTAB
  return FC_OK;
}

int fc_get_factor_values(fc_handle h, int slot, int64_t n, double* out) {
  if (!h || slot < 0 || slot > 1 || !out) return fail(FC_ERR_INVALID, "fc_get_factor_values: bad argument");
  OrderSys& S = h->sys[slot];
  if (!S.structured) return fail(FC_ERR_NOT_READY, "fc_solver_setup not called for this slot");
  if (n != S.f_nnz) return fail(FC_ERR_INVALID, "fc_get_factor_values: size differs from the uploaded factors");
  if (S.bits != 64) {  // compressed factors: widened back to fp64 on the host
    HIPCHK(hipSetDevice(h->device));
    if (S.bits == 32) {
      std::vector<float> tmp((size_t)n);
      HIPCHK(hipMemcpy(tmp.data(), S.f_val32.p, (size_t)n * sizeof(float), hipMemcpyDeviceToHost));
      for (int64_t i = 0; i < n; ++i) out[i] = (double)tmp[(size_t)i];
    } else {
      std::vector<unsigned short> tmp((size_t)n);
      HIPCHK(hipMemcpy(tmp.data(), S.f_val16.p, (size_t)n * sizeof(unsigned short), hipMemcpyDeviceToHost));
      for (int64_t i = 0; i < n; ++i) {
        const uint32_t u = (uint32_t)tmp[(size_t)i] << 16;
        float f;
        std::memcpy(&f, &u, sizeof f);
        out[i] = (double)f;
      }
    }
    return FC_OK;
  }
  HIPCHK(hipSetDevice(h->device));
  HIPCHK(hipMemcpyAsync(out, S.f_val.p, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  return FC_OK;
}

int fc_set_rhs_operator(fc_handle h, int slot, const int32_t* rowptr, const int32_t* col, const double* val) {
  if (!h || slot < 0 || slot > 1) return fail(FC_ERR_INVALID, "fc_set_rhs_operator: bad argument");
  HIPCHK(hipSetDevice(h->device));
  FCCHK(quiesce(h));
  OrderSys& S = h->sys[slot];
  if (!rowptr) {
    S.have_c = false;
    return FC_OK;
  }
  if (!col || !val) return fail(FC_ERR_INVALID, "fc_set_rhs_operator: null argument");
  const int N = h->N;
  const int nz = rowptr[N];
  if (rowptr[0] != 0 || nz < 0) return fail(FC_ERR_INVALID, "fc_set_rhs_operator: bad row pointers");
  for (int i = 0; i < N; ++i)
    if (rowptr[i + 1] < rowptr[i]) return fail(FC_ERR_INVALID, "fc_set_rhs_operator: bad row pointers");
  for (int k = 0; k < nz; ++k)
    if (col[k] < 0 || col[k] >= 2 * h->nn) return fail(FC_ERR_INVALID, "fc_set_rhs_operator: column is not a velocity dof");
  if (!h->have_perm) return fail(FC_ERR_NOT_READY, "fc_set_rhs_operator: the rows are in the solver's permuted numbering (fc_setup_solver / fc_set_permutation first)");
  // the columns address u_n, which lives in the permuted numbering on the device: W dof -> permuted position
  std::vector<int> ip((size_t)N), colp((size_t)std::max(1, nz), 0);
  for (int i = 0; i < N; ++i) ip[(size_t)h->h_perm[i]] = i;
  for (int k = 0; k < nz; ++k) colp[(size_t)k] = ip[(size_t)col[k]];
  FCCHK(S.c_rowptr.upload(rowptr, (size_t)N + 1, h->stream));
  FCCHK(S.c_col.upload(colp, h->stream));
  FCCHK(S.c_val.upload(val, (size_t)std::max(1, nz), h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  S.have_c = true;
  return FC_OK;
}

int fc_set_energy_matrix(fc_handle h, const int32_t* rowptr, const int32_t* col, const double* val) {
  if (!h || !rowptr || !col || !val) return fail(FC_ERR_INVALID, "fc_set_energy_matrix: null argument");
  HIPCHK(hipSetDevice(h->device));
  FCCHK(quiesce(h));
  const int N = h->N;
  const int nz = rowptr[N];
  if (rowptr[0] != 0 || nz < 0) return fail(FC_ERR_INVALID, "fc_set_energy_matrix: bad row pointers");
  for (int i = 0; i < N; ++i)
    if (rowptr[i + 1] < rowptr[i]) return fail(FC_ERR_INVALID, "fc_set_energy_matrix: bad row pointers");
  for (int k = 0; k < nz; ++k)
    if (col[k] < 0 || col[k] >= N) return fail(FC_ERR_INVALID, "fc_set_energy_matrix: column out of range");
  FCCHK(h->mp_rowptr.upload(rowptr, (size_t)N + 1, h->stream));
  FCCHK(h->mp_col.upload(col, (size_t)std::max(1, nz), h->stream));
  FCCHK(h->mp_val.upload(val, (size_t)std::max(1, nz), h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  h->have_mp = true;
  return FC_OK;
}

int fc_set_solver_options(fc_handle h, int method, int max_iter, double rtol, int check_residual) {
  if (!h || max_iter < 0 || max_iter > 1000 || check_residual < -1) return fail(FC_ERR_INVALID, "fc_set_solver_options: bad argument");
  if (method != FC_METHOD_REFINE && method != FC_METHOD_BICGSTAB && method != FC_METHOD_GMRES)
    return fail(FC_ERR_INVALID, "fc_set_solver_options: unknown method (REFINE, BICGSTAB, GMRES)");
  if (method != FC_METHOD_REFINE && (max_iter < 1 || !(rtol > 0.0)))
    return fail(FC_ERR_INVALID, "fc_set_solver_options: the Krylov methods need max_iter >= 1 and rtol > 0");
  h->method = method;
  h->max_iter = max_iter;
  h->rtol = rtol;
  h->check_residual = check_residual;
  return FC_OK;
}

int fc_set_state(fc_handle h, const double* u_n, const double* u_nn, const double* p_n) {
  if (!h || !u_n || !u_nn) return fail(FC_ERR_INVALID, "fc_set_state: null argument");
  h->pre_slot = -1;
  h->undo_ok = false;
  HIPCHK(hipSetDevice(h->device));
  FCCHK(quiesce(h));
  const size_t nv2 = 2 * (size_t)h->nn, N = (size_t)h->N;
  // W-layout vectors [u | p] of both time levels (p_n = NULL keeps the present pressure)
  std::vector<double> wn(N, 0.0), wnn(N, 0.0);
  if (!p_n) {
    if (h->have_perm && h->state_live) FCCHK(state_download(h, wn.data(), nullptr));
    else if (!h->hs_n.empty()) wn = h->hs_n;
  }
  std::copy(u_n, u_n + nv2, wn.begin());
  std::copy(u_nn, u_nn + nv2, wnn.begin());
  if (p_n) std::copy(p_n, p_n + h->nv, wn.begin() + (std::ptrdiff_t)nv2);
  std::copy(wn.begin() + (std::ptrdiff_t)nv2, wn.end(), wnn.begin() + (std::ptrdiff_t)nv2);
  HIPCHK(hipMemsetAsync(h->flag.p, 0, sizeof(int), h->stream));
  if (h->have_perm) {
    FCCHK(state_upload(h, wn.data(), wnn.data()));
    h->hs_n.clear(), h->hs_nn.clear();
  } else {  // no numbering to store it in yet: fc_set_permutation takes it from here
    h->hs_n.swap(wn), h->hs_nn.swap(wnn);
    HIPCHK(hipStreamSynchronize(h->stream));
  }
  return FC_OK;
}

int fc_undo_step(fc_handle h) {
  if (!h) return fail(FC_ERR_INVALID, "null handle");
  if (h->step_pending) return fail(FC_ERR_INVALID, "fc_undo_step: collect the step first (fc_step_end)");
  if (!h->undo_ok) return fail(FC_ERR_NOT_READY, "fc_undo_step: the last state change was not a single fc_step");
  HIPCHK(hipSetDevice(h->device));
  FCCHK(quiesce(h));
  HIPCHK(hipStreamSynchronize(h->stream));  // (a speculated element loop may still be reading the withdrawn state)
  // the step wrote its solution into a ring slot of its own: the three older levels are still where they were
  h->cur = (h->cur + 3) % 4;
  ring_point(h);
  HIPCHK(hipMemsetAsync(h->flag.p, 0, sizeof(int), h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  h->pre_slot = -1;  // a speculative element loop of the next step read the state that was just withdrawn
  h->undo_ok = false;
  return FC_OK;
}

int fc_get_state(fc_handle h, double* u_n, double* u_nn, double* p_n) {
  if (!h) return fail(FC_ERR_INVALID, "null handle");
  HIPCHK(hipSetDevice(h->device));
  FCCHK(quiesce(h));
  const size_t nv2 = 2 * (size_t)h->nn, N = (size_t)h->N;
  std::vector<double> wn(N, 0.0), wnn(N, 0.0);
  if (h->have_perm && h->state_live) FCCHK(state_download(h, wn.data(), u_nn ? wnn.data() : nullptr));
  else if (!h->hs_n.empty()) wn = h->hs_n, wnn = h->hs_nn;
  if (u_n) std::copy(wn.begin(), wn.begin() + (std::ptrdiff_t)nv2, u_n);
  if (u_nn) std::copy(wnn.begin(), wnn.begin() + (std::ptrdiff_t)nv2, u_nn);
  if (p_n) std::copy(wn.begin() + (std::ptrdiff_t)nv2, wn.end(), p_n);
  return FC_OK;
}

int fc_get_solution(fc_handle h, double* up) {
  if (!h || !up) return fail(FC_ERR_INVALID, "fc_get_solution: null argument");
  HIPCHK(hipSetDevice(h->device));
  FCCHK(quiesce(h));
  // the last step's solution IS the state (u_n, p_n)
  if (h->have_perm && h->state_live) return state_download(h, up, nullptr);
  std::fill(up, up + h->N, 0.0);
  if (!h->hs_n.empty()) std::copy(h->hs_n.begin(), h->hs_n.end(), up);
  return FC_OK;
}

// Behind a synchronous step: the element loop of the next right-hand side reads the state only (BC
// actuation enters in fc_rhs_gather), so it is enqueued now and runs while the host is between two
// fc_step calls; enqueue_rhs skips its own launch when the prediction (same scheme, BDF2 after BDF1)
// holds and nothing touched the state in between.  (Body forces enter in fc_rhs_gather too: build_force_vectors.)
void speculate_next_rhs(fc_ctx* h, int order_slot, hipStream_t stream = nullptr) {
  if (!stream) stream = h->stream;
  static const bool enabled = [] {
    const char* e = std::getenv("FC_SPECULATE");
    return !(e && e[0] == '0');
  }();
  h->pre_slot = -1;
  const int ncl = h->partitioned ? h->ncl : h->nc;
  if (!enabled || ncl <= 0 || h->phase_timing) return;  // (phase timing: the element loop is charged to the step it belongs to)
  const int next = h->sys[order_slot].have_c ? order_slot : FC_SLOT_BDF2;
  if (!h->sys[next].ready || !h->sys[next].have_lift) return;
  const StepCoeffs c = coeffs_for(h, next);
  launch_step_elem(h, stream, c, h->pin_dev, ncl);
  if (hipGetLastError() == hipSuccess) h->pre_slot = next;
}

// fc_step in two halves: fc_step_begin writes the controls into the mapped record and enqueues the step's launches (the GPU
// works from here on), fc_step_end waits for the record.  A host program can do its own per-step bookkeeping in between
// (FlowSolver.step appends the previous step's log row there); fc_step is begin + end.
constexpr int kLateRec = 8010;  // late records (two step parities) of the overlapped tail in the pinned page: [E, sum r^2, sum b^2, seq, checksum, checksum, -, -]

// the overlapped form of a step (fc_ctx::stream2): see the comment there
static bool step_can_overlap(const fc_ctx* h, int order_slot) {
  const OrderSys& S = h->sys[order_slot];
  // (factors that stream from HBM -- S.nt -- leave no slack for a concurrent matrix pass: measured on the refined cylinder, the pinball and
  //  cavity_fine the overlapped tail costs 6-8 % of the step rate where the cache-resident O1 gains 10 %; profiles/r04_overlap.txt)
  static const bool overlap_nt = [] { const char* e = std::getenv("FC_OVERLAP_NT"); return e && e[0] == '1'; }();  // experiment: overlap on streaming meshes too
  return h->overlap && !h->want_all && (!S.nt || overlap_nt) && !h->partitioned && h->method == FC_METHOD_REFINE && !S.inexact && !S.truncated && S.bits == 64 && use_fused_tail(h) && !h->timing &&
         !h->phase_timing;
}

static int step_enqueue_overlapped(fc_ctx* h) {
  double* dev = h->pin_dev;
  const int order_slot = h->pend_slot, compute_energy = h->pend_energy;
  OrderSys& S = h->sys[order_slot];
  if (!S.ready) return fail(FC_ERR_NOT_READY, "fc_solver_setup not called for this order");
  const int par = (int)(h->step_count & 1);
  // the late tail of step n - 2 read b(n - 2) -- the buffer this step assembles into -- and, two steps on, its ring slot would be
  // overwritten: make sure it has finished (it did, a step ago; this is a read of a host-mapped word)
  FCCHK(collect_late(h, par));
  h->b.p = h->bstore.p + (size_t)par * (size_t)h->N;
  FCCHK(enqueue_rhs(h, order_slot, dev, dev + 32));
  h->sweep_check = true;  // the down-sweep launches raise h->flag on a non-finite velocity entry as they write it
  const int code = apply_factors(h, S);
  h->sweep_check = false;
  FCCHK(code);
  const double* x = h->buf.p + h->N;
  const double* b_now = h->b.p;
  hipLaunchKernelGGL(fc_early, dim3(1), dim3(256), 0, h->stream, h->n_sens, h->s_rowptr.p, h->s_idxp.p, h->s_w.p, x, dev + 64, h->flag.p, dev + 136, dev + 137,
                     h->pend_seq, h->solved.p);
  ring_advance(h);  // the solution just written is the state from here on
  h->state_live = true;
  const int every = residual_every(h, S);
  const bool res = every != 0 && (h->step_count % (uint64_t)every) == 0;
  h->last_checked = res;
  ++h->step_count;
  speculate_next_rhs(h, order_slot);  // the next step's element loop, on the main stream behind fc_early as ever
  // ---- side stream: [gate: this step's solve has finished] residual monitor + energy -> this step's late record
  // (with factors that stream from HBM a concurrent matrix pass costs more than it hides: step_can_overlap.  The same late tail kept on the
  //  MAIN stream behind fc_early -- so that the host has y and enqueues the next step while the tail runs -- was measured too: 3-4 % slower
  //  than the plain one-stream step on the three large meshes, the next step's first launch starts 14 us after fc_final_late although it was
  //  enqueued 45 us earlier; profiles/EXPERIMENTS.md)
  hipStream_t ts = h->stream2;
  hipLaunchKernelGGL(fc_wait_solved, dim3(1), dim3(1), 0, ts, h->solved.p, (fc_u64)h->pend_seq, h->side_err.p, h->gate_spin);
  const int reps = std::max(1, nblocks(h->N, 32 * 2048));
  const int g_rows = res ? nblocks(h->N, 32 * reps) : 0, g_cells = (compute_energy && h->nc > 0) ? nblocks(h->nc, 32 * reps) : 0;
  const int g = g_rows + g_cells;
  if (g > h->nblk_N) return fail(FC_ERR_INVALID, "step: partial buffer too small");
  if (g > 0)
    hipLaunchKernelGGL(fc_tail<false>, dim3(g), dim3(256), 0, ts, h->N, h->velrow_p.p, x, b_now, res ? S.Ap_rowptr.p : nullptr, S.Ap_col.p, S.Ap_val.p, g_rows,
                       0, reps, h->nc, g_cells > 0 ? h->cnp.p : nullptr, h->geom.p, (const unsigned char*)nullptr, (const int*)nullptr, h->nc, h->flag2.p,
                       h->partial.p, FcFin{});
  hipLaunchKernelGGL(fc_final_late, dim3(1), dim3(256), 0, ts, g_cells > 0 ? g : 0, g_cells > 0 ? h->partial.p + 2 * (size_t)g : nullptr, res ? g : 0,
                     res ? h->partial.p : nullptr, dev + kLateRec + 8 * par, h->pend_seq, (const int*)h->side_err.p);
  HIPCHK(hipGetLastError());
  h->late[par].pending = true;
  h->late[par].seq = h->pend_seq;
  h->late[par].energy = compute_energy;
  h->late[par].checked = res;
  h->side_busy = true;
  return FC_OK;
}

static int step_enqueue(fc_ctx* h) {
  double* dev = h->pin_dev;
  h->pend_seq = (double)(++h->seq);
  h->pend_overlapped = step_can_overlap(h, h->pend_slot);
  if (h->pend_overlapped) {
    h->pend_par = (int)(h->step_count & 1);
    FCCHK(step_enqueue_overlapped(h));
    h->pend_checked = h->last_checked;
    h->undo_ok = true;
    return FC_OK;
  }
  FCCHK(quiesce(h));
  h->b.p = h->bstore.p;
  FCCHK(enqueue_step(h, h->pend_slot, dev, dev + 64, dev + 128, dev + 129, dev + 136, h->pend_energy, dev + 32, dev + 137, h->pend_seq));
  h->pend_checked = h->last_checked;
  h->undo_ok = true;  // the step wrote into a ring slot of its own: the older levels are intact
  speculate_next_rhs(h, h->pend_slot);
  return FC_OK;
}

int fc_step_begin(fc_handle h, int order_slot, const double* u_ctrl, const double* u_force, int compute_energy) {
  FCCHK(check_step_ready(h, order_slot));
  if (h->n_act > 0 && !u_ctrl) return fail(FC_ERR_INVALID, "fc_step: u_ctrl is null");
  if (h->n_act > 32) return fail(FC_ERR_INVALID, "fc_step: at most 32 actuators");
  if (h->step_pending) return fail(FC_ERR_INVALID, "fc_step_begin: the previous step was not collected (fc_step_end)");
  HIPCHK(hipSetDevice(h->device));
  // zero-copy record in pinned, device-mapped host memory: the kernels read u_ctrl from it and the
  // last kernel of the step writes (y, dE, |r|^2, |b|^2, flag) into it — no memcpy on the stream.
  volatile double* pin = h->pin;
  for (int k = 0; k < h->n_act; ++k) {
    pin[k] = u_ctrl[k];
    pin[32 + k] = u_force ? u_force[k] : u_ctrl[k];  // body-force amplitudes (CN: mean of new and old)
  }
  h->pend_slot = order_slot;
  h->pend_energy = compute_energy;
  FCCHK(step_enqueue(h));
  h->step_pending = true;
  return FC_OK;
}

// the late record of an overlapped step (fc_final_late on the side stream): [E, sum r^2, sum b^2, seq, checksum, checksum], one per step parity
int collect_late(fc_ctx* h, int par) {
  fc_ctx::Late& L = h->late[par];
  if (!L.pending) return FC_OK;
  volatile double* rec = h->pin + kLateRec + 8 * par;
  const double seq = L.seq;
  auto bits = [](double v) {
    unsigned long long u;
    std::memcpy(&u, &v, sizeof u);
    return u;
  };
  auto ok = [&]() {
    if (rec[3] != seq) return false;
    unsigned long long x = bits(seq), w = x, k = 3;
    for (int i = 0; i < 4; ++i, k += 2) {
      const unsigned long long v = bits(rec[i < 3 ? i : 6]);
      x ^= v;
      w += k * v;
    }
    return x == bits(rec[4]) && w == bits(rec[5]);
  };
  bool seen = false;
  for (long spin = 0; spin < 20000000L; ++spin) {
    if (ok()) {
      seen = true;
      break;
    }
    __builtin_ia32_pause();
  }
  if (!seen) {
    HIPCHK(hipStreamSynchronize(h->stream2));
    if (!ok()) return fail(FC_ERR_HIP, "fc_step: the late record (residual monitor, energy) never arrived or failed its checksum");
  }
  L.pending = false;
  if (rec[6] != 0.0)  // the side stream's gate gave up: its tail may have read buffers the main stream was still writing
    return fail(FC_ERR_HIP, "fc_step: the side stream stopped waiting for the step's solve (main stream delayed or a launch failed): "
                            "residual monitor and energy of that step are not valid");
  if (par == h->last_par) {  // the values fc_step_collect hands out are those of the LAST step that ended
    const double nan = std::numeric_limits<double>::quiet_NaN();
    const double r2 = rec[1], b2 = rec[2];
    h->last_dE = L.energy ? rec[0] : nan;
    h->last_info[1] = L.checked ? std::sqrt(r2 / (b2 > 0 ? b2 : 1.0)) : nan;
    h->last_info[2] = L.checked ? std::sqrt(b2) : nan;
  }
  return FC_OK;
}

int fc_step_end(fc_handle h, double* y_out, double* dE_out, double* info_out) {
  if (!h || !h->step_pending) return fail(FC_ERR_INVALID, "fc_step_end: no step in flight (fc_step_begin)");
  h->step_pending = false;
  HIPCHK(hipSetDevice(h->device));
  volatile double* pin = h->pin;
  const int compute_energy = h->pend_energy;
  {
    const double seq = h->pend_seq;
    // the last kernel publishes the record and `seq` with no fence: poll the host-mapped words (bounded), then fall
    // back to a stream synchronisation — which is also what reports a faulted kernel.
    // A record is accepted only when both of its checksums (fc_publish) agree with the words actually read: the
    // individual device writes may become visible to the host in any order.
    auto bits = [](double v) {
      unsigned long long u;
      std::memcpy(&u, &v, sizeof u);
      return u;
    };
    auto record_ok = [&]() {
      if (pin[137] != seq) return false;
      unsigned long long x = bits(seq), w = x, k = 3;
      for (int s = 0; s < h->n_sens; ++s, k += 2) {
        const unsigned long long v = bits(pin[64 + s]);
        x ^= v;
        w += k * v;
      }
      const unsigned long long tail[4] = {bits(pin[128]), bits(pin[129]), bits(pin[130]), bits(pin[136])};
      for (int i = 0; i < 4; ++i, k += 2) {
        x ^= tail[i];
        w += k * tail[i];
      }
      return x == bits(pin[138]) && w == bits(pin[139]);
    };
    bool seen = false;
    if (!h->timing && !h->phase_timing) {
      for (long spin = 0; spin < 20000000L; ++spin) {
        if (record_ok()) {
          seen = true;
          break;
        }
        __builtin_ia32_pause();
      }
    }
    if (!seen) {
      HIPCHK(hipStreamSynchronize(h->stream));
      FCCHK(time_collect(h));
      FCCHK(phase_collect(h));
      if (!record_ok()) return fail(FC_ERR_HIP, "fc_step: the step record failed its checksum after stream synchronisation");
    }
  }
  for (int s = 0; s < h->n_sens; ++s)
    if (y_out) y_out[s] = pin[64 + s];
  const int flag = ((int)pin[136]) % 1024;
  const double nan = std::numeric_limits<double>::quiet_NaN();
  h->last_info[0] = h->method == FC_METHOD_REFINE ? h->max_iter : h->last_krylov_iters;  // refinement sweeps / Krylov iterations
  h->last_info[3] = flag;
  if (h->pend_overlapped) {
    h->last_par = h->pend_par;
    // energy and residual norms follow in the late record: wait for it only if the caller wants them now
    if (dE_out || info_out) FCCHK(collect_late(h, h->last_par));
  } else {
    const double r2 = pin[129], b2 = pin[130];
    const bool checked = h->pend_checked;  // (check_residual = n > 1: the monitor ran on every n-th step only)
    h->last_dE = compute_energy ? pin[128] : nan;
    h->last_info[1] = checked ? std::sqrt(r2 / (b2 > 0 ? b2 : 1.0)) : nan;
    h->last_info[2] = checked ? std::sqrt(b2) : nan;
  }
  if (dE_out) *dE_out = h->last_dE;
  if (info_out) std::copy(h->last_info, h->last_info + 4, info_out);
  if (flag) return fail(FC_ERR_DIVERGED, "non-finite velocity after solve");
  return FC_OK;
}

// energy and solve info of the last fc_step_end that was called WITHOUT dE_out / info_out (the overlapped tail computes them while the host
// is already busy with the next step): blocks until they are there.  Always valid after any fc_step_end.
int fc_step_collect(fc_handle h, double* dE_out, double* info_out) {
  if (!h) return fail(FC_ERR_INVALID, "null handle");
  HIPCHK(hipSetDevice(h->device));
  FCCHK(collect_late(h, h->last_par));
  if (dE_out) *dE_out = h->last_dE;
  if (info_out) std::copy(h->last_info, h->last_info + 4, info_out);
  return FC_OK;
}

int fc_step(fc_handle h, int order_slot, const double* u_ctrl, const double* u_force, double* y_out, double* dE_out,
            int compute_energy, double* info_out) {
  if (!h) return fail(FC_ERR_INVALID, "null handle");
  h->want_all = dE_out != nullptr || info_out != nullptr;
  const int code = fc_step_begin(h, order_slot, u_ctrl, u_force, compute_energy);
  h->want_all = false;
  FCCHK(code);
  return fc_step_end(h, y_out, dE_out, info_out);
}

int fc_run(fc_handle h, int first_order_slot, int32_t n_steps, const double* u_ctrl, int u_ctrl_is_sequence,
           double* y_seq, double* dE_seq, int compute_energy) {
  FCCHK(check_step_ready(h, first_order_slot));
  if (n_steps <= 0) return fail(FC_ERR_INVALID, "fc_run: n_steps must be positive");
  h->undo_ok = false;  // (fc_undo_step withdraws a single fc_step)
  h->b.p = h->bstore.p;
  if (h->n_act > 0 && !u_ctrl) return fail(FC_ERR_INVALID, "fc_run: u_ctrl is null");
  HIPCHK(hipSetDevice(h->device));
  FCCHK(quiesce(h));
  const int na = std::max(1, h->n_act), ns = std::max(1, h->n_sens);
  const size_t nu = u_ctrl_is_sequence ? (size_t)n_steps * na : (size_t)na;
  std::vector<double> uh(nu, 0.0);
  if (h->n_act) std::copy(u_ctrl, u_ctrl + (u_ctrl_is_sequence ? (size_t)n_steps * h->n_act : (size_t)h->n_act), uh.begin());
  FCCHK(h->useq.upload(uh, h->stream));
  FCCHK(h->yseq.alloc((size_t)n_steps * ns));
  FCCHK(h->Eseq.alloc((size_t)n_steps));
  std::vector<double> yh((size_t)n_steps * ns), Eh(n_steps);
  int flag = 0;
  for (int s = 0; s < n_steps; ++s) {
    const int order = s == 0 ? first_order_slot : FC_SLOT_BDF2;
    const double* du = h->useq.p + (u_ctrl_is_sequence ? (size_t)s * h->n_act : 0);
    FCCHK(enqueue_step(h, order, du, h->yseq.p + (size_t)s * ns, h->Eseq.p + s, h->scal.p + 1, nullptr, compute_energy));
  }
  HIPCHK(hipMemcpyAsync(yh.data(), h->yseq.p, yh.size() * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipMemcpyAsync(Eh.data(), h->Eseq.p, Eh.size() * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipMemcpyAsync(&flag, h->flag.p, sizeof(int), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  FCCHK(time_collect(h));
  FCCHK(phase_collect(h));
  if (y_seq)
    for (int s = 0; s < n_steps; ++s)
      for (int k = 0; k < h->n_sens; ++k) y_seq[(size_t)s * h->n_sens + k] = yh[(size_t)s * ns + k];
  if (dE_seq)
    for (int s = 0; s < n_steps; ++s) dE_seq[s] = compute_energy ? Eh[s] : std::numeric_limits<double>::quiet_NaN();
  if (flag) return fail(FC_ERR_DIVERGED, "non-finite velocity after solve");
  return FC_OK;
}

int fc_assemble_rhs(fc_handle h, int order_slot, const double* u_ctrl, double* b_out) {
  FCCHK(check_step_ready(h, order_slot));
  if (!b_out || (h->n_act > 0 && !u_ctrl)) return fail(FC_ERR_INVALID, "fc_assemble_rhs: null argument");
  HIPCHK(hipSetDevice(h->device));
  FCCHK(quiesce(h));
  if (h->n_act) HIPCHK(hipMemcpyAsync(h->uctrl.p, u_ctrl, h->n_act * sizeof(double), hipMemcpyHostToDevice, h->stream));
  FCCHK(enqueue_rhs(h, order_slot, h->uctrl.p));
  hipLaunchKernelGGL(fc_scatter_perm, dim3(nblocks(h->N, 256)), dim3(256), 0, h->stream, h->N, h->perm.p, h->b.p,
                     (const double*)nullptr, h->tmpN.p);
  HIPCHK(hipMemcpyAsync(b_out, h->tmpN.p, (size_t)h->N * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  return FC_OK;
}

static int solve_once(fc_handle h, int slot, const double* b, double* x, double* info_out);

int fc_solve(fc_handle h, int slot, const double* b, double* x, double* info_out) {
  if (!h || slot < 0 || slot > 1 || !b || !x) return fail(FC_ERR_INVALID, "fc_solve: bad argument");
  return solve_once(h, slot, b, x, info_out);
}

static int solve_once(fc_handle h, int slot, const double* b, double* x, double* info_out) {
  OrderSys& S = h->sys[slot];
  if (!S.ready) return fail(FC_ERR_NOT_READY, "fc_solver_setup not called for this slot");
  if (S.factor_free && h->method == FC_METHOD_REFINE)
    return fail(FC_ERR_INVALID, "fc_solve: this slot has no factors (fc_setup_krylov): set FC_METHOD_GMRES or FC_METHOD_BICGSTAB");
  HIPCHK(hipSetDevice(h->device));
  FCCHK(quiesce(h));
  h->b.p = h->bstore.p;
  const int N = h->N, g = nblocks(N, 256);
  HIPCHK(hipMemcpyAsync(h->tmpN.p, b, (size_t)N * sizeof(double), hipMemcpyHostToDevice, h->stream));
  hipLaunchKernelGGL(fc_gather_perm, dim3(g), dim3(256), 0, h->stream, N, h->perm.p, h->tmpN.p, h->b.p);
  const bool dist = h->partitioned && exchanges(h);
  KrylovOverride inexact_factors(h, S);
  const bool krylov = h->method == FC_METHOD_BICGSTAB || h->method == FC_METHOD_GMRES;
  if (dist) {
    // every rank was handed the whole right-hand side: keep the rows this rank accounts for (its own, and the root's on
    // the lead rank -- the apply sums the root rows over the ranks; the Krylov vectors keep the root rows on every rank),
    // solve, then merge the ranks' parts of the solution
    hipLaunchKernelGGL(fc_mask_rows, dim3(g), dim3(256), 0, h->stream, N, h->rowkind_p.p, krylov ? 1 : (h->lead ? 1 : 0), h->b.p);
  }
  hipLaunchKernelGGL(fc_copy, dim3(g), dim3(256), 0, h->stream, N, h->b.p, h->buf.p);
  if (krylov) {
    if (h->partitioned && !dist) return fail(FC_ERR_INVALID, "fc_solve: partitioned handle without an exchange");
    int iters = 0;
    double relres = 0.0;
    const int code = h->method == FC_METHOD_GMRES ? gmres_permuted(h, S, &iters, &relres) : bicgstab_permuted(h, S, &iters, &relres);
    if (info_out) {
      info_out[0] = iters;
      info_out[1] = relres;
      info_out[2] = 0.0;
      info_out[3] = 0.0;
    }
    if (code != FC_OK) return code;
    if (dist) {
      hipLaunchKernelGGL(fc_mask_rows, dim3(g), dim3(256), 0, h->stream, N, h->rowkind_p.p, h->lead ? 1 : 0, h->kry.p);
      FCCHK(exchange(h, h->kry.p, (size_t)N));
    }
    hipLaunchKernelGGL(fc_scatter_perm, dim3(g), dim3(256), 0, h->stream, N, h->perm.p, h->kry.p, (const double*)nullptr, h->tmpN2.p);
    HIPCHK(hipMemcpyAsync(x, h->tmpN2.p, (size_t)N * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return FC_OK;
  }
  const double *xs = nullptr, *dx = nullptr;
  int nrp = 0;
  FCCHK(solve_permuted(h, S, &xs, &dx, &nrp));
  if (nrp > 0) hipLaunchKernelGGL(fc_reduce_final, dim3(2), dim3(256), 0, h->stream, nrp, h->partial.p, 1.0, h->scal.p + 1);
  if (dist) {
    // x (+ dx after refinement): this rank's rows and the (replicated) root rows are valid; |r|^2, |b|^2: this rank's rows
    if (dx) {
      hipLaunchKernelGGL(fc_axpy, dim3(g), dim3(256), 0, h->stream, N, 1.0, dx, h->xsol.p);
      hipLaunchKernelGGL(fc_copy, dim3(g), dim3(256), 0, h->stream, N, h->xsol.p, h->buf.p + N);
      xs = h->buf.p + N;
      dx = nullptr;
    }
    hipLaunchKernelGGL(fc_mask_rows, dim3(g), dim3(256), 0, h->stream, N, h->rowkind_p.p, h->lead ? 1 : 0, h->buf.p + N);
    FCCHK(exchange(h, h->buf.p + N, (size_t)N));
    if (nrp > 0) FCCHK(exchange(h, h->scal.p + 1, 2));
  }
  hipLaunchKernelGGL(fc_scatter_perm, dim3(g), dim3(256), 0, h->stream, N, h->perm.p, xs, dx, h->tmpN2.p);
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpyAsync(x, h->tmpN2.p, (size_t)N * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipMemcpyAsync(h->pin, h->scal.p, 8 * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  if (info_out) {
    info_out[0] = h->max_iter;
    info_out[1] = h->check_residual ? std::sqrt(h->pin[1] / (h->pin[2] > 0 ? h->pin[2] : 1.0)) : std::numeric_limits<double>::quiet_NaN();
    info_out[2] = h->check_residual ? std::sqrt(h->pin[2]) : std::numeric_limits<double>::quiet_NaN();
    info_out[3] = 0.0;
  }
  return FC_OK;
}

int fc_energy(fc_handle h, const double* u, double* E) {
  if (!h || !u || !E) return fail(FC_ERR_INVALID, "fc_energy: null argument");
  if (!h->slot_ok[FC_SLOT_MASS]) return fail(FC_ERR_NOT_READY, "FC_SLOT_MASS not assembled");
  HIPCHK(hipSetDevice(h->device));
  FCCHK(quiesce(h));
  HIPCHK(hipMemcpyAsync(h->tmpN.p, u, 2 * (size_t)h->nn * sizeof(double), hipMemcpyHostToDevice, h->stream));
  FCCHK(enqueue_energy(h, h->tmpN.p, h->scal.p + 3));
  HIPCHK(hipMemcpyAsync(h->pin, h->scal.p + 3, sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  *E = h->pin[0];
  return FC_OK;
}

int fc_measure(fc_handle h, const double* up, double* y) {
  if (!h || !up || (h->n_sens > 0 && !y)) return fail(FC_ERR_INVALID, "fc_measure: null argument");
  HIPCHK(hipSetDevice(h->device));
  FCCHK(quiesce(h));
  if (h->n_sens == 0) return FC_OK;
  HIPCHK(hipMemcpyAsync(h->tmpN.p, up, (size_t)h->N * sizeof(double), hipMemcpyHostToDevice, h->stream));
  hipLaunchKernelGGL(fc_sensors, dim3(h->n_sens), dim3(64), 0, h->stream, h->n_sens, h->s_rowptr.p, h->s_idx.p, h->s_w.p,
                     h->tmpN.p, h->ydev.p);
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpyAsync(h->pin, h->ydev.p, h->n_sens * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  for (int s = 0; s < h->n_sens; ++s) y[s] = h->pin[s];
  return FC_OK;
}

int fc_bench_sweeps(fc_handle h, int slot, int reps, double* ms_per_apply, int32_t* launches_per_apply) {
  if (!h || slot < 0 || slot > 1 || reps <= 0 || !ms_per_apply) return fail(FC_ERR_INVALID, "fc_bench_sweeps: bad argument");
  OrderSys& S = h->sys[slot];
  if (!S.ready) return fail(FC_ERR_NOT_READY, "fc_solver_setup not called for this slot");
  HIPCHK(hipSetDevice(h->device));
  FCCHK(quiesce(h));
  const int N = h->N, g = nblocks(N, 256);
  // rhs = last assembled b (any finite data); re-copied each rep so values stay bounded
  for (int i = 0; i < 2; ++i) {
    hipLaunchKernelGGL(fc_copy, dim3(g), dim3(256), 0, h->stream, N, h->b.p, h->buf.p);
    FCCHK(apply_factors(h, S));
  }
  HIPCHK(hipEventRecord(h->ev0, h->stream));
  for (int i = 0; i < reps; ++i) {
    hipLaunchKernelGGL(fc_copy, dim3(g), dim3(256), 0, h->stream, N, h->b.p, h->buf.p);
    FCCHK(apply_factors(h, S));
  }
  HIPCHK(hipEventRecord(h->ev1, h->stream));
  HIPCHK(hipEventSynchronize(h->ev1));
  float ms = 0.f;
  HIPCHK(hipEventElapsedTime(&ms, h->ev0, h->ev1));
  *ms_per_apply = (double)ms / reps;
  if (launches_per_apply) {
    const bool upc = up_column_wanted(h, S) && h->upc.ready && !S.truncated && S.bits == 64;
    int n = 0;
    for (const Stage& st : S.stages) n += (st.nrows > 0 && !(upc && st.kind == 0)) ? 1 : 0;
    if (upc)
      for (const fc_ctx::UpCol::Lvl& L : h->upc.lv) n += (L.count > 0 ? 1 : 0) + (L.fold_nrows > 0 ? 1 : 0);
    *launches_per_apply = n;
  }
  return FC_OK;
}

int fc_set_partition(fc_handle h, int32_t n_local_cells, const int32_t* local_cells, const uint8_t* rowkind, int lead) {
  if (h) {
    h->fvec_ok = false;
    h->pre_slot = h->bat.pre_slot = -1;  // a speculated element loop filled ev for the OLD cell list
  }
  if (!h || n_local_cells < 0 || (n_local_cells > 0 && !local_cells) || !rowkind)
    return fail(FC_ERR_INVALID, "fc_set_partition: bad argument");
  HIPCHK(hipSetDevice(h->device));
  FCCHK(quiesce(h));
  const int N = h->N, nc = h->nc;
  std::vector<unsigned char> mine(nc, 0);
  for (int k = 0; k < n_local_cells; ++k) {
    if (local_cells[k] < 0 || local_cells[k] >= nc || mine[local_cells[k]]) return fail(FC_ERR_INVALID, "fc_set_partition: bad cell list");
    mine[local_cells[k]] = 1;
  }
  for (int i = 0; i < N; ++i)
    if (rowkind[i] > 2) return fail(FC_ERR_INVALID, "fc_set_partition: rowkind must be 0, 1 or 2");
  h->h_cell_list.assign(local_cells, local_cells + n_local_cells);
  h->h_rowkind.assign(rowkind, rowkind + N);
  h->ncl = n_local_cells;
  h->lead = lead != 0;
  h->partitioned = true;
  std::vector<int> cl(std::max(1, n_local_cells), 0);
  std::copy(local_cells, local_cells + n_local_cells, cl.begin());
  FCCHK(h->cell_list.upload(cl, h->stream));
  {
    // cells whose element MATRICES this rank needs: every cell with a dof it owns or a root dof (its own cells, and the other ranks'
    // cells along the separator).  The element matrices of all other cells are zeroed once and never written again.
    std::vector<int> ac;
    for (int c = 0; c < nc; ++c) {
      bool need = false;
      for (int k = 0; k < 15 && !need; ++k) need = rowkind[h->h_cell_dofs[(size_t)c * 15 + k]] != 0;
      if (need) ac.push_back(c);
    }
    h->n_asm = (int)ac.size();
    if (ac.empty()) ac.push_back(0);
    FCCHK(h->asm_cells.upload(ac, h->stream));
    FCCHK(h->rowkind_w.upload(rowkind, (size_t)N, h->stream));
    FCCHK(h->em.zero(h->stream));
  }
  // element -> row gather lists restricted to this rank's cells (same fixed order as the serial lists)
  {
    std::vector<int> cnh((size_t)6 * nc);
    HIPCHK(hipMemcpyAsync(cnh.data(), h->cn.p, cnh.size() * sizeof(int), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    std::vector<int> gp(N + 1, 0);
    for (int c = 0; c < nc; ++c) {
      if (!mine[c]) continue;
      for (int k = 0; k < 12; ++k) gp[cnh[(size_t)(k % 6) * nc + c] + (k / 6) * h->nn + 1]++;
    }
    for (int r = 0; r < N; ++r) gp[r + 1] += gp[r];
    std::vector<int> gi(gp[N]);
    std::vector<int> fill(gp.begin(), gp.end() - 1);
    for (int c = 0; c < nc; ++c) {
      if (!mine[c]) continue;
      for (int k = 0; k < 12; ++k) gi[fill[cnh[(size_t)(k % 6) * nc + c] + (k / 6) * h->nn]++] = k * nc + c;
    }
    h->h_gptr.swap(gp);
    h->h_gidx.swap(gi);
  }
  HIPCHK(hipStreamSynchronize(h->stream));
  if (h->have_perm) FCCHK(refresh_permuted(h));
  FCCHK(upload_sensors(h));
  return FC_OK;
}

int fc_comm_unique_id(char* out128) {
  if (!out128) return fail(FC_ERR_INVALID, "fc_comm_unique_id: null argument");
  FCCHK(rccl_load());
  FcNcclId id;
  NCCLCHK(g_rccl.GetUniqueId(&id));
  std::memcpy(out128, id.internal, 128);
  return FC_OK;
}

// the handle becomes (or stops being) one rank of several: the elimination tree gets another root, so whatever was set up
// for the old role is void
static void forget_solver_structure(fc_ctx* h) {
  h->sym_ready = false;
  h->upc.ready = h->upc.tried = false;
  h->have_plan = false;
  h->bat.tables = h->bat.tb_built = false;
  h->bat.k = h->bat.KB = 0;
  for (int o = 0; o < 2; ++o) h->sys[o].ready = h->sys[o].structured = false;
}

int fc_comm_init(fc_handle h, int nranks, int rank, const char* id128) {
  if (!h || nranks < 1 || rank < 0 || rank >= nranks || !id128) return fail(FC_ERR_INVALID, "fc_comm_init: bad argument");
  FCCHK(rccl_load());
  HIPCHK(hipSetDevice(h->device));
  FcNcclId id;
  std::memcpy(id.internal, id128, 128);
  void* comm = nullptr;
  NCCLCHK(g_rccl.CommInitRank(&comm, nranks, id, rank));
  h->comm = comm;
  h->nranks = nranks;
  h->rank = rank;
  forget_solver_structure(h);
  return FC_OK;
}

// Can THIS rank create an RCCL communicator at all (library loadable, symbols there)?  Called by every rank before fc_comm_init so
// that the ranks can agree, over their own process group, to fall back to the host exchange TOGETHER: ncclCommInitRank is a
// collective, a rank that cannot even load RCCL would leave the others waiting inside it.  FC_RCCL_FAIL_RANK=<r> (test aid): rank r
// reports failure.
int fc_comm_probe(int rank) {
  if (const char* e = std::getenv("FC_RCCL_FAIL_RANK"); e && std::atoi(e) == rank)
    return fail(FC_ERR_HIP, "cannot load RCCL: disabled for this rank by FC_RCCL_FAIL_RANK");
  return rccl_load();
}

// drop the handle's RCCL communicator (a mixed outcome of fc_comm_init over the ranks: the ranks that did get one give it back and
// everybody goes on over the host exchange)
int fc_comm_destroy(fc_handle h) {
  if (!h) return fail(FC_ERR_INVALID, "null handle");
  if (h->comm) {
    HIPCHK(hipSetDevice(h->device));
    FCCHK(quiesce(h));
    if (g_rccl.CommDestroy) (void)g_rccl.CommDestroy(h->comm);
    h->comm = nullptr;
    h->nranks = 1;
    h->rank = 0;
    forget_solver_structure(h);
  }
  return FC_OK;
}

// what the handle's exchange actually is: transport 0 = none (single GPU), 1 = RCCL communicator — nranks / rank are then READ
// BACK from the communicator (ncclCommCount / ncclCommUserRank), not echoed from fc_comm_init —, 2 = host callback
int fc_comm_info(fc_handle h, int32_t* nranks, int32_t* rank, int32_t* transport) {
  if (!h) return fail(FC_ERR_INVALID, "null handle");
  int n = 1, r = 0, t = 0;
  if (h->comm) {
    t = 1;
    if (!g_rccl.CommCount || !g_rccl.CommUserRank) return fail(FC_ERR_HIP, "fc_comm_info: ncclCommCount / ncclCommUserRank not found");
    NCCLCHK(g_rccl.CommCount(h->comm, &n));
    NCCLCHK(g_rccl.CommUserRank(h->comm, &r));
  } else if (h->host_xchg) {
    t = 2;
    n = h->nranks;
    r = h->rank;
  }
  if (nranks) *nranks = n;
  if (rank) *rank = r;
  if (transport) *transport = t;
  return FC_OK;
}

int fc_set_host_exchange(fc_handle h, int nranks, int rank, fc_exchange_fn fn, void* user) {
  if (!h || nranks < 1 || rank < 0 || rank >= nranks) return fail(FC_ERR_INVALID, "fc_set_host_exchange: bad argument");
  if (h->comm) return fail(FC_ERR_INVALID, "fc_set_host_exchange: the handle already has an RCCL communicator");
  h->host_xchg = fn;
  h->host_xchg_user = user;
  h->nranks = nranks;
  h->rank = rank;
  forget_solver_structure(h);
  return FC_OK;
}

int fc_comm_selftest(fc_handle h, double* max_err_out) {
  if (!h) return fail(FC_ERR_INVALID, "null handle");
  if (max_err_out) *max_err_out = 0.0;
  if (!exchanges(h)) return FC_OK;  // a single-GPU handle has nothing to test
  HIPCHK(hipSetDevice(h->device));
  constexpr int n = 256;
  std::vector<double> v(n);
  for (int i = 0; i < n; ++i) v[i] = (double)(h->rank + 1) * (double)(i + 1);
  DevBuf<double> d;
  FCCHK(d.upload(v, h->stream));
  FCCHK(exchange(h, d.p, (size_t)n));
  HIPCHK(hipMemcpyAsync(v.data(), d.p, n * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  int nr = h->nranks;
  if (h->comm && g_rccl.CommCount) NCCLCHK(g_rccl.CommCount(h->comm, &nr));
  if (nr != h->nranks)
    return fail(FC_ERR_HIP, "fc_comm_selftest: the communicator holds " + std::to_string(nr) + " ranks, the handle was told " + std::to_string(h->nranks));
  const double tri = 0.5 * (double)nr * (double)(nr + 1);
  double worst = 0.0;
  int bad = -1;
  for (int i = 0; i < n; ++i) {
    const double e = std::fabs(v[i] - tri * (double)(i + 1));
    if (!(e <= worst)) {  // (a NaN lands here too)
      worst = e;
      if (bad < 0 && !(e == 0.0)) bad = i;
    }
  }
  if (max_err_out) *max_err_out = worst;
  if (bad >= 0)
    return fail(FC_ERR_HIP, "fc_comm_selftest: rank " + std::to_string(h->rank) + " of " + std::to_string(nr) + ": all-reduce of a known vector is wrong at entry " +
                                std::to_string(bad) + " (got " + std::to_string(v[bad]) + ", expected " + std::to_string(tri * (bad + 1)) + ")");
  return FC_OK;
}

int fc_set_phase_timing(fc_handle h, int on) {
  if (!h) return fail(FC_ERR_INVALID, "null handle");
  if (h->step_pending) return fail(FC_ERR_INVALID, "fc_set_phase_timing: a step is in flight");
  HIPCHK(hipSetDevice(h->device));
  FCCHK(quiesce(h));
  HIPCHK(hipStreamSynchronize(h->stream));
  h->phase_timing = on != 0;
  h->pused = 0;
  h->pre_slot = h->phase_timing ? -1 : h->pre_slot;  // a speculated element loop would fall outside the marks
  for (int p = 0; p < FC_N_PHASES; ++p) h->ph_us[p] = 0.0;
  h->ph_steps = 0;
  return FC_OK;
}

int fc_get_phase_timing(fc_handle h, double* us, int64_t* steps) {
  if (!h || !us) return fail(FC_ERR_INVALID, "fc_get_phase_timing: null argument");
  for (int p = 0; p < FC_N_PHASES; ++p) us[p] = h->ph_us[p];
  if (steps) *steps = h->ph_steps;
  return FC_OK;
}

int fc_set_timing(fc_handle h, int on) {
  if (!h) return fail(FC_ERR_INVALID, "null handle");
  HIPCHK(hipSetDevice(h->device));
  FCCHK(quiesce(h));
  HIPCHK(hipStreamSynchronize(h->stream));
  h->timing = on != 0;
  h->tused = 0;
  h->tkind.clear();
  h->tcount.clear();
  h->t_ms[0] = h->t_ms[1] = 0.0;
  h->t_cnt[0] = h->t_cnt[1] = 0;
  return FC_OK;
}

int fc_get_timing(fc_handle h, double* sweep_ms, int64_t* sweep_launches, double* spmv_ms, int64_t* spmv_launches) {
  if (!h) return fail(FC_ERR_INVALID, "null handle");
  if (sweep_ms) *sweep_ms = h->t_ms[0];
  if (sweep_launches) *sweep_launches = h->t_cnt[0];
  if (spmv_ms) *spmv_ms = h->t_ms[1];
  if (spmv_launches) *spmv_launches = h->t_cnt[1];
  return FC_OK;
}

int fc_algorithmic_bytes(fc_handle h, int slot, double* sweep_bytes, double* spmv_bytes) {
  if (!h || slot < 0 || slot > 1) return fail(FC_ERR_INVALID, "fc_algorithmic_bytes: bad argument");
  if (sweep_bytes) *sweep_bytes = h->sys[slot].sweep_bytes;
  // the in-step SpMV runs on the permuted system matrix (same nnz as the slot up to explicit zeros)
  const double nz = h->sys[slot].Ap_nnz > 0 ? (double)h->sys[slot].Ap_nnz : (double)h->nnz;
  if (spmv_bytes) *spmv_bytes = nz * 12.0 + (double)h->N * 16.0 + (double)(h->N + 1) * 4.0;
  return FC_OK;
}

// storage width of the factor values of every slot set up from now on: 64 = exact selected inverse (default), 32 / 16 =
// COMPRESSED factors (fp32 / bfloat16: 50 % / 25 % of the memory; never materialised in fp64).  Compressed factors are a
// preconditioner: fc_solve / fc_step then need FC_METHOD_GMRES or FC_METHOD_BICGSTAB.
int fc_set_factor_precision(fc_handle h, int bits) {
  if (!h || (bits != 64 && bits != 32 && bits != 16)) return fail(FC_ERR_INVALID, "fc_set_factor_precision: bits must be 64, 32 or 16");
  if (bits != h->factor_bits) {
    h->factor_bits = bits;
    for (int o = 0; o < 2; ++o) h->sys[o].ready = h->sys[o].structured = false;  // the value arrays are laid out anew
    h->bat.ftile_ok[0] = h->bat.ftile_ok[1] = false;
  }
  return FC_OK;
}

int fc_get_factor_storage(fc_handle h, int slot, int32_t* bits, int64_t* bytes) {
  if (!h || slot < 0 || slot > 1) return fail(FC_ERR_INVALID, "fc_get_factor_storage: bad argument");
  const OrderSys& S = h->sys[slot];
  if (bits) *bits = S.bits;
  if (bytes) *bytes = (int64_t)(S.f_val.n * sizeof(double) + S.f_val32.n * sizeof(float) + S.f_val16.n * sizeof(FcBf16));
  return FC_OK;
}

// ── base-flow (steady-state) iterations behind the C ABI: SteadyStateSolver.picard / .newton (steadystate.py:60-159) ──────
// Every iteration assembles its operator with the HIP element loop, eliminates the Dirichlet dofs, factorises on the device
// and solves with the sweep kernels; the host side here only forms the residual and the update (vectors cross the
// boundary as host arrays: this is the setup path).  Slots used: FC_SLOT_SCRATCH (operator of the residual), FC_SLOT_BDF1
// (the iteration's system matrix and its factors) — assemble the time-stepping operators afterwards.
int fc_set_baseflow_bc(fc_handle h, int32_t n_bc, const int32_t* bc_dofs, const double* bc_values) {
  if (!h || n_bc < 0 || (n_bc > 0 && (!bc_dofs || !bc_values))) return fail(FC_ERR_INVALID, "fc_set_baseflow_bc: bad argument");
  h->ss_dofs.assign(bc_dofs, bc_dofs + n_bc);
  h->ss_vals.assign(bc_values, bc_values + n_bc);
  // the increments of both iterations vanish on the Dirichlet dofs: homogeneous symmetric elimination, no lifting
  std::vector<double> zero((size_t)std::max(1, n_bc), 0.0);
  return fc_set_bc(h, n_bc, bc_dofs, 1, zero.data());
}

static int steady_solve_increment(fc_ctx* h, double nu, const double* adv, const double* lin, const std::vector<double>& rhs, std::vector<double>& delta) {
  FCCHK(fc_assemble_matrix(h, FC_SLOT_BDF1, 0.0, nu, adv, 1.0, lin, 1.0, -1.0, -1.0));
  FCCHK(fc_apply_bc(h, FC_SLOT_BDF1));
  FCCHK(fc_setup_solver(h, FC_SLOT_BDF1, 0, 2, 0, 2, 1));  // first call: + symbolic phase; later: numeric factorisation only
  double info[4];
  delta.assign((size_t)h->N, 0.0);
  FCCHK(fc_solve(h, FC_SLOT_BDF1, rhs.data(), delta.data(), info));
  if (!(info[1] < 1e-8)) return fail(FC_ERR_NOT_CONVERGED, "base-flow iteration: linear solve residual " + std::to_string(info[1]));
  return FC_OK;
}

int fc_picard_step(fc_handle h, double nu, double* up, const double* load, double* rel_change) {
  if (!h || !up || !(nu > 0.0)) return fail(FC_ERR_INVALID, "fc_picard_step: bad argument");
  const int N = h->N, nn2 = 2 * h->nn;
  // Oseen operator with the advecting velocity frozen at the iterate (nsforms.py:149-187); x~ = iterate with the boundary
  // values imposed; A d = load - A x~ on the free rows, d = 0 on the Dirichlet dofs; next iterate = x~ + d
  std::vector<double> adv(up, up + nn2), xt(up, up + N), r((size_t)N), d;
  for (size_t k = 0; k < h->ss_dofs.size(); ++k) xt[(size_t)h->ss_dofs[k]] = h->ss_vals[k];
  FCCHK(fc_assemble_matrix(h, FC_SLOT_SCRATCH, 0.0, nu, adv.data(), 1.0, nullptr, 1.0, -1.0, -1.0));
  FCCHK(fc_spmv(h, FC_SLOT_SCRATCH, xt.data(), r.data()));
  for (int i = 0; i < N; ++i) r[(size_t)i] = (load ? load[i] : 0.0) - r[(size_t)i];
  for (int dof : h->ss_dofs) r[(size_t)dof] = 0.0;
  FCCHK(steady_solve_increment(h, nu, adv.data(), nullptr, r, d));
  double diff = 0.0, base = 0.0;
  for (int i = 0; i < N; ++i) {
    const double nw = xt[(size_t)i] + d[(size_t)i];
    diff += (nw - up[i]) * (nw - up[i]);
    base += up[i] * up[i];
    up[i] = nw;
  }
  if (rel_change) *rel_change = std::sqrt(diff) / (std::sqrt(base) + 1e-14);
  return FC_OK;
}

int fc_newton_step(fc_handle h, double nu, double* up, const double* load, double* res_norm, int update) {
  if (!h || !up || !(nu > 0.0)) return fail(FC_ERR_INVALID, "fc_newton_step: bad argument");
  const int N = h->N, nn2 = 2 * h->nn;
  for (size_t k = 0; k < h->ss_dofs.size(); ++k) up[(size_t)h->ss_dofs[k]] = h->ss_vals[k];
  // residual of the steady equations: ((U.grad)U, v) + nu (grad U, grad v) - (P, div v) - (q, div U) - load (nsforms.py:116-147)
  std::vector<double> u(up, up + nn2), F((size_t)N), d;
  FCCHK(fc_assemble_matrix(h, FC_SLOT_SCRATCH, 0.0, nu, u.data(), 1.0, nullptr, 1.0, -1.0, -1.0));
  FCCHK(fc_spmv(h, FC_SLOT_SCRATCH, up, F.data()));
  if (load)
    for (int i = 0; i < N; ++i) F[(size_t)i] -= load[i];
  for (int dof : h->ss_dofs) F[(size_t)dof] = 0.0;
  double r2 = 0.0;
  for (double v : F) r2 += v * v;
  if (res_norm) *res_norm = std::sqrt(r2);
  if (!update) return FC_OK;
  FCCHK(steady_solve_increment(h, nu, u.data(), u.data(), F, d));  // Jacobian: advecting AND linearised field = the iterate
  for (int i = 0; i < N; ++i) up[i] -= d[(size_t)i];
  return FC_OK;
}

// ── shared-operator batched stepping (fc_batch.hip.h): k lock-step simulations per handle ─────────────────────────────
// Replaces k independent FlowSolver instances stepping the SAME operator (IC sweeps
// examples/lidcavity/batch_run_lidcavity.py:197-215, controller optimisation utils/optim.py:95-102).
constexpr int kRecStride = 160;  // doubles per simulation in the host-mapped record: the single-simulation layout, repeated

// launch tables of the batched factor apply from the symbolic phase of fc_setup_solver: per tree level one block launch
// (-L blocks on the way up, [D^-1 | -U] blocks on the way down) and, on the way up, one fold launch
static inline double* bat_slot(const fc_ctx* h, int k) { return h->bat.ring.p + (size_t)(((k % 4) + 4) % 4) * h->bat.slot_doubles; }
static inline double* bat_n(const fc_ctx* h) { return bat_slot(h, h->bat.cur) + (size_t)h->N * h->bat.KB; }       // (u_n, p_n) of all simulations
static inline double* bat_nn(const fc_ctx* h) { return bat_slot(h, h->bat.cur + 3) + (size_t)h->N * h->bat.KB; }  // u_nn
static inline void bat_point(fc_ctx* h) { h->bat.buf.p = bat_slot(h, h->bat.cur + 1); }

// row blocks of the batched tail over the permuted pattern (the same for both slots): tabulated per batch width, because a block's
// solution rows [column][KB] sit in LDS
static int build_tail_blocks(fc_ctx* h, int KB) {
  fc_ctx::Batch& B = h->bat;
  const int N = h->N;
  const std::vector<int>& rp = h->sym_plan.Ap_rowptr;
  const std::vector<int>& cl = h->sym_plan.Ap_col;
  // width of a row block's column set.  Alone on the device the tail is fastest with many small blocks (128 columns: ~8 rows per block on
  // O1, k = 16: 45.8 us against 53.8 with 256), but the default batched step runs it on the second stream BESIDE the next step's
  // launches, and there the 256-column blocks (~14.5 rows, half the operand traffic, a third fewer workgroups) leave more of the machine
  // to the main stream: k = 16 75.0 -> 83.7 k simulated steps/s on O1, k = 8 48.4 -> 50.0 (same box; 192 / 320 / 448: 82.0 / 83.0 / 80.6).
  // 384 x 16 simulations x 8 B + the staged matrix entries (FC_TB_NNZ x 10 B) + the static 4 KB stay below the 64 KB a launch gets without opting in.
  int want;
  {
    const char* e = std::getenv("FC_TB_COLS");  // tuning aid
    bool streams = false;  // any slot whose factors stream from HBM: the batched step stays on one stream there (step_batch_begin)
    for (int o = 0; o < 2; ++o) streams = streams || (h->sys[o].structured && h->sys[o].nt);
    // (32 simulations, [3,3,2,2] tree of O1: 144 columns 118.2 k simulated steps/s, 128: 116.5 k, 224: 113.3 k, 112: 97.5 k -- the LDS a block
    //  takes decides how many of them sit beside the main stream's workgroups)
    const int v = e ? std::atoi(e) : ((h->overlap && !streams) ? (KB > 16 ? 144 : 256) : FC_TB_COLS);
    want = std::min(384 * 16 / std::max(16, KB), std::max(FC_TB_ROWS, v));  // (KB = 32: at most 192 columns)
  }
  if (B.tblocks.p && B.tb_cols == want && B.tb_built) return FC_OK;
  B.tb_cols = want;
  const size_t tb_cols = (size_t)B.tb_cols;
  // The residual only needs sums over ALL rows, so the tail is free to visit them in an order of its own: cell by cell (the cells are
  // Morton-ordered), a node's two velocity rows next to each other.  Rows that are neighbours in the solver's permuted numbering share
  // few columns (inside a tree node the dofs are in index order: all x-velocities, then all y-velocities, then the pressures --
  // 16 such rows touch ~280 distinct columns, i.e. every staged solution row served 1.7 matrix rows); 16 rows of two or three
  // neighbouring cells touch ~90.
  std::vector<int> ip((size_t)N), rorder;
  rorder.reserve((size_t)N);
  for (int i = 0; i < N; ++i) ip[(size_t)h->h_perm[(size_t)i]] = i;
  {
    std::vector<unsigned char> seen((size_t)N, 0);
    auto take = [&](int w) {
      if (!seen[(size_t)w]) seen[(size_t)w] = 1, rorder.push_back(ip[(size_t)w]);
    };
    for (int c = 0; c < h->nc; ++c) {
      const int* cd = h->h_cell_dofs.data() + (size_t)c * 15;
      for (int a = 0; a < 6; ++a) take(cd[a]), take(cd[6 + a]);
      for (int a = 12; a < 15; ++a) take(cd[a]);
    }
    for (int w = 0; w < N; ++w) take(w);  // (dofs of no cell: none on a conforming mesh)
  }
  std::vector<FcTBlock> tb;
  std::vector<int> tcols, trowd;  // trowd: one int4 per row (FcTBlock)
  std::vector<unsigned short> lidx(cl.size(), 0);
  std::vector<int> mark((size_t)N, -1), cur;
  trowd.reserve(4 * (size_t)N);
  size_t q0 = 0;
  while (q0 < (size_t)N) {
    cur.clear();
    size_t q = q0;
    int nnz = 0;
    for (; q < (size_t)N && q - q0 < FC_TB_ROWS; ++q) {
      const int r = rorder[q];
      const int len = rp[(size_t)r + 1] - rp[(size_t)r];
      if (q > q0 && nnz + len > FC_TB_NNZ) break;  // the block's entries are staged in LDS
      size_t added = 0;
      for (int k = rp[(size_t)r]; k < rp[(size_t)r + 1]; ++k)
        if (mark[(size_t)cl[(size_t)k]] != (int)q0) {
          mark[(size_t)cl[(size_t)k]] = (int)q0;
          cur.push_back(cl[(size_t)k]);
          ++added;
        }
      if (cur.size() > tb_cols && q > q0) {  // this row does not fit any more: it starts the next block
        for (size_t u = 0; u < added; ++u) mark[(size_t)cur[cur.size() - 1 - u]] = -1;
        cur.resize(cur.size() - added);
        break;
      }
      nnz += len;
    }
    if (cur.size() > tb_cols || nnz > FC_TB_NNZ) return fail(FC_ERR_INVALID, "fc_set_batch: a matrix row has more entries than a row block holds");
    std::sort(cur.begin(), cur.end());
    const int c0 = (int)tcols.size();
    for (size_t u = 0; u < cur.size(); ++u) mark[(size_t)cur[u]] = -2 - (int)u;  // local position
    int off = 0;
    for (size_t u = q0; u < q; ++u) {
      const int r = rorder[u];
      trowd.push_back(r), trowd.push_back(rp[(size_t)r]), trowd.push_back((rp[(size_t)r + 1] - rp[(size_t)r]) | ((h->h_perm[(size_t)r] < 2 * h->nn ? 1 : 0) << 16)), trowd.push_back(off);
      off += rp[(size_t)r + 1] - rp[(size_t)r];
      for (int k = rp[(size_t)r]; k < rp[(size_t)r + 1]; ++k) lidx[(size_t)k] = (unsigned short)(-2 - mark[(size_t)cl[(size_t)k]]);
    }
    for (int c : cur) mark[(size_t)c] = -1;
    tcols.insert(tcols.end(), cur.begin(), cur.end());
    tb.push_back(FcTBlock{(int)q0, (int)(q - q0), c0, (int)cur.size()});
    q0 = q;
  }
  if (tcols.empty()) tcols.push_back(0);
  if (lidx.empty()) lidx.push_back(0);
  B.n_tblocks = (int)tb.size();
  FCCHK(B.tblocks.upload(tb, h->stream));
  FCCHK(B.tcols.upload(tcols, h->stream));
  FCCHK(B.tlidx.upload(lidx, h->stream));
  FCCHK(B.trowd.upload(trowd, h->stream));
  B.tb_built = true;
  HIPCHK(hipStreamSynchronize(h->stream));
  return FC_OK;
}

static int build_batch_tables(fc_ctx* h) {
  fc_ctx::Batch& B = h->bat;
  if (B.tables) return FC_OK;
  if (!h->sym_ready) return fail(FC_ERR_NOT_READY, "fc_set_batch: fc_setup_solver must have run (batched stepping uses the in-library analysis)");
  if (h->partitioned || h->sym_truncate > 0) return fail(FC_ERR_INVALID, "fc_set_batch: single-GPU handles with full factors only");
  const fcsym::Factors& fac = h->sym_fac;
  const fcsym::Tree& t = h->sym_tree;
  const int N = h->N;
  const size_t G = fac.nodes.size() / 7;
  auto nd = [&](size_t g, int f) { return fac.nodes[g * 7 + (size_t)f]; };  // level, n, i0, ni, nb, voff, ioff
  std::vector<int64_t> soff(G, 0);
  int64_t S = 0;
  for (size_t g = 0; g < G; ++g) {
    soff[g] = S;
    S += nd(g, 4);
  }
  if (2 * (int64_t)N + S > (int64_t)std::numeric_limits<int>::max() / 32) return fail(FC_ERR_INVALID, "fc_set_batch: mesh too large for int32 buffer rows");
  // fold lists: destination row -> scratch rows, nodes in elimination order (deepest first): the order of the
  // single-simulation segment lists
  std::vector<int> fptr((size_t)N + 1, 0);
  for (size_t g = 0; g < G; ++g)
    for (int64_t j = 0; j < nd(g, 4); ++j) fptr[(size_t)(fac.idx[(size_t)(nd(g, 6) + j)] - N) + 1]++;
  for (int i = 0; i < N; ++i) fptr[(size_t)i + 1] += fptr[(size_t)i];
  std::vector<int> fsrc((size_t)std::max(1, fptr[(size_t)N]));
  {
    std::vector<int> fill(fptr.begin(), fptr.end() - 1);
    for (size_t g = 0; g < G; ++g)
      for (int64_t j = 0; j < nd(g, 4); ++j) fsrc[(size_t)fill[(size_t)(fac.idx[(size_t)(nd(g, 6) + j)] - N)]++] = (int)(2 * (int64_t)N + soff[g] + j);
  }
  // operand row lists: node g's columns [0, ni) are its own y rows, [ni, nf) the x rows of its boundary; padded to a
  // multiple of 32 with the index of the buffer's zero row (the row behind the scratch rows: never written); olist[0..8)
  // is the "null group" the block kernel reads for column groups past the end of a block
  const int zero_row = (int)(2 * (int64_t)N + S);
  std::vector<int> olist(32, zero_row);
  std::vector<int> ooff(G, 0), ooff_up(G, 0);
  for (size_t g = 0; g < G; ++g) {
    ooff[g] = (int)olist.size();
    for (int64_t c = 0; c < nd(g, 3); ++c) olist.push_back((int)(nd(g, 2) + c));
    for (int64_t j = 0; j < nd(g, 4); ++j) olist.push_back(fac.idx[(size_t)(nd(g, 6) + j)]);
    while (olist.size() & 31) olist.push_back(zero_row);
    // the -L block's operand is the node's own y rows only: its list must END with the padding
    ooff_up[g] = ooff[g];
    if (nd(g, 4) > 0 && (nd(g, 3) & 31)) {
      ooff_up[g] = (int)olist.size();
      for (int64_t c = 0; c < nd(g, 3); ++c) olist.push_back((int)(nd(g, 2) + c));
      while (olist.size() & 31) olist.push_back(zero_row);
    }
  }
  std::vector<FcBTask> tasks;
  int64_t tiled_off = 0;
  B.pslots = 0;
  B.launches.clear();
  B.factor_values = 0;
  B.vec_rows = 0.0;
  static const int force_cg = [] { const char* e = std::getenv("FC_BATCH_CG"); return e ? std::atoi(e) : 0; }();
  static const double chunks_per_wave = [] { const char* e = std::getenv("FC_BATCH_CPW"); return e ? std::max(0.5, std::atof(e)) : 1.5; }();
  // one block launch for all nodes of `level`: up = the -L blocks (nb x ni), else the [D^-1 | -U] rows (ni x nf);
  // one task per 16-row tile, cg waves per tile (about chunks_per_wave 32-column chunks per wave)
  auto emit = [&](int level, bool up) {
    std::vector<size_t> sel;
    double cols_w = 0.0, w = 0.0;
    for (size_t g = 0; g < G; ++g)
      if (nd(g, 0) == level && nd(g, 3) > 0 && (!up || nd(g, 4) > 0)) {
        sel.push_back(g);
        const double rows = (double)(up ? nd(g, 4) : nd(g, 3)), cols = (double)(up ? nd(g, 3) : nd(g, 3) + nd(g, 4));
        cols_w += rows * cols;  // mean columns, weighted by the values behind them
        w += rows;
      }
    if (sel.empty()) return;
    const double mean_chunks = cols_w / std::max(w, 1.0) / 32.0;
    int cg = 1;
    while (cg < 16 && mean_chunks / cg > chunks_per_wave) cg *= 2;
    if (force_cg == 1 || force_cg == 2 || force_cg == 4 || force_cg == 8 || force_cg == 16) cg = force_cg;
    const int first = (int)tasks.size();
    // XCD placement (speed only): workgroups b and b + 8 share an XCD and its L2 (MI355X_MICROARCH.md, Workgroup dispatch),
    // and all tiles of a node read the same operand rows — so a node's tiles (for the few big nodes near the root: a run of
    // them) go to ONE of eight classes, and position p of the launch takes the next task of class p % 8: the node's operand
    // crosses the fabric once per XCD that hosts it instead of once per tile.  FC_BATCH_XCD=0: node-major order.
    static const bool xcd_order = [] { const char* e = std::getenv("FC_BATCH_XCD"); return !(e && e[0] == '0'); }();
    std::vector<FcBTask> cls[8];
    // Wide tiles are cut into parts of about `split_chunks` 32-column chunks, one workgroup each (the kernel's last arriver adds a tile's
    // parts in part order): a launch with fewer 16-row tiles than compute units -- O1's root is ONE node of 143 tiles x 72 chunks -- or with
    // a few more than a whole number of rounds of 1024-thread workgroups otherwise streams at half the rate of the others.
    // FC_BATCH_SPLIT=<chunks> (0: never split)
    static const int split_chunks = [] { const char* e = std::getenv("FC_BATCH_SPLIT"); return e ? std::max(0, std::atoi(e)) : 16; }();
    auto push_parts = [&](const FcBTask& whole, std::vector<FcBTask>& out) {
      const int nchunk = (whole.ncols + 31) / 32;
      const int parts = split_chunks > 0 && 2 * nchunk >= 3 * split_chunks ? std::min(255, (nchunk + split_chunks / 2) / split_chunks) : 1;
      if (parts <= 1) {
        out.push_back(whole);
        return;
      }
      for (int q = 0; q < parts; ++q) {
        const int c0 = (int)((int64_t)nchunk * q / parts), c1 = (int)((int64_t)nchunk * (q + 1) / parts);
        FcBTask tk = whole;
        tk.src += 32 * (int64_t)c0;
        tk.val += 512 * (int64_t)c0;
        tk.op += 32 * c0;
        tk.ncols = std::min(whole.ncols, 32 * c1) - 32 * c0;
        tk.split = q | (parts << 8);
        tk.pslot = (int)B.pslots;
        out.push_back(tk);
      }
      B.pslots += parts;
    };
    int64_t tiles_total = 0;
    for (size_t g : sel) tiles_total += ((up ? nd(g, 4) : nd(g, 3)) + 15) / 16;
    const int64_t run_max = std::max<int64_t>(1, (tiles_total + 7) / 8);  // a node with more tiles is cut into runs of this many
    for (size_t g : sel) {
      const int64_t i0 = nd(g, 2), ni = nd(g, 3), nb = nd(g, 4), voff = nd(g, 5), nf = ni + nb;
      const int64_t rows = up ? nb : ni, ld = up ? ni : nf, base = up ? voff + ni * nf : voff;
      int64_t in_run = 0;
      int c = 0;
      for (int64_t r0 = 0; r0 < rows; r0 += 16) {
        FcBTask tk = {};
        tk.src = base + r0 * ld;
        tk.val = tiled_off;
        tiled_off += 512 * ((ld + 31) / 32);
        tk.ld = (int)ld;
        tk.nrows = (int)std::min<int64_t>(16, rows - r0);
        tk.ncols = (int)ld;
        tk.op = up ? ooff_up[g] : ooff[g];
        tk.dst = up ? (int)(2 * (int64_t)N + soff[g] + r0) : (int)(N + i0 + r0);
        if (!xcd_order) {
          push_parts(tk, tasks);
          continue;
        }
        if (in_run == 0) {  // a new run: the class with the fewest tasks so far
          c = 0;
          for (int q = 1; q < 8; ++q)
            if (cls[q].size() < cls[c].size()) c = q;
        }
        push_parts(tk, cls[c]);
        if (++in_run == run_max) in_run = 0;
      }
      B.factor_values += rows * ld;
      B.vec_rows += (double)ld + (double)rows;
    }
    if (xcd_order) {
      size_t pos[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      size_t left = 0;
      for (int q = 0; q < 8; ++q) left += cls[q].size();
      for (size_t p = 0; left > 0; ++p) {
        int c = (int)(p & 7);
        if (pos[c] >= cls[c].size()) {  // this class is exhausted: take from the one with the most tasks left
          size_t best = 0;
          for (int q = 0; q < 8; ++q)
            if (cls[q].size() - pos[q] > best) best = cls[q].size() - pos[q], c = q;
        }
        tasks.push_back(cls[c][pos[c]++]);
        --left;
      }
    }
    // (column-group waves of the launch: from the chunks of its TASKS -- a part of a split tile is a task)
    double tcols = 0.0, trows = 0.0;
    for (size_t q = (size_t)first; q < tasks.size(); ++q) tcols += (double)tasks[q].nrows * tasks[q].ncols, trows += (double)tasks[q].nrows;
    const double task_chunks = tcols / std::max(trows, 1.0) / 32.0;
    cg = 1;
    while (cg < 16 && task_chunks / cg > chunks_per_wave) cg *= 2;
    if (force_cg == 1 || force_cg == 2 || force_cg == 4 || force_cg == 8 || force_cg == 16) cg = force_cg;
    bool any_split = false;
    for (size_t q = (size_t)first; q < tasks.size(); ++q) any_split = any_split || tasks[q].split != 0;
    B.launches.push_back({0, first, (int)tasks.size() - first, cg, any_split ? 1 : 0, 0, 0, 0});  // (row0 of a block launch: it has split tiles)
    B.launches.back().mean_chunks = task_chunks;
  };
  for (int k = t.depth; k >= 1; --k) {
    emit(k, true);
    const int64_t r0 = t.node_ptr[(size_t)k - 1].front(), r1 = t.node_ptr[(size_t)k - 1].back();
    if (r1 > r0) {
      B.launches.push_back({1, 0, 0, 0, (int)r0, (int)(r1 - r0), 0, 1});
      B.vec_rows += 2.0 * (double)(r1 - r0) + (double)(fptr[(size_t)r1] - fptr[(size_t)r0]);
    }
  }
  for (int k = 0; k <= t.depth; ++k) emit(k, false);
  if (tasks.empty()) return fail(FC_ERR_INVALID, "fc_set_batch: empty factor structure");
  B.scratch_rows = S;
  B.tiled_values = tiled_off;
  B.ftile_ok[0] = B.ftile_ok[1] = false;
  FCCHK(B.tasks.upload(tasks, h->stream));
  FCCHK(B.part.alloc((size_t)std::max<int64_t>(1, B.pslots) * 512));
  FCCHK(B.ticket.alloc((size_t)std::max<int64_t>(1, B.pslots)));
  FCCHK(B.ticket.zero(h->stream));
  FCCHK(B.fptr.upload(fptr, h->stream));
  FCCHK(B.fsrc.upload(fsrc, h->stream));
  FCCHK(B.olist.upload(olist, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  B.tables = true;
  return FC_OK;
}

#define FC_KB_DISPATCH(KBV, CALL4, CALL8, CALL16, CALL32) \
  do {                                            \
    if ((KBV) == 4) {                             \
      CALL4;                                      \
    } else if ((KBV) == 8) {                      \
      CALL8;                                      \
    } else if ((KBV) == 16) {                     \
      CALL16;                                     \
    } else {                                      \
      CALL32;                                     \
    }                                             \
  } while (0)

// tiled copy of the slot's factor values for the batched block kernel (after every numeric factorisation)
static int batch_repack(fc_ctx* h, int slot) {
  fc_ctx::Batch& B = h->bat;
  if (!B.tables || B.tasks.n == 0) return FC_OK;
  OrderSys& S = h->sys[slot];
  if (!S.structured || S.f_val.n == 0 || S.bits != 64) return FC_OK;  // batched steps apply exact (fp64) factors
  if (B.ftile[slot].n != (size_t)B.tiled_values) FCCHK(B.ftile[slot].alloc((size_t)B.tiled_values));
  hipLaunchKernelGGL(fc_b_repack, dim3((unsigned)B.tasks.n), dim3(256), 0, h->stream, B.tasks.p, S.f_val.p, B.ftile[slot].p);
  HIPCHK(hipGetLastError());
  B.ftile_ok[slot] = true;
  return FC_OK;
}

// x (rows N .. 2N of bat.buf) = M^-1 y (rows 0 .. N) for all KB columns
static int batch_apply(fc_ctx* h, int slot, bool check = false) {
  fc_ctx::Batch& B = h->bat;
  const unsigned char* vr = check ? h->velrow_p.p : nullptr;  // overlapped tail: the down tiles test what they write for finiteness
  if (!B.ftile_ok[slot]) return fail(FC_ERR_NOT_READY, "batched apply: the slot's factors have no tiled copy (fc_set_batch after fc_setup_solver)");
  double* buf = B.buf.p;
  const double* tiled = B.ftile[slot].p;
  const bool nt = h->sys[slot].nt;
  FCCHK(time_begin(h, 0, (int)B.launches.size()));
  // column-group waves per tile: about 1.5 chunks of 32 columns per wave, 3 at 32 simulations (two accumulators: a wave does twice the
  // matrix work per chunk it loads; k = 32 109.7 -> 113.7 k simulated steps/s on O1, k = 16 unchanged or worse with more)
  static const double cpw_env = [] { const char* e = std::getenv("FC_BATCH_CPW"); return e ? std::max(0.5, std::atof(e)) : 0.0; }();
  static const int cg_env = [] { const char* e = std::getenv("FC_BATCH_CG"); return e ? std::atoi(e) : 0; }();
  const double cpw = cpw_env > 0.0 ? cpw_env : (B.KB > 16 ? 3.0 : 1.5);
  for (fc_ctx::BLaunch L : B.launches) {
    if (L.kind == 0) {
      if (!(cg_env == 1 || cg_env == 2 || cg_env == 4 || cg_env == 8 || cg_env == 16)) {
        // (launches with split tiles: 3-4 chunks per wave -- their parts are long enough for the register pipeline to reach its steady state)
        static const double split_cpw = [] { const char* e = std::getenv("FC_BATCH_SPLIT_CPW"); return e ? std::max(0.5, std::atof(e)) : 0.0; }();
        const double want = !L.row0 ? cpw : (split_cpw > 0.0 ? split_cpw : (B.KB > 16 ? 4.0 : 3.0));
        int cg = 1;
        while (cg < 16 && L.mean_chunks / cg > want) cg *= 2;
        L.cg = cg;
      }
      const FcBTask* tp = B.tasks.p + L.first;
#define FC_BLK(K) \
  do { \
    const size_t lds = L.cg > 1 ? (size_t)L.cg * 2048 * (K > 16 ? 2 : 1) : 0; \
    if (nt) \
      hipLaunchKernelGGL((fc_nd_block_b<K, true>), dim3(L.count), dim3(64 * L.cg), lds, h->stream, tp, B.olist.p, tiled, buf, L.cg, vr, h->N, B.flag.p, B.part.p, B.ticket.p); \
    else \
      hipLaunchKernelGGL((fc_nd_block_b<K>), dim3(L.count), dim3(64 * L.cg), lds, h->stream, tp, B.olist.p, tiled, buf, L.cg, vr, h->N, B.flag.p, B.part.p, B.ticket.p); \
  } while (0)
      FC_KB_DISPATCH(B.KB, FC_BLK(4), FC_BLK(8), FC_BLK(16), FC_BLK(32));
#undef FC_BLK
    } else {
      const int g = nblocks((int64_t)L.nrows * B.KB, 256);
#define FC_FOLD(K) hipLaunchKernelGGL((fc_nd_fold_b<K>), dim3(g), dim3(256), 0, h->stream, L.nrows, L.row0, B.fptr.p, B.fsrc.p, buf, L.dst_off, L.accumulate)
      FC_KB_DISPATCH(B.KB, FC_FOLD(4), FC_FOLD(8), FC_FOLD(16), FC_FOLD(32));
#undef FC_FOLD
    }
  }
  FCCHK(time_end(h));
  HIPCHK(hipGetLastError());
  return FC_OK;
}

int fc_set_batch(fc_handle h, int32_t k) {
  if (!h || k < 0 || k > 32) return fail(FC_ERR_INVALID, "fc_set_batch: k must be in [0, 32]");
  if (h->bat.pending || h->step_pending) return fail(FC_ERR_INVALID, "fc_set_batch: a step is in flight");
  HIPCHK(hipSetDevice(h->device));
  FCCHK(quiesce(h));
  fc_ctx::Batch& B = h->bat;
  if (k == 0) {
    HIPCHK(hipStreamSynchronize(h->stream));
    batch_drop_graphs(h);
    B.ring.release(), B.bstore.release(), B.ev.release(), B.partial.release();
    B.b = fc_ctx::BufView{};
    B.buf = fc_ctx::BufView{};
    B.flag.release();
    // ... and everything fc_set_batch(k > 0) tabulates: the tiled copies of both slots' factors (each >= the factor size), task
    // lists, fold lists, tail blocks -- a handle that goes back to single-run stepping must not keep holding them, nor pay a
    // repack per fc_refactor
    B.ftile[0].release(), B.ftile[1].release();
    B.ftile_ok[0] = B.ftile_ok[1] = false;
    B.tasks.release(), B.part.release(), B.ticket.release(), B.olist.release(), B.fptr.release(), B.fsrc.release(), B.tblocks.release(), B.tcols.release(), B.tlidx.release(), B.trowd.release(), B.ctrl_rows[0].release(), B.ctrl_rows[1].release();
    B.ctrl_ok[0] = B.ctrl_ok[1] = false;
    B.launches.clear();
    B.tables = B.tb_built = false;
    B.k = B.KB = 0;
    return FC_OK;
  }
  FCCHK(build_batch_tables(h));
  h->bat.pre_slot = -1;  // the state below is zeroed
  const int KB = k <= 4 ? 4 : (k <= 8 ? 8 : (k <= 16 ? 16 : 32));
  const size_t N = (size_t)h->N;
  FCCHK(build_tail_blocks(h, KB));  // (the width of a row block's column set depends on KB: its rows [column][KB] sit in LDS)
  if (KB != B.KB || !B.ring.p) {
    B.slot_doubles = (2 * N + (size_t)B.scratch_rows + 1) * KB;  // + the zero row of the operand lists
    FCCHK(B.ring.alloc(4 * B.slot_doubles));
    B.buf.n = B.slot_doubles;
    FCCHK(B.bstore.alloc(4 * N * KB));  // one right-hand side per ring phase: b(n) for step n's late tail, b(n + 1) being gathered ahead
    B.b.n = N * KB;
    FCCHK(B.ev.alloc((size_t)12 * h->nc * KB));
    FCCHK(B.partial.alloc((size_t)3 * ((size_t)B.n_tblocks + (size_t)nblocks(h->nc, FC_EB_TRIPS * (256 / KB)) + 1) * KB));
    FCCHK(B.flag.alloc(32));
  }
  B.k = k;
  B.KB = KB;
  for (int o = 0; o < 2; ++o)
    if (h->sys[o].ready && !B.ftile_ok[o]) FCCHK(batch_repack(h, o));
  for (DevBuf<double>* d : {&B.ring, &B.bstore, &B.ev, &B.partial}) FCCHK(d->zero(h->stream));
  B.cur = 0;
  bat_point(h);
  B.b.p = B.bstore.p;
  B.late[0] = B.late[1] = fc_ctx::Batch::Late{};
  B.last_dE.assign(32, 0.0), B.last_r.assign(32, 0.0), B.last_b.assign(32, 0.0);
  FCCHK(B.flag.zero(h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  return FC_OK;
}

static int batch_check(fc_ctx* h, int32_t k, const char* who) {
  if (!h) return fail(FC_ERR_INVALID, "null handle");
  if (h->bat.k == 0) return fail(FC_ERR_NOT_READY, std::string(who) + ": fc_set_batch not called");
  if (k != h->bat.k) return fail(FC_ERR_INVALID, std::string(who) + ": k differs from fc_set_batch");
  return FC_OK;
}

// host [k][n] <-> device [n][KB]
// (perm != nullptr: device rows are in the solver's permuted numbering, host vectors in the W layout)
static int batch_copy(fc_ctx* h, int n, double* host, double* dev, bool to_device, const int* perm = nullptr) {
  fc_ctx::Batch& B = h->bat;
  DevBuf<double> stage;
  FCCHK(stage.alloc((size_t)n * B.k));
  const int g = nblocks((int64_t)n * B.KB, 256);
  if (to_device) HIPCHK(hipMemcpyAsync(stage.p, host, (size_t)n * B.k * sizeof(double), hipMemcpyHostToDevice, h->stream));
#define FC_IL(K) hipLaunchKernelGGL((fc_b_interleave<K>), dim3(g), dim3(256), 0, h->stream, n, B.k, to_device ? stage.p : dev, to_device ? dev : stage.p, to_device ? 1 : 0, perm)
  FC_KB_DISPATCH(B.KB, FC_IL(4), FC_IL(8), FC_IL(16), FC_IL(32));
#undef FC_IL
  if (!to_device) HIPCHK(hipMemcpyAsync(host, stage.p, (size_t)n * B.k * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  return FC_OK;
}

// W-layout host vectors [k][N] = [u (2 nn) | p (nv)] of every simulation from the caller's split arrays
static void batch_pack_w(const fc_ctx* h, int k, const double* u, const double* p, std::vector<double>& w) {
  const size_t N = (size_t)h->N, nv2 = 2 * (size_t)h->nn;
  for (int s = 0; s < k; ++s) {
    if (u) std::copy(u + (size_t)s * nv2, u + (size_t)(s + 1) * nv2, w.begin() + (std::ptrdiff_t)(s * N));
    if (p) std::copy(p + (size_t)s * h->nv, p + (size_t)(s + 1) * h->nv, w.begin() + (std::ptrdiff_t)(s * N + nv2));
  }
}

int fc_set_state_batch(fc_handle h, int32_t k, const double* u_n, const double* u_nn, const double* p_n) {
  FCCHK(batch_check(h, k, "fc_set_state_batch"));
  if (!u_n || !u_nn) return fail(FC_ERR_INVALID, "fc_set_state_batch: null argument");
  if (!h->have_perm) return fail(FC_ERR_NOT_READY, "fc_set_state_batch: no permutation (fc_setup_solver)");
  HIPCHK(hipSetDevice(h->device));
  FCCHK(quiesce(h));
  fc_ctx::Batch& B = h->bat;
  B.pre_slot = -1;  // element vectors of the old state
  const size_t N = (size_t)h->N;
  std::vector<double> wn((size_t)k * N, 0.0), wnn((size_t)k * N, 0.0);
  if (!p_n) FCCHK(batch_copy(h, h->N, wn.data(), bat_n(h), false, h->perm.p));  // keeps the present pressure
  batch_pack_w(h, k, u_n, p_n, wn);
  for (int s = 0; s < k; ++s) std::copy(wn.begin() + (std::ptrdiff_t)(s * N), wn.begin() + (std::ptrdiff_t)((s + 1) * N), wnn.begin() + (std::ptrdiff_t)(s * N));
  batch_pack_w(h, k, u_nn, nullptr, wnn);
  FCCHK(batch_copy(h, h->N, wn.data(), bat_n(h), true, h->perm.p));
  FCCHK(batch_copy(h, h->N, wnn.data(), bat_nn(h), true, h->perm.p));
  FCCHK(B.flag.zero(h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  return FC_OK;
}

int fc_get_state_batch(fc_handle h, int32_t k, double* u_n, double* u_nn, double* p_n) {
  FCCHK(batch_check(h, k, "fc_get_state_batch"));
  HIPCHK(hipSetDevice(h->device));
  const size_t N = (size_t)h->N, nv2 = 2 * (size_t)h->nn;
  std::vector<double> w((size_t)k * N);
  if (u_n || p_n) {
    FCCHK(batch_copy(h, h->N, w.data(), bat_n(h), false, h->perm.p));
    for (int s = 0; s < k; ++s) {
      if (u_n) std::copy(w.begin() + (std::ptrdiff_t)(s * N), w.begin() + (std::ptrdiff_t)(s * N + nv2), u_n + (size_t)s * nv2);
      if (p_n) std::copy(w.begin() + (std::ptrdiff_t)(s * N + nv2), w.begin() + (std::ptrdiff_t)((s + 1) * N), p_n + (size_t)s * h->nv);
    }
  }
  if (u_nn) {
    FCCHK(batch_copy(h, h->N, w.data(), bat_nn(h), false, h->perm.p));
    for (int s = 0; s < k; ++s) std::copy(w.begin() + (std::ptrdiff_t)(s * N), w.begin() + (std::ptrdiff_t)(s * N + nv2), u_nn + (size_t)s * nv2);
  }
  return FC_OK;
}

int fc_get_solution_batch(fc_handle h, int32_t k, double* up) {
  FCCHK(batch_check(h, k, "fc_get_solution_batch"));
  if (!up) return fail(FC_ERR_INVALID, "fc_get_solution_batch: null argument");
  HIPCHK(hipSetDevice(h->device));
  return batch_copy(h, h->N, up, bat_n(h), false, h->perm.p);  // the last step's solution IS the state (u_n, p_n)
}

// Take simulation s out of the batch's dynamics: its state (both time levels) becomes zero and its non-finite flag is cleared.
// What a host does with a run that diverged (FC_ERR_DIVERGED, info[s][3]): the columns are independent, so the other runs are
// unaffected either way, but a column full of Inf / NaN would raise the flag again on every later step.
int fc_reset_sim_batch(fc_handle h, int32_t s) {
  if (!h || h->bat.k == 0) return fail(FC_ERR_NOT_READY, "fc_reset_sim_batch: fc_set_batch not called");
  if (s < 0 || s >= h->bat.k) return fail(FC_ERR_INVALID, "fc_reset_sim_batch: no such simulation");
  if (h->bat.pending) return fail(FC_ERR_INVALID, "fc_reset_sim_batch: a step is in flight");
  HIPCHK(hipSetDevice(h->device));
  fc_ctx::Batch& B = h->bat;
  B.pre_slot = -1;
  // the late tail of the step that just ended (side stream) may still be reading the state columns zeroed below: let it finish
  // (its late record stays collectable -- fc_step_batch_collect -- with the values of the step as it ran)
  if (B.side_busy) {
    HIPCHK(hipStreamSynchronize(h->stream2));
    B.side_busy = false;
  }
  const int g = nblocks(h->N, 256);
  for (double* d : {bat_n(h), bat_nn(h)}) {
#define FC_ZC(K) hipLaunchKernelGGL((fc_b_zero_column<K>), dim3(g), dim3(256), 0, h->stream, h->N, s, d)
    FC_KB_DISPATCH(B.KB, FC_ZC(4), FC_ZC(8), FC_ZC(16), FC_ZC(32));
#undef FC_ZC
  }
  const int zero = 0;
  HIPCHK(hipMemcpyAsync(B.flag.p + s, &zero, sizeof(int), hipMemcpyHostToDevice, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  return FC_OK;
}

// the launches of one batched step; controls are read from the host-mapped record (uctrl at s * kRecStride, body-force
// amplitudes at s * kRecStride + 32, the step's sequence number at kSeqSlot), every simulation's outputs go to its own record
constexpr int kSeqSlot = 8000;
// lead_elem: the step starts with its element loop; spec_slot >= 0: it ENDS with the element loop of the next step (scheme of
// that slot) -- the loop depends on the state only, so it runs while the host is between two fc_step_batch calls, as
// speculate_next_rhs does for the single simulation
constexpr int kLateRecB = 144;  // late record of a simulation inside its record (kRecStride): + 8 x step parity
static int batch_tail_geometry(fc_ctx* h, int compute_energy, int* n_row_blocks, int* n_cell_blocks) {
  fc_ctx::Batch& B = h->bat;
  // (a step off the residual monitor's cadence has no row blocks: B.pend_checked, set by step_batch_begin before the launches)
  const int cpb = FC_EB_TRIPS * (256 / B.KB);  // cells per cell workgroup of fc_tail_b (fc_energy_b_block)
  *n_row_blocks = B.pend_checked ? B.n_tblocks : 0;
  *n_cell_blocks = compute_energy ? nblocks(h->nc, cpb) : 0;
  if ((size_t)3 * (*n_row_blocks + *n_cell_blocks) * B.KB > B.partial.n) return fail(FC_ERR_INVALID, "fc_step_batch: partial buffer too small");
  return FC_OK;
}

// overlapped == false: the whole step on the main stream (tail and final behind the apply, one record per simulation).
// overlapped == true: the main stream ends with fc_early_b (what the host waits for) and the next step's element loop; residual monitor and
// energy are batch_launches_side's, on the side stream.
// lead: 2 = element loop + full gather, 1 = full gather (element vectors left by the previous step), 0 = control rows only (the
// previous step gathered the control-independent right-hand side as well); spec_gather: gather the NEXT step's control-independent
// right-hand side behind its element loop (into the b / y buffers that step will use)
static int batch_launches(fc_ctx* h, int order_slot, int compute_energy, int lead = 2, int spec_slot = -1, bool overlapped = false, bool spec_gather = false) {
  fc_ctx::Batch& B = h->bat;
  OrderSys& S = h->sys[order_slot];
  const int KB = B.KB, N = h->N, nc = h->nc;
  const int par = B.cur & 1;
  const double* uc = h->pin_dev;
  const double* uf = h->pin_dev + 32;
  const double* seqp = h->pin_dev + kSeqSlot + par;
  const int g_elem = nblocks(nc, 256 / (8 * (KB / 2))), g_rows = nblocks((int64_t)N * (KB / 2), 256);  // (element loop: thread = (cell, lane8, simulation pair))
  // the state ring: this step reads (u_n, u_nn) from slots cur, cur - 1 and writes its solution -- the new state -- into the x
  // half of slot cur + 1 (= B.buf); the caller moves `cur` on afterwards
  const double* un = bat_n(h);
  const double* unn = bat_nn(h);
  double* xnew = B.buf.p + (size_t)N * KB;
  // element loop: one thread per (cell, simulation), everything in registers (fc_rhs_elem_breg); FC_BATCH_ELEM=lds: the LDS-shared form
  // (thread = (cell, lane8, simulation pair), fc_rhs_elem_b)
  static const bool elem_reg = [] { const char* e = std::getenv("FC_BATCH_ELEM"); return !(e && std::string(e) == "lds"); }();
  auto element_loop = [&](const StepCoeffs& c, const double* u1, const double* u2) {
    if (elem_reg) {
      const int g_reg = nblocks(nc, 256 / KB);
#define FC_ELEMR(K)                                                                                                                                \
  do {                                                                                                                                             \
    if (h->have_force)                                                                                                                             \
      hipLaunchKernelGGL((fc_rhs_elem_breg<K, true>), dim3(g_reg), dim3(256), 0, h->stream, nc, h->nn, h->cn.p, h->cnp.p, h->geom.p, u1, u2,       \
                         h->fprof.p, h->n_act, uf, kRecStride, c.cm_n, c.cm_nn, c.cc_n, c.cc_nn, B.ev.p);                                          \
    else                                                                                                                                           \
      hipLaunchKernelGGL((fc_rhs_elem_breg<K, false>), dim3(g_reg), dim3(256), 0, h->stream, nc, h->nn, h->cn.p, h->cnp.p, h->geom.p, u1, u2,      \
                         (const double*)nullptr, 0, uf, kRecStride, c.cm_n, c.cm_nn, c.cc_n, c.cc_nn, B.ev.p);                                     \
  } while (0)
      FC_KB_DISPATCH(KB, FC_ELEMR(4), FC_ELEMR(8), FC_ELEMR(16), FC_ELEMR(32));
#undef FC_ELEMR
      return;
    }
#define FC_ELEM(K) hipLaunchKernelGGL((fc_rhs_elem_b<K>), dim3(g_elem), dim3(256), 0, h->stream, nc, h->nn, h->cn.p, h->cnp.p, h->geom.p, u1, u2, \
                                      h->have_force ? h->fprof.p : nullptr, h->have_force ? h->n_act : 0, uf, kRecStride, c.cm_n, c.cm_nn, c.cc_n, c.cc_nn, B.ev.p)
    FC_KB_DISPATCH(KB, FC_ELEM(4), FC_ELEM(8), FC_ELEM(16), FC_ELEM(32));
#undef FC_ELEM
  };
  // the side stream's gate opens behind the gather that runs ahead for the next step, if there is one (FC_LATE_GATE=0: behind fc_early_b)
  static const bool late_gate_on = [] { const char* e = std::getenv("FC_LATE_GATE"); return !(e && e[0] == '0'); }();
  const bool late_gate = late_gate_on && overlapped && spec_slot >= 0 && spec_gather;
  if (lead == 2) element_loop(coeffs_for(h, order_slot), un, unn);
  if (lead >= 1) {
#define FC_GATH(K) hipLaunchKernelGGL((fc_rhs_gather_b<K>), dim3(g_rows), dim3(256), 0, h->stream, N, h->gptr_p.p, h->gidx_p.p, B.ev.p, h->bcslot_p.p, \
                                      h->bcprof.p, S.lift_p.p, h->n_act, uc, kRecStride, B.b.p, B.buf.p, S.have_c ? S.c_rowptr.p : nullptr, S.c_col.p, \
                                      S.c_val.p, un, 1)
    FC_KB_DISPATCH(KB, FC_GATH(4), FC_GATH(8), FC_GATH(16), FC_GATH(32));
#undef FC_GATH
  } else if (B.n_ctrl_rows[order_slot] > 0) {
    const int nr = B.n_ctrl_rows[order_slot];
#define FC_CTRL(K) hipLaunchKernelGGL((fc_rhs_ctrl_b<K>), dim3(nblocks((int64_t)nr * (K / 2), 256)), dim3(256), 0, h->stream, nr, B.ctrl_rows[order_slot].p, N, \
                                      h->bcslot_p.p, h->bcprof.p, S.lift_p.p, h->n_act, uc, kRecStride, B.b.p, B.buf.p)
    FC_KB_DISPATCH(KB, FC_CTRL(4), FC_CTRL(8), FC_CTRL(16), FC_CTRL(32));
#undef FC_CTRL
  }
  // (the down-sweep tiles test what they write for finiteness whenever no tail pass of this stream does it: the overlapped step,
  //  and a step off the residual monitor's cadence)
  FCCHK(batch_apply(h, order_slot, overlapped || !B.pend_checked));
  if ((int64_t)B.tlidx.n != S.Ap_nnz && S.Ap_nnz > 0) return fail(FC_ERR_INVALID, "fc_step_batch: tail tables and system pattern disagree");
  if (overlapped) {
#define FC_EARLYB(K) hipLaunchKernelGGL((fc_early_b<K>), dim3(B.k), dim3(256), 0, h->stream, h->n_sens, h->s_rowptr.p, h->s_idxp.p, h->s_w.p, xnew, B.flag.p, \
                                        h->pin_dev, kRecStride, seqp, late_gate ? (unsigned long long*)nullptr : (unsigned long long*)h->solved.p)
    FC_KB_DISPATCH(KB, FC_EARLYB(4), FC_EARLYB(8), FC_EARLYB(16), FC_EARLYB(32));
#undef FC_EARLYB
  } else {
    // tail: residual monitor, non-finite flags, energy (the solution stays where it is: it is the new state)
    int n_row_blocks = 0, n_cell_blocks = 0;
    FCCHK(batch_tail_geometry(h, compute_energy, &n_row_blocks, &n_cell_blocks));
    const int G = n_row_blocks + n_cell_blocks;
#define FC_TAILB(K) hipLaunchKernelGGL((fc_tail_b<K>), dim3(G), dim3(256), fc_tail_b_lds(B.tb_cols, K), h->stream, N, h->velrow_p.p, xnew, B.b.p, B.tblocks.p, \
                                       B.tcols.p, (const int4*)B.trowd.p, B.tlidx.p, S.Ap_val.p, B.flag.p, B.partial.p, G, n_cell_blocks, nc, h->cnp.p, h->geom.p, B.tb_cols)
    if (G > 0) FC_KB_DISPATCH(KB, FC_TAILB(4), FC_TAILB(8), FC_TAILB(16), FC_TAILB(32));
#undef FC_TAILB
#define FC_FINB(K) hipLaunchKernelGGL((fc_final_b<K>), dim3(B.k), dim3(1024), 0, h->stream, G, n_row_blocks, B.partial.p, h->n_sens, h->s_rowptr.p, h->s_idxp.p, \
                                      h->s_w.p, xnew, B.flag.p, h->pin_dev, kRecStride, seqp, compute_energy)
    FC_KB_DISPATCH(KB, FC_FINB(4), FC_FINB(8), FC_FINB(16), FC_FINB(32));
#undef FC_FINB
  }
  if (spec_slot >= 0) element_loop(coeffs_for(h, spec_slot), xnew, un);  // the NEXT step's loop: its (u_n, u_nn) = (this solution, this u_n)
  if (spec_slot >= 0 && spec_gather) {
    // ... and the control-independent part of its right-hand side, into the buffers that step will use: b of the next ring phase (none of
    // the two late tails that may still run reads it) and the y half of its work buffer
    OrderSys& Sn = h->sys[spec_slot];
    double* bnext = B.bstore.p + (overlapped ? (size_t)((B.cur + 1) % 4) * (size_t)N * KB : 0);
    double* ynext = bat_slot(h, B.cur + 2);
#define FC_GATH(K) hipLaunchKernelGGL((fc_rhs_gather_b<K>), dim3(g_rows), dim3(256), 0, h->stream, N, h->gptr_p.p, h->gidx_p.p, B.ev.p, h->bcslot_p.p, \
                                      h->bcprof.p, Sn.lift_p.p, h->n_act, uc, kRecStride, bnext, ynext, (const int*)nullptr, Sn.c_col.p, Sn.c_val.p, xnew, 0, \
                                      late_gate ? (unsigned long long*)h->solved.p : (unsigned long long*)nullptr, seqp)
    FC_KB_DISPATCH(KB, FC_GATH(4), FC_GATH(8), FC_GATH(16), FC_GATH(32));
#undef FC_GATH
  }
  HIPCHK(hipGetLastError());
  return FC_OK;
}

// rows of `slot` whose right-hand side depends on u_ctrl: Dirichlet rows and rows with an entry in a lifting vector (permuted numbering)
static int build_ctrl_rows(fc_ctx* h, int slot) {
  fc_ctx::Batch& B = h->bat;
  OrderSys& S = h->sys[slot];
  const int N = h->N;
  std::vector<double> lift((size_t)std::max(1, h->n_act) * N, 0.0);
  if (h->n_act > 0 && S.lift_p.n >= (size_t)h->n_act * N) {
    HIPCHK(hipMemcpyAsync(lift.data(), S.lift_p.p, (size_t)h->n_act * N * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
  }
  std::vector<int> ip((size_t)N), rows;
  for (int i = 0; i < N; ++i) ip[(size_t)h->h_perm[(size_t)i]] = i;
  std::vector<unsigned char> isbc((size_t)N, 0);
  for (int k = 0; k < h->n_bc; ++k) isbc[(size_t)ip[(size_t)h->h_bc_dofs[(size_t)k]]] = 1;
  for (int i = 0; i < N; ++i) {
    bool dep = isbc[(size_t)i] != 0;
    for (int k = 0; k < h->n_act && !dep; ++k) dep = lift[(size_t)k * N + i] != 0.0;
    if (dep) rows.push_back(i);
  }
  B.n_ctrl_rows[slot] = (int)rows.size();
  if (rows.empty()) rows.push_back(0);
  FCCHK(B.ctrl_rows[slot].upload(rows, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  B.ctrl_ok[slot] = true;
  return FC_OK;
}

// the side stream's share of an overlapped batched step: [gate] residual monitor + energy of all simulations -> their late records
static int batch_launches_side(fc_ctx* h, int order_slot, int compute_energy) {
  fc_ctx::Batch& B = h->bat;
  OrderSys& S = h->sys[order_slot];
  const int KB = B.KB, N = h->N, nc = h->nc, par = B.cur & 1;
  const double* seqp = h->pin_dev + kSeqSlot + par;
  const double* xnew = B.buf.p + (size_t)N * KB;
  int n_row_blocks = 0, n_cell_blocks = 0;
  FCCHK(batch_tail_geometry(h, compute_energy, &n_row_blocks, &n_cell_blocks));
  const int G = n_row_blocks + n_cell_blocks;
  hipLaunchKernelGGL(fc_wait_solved_b, dim3(1), dim3(1), 0, h->stream2, (const unsigned long long*)h->solved.p, seqp, h->side_err.p, h->gate_spin);
#define FC_TAILB(K) hipLaunchKernelGGL((fc_tail_b<K>), dim3(G), dim3(256), fc_tail_b_lds(B.tb_cols, K), h->stream2, N, h->velrow_p.p, xnew, B.b.p, B.tblocks.p, \
                                       B.tcols.p, (const int4*)B.trowd.p, B.tlidx.p, S.Ap_val.p, h->flag2x.p, B.partial.p, G, n_cell_blocks, nc, h->cnp.p, h->geom.p, B.tb_cols)
  if (G > 0) FC_KB_DISPATCH(KB, FC_TAILB(4), FC_TAILB(8), FC_TAILB(16), FC_TAILB(32));
#undef FC_TAILB
#define FC_FINLB(K) hipLaunchKernelGGL((fc_final_late_b<K>), dim3(B.k), dim3(1024), 0, h->stream2, G, n_row_blocks, B.partial.p, h->pin_dev, kRecStride, \
                                       kLateRecB + 8 * par, seqp, compute_energy, (const int*)h->side_err.p)
  FC_KB_DISPATCH(KB, FC_FINLB(4), FC_FINLB(8), FC_FINLB(16), FC_FINLB(32));
#undef FC_FINLB
  HIPCHK(hipGetLastError());
  return FC_OK;
}

// everything a kernel argument of batch_launches is taken from, hashed (FNV-1a): a captured graph is replayed only while
// this is unchanged
static uint64_t batch_signature(fc_ctx* h, int order_slot, int compute_energy, int lead_elem, int spec_slot, bool spec_gather) {
  fc_ctx::Batch& B = h->bat;
  OrderSys& S = h->sys[order_slot];
  const StepCoeffs c = coeffs_for(h, order_slot);
  auto bits = [](double v) {
    uint64_t u;
    std::memcpy(&u, &v, sizeof u);
    return u;
  };
  const uint64_t words[] = {
      (uint64_t)(uintptr_t)h->cn.p, (uint64_t)(uintptr_t)h->geom.p, (uint64_t)(uintptr_t)bat_n(h), (uint64_t)(uintptr_t)bat_nn(h), (uint64_t)(uintptr_t)h->cnp.p, (uint64_t)(uintptr_t)h->fprof.p,
      (uint64_t)(uintptr_t)B.ev.p, (uint64_t)(uintptr_t)h->gptr_p.p, (uint64_t)(uintptr_t)h->gidx_p.p, (uint64_t)(uintptr_t)h->bcslot_p.p,
      (uint64_t)(uintptr_t)h->bcprof.p, (uint64_t)(uintptr_t)S.lift_p.p, (uint64_t)(uintptr_t)B.b.p, (uint64_t)(uintptr_t)B.buf.p, (uint64_t)(uintptr_t)S.c_rowptr.p,
      (uint64_t)(uintptr_t)S.c_col.p, (uint64_t)(uintptr_t)S.c_val.p, (uint64_t)(uintptr_t)B.tasks.p, (uint64_t)(uintptr_t)B.olist.p, (uint64_t)(uintptr_t)B.ftile[order_slot].p,
      (uint64_t)(uintptr_t)B.fptr.p, (uint64_t)(uintptr_t)B.fsrc.p, (uint64_t)(uintptr_t)h->perm.p, (uint64_t)(uintptr_t)h->iperm.p, (uint64_t)(uintptr_t)S.Ap_rowptr.p,
      (uint64_t)(uintptr_t)S.Ap_col.p, (uint64_t)(uintptr_t)S.Ap_val.p, (uint64_t)(uintptr_t)h->velrow_p.p, (uint64_t)(uintptr_t)h->s_idxp.p, (uint64_t)(uintptr_t)B.flag.p,
      (uint64_t)(uintptr_t)B.partial.p, (uint64_t)(uintptr_t)h->s_rowptr.p, (uint64_t)(uintptr_t)h->s_idx.p, (uint64_t)(uintptr_t)h->s_w.p,
      (uint64_t)(uintptr_t)h->pin_dev, (uint64_t)(uintptr_t)B.tblocks.p, (uint64_t)(uintptr_t)B.tcols.p, (uint64_t)(uintptr_t)B.tlidx.p, (uint64_t)B.n_tblocks, (uint64_t)B.k, (uint64_t)B.KB, (uint64_t)h->n_act, (uint64_t)h->n_sens, (uint64_t)(h->have_force ? 1 : 0),
      (uint64_t)(S.have_c ? 1 : 0), (uint64_t)compute_energy, (uint64_t)B.launches.size(), (uint64_t)B.tasks.n, bits(c.cm_n), bits(c.cm_nn), bits(c.cc_n),
      bits(c.cc_nn), (uint64_t)lead_elem, (uint64_t)(spec_gather ? 1 : 0), (uint64_t)(B.pend_checked ? 1 : 0), (uint64_t)(uintptr_t)B.ctrl_rows[order_slot].p, (uint64_t)B.n_ctrl_rows[order_slot],
      (uint64_t)(uintptr_t)B.bstore.p, (uint64_t)(uintptr_t)B.part.p, (uint64_t)(uintptr_t)B.ticket.p, (uint64_t)(spec_slot + 1), spec_slot >= 0 ? bits(coeffs_for(h, spec_slot).cm_n) : 0,
      spec_slot >= 0 ? bits(coeffs_for(h, spec_slot).cm_nn) : 0, spec_slot >= 0 ? bits(coeffs_for(h, spec_slot).cc_n) : 0,
      spec_slot >= 0 ? bits(coeffs_for(h, spec_slot).cc_nn) : 0};
  uint64_t hsh = 1469598103934665603ull;
  for (uint64_t w : words)
    for (int b = 0; b < 8; ++b) {
      hsh ^= (w >> (8 * b)) & 0xffu;
      hsh *= 1099511628211ull;
    }
  return hsh ? hsh : 1;
}

static void batch_drop_graphs(fc_ctx* h) {
  for (int ph = 0; ph < 4; ++ph)
    for (int o = 0; o < 2; ++o)
      for (int e = 0; e < 4; ++e)
        for (int l = 0; l < 3; ++l) {
          if (h->bat.gexec[ph][o][e][l]) (void)hipGraphExecDestroy(h->bat.gexec[ph][o][e][l]);
          h->bat.gexec[ph][o][e][l] = nullptr;
          h->bat.gsig[ph][o][e][l] = 0;
        }
  for (int ph = 0; ph < 4; ++ph)
    for (int e = 0; e < 4; ++e) {
      if (h->bat.gside[ph][e]) (void)hipGraphExecDestroy(h->bat.gside[ph][e]);
      h->bat.gside[ph][e] = nullptr;
      h->bat.gside_sig[ph][e] = 0;
    }
  h->bat.pre_slot = -1;
}

static int batch_enqueue(fc_ctx* h, int order_slot, int compute_energy, bool overlapped) {
  // FC_BATCH_GRAPH=1: replay the step as a HIP graph per (ring phase, slot, flags).  Default since round 5: plain launches -- the first
  // kernel of a step starts ~10 us sooner than behind hipGraphLaunch and the other ~16 launches are enqueued while the GPU works
  // (k = 8 / 16 / 32 on O1: 49.2 / 90.3 / 139.0 k simulated steps/s against 50.3 / 87.9 / 135.6 k with graphs; the host spends
  // ~60 us per step launching, a third of the step)
  static const bool use_graph = [] {
    const char* e = std::getenv("FC_BATCH_GRAPH");
    return e && e[0] == '1';
  }();
  fc_ctx::Batch& B = h->bat;
  // element loop of the NEXT step behind this one (same prediction as speculate_next_rhs: BDF2 follows, or the same CN slot);
  // body forces make the loop depend on u_ctrl: no speculation then
  static const bool speculate = [] {
    const char* e = std::getenv("FC_SPECULATE");
    return !(e && e[0] == '0');
  }();
  int spec_slot = -1;
  if (speculate && !h->have_force) {
    const int next = h->sys[order_slot].have_c ? order_slot : FC_SLOT_BDF2;
    if (h->sys[next].ready && h->sys[next].have_lift) spec_slot = next;
  }
  // ... and the control-independent part of its right-hand side behind it (FC_SPECULATE_GATHER=0: not): what is left for the next step
  // before its factor sweeps is the handful of rows that see u_ctrl.  Not for Crank-Nicolson slots (their explicit operator would be
  // subtracted in another order than the full gather does it: the batched and the single run are kept bit-identical)
  static const bool speculate_gather = [] {
    const char* e = std::getenv("FC_SPECULATE_GATHER");
    return !(e && e[0] == '0');
  }();
  bool spec_gather = false;
  if (spec_slot >= 0 && speculate_gather && !h->sys[spec_slot].have_c) {
    if (!B.ctrl_ok[spec_slot]) FCCHK(build_ctrl_rows(h, spec_slot));
    spec_gather = true;
  }
  const bool have_ev = B.pre_slot == order_slot && !h->have_force;
  B.b.p = B.bstore.p + (overlapped ? (size_t)(B.cur % 4) * (size_t)h->N * B.KB : 0);  // (overlapped: b(n) stays intact for step n's late tail)
  // (the gathered right-hand side counts only if it sits where this step reads it: a step that switches between the overlapped and the
  //  one-stream form -- fc_step_batch after fc_step_batch_begin / _end_early -- gathers again)
  const int lead = !have_ev ? 2 : ((B.pre_gather && B.ctrl_ok[order_slot] && B.pre_b == B.b.p && B.pre_y == B.buf.p) ? 0 : 1);
  B.pre_slot = -1;
  B.pre_gather = false;
  const double* spec_b = B.bstore.p + (overlapped ? (size_t)((B.cur + 1) % 4) * (size_t)h->N * B.KB : 0);
  const double* spec_y = bat_slot(h, B.cur + 2);
  auto advance = [&]() {  // the solution this step writes is the state from here on
    B.cur = (B.cur + 1) % 4;
    bat_point(h);
    B.pre_slot = spec_slot;
    B.pre_gather = spec_gather;
    B.pre_b = spec_b, B.pre_y = spec_y;
  };
  if (!use_graph || h->timing) {
    FCCHK(batch_launches(h, order_slot, compute_energy, lead, spec_slot, overlapped, spec_gather));
    if (overlapped) FCCHK(batch_launches_side(h, order_slot, compute_energy));
    advance();
    return FC_OK;
  }
  // one graph per ring phase: the buffers rotate with period four (the overlapped step is two graphs, one per stream)
  const int ph = B.cur, ei = (compute_energy ? 1 : 0) + (B.pend_checked ? 2 : 0), li = lead;
  const uint64_t sig = batch_signature(h, order_slot, compute_energy, lead, spec_slot, spec_gather) ^ (overlapped ? 0x9e3779b97f4a7c15ull : 0ull);
  hipGraphExec_t& gx = B.gexec[ph][order_slot][ei][li];
  if (!gx || B.gsig[ph][order_slot][ei][li] != sig) {
    FCCHK(capture_graph(h, h->stream, &gx, [&] { return batch_launches(h, order_slot, compute_energy, lead, spec_slot, overlapped, spec_gather); }));
    B.gsig[ph][order_slot][ei][li] = sig;
  }
  HIPCHK(hipGraphLaunch(gx, h->stream));
  if (overlapped) {
    const uint64_t ssig = sig ^ ((uint64_t)order_slot + 1) * 0x100000001b3ull;
    hipGraphExec_t& gs = B.gside[ph][ei];
    if (!gs || B.gside_sig[ph][ei] != ssig) {
      FCCHK(capture_graph(h, h->stream2, &gs, [&] { return batch_launches_side(h, order_slot, compute_energy); }));
      B.gside_sig[ph][ei] = ssig;
    }
    HIPCHK(hipGraphLaunch(gs, h->stream2));
  }
  advance();
  return FC_OK;
}

static int batch_ready(fc_ctx* h, int order_slot, int32_t k, const char* who) {
  FCCHK(batch_check(h, k, who));
  FCCHK(check_step_ready(h, order_slot));
  OrderSys& S = h->sys[order_slot];
  if (!S.ready) return fail(FC_ERR_NOT_READY, std::string(who) + ": fc_setup_solver not called for this order");
  if (h->method != FC_METHOD_REFINE || h->max_iter != 0) return fail(FC_ERR_INVALID, std::string(who) + ": batched steps apply the factors directly (FC_METHOD_REFINE, no refinement sweeps)");
  if (h->partitioned || S.truncated) return fail(FC_ERR_INVALID, std::string(who) + ": single-GPU handles with full factors only");
  if (S.inexact) return fail(FC_ERR_INVALID, std::string(who) + ": this slot's factors are inexact (a preconditioner for GMRES): batched stepping applies them directly");
  if (S.factor_free) return fail(FC_ERR_INVALID, std::string(who) + ": this slot has no factors (fc_setup_krylov): batched stepping applies factors directly");
  if (h->n_act > 32 || h->n_sens > 64) return fail(FC_ERR_INVALID, std::string(who) + ": at most 32 actuators and 64 sensors");
  if (kRecStride * 32 > kSeqSlot) return fail(FC_ERR_INVALID, "record too small");
  return FC_OK;
}

// late records of an overlapped batched step, parity `par`: all k simulations
int collect_late_batch(fc_ctx* h, int par) {
  fc_ctx::Batch& B = h->bat;
  fc_ctx::Batch::Late& L = B.late[par];
  if (!L.pending) return FC_OK;
  const double seq = L.seq;
  auto bits = [](double v) {
    unsigned long long u;
    std::memcpy(&u, &v, sizeof u);
    return u;
  };
  auto ok1 = [&](int s) {
    volatile double* rec = h->pin + (size_t)s * kRecStride + kLateRecB + 8 * par;
    if (rec[3] != seq) return false;
    unsigned long long x = bits(seq), w = x, kk = 3;
    for (int i = 0; i < 4; ++i, kk += 2) {
      const unsigned long long v = bits(rec[i < 3 ? i : 6]);
      x ^= v;
      w += kk * v;
    }
    return x == bits(rec[4]) && w == bits(rec[5]);
  };
  auto ok = [&]() {
    for (int s = 0; s < B.k; ++s)
      if (!ok1(s)) return false;
    return true;
  };
  bool seen = false;
  for (long spin = 0; spin < 20000000L; ++spin) {
    if (ok()) {
      seen = true;
      break;
    }
    __builtin_ia32_pause();
  }
  if (!seen) {
    HIPCHK(hipStreamSynchronize(h->stream2));
    if (!ok()) return fail(FC_ERR_HIP, "fc_step_batch: a late record (residual monitor, energy) never arrived or failed its checksum");
  }
  L.pending = false;
  if (B.k > 0 && h->pin[kLateRecB + 8 * par + 6] != 0.0)
    return fail(FC_ERR_HIP, "fc_step_batch: the side stream stopped waiting for the step's solve (main stream delayed or a launch failed): "
                            "residual monitor and energy of that step are not valid");
  if (par == B.last_par)
    for (int s = 0; s < B.k; ++s) {
      volatile double* rec = h->pin + (size_t)s * kRecStride + kLateRecB + 8 * par;
      B.last_dE[(size_t)s] = L.energy ? rec[0] : std::numeric_limits<double>::quiet_NaN();
      B.last_r[(size_t)s] = L.checked ? (double)rec[1] : std::numeric_limits<double>::quiet_NaN();  // (off the monitor's cadence: NaN)
      B.last_b[(size_t)s] = L.checked ? (double)rec[2] : std::numeric_limits<double>::quiet_NaN();
    }
  return FC_OK;
}

static int step_batch_begin(fc_handle h, int order_slot, int32_t k, const double* u_ctrl, const double* u_force, int compute_energy, bool want_all) {
  FCCHK(batch_ready(h, order_slot, k, "fc_step_batch"));
  if (h->n_act > 0 && !u_ctrl) return fail(FC_ERR_INVALID, "fc_step_batch: u_ctrl is null");
  if (h->bat.pending || h->step_pending) return fail(FC_ERR_INVALID, "fc_step_batch_begin: the previous step was not collected");
  HIPCHK(hipSetDevice(h->device));
  if (h->side_busy) FCCHK(quiesce(h));  // (a single simulation's late tail on this handle)
  h->pre_slot = -1;
  fc_ctx::Batch& B = h->bat;
  const OrderSys& S = h->sys[order_slot];
  const bool overlapped = h->overlap && !want_all && !S.nt && !h->timing;
  const int par = B.cur & 1;
  // the late tail of the step two back read the b buffer / sequence slot this step is about to write: it finished a step ago
  FCCHK(collect_late_batch(h, par));
  if (!overlapped && B.side_busy) {
    FCCHK(collect_late_batch(h, par ^ 1));
    B.side_busy = false;
  }
  volatile double* pin = h->pin;
  for (int s = 0; s < h->bat.KB; ++s)
    for (int a = 0; a < h->n_act; ++a) {
      pin[s * kRecStride + a] = s < k ? u_ctrl[(size_t)s * h->n_act + a] : 0.0;
      pin[s * kRecStride + 32 + a] = s < k ? (u_force ? u_force[(size_t)s * h->n_act + a] : u_ctrl[(size_t)s * h->n_act + a]) : 0.0;
    }
  // the residual monitor's cadence (fc_set_solver_options check_residual = n: every n-th batched step; the reference forms no residual at
  // all, flowsolver.py:728-737); the non-finite test runs on every step
  const int every = residual_every(h, S);
  B.pend_checked = every != 0 && (B.step_count % (uint64_t)every) == 0;
  ++B.step_count;
  B.pend_slot = order_slot;
  B.pend_energy = compute_energy;
  B.pend_seq = (double)(++h->seq);
  B.pend_par = par;
  B.pend_overlapped = overlapped;
  pin[kSeqSlot + par] = B.pend_seq;
  FCCHK(batch_enqueue(h, order_slot, compute_energy, overlapped));
  if (overlapped) {
    B.late[par].pending = true;
    B.late[par].seq = B.pend_seq;
    B.late[par].energy = compute_energy;
    B.late[par].checked = B.pend_checked;
    B.side_busy = true;
  }
  B.pending = true;
  return FC_OK;
}

int fc_step_batch_begin(fc_handle h, int order_slot, int32_t k, const double* u_ctrl, const double* u_force, int compute_energy) {
  return step_batch_begin(h, order_slot, k, u_ctrl, u_force, compute_energy, false);
}

// flags_out [k] (may be NULL): 1 where a simulation's velocity became non-finite in this step
static int step_batch_end(fc_handle h, int32_t k, double* y_out, double* dE_out, double* info_out, int32_t* flags_out) {
  FCCHK(batch_check(h, k, "fc_step_batch_end"));
  fc_ctx::Batch& B = h->bat;
  if (!B.pending) return fail(FC_ERR_INVALID, "fc_step_batch_end: no step in flight");
  B.pending = false;
  HIPCHK(hipSetDevice(h->device));
  volatile double* pin = h->pin;
  const double seq = B.pend_seq;
  auto bits = [](double v) {
    unsigned long long u;
    std::memcpy(&u, &v, sizeof u);
    return u;
  };
  auto record_ok = [&](int s) {  // the checksummed record of fc_publish, simulation s
    volatile double* r = pin + (size_t)s * kRecStride;
    if (r[137] != seq) return false;
    unsigned long long x = bits(seq), w = x, kk = 3;
    for (int q = 0; q < h->n_sens; ++q, kk += 2) {
      const unsigned long long v = bits(r[64 + q]);
      x ^= v;
      w += kk * v;
    }
    const unsigned long long tail[4] = {bits(r[128]), bits(r[129]), bits(r[130]), bits(r[136])};
    for (int i = 0; i < 4; ++i, kk += 2) {
      x ^= tail[i];
      w += kk * tail[i];
    }
    return x == bits(r[138]) && w == bits(r[139]);
  };
  auto all_ok = [&]() {
    for (int s = 0; s < k; ++s)
      if (!record_ok(s)) return false;
    return true;
  };
  bool seen = false;
  if (!h->timing) {
    for (long spin = 0; spin < 20000000L; ++spin) {
      if (all_ok()) {
        seen = true;
        break;
      }
      __builtin_ia32_pause();
    }
  }
  if (!seen) {
    HIPCHK(hipStreamSynchronize(h->stream));
    FCCHK(time_collect(h));
    if (!all_ok()) return fail(FC_ERR_HIP, "fc_step_batch: a step record failed its checksum after stream synchronisation");
  }
  B.last_par = B.pend_par;
  int any = 0;
  for (int s = 0; s < k; ++s) {
    volatile double* r = pin + (size_t)s * kRecStride;
    for (int q = 0; q < h->n_sens; ++q)
      if (y_out) y_out[(size_t)s * h->n_sens + q] = r[64 + q];
    const int flag = ((int)r[136]) % 1024;
    any |= flag;
    if (flags_out) flags_out[s] = flag;
    if (info_out) info_out[4 * s + 3] = flag;
    if (!B.pend_overlapped) {
      B.last_dE[(size_t)s] = B.pend_energy ? r[128] : std::numeric_limits<double>::quiet_NaN();
      B.last_r[(size_t)s] = B.pend_checked ? (double)r[129] : std::numeric_limits<double>::quiet_NaN();
      B.last_b[(size_t)s] = B.pend_checked ? (double)r[130] : std::numeric_limits<double>::quiet_NaN();
    }
  }
  if (B.pend_overlapped && (dE_out || info_out)) FCCHK(collect_late_batch(h, B.last_par));
  for (int s = 0; s < k; ++s) {
    if (dE_out) dE_out[s] = B.last_dE[(size_t)s];
    if (info_out) {
      const double r2 = B.last_r[(size_t)s], b2 = B.last_b[(size_t)s];
      info_out[4 * s + 0] = 0.0;
      info_out[4 * s + 1] = std::sqrt(r2 / (b2 > 0 ? b2 : 1.0));
      info_out[4 * s + 2] = std::sqrt(b2);
    }
  }
  if (any) return fail(FC_ERR_DIVERGED, "non-finite velocity after solve (info[s][3] / flags mark the simulations)");
  return FC_OK;
}

int fc_step_batch_end(fc_handle h, int32_t k, double* y_out, double* dE_out, double* info_out) {
  return step_batch_end(h, k, y_out, dE_out, info_out, nullptr);
}

// fc_step_end / fc_step_collect for k simulations: the early end hands over the measurements and the non-finite flags as soon as the
// solve is done; energy and solve info of that step follow (fc_step_batch_collect blocks until they exist; info[s][3] repeats the flag)
int fc_step_batch_end_early(fc_handle h, int32_t k, double* y_out, int32_t* flags_out) {
  return step_batch_end(h, k, y_out, nullptr, nullptr, flags_out);
}

int fc_step_batch_collect(fc_handle h, int32_t k, double* dE_out, double* info_out) {
  FCCHK(batch_check(h, k, "fc_step_batch_collect"));
  HIPCHK(hipSetDevice(h->device));
  fc_ctx::Batch& B = h->bat;
  FCCHK(collect_late_batch(h, B.last_par));
  for (int s = 0; s < k; ++s) {
    if (dE_out) dE_out[s] = B.last_dE[(size_t)s];
    if (info_out) {
      const double r2 = B.last_r[(size_t)s], b2 = B.last_b[(size_t)s];
      info_out[4 * s + 0] = 0.0;
      info_out[4 * s + 1] = std::sqrt(r2 / (b2 > 0 ? b2 : 1.0));
      info_out[4 * s + 2] = std::sqrt(b2);
      info_out[4 * s + 3] = std::numeric_limits<double>::quiet_NaN();  // (the flags came with the early end)
    }
  }
  return FC_OK;
}

int fc_step_batch(fc_handle h, int order_slot, int32_t k, const double* u_ctrl, const double* u_force, double* y_out, double* dE_out,
                  int compute_energy, double* info_out) {
  FCCHK(step_batch_begin(h, order_slot, k, u_ctrl, u_force, compute_energy, dE_out != nullptr || info_out != nullptr));
  return fc_step_batch_end(h, k, y_out, dE_out, info_out);
}

int fc_get_batch_info(fc_handle h, double* info) {
  if (!h || !info) return fail(FC_ERR_INVALID, "fc_get_batch_info: null argument");
  const fc_ctx::Batch& B = h->bat;
  int nblk = 0, nfold = 0;
  for (const fc_ctx::BLaunch& L : B.launches) (L.kind == 0 ? nblk : nfold)++;
  info[0] = B.k;
  info[1] = B.KB;
  info[2] = (double)B.scratch_rows;
  info[3] = nblk;
  info[4] = nfold;
  info[5] = 8.0 * (double)B.tiled_values;               // factor bytes STREAMED by one batched apply (tiled copy incl. zero padding; serves KB simulated steps)
  info[6] = 8.0 * B.vec_rows * (double)std::max(1, B.KB);  // operand / result / fold bytes of one batched apply
  info[7] = (double)B.tasks.n;
  return FC_OK;
}

int fc_bench_batch_apply(fc_handle h, int slot, int reps, double* ms_per_apply) {
  if (!h || slot < 0 || slot > 1 || reps <= 0 || !ms_per_apply) return fail(FC_ERR_INVALID, "fc_bench_batch_apply: bad argument");
  if (h->bat.k == 0) return fail(FC_ERR_NOT_READY, "fc_bench_batch_apply: fc_set_batch not called");
  OrderSys& S = h->sys[slot];
  if (!S.ready) return fail(FC_ERR_NOT_READY, "fc_setup_solver not called for this slot");
  HIPCHK(hipSetDevice(h->device));
  FCCHK(quiesce(h));
  fc_ctx::Batch& B = h->bat;
  const size_t bytes = (size_t)h->N * B.KB * sizeof(double);
  for (int i = 0; i < 2; ++i) {
    HIPCHK(hipMemcpyAsync(B.buf.p, B.b.p, bytes, hipMemcpyDeviceToDevice, h->stream));
    FCCHK(batch_apply(h, slot));
  }
  HIPCHK(hipEventRecord(h->ev0, h->stream));
  for (int i = 0; i < reps; ++i) {
    HIPCHK(hipMemcpyAsync(B.buf.p, B.b.p, bytes, hipMemcpyDeviceToDevice, h->stream));
    FCCHK(batch_apply(h, slot));
  }
  HIPCHK(hipEventRecord(h->ev1, h->stream));
  HIPCHK(hipEventSynchronize(h->ev1));
  float ms = 0.f;
  HIPCHK(hipEventElapsedTime(&ms, h->ev0, h->ev1));
  *ms_per_apply = (double)ms / reps;
  return FC_OK;
}

// parity hook: X = A_bc^-1 B for k right-hand sides through the batched factor apply (B, X: [k][N], W numbering)
int fc_solve_batch(fc_handle h, int slot, int32_t k, const double* b, double* x) {
  FCCHK(batch_check(h, k, "fc_solve_batch"));
  if (slot < 0 || slot > 1 || !b || !x) return fail(FC_ERR_INVALID, "fc_solve_batch: bad argument");
  OrderSys& S = h->sys[slot];
  if (!S.ready) return fail(FC_ERR_NOT_READY, "fc_setup_solver not called for this slot");
  if (S.inexact) return fail(FC_ERR_INVALID, "fc_solve_batch: this slot's factors are inexact (a preconditioner for GMRES): the batched path applies them directly");
  HIPCHK(hipSetDevice(h->device));
  FCCHK(quiesce(h));
  fc_ctx::Batch& B = h->bat;
  const int N = h->N;
  // permute on the host (setup-path helper): y_p[i] = b[perm[i]]
  std::vector<double> bp((size_t)k * N), xp((size_t)k * N);
  for (int s = 0; s < k; ++s)
    for (int i = 0; i < N; ++i) bp[(size_t)s * N + i] = b[(size_t)s * N + h->h_perm[i]];
  FCCHK(batch_copy(h, N, bp.data(), B.buf.p, true));
  FCCHK(batch_apply(h, slot));
  FCCHK(batch_copy(h, N, xp.data(), B.buf.p + (size_t)N * B.KB, false));
  for (int s = 0; s < k; ++s)
    for (int i = 0; i < N; ++i) x[(size_t)s * N + h->h_perm[i]] = xp[(size_t)s * N + i];
  return FC_OK;
}

}  // extern "C"
