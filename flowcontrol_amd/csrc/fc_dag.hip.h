// One-launch factor apply: the level sweeps of fc_kernels.hip.h as ONE grid in which every workgroup is a
// task of the elimination tree and waits for the tasks it depends on through per-node arrival counters
// (instead of a kernel boundary per tree level).  Included by fc_hip.hip only.
//
//   up task    rows of tree node t (segment lists, y[r] += sum -L . y_deeper): needs the up tasks of t's
//              children (by induction: of every descendant) to have arrived
//   down task  a tile of rows of node t ([D^-1 | -U] block x shared operand): needs the down tasks of t's
//              parent (by induction: of every ancestor); the root needs its own up tasks
//
// Tasks are numbered in topological order (level by level) and blockIdx.x = task number, so a task's
// dependencies have smaller block indices: they were handed to the dispatcher earlier and are resident or done.
// Every wait is nevertheless BOUNDED: a workgroup that gives up raises `err`, the step's tail then leaves the
// state untouched and the host redoes the step with the level launches (fc_hip.hip: dag_failed).
//
// Why this is faster than a launch per level (MI355X_MICROARCH.md price list): the factor VALUES of a task do
// not depend on any other task, so every workgroup issues its value loads (a whole tile, <= 24 fp64 per lane)
// BEFORE it waits; while one level's dependencies resolve, the value stream of the next levels is already in
// flight, and the wide levels of different sub-trees overlap freely.  A level costs one point-to-point hand-off
// instead of a launch boundary + a cold dependent-load chain.
//
// Hand-off protocol (cdna_hip_programming.md Guideline 16, form R1 with write-through payload): every word of
// the work buffer `buf` is stored with sc1 stores and loaded with sc1 loads inside this kernel (agent-scope
// relaxed atomics on 8-byte words: global_store/load_dwordx2 ... sc1, never through L1); a producer drains
// its stores (s_waitcnt vmcnt(0) in every wave), the workgroup meets at a barrier, ONE lane adds 1 to its
// node's counter (agent-scope atomic); a consumer polls the counters with ONE wave (relaxed agent-scope loads),
// the workgroup meets at a barrier, then every wave loads.  Counters are never reset: the target of the k-th
// apply is k x (tasks of the node), compared modulo 2^32.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

struct __attribute__((aligned(16))) FcDagTask {  // 64 bytes: everything a workgroup needs to issue its value loads
  int kind;   // 0: up rows (expanded sparse rows, accumulate into y); 1: down tile (dense rows, assign into x)
  int lpr;    // lanes per row (8 .. 256, power of two); a task covers 256 / lpr rows of ONE tree node
  int nrows;
  int dest0;  // buf index of the first row's result (up: the row in y; down: the row in x)
  int dep0, ndep;  // dependency records
  int sig;         // counter word this task adds 1 to when it is done
  int stride;      // values per row: down = ni + nb; up = padded row length of the tile (ELL)
  long long val;   // down: offset of the first row in the factor values; up: offset of the tile in up_val / up_col
  int i0, ni;      // down: operand = [ y[i0 .. i0+ni) | x[idx[ioff .. ioff+nb)] ]
  int ioff, nb;
  int pad[2];
};
struct __attribute__((aligned(16))) FcDagDep {
  int base;        // first counter word
  int nshard;      // words base, base + 32, ... (one 128-B line each)
  unsigned target; // arrivals per apply
  int pad;
};

#define FC_DAG_SHARD_STRIDE 32  // counter words per 128-byte line
#define FC_DAG_MAX_SHARDS 16
#define FC_DAG_PV 16            // fp64 values a lane holds in registers while it waits
#define FC_DAG_TILE 2048        // operand entries staged in LDS at a time
#define FC_DAG_SPIN_LIMIT 200000
#ifndef FC_DAG_SLEEP_MAX
#define FC_DAG_SLEEP_MAX 32  // poll back-off cap in units of 64 clocks (waiters of one node poll the same counter lines)
#endif
#ifndef FC_DAG_WAVES_PER_SIMD
#define FC_DAG_WAVES_PER_SIMD 3  // register budget for 3 resident workgroups of 256 threads per CU (the launch keeps it to that)
#endif

// wave 0 only: wait until every dependency's counters have reached  epoch x target  (mod 2^32).
// 16 lanes per dependency (one per shard), four dependencies per pass.
__device__ __forceinline__ bool fc_dag_wait(const FcDagDep* __restrict__ deps, int dep0, int ndep, const unsigned* cnt,
                                            unsigned epoch, int* err) {
  const int lane = threadIdx.x & 63;
  const int grp = lane >> 4, s = lane & 15;
  for (int d0 = 0; d0 < ndep; d0 += 4) {
    const int d = d0 + grp;
    const bool has = d < ndep;
    FcDagDep dp = {0, 0, 0u, 0};
    if (has) dp = deps[dep0 + d];
    const bool mine = has && s < dp.nshard;
    const unsigned* p = cnt + dp.base + (mine ? s : 0) * FC_DAG_SHARD_STRIDE;
    const unsigned want = dp.target * epoch;
    int spin = 0, nap = 1;
    for (;;) {
      unsigned v = mine ? __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
      v += __shfl_xor(v, 8, 16);
      v += __shfl_xor(v, 4, 16);
      v += __shfl_xor(v, 2, 16);
      v += __shfl_xor(v, 1, 16);
      const bool ok = !has || v == want;
      if (__all(ok)) break;
      if (++spin > FC_DAG_SPIN_LIMIT) {
        if (lane == 0) __hip_atomic_store(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return false;
      }
      if ((spin & 1023) == 0 && __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return false;
      for (int k = 0; k < nap; ++k) __builtin_amdgcn_s_sleep(1);
      if (nap < FC_DAG_SLEEP_MAX) nap *= 2;
    }
  }
  return true;
}

#ifdef FC_WITH_DAG  // the kernel is compiled only into builds that ask for it (hipcc -DFC_WITH_DAG): measured slower than the level launches on every mesh
// PERSISTENT: workgroup w of G runs tasks w, w + G, w + 2G, ... (topological order: a task's dependencies have
// smaller numbers, so they belong to workgroups that are resident and get to them first; all G workgroups must be
// resident, which the launch ensures by its size — and every wait is bounded anyway).  Software pipeline: the
// descriptor of the NEXT task is fetched while the current one runs, and its values are requested right after the
// current task's results are out — before the wait for its dependencies.
__global__ __launch_bounds__(256, FC_DAG_WAVES_PER_SIMD) void fc_nd_dag(
    const FcDagTask* __restrict__ tasks, int ntasks, const FcDagDep* __restrict__ deps, unsigned* cnt, unsigned epoch, int* err,
    const int* __restrict__ up_col, const double* __restrict__ up_val, const int* __restrict__ idx,
    const double* __restrict__ val, double* buf, fc_u64* trace) {
  // trace (diagnostic launches only, fc_debug_trace_apply): 100 MHz wall-clock stamps of thread 0 per task at
  // values requested / dependencies met / products done / stores drained
  __shared__ double xs[FC_DAG_TILE];
  __shared__ double part[4];
  __shared__ int go;
  const int tid = threadIdx.x;
  const int G = gridDim.x;
  int i = blockIdx.x;
  if (i >= ntasks) return;
#define FC_STAMP(k) \
  if (trace && tid == 0) trace[(size_t)i * 8 + (k)] = wall_clock64()
  constexpr int NG = FC_DAG_TILE / 256;
  FcDagTask t = tasks[i];
  double pv[FC_DAG_PV];
  int pa[FC_DAG_PV];  // up: buffer column of every value; down: [0, NG) the coupling part's index list
  // values (and index data) of task t into registers: depends on nothing but the descriptor.  Every load is
  // UNCONDITIONAL with a clamped address (a select around a load makes hipcc branch around it and wait at once)
  auto prefetch = [&]() {
    const int LPR = t.lpr;
    const int slot = tid / LPR, l = tid % LPR;
    const long long row = t.val + (long long)(slot < t.nrows ? slot : 0) * t.stride;
    if (t.kind == 1) {
      const double* __restrict__ v = val + row;
#pragma unroll
      for (int u = 0; u < FC_DAG_PV; ++u) {
        const int col = l + u * LPR;
        pv[u] = v[col < t.stride ? col : 0];
      }
      const int nbm1 = t.nb > 0 ? t.nb - 1 : 0;
#pragma unroll
      for (int k = 0; k < NG; ++k) {
        int o = tid + 256 * k - t.ni;
        o = o < 0 ? 0 : (o > nbm1 ? nbm1 : o);
        pa[k] = idx[t.ioff + o];
      }
    } else {
      const double* __restrict__ v = up_val + row;
      const int* __restrict__ c = up_col + row;
#pragma unroll
      for (int u = 0; u < FC_DAG_PV; ++u) {
        const int e = l + u * LPR;
        const int ee = e < t.stride ? e : 0;
        pv[u] = v[ee];
        pa[u] = c[ee];
      }
    }
  };
  prefetch();
  for (;;) {
    FC_STAMP(0);
    const int inext = i + G;
    FcDagTask tn = t;
    if (inext < ntasks) tn = tasks[inext];  // in flight while this task runs
    if (t.ndep > 0) {
      if (tid < 64) {
        const bool ok = fc_dag_wait(deps, t.dep0, t.ndep, cnt, epoch, err);
        if (tid == 0) go = ok ? 1 : 0;
      }
      __syncthreads();
      if (!go) return;
    }
    FC_STAMP(1);
    const int LPR = t.lpr;
    const int slot = tid / LPR, l = tid % LPR;
    const bool rowok = slot < t.nrows;
    double* dst = buf + t.dest0 + (rowok ? slot : 0);
    double acc = 0.0, own = 0.0;
    if (t.kind == 1) {
      // ── down tile: operand [ y[i0..i0+ni) | x[idx[..nb)] ] shared by all rows -> LDS (sc1 loads, all in flight) ──
      const int wd = t.stride;
      const double* __restrict__ vrow = val + t.val + (long long)(rowok ? slot : 0) * wd;
      const int npre = FC_DAG_PV * LPR;  // columns held in registers
      const int wdm1 = wd - 1;
      auto consume = [&](int t0, int tl) {
        if (t0 < npre) {
#pragma unroll
          for (int u = 0; u < FC_DAG_PV; ++u) {
            const int c = l + u * LPR - t0;
            const bool in = c >= 0 && c < tl;
            acc += (in ? pv[u] : 0.0) * xs[in ? c : 0];
          }
        }
        int c0 = t0 > npre ? t0 : npre;
        c0 += (l - (c0 % LPR) + LPR) % LPR;  // first column >= max(t0, npre) owned by this lane
        for (int col = c0; col < t0 + tl; col += LPR) acc += vrow[col] * xs[col - t0];
      };
      {
        const int tl = wd < FC_DAG_TILE ? wd : FC_DAG_TILE;
        double ov[NG];
#pragma unroll
        for (int k = 0; k < NG; ++k) {
          int col = tid + 256 * k;
          col = col > wdm1 ? wdm1 : col;
          ov[k] = fc_ld_sc1(buf + (col < t.ni ? t.i0 + col : pa[k]));
        }
#pragma unroll
        for (int k = 0; k < NG; ++k)
          if (tid + 256 * k < tl) xs[tid + 256 * k] = ov[k];
        __syncthreads();
        consume(0, tl);
      }
      for (int t0 = FC_DAG_TILE; t0 < wd; t0 += FC_DAG_TILE) {  // fronts wider than one tile (large meshes, near the root)
        const int tl = wd - t0 < FC_DAG_TILE ? wd - t0 : FC_DAG_TILE;
        __syncthreads();
        for (int j = tid; j < tl; j += 256) {
          const int col = t0 + j;
          xs[j] = fc_ld_sc1(buf + (col < t.ni ? t.i0 + col : idx[t.ioff + (col - t.ni)]));
        }
        __syncthreads();
        consume(t0, tl);
      }
    } else {
      // ── up rows: one trip of sc1 operand loads (all in flight), products with the values held in registers ──
      own = fc_ld_sc1(dst);
      double xv[FC_DAG_PV];
#pragma unroll
      for (int u = 0; u < FC_DAG_PV; ++u) xv[u] = fc_ld_sc1(buf + pa[u]);
#pragma unroll
      for (int u = 0; u < FC_DAG_PV; ++u) acc += (l + u * LPR < t.stride ? pv[u] : 0.0) * xv[u];
      const long long row = t.val + (long long)(rowok ? slot : 0) * t.stride;
      for (int e = l + FC_DAG_PV * LPR; e < t.stride; e += LPR) acc += up_val[row + e] * fc_ld_sc1(buf + up_col[row + e]);  // rows longer than the register tile
    }
    // row sums: lanes of a row are consecutive; LPR <= 64: inside the wave, 128 / 256: through LDS in a fixed order
    double s = acc;
    const int w = LPR < 64 ? LPR : 64;
    for (int off = w >> 1; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    if (LPR > 64) {
      if ((tid & 63) == 0) part[tid >> 6] = s;
      __syncthreads();
      if (l == 0 && rowok) {
        const int w0 = tid >> 6;
        fc_st_sc1(dst, own + (LPR == 128 ? part[w0] + part[w0 + 1] : (part[0] + part[1]) + (part[2] + part[3])));
      }
    } else if (l == 0 && rowok) {
      fc_st_sc1(dst, own + s);
    }
    FC_STAMP(2);
    // ── arrival: every wave drains its write-through stores, the workgroup meets, ONE lane signals ──
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    FC_STAMP(3);
    if (tid == 0) __hip_atomic_fetch_add(cnt + t.sig, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (inext >= ntasks) break;
    t = tn;
    i = inext;
    prefetch();  // the next task's values: requested before its dependencies are waited for
  }
#undef FC_STAMP
}
#endif  // FC_WITH_DAG
