// Micro-benchmark: what does a barrier among the workgroups of ONE XCD cost on MI355X (8 XCDs x 32 CUs; workgroup b runs on XCD b % 8)?
// Decides whether tree levels whose sub-trees can be pinned to an XCD may be separated by such a barrier inside a persistent kernel
// instead of a kernel boundary (~3.5 us floor per sweep launch).  Data written before the barrier by one CU and read after it by another CU of
// the same XCD goes through that XCD's L2: stores are write-through (drained with s_waitcnt), the reader invalidates its L1 (buffer_inv sc0... here: an
// agent-scope acquire fence, the portable form) -- the counter itself is an L2 atomic.
//   hipcc --offload-arch=gfx950 -O3 scripts/micro/xcd_barrier.hip -o gpurun_out/xcd_barrier && gpurun_out/xcd_barrier
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

#define CHK(x)                                                              \
  do {                                                                      \
    hipError_t e_ = (x);                                                    \
    if (e_ != hipSuccess) {                                                 \
      std::printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); \
      return 1;                                                             \
    }                                                                       \
  } while (0)

// MODE 0: relaxed agent-scope atomics, no fences (cost of the rendezvous alone)
// MODE 1: + release (drain) before arriving, agent-scope acquire fence after (what correct hand-over of plain stores needs, portable form)
// MODE 2: + drain before, workgroup-scope fence after and the dependent loads done as agent-scope atomic loads (bypass L1)
// MODE 3: as 2 with a two-level rendezvous: groups of 16 members count on their own line, the last of a group counts on the team's line
template <int MODE>
__global__ void k_team(int iters, double* data, int n, unsigned int* ctr, int* timeout_flag, unsigned long long* xcc_out) {
  const int team = blockIdx.x & 7, member = blockIdx.x >> 3, team_size = gridDim.x >> 3;
  unsigned int* c = ctr + 64 * team;  // one 256-byte line per team
  if (threadIdx.x == 0 && member == 0) {
    unsigned xcc = 0;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    xcc_out[team] = xcc & 0xf;
  }
  const int per_team = n / 8;
  double* d = data + (size_t)team * per_team * 2;
  const int tid = member * blockDim.x + threadIdx.x, nt = team_size * blockDim.x;
  for (int it = 0; it < iters; ++it) {
    // a little dependent work: every thread reads what a thread of ANOTHER workgroup of its team wrote last round
    for (int i = tid; i < per_team; i += nt) {
      const double* src = d + ((it + 1) & 1) * per_team + (i + 4099) % per_team;
      double v;
      if (MODE >= 2) {
        v = __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<const unsigned long long*>(src), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
      } else {
        v = *src;
      }
      d[(it & 1) * per_team + i] = v + 1.0;
    }
    if (MODE >= 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
      unsigned int target = (unsigned int)team_size * (unsigned int)(it + 1);
      if (MODE == 3) {
        const int ngroups = team_size / 16;
        unsigned int* gc = ctr + 64 * 8 + 64 * (team * 64 + member / 16);
        const unsigned int old = __hip_atomic_fetch_add(gc, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((old & 15u) == 15u) __hip_atomic_fetch_add(c, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        target = (unsigned int)ngroups * (unsigned int)(it + 1);
      } else {
        __hip_atomic_fetch_add(c, 1u, MODE == 1 ? __ATOMIC_RELEASE : __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      long spins = 0;
      while (__hip_atomic_load(c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > 20000000L) {
          *timeout_flag = 1;
          break;
        }
      }
    }
    __syncthreads();
    if (MODE == 1) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  }
}

int main() {
  hipDeviceProp_t prop;
  CHK(hipGetDeviceProperties(&prop, 0));
  std::printf("%s CUs %d\n", prop.name, prop.multiProcessorCount);
  hipStream_t st;
  CHK(hipStreamCreate(&st));
  const int n = 8 * 8192;
  double* data;
  unsigned int* ctr;
  int* tflag;
  unsigned long long* xcc;
  CHK(hipMalloc(&data, 2 * n * sizeof(double)));
  CHK(hipMalloc(&ctr, (8 * 64 + 8 * 64 * 64) * 4));
  CHK(hipMalloc(&tflag, 4));
  CHK(hipMalloc(&xcc, 64));
  hipEvent_t e0, e1;
  CHK(hipEventCreate(&e0));
  CHK(hipEventCreate(&e1));
  const int iters = 400;
  for (int wgs_per_cu : {1, 2, 4, 8}) {
    for (int threads : {256, 1024}) {
      if (wgs_per_cu * threads > 2048) continue;
      const int grid = prop.multiProcessorCount * wgs_per_cu;  // all resident: 256 CUs x wgs_per_cu (cooperative launch guarantees it)
      for (int mode = 0; mode < 4; ++mode) {
        if (mode == 1 || (mode == 3 && (grid / 8) % 16)) continue;
        float best = 1e30f;
        int tf = 0;
        for (int rep = 0; rep < 3; ++rep) {
          CHK(hipMemsetAsync(data, 0, 2 * n * sizeof(double), st));
          CHK(hipMemsetAsync(ctr, 0, (8 * 64 + 8 * 64 * 64) * 4, st));
          CHK(hipMemsetAsync(tflag, 0, 4, st));
          int it = iters, nn = n;
          void* args[] = {&it, &data, &nn, &ctr, &tflag, &xcc};
          const void* fn = mode == 0 ? (const void*)k_team<0> : (mode == 1 ? (const void*)k_team<1> : (mode == 2 ? (const void*)k_team<2> : (const void*)k_team<3>));
          CHK(hipEventRecord(e0, st));
          CHK(hipLaunchCooperativeKernel(fn, dim3(grid), dim3(threads), args, 0, st));
          CHK(hipEventRecord(e1, st));
          CHK(hipEventSynchronize(e1));
          float ms;
          CHK(hipEventElapsedTime(&ms, e0, e1));
          best = ms < best ? ms : best;
          CHK(hipMemcpy(&tf, tflag, 4, hipMemcpyDeviceToHost));
        }
        // check: after `iters` rounds every entry of the last written half equals iters (each round adds 1 to a value of the round before)
        std::vector<double> hd(2 * n);
        CHK(hipMemcpy(hd.data(), data, 2 * n * sizeof(double), hipMemcpyDeviceToHost));
        int bad = 0;
        const int per_team = n / 8;
        for (int t = 0; t < 8; ++t)
          for (int i = 0; i < per_team; ++i) bad += hd[(size_t)t * per_team * 2 + ((iters - 1) & 1) * per_team + i] != (double)iters;
        unsigned long long hx[8];
        CHK(hipMemcpy(hx, xcc, 64, hipMemcpyDeviceToHost));
        std::printf("team barrier mode %d  grid %4d x %4d (%2d per team): %.2f us per round  wrong %d timeout %d  xcc of teams %llu%llu%llu%llu%llu%llu%llu%llu\n", mode, grid, threads,
                    grid / 8, 1e3 * best / iters, bad, tf, hx[0], hx[1], hx[2], hx[3], hx[4], hx[5], hx[6], hx[7]);
        std::fflush(stdout);
      }
    }
  }
  return 0;
}
