// Micro-benchmark: what does a device-wide barrier cost on MI355X (256 CUs, 8 XCDs)?
// Decides whether the 11 dependent sweep launches of one factor apply (~5 us floor each) are worth
// replacing by one persistent kernel with grid barriers between levels.
//   hipcc --offload-arch=gfx950 -O3 scripts/micro/grid_barrier.hip -o gpurun_out/grid_barrier
#include <hip/hip_cooperative_groups.h>
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>
namespace cg = cooperative_groups;

#define CHK(x)                                                                  \
  do {                                                                          \
    hipError_t e_ = (x);                                                        \
    if (e_ != hipSuccess) {                                                     \
      std::printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__);     \
      return 1;                                                                 \
    }                                                                           \
  } while (0)

__global__ void k_cg(int iters, double* data, int n) {
  cg::grid_group g = cg::this_grid();
  const int tid = blockIdx.x * blockDim.x + threadIdx.x, nt = gridDim.x * blockDim.x;
  for (int it = 0; it < iters; ++it) {
    // a little dependent work: every thread reads what a thread of another workgroup wrote last round
    if (tid < n) data[(it & 1) * n + tid] = data[((it + 1) & 1) * n + (tid + 4099) % n] + 1.0;
    g.sync();
  }
  (void)nt;
}

// hand-rolled: monotone ticket counter, agent-scope release/acquire, bounded spin (exits on timeout)
__global__ void k_own(int iters, double* data, int n, unsigned int* ctr, int* timeout_flag) {
  const int tid = blockIdx.x * blockDim.x + threadIdx.x;
  const unsigned int G = gridDim.x;
  for (int it = 0; it < iters; ++it) {
    if (tid < n) data[(it & 1) * n + tid] = data[((it + 1) & 1) * n + (tid + 4099) % n] + 1.0;
    __syncthreads();
    if (threadIdx.x == 0) {
      __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned int target = G * (unsigned int)(it + 1);
      long spins = 0;
      while (__hip_atomic_load(ctr, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > 20000000L) {
          *timeout_flag = 1;
          break;
        }
      }
    }
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  }
}

__global__ void k_empty() {}

int main() {
  hipDeviceProp_t prop;
  CHK(hipGetDeviceProperties(&prop, 0));
  std::printf("%s CUs %d cooperative %d\n", prop.name, prop.multiProcessorCount, prop.cooperativeLaunch);
  hipStream_t st;
  CHK(hipStreamCreate(&st));
  const int n = 1 << 16;
  double* data;
  unsigned int* ctr;
  int* tflag;
  CHK(hipMalloc(&data, 2 * n * sizeof(double)));
  CHK(hipMemset(data, 0, 2 * n * sizeof(double)));
  CHK(hipMalloc(&ctr, 4));
  CHK(hipMalloc(&tflag, 4));
  hipEvent_t e0, e1;
  CHK(hipEventCreate(&e0));
  CHK(hipEventCreate(&e1));
  int iters = 200;
  for (int wgs_per_cu : {1, 2, 4}) {
    for (int threads : {256, 1024}) {
      int maxb = 0;
      CHK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&maxb, k_cg, threads, 0));
      if (wgs_per_cu > maxb) continue;
      const int grid = prop.multiProcessorCount * wgs_per_cu;
      int nn = n;
      double* dd = data;
      void* args[] = {&iters, &dd, &nn};
      CHK(hipLaunchCooperativeKernel((void*)k_cg, dim3(grid), dim3(threads), args, 0, st));
      CHK(hipStreamSynchronize(st));
      CHK(hipEventRecord(e0, st));
      CHK(hipLaunchCooperativeKernel((void*)k_cg, dim3(grid), dim3(threads), args, 0, st));
      CHK(hipEventRecord(e1, st));
      CHK(hipEventSynchronize(e1));
      float ms;
      CHK(hipEventElapsedTime(&ms, e0, e1));
      std::printf("cg::grid.sync  grid %5d x %4d: %.2f us per barrier(+tiny work)\n", grid, threads, 1e3 * ms / iters);
      std::fflush(stdout);
      // own barrier (launched cooperatively as well, so co-residency is guaranteed)
      CHK(hipMemsetAsync(ctr, 0, 4, st));
      CHK(hipMemsetAsync(tflag, 0, 4, st));
      void* args2[] = {&iters, &dd, &nn, &ctr, &tflag};
      CHK(hipLaunchCooperativeKernel((void*)k_own, dim3(grid), dim3(threads), args2, 0, st));
      CHK(hipStreamSynchronize(st));
      CHK(hipMemsetAsync(ctr, 0, 4, st));
      CHK(hipEventRecord(e0, st));
      CHK(hipLaunchCooperativeKernel((void*)k_own, dim3(grid), dim3(threads), args2, 0, st));
      CHK(hipEventRecord(e1, st));
      CHK(hipEventSynchronize(e1));
      CHK(hipEventElapsedTime(&ms, e0, e1));
      int tf = 0;
      CHK(hipMemcpy(&tf, tflag, 4, hipMemcpyDeviceToHost));
      std::printf("own ticket     grid %5d x %4d: %.2f us per barrier(+tiny work) timeout=%d\n", grid, threads, 1e3 * ms / iters, tf);
      std::fflush(stdout);
    }
  }
  // reference: dependent empty launches
  for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(k_empty, dim3(1024), dim3(256), 0, st);
  CHK(hipStreamSynchronize(st));
  CHK(hipEventRecord(e0, st));
  for (int i = 0; i < 1000; ++i) hipLaunchKernelGGL(k_empty, dim3(1024), dim3(256), 0, st);
  CHK(hipEventRecord(e1, st));
  CHK(hipEventSynchronize(e1));
  float ms;
  CHK(hipEventElapsedTime(&ms, e0, e1));
  std::printf("empty kernel 1024x256 back-to-back: %.2f us per launch\n", ms);
  return 0;
}
