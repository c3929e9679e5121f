// Micro-benchmark: practical read bandwidth of one MI355X for a plain streaming reduction, as a
// calibration of the 8 TB/s roofline the sweep kernels are quoted against.
//   hipcc --offload-arch=gfx950 -O3 scripts/micro/stream_read.hip -o /tmp/stream_read && /tmp/stream_read
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

template <int W>  // doubles per load (1: 8 B, 2: 16 B), U loads in flight per lane
__global__ __launch_bounds__(256) void k_sum(const double* __restrict__ a, size_t n, double* __restrict__ out) {
  constexpr int U = 4;
  const size_t stride = (size_t)gridDim.x * blockDim.x * W;
  size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * W;
  double s = 0.0;
  for (; i + (U - 1) * stride + W <= n; i += U * stride) {
    if (W == 1) {
      double v[U];
#pragma unroll
      for (int u = 0; u < U; ++u) v[u] = a[i + u * stride];
#pragma unroll
      for (int u = 0; u < U; ++u) s += v[u];
    } else {
      double2 v[U];
#pragma unroll
      for (int u = 0; u < U; ++u) v[u] = *reinterpret_cast<const double2*>(a + i + u * stride);
#pragma unroll
      for (int u = 0; u < U; ++u) s += v[u].x + v[u].y;
    }
  }
  for (; i + W <= n; i += stride)
    for (int w = 0; w < W; ++w) s += a[i + w];
  if (s == 123.456) out[0] = s;  // keep the loads
}

int main() {
  hipStream_t st;
  CHK(hipStreamCreate(&st));
  hipEvent_t e0, e1;
  CHK(hipEventCreate(&e0));
  CHK(hipEventCreate(&e1));
  double* out;
  CHK(hipMalloc(&out, 8));
  for (size_t mb : {100, 1000, 4000}) {
    const size_t n = mb * 1000 * 1000 / 8;
    double* a;
    CHK(hipMalloc(&a, n * 8));
    CHK(hipMemset(a, 0, n * 8));
    for (int width : {1, 2}) {
      for (int wg_per_cu : {4, 8, 16, 32}) {
        const int grid = 256 * wg_per_cu;
        auto launch = [&]() {
          if (width == 1) hipLaunchKernelGGL(k_sum<1>, dim3(grid), dim3(256), 0, st, a, n, out);
          else hipLaunchKernelGGL(k_sum<2>, dim3(grid), dim3(256), 0, st, a, n, out);
        };
        for (int i = 0; i < 3; ++i) launch();
        CHK(hipStreamSynchronize(st));
        const int reps = mb >= 1000 ? 10 : 50;
        CHK(hipEventRecord(e0, st));
        for (int i = 0; i < reps; ++i) launch();
        CHK(hipEventRecord(e1, st));
        CHK(hipEventSynchronize(e1));
        float ms;
        CHK(hipEventElapsedTime(&ms, e0, e1));
        std::printf("%5zu MB  %2d B/lane  %2d WG/CU : %.2f TB/s\n", mb, 8 * width, wg_per_cu, (double)n * 8 * reps / (ms * 1e-3) / 1e12);
      }
    }
    CHK(hipFree(a));
  }
  return 0;
}
