// Calibration of rocprofv3's FETCH_SIZE for the access widths of this repo's kernels (MI355X_MICROARCH.md calibrates 16 B per lane only:
// "FETCH_SIZE reports exactly 1/2 of the bytes of a wide coalesced streaming read ... other access widths are uncalibrated").
// Every kernel below reads a KNOWN byte count once from a 2 GB buffer (far beyond the 256 MiB Infinity Cache):
//   k_read16   16 B per lane, a wave reads 1 KB contiguous          (the guide's case)
//   k_read8     8 B per lane, a wave reads 512 B contiguous         (fc_spmv_csr values, fc_nd_sweep segments)
//   k_rows8     8 lanes x 8 B = 64-B pieces, 8 pieces per wave from 8 rows 4 KB apart   (8-lane rows: fc_nd_down_block<8,...>, CSR rows)
//   k_rows16   16 lanes x 8 B = 128-B pieces from rows 4 KB apart   (16-lane rows)
//   k_read8nt   as k_read8 with nontemporal loads                   (the streamed factors: S.nt)
//   hipcc --offload-arch=gfx950 -O3 scripts/micro/fetch_calib.hip -o /tmp/fetch_calib
//   rocprofv3 --pmc FETCH_SIZE --kernel-trace -d out -- /tmp/fetch_calib      (scripts/fetch_calib.sh)
#include <hip/hip_runtime.h>

#include <cstdio>

#define CHK(x)                                                              \
  do {                                                                      \
    hipError_t e_ = (x);                                                    \
    if (e_ != hipSuccess) {                                                 \
      std::printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); \
      return 1;                                                             \
    }                                                                       \
  } while (0)

__global__ __launch_bounds__(256) void k_read16(const double* __restrict__ a, size_t n, double* __restrict__ out) {
  double s = 0.0;
  for (size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 2; i + 2 <= n; i += (size_t)gridDim.x * blockDim.x * 2) {
    const double2 v = *reinterpret_cast<const double2*>(a + i);
    s += v.x + v.y;
  }
  if (s == 123.456) out[0] = s;
}
__global__ __launch_bounds__(256) void k_read8(const double* __restrict__ a, size_t n, double* __restrict__ out) {
  double s = 0.0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) s += a[i];
  if (s == 123.456) out[0] = s;
}
__global__ __launch_bounds__(256) void k_read8nt(const double* __restrict__ a, size_t n, double* __restrict__ out) {
  double s = 0.0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) s += __builtin_nontemporal_load(a + i);
  if (s == 123.456) out[0] = s;
}
// rows of 512 doubles (4 KB); LANES lanes walk one row, 64 / LANES rows per wave
template <int LANES>
__global__ __launch_bounds__(256) void k_rows(const double* __restrict__ a, size_t nrows, double* __restrict__ out) {
  const int lane = threadIdx.x % LANES;
  double s = 0.0;
  for (size_t r = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) / LANES; r < nrows; r += (size_t)gridDim.x * blockDim.x / LANES)
    for (int j = lane; j < 512; j += LANES) s += a[r * 512 + j];
  if (s == 123.456) out[0] = s;
}

int main() {
  const size_t n = (size_t)2 << 27;  // 2^28 doubles = 2 GiB
  double *a, *out;
  CHK(hipMalloc(&a, n * 8));
  CHK(hipMalloc(&out, 8));
  CHK(hipMemset(a, 0, n * 8));
  CHK(hipDeviceSynchronize());
  const int grid = 256 * 16;
  hipLaunchKernelGGL(k_read16, dim3(grid), dim3(256), 0, 0, a, n, out);
  hipLaunchKernelGGL(k_read8, dim3(grid), dim3(256), 0, 0, a, n, out);
  hipLaunchKernelGGL(k_read8nt, dim3(grid), dim3(256), 0, 0, a, n, out);
  hipLaunchKernelGGL(k_rows<8>, dim3(grid), dim3(256), 0, 0, a, n / 512, out);
  hipLaunchKernelGGL(k_rows<16>, dim3(grid), dim3(256), 0, 0, a, n / 512, out);
  CHK(hipDeviceSynchronize());
  std::printf("bytes read by every kernel: %zu\n", n * 8);
  return 0;
}
