// stream_lab.hip -- what does the MI355X memory system deliver for the Jacobi traffic mix (2 streams read, 1 written)?
// Calibration for roofline.frac: pure float4 streaming kernels over 512^3-sized arrays, timed with HIP events.
//   hipcc --offload-arch=gfx950 -O3 tools/stream_lab.hip -o gpurun_out/stream_lab && gpurun_out/stream_lab
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__global__ void __launch_bounds__(256) k_copy(const float4* __restrict__ a, float4* __restrict__ c, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) c[i] = a[i];
}
__global__ void __launch_bounds__(256) k_triad(const float4* __restrict__ a, const float4* __restrict__ b, float4* __restrict__ c, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    float4 x = a[i], y = b[i];
    c[i] = make_float4(x.x + 0.8f * y.x, x.y + 0.8f * y.y, x.z + 0.8f * y.z, x.w + 0.8f * y.w);
  }
}
__global__ void __launch_bounds__(256) k_triad_nt(const float4* __restrict__ a, const float4* __restrict__ b, float4* __restrict__ c, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    typedef float v4 __attribute__((ext_vector_type(4)));
    v4 x = __builtin_nontemporal_load((const v4*)&a[i]), y = __builtin_nontemporal_load((const v4*)&b[i]);
    v4 r = x + 0.8f * y;
    __builtin_nontemporal_store(r, (v4*)&c[i]);
  }
}
__global__ void __launch_bounds__(256) k_read2(const float4* __restrict__ a, const float4* __restrict__ b, float* __restrict__ out, size_t n) {
  float s = 0.f;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    float4 x = a[i], y = b[i];
    s += x.x + x.y + x.z + x.w + y.x + y.y + y.z + y.w;
  }
  if (s == 123.456f) out[0] = s;
}
__global__ void __launch_bounds__(256) k_write(float4* __restrict__ c, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) c[i] = make_float4(1.f, 2.f, 3.f, 4.f);
}
// contiguous chunk per block instead of grid-stride (each block streams its own 1/grid slice)
__global__ void __launch_bounds__(256) k_triad_chunk(const float4* __restrict__ a, const float4* __restrict__ b, float4* __restrict__ c, size_t n) {
  const size_t per = (n + gridDim.x - 1) / gridDim.x;
  const size_t lo = (size_t)blockIdx.x * per, hi = lo + per < n ? lo + per : n;
  for (size_t i = lo + threadIdx.x; i < hi; i += 256) {
    float4 x = a[i], y = b[i];
    c[i] = make_float4(x.x + 0.8f * y.x, x.y + 0.8f * y.y, x.z + 0.8f * y.z, x.w + 0.8f * y.w);
  }
}

int main() {
  const size_t n = (size_t)516 * 516 * 516 / 4;  // float4 count of one 512^3 S3D array
  float4 *a, *b, *c;
  float* out;
  CK(hipMalloc(&a, n * 16)); CK(hipMalloc(&b, n * 16)); CK(hipMalloc(&c, n * 16)); CK(hipMalloc(&out, 4));
  CK(hipMemset(a, 0, n * 16)); CK(hipMemset(b, 0, n * 16)); CK(hipMemset(c, 0, n * 16));
  hipLaunchKernelGGL(k_write, dim3(4096), dim3(256), 0, 0, a, n);
  hipLaunchKernelGGL(k_write, dim3(4096), dim3(256), 0, 0, b, n);
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int grids[] = {1024, 2048, 4096, 8192, 16384, 65536};
  printf("%-14s %7s %9s %9s\n", "kernel", "grid", "ms(med)", "GB/s");
  for (int which = 0; which < 6; which++) {
    for (int g : grids) {
      std::vector<float> ts;
      for (int rep = 0; rep < 7; rep++) {
        CK(hipEventRecord(e0, 0));
        double bytes = 0;
        const char* name = "";
        switch (which) {
          case 0: hipLaunchKernelGGL(k_copy, dim3(g), dim3(256), 0, 0, a, c, n); bytes = 2.0 * n * 16; name = "copy"; break;
          case 1: hipLaunchKernelGGL(k_triad, dim3(g), dim3(256), 0, 0, a, b, c, n); bytes = 3.0 * n * 16; name = "triad(2r1w)"; break;
          case 2: hipLaunchKernelGGL(k_triad_nt, dim3(g), dim3(256), 0, 0, a, b, c, n); bytes = 3.0 * n * 16; name = "triad_nt"; break;
          case 3: hipLaunchKernelGGL(k_read2, dim3(g), dim3(256), 0, 0, a, b, out, n); bytes = 2.0 * n * 16; name = "read2"; break;
          case 4: hipLaunchKernelGGL(k_write, dim3(g), dim3(256), 0, 0, c, n); bytes = 1.0 * n * 16; name = "write"; break;
          case 5: hipLaunchKernelGGL(k_triad_chunk, dim3(g), dim3(256), 0, 0, a, b, c, n); bytes = 3.0 * n * 16; name = "triad_chunk"; break;
        }
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        ts.push_back(ms);
        if (rep == 6) {
          std::sort(ts.begin(), ts.end());
          printf("%-14s %7d %9.4f %9.0f\n", name, g, ts[3], bytes / (ts[3] * 1e-3) / 1e9);
        }
      }
    }
  }
  return 0;
}
