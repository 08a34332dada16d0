// fetch_lab.hip -- what does FETCH_SIZE count for the access shape of pcr_line_reg_k?  ("whether FETCH_SIZE needs the x2 here
// is not established".)  MI355X_MICROARCH.md calibrates the counter for wide coalesced streams only (16 B per lane: FETCH_SIZE reports half
// the bytes); the register form of the line solvers reads RUNS -- every lane M = 8 consecutive floats of its k-line as two dword-aligned
// global_load_dwordx4 (a line starts at padded element 3, so nothing is 16-byte aligned), lanes 32 B apart.  Every kernel below reads each
// byte of one 516^3 float array exactly once (the same rows, the same bytes), so the factor between the counter and the known byte count
// is the correction for that shape:
//     rocprofv3 --pmc FETCH_SIZE -d out --output-format csv -- tools/bin/fetch_lab
//   hipcc --offload-arch=gfx950 -O3 tools/fetch_lab.hip -o tools/bin/fetch_lab
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CK(x)                                                                           \
  do {                                                                                  \
    hipError_t e_ = (x);                                                                \
    if (e_ != hipSuccess) {                                                             \
      printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__);                  \
      exit(1);                                                                          \
    }                                                                                   \
  } while (0)

typedef float Run4 __attribute__((ext_vector_type(4), aligned(4)));  // dword-aligned dwordx4, as load_run of cz_k_linesor.h
constexpr int NKP = 516;                                              // padded row length of a 512^3 grid

// (a) the calibrated shape: 16 B per lane, fully coalesced, 16-byte aligned
__global__ void __launch_bounds__(256) fetch_coalesced16(const float4* __restrict__ a, size_t n4, float* out) {
  float s = 0.f;
  for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    const float4 x = a[i];
    s += x.x + x.y + x.z + x.w;
  }
  if (s == 123.456f) out[0] = s;
}
// (b) 4 B per lane, coalesced
__global__ void __launch_bounds__(256) fetch_coalesced4(const float* __restrict__ a, size_t n, float* out) {
  float s = 0.f;
  for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) s += a[i];
  if (s == 123.456f) out[0] = s;
}
// (c) the shape of pcr_line_reg_k: one wave per row, lane l reads elements off + 8 l .. off + 8 l + 7 of the row as two dword-aligned dwordx4
// (off = 3: the first inner k of a line).  64 x 8 = 512 of the row's 516 elements; the 4 left over are read by lane 0 .. 3 with a dword load so
// that every byte is read once.
template <int off>
__global__ void __launch_bounds__(256) fetch_runs8(const float* __restrict__ a, size_t rows, float* out) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  float s = 0.f;
  for (size_t r = blockIdx.x * (size_t)4 + wave; r < rows; r += (size_t)gridDim.x * 4) {
    const float* row = a + r * NKP;
    const float* p = row + off + lane * 8;
    if (off + lane * 8 + 8 <= NKP) {
      const Run4 x = *reinterpret_cast<const Run4*>(p);
      const Run4 y = *reinterpret_cast<const Run4*>(p + 4);
      s += x[0] + x[1] + x[2] + x[3] + y[0] + y[1] + y[2] + y[3];
    } else {
      for (int c = 0; c < 8; c++)
        if (off + lane * 8 + c < NKP) s += p[c];
    }
    if (lane < off) s += row[lane];
  }
  if (s == 123.456f) out[0] = s;
}
// (d) the same runs, 16-byte aligned (off = 0 with rows of 516 floats = 2064 B: every row start is 16-byte aligned)
// -> fetch_runs8<0>

int main(int argc, char** argv) {
  const size_t rows = (size_t)516 * 516, n = rows * NKP;
  float *a, *out;
  CK(hipMalloc(&a, n * sizeof(float)));
  CK(hipMalloc(&out, 4));
  CK(hipMemset(a, 0, n * sizeof(float)));
  const int reps = argc > 1 ? atoi(argv[1]) : 4;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  printf("array: %zu floats = %.1f MB, every kernel reads each byte once\n", n, n * 4e-6);
  for (int which = 0; which < 4; which++) {
    float best = 1e30f;
    const char* name = "";
    for (int rep = 0; rep < reps; rep++) {
      CK(hipEventRecord(e0, 0));
      switch (which) {
        case 0: hipLaunchKernelGGL(fetch_coalesced16, dim3(8192), dim3(256), 0, 0, reinterpret_cast<const float4*>(a), n / 4, out); name = "fetch_coalesced16"; break;
        case 1: hipLaunchKernelGGL(fetch_coalesced4, dim3(8192), dim3(256), 0, 0, a, n, out); name = "fetch_coalesced4"; break;
        case 2: hipLaunchKernelGGL(fetch_runs8<3>, dim3(8192), dim3(256), 0, 0, a, rows, out); name = "fetch_runs8 (offset 3: pcr_line_reg_k)"; break;
        case 3: hipLaunchKernelGGL(fetch_runs8<0>, dim3(8192), dim3(256), 0, 0, a, rows, out); name = "fetch_runs8 (offset 0: 16-byte aligned)"; break;
      }
      CK(hipEventRecord(e1, 0));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      if (ms < best) best = ms;
    }
    printf("%-42s %.3f ms  %.0f GB/s\n", name, best, n * 4.0 / best * 1e-6);
  }
  return 0;
}
