// pk_lab: issue cost of packed FP32 vector instructions on gfx950 (v_pk_mul_f32 / v_pk_add_f32 / v_pk_mov_b32) against the
// single-lane forms, in a stream of independent instructions, at 1, 2 and 4 waves per SIMD.  Cycles per instruction per wave.
//   hipcc --offload-arch=gfx950 -O3 -o tools/bin/pk_lab tools/pk_lab.hip && tools/bin/pk_lab
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float v2f __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ void k(float* out, long long* cyc, int iters) {
  float a[8];
  v2f p[8];
  const float s = out[threadIdx.x & 63] + 1.0f;
#pragma unroll
  for (int i = 0; i < 8; i++) { a[i] = s + i; p[i] = v2f{s + i, s - i}; }
  const v2f sc = v2f{1.0000001f, 0.9999999f};
  __syncthreads();
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < 8; i++) {
      if (MODE == 0) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(sc.x));
      if (MODE == 1) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(sc));
      if (MODE == 2) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(sc.x));
      if (MODE == 3) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(sc));
      if (MODE == 4) asm volatile("v_pk_mov_b32 %0, %0, %1 op_sel:[1,0]" : "+v"(p[i]) : "v"(sc));
      if (MODE == 5) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p[i]) : "v"(sc));
      if (MODE == 6) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(sc.x));
      if (MODE == 7) asm volatile("v_pk_mul_f32 %0, %0, %1 op_sel_hi:[1,0]" : "+v"(p[i]) : "s"(sc));
    }
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  float r = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) r += a[i] + p[i].x + p[i].y;
  if (r == 12345.678f) out[0] = r;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

template <int MODE>
void run(const char* name, float* d, long long* c) {
  const int iters = 4096;
  for (int waves : {1, 2, 4}) {  // per SIMD
    const int threads = 64 * 4 * waves;
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 0, 0, d, c, iters);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 0, 0, d, c, iters);
    hipDeviceSynchronize();
    std::vector<long long> h(256 * threads / 64);
    hipMemcpy(h.data(), c, h.size() * sizeof(long long), hipMemcpyDeviceToHost);
    double sum = 0;
    for (auto x : h) sum += (double)x;
    const double per = sum / h.size() / (iters * 8.0);
    printf("%-34s waves/SIMD %d: %6.2f cycles per instruction per wave -> %5.2f cycles of the SIMD per instruction\n", name, waves, per, per / waves);
  }
}
int main() {
  float* d; long long* c;
  hipMalloc(&d, 4096); hipMemset(d, 0, 4096); hipMalloc(&c, 1 << 20);
  run<0>("v_mul_f32", d, c);
  run<1>("v_pk_mul_f32", d, c);
  run<7>("v_pk_mul_f32 (sgpr pair, op_sel_hi)", d, c);
  run<2>("v_add_f32", d, c);
  run<3>("v_pk_add_f32", d, c);
  run<6>("v_fma_f32", d, c);
  run<5>("v_pk_fma_f32", d, c);
  run<4>("v_pk_mov_b32", d, c);
  return 0;
}
