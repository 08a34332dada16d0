// div_lab: the short form of the division by a loop-invariant divisor (cz_k_fastdiv.h, shortdiv) against `n / d` for ALL 2^32 float
// numerators, for a list of divisors; and the hoisted form as a control.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -std=c++17 -Icubez_amd/csrc -Iinclude tools/div_lab.hip -o tools/bin/div_lab
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include "cz_internal.h"
typedef CZ_REAL REAL;
namespace {
#include "cz_k_fastdiv.h"
}
int main(int argc, char** argv) {
  unsigned long long* bad;
  (void)hipMalloc(&bad, 16);
  const float ds[] = {6.0f, -6.0f, 3.0f, 7.0f, 1.5f, 0.1f, 1e-3f, 1e10f, 6.0000005f, 5.9999995f, 1.0f, 2.0f, 0.75f, 1e-20f, 1e20f, 12.566371f, 1.0000001f, 1.9999999f};
  for (float d : ds) {
    for (int sh = 0; sh < 3; sh++) {
      (void)hipMemset(bad, 0, 16);
      hipEvent_t a, b;
      (void)hipEventCreate(&a); (void)hipEventCreate(&b);
      (void)hipEventRecord(a);
      if (sh == 2) hipLaunchKernelGGL(fastdiv_check_k<2>, dim3(4096), dim3(256), 0, 0, d, bad);
      else if (sh) hipLaunchKernelGGL(fastdiv_check_k<1>, dim3(4096), dim3(256), 0, 0, d, bad);
      else hipLaunchKernelGGL(fastdiv_check_k<0>, dim3(4096), dim3(256), 0, 0, d, bad);
      (void)hipEventRecord(b);
      (void)hipDeviceSynchronize();
      float ms = 0; (void)hipEventElapsedTime(&ms, a, b);
      unsigned long long h = 0;
      (void)hipMemcpy(&h, bad, 8, hipMemcpyDeviceToHost);
      printf("d = %-14.9g %s form: %llu of 2^32 numerators differ from n / d  (%.1f ms)\n", d, sh == 2 ? "medium " : sh ? "short  " : "hoisted", h, ms);
    }
  }
  return 0;
}
