// tools/pair_lab.hip -- stand-alone A/B bench of the two-stage pass kernels (jacobi2_k vs jacobi2p_k) on random fields:
// bitwise comparison of the output field, both residual sums, ms per launch.  Compiles in seconds (the library's translation unit
// takes minutes), which is what the kernel work of round 2 iterated on.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-slp-vectorize -std=c++17 -Icubez_amd/csrc -Iinclude -Itools tools/pair_lab.hip -o tools/bin/pair_lab [-DCZ_REAL_IS_DOUBLE]
//   tools/bin/pair_lab N reps rb ni nj nk TBxTJ [TBxTJ ...]      (ni = 0: cube of N; -DCZ_P2_PLAIN_DIV: jacobi2p_k with the ordinary division)
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "cz_internal.h"

typedef CZ_REAL REAL;
#ifdef CZ_REAL_IS_DOUBLE
constexpr int VW = 2;
#else
constexpr int VW = 4;
#endif

namespace {
#include "cz_k_common.h"
#include "cz_k_fastdiv.h"
#include "cz_k_pair.h"
#include "pair_lab_v1.h"
#include "cz_k_pair2.h"

__global__ void fill_k(REAL* x, size_t n, unsigned seed, REAL scale) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    unsigned h = (unsigned)i * 2654435761u ^ seed;
    h ^= h >> 16, h *= 0x85ebca6bu, h ^= h >> 13, h *= 0xc2b2ae35u, h ^= h >> 16;
    x[i] = scale * ((REAL)(h & 0xffffff) / (REAL)0x800000 - (REAL)1.0);
  }
}
__global__ void diff_k(const REAL* a, const REAL* b, size_t n, unsigned long long* cnt) {
  unsigned long long c = 0;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    if (sizeof(REAL) == 4 ? (reinterpret_cast<const unsigned*>(a)[i] != reinterpret_cast<const unsigned*>(b)[i])
                          : (reinterpret_cast<const unsigned long long*>(a)[i] != reinterpret_cast<const unsigned long long*>(b)[i]))
      c++;
  if (c) atomicAdd(cnt, c);
}

struct Lab {
  int ni, nj, nk, nip, njp, nkp;
  REAL *U, *B, *W1, *W2;
  double *partials, *dst;
  unsigned* counter;
  Geom2 g;
  Coef c;
  int nblk;
  size_t lds1, lds2;
};

template <int TB, int MV>
bool geom(Lab& L, int tj, int rb) {
  constexpr int V = VW;
  Geom2& g = L.g;
  g.R = L.nkp / V;
  if (2 * g.R >= TB * MV / 2 || 2 * g.R > TB) return false;
  g.PSV = (long long)g.R * L.nip;
  g.nkp = L.nkp, g.PSB = (long long)L.nkp * L.nip * (long long)sizeof(REAL), g.jlast = L.njp - 1, g.last_off = (unsigned)(g.PSB - (long long)sizeof(Vec<V>));
  // single-domain inner box: 1-based (2..n-1) -> padded 0-based (2+1 .. n-1+1) with g = 2
  g.kk0 = 3, g.kk1 = L.nk, g.jj0 = 3, g.jj1 = L.nj;
  const int ii0 = 3, ii1 = L.ni;
  g.F0 = (long long)ii0 * g.R, g.Fend = (long long)(ii1 + 1) * g.R;
  g.kk0a = g.kk0, g.kk1a = g.kk1, g.jj0a = g.jj0, g.jj1a = g.jj1, g.F0a = g.F0, g.Fenda = g.Fend;
  g.S = TB * MV - 2 * g.R;
  g.par = rb ? 1 : 0;
  g.zero_u = 0;
  g.nseg = (int)((g.Fend - g.F0 + g.S - 1) / g.S);
  const int nplanes = g.jj1 - g.jj0 + 1;
  g.TJ = std::min(tj, nplanes);
  const int nchunk = (nplanes + g.TJ - 1) / g.TJ;
  g.band = 1;
  g.map = nullptr;
  L.nblk = 8 * ((g.nseg + 7) / 8) * nchunk;
  L.lds1 = (size_t)2 * ((g.S + 4 * g.R) + (g.S + 2 * g.R)) * sizeof(Vec<V>) + 18 * sizeof(double);
  L.lds2 = L.lds1;
  return L.lds1 <= 160 * 1024 && g.nseg >= 8;
}

template <int TB, int MV, int RB>
float run(Lab& L, int which, int reps, double* sums) {
  constexpr int V = VW;
  Fin2 fin;
  fin.dst = L.dst, fin.counter = L.counter, fin.single = RB;
  REAL* W = which == 1 ? L.W1 : L.W2;
  auto launch = [&]() {
    if (which == 1)
      hipLaunchKernelGGL((jacobi2_k<V, TB, MV, RB>), dim3(L.nblk), dim3(TB), L.lds1, 0, L.U, L.B, W, L.c, L.g, L.partials, nullptr, fin);
    else
      hipLaunchKernelGGL((jacobi2p_k<V, TB, MV, RB, 0>), dim3(L.nblk), dim3(TB), L.lds2, 0, L.U, L.B, W, L.c, L.g, L.partials, nullptr, fin, MafArgs(), BSrc());
  };
  if (which == 1)
    HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&jacobi2_k<V, TB, MV, RB>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  else
    HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&jacobi2p_k<V, TB, MV, RB, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  for (int i = 0; i < 3; i++) launch();
  HIP_CHECK(hipDeviceSynchronize());
  HIP_CHECK(hipGetLastError());
  hipEvent_t a, b;
  HIP_CHECK(hipEventCreate(&a));
  HIP_CHECK(hipEventCreate(&b));
  HIP_CHECK(hipEventRecord(a, 0));
  for (int i = 0; i < reps; i++) launch();
  HIP_CHECK(hipEventRecord(b, 0));
  HIP_CHECK(hipEventSynchronize(b));
  float ms = 0;
  HIP_CHECK(hipEventElapsedTime(&ms, a, b));
  HIP_CHECK(hipMemcpy(sums, L.dst, 2 * sizeof(double), hipMemcpyDeviceToHost));
  return ms / reps;
}

template <int TB, int MV>
int ab(Lab& L, int tj, int reps, int rb) {
  if (!geom<TB, MV>(L, tj, rb)) {
    printf("(%d,%d) tj %d: geometry not supported\n", TB, MV, tj);
    return 0;
  }
  const size_t n = (size_t)L.nip * L.njp * L.nkp;
  HIP_CHECK(hipMemset(L.W1, 0, n * sizeof(REAL)));
  HIP_CHECK(hipMemset(L.W2, 0, n * sizeof(REAL)));
  double s1[2], s2[2];
  const float t1 = rb ? run<TB, MV, 1>(L, 1, reps, s1) : run<TB, MV, 0>(L, 1, reps, s1);
  const float t2 = rb ? run<TB, MV, 1>(L, 2, reps, s2) : run<TB, MV, 0>(L, 2, reps, s2);
  unsigned long long* cnt;
  HIP_CHECK(hipMalloc(&cnt, 8));
  HIP_CHECK(hipMemset(cnt, 0, 8));
  hipLaunchKernelGGL(diff_k, dim3(2048), dim3(256), 0, 0, L.W1, L.W2, n, cnt);
  unsigned long long h = 0;
  HIP_CHECK(hipMemcpy(&h, cnt, 8, hipMemcpyDeviceToHost));
  const double pts = (double)(L.ni - 2) * (L.nj - 2) * (L.nk - 2);
  const double alg = 2.0 * pts * 3 * sizeof(REAL);
  printf("(%4d,%d) tj %3d rb %d nblk %5d | v1 %.4f ms %7.0f GB/s | v2 %.4f ms %7.0f GB/s %8.0f MLUPS | diff words %llu | sums rel %.2e %.2e\n", TB, MV,
         L.g.TJ, rb, L.nblk, t1, alg / t1 / 1e6, t2, alg / t2 / 1e6, 2.0 * pts / t2 / 1e3, h, fabs(s1[0] - s2[0]) / fabs(s1[0] + 1e-300),
         fabs(s1[1] - s2[1]) / fabs(s1[1] + 1e-300));
  fflush(stdout);
  return h == 0 ? 0 : 1;
}
}  // namespace

int main(int argc, char** argv) {
  // pair_lab N reps rb ni nj nk  TBxTJ [TBxTJ ...]     (ni = 0: cube of N)
  const int N = argc > 1 ? atoi(argv[1]) : 512;
  const int reps = argc > 2 ? atoi(argv[2]) : 30;
  const int rb = argc > 3 ? atoi(argv[3]) : 0;
  Lab L;
  const bool box = argc > 6 && atoi(argv[4]) > 0;
  L.ni = box ? atoi(argv[4]) : N, L.nj = box ? atoi(argv[5]) : N, L.nk = box ? atoi(argv[6]) : N;
  L.nip = L.ni + 4, L.njp = L.nj + 4, L.nkp = L.nk + 4;
  const size_t n = (size_t)L.nip * L.njp * L.nkp;
  HIP_CHECK(hipMalloc(&L.U, n * sizeof(REAL)));
  HIP_CHECK(hipMalloc(&L.B, n * sizeof(REAL)));
  HIP_CHECK(hipMalloc(&L.W1, n * sizeof(REAL)));
  HIP_CHECK(hipMalloc(&L.W2, n * sizeof(REAL)));
  HIP_CHECK(hipMalloc(&L.partials, 65536 * sizeof(double)));
  HIP_CHECK(hipMalloc(&L.dst, 16 * sizeof(double)));
  HIP_CHECK(hipMalloc(&L.counter, 64));
  HIP_CHECK(hipMemset(L.counter, 0, 64));
  hipLaunchKernelGGL(fill_k, dim3(4096), dim3(256), 0, 0, L.U, n, 12345u, (REAL)1.0);
  hipLaunchKernelGGL(fill_k, dim3(4096), dim3(256), 0, 0, L.B, n, 777u, (REAL)0.5);
  HIP_CHECK(hipDeviceSynchronize());
  L.c.c1 = (REAL)1.1, L.c.c2 = (REAL)0.9, L.c.c3 = (REAL)1.05, L.c.c4 = (REAL)0.95, L.c.c5 = (REAL)1.2, L.c.c6 = (REAL)0.8, L.c.dd = (REAL)6.3,
  L.c.omg = (REAL)0.8;
  int bad = 0;
  printf("grid %d x %d x %d %s rb %d\n", L.ni, L.nj, L.nk, sizeof(REAL) == 4 ? "f32" : "f64", rb);
  if (argc <= 7) {
    bad += ab<512, 2>(L, 16, reps, rb);
    bad += ab<1024, 2>(L, 16, reps, rb);
  }
  for (int a = 7; a < argc; a++) {
    int tb = 512, tj = 16;
    sscanf(argv[a], "%dx%d", &tb, &tj);
    bad += tb == 1024 ? ab<1024, 2>(L, tj, reps, rb) : ab<512, 2>(L, tj, reps, rb);
  }
  printf(bad ? "MISMATCH\n" : "all fields bit-identical\n");
  return bad;
}
