// tools/psor_lab.hip -- stand-alone A/B of the lexicographic point SOR sweep: psor_tile_k (a launch per tile hyperplane) against psor_col_k
// (the whole sweep in one launch, columns of workgroups handing their faces on through memory): bitwise comparison of the field, residual, ms.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-slp-vectorize -std=c++17 -Icubez_amd/csrc -Iinclude tools/psor_lab.hip -o tools/bin/psor_lab [-DCZ_REAL_IS_DOUBLE]
//   tools/bin/psor_lab ni nj nk reps [maf] [wg_per_cu]
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
#include "cz_k_linesor.h"
#include "cz_k_psor.h"

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
__global__ void sum_k(const double* p, int n, double* out) {
  double s = 0;
  for (int i = 0; i < n; i++) s += p[i];
  *out = s;
}
}  // namespace

int main(int argc, char** argv) {
  const int ni = argc > 1 ? atoi(argv[1]) : 128, nj = argc > 2 ? atoi(argv[2]) : ni, nk = argc > 3 ? atoi(argv[3]) : ni;
  const int reps = argc > 4 ? atoi(argv[4]) : 5, maf = argc > 5 ? atoi(argv[5]) : 0, per_cu = argc > 6 ? atoi(argv[6]) : 1;
  const int nip = ni + 4, njp = nj + 4, nkp = nk + 4;
  const size_t n = (size_t)nip * njp * nkp;
  REAL *P0, *B, *P1, *P2;
  HIP_CHECK(hipMalloc(&P0, n * sizeof(REAL)));
  HIP_CHECK(hipMalloc(&B, n * sizeof(REAL)));
  HIP_CHECK(hipMalloc(&P1, n * sizeof(REAL)));
  HIP_CHECK(hipMalloc(&P2, n * sizeof(REAL)));
  hipLaunchKernelGGL(fill_k, dim3(4096), dim3(256), 0, 0, P0, n, 12345u, (REAL)1.0);
  hipLaunchKernelGGL(fill_k, dim3(4096), dim3(256), 0, 0, B, n, 777u, (REAL)0.5);
  Coef c;
  c.c1 = (REAL)1.1, c.c2 = (REAL)0.9, c.c3 = (REAL)1.05, c.c4 = (REAL)0.95, c.c5 = (REAL)1.2, c.c6 = (REAL)0.8, c.dd = (REAL)6.3, c.omg = (REAL)1.2;
  MafArgs ma = MafArgs();
  if (maf) {  // a stretched grid
    std::vector<REAL> xc(nip), yc(njp), zc(nkp);
    double x = 0;
    for (int i = 0; i < nip; i++) xc[i] = (REAL)(x += 0.5 + 0.37 * ((i * 7919) % 13) / 13.0);
    x = 0;
    for (int i = 0; i < njp; i++) yc[i] = (REAL)(x += 0.6 + 0.29 * ((i * 104729) % 11) / 11.0);
    x = 0;
    for (int i = 0; i < nkp; i++) zc[i] = (REAL)(x += 0.7 + 0.21 * ((i * 1299709) % 7) / 7.0);
    REAL *dx, *dy, *dz;
    HIP_CHECK(hipMalloc(&dx, nip * sizeof(REAL)));
    HIP_CHECK(hipMalloc(&dy, njp * sizeof(REAL)));
    HIP_CHECK(hipMalloc(&dz, nkp * sizeof(REAL)));
    HIP_CHECK(hipMemcpy(dx, xc.data(), nip * sizeof(REAL), hipMemcpyHostToDevice));
    HIP_CHECK(hipMemcpy(dy, yc.data(), njp * sizeof(REAL), hipMemcpyHostToDevice));
    HIP_CHECK(hipMemcpy(dz, zc.data(), nkp * sizeof(REAL), hipMemcpyHostToDevice));
    ma.xc = dx, ma.yc = dy, ma.zc = dz;
  }
  // single-domain inner box: 1-based 2 .. n-1 -> padded 3 .. n
  const int kk0 = 3, kk1 = nk, ii0 = 3, ii1 = ni, jj0 = 3, jj1 = nj;
  // ---- A: tile hyperplanes
  constexpr int T = 16;
  PsorGeom g;
  g.nkp = nkp, g.nip = nip, g.njp = njp, g.kk0 = kk0, g.kk1 = kk1, g.ii0 = ii0, g.ii1 = ii1, g.jj0 = jj0, g.jj1 = jj1;
  g.ntk = (kk1 - kk0 + T) / T, g.nti = (ii1 - ii0 + T) / T, g.ntj = (jj1 - jj0 + T) / T;
  const size_t ntiles = (size_t)g.ntk * g.nti * g.ntj;
  double *partials, *dst;
  HIP_CHECK(hipMalloc(&partials, std::max<size_t>(ntiles, 65536) * sizeof(double)));
  HIP_CHECK(hipMalloc(&dst, 16 * sizeof(double)));
  const size_t lds = ((size_t)(T + 2) * (T + 2) * (T + 2) + (size_t)T * T * T) * sizeof(REAL);
  HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&psor_tile_k<T, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
  HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&psor_tile_k<T, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
  hipEvent_t e0, e1;
  HIP_CHECK(hipEventCreate(&e0));
  HIP_CHECK(hipEventCreate(&e1));
  float bestA = 1e30f, bestB = 1e30f;
  double resA = 0, resB = 0;
  for (int rep = 0; rep < reps; rep++) {
    HIP_CHECK(hipMemcpy(P1, P0, n * sizeof(REAL), hipMemcpyDeviceToDevice));
    HIP_CHECK(hipMemset(partials, 0, ntiles * sizeof(double)));
    HIP_CHECK(hipEventRecord(e0, 0));
    for (int H = 0; H <= g.ntk + g.nti + g.ntj - 3; H++) {
      if (maf) hipLaunchKernelGGL((psor_tile_k<T, 1>), dim3(g.nti, g.ntj), dim3(T * T), lds, 0, P1, B, c, g, H, partials, nullptr, ma);
      else hipLaunchKernelGGL((psor_tile_k<T, 0>), dim3(g.nti, g.ntj), dim3(T * T), lds, 0, P1, B, c, g, H, partials, nullptr, ma);
    }
    HIP_CHECK(hipEventRecord(e1, 0));
    HIP_CHECK(hipEventSynchronize(e1));
    float ms;
    HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
    bestA = std::min(bestA, ms);
    hipLaunchKernelGGL(sum_k, dim3(1), dim3(1), 0, 0, partials, (int)ntiles, dst);
    HIP_CHECK(hipMemcpy(&resA, dst, sizeof(double), hipMemcpyDeviceToHost));
  }
  // ---- B: columns
  PsorColGeom q;
  q.nkp = nkp, q.nip = nip, q.njp = njp, q.kk0 = kk0, q.nk = kk1 - kk0 + 1, q.ii0 = ii0, q.ii1 = ii1, q.jj0 = jj0, q.jj1 = jj1;
  q.nti = (ii1 - ii0 + PC_T) / PC_T, q.ntj = (jj1 - jj0 + PC_T) / PC_T;
  q.face_words = (long long)(q.nk + PC_T) * PC_T * kPsorColHW;
  const int ncols = q.nti * q.ntj;
#ifndef PSOR_NC
#define PSOR_NC 1
#endif
  constexpr int NC = PSOR_NC;
  std::vector<int> order;
  for (int d = 0; d <= q.nti + q.ntj - 2; d++) {
    int nn = 0;
    for (int a = std::max(0, d - (q.ntj - 1)); a <= std::min(q.nti - 1, d); a++, nn++) order.push_back(a + q.nti * (d - a));
    while (nn % NC) order.push_back(-1), nn++;
  }
  const int ntickets = (int)(order.size() / NC);
  int* d_order;
  unsigned *ctl, *counter;
  unsigned long long* faces;
  HIP_CHECK(hipMalloc(&d_order, order.size() * sizeof(int)));
  HIP_CHECK(hipMemcpy(d_order, order.data(), order.size() * sizeof(int), hipMemcpyHostToDevice));
  HIP_CHECK(hipMalloc(&ctl, 256));
  HIP_CHECK(hipMalloc(&counter, 64));
  HIP_CHECK(hipMemset(counter, 0, 64));
  HIP_CHECK(hipMalloc(&faces, (size_t)2 * ncols * q.face_words * sizeof(unsigned long long)));
  HIP_CHECK(hipMemset(faces, 0, (size_t)2 * ncols * q.face_words * sizeof(unsigned long long)));
  hipDeviceProp_t prop;
  HIP_CHECK(hipGetDeviceProperties(&prop, 0));
  const int nblk = std::min(ntickets, prop.multiProcessorCount * per_cu);
  unsigned seq = 0;
  long long* prof = nullptr;
  const bool want_prof = getenv("PSOR_PROF") != nullptr;
  if (want_prof) {
    HIP_CHECK(hipMalloc(&prof, (size_t)4 * ncols * sizeof(long long)));
    HIP_CHECK(hipMemset(prof, 0, (size_t)4 * ncols * sizeof(long long)));
  }
  for (int rep = 0; rep < reps; rep++) {
    HIP_CHECK(hipMemcpy(P2, P0, n * sizeof(REAL), hipMemcpyDeviceToDevice));
    seq++;
    HIP_CHECK(hipEventRecord(e0, 0));
    HIP_CHECK(hipMemsetAsync(ctl, 0, 256, 0));
    if (maf) hipLaunchKernelGGL((psor_col_k<1, NC, 4>), dim3(nblk), dim3(psor_col_threads(NC)), 0, 0, P2, B, c, q, d_order, ntickets, ctl, faces, seq, 200000000LL, partials, dst + 1, 0, counter, nullptr, ma, prof);
    else hipLaunchKernelGGL((psor_col_k<0, NC, (sizeof(REAL) == 4 ? 8 : 4)>), dim3(nblk), dim3(psor_col_threads(NC)), 0, 0, P2, B, c, q, d_order, ntickets, ctl, faces, seq, 200000000LL, partials, dst + 1, 0, counter, nullptr, ma, prof);
    HIP_CHECK(hipEventRecord(e1, 0));
    HIP_CHECK(hipEventSynchronize(e1));
    HIP_CHECK(hipGetLastError());
    float ms;
    HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
    bestB = std::min(bestB, ms);
    HIP_CHECK(hipMemcpy(&resB, dst + 1, sizeof(double), hipMemcpyDeviceToHost));
  }
  if (want_prof) {
    std::vector<long long> h((size_t)4 * ncols);
    HIP_CHECK(hipMemcpy(h.data(), prof, h.size() * sizeof(long long), hipMemcpyDeviceToHost));
    long long t0 = h[0];
    for (int cc = 0; cc < ncols; cc++) t0 = std::min(t0, h[4 * cc]);
    const int nsteps = q.nk + 30;
    // per diagonal: when its columns start / end, how long a column takes, how far behind its predecessor (a-1, b) a column starts
    printf("diag: columns | start us (min..max) | duration us (mean) -> us per step | start behind (a-1,b) us (mean)\n");
    for (int d = 0; d <= q.nti + q.ntj - 2; d += std::max(1, (q.nti + q.ntj) / 16)) {
      double smin = 1e30, smax = 0, dur = 0, lag = 0;
      int cntc = 0, nl = 0;
      for (int a = std::max(0, d - (q.ntj - 1)); a <= std::min(q.nti - 1, d); a++) {
        const int b = d - a, cc = a + q.nti * b;
        const double st = (h[4 * cc] - t0) * 0.01, en = (h[4 * cc + 1] - t0) * 0.01;
        smin = std::min(smin, st), smax = std::max(smax, st), dur += en - st, cntc++;
        if (a > 0) lag += st - (h[4 * (cc - 1)] - t0) * 0.01, nl++;
      }
      double cyc = 0;
      for (int a = std::max(0, d - (q.ntj - 1)); a <= std::min(q.nti - 1, d); a++) cyc += (double)h[4 * (a + q.nti * (d - a)) + 3];
      printf("  %3d: %3d | %8.1f .. %8.1f | %7.1f -> %.3f | %6.1f | %.0f cycles per step = %.0f MHz\n", d, cntc, smin, smax, dur / cntc, dur / cntc / nsteps, nl ? lag / nl : 0.0,
             cyc / cntc / nsteps, cyc / dur);
    }
  }
  unsigned long long* cnt;
  HIP_CHECK(hipMalloc(&cnt, 8));
  HIP_CHECK(hipMemset(cnt, 0, 8));
  hipLaunchKernelGGL(diff_k, dim3(2048), dim3(256), 0, 0, P1, P2, n, cnt);
  unsigned long long h = 0;
  HIP_CHECK(hipMemcpy(&h, cnt, 8, hipMemcpyDeviceToHost));
  const double pts = (double)(ni - 2) * (nj - 2) * (nk - 2);
  printf("%d x %d x %d %s maf %d | tiles %.3f ms %8.0f MLUPS | columns (%d workgroups, %d columns) %.3f ms %8.0f MLUPS | differing words %llu | residual rel diff %.2e\n", ni, nj,
         nk, sizeof(REAL) == 4 ? "f32" : "f64", maf, bestA, pts / bestA / 1e3, nblk, ncols, bestB, pts / bestB / 1e3, h, fabs(resA - resB) / fabs(resA + 1e-300));
  return h == 0 ? 0 : 1;
}
