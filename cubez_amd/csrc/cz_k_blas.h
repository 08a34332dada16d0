// cz_k_blas.h -- part of cz_kernels.hip (ONE translation unit per precision; this file is included inside its anonymous
// namespace and is not a stand-alone header): reductions, convergence bookkeeping, BLAS-like element-wise kernels, dots, pivot, shell copy, boundary faces.
// sum of n partials in a fixed order -> dst[0] (= or +=).  One workgroup: deterministic.
__global__ void __launch_bounds__(1024)
reduce_partials_k(const double* __restrict__ partials, int n, double* __restrict__ dst, int accumulate,
                  const int* __restrict__ skip) {
  if (skip != nullptr && *skip != 0) return;
  __shared__ double wsum[16];
  double x = 0.0;
  for (int i = threadIdx.x; i < n; i += 1024) x += partials[i];
  const double s = block_sum<1024>(x, wsum);
  if (threadIdx.x == 0) dst[0] = accumulate ? dst[0] + s : s;
}

// cz_Poisson.cpp:67-77 on the device.  snap (optional) receives the flag as it stands after this test: the lagged loops of
// CZ::JACOBI / CZ::RBSOR hand that copy -- which no later test can change -- to the pass two passes further on, so that every
// workgroup of a pass sees the same value (the flag itself may be rewritten by the next test while the pass is running).
__global__ void check_k(const double* res_dev, double res_normal, double eps, int itr, double* hist, int* flag,
                        int* conv_itr, int* snap) {
  if (*flag == 0) {
    double r = res_dev[0];
    r *= res_normal;
    r = sqrt(r);
    hist[itr] = r;
    if (r < eps) {
      *flag = 1;
      *conv_itr = itr;
    }
  }
  if (snap) *snap = *flag;
}

// the same bookkeeping for a fused pair (iterations itr, itr+1) whose two sums were all-reduced first
__global__ void check2_k(const double* res_dev, double res_normal, double eps, int itr, double* hist, int* flag,
                         int* conv_itr, int* snap) {
  if (*flag == 0) {
    double r = sqrt(res_dev[0] * res_normal);
    hist[itr] = r;
    if (r < eps) {
      *flag = 1;
      *conv_itr = itr;
    } else {
      r = sqrt(res_dev[1] * res_normal);
      hist[itr + 1] = r;
      if (r < eps) {
        *flag = 1;
        *conv_itr = itr + 1;
      }
    }
  }
  if (snap) *snap = *flag;
}

// ------------------------------------------------------------------------------------------------------------
// element-wise kernels on the inner box (cz_blas.f90): one vector per thread, blockIdx.y = plane
// ------------------------------------------------------------------------------------------------------------
enum { OP_TRIAD = 0, OP_BICG1 = 1, OP_BICG2 = 2, OP_COPY = 3 };

struct EGeom {
  int R;
  long long PSV;
  int kk0, kk1, jj0;
  long long F0, Fend;
  // rows as R vectors from a vector boundary each; in memory a row is nkp elements and a plane PSE elements (Geom2, cz_k_pair.h)
  int nkp = 0;
  long long PSE = 0;
  // scalars of the update kept on the device (BiCGSTAB's alpha / omega, made by bicg_scal_k from the dot products of the launch before):
  // where set they replace the by-value arguments a / b of ewise_k and triad_dots_k -- the host need not read the dot products back first
  const REAL* pa = nullptr;
  const REAL* pb = nullptr;
};
// element offset of vector f of the row view inside a plane in memory
template <int V>
__device__ __forceinline__ long long egeom_eo(const EGeom& g, long long f) {
  const long long r = f / g.R;
  return r * g.nkp + (f - r * g.R) * V;
}

template <int V, int OP>
__global__ void __launch_bounds__(256)
ewise_k(REAL* Z, const REAL* X, const REAL* Y, REAL a, REAL b, EGeom g) {
  const long long f = g.F0 + (long long)blockIdx.x * 256 + threadIdx.x;
  if (f >= g.Fend) return;
  if (g.pa) a = *g.pa;
  if (g.pb) b = *g.pb;
  const long long pe = (long long)(g.jj0 + blockIdx.y) * g.PSE + egeom_eo<V>(g, f);
  const int kv = (int)(f % g.R);
  unsigned mk = 0;
#pragma unroll
  for (int cc = 0; cc < V; cc++) {
    const int kk = kv * V + cc;
    if (kk >= g.kk0 && kk <= g.kk1) mk |= 1u << cc;
  }
  if (mk == 0) return;
  Vec<V> x = ldve<V>(X, pe), y, z, o;
  if (OP != OP_COPY) y = ldve<V>(Y, pe);
  if (OP == OP_BICG1 || OP == OP_BICG2) z = ldve<V>(Z, pe);
#pragma unroll
  for (int cc = 0; cc < V; cc++) {
    if (OP == OP_TRIAD) o.v[cc] = a * x.v[cc] + y.v[cc];                              // cz_blas.f90:297
    if (OP == OP_BICG1) o.v[cc] = x.v[cc] + a * (z.v[cc] - b * y.v[cc]);              // :490  p = r + beta*(p - omg*q)
    if (OP == OP_BICG2) o.v[cc] = a * x.v[cc] + b * y.v[cc] + z.v[cc];                // :554
    if (OP == OP_COPY) o.v[cc] = x.v[cc];
  }
  if (mk == (1u << V) - 1) {
    stve<V>(Z, pe, o);
  } else {
#pragma unroll
    for (int cc = 0; cc < V; cc++)
      if (mk & (1u << cc)) Z[pe + cc] = o.v[cc];
  }
}

// BiCGSTAB's scalars on the device (cz_Poisson.cpp:427, 464), from the dot products the launch before left in `dots` (all-reduced in a
// decomposed run), with the host's own operations -- the double sums rounded to REAL, then one REAL division: the same bits as the host path.
//   STEP 1: alpha = rho / (q . r0)              sc[0] = alpha, sc[2] = -alpha
//   STEP 2: omega = (t . s) / (t . t)           sc[1] = omega, sc[3] = -omega
template <int STEP>
__global__ void bicg_scal_k(const double* __restrict__ dots, REAL rho, REAL* __restrict__ sc) {
  if (STEP == 1) {
    const REAL q_r0 = (REAL)dots[0];
    const REAL alpha = rho / q_r0;
    sc[0] = alpha, sc[2] = -alpha;
  } else {
    const REAL ts = (REAL)dots[0], tt = (REAL)dots[1];
    const REAL omega = ts / tt;
    sc[1] = omega, sc[3] = -omega;
  }
}

// search_pivot (cz_blas.f90:947-1039): pvt = 1 / max(|row entries|) on the inner box
template <int V>
__global__ void __launch_bounds__(256)
pivot_k(REAL* PVT, EGeom g, MafArgs ma, int nkp, int nip) {
  const long long f = g.F0 + (long long)blockIdx.x * 256 + threadIdx.x;
  if (f >= g.Fend) return;
  const int jj = g.jj0 + blockIdx.y;
  const long long row = f / g.R;
  const int kv = (int)(f - row * g.R);
  const long long pe = (long long)jj * g.PSE + row * g.nkp + (long long)kv * V;
  const int ii = (int)row;
  const REAL xm = ma.xc[ii - 1], x0 = ma.xc[ii], xp = ma.xc[ii + 1];
  const REAL ym = ma.yc[jj - 1], y0 = ma.yc[jj], yp = ma.yc[jj + 1];
  const REAL XG = (REAL)0.5 * (xp - xm), XGG = xp - (REAL)2.0 * x0 + xm;
  const REAL YE = (REAL)0.5 * (yp - ym), YEE = yp - (REAL)2.0 * y0 + ym;
#pragma unroll
  for (int cc = 0; cc < V; cc++) {
    const int kk = kv * V + cc;
    if (kk < g.kk0 || kk > g.kk1) continue;
    const REAL zm = ma.zc[kk - 1], z0 = ma.zc[kk], zp = ma.zc[kk + 1];
    const MafW w = maf_weights(XG, XGG, YE, YEE, (REAL)0.5 * (zp - zm), zp - (REAL)2.0 * z0 + zm);
    REAL ss = fmax(fabs(w.w1), fabs(w.w2));  // max(s1..s7), left to right (cz_blas.f90:1024)
    ss = fmax(ss, fabs(w.w3));
    ss = fmax(ss, fabs(w.w4));
    ss = fmax(ss, fabs(w.w5));
    ss = fmax(ss, fabs(w.w6));
    ss = fmax(ss, fabs(w.dd));
    PVT[pe + cc] = (REAL)1.0 / ss;
  }
  (void)nkp;
  (void)nip;
}

// dot products (cz_blas.f90:361-362, :426): per-point product in REAL, accumulated in double.  A workgroup strides over
// the planes (few thousand workgroups in all); the last one to finish sums the partials in fixed order into dst[0]
// (same write-through hand-off as the sweeps: no second launch).
template <int V, int TWO>
__global__ void __launch_bounds__(256)
dot_k(const REAL* X, const REAL* Y, EGeom g, int nplanes, double* partials, double* dst, unsigned* counter) {
  __shared__ double wsum[4];
  __shared__ int last_flag;
  const long long f = g.F0 + (long long)blockIdx.x * 256 + threadIdx.x;
  double acc = 0.0;
  if (f < g.Fend) {
    const int kv = (int)(f % g.R);
    unsigned mk = 0;
#pragma unroll
    for (int cc = 0; cc < V; cc++) {
      const int kk = kv * V + cc;
      if (kk >= g.kk0 && kk <= g.kk1) mk |= 1u << cc;
    }
    const long long eo = egeom_eo<V>(g, f);
    for (int pl = blockIdx.y; pl < nplanes; pl += gridDim.y) {
      const long long pe = (long long)(g.jj0 + pl) * g.PSE + eo;
      const Vec<V> x = ldve<V>(X, pe);
      Vec<V> y = x;
      if (TWO) y = ldve<V>(Y, pe);
#pragma unroll
      for (int cc = 0; cc < V; cc++) {
        const REAL tt = x.v[cc] * y.v[cc];
        if (mk & (1u << cc)) acc += (double)tt;
      }
    }
  }
  const double s = block_sum<256>(acc, wsum);
  const int nblk = gridDim.x * gridDim.y;
  const int me = blockIdx.y * gridDim.x + blockIdx.x;
  if (threadIdx.x == 0) {
    __hip_atomic_store(&partials[me], s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    last_flag = arrive_and_test_last(counter, nblk);
  }
  __syncthreads();
  if (last_flag) {
    double x = 0.0;
    for (int i = threadIdx.x; i < nblk; i += 256) x += __hip_atomic_load(&partials[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    const double tot = block_sum<256>(x, wsum);
    if (threadIdx.x == 0) {
      dst[0] = tot;
      *counter = 0u;
    }
  }
}

// z = a*x + y on the inner box (blas_triad, cz_blas.f90:297) with the two dot products that follow it in BiCGSTAB folded
// in: dst[0] = sum z*z (cz_Poisson.cpp:481), dst[1] = sum z*w (the next iteration's rho, :376).  Same structure as dot_k.
template <int V>
__global__ void __launch_bounds__(256)
triad_dots_k(REAL* Z, const REAL* X, const REAL* Y, const REAL* W, REAL a, EGeom g, int nplanes, double* partials, double* dst,
             unsigned* counter) {
  __shared__ double wsum[4];
  __shared__ int last_flag;
  const long long f = g.F0 + (long long)blockIdx.x * 256 + threadIdx.x;
  if (g.pa) a = *g.pa;
  double acc1 = 0.0, acc2 = 0.0;
  if (f < g.Fend) {
    const int kv = (int)(f % g.R);
    unsigned mk = 0;
#pragma unroll
    for (int cc = 0; cc < V; cc++) {
      const int kk = kv * V + cc;
      if (kk >= g.kk0 && kk <= g.kk1) mk |= 1u << cc;
    }
    if (mk != 0) {
      const long long eo = egeom_eo<V>(g, f);
      for (int pl = blockIdx.y; pl < nplanes; pl += gridDim.y) {
        const long long pe = (long long)(g.jj0 + pl) * g.PSE + eo;
        const Vec<V> x = ldve<V>(X, pe), y = ldve<V>(Y, pe), w = ldve<V>(W, pe);
        Vec<V> o;
#pragma unroll
        for (int cc = 0; cc < V; cc++) {
          o.v[cc] = a * x.v[cc] + y.v[cc];
          const REAL zz = o.v[cc] * o.v[cc];
          const REAL zw = o.v[cc] * w.v[cc];
          if (mk & (1u << cc)) {
            acc1 += (double)zz;
            acc2 += (double)zw;
          }
        }
        if (mk == (1u << V) - 1) {
          stve<V>(Z, pe, o);
        } else {
#pragma unroll
          for (int cc = 0; cc < V; cc++)
            if (mk & (1u << cc)) Z[pe + cc] = o.v[cc];
        }
      }
    }
  }
  const double s1 = block_sum<256>(acc1, wsum);
  __syncthreads();
  const double s2 = block_sum<256>(acc2, wsum);
  const int nblk = gridDim.x * gridDim.y;
  const int me = blockIdx.y * gridDim.x + blockIdx.x;
  if (threadIdx.x == 0) {
    __hip_atomic_store(&partials[me], s1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(&partials[nblk + me], s2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    last_flag = arrive_and_test_last(counter, nblk);
  }
  __syncthreads();
  if (last_flag) {
    double x1 = 0.0, x2 = 0.0;
    for (int i = threadIdx.x; i < nblk; i += 256) {
      x1 += __hip_atomic_load(&partials[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      x2 += __hip_atomic_load(&partials[nblk + i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    const double t1 = block_sum<256>(x1, wsum);
    __syncthreads();
    const double t2 = block_sum<256>(x2, wsum);
    if (threadIdx.x == 0) {
      dst[0] = t1;
      dst[1] = t2;
      *counter = 0u;
    }
  }
}

// ------------------------------------------------------------------------------------------------------------
// bc_k (cz_solver.f90:22-191): the sin*sin table is evaluated on the HOST with the host libm -- the same sinf/sin
// the reference's Fortran calls -- so the Dirichlet data are bit-identical to the reference's; the kernels only
// scatter it.  Three launches in the reference's order: K faces, then I faces, then J faces (edges end up 0).
// ------------------------------------------------------------------------------------------------------------
__global__ void bc_kface_k(REAL* p, const REAL* __restrict__ tab, int ix, int jx, int kface, int g, int nkp, int nip) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x + 1, j = blockIdx.y + 1;
  if (i > ix) return;
  p[(size_t)(kface + g - 1) + (size_t)(i + g - 1) * nkp + (size_t)(j + g - 1) * nkp * nip] = tab[(size_t)(j - 1) * ix + (i - 1)];
}
__global__ void bc_iface_k(REAL* p, int jx, int kx, int iface, int g, int nkp, int nip) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x + 1, j = blockIdx.y + 1;
  if (k > kx) return;
  p[(size_t)(k + g - 1) + (size_t)(iface + g - 1) * nkp + (size_t)(j + g - 1) * nkp * nip] = (REAL)0;
}
__global__ void bc_jface_k(REAL* p, int ix, int kx, int jface, int g, int nkp, int nip) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x + 1, i = blockIdx.y + 1;
  if (k > kx) return;
  p[(size_t)(k + g - 1) + (size_t)(i + g - 1) * nkp + (size_t)(jface + g - 1) * nkp * nip] = (REAL)0;
}

// copy every element OUTSIDE the inner box (guide cells, Dirichlet faces) from src to dst: one wave per k-row.
// Used to give the ping-pong partner buffer of a Jacobi solve the same boundary data as the solution array.
__global__ void __launch_bounds__(256)
copy_shell_k(REAL* __restrict__ dst, const REAL* __restrict__ src, int nkp, int nip, int njp, int kk0, int kk1, int ii0, int ii1,
             int jj0, int jj1) {
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= (long long)nip * njp) return;
  const int lane = threadIdx.x & 63;
  const int jj = (int)(row / nip), ii = (int)(row - (long long)jj * nip);
  const bool inner_row = ii >= ii0 && ii <= ii1 && jj >= jj0 && jj <= jj1;
  const size_t base = (size_t)row * nkp;
  if (inner_row) {
    for (int kk = lane; kk < kk0; kk += 64) dst[base + kk] = src[base + kk];
    for (int kk = kk1 + 1 + lane; kk < nkp; kk += 64) dst[base + kk] = src[base + kk];
  } else {
    for (int kk = lane; kk < nkp; kk += 64) dst[base + kk] = src[base + kk];
  }
}
