// cz_k_linesor.h -- part of cz_kernels.hip (ONE translation unit per precision; this file is included inside its anonymous
// namespace and is not a stand-alone header): line SOR by parallel cyclic reduction (pcr_*), lexicographic point SOR (psor), imask_k.
// ------------------------------------------------------------------------------------------------------------
// Line SOR by parallel cyclic reduction, pcr_rb (cz_solver.f90:497-662).
// One wave64 per (i,j) column of the active checkerboard colour, four columns per workgroup.  The column's tridiagonal
// system along k lives in LDS (a, c, d and their successors a1, c1, d1, ping-ponged instead of copied back); lanes
// stride over k, so every global access is coalesced along the unit-stride axis.  pn-1 reduction stages of stride
// 2^(p-1), then the 2x2 systems of stride 2^(pn-1), then the relaxation -- operation for operation the reference's
// arithmetic (serial build: entries outside kst..ked are zero, kept in two pad slots).  sum dp^2 is accumulated in double.
// ------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void wave_lds_sync() {
  // the lanes of ONE wave hand data to each other through LDS: order the accesses, no workgroup barrier
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

struct PcrGeom {
  int nkp, nip;                 // padded extents
  int kk0, n;                   // padded index of kst, number of unknowns per column
  int ii0, ni, jj0, nj;         // inner (i,j) range in padded indices / counts
  int ist1, jst1;               // 1-based ist, jst (colour rule mod(i+j,2) == color uses 1-based indices)
  int pn, color;
  int nhalf;                    // columns of one colour per j row (upper bound)
};

// ORDER 0: one colour; ORDER 1: the columns of one diagonal (i-ist)+(j-jst) = g.color of the lexicographic order; ORDER 2: all columns from
// the old field, results into WOUT (pcr_j_esa; see pcr_rb2_k).
// MAF = 1: the matrix comes from the metrics of the 1-D grids (cz_maf.f90:442-1560: pcr_rb_maf, pcr_maf and their _eda/_esa
// forms): it differs from line to line, so this literal form is the only one that applies.
// GS = 1: a, c, d and their successors live in a per-wave piece of GLOBAL scratch instead of LDS -- the form without a limit on the line
// length (the reference allocates its work arrays by kx, cz_Evaluate.cpp:257-262, and has none).  A line still never leaves its wave, and
// a wave's own stores are visible to its later loads (one CU, one L1; program order), so the wave-scope synchronisation is the same.  The
// workgroups are persistent there (`ncol` columns dealt round-robin), the scratch is the launch's workgroups x NW lines.
template <int NW, int ORDER, int MAF, int FINAL4 = 0, int GS = 0>
__global__ void __launch_bounds__(64 * NW)
pcr_rb_k(REAL* X, REAL* WOUT, const REAL* __restrict__ MSK, const REAL* __restrict__ RHS, PcrGeom g, REAL omg, double* partials,
         double* dst, int accumulate, unsigned* counter, MafArgs ma, REAL* scratch, long long ncol) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int n = g.n, LD = n + 2;  // slot 0 and slot n+1 are the zero entries k = kst-1 / ked+1
  REAL* base = GS ? scratch + ((size_t)blockIdx.x * NW + wave) * 6 * LD : reinterpret_cast<REAL*>(smem) + (size_t)wave * 6 * LD;
  REAL* A[2] = {base, base + 3 * LD};          // [buf][a | c | d]
  double* wsum = reinterpret_cast<double*>(reinterpret_cast<REAL*>(smem) + (GS ? 0 : (size_t)NW * 6 * LD) + 4);
  wsum = reinterpret_cast<double*>((reinterpret_cast<size_t>(wsum) + 15) & ~(size_t)15);
  double acc = 0.0;

  for (long long col = (long long)blockIdx.x * NW + wave; col < ncol; col += (long long)gridDim.x * NW) {  // (GS = 0: one column per wave)
  bool active;
  int ii = 0, jj = 0;
  if (ORDER == 0) {
    const int jrow = (int)(col / g.nhalf), ih = (int)(col % g.nhalf);
    active = jrow < g.nj;
    if (active) {
      const int j1 = g.jst1 + jrow;
      int i1 = g.ist1 + 2 * ih;
      if (((i1 + j1) & 1) != g.color) i1 += 1;   // first i of this colour in the row
      active = (i1 - g.ist1) < g.ni;
      ii = g.ii0 + (i1 - g.ist1);
      jj = g.jj0 + jrow;
    }
  } else if (ORDER == 1) {
    const int dlo = max(0, g.color - (g.nj - 1));
    active = col < (long long)(min(g.ni - 1, g.color) - dlo + 1);
    const int io = dlo + (int)col;
    ii = g.ii0 + io, jj = g.jj0 + (g.color - io);
  } else {
    active = col < (long long)g.ni * g.nj;
    ii = g.ii0 + (int)(col % g.ni), jj = g.jj0 + (int)(col / g.ni);
  }
  if (!active) ii = g.ii0, jj = g.jj0;
  const REAL r = (REAL)1.0 / (REAL)6.0;
  const size_t rowlen = (size_t)g.nkp, plane = (size_t)g.nkp * g.nip;
  const size_t c0 = (size_t)g.kk0 + (size_t)ii * rowlen + (size_t)jj * plane;  // element (kst, i, j)

  // ---- set-up: coefficients and source term (:545-568)
  if (active) {
    REAL* a = A[0];
    REAL* c = a + LD;
    REAL* d = c + LD;
    if (lane == 0) {
      for (int b = 0; b < 2; b++)
        for (int v = 0; v < 3; v++) A[b][v * LD] = (REAL)0, A[b][v * LD + n + 1] = (REAL)0;
    }
    if (MAF) {  // cz_maf.f90:489-546 (padded index == index into xc / yc / zc for g = 2, see MafArgs)
      const REAL GX = (REAL)2.0 / (ma.xc[ii + 1] - ma.xc[ii - 1]);
      const REAL EY = (REAL)2.0 / (ma.yc[jj + 1] - ma.yc[jj - 1]);
      const REAL C1 = GX * GX, C2 = EY * EY;
      const REAL C7 = -(ma.xc[ii + 1] - (REAL)2.0 * ma.xc[ii] + ma.xc[ii - 1]) * C1 * GX;
      const REAL C8 = -(ma.yc[jj + 1] - (REAL)2.0 * ma.yc[jj] + ma.yc[jj - 1]) * C2 * EY;
      const REAL dd1 = C1 + (REAL)0.5 * C7, dd2 = C1 - (REAL)0.5 * C7, cc1 = C2 + (REAL)0.5 * C8, cc2 = C2 - (REAL)0.5 * C8;
      for (int k = lane; k < n; k += 64) {
        const size_t e = c0 + k;
        const int kk = g.kk0 + k;
        const REAL f1 = ma.zc[kk + 1], f2 = ma.zc[kk - 1];
        const REAL TZ = (REAL)2.0 / (f1 - f2);
        const REAL ZTT = f1 - (REAL)2.0 * ma.zc[kk] + f2;
        const REAL f3 = TZ * TZ;
        const REAL aw = f3, cw = -ZTT * f3 * TZ, dw = (REAL)0.5 / (C1 + C2 + f3);
        a[k + 1] = (k == 0 && n > 1) ? (REAL)0 : -(aw - (REAL)0.5 * cw) * dw;  // (n = 1: :527 overwrites :513)
        c[k + 1] = (k == n - 1) ? (REAL)0 : -(aw + (REAL)0.5 * cw) * dw;
        const REAL mk = MSK[e];
        REAL dv = (dd1 * X[e + rowlen] + dd2 * X[e - rowlen] + cc1 * X[e + plane] + cc2 * X[e - plane] - RHS[e]) * dw * mk;
        if (k == 0) dv = (dv + (aw - (REAL)0.5 * cw) * dw * X[e - 1]) * mk;
        if (k == n - 1) dv = (dv + (aw + (REAL)0.5 * cw) * dw * X[e + 1]) * mk;
        d[k + 1] = dv;
      }
    } else {
      for (int k = lane; k < n; k += 64) {
        const size_t e = c0 + k;
        a[k + 1] = (k == 0) ? (REAL)0 : -r;
        c[k + 1] = (k == n - 1) ? (REAL)0 : -r;
        REAL dv = ((X[e - plane] + X[e + plane] + X[e - rowlen] + X[e + rowlen] - RHS[e]) * r) * MSK[e];
        if (k == 0) dv = (dv + X[e - 1] * r) * MSK[e];
        if (k == n - 1) dv = (dv + X[e + 1] * r) * MSK[e];
        d[k + 1] = dv;
      }
    }
  }
  wave_lds_sync();  // a line never leaves its wave: no workgroup barrier between the stages
  // ---- PCR stages (:572-595)
  int cur = 0;
  for (int p = 1; p <= (FINAL4 ? g.pn - 2 : g.pn - 1); p++) {
    const int s = 1 << (p - 1);
    if (active) {
      const REAL* a = A[cur];
      const REAL* c = a + LD;
      const REAL* d = c + LD;
      REAL* a1 = A[cur ^ 1];
      REAL* c1 = a1 + LD;
      REAL* d1 = c1 + LD;
      for (int k = lane; k < n; k += 64) {
        const int x = k + 1;
        const int kl = (k - s >= 0) ? x - s : 0;
        const int kr = (k + s <= n - 1) ? x + s : n + 1;
        const REAL ap = a[x], cp = c[x];
        const REAL e = (REAL)1.0 / ((REAL)1.0 - ap * c[kl] - cp * a[kr]);
        a1[x] = -e * ap * a[kl];
        c1[x] = -e * cp * c[kr];
        d1[x] = e * (d[x] - ap * d[kl] - cp * d[kr]);
      }
    }
    wave_lds_sync();
    cur ^= 1;
  }
  // ---- 4x4 systems of the last stage by Cramer's rule (pcr :787-842, pcr_esa, pcr_rb_esa; the coefficients pcr_coef_k tabulates are taken
  // from this line's own a and c: same operations, same values), result into the d slot of the other buffer
  if (FINAL4) {
    const int s = 1 << (g.pn - 2);
    if (active) {
      const REAL* a = A[cur];
      const REAL* c = a + LD;
      const REAL* d = c + LD;
      REAL* d1 = A[cur ^ 1] + 2 * LD;
      for (int k = lane; k < s && k < n; k += 64) {
        const int x = k + 1;
        const int kl = (k + s <= n - 1) ? x + s : n + 1, km = (k + 2 * s <= n - 1) ? x + 2 * s : n + 1, kr = (k + 3 * s <= n - 1) ? x + 3 * s : n + 1;
        const REAL cc1 = c[x], cc2 = c[kl], cc3 = c[km], aa2 = a[kl], aa3 = a[km], aa4 = a[kr];
        const REAL inv_detA = (REAL)1.0 / ((REAL)1.0 - aa4 * cc3 - aa3 * cc2 - aa2 * cc1 * ((REAL)1.0 - cc3 * aa4));
        const REAL dd1 = d[x], dd2 = d[kl], dd3 = d[km], dd4 = d[kr];
        const REAL detA1 = -cc3 * (aa4 * dd1 + cc1 * cc2 * dd4 - aa4 * cc1 * dd2) + dd1 + cc1 * cc2 * dd3 - aa3 * cc2 * dd1 - cc1 * dd2;
        const REAL detA2 = dd2 + cc2 * cc3 * dd4 - aa4 * cc3 * dd2 - cc2 * dd3 - aa2 * (dd1 - aa4 * cc3 * dd1);
        const REAL detA3 = dd3 - cc3 * dd4 - aa3 * dd2 - aa2 * (cc1 * dd3 - cc1 * cc3 * dd4 - aa3 * dd1);
        const REAL detA4 = dd4 + aa3 * aa4 * dd2 - aa4 * dd3 - aa3 * cc2 * dd4 - aa2 * (cc1 * dd4 + aa3 * aa4 * dd1 - aa4 * cc1 * dd3);
        d1[x] = detA1 * inv_detA;
        if (kl <= n) d1[kl] = detA2 * inv_detA;
        if (km <= n) d1[km] = detA3 * inv_detA;
        if (kr <= n) d1[kr] = detA4 * inv_detA;
      }
    }
    wave_lds_sync();
  } else {  // ---- 2x2 systems of the last stage (:599-616), result into the d slot of the other buffer
    const int s = 1 << (g.pn - 1);
    if (active) {
      const REAL* a = A[cur];
      const REAL* c = a + LD;
      const REAL* d = c + LD;
      REAL* d1 = A[cur ^ 1] + 2 * LD;
      for (int k = lane; k < s && k < n; k += 64) {
        const int x = k + 1;
        const int kr = (k + s <= n - 1) ? x + s : n + 1;
        const REAL cc1 = c[x], aa2 = a[kr], f1 = d[x], f2 = d[kr];
        const REAL jj2 = (REAL)1.0 / ((REAL)1.0 - aa2 * cc1);
        const REAL dd1 = (f1 - cc1 * f2) * jj2;
        const REAL dd2 = (f2 - aa2 * f1) * jj2;
        d1[x] = dd1;
        if (kr <= n) d1[kr] = dd2;  // (the reference also stores the k = ked+1 dummy, which nothing reads)
      }
    }
    wave_lds_sync();
  }
  // ---- relaxation (:626-633)
  if (active) {
    const REAL* d1 = A[cur ^ 1] + 2 * LD;
    for (int k = lane; k < n; k += 64) {
      const size_t e = c0 + k;
      const REAL pp = X[e];
      const REAL dp = (d1[k + 1] - pp) * omg * MSK[e];
      if (ORDER == 2) WOUT[e] = pp + dp;
      else X[e] = pp + dp;
      const REAL d2 = dp * dp;
      acc += (double)d2;
    }
  }
  wave_lds_sync();  // (the next column's set-up overwrites the buffers)
  }  // columns of this wave
  // ---- residual: partial per workgroup, fixed-order sum by the last one (write-through hand-off as in stencil_k)
  __syncthreads();
  const double sblk = block_sum<64 * NW>(acc, wsum);
  int* last_flag = reinterpret_cast<int*>(wsum + 8);
  const int nblk = gridDim.x;
  if (threadIdx.x == 0) {
    __hip_atomic_store(&partials[blockIdx.x], sblk, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    *last_flag = arrive_and_test_last(counter, nblk);
  }
  __syncthreads();
  if (*last_flag) {
    double x = 0.0;
    for (int i = threadIdx.x; i < nblk; i += 64 * NW) x += __hip_atomic_load(&partials[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    const double tot = block_sum<64 * NW>(x, wsum);
    if (threadIdx.x == 0) {
      dst[0] = accumulate ? dst[0] + tot : tot;
      *counter = 0u;
    }
  }
}

// ------------------------------------------------------------------------------------------------------------
// psor / psor_maf (cz_solver.f90:207-269, cz_maf.f90:23-112): lexicographic in-place SOR.  In the
// order (j outer, i, k inner) an update sees the NEW values of its k-1, i-1, j-1 neighbours and the OLD ones of k+1, i+1,
// j+1, so all points of a hyperplane k+i+j = const are independent: the sweep is a wavefront, and what one thread of the
// reference computes can be reproduced bit for bit in parallel.  Two levels: the box is cut into T^3 tiles, the tiles of
// one tile-hyperplane tk+ti+tj = H are independent (one launch per H, 3N/T - 2 launches per sweep); inside a tile, staged
// in LDS with one halo layer (new values from the tiles before, old values from the tiles after), thread (i,j) owns a
// column and updates k = h - i - j at step h (3T - 2 barrier-separated steps).
// ------------------------------------------------------------------------------------------------------------
struct PsorGeom {
  int nkp, nip, njp;
  int kk0, kk1, ii0, ii1, jj0, jj1;  // inner box, padded 0-based
  int ntk, nti, ntj;                 // tiles per axis
};

template <int T, int MAF>
__global__ void __launch_bounds__(T * T)
psor_tile_k(REAL* __restrict__ P, const REAL* __restrict__ B, Coef c, PsorGeom g, int H, double* __restrict__ tile_partials,
            const int* __restrict__ skip, MafArgs ma) {
  if (skip != nullptr && *skip != 0) return;
  const int ti = blockIdx.x, tj = blockIdx.y, tk = H - ti - tj;
  if (tk < 0 || tk >= g.ntk) return;  // uniform per workgroup
  constexpr int L1 = T + 2, L2 = (T + 2) * (T + 2);
  extern __shared__ __attribute__((aligned(16))) char smem[];
  __shared__ double wsum[T * T / 64 + 2];
  REAL* lp = reinterpret_cast<REAL*>(smem);  // p tile with one halo layer: [j][i][k]
  REAL* lb = lp + L2 * L1;                   // b tile
  const int t = threadIdx.x;
  const int K0 = g.kk0 + tk * T, I0 = g.ii0 + ti * T, J0 = g.jj0 + tj * T;  // first cell of the tile
  const size_t si = (size_t)g.nkp, sj = (size_t)g.nkp * g.nip;
  {  // every global read of the tile is issued before the first use (one memory latency per tile)
    constexpr int NP = (L2 * L1 + T * T - 1) / (T * T), NB = T;
    REAL rp[NP], rb[NB];
#pragma unroll
    for (int m = 0; m < NP; m++) {
      const int e = t + m * T * T;
      const int k = e % L1, r = e / L1, i = r % L1, j = r / L1;
      const int gk = K0 - 1 + k, gi = I0 - 1 + i, gj = J0 - 1 + j;
      rp[m] = (e < L2 * L1 && gk < g.nkp && gi < g.nip && gj < g.njp) ? P[(size_t)gk + (size_t)gi * si + (size_t)gj * sj] : (REAL)0;
    }
#pragma unroll
    for (int m = 0; m < NB; m++) {
      const int e = t + m * T * T;
      const int k = e % T, r = e / T, i = r % T, j = r / T;
      const int gk = K0 + k, gi = I0 + i, gj = J0 + j;
      rb[m] = (gk <= g.kk1 && gi <= g.ii1 && gj <= g.jj1) ? B[(size_t)gk + (size_t)gi * si + (size_t)gj * sj] : (REAL)0;
    }
#pragma unroll
    for (int m = 0; m < NP; m++)
      if (t + m * T * T < L2 * L1) lp[t + m * T * T] = rp[m];
#pragma unroll
    for (int m = 0; m < NB; m++) lb[t + m * T * T] = rb[m];
  }
  const int i = t % T, j = t / T;
  const int gi = I0 + i, gj = J0 + j;
  const bool col_in = gi <= g.ii1 && gj <= g.jj1;
  REAL XG = 0, XGG = 0, YE = 0, YEE = 0;
  if (MAF && col_in) {  // padded index == index into xc / yc / zc for g = 2 (see MafArgs)
    const REAL xm = ma.xc[gi - 1], x0 = ma.xc[gi], xp = ma.xc[gi + 1];
    const REAL ym = ma.yc[gj - 1], y0 = ma.yc[gj], yp = ma.yc[gj + 1];
    XG = (REAL)0.5 * (xp - xm), XGG = xp - (REAL)2.0 * x0 + xm;
    YE = (REAL)0.5 * (yp - ym), YEE = yp - (REAL)2.0 * y0 + ym;
  }
  __syncthreads();
  double acc = 0.0;
  for (int h = 0; h <= 3 * T - 3; h++) {
    const int k = h - i - j;
    if (k >= 0 && k < T && col_in && K0 + k <= g.kk1) {
      const int x = (k + 1) + L1 * (i + 1) + L2 * (j + 1);
      const REAL pp = lp[x];
      const REAL bb = lb[k + T * (i + T * j)];
      if (MAF) {
        const int gk = K0 + k;
        const REAL zm = ma.zc[gk - 1], z0 = ma.zc[gk], zp = ma.zc[gk + 1];
        const MafW w = maf_weights(XG, XGG, YE, YEE, (REAL)0.5 * (zp - zm), zp - (REAL)2.0 * z0 + zm);
        const REAL rp = w.w1 * lp[x + L1] + w.w2 * lp[x - L1] + w.w3 * lp[x + L2] + w.w4 * lp[x - L2] + w.w5 * lp[x + 1] +
                        w.w6 * lp[x - 1] + bb;  // cz_maf.f90:93-99
        const REAL dp = (rp / w.dd - pp) * c.omg;
        lp[x] = pp + dp;
        const REAL d2 = dp * dp;
        acc += (double)d2;
      } else {
        Vec<1> pc, im, ip, pm, pn, bv;
        pc.v[0] = pp, im.v[0] = lp[x - L1], ip.v[0] = lp[x + L1], pm.v[0] = lp[x - L2], pn.v[0] = lp[x + L2], bv.v[0] = bb;
        lp[x] = relax_vec<1>(pc, im, ip, pm, pn, lp[x - 1], lp[x + 1], bv, c, PlainDiv{c.dd}, 1u, 1u, acc).v[0];
      }
    }
    __syncthreads();
  }
  for (int e = t; e < T * T * T; e += T * T) {
    const int k = e % T, r = e / T, i2 = r % T, j2 = r / T;
    const int gk = K0 + k, gi2 = I0 + i2, gj2 = J0 + j2;
    if (gk <= g.kk1 && gi2 <= g.ii1 && gj2 <= g.jj1) P[(size_t)gk + (size_t)gi2 * si + (size_t)gj2 * sj] = lp[(k + 1) + L1 * (i2 + 1) + L2 * (j2 + 1)];
  }
  const double sblk = block_sum<T * T>(acc, wsum);
  if (t == 0) tile_partials[(size_t)tk + (size_t)g.ntk * (ti + (size_t)g.nti * tj)] = sblk;
}

// ------------------------------------------------------------------------------------------------------------
// pcr_rb, fast form.  The matrix of every k-line is the same (a = c = -1/6, zero at the ends; cz_solver.f90:545-556), so
// the a/c recurrences of the reduction and the reciprocals e = 1/(1 - ap*c(kl) - cp*a(kr)) (:572-595), and cc1/aa2/jj of
// the final 2x2 systems (:599-616), are identical for all lines: pcr_coef_k evaluates them ONCE, with the reference's
// operations in the reference's order, into a table; the per-line work that remains is the right-hand side
//     d1(k) = e * (d(k) - ap*d(kl) - cp*d(kr))
// -- the very expression of :590 with the very same operand values, hence the same bits -- without the division and the
// two coefficient updates (14 -> 5 flop per entry and stage).  pcr_rb2_k keeps the table in LDS, is persistent (the table
// is loaded once per workgroup), gives each wave L lines at a time (one table read serves L lines) and synchronises
// waves individually (a line never leaves its wave).
// ------------------------------------------------------------------------------------------------------------
// table layout: stage p = 1..nstage: [e | ap | cp] x n entries each, then the final stage x nfin entries:
//   final4 = 0 (pcr_rb, pcr_j_esa): nstage = pn-1, 2x2 systems (:599-616): [jj | cc1 | aa2]
//   final4 = 1 (pcr, pcr_esa, pcr_rb_esa): nstage = pn-2, 4x4 systems by Cramer's rule (:787-842): [inv_detA | cc1 | cc2 | cc3 | aa2 | aa3 | aa4]
__global__ void __launch_bounds__(256)
pcr_coef_k(REAL* __restrict__ tab, int n, int pn, int nfin, int final4) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int t = threadIdx.x, LD = n + 2;
  REAL* A[2] = {reinterpret_cast<REAL*>(smem), reinterpret_cast<REAL*>(smem) + 2 * LD};  // [buf][a | c]
  const REAL r = (REAL)1.0 / (REAL)6.0;
  for (int k = t; k < n; k += 256) {
    A[0][k + 1] = (k == 0) ? (REAL)0 : -r;
    A[0][LD + k + 1] = (k == n - 1) ? (REAL)0 : -r;
  }
  if (t == 0)
    for (int b = 0; b < 2; b++)
      for (int v = 0; v < 2; v++) A[b][v * LD] = (REAL)0, A[b][v * LD + n + 1] = (REAL)0;
  __syncthreads();
  int cur = 0;
  const int nstage = final4 ? pn - 2 : pn - 1;
  for (int p = 1; p <= nstage; p++) {
    const int s = 1 << (p - 1);
    const REAL* a = A[cur];
    const REAL* c = a + LD;
    REAL* a1 = A[cur ^ 1];
    REAL* c1 = a1 + LD;
    REAL* T = tab + (size_t)(p - 1) * 3 * n;
    for (int k = t; k < n; k += 256) {
      const int x = k + 1;
      const int kl = (k - s >= 0) ? x - s : 0;
      const int kr = (k + s <= n - 1) ? x + s : n + 1;
      const REAL ap = a[x], cp = c[x];
      const REAL e = (REAL)1.0 / ((REAL)1.0 - ap * c[kl] - cp * a[kr]);
      a1[x] = -e * ap * a[kl];
      c1[x] = -e * cp * c[kr];
      T[k] = e, T[n + k] = ap, T[2 * n + k] = cp;
    }
    __syncthreads();
    cur ^= 1;
  }
  const REAL* a = A[cur];
  const REAL* c = a + LD;
  REAL* F = tab + (size_t)nstage * 3 * n;
  if (!final4) {
    const int s = 1 << (pn - 1);
    for (int k = t; k < nfin; k += 256) {
      const int x = k + 1;
      const int kr = (k + s <= n - 1) ? x + s : n + 1;
      const REAL cc1 = c[x], aa2 = a[kr];
      F[k] = (REAL)1.0 / ((REAL)1.0 - aa2 * cc1);
      F[nfin + k] = cc1;
      F[2 * nfin + k] = aa2;
    }
  } else {
    const int s = 1 << (pn - 2);
    for (int k = t; k < nfin; k += 256) {
      const int x = k + 1;
      const int kl = (k + s <= n - 1) ? x + s : n + 1, km = (k + 2 * s <= n - 1) ? x + 2 * s : n + 1, kr = (k + 3 * s <= n - 1) ? x + 3 * s : n + 1;
      const REAL cc1 = c[x], cc2 = c[kl], cc3 = c[km], aa2 = a[kl], aa3 = a[km], aa4 = a[kr];
      F[k] = (REAL)1.0 / ((REAL)1.0 - aa4 * cc3 - aa3 * cc2 - aa2 * cc1 * ((REAL)1.0 - cc3 * aa4));
      F[nfin + k] = cc1, F[2 * nfin + k] = cc2, F[3 * nfin + k] = cc3;
      F[4 * nfin + k] = aa2, F[5 * nfin + k] = aa3, F[6 * nfin + k] = aa4;
    }
  }
}

// ORDER selects the columns of one launch (the line-SOR variants of cz_solver.f90 differ in the column order):
//   0  one checkerboard colour, in place          pcr_rb (:540), pcr_rb_esa (:1324)           g.color = colour
//   1  one diagonal (i-ist)+(j-jst) = g.color of the lexicographic order, in place: a column of pcr (:718-719) / pcr_esa sees
//      the new values of its i-1 and j-1 neighbours, which lie on the diagonal before => diagonals in sequence, columns of one in parallel
//   2  all columns from the old field, result into WOUT (pcr_j_esa :1553-1632; the caller copies back, :1655-1663)
//   TG = 1: the table stays in global memory (read through L1 / L2: it is the same for every line, a few hundred KB at most) and LDS holds the
//   right-hand sides only -- lines whose table does not fit LDS beside them (FP64 beyond ~640 unknowns, FP32 beyond ~1 290), up to the
//   ~10 000 / 20 000 unknowns whose two right-hand-side buffers fill the 160 KiB.  Same operations on the same values => same bits.
template <int NW, int L, int FINAL4, int ORDER, int TG = 0>
__global__ void __launch_bounds__(64 * NW)
pcr_rb2_k(REAL* X, REAL* WOUT, const REAL* __restrict__ MSK, const REAL* __restrict__ RHS, PcrGeom g, REAL omg,
          const REAL* __restrict__ tab, int tab_len, int nfin, double* partials, double* dst, int accumulate, unsigned* counter) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int n = g.n, LD = n + 2;  // slot 0 and slot n+1 are the zero entries k = kst-1 / ked+1
  REAL* Tl = reinterpret_cast<REAL*>(smem);
  const size_t tl = TG ? 0 : (size_t)tab_len;           // words of LDS in front of the right-hand sides
  const REAL* T = TG ? tab : Tl;                        // (TG is a compile-time constant: the address space of T is known per instantiation)
  REAL* D = Tl + tl + (size_t)wave * 2 * L * LD;        // [buf][line][LD]
  double* wsum = reinterpret_cast<double*>(Tl + tl + (size_t)NW * 2 * L * LD + 4);
  wsum = reinterpret_cast<double*>((reinterpret_cast<size_t>(wsum) + 15) & ~(size_t)15);

  if (!TG)
    for (int i = threadIdx.x; i < tab_len; i += 64 * NW) Tl[i] = tab[i];
  if (lane < 2 * L) D[lane * LD] = (REAL)0, D[lane * LD + n + 1] = (REAL)0;
  __syncthreads();

  const REAL r = (REAL)1.0 / (REAL)6.0;
  const size_t rowlen = (size_t)g.nkp, plane = (size_t)g.nkp * g.nip;
  const int dlo = (ORDER == 1) ? max(0, g.color - (g.nj - 1)) : 0;  // first i offset on the diagonal
  const long long ncol = (ORDER == 0) ? (long long)g.nhalf * g.nj
                         : (ORDER == 1) ? (long long)(min(g.ni - 1, g.color) - dlo + 1)
                                        : (long long)g.ni * g.nj;
  const long long ngroups = (ncol + L - 1) / L;
  double acc = 0.0;
  for (long long q = (long long)blockIdx.x * NW + wave; q < ngroups; q += (long long)gridDim.x * NW) {
    size_t c0[L];
    bool act[L];
#pragma unroll
    for (int l = 0; l < L; l++) {
      const long long col = q * L + l;  // column ordinal among the launch's columns
      int ii = 0, jj = 0;
      if (ORDER == 0) {
        const int jrow = (int)(col / g.nhalf), ih = (int)(col % g.nhalf);
        act[l] = jrow < g.nj;
        if (act[l]) {
          const int j1 = g.jst1 + jrow;
          int i1 = g.ist1 + 2 * ih;
          if (((i1 + j1) & 1) != g.color) i1 += 1;  // first i of this colour in the row
          act[l] = (i1 - g.ist1) < g.ni;
          ii = g.ii0 + (i1 - g.ist1);
          jj = g.jj0 + jrow;
        }
      } else if (ORDER == 1) {
        act[l] = col < ncol;
        const int io = dlo + (int)col;
        ii = g.ii0 + io, jj = g.jj0 + (g.color - io);
      } else {
        act[l] = col < ncol;
        ii = g.ii0 + (int)(col % g.ni), jj = g.jj0 + (int)(col / g.ni);
      }
      if (!act[l]) ii = g.ii0, jj = g.jj0;
      c0[l] = (size_t)g.kk0 + (size_t)ii * rowlen + (size_t)jj * plane;  // element (kst, i, j)
    }
    // ---- source term (:558-568)
#pragma unroll
    for (int l = 0; l < L; l++) {
      REAL* d = D + l * LD;
      for (int k = lane; k < n; k += 64) {
        REAL dv = (REAL)0;
        if (act[l]) {
          const size_t e = c0[l] + k;
          const REAL mk = MSK[e];
          dv = ((X[e - plane] + X[e + plane] + X[e - rowlen] + X[e + rowlen] - RHS[e]) * r) * mk;
          if (k == 0) dv = (dv + X[e - 1] * r) * mk;
          if (k == n - 1) dv = (dv + X[e + 1] * r) * mk;
        }
        d[k + 1] = dv;
      }
    }
    wave_lds_sync();
    // ---- PCR stages (:572-595), right-hand side only
    int cur = 0;
    const int nstage = FINAL4 ? g.pn - 2 : g.pn - 1;
    for (int p = 1; p <= nstage; p++) {
      const int s = 1 << (p - 1);
      const REAL* Tp = T + (size_t)(p - 1) * 3 * n;
      const REAL* dc = D + (size_t)cur * L * LD;
      REAL* dn = D + (size_t)(cur ^ 1) * L * LD;
      for (int k = lane; k < n; k += 64) {
        const int x = k + 1;
        const int kl = (k - s >= 0) ? x - s : 0;
        const int kr = (k + s <= n - 1) ? x + s : n + 1;
        const REAL e = Tp[k], ap = Tp[n + k], cp = Tp[2 * n + k];
#pragma unroll
        for (int l = 0; l < L; l++) dn[l * LD + x] = e * (dc[l * LD + x] - ap * dc[l * LD + kl] - cp * dc[l * LD + kr]);
      }
      wave_lds_sync();
      cur ^= 1;
    }
    // ---- 4x4 systems of the last stage by Cramer's rule (:787-842; pcr, pcr_esa, pcr_rb_esa)
    if (FINAL4) {
      const int s = 1 << (g.pn - 2);
      const REAL* F = T + (size_t)nstage * 3 * n;
      const REAL* dc = D + (size_t)cur * L * LD;
      REAL* dn = D + (size_t)(cur ^ 1) * L * LD;
      for (int k = lane; k < nfin; k += 64) {
        const int x = k + 1;
        const int kl = (k + s <= n - 1) ? x + s : n + 1, km = (k + 2 * s <= n - 1) ? x + 2 * s : n + 1, kr = (k + 3 * s <= n - 1) ? x + 3 * s : n + 1;
        const REAL inv_detA = F[k], cc1 = F[nfin + k], cc2 = F[2 * nfin + k], cc3 = F[3 * nfin + k];
        const REAL aa2 = F[4 * nfin + k], aa3 = F[5 * nfin + k], aa4 = F[6 * nfin + k];
#pragma unroll
        for (int l = 0; l < L; l++) {
          const REAL dd1 = dc[l * LD + x], dd2 = dc[l * LD + kl], dd3 = dc[l * LD + km], dd4 = dc[l * LD + kr];
          const REAL detA1 = -cc3 * (aa4 * dd1 + cc1 * cc2 * dd4 - aa4 * cc1 * dd2) + dd1 + cc1 * cc2 * dd3 - aa3 * cc2 * dd1 - cc1 * dd2;
          const REAL detA2 = dd2 + cc2 * cc3 * dd4 - aa4 * cc3 * dd2 - cc2 * dd3 - aa2 * (dd1 - aa4 * cc3 * dd1);
          const REAL detA3 = dd3 - cc3 * dd4 - aa3 * dd2 - aa2 * (cc1 * dd3 - cc1 * cc3 * dd4 - aa3 * dd1);
          const REAL detA4 = dd4 + aa3 * aa4 * dd2 - aa4 * dd3 - aa3 * cc2 * dd4 - aa2 * (cc1 * dd4 + aa3 * aa4 * dd1 - aa4 * cc1 * dd3);
          dn[l * LD + x] = detA1 * inv_detA;
          if (kl <= n) dn[l * LD + kl] = detA2 * inv_detA;
          if (km <= n) dn[l * LD + km] = detA3 * inv_detA;
          if (kr <= n) dn[l * LD + kr] = detA4 * inv_detA;
        }
      }
      wave_lds_sync();
    } else {  // ---- 2x2 systems of the last stage (:599-616)
      const int s = 1 << (g.pn - 1);
      const REAL* F = T + (size_t)nstage * 3 * n;
      const REAL* dc = D + (size_t)cur * L * LD;
      REAL* dn = D + (size_t)(cur ^ 1) * L * LD;
      for (int k = lane; k < nfin; k += 64) {
        const int x = k + 1;
        const int kr = (k + s <= n - 1) ? x + s : n + 1;
        const REAL jj2 = F[k], cc1 = F[nfin + k], aa2 = F[2 * nfin + k];
#pragma unroll
        for (int l = 0; l < L; l++) {
          const REAL f1 = dc[l * LD + x], f2 = dc[l * LD + kr];
          const REAL dd1 = (f1 - cc1 * f2) * jj2;
          const REAL dd2 = (f2 - aa2 * f1) * jj2;
          dn[l * LD + x] = dd1;
          if (kr <= n) dn[l * LD + kr] = dd2;
        }
      }
      wave_lds_sync();
    }
    // ---- relaxation (:626-633)
    {
      const REAL* d1 = D + (size_t)(cur ^ 1) * L * LD;
#pragma unroll
      for (int l = 0; l < L; l++) {
        if (!act[l]) continue;
        for (int k = lane; k < n; k += 64) {
          const size_t e = c0[l] + k;
          const REAL pp = X[e];
          const REAL dp = (d1[l * LD + k + 1] - pp) * omg * MSK[e];
          if (ORDER == 2) WOUT[e] = pp + dp;
          else X[e] = pp + dp;
          const REAL d2 = dp * dp;
          acc += (double)d2;
        }
      }
    }
    wave_lds_sync();  // the next group's source term overwrites buffer 0
  }
  // ---- residual: partial per workgroup, fixed-order sum by the last one (write-through hand-off as in stencil_k)
  __syncthreads();
  const double sblk = block_sum<64 * NW>(acc, wsum);
  int* last_flag = reinterpret_cast<int*>(wsum + 16);
  const int nblk = gridDim.x;
  if (threadIdx.x == 0) {
    __hip_atomic_store(&partials[blockIdx.x], sblk, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    *last_flag = arrive_and_test_last(counter, nblk);
  }
  __syncthreads();
  if (*last_flag) {
    double x = 0.0;
    for (int i = threadIdx.x; i < nblk; i += 64 * NW) x += __hip_atomic_load(&partials[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    const double tot = block_sum<64 * NW>(x, wsum);
    if (threadIdx.x == 0) {
      dst[0] = accumulate ? dst[0] + tot : tot;
      *counter = 0u;
    }
  }
}

// ------------------------------------------------------------------------------------------------------------
// Line SOR, register form.  A wave owns a k-line (L lines at a time); lane `lane` holds the M consecutive entries
// k = lane*M .. lane*M + M-1 of the right-hand side d in registers.  A reduction stage needs d(k-s) and d(k+s): for s < M
// they are in the lane's own registers except at the edges of its block (one value from lane-1 / lane+1), for s >= M they are
// entry m of lane -/+ s/M -- a cross-lane shuffle, no LDS memory and no barrier at all.  LDS holds only the
// line-independent coefficient table (pcr_coef_k's values, re-ordered [m][lane] so that reads are conflict-free), loaded
// once per persistent workgroup.  Same operations on the same operand values as pcr_rb2_k => same bits.
// ------------------------------------------------------------------------------------------------------------
// table: stage p = 1..nstage: [e | ap | cp] x NE, then the final stage [7 x NE if FINAL4 else 3 x NE], NE = 64*M, entry of
// element k = lane*M + m at m*64 + lane; entries of k >= n are zero
__global__ void __launch_bounds__(256)
pcr_coef_perm_k(const REAL* __restrict__ nat, REAL* __restrict__ tab, int n, int pn, int nfin, int final4, int M) {
  const int NE = 64 * M;
  const int nstage = final4 ? pn - 2 : pn - 1;
  const int s = 1 << nstage;  // stride of the final stage
  for (int k = threadIdx.x; k < NE; k += 256) {
    const int x = (k % M) * 64 + k / M;
    for (int p = 0; p < nstage; p++)
      for (int v = 0; v < 3; v++) tab[(size_t)(p * 3 + v) * NE + x] = (k < n) ? nat[(size_t)p * 3 * n + (size_t)v * n + k] : (REAL)0;
    const REAL* F = nat + (size_t)nstage * 3 * n;
    REAL* G = tab + (size_t)nstage * 3 * NE;
    const int kb = k % s;  // base element of the 2x2 / 4x4 system this element belongs to
    const int nf = final4 ? 7 : 3;
    for (int v = 0; v < nf; v++) G[(size_t)v * NE + x] = (k < n && kb < nfin) ? F[(size_t)v * nfin + kb] : (REAL)0;
  }
}

// M consecutive elements from / to an address that is only element-aligned (a k-line starts at padded index g): the hardware
// takes multi-dword global accesses at dword alignment
#ifdef CZ_REAL_IS_DOUBLE
typedef double RunVec __attribute__((ext_vector_type(2), aligned(8)));
constexpr int kRunW = 2;
#else
typedef float RunVec __attribute__((ext_vector_type(4), aligned(4)));
constexpr int kRunW = 4;
#endif
template <int M>
__device__ __forceinline__ void load_run(const REAL* __restrict__ p, REAL (&o)[M]) {
  if (M % kRunW == 0) {
#pragma unroll
    for (int c = 0; c < M; c += kRunW) {
      const RunVec v = *reinterpret_cast<const RunVec*>(p + c);
#pragma unroll
      for (int w = 0; w < kRunW; w++) o[c + w] = v[w];
    }
  } else {
#pragma unroll
    for (int c = 0; c < M; c++) o[c] = p[c];
  }
}
template <int M>
__device__ __forceinline__ void store_run(REAL* __restrict__ p, const REAL (&o)[M], int nvalid) {
  if (M % kRunW == 0 && nvalid >= M) {
#pragma unroll
    for (int c = 0; c < M; c += kRunW) {
      RunVec v;
#pragma unroll
      for (int w = 0; w < kRunW; w++) v[w] = o[c + w];
      *reinterpret_cast<RunVec*>(p + c) = v;
    }
  } else {
#pragma unroll
    for (int c = 0; c < M; c++)
      if (c < nvalid) p[c] = o[c];
  }
}

template <typename T>
__device__ __forceinline__ T lane_up(T v, int q, int lane) {  // value of lane - q, zero below lane 0
  const T r = __shfl_up(v, (unsigned)q, 64);
  return (q < 64 && lane >= q) ? r : (T)0;
}
template <typename T>
__device__ __forceinline__ T lane_down(T v, int q, int lane) {  // value of lane + q, zero above lane 63
  const T r = __shfl_down(v, (unsigned)q, 64);
  return (q < 64 && lane + q < 64) ? r : (T)0;
}

// The reduction stages and the final systems of L lines held in registers (lane = entries k0 .. k0+M-1 of each line): d -> sol.
// (a function of its own so that other kernels can hold lines in registers the same way)
template <int M, int L, int FINAL4>
__device__ __forceinline__ void pcr_reg_solve(REAL (&d)[L][M], REAL (&sol)[L][M], const REAL* T, int nstage, int n, int lane) {
  constexpr int NE = 64 * M;
  const int k0 = lane * M;
  // ---- PCR stages (:572-595), right-hand side only; every index below is a compile-time constant
#pragma unroll
  for (int sidx = 0; sidx < 20; sidx++) {
    if ((1 << sidx) >= NE) break;  // compile time
    if (sidx < nstage) {
      const int s = 1 << sidx;
      const REAL* Tp = T + (size_t)sidx * 3 * NE;
      REAL nd[L][M];
#pragma unroll
      for (int m = 0; m < M; m++) {
        const REAL e = Tp[m * 64 + lane], ap = Tp[NE + m * 64 + lane], cp = Tp[2 * NE + m * 64 + lane];
#pragma unroll
        for (int l = 0; l < L; l++) {
          REAL dl, dr;
          if (s < M) {
            dl = (m - s >= 0) ? d[l][(m - s >= 0) ? m - s : 0] : lane_up(d[l][(m - s + M) % M], 1, lane);
            dr = (m + s < M) ? d[l][(m + s < M) ? m + s : 0] : lane_down(d[l][(m + s) % M], 1, lane);
          } else {
            dl = lane_up(d[l][m], s / M, lane);
            dr = lane_down(d[l][m], s / M, lane);
          }
          nd[l][m] = e * (d[l][m] - ap * dl - cp * dr);
        }
      }
#pragma unroll
      for (int l = 0; l < L; l++)
#pragma unroll
        for (int m = 0; m < M; m++) d[l][m] = (k0 + m < n) ? nd[l][m] : (REAL)0;
    }
  }
  // ---- final stage: every entry solves for itself
  {
    const int s = 1 << nstage;
    const int qf = s / M;  // s >= M always (s >= n/4 > 8M .. see launch_pcr_reg)
    const REAL* F = T + (size_t)nstage * 3 * NE;
#pragma unroll
    for (int m = 0; m < M; m++) {
      const int k = k0 + m;
      const int rr = k >> nstage;  // position of this entry in its 2x2 / 4x4 system (s = 2^nstage)
      const int x = m * 64 + lane;
#pragma unroll
      for (int l = 0; l < L; l++) {
        const REAL me = d[l][m];
        if (!FINAL4) {  // (:599-616)
          const REAL jj2 = F[x], cc1 = F[NE + x], aa2 = F[2 * NE + x];
          const REAL up = lane_up(me, qf, lane), dn = lane_down(me, qf, lane);
          const REAL f1 = rr == 0 ? me : up, f2 = rr == 0 ? dn : me;
          sol[l][m] = rr == 0 ? (f1 - cc1 * f2) * jj2 : (f2 - aa2 * f1) * jj2;
        } else {  // Cramer's rule (:787-842)
          const REAL inv_detA = F[x], cc1 = F[NE + x], cc2 = F[2 * NE + x], cc3 = F[3 * NE + x];
          const REAL aa2 = F[4 * NE + x], aa3 = F[5 * NE + x], aa4 = F[6 * NE + x];
          const REAL u1 = lane_up(me, qf, lane), u2 = lane_up(me, 2 * qf, lane), u3 = lane_up(me, 3 * qf, lane);
          const REAL w1 = lane_down(me, qf, lane), w2 = lane_down(me, 2 * qf, lane), w3 = lane_down(me, 3 * qf, lane);
          const REAL dd1 = rr == 0 ? me : rr == 1 ? u1 : rr == 2 ? u2 : u3;
          const REAL dd2 = rr == 0 ? w1 : rr == 1 ? me : rr == 2 ? u1 : u2;
          const REAL dd3 = rr == 0 ? w2 : rr == 1 ? w1 : rr == 2 ? me : u1;
          const REAL dd4 = rr == 0 ? w3 : rr == 1 ? w2 : rr == 2 ? w1 : me;
          REAL det;
          if (rr == 0) det = -cc3 * (aa4 * dd1 + cc1 * cc2 * dd4 - aa4 * cc1 * dd2) + dd1 + cc1 * cc2 * dd3 - aa3 * cc2 * dd1 - cc1 * dd2;
          else if (rr == 1) det = dd2 + cc2 * cc3 * dd4 - aa4 * cc3 * dd2 - cc2 * dd3 - aa2 * (dd1 - aa4 * cc3 * dd1);
          else if (rr == 2) det = dd3 - cc3 * dd4 - aa3 * dd2 - aa2 * (cc1 * dd3 - cc1 * cc3 * dd4 - aa3 * dd1);
          else det = dd4 + aa3 * aa4 * dd2 - aa4 * dd3 - aa3 * cc2 * dd4 - aa2 * (cc1 * dd4 + aa3 * aa4 * dd1 - aa4 * cc1 * dd3);
          sol[l][m] = det * inv_detA;
        }
      }
    }
  }
}

template <int M, int NW, int L, int FINAL4, int ORDER>
__global__ void __launch_bounds__(64 * NW)
pcr_line_reg_k(REAL* X, REAL* WOUT, const REAL* __restrict__ MSK, const REAL* __restrict__ RHS, PcrGeom g, REAL omg,
               const REAL* __restrict__ tab, int tab_len, double* partials, double* dst, int accumulate, unsigned* counter) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int n = g.n;
  REAL* T = reinterpret_cast<REAL*>(smem);
  double* wsum = reinterpret_cast<double*>(T + tab_len + 4);
  wsum = reinterpret_cast<double*>((reinterpret_cast<size_t>(wsum) + 15) & ~(size_t)15);
  for (int i = threadIdx.x; i < tab_len; i += 64 * NW) T[i] = tab[i];
  __syncthreads();

  const REAL r = (REAL)1.0 / (REAL)6.0;
  const size_t rowlen = (size_t)g.nkp, plane = (size_t)g.nkp * g.nip;
  const int dlo = (ORDER == 1) ? max(0, g.color - (g.nj - 1)) : 0;
  const long long ncol = (ORDER == 0) ? (long long)g.nhalf * g.nj
                         : (ORDER == 1) ? (long long)(min(g.ni - 1, g.color) - dlo + 1)
                                        : (long long)g.ni * g.nj;
  const long long ngroups = (ncol + L - 1) / L;
  const int nstage = FINAL4 ? g.pn - 2 : g.pn - 1;
  const int k0 = lane * M;
  double acc = 0.0;
  for (long long q = (long long)blockIdx.x * NW + wave; q < ngroups; q += (long long)gridDim.x * NW) {
    size_t c0[L];
    bool act[L];
#pragma unroll
    for (int l = 0; l < L; l++) {
      const long long col = q * L + l;
      int ii = 0, jj = 0;
      if (ORDER == 0) {
        const int jrow = (int)(col / g.nhalf), ih = (int)(col % g.nhalf);
        act[l] = jrow < g.nj;
        if (act[l]) {
          const int j1 = g.jst1 + jrow;
          int i1 = g.ist1 + 2 * ih;
          if (((i1 + j1) & 1) != g.color) i1 += 1;
          act[l] = (i1 - g.ist1) < g.ni;
          ii = g.ii0 + (i1 - g.ist1);
          jj = g.jj0 + jrow;
        }
      } else if (ORDER == 1) {
        act[l] = col < ncol;
        const int io = dlo + (int)col;
        ii = g.ii0 + io, jj = g.jj0 + (g.color - io);
      } else {
        act[l] = col < ncol;
        ii = g.ii0 + (int)(col % g.ni), jj = g.jj0 + (int)(col / g.ni);
      }
      if (!act[l]) ii = g.ii0, jj = g.jj0;
      c0[l] = (size_t)g.kk0 + (size_t)ii * rowlen + (size_t)jj * plane;
    }
    // ---- source term (:558-568)
    // (a run may reach past the end of its line: those values are read from the rows behind it -- the array continues for at
    // least one more plane -- and discarded)
    REAL d[L][M];
#pragma unroll
    for (int l = 0; l < L; l++) {
      if (act[l] && k0 < n) {
        const size_t e0 = c0[l] + k0;
        REAL xjm[M], xjp[M], xim[M], xip[M], rh[M], mk[M];
        load_run<M>(X + e0 - plane, xjm);
        load_run<M>(X + e0 + plane, xjp);
        load_run<M>(X + e0 - rowlen, xim);
        load_run<M>(X + e0 + rowlen, xip);
        load_run<M>(RHS + e0, rh);
        load_run<M>(MSK + e0, mk);
#pragma unroll
        for (int m = 0; m < M; m++) {
          const int k = k0 + m;
          REAL dv = ((xjm[m] + xjp[m] + xim[m] + xip[m] - rh[m]) * r) * mk[m];
          if (k == 0) dv = (dv + X[e0 - 1] * r) * mk[m];
          if (k == n - 1) dv = (dv + X[e0 + m + 1] * r) * mk[m];
          d[l][m] = (k < n) ? dv : (REAL)0;
        }
      } else {
#pragma unroll
        for (int m = 0; m < M; m++) d[l][m] = (REAL)0;
      }
    }
    // ---- PCR stages (:572-595) and the final systems, right-hand side only
    REAL sol[L][M];
    pcr_reg_solve<M, L, FINAL4>(d, sol, T, nstage, n, lane);
    {
      // ---- relaxation (:626-633)
#pragma unroll
      for (int l = 0; l < L; l++) {
        if (!act[l] || k0 >= n) continue;
        const size_t e0 = c0[l] + k0;
        REAL pp[M], mk[M], out[M];
        load_run<M>(X + e0, pp);
        load_run<M>(MSK + e0, mk);
#pragma unroll
        for (int m = 0; m < M; m++) {
          const REAL dp = (sol[l][m] - pp[m]) * omg * mk[m];
          out[m] = pp[m] + dp;
          const REAL d2 = dp * dp;
          if (k0 + m < n) acc += (double)d2;
        }
        store_run<M>((ORDER == 2 ? WOUT : X) + e0, out, n - k0);
      }
    }
  }
  // ---- residual: partial per workgroup, fixed-order sum by the last one (write-through hand-off as in stencil_k)
  __syncthreads();
  const double sblk = block_sum<64 * NW>(acc, wsum);
  int* last_flag = reinterpret_cast<int*>(wsum + 16);
  const int nblk = gridDim.x;
  if (threadIdx.x == 0) {
    __hip_atomic_store(&partials[blockIdx.x], sblk, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    *last_flag = arrive_and_test_last(counter, nblk);
  }
  __syncthreads();
  if (*last_flag) {
    double x = 0.0;
    for (int i = threadIdx.x; i < nblk; i += 64 * NW) x += __hip_atomic_load(&partials[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    const double tot = block_sum<64 * NW>(x, wsum);
    if (threadIdx.x == 0) {
      dst[0] = accumulate ? dst[0] + tot : tot;
      *counter = 0u;
    }
  }
}

// ------------------------------------------------------------------------------------------------------------
// Register form of the MAF line solvers (cz_maf.f90:442-1560): the matrix differs from line to line (metrics of the 1-D grids), so
// a, c AND d of a line live in the registers of its wave (lane = M consecutive entries, as in pcr_line_reg_k) and all three are
// reduced: e = 1/(1 - ap*c(kl) - cp*a(kr)), a1 = -e*ap*a(kl), c1 = -e*cp*c(kr), d1 = e*(d - ap*d(kl) - cp*d(kr)) (:551-574), then
// the 2x2 systems (:578-596).  No table, no LDS memory, no barrier.  ORDER 0: one colour; ORDER 1: one diagonal of the lexicographic order.
// ------------------------------------------------------------------------------------------------------------
template <int M, int NW, int L, int ORDER>
__global__ void __launch_bounds__(64 * NW)
pcr_line_reg_maf_k(REAL* X, const REAL* __restrict__ MSK, const REAL* __restrict__ RHS, PcrGeom g, REAL omg, MafArgs ma, double* partials,
                   double* dst, int accumulate, unsigned* counter) {
  __shared__ double wsum[NW + 20];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int n = g.n;
  const size_t rowlen = (size_t)g.nkp, plane = (size_t)g.nkp * g.nip;
  const int dlo = (ORDER == 1) ? max(0, g.color - (g.nj - 1)) : 0;
  const long long ncol = (ORDER == 0) ? (long long)g.nhalf * g.nj : (long long)(min(g.ni - 1, g.color) - dlo + 1);
  const long long ngroups = (ncol + L - 1) / L;
  const int nstage = g.pn - 1;
  const int k0 = lane * M;
  double acc = 0.0;
  for (long long q = (long long)blockIdx.x * NW + wave; q < ngroups; q += (long long)gridDim.x * NW) {
    size_t c0[L];
    bool act[L];
    int iis[L], jjs[L];
#pragma unroll
    for (int l = 0; l < L; l++) {
      const long long col = q * L + l;
      int ii = g.ii0, jj = g.jj0;
      if (ORDER == 0) {
        const int jrow = (int)(col / g.nhalf), ih = (int)(col % g.nhalf);
        act[l] = jrow < g.nj;
        if (act[l]) {
          const int j1 = g.jst1 + jrow;
          int i1 = g.ist1 + 2 * ih;
          if (((i1 + j1) & 1) != g.color) i1 += 1;
          act[l] = (i1 - g.ist1) < g.ni;
          if (act[l]) ii = g.ii0 + (i1 - g.ist1), jj = g.jj0 + jrow;
        }
      } else {
        act[l] = col < ncol;
        if (act[l]) ii = g.ii0 + dlo + (int)col, jj = g.jj0 + (g.color - dlo - (int)col);
      }
      iis[l] = ii, jjs[l] = jj;
      c0[l] = (size_t)g.kk0 + (size_t)ii * rowlen + (size_t)jj * plane;
    }
    // ---- coefficients and source term (:489-546)
    REAL a[L][M], c[L][M], d[L][M];
#pragma unroll
    for (int l = 0; l < L; l++) {
      if (act[l] && k0 < n) {
        const int ii = iis[l], jj = jjs[l];
        const REAL GX = (REAL)2.0 / (ma.xc[ii + 1] - ma.xc[ii - 1]);
        const REAL EY = (REAL)2.0 / (ma.yc[jj + 1] - ma.yc[jj - 1]);
        const REAL C1 = GX * GX, C2 = EY * EY;
        const REAL C7 = -(ma.xc[ii + 1] - (REAL)2.0 * ma.xc[ii] + ma.xc[ii - 1]) * C1 * GX;
        const REAL C8 = -(ma.yc[jj + 1] - (REAL)2.0 * ma.yc[jj] + ma.yc[jj - 1]) * C2 * EY;
        const REAL dd1 = C1 + (REAL)0.5 * C7, dd2 = C1 - (REAL)0.5 * C7, cc1 = C2 + (REAL)0.5 * C8, cc2 = C2 - (REAL)0.5 * C8;
        const size_t e0 = c0[l] + k0;
        REAL xip[M], xim[M], xjp[M], xjm[M], rh[M], mk[M];
        load_run<M>(X + e0 + rowlen, xip);
        load_run<M>(X + e0 - rowlen, xim);
        load_run<M>(X + e0 + plane, xjp);
        load_run<M>(X + e0 - plane, xjm);
        load_run<M>(RHS + e0, rh);
        load_run<M>(MSK + e0, mk);
#pragma unroll
        for (int m = 0; m < M; m++) {
          const int k = k0 + m;
          const int kk = g.kk0 + (k < n ? k : 0);
          const REAL f1 = ma.zc[kk + 1], f2 = ma.zc[kk - 1];
          const REAL TZ = (REAL)2.0 / (f1 - f2);
          const REAL ZTT = f1 - (REAL)2.0 * ma.zc[kk] + f2;
          const REAL f3 = TZ * TZ;
          const REAL aw = f3, cw = -ZTT * f3 * TZ, dw = (REAL)0.5 / (C1 + C2 + f3);
          const REAL av = (k == 0 && n > 1) ? (REAL)0 : -(aw - (REAL)0.5 * cw) * dw;
          const REAL cv = (k == n - 1) ? (REAL)0 : -(aw + (REAL)0.5 * cw) * dw;
          REAL dv = (dd1 * xip[m] + dd2 * xim[m] + cc1 * xjp[m] + cc2 * xjm[m] - rh[m]) * dw * mk[m];
          if (k == 0) dv = (dv + (aw - (REAL)0.5 * cw) * dw * X[e0 - 1]) * mk[m];
          if (k == n - 1) dv = (dv + (aw + (REAL)0.5 * cw) * dw * X[e0 + m + 1]) * mk[m];
          a[l][m] = (k < n) ? av : (REAL)0, c[l][m] = (k < n) ? cv : (REAL)0, d[l][m] = (k < n) ? dv : (REAL)0;
        }
      } else {
#pragma unroll
        for (int m = 0; m < M; m++) a[l][m] = c[l][m] = d[l][m] = (REAL)0;
      }
    }
    // ---- PCR stages (:551-574)
#pragma unroll
    for (int sidx = 0; sidx < 20; sidx++) {
      if ((1 << sidx) >= 64 * M) break;  // compile time
      if (sidx < nstage) {
        const int s = 1 << sidx;
        REAL na[L][M], nc[L][M], nd[L][M];
#pragma unroll
        for (int m = 0; m < M; m++) {
#pragma unroll
          for (int l = 0; l < L; l++) {
            REAL al, cl, dl, ar, cr, dr;
            if (s < M) {
              if (m - s >= 0) {
                al = a[l][(m - s >= 0) ? m - s : 0], cl = c[l][(m - s >= 0) ? m - s : 0], dl = d[l][(m - s >= 0) ? m - s : 0];
              } else {
                al = lane_up(a[l][(m - s + M) % M], 1, lane), cl = lane_up(c[l][(m - s + M) % M], 1, lane), dl = lane_up(d[l][(m - s + M) % M], 1, lane);
              }
              if (m + s < M) {
                ar = a[l][(m + s < M) ? m + s : 0], cr = c[l][(m + s < M) ? m + s : 0], dr = d[l][(m + s < M) ? m + s : 0];
              } else {
                ar = lane_down(a[l][(m + s) % M], 1, lane), cr = lane_down(c[l][(m + s) % M], 1, lane), dr = lane_down(d[l][(m + s) % M], 1, lane);
              }
            } else {
              al = lane_up(a[l][m], s / M, lane), cl = lane_up(c[l][m], s / M, lane), dl = lane_up(d[l][m], s / M, lane);
              ar = lane_down(a[l][m], s / M, lane), cr = lane_down(c[l][m], s / M, lane), dr = lane_down(d[l][m], s / M, lane);
            }
            const REAL ap = a[l][m], cp = c[l][m];
            const REAL e = (REAL)1.0 / ((REAL)1.0 - ap * cl - cp * ar);
            na[l][m] = -e * ap * al;
            nc[l][m] = -e * cp * cr;
            nd[l][m] = e * (d[l][m] - ap * dl - cp * dr);
          }
        }
#pragma unroll
        for (int l = 0; l < L; l++)
#pragma unroll
          for (int m = 0; m < M; m++) {
            const bool in = k0 + m < n;
            a[l][m] = in ? na[l][m] : (REAL)0, c[l][m] = in ? nc[l][m] : (REAL)0, d[l][m] = in ? nd[l][m] : (REAL)0;
          }
      }
    }
    // ---- 2x2 systems (:578-596), every entry solves for itself, then the relaxation (:626-637)
    {
      const int qf = (1 << nstage) / M;
#pragma unroll
      for (int l = 0; l < L; l++) {
        REAL sol[M];
#pragma unroll
        for (int m = 0; m < M; m++) {
          const int rr = (k0 + m) >> nstage;
          const REAL a_dn = lane_down(a[l][m], qf, lane), d_dn = lane_down(d[l][m], qf, lane);
          const REAL c_up = lane_up(c[l][m], qf, lane), d_up = lane_up(d[l][m], qf, lane);
          const REAL cc1 = rr == 0 ? c[l][m] : c_up, aa2 = rr == 0 ? a_dn : a[l][m];
          const REAL f1 = rr == 0 ? d[l][m] : d_up, f2 = rr == 0 ? d_dn : d[l][m];
          const REAL jj2 = (REAL)1.0 / ((REAL)1.0 - aa2 * cc1);
          sol[m] = rr == 0 ? (f1 - cc1 * f2) * jj2 : (f2 - aa2 * f1) * jj2;
        }
        if (!act[l] || k0 >= n) continue;
        const size_t e0 = c0[l] + k0;
        REAL pp[M], mk[M], out[M];
        load_run<M>(X + e0, pp);
        load_run<M>(MSK + e0, mk);
#pragma unroll
        for (int m = 0; m < M; m++) {
          const REAL dp = (sol[m] - pp[m]) * omg * mk[m];
          out[m] = pp[m] + dp;
          const REAL d2 = dp * dp;
          if (k0 + m < n) acc += (double)d2;
        }
        store_run<M>(X + e0, out, n - k0);
      }
    }
  }
  // ---- residual (as in pcr_line_reg_k)
  const double sblk = block_sum<64 * NW>(acc, wsum);
  int* last_flag = reinterpret_cast<int*>(wsum + NW + 2);
  const int nblk = gridDim.x;
  if (threadIdx.x == 0) {
    __hip_atomic_store(&partials[blockIdx.x], sblk, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    *last_flag = arrive_and_test_last(counter, nblk);
  }
  __syncthreads();
  if (*last_flag) {
    double x = 0.0;
    for (int i = threadIdx.x; i < nblk; i += 64 * NW) x += __hip_atomic_load(&partials[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    const double tot = block_sum<64 * NW>(x, wsum);
    if (threadIdx.x == 0) {
      dst[0] = accumulate ? dst[0] + tot : tot;
      *counter = 0u;
    }
  }
}

// ------------------------------------------------------------------------------------------------------------
// pcr / pcr_esa (cz_solver.f90:666-878, :1036-1250): the lexicographic line SOR in ONE launch per sweep instead of one per diagonal.
// In the order (j outer, i inner) the line (i,j) sees the new values of (i-1,j) and (i,j-1) and the old ones of (i+1,j) and (i,j+1):
// the chain (i,j) -> (i+1,j), (i,j) -> (i,j+1) is (ni + nj) line solves long whatever the number of lines in flight, so the sweep is
// bounded by the LATENCY of one line solve plus the time to hand a line to the row below -- not by throughput.  Hence:
//   * a line is solved by NT >= n threads, one entry each: the right-hand side ping-pongs between two LDS rows (pcr_rb2_k's arithmetic,
//     one workgroup barrier per stage that orders LDS only -- the global loads of the next line stay in flight across it), the entry's
//     coefficients of every stage are read from an LDS copy of pcr_coef_k's table together with the three right-hand sides (one LDS
//     round trip per stage);
//   * a group of NT threads owns a row j and walks i = ist..ied, one line per step:
//       (i-1,j) new   the thread's own previous result, in a register
//       (i+1,j) old   the row's next line, loaded a step ahead (it is also the centre line `pp` of the next step)
//       (i,j+1) old   loaded a step ahead -- the owner of row j+1 cannot touch it before this row has published (i,j)
//       (i,j-1) new   from the row above: inside a workgroup (R groups, Q rows per thread: a strip of R*Q rows in lock step, row rs one
//                     line behind row rs-1) through LDS or the thread's own registers; between workgroups through memory, in the manner
//                     of RCCL's LL protocol: the last row of a strip writes every entry of its line as ONE 64-bit word {sequence number of
//                     the line | value bits} (FP64: two words, each with half of the bits) into a ring of `nslots` lines, with agent-scope
//                     relaxed atomic stores (`sc1`, write-through); the thread of the strip below that needs the entry reads the word with
//                     an agent-scope load -- asked for late in the step before -- and takes the value once the word carries the number it
//                     expects, reading again until it does.  A 64-bit store is single-copy atomic, so value and number arrive together:
//                     no drain of the stores, no separate flag, no barrier, one memory round trip.  Sequence numbers grow from sweep to
//                     sweep (`seq_base`), so a stale word never matches.  (The old line (i,j+1) that row j reads with an ordinary load is
//                     overwritten by row j+1 only after row j+1 has received the word of (i,j), and that word holds a value computed FROM
//                     the loaded entry -- the load has returned before the word is stored, so the later store cannot reach it.)  The strip below publishes the number of lines it has taken
//                     (`ctl`, one store per step); the strip above reads that a step ahead and does not overwrite a slot that is not free.
//                     Measured at 512^3 FP32 (profiles/r02/pcr_lex_*): a step takes 2.1 us (8 stages x 0.13 us + source term and fetches
//                     0.44 + final systems and relaxation 0.40 + rotation 0.2), a strip starts 2.6-2.8 us behind the one above; the sweep is
//                     nj x (lag + step').  More rows per workgroup make every step slower by more than the hand-offs they save.
// Progress: strips are handed out by a ticket (`ctl[0]`), so the strip a workgroup waits for was taken earlier by a workgroup that is
// running or has finished.  The ring adds one condition: a strip may run at most nslots lines ahead of the strip below, so the launcher
// starts no more workgroups than the chip holds at once and sizes nslots such that the first strip of the resident window can finish
// (and free its workgroup for the next strip) however far the window's last strip is held back.  Every wait is bounded (`spin_limit` ticks of the
// 100 MHz wall clock): on expiry `ctl[1]` is set, all workgroups leave, and the residual is NaN (a lost hand-off must not pass for a result).
// Same operations on the same operand values as the launch-per-diagonal path (pcr_line_reg_k<ORDER=1>) and the reference => same bits.
// ------------------------------------------------------------------------------------------------------------
constexpr int kPipeCtlStride = 32;  // one 128-byte line per strip counter; ctl[0] = next strip, ctl[1] = error, counters from ctl[kPipeCtlStride]

// one poll of a bounded wait: true when the wait must be given up (time is up, or another workgroup has given up)
#ifndef CZ_PIPE_SLEEP
#define CZ_PIPE_SLEEP 1  // (2, 4 and 8 measured in round 4: see profiles/r04/pcr_lex_what_bounds_it.txt)
#endif
__device__ __forceinline__ bool pipe_give_up(unsigned& polls, long long& t0, long long limit, unsigned* ctl) {
  __builtin_amdgcn_s_sleep(CZ_PIPE_SLEEP);
  if ((++polls & 255u) != 0) return false;
  if (__hip_atomic_load(&ctl[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return true;
  const long long now = (long long)wall_clock64();
  if (t0 == 0) {
    t0 = now;
    return false;
  }
  if (now - t0 <= limit) return false;
  __hip_atomic_store(&ctl[1], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return true;
}

__device__ __forceinline__ double block_sum_rt(double x, double* wsum, int nwaves) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off, 64);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (lane == 0) wsum[wave] = x;
  __syncthreads();
  double s = 0.0;
  if (threadIdx.x == 0)
    for (int w = 0; w < nwaves; w++) s += wsum[w];
  return s;  // valid on thread 0
}

// workgroup barrier that orders LDS accesses only: __syncthreads() would also wait for every global load and store in flight
// (vmcnt(0)), i.e. for the operands fetched a step ahead
__device__ __forceinline__ void lds_barrier() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// (FP64 keeps twice the registers per value: its workgroups are bounded at 512 threads so that the compiler may use 256 registers per thread --
// with 1 024 it spilled 104 bytes per thread and every reload drained the fetches in flight: 21 400 instead of 26 800 MLUPS)
// (the MAF form spilled 132 bytes in FP32 as well: same bound)
constexpr int lex_max_threads(int maf) { return (sizeof(REAL) == 8 || maf) ? 512 : 1024; }

// The per-strip time stamps (CZHIP_PCR_PIPE_PROF) cost eighteen registers: compiled in only with -DCZ_LEX_PROF (how the strip profiles under
// profiles/r02 were taken); without them the MAF form gained 15 % (27 100 -> 31 300 MLUPS).
#ifdef CZ_LEX_PROF
constexpr bool kLexProf = true;
#else
constexpr bool kLexProf = false;
#endif

template <int FINAL4, int NT, int Q, int MAF>
__global__ void __launch_bounds__(lex_max_threads(MAF))
pcr_lex_wg_k(REAL* X, const REAL* __restrict__ MSK, const REAL* __restrict__ RHS, PcrGeom g, REAL omg, const REAL* __restrict__ tab, int nfin,
             int R, unsigned* ctl, unsigned long long* hb, int nslots, unsigned seq_base, int nstrips, long long spin_limit, double* partials,
             double* dst, int accumulate, unsigned* counter, long long* prof, MafArgs ma) {
  // MAF = 1 (pcr_maf, pcr_eda_maf, pcr_esa_maf; cz_maf.f90:442-1560): the matrix of a line comes from the metrics of the 1-D grids and differs from
  // line to line, so a and c are reduced together with d (three LDS rows each way instead of one, no table; pcr_rb_k<MAF>'s arithmetic) and the
  // final systems are the 2x2 ones (FINAL4 = 0).
  // NT threads per line (one entry each), R groups of NT threads, Q rows per group (a thread holds the same entry of Q consecutive rows:
  // one barrier per stage serves Q line solves, the coefficients are read once for all of them, and row q+1 takes (i,j-1) from the
  // registers of row q).  A strip = R*Q consecutive rows, row rs one line behind row rs-1.
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int MAXST = 10;  // n <= 1024
  const int t = threadIdx.x;
  const int rg = __builtin_amdgcn_readfirstlane(t / NT);  // NT is a multiple of 64: a wave belongs to one group (uniform: row addresses stay scalar)
  const int k = t - rg * NT;
  const int n = g.n, LD = NT + 2, x = k + 1, RS = R * Q;
  // [3][RS][LD]: buffers 0 and 1 ping-pong through the stages, buffer 2 takes the source term -- the first stage reads it -- so that a thread
  // may start the next line while others still read the final systems of this one (no barrier at the end of a step when R == 1);
  // slot 0 and slot n+1 of every row are the zero entries k = kst-1 / ked+1
  REAL* D = reinterpret_cast<REAL*>(smem);
  REAL* AA = D + (size_t)3 * RS * LD;               // MAF: the sub- and super-diagonal of every line, laid out like D
  REAL* CC = AA + (MAF ? (size_t)3 * RS * LD : 0);
  REAL* NLINE = CC + (MAF ? (size_t)3 * RS * LD : 0);  // [R][NT]: the line the last row of each group has just finished, for the group below
  REAL* TAB = NLINE + (size_t)R * NT;               // [3*nstage + NF][NT]: this entry's e | ap | cp of every stage, then the final system's coefficients
  const int nstage = FINAL4 ? g.pn - 2 : g.pn - 1;
  const int ntab = MAF ? 0 : 3 * nstage + (FINAL4 ? 7 : 3);
  int* sh = reinterpret_cast<int*>(TAB + (size_t)ntab * NT);  // [0] strip, [2] a wait was given up
  double* wsum = reinterpret_cast<double*>((reinterpret_cast<size_t>(sh + 8) + 15) & ~(size_t)15);
  const int nwaves = (NT * R) >> 6;
  const bool kin = k < n;

  // ---- the coefficients (pcr_coef_k's table, natural order) into LDS, entry k of every row at [row][k]; zero where the reference has no entry
  const int sfin = 1 << nstage;
  const int kb = k & (sfin - 1), rr = k >> nstage;  // base entry and position in the final 2x2 / 4x4 system
  for (int row = rg; row < ntab; row += R) {  // (MAF: no table)
    REAL v = (REAL)0;
    if (row < 3 * nstage) {
      if (kin) v = tab[(size_t)row * n + k];
    } else if (kin && kb < nfin) {
      v = tab[(size_t)nstage * 3 * n + (size_t)(row - 3 * nstage) * nfin + kb];
    }
    TAB[(size_t)row * NT + k] = v;
  }
  const REAL* Tk = TAB + k;
  const int f1i = kb + 1;
  const int f2i = (kb + sfin <= n - 1) ? kb + 1 + sfin : n + 1;
  const int f3i = (kb + 2 * sfin <= n - 1) ? kb + 1 + 2 * sfin : n + 1;
  const int f4i = (kb + 3 * sfin <= n - 1) ? kb + 1 + 3 * sfin : n + 1;
  for (int e = t; e < (MAF ? 9 : 3) * RS * LD; e += NT * R) D[e] = (REAL)0;  // (AA and CC follow D)
  if (t == 0) sh[2] = 0;
  __syncthreads();
  REAL cfv[FINAL4 ? 7 : 3];  // this entry's coefficients of the final system: the same for every line, kept in registers
#pragma unroll
  for (int v = 0; v < (FINAL4 ? 7 : 3); v++) cfv[v] = MAF ? (REAL)0 : Tk[(size_t)(3 * nstage + v) * NT];
  // MAF: what depends on k only (cz_maf.f90:497-512; padded index == index into zc for g = 2, see MafArgs)
  REAL mz_f3 = 0, mz_lo = 0, mz_hi = 0;  // TZ^2, aw - cw/2, aw + cw/2
  if (MAF) {
    const int kk = g.kk0 + min(k, n - 1);
    const REAL f1 = ma.zc[kk + 1], f2 = ma.zc[kk - 1];
    const REAL TZ = (REAL)2.0 / (f1 - f2);
    const REAL ZTT = f1 - (REAL)2.0 * ma.zc[kk] + f2;
    mz_f3 = TZ * TZ;
    const REAL cw = -ZTT * mz_f3 * TZ;
    mz_lo = mz_f3 - (REAL)0.5 * cw, mz_hi = mz_f3 + (REAL)0.5 * cw;
  }

  const REAL r = (REAL)1.0 / (REAL)6.0;
  const size_t rowlen = (size_t)g.nkp, plane = (size_t)g.nkp * g.nip;
  const int kc = min(k, n - 1);
  const bool edge_lo = k == 0, edge_hi = k == n - 1;
  unsigned polls = 0;
  long long t0 = 0;

  for (;;) {
    __syncthreads();
    if (t == 0) {
      sh[0] = (int)__hip_atomic_fetch_add(&ctl[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (__hip_atomic_load(&ctl[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) sh[2] = 1;
    }
    __syncthreads();
    const int strip = __builtin_amdgcn_readfirstlane(sh[0]);
    if (strip >= nstrips || sh[2] != 0) break;
    const int rlast = min(RS, g.nj - strip * RS) - 1;                      // last row of this strip that exists
    const bool from_mem = strip > 0;                                       // row 0 of the strip reads (i,j-1) from the strip above
    const bool feeds = strip * RS + rlast + 1 < g.nj;                      // a strip below reads row rlast
    unsigned* my_prog = ctl + (size_t)kPipeCtlStride * (1 + strip);        // lines of the strip above this strip has taken
    unsigned* dn_prog = ctl + (size_t)kPipeCtlStride * (2 + strip);        // lines of this strip the strip below has taken
    // hand-off words of entry kc: {sequence number | value bits}; FP64: two words, each with half of the bits
    constexpr int HW = sizeof(REAL) == 8 ? 2 : 1;
    unsigned long long* hb_out = hb + ((size_t)strip * nslots * NT + kc) * HW;
    const unsigned long long* hb_in = hb + ((size_t)(strip > 0 ? strip - 1 : 0) * nslots * NT + kc) * HW;
    const int smask = nslots - 1;

    // Every load below is unconditional, from an address clamped into the array (entry kc, a row and a line that exist): a load in a
    // divergent branch makes the compiler wait for ALL loads in flight at the next use of any of them, and the operands fetched a step
    // ahead would be waited for at once.  What a clamped load returns for an entry, a row or a line that does not exist is never used.
    size_t c0[Q];  // element (kst, ist, j) of row q
    bool rowok[Q];
    REAL xim[Q], pp[Q], xip[Q], xjp[Q], rh[Q], mk[Q], klo[Q], khi[Q];
    REAL nxjm = (REAL)0;
#pragma unroll
    for (int q = 0; q < Q; q++) {
      const int rs = rg * Q + q;
      rowok[q] = rs <= rlast;
      c0[q] = (size_t)g.kk0 + (size_t)g.ii0 * rowlen + (size_t)(g.jj0 + strip * RS + min(rs, rlast)) * plane;
      const size_t e0 = c0[q] + kc;
      xim[q] = X[e0 - rowlen], pp[q] = X[e0], xip[q] = X[e0 + rowlen], xjp[q] = X[e0 + plane], rh[q] = RHS[e0], mk[q] = MSK[e0];
      klo[q] = X[c0[q] - 1], khi[q] = X[c0[q] + n];
    }
    unsigned long long hw[HW];  // row 0 below another strip: the hand-off words of the coming line, fetched a step ahead
#pragma unroll
    for (int w = 0; w < HW; w++) hw[w] = 0ull;
    REAL my_C2[Q], my_cc1[Q], my_cc2[Q];  // MAF: what depends on j only (:489-496)
#pragma unroll
    for (int q = 0; q < Q; q++) {
      my_C2[q] = my_cc1[q] = my_cc2[q] = (REAL)0;
      if (MAF) {
        const int jj = g.jj0 + strip * RS + min(rg * Q + q, rlast);
        const REAL EY = (REAL)2.0 / (ma.yc[jj + 1] - ma.yc[jj - 1]);
        const REAL C2 = EY * EY;
        const REAL C8 = -(ma.yc[jj + 1] - (REAL)2.0 * ma.yc[jj] + ma.yc[jj - 1]) * C2 * EY;
        my_C2[q] = C2, my_cc1[q] = C2 + (REAL)0.5 * C8, my_cc2[q] = C2 - (REAL)0.5 * C8;
      }
    }
    if (rg == 0) {
      if (from_mem) {
#pragma unroll
        for (int w = 0; w < HW; w++) hw[w] = __hip_atomic_load(hb_in + w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      } else {
        nxjm = X[c0[0] + kc - plane];  // the boundary plane
      }
    }
    int dn_seen = 0, dn_next = 0;  // lower bounds of dn_prog (the group that hands its row down)
    double acc = 0.0;
    long long pf_start = 0, pf_first = 0, pf_wait = 0, pf_nwait = 0;  // CZHIP_PCR_PIPE_PROF (thread 0)
    long long pf_ph[4] = {0, 0, 0, 0}, pf_m = 0;  // ticks up to the first barrier / in the stages / final + relax / rotation + last barrier
    if (kLexProf && prof && t == 0) pf_start = (long long)wall_clock64();
    const int nsteps = g.ni + rlast;

    for (int st = 0; st < nsteps; st++) {
      // ---- (i,j-1) of row 0 below another strip: the words fetched a step ahead are the line st once they carry its sequence number;
      // until then read them again (every thread for itself: no barrier, no separate flag)
      if (rg == 0 && from_mem && st < g.ni) {
        const unsigned want = seq_base + (unsigned)st + 1u;
        long long pa = 0;
        if (kLexProf && prof && t == 0) pa = (long long)wall_clock64();
        bool ok = true;
#pragma unroll
        for (int w = 0; w < HW; w++) ok = ok && (unsigned)(hw[w] >> 32) == want;
        while (!ok) {
          const unsigned long long* src = hb_in + (size_t)(st & smask) * NT * HW;
#pragma unroll
          for (int w = 0; w < HW; w++) hw[w] = __hip_atomic_load(src + w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          ok = true;
#pragma unroll
          for (int w = 0; w < HW; w++) ok = ok && (unsigned)(hw[w] >> 32) == want;
          if (!ok && pipe_give_up(polls, t0, spin_limit, ctl)) {
            sh[2] = 1;
            break;
          }
        }
        t0 = 0;
        if (kLexProf && prof && t == 0) {
          const long long pb = (long long)wall_clock64();
          pf_wait += pb - pa, pf_nwait += (pb - pa > 20);
          if (st == 0) pf_first = pb;
        }
      }
      if (kLexProf && prof && t == 0) pf_m = (long long)wall_clock64();
      bool act[Q], on[Q];
      size_t cl[Q], en[Q];
#pragma unroll
      for (int q = 0; q < Q; q++) {
        const int i = st - (rg * Q + q);
        act[q] = rowok[q] && i >= 0 && i < g.ni;
        on[q] = act[q] && kin;
        const int ic = min(max(i, 0), g.ni - 1);
        cl[q] = c0[q] + (size_t)ic * rowlen;                       // element (kst, ist+i, j)
        en[q] = c0[q] + (size_t)min(ic + 1, g.ni - 1) * rowlen;    // element (kst, ist+i+1, j)
      }
      // ---- source terms (:558-568)
#pragma unroll
      for (int q = 0; q < Q; q++) {
        REAL xjm;
        if (q > 0) {
          xjm = xim[q - 1];  // what row q-1 finished in the step before: its line i
        } else if (rg == 0) {
          if (from_mem) {
            if (sizeof(REAL) == 8) {
              const unsigned long long bits = (hw[0] & 0xffffffffull) | (hw[HW - 1] << 32);
              xjm = (REAL)__longlong_as_double((long long)bits);
            } else {
              xjm = (REAL)__uint_as_float((unsigned)(hw[0] & 0xffffffffull));
            }
          } else {
            xjm = nxjm;
          }
        } else {
          xjm = NLINE[(size_t)(rg - 1) * NT + kc];
        }
        if (!MAF) {
          REAL dv = ((xjm + xjp[q] + xim[q] + xip[q] - rh[q]) * r) * mk[q];
          if (edge_lo) dv = (dv + klo[q] * r) * mk[q];
          if (edge_hi) dv = (dv + khi[q] * r) * mk[q];
          if (on[q]) D[(size_t)(2 * RS + rg * Q + q) * LD + x] = dv;
        } else {  // cz_maf.f90:489-546: the metrics of this line's i, then coefficients and source term of entry k
          const int ii = g.ii0 + min(max(st - (rg * Q + q), 0), g.ni - 1);
          const REAL GX = (REAL)2.0 / (ma.xc[ii + 1] - ma.xc[ii - 1]);
          const REAL C1 = GX * GX;
          const REAL C7 = -(ma.xc[ii + 1] - (REAL)2.0 * ma.xc[ii] + ma.xc[ii - 1]) * C1 * GX;
          const REAL dd1 = C1 + (REAL)0.5 * C7, dd2 = C1 - (REAL)0.5 * C7;
          const REAL dw = (REAL)0.5 / (C1 + my_C2[q] + mz_f3);
          const REAL av = (edge_lo && n > 1) ? (REAL)0 : -mz_lo * dw;  // (n = 1: :527 overwrites :513)
          const REAL cv = edge_hi ? (REAL)0 : -mz_hi * dw;
          REAL dv = (dd1 * xip[q] + dd2 * xim[q] + my_cc1[q] * xjp[q] + my_cc2[q] * xjm - rh[q]) * dw * mk[q];
          if (edge_lo) dv = (dv + mz_lo * dw * klo[q]) * mk[q];
          if (edge_hi) dv = (dv + mz_hi * dw * khi[q]) * mk[q];
          if (on[q]) {
            const size_t o = (size_t)(2 * RS + rg * Q + q) * LD + x;
            AA[o] = av, CC[o] = cv, D[o] = dv;
          }
        }
      }
      // ---- operands of the next lines, most of a step ahead of their use: fetched behind the second stage (behind the drain of the
      // row that feeds the next strip), consumed at the end of the step
      // (nxjm, hw and dn_next are written in place -- their old values were used at the top of the step -- and never reset: a register
      // that is zeroed on one path and loaded on another makes the compiler wait for the loads in flight at the zeroing)
      REAL n_xip[Q], n_xjp[Q], n_rh[Q], n_mk[Q], n_klo[Q], n_khi[Q];
      auto fetch_next = [&]() {
#pragma unroll
        for (int q = 0; q < Q; q++) {
          n_xip[q] = X[en[q] + rowlen + kc], n_xjp[q] = X[en[q] + plane + kc], n_rh[q] = RHS[en[q] + kc], n_mk[q] = MSK[en[q] + kc];
          n_klo[q] = X[en[q] - 1], n_khi[q] = X[en[q] + n];
        }
        // (no branch here, whatever the group or the strip: a register that is loaded on one path and kept on another is merged by a
        // copy, and the copy waits for every load in flight.  Groups that need neither of the two read them all the same.)
        nxjm = X[en[0] + kc - plane];
        dn_next = (int)__hip_atomic_load(dn_prog, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // used at the end of the next step
      };
      fetch_next();
      lds_barrier();
      int gave_up = 0;
      if (kLexProf && prof && t == 0) {
        const long long nw = (long long)wall_clock64();
        pf_ph[0] += nw - pf_m, pf_m = nw;
      }
      // ---- PCR stages (:572-595), right-hand side only
#pragma unroll
      for (int sidx = 0; sidx < MAXST; sidx++) {
        if (sidx < nstage) {
          const int s = 1 << sidx;
          if (MAF) {  // a, c and d of every line reduced together (cz_maf.f90:551-574)
            const size_t oc = (size_t)((sidx == 0 ? 2 : (sidx & 1)) * RS + rg * Q) * LD, on_ = (size_t)(((sidx & 1) ^ 1) * RS + rg * Q) * LD;
            if (sidx == 0) gave_up = sh[2];  // (read with the operands of the first stage: nobody writes it between the two barriers around)
            const int kl = (k - s >= 0) ? x - s : 0;
            const int kr = (k + s <= n - 1) ? x + s : n + 1;
            REAL ap[Q], cp[Q], d0[Q], al[Q], cl[Q], dl[Q], ar[Q], cr[Q], dr[Q];
#pragma unroll
            for (int q = 0; q < Q; q++) {
              const size_t o = oc + (size_t)q * LD;
              ap[q] = AA[o + x], cp[q] = CC[o + x], d0[q] = D[o + x];
              al[q] = AA[o + kl], cl[q] = CC[o + kl], dl[q] = D[o + kl];
              ar[q] = AA[o + kr], cr[q] = CC[o + kr], dr[q] = D[o + kr];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q = 0; q < Q; q++) {
              const REAL e = (REAL)1.0 / ((REAL)1.0 - ap[q] * cl[q] - cp[q] * ar[q]);
              const REAL na = -e * ap[q] * al[q], nc = -e * cp[q] * cr[q], nd = e * (d0[q] - ap[q] * dl[q] - cp[q] * dr[q]);
              if (on[q]) {
                const size_t o = on_ + (size_t)q * LD + x;
                AA[o] = na, CC[o] = nc, D[o] = nd;
              }
            }
          } else {  // all LDS reads in flight at once (this entry's coefficients of the stage and three right-hand sides per row), one wait
            const REAL* dc = D + (size_t)((sidx == 0 ? 2 : (sidx & 1)) * RS + rg * Q) * LD;
            REAL* dn = D + (size_t)(((sidx & 1) ^ 1) * RS + rg * Q) * LD;
            if (sidx == 0) gave_up = sh[2];  // (read with the operands of the first stage: nobody writes it between the two barriers around)
            const int kl = (k - s >= 0) ? x - s : 0;
            const int kr = (k + s <= n - 1) ? x + s : n + 1;
            const REAL e = Tk[(3 * sidx) * NT], ap = Tk[(3 * sidx + 1) * NT], cp = Tk[(3 * sidx + 2) * NT];
            REAL d0[Q], dl[Q], dr[Q];
#pragma unroll
            for (int q = 0; q < Q; q++) d0[q] = dc[q * LD + x], dl[q] = dc[q * LD + kl], dr[q] = dc[q * LD + kr];
            __builtin_amdgcn_sched_barrier(0);  // (all reads issued before the first use: one LDS round trip per stage, not two)
#pragma unroll
            for (int q = 0; q < Q; q++) {
              const REAL nd = e * (d0[q] - ap * dl[q] - cp * dr[q]);
              if (on[q]) dn[q * LD + x] = nd;
            }
          }
          lds_barrier();
          if (sidx == 0 && gave_up) break;  // a wait was given up by a thread of this workgroup (every thread read the same value)
        }
      }
      if (gave_up) break;
      if (kLexProf && prof && t == 0) {
        const long long nw = (long long)wall_clock64();
        pf_ph[1] += nw - pf_m, pf_m = nw;
      }
      // ---- the hand-off words of the next line: asked for as late as the step allows (the strip above stores them at the end of ITS step),
      // checked at the top of the next step
      {
        const unsigned long long* src = hb_in + (size_t)((st + 1) & smask) * NT * HW;
#pragma unroll
        for (int w = 0; w < HW; w++) hw[w] = __hip_atomic_load(src + w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      // ---- final systems (:599-616 / Cramer's rule :787-842), every entry solves for itself; relaxation (:626-633)
      REAL out[Q];
      {
        const REAL* dc = D + (size_t)((nstage & 1) * RS + rg * Q) * LD;  // (pointer arithmetic, not a select of two pointers: the address space must stay LDS)
#pragma unroll
        for (int q = 0; q < Q; q++) {
          REAL sol;
          if (MAF) {  // 2x2 systems with the line's own coefficients (cz_maf.f90:578-596)
            const size_t o = (size_t)((nstage & 1) * RS + rg * Q + q) * LD;
            const REAL cc1 = CC[o + f1i], aa2 = AA[o + f2i], f1 = D[o + f1i], f2 = D[o + f2i];
            const REAL jj2 = (REAL)1.0 / ((REAL)1.0 - aa2 * cc1);
            sol = rr == 0 ? (f1 - cc1 * f2) * jj2 : (f2 - aa2 * f1) * jj2;
          } else if (!FINAL4) {
            const REAL jj2 = cfv[0], cc1 = cfv[1], aa2 = cfv[2];
            const REAL f1 = dc[q * LD + f1i], f2 = dc[q * LD + f2i];
            sol = rr == 0 ? (f1 - cc1 * f2) * jj2 : (f2 - aa2 * f1) * jj2;
          } else {
            const REAL inv_detA = cfv[0], cc1 = cfv[1], cc2 = cfv[2], cc3 = cfv[3], aa2 = cfv[4], aa3 = cfv[5], aa4 = cfv[6];
            const REAL dd1 = dc[q * LD + f1i], dd2 = dc[q * LD + f2i], dd3 = dc[q * LD + f3i], dd4 = dc[q * LD + f4i];
            __builtin_amdgcn_sched_barrier(0);
            REAL det;
            if (rr == 0) det = -cc3 * (aa4 * dd1 + cc1 * cc2 * dd4 - aa4 * cc1 * dd2) + dd1 + cc1 * cc2 * dd3 - aa3 * cc2 * dd1 - cc1 * dd2;
            else if (rr == 1) det = dd2 + cc2 * cc3 * dd4 - aa4 * cc3 * dd2 - cc2 * dd3 - aa2 * (dd1 - aa4 * cc3 * dd1);
            else if (rr == 2) det = dd3 - cc3 * dd4 - aa3 * dd2 - aa2 * (cc1 * dd3 - cc1 * cc3 * dd4 - aa3 * dd1);
            else det = dd4 + aa3 * aa4 * dd2 - aa4 * dd3 - aa3 * cc2 * dd4 - aa2 * (cc1 * dd4 + aa3 * aa4 * dd1 - aa4 * cc1 * dd3);
            sol = det * inv_detA;
          }
          const REAL dp = (sol - pp[q]) * omg * mk[q];
          out[q] = pp[q] + dp;
          const REAL d2 = dp * dp;
          if (on[q]) {
            acc += (double)d2;
            X[cl[q] + k] = out[q];
            if (q == Q - 1) NLINE[(size_t)rg * NT + k] = out[q];
            if (feeds && rg * Q + q == rlast) {
              // hand the line down: slot i mod nslots, free once the strip below has taken line i - nslots
              const int i = st - rlast;
              dn_seen = max(dn_seen, dn_next);  // (fetched during the step before)
              bool slot_free = true;
              while (dn_seen < i - nslots + 1) {
                dn_seen = (int)__hip_atomic_load(dn_prog, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (dn_seen < i - nslots + 1 && pipe_give_up(polls, t0, spin_limit, ctl)) {
                  sh[2] = 1;
                  slot_free = false;  // the wait was given up: the slot still holds a line the strip below has not taken -- leave it alone
                  break;
                }
              }
              t0 = 0;
              const unsigned long long tag = (unsigned long long)(seq_base + (unsigned)i + 1u) << 32;
              unsigned long long* dstw = hb_out + (size_t)(i & smask) * NT * HW;
              if (!slot_free) {
              } else if (sizeof(REAL) == 8) {
                const unsigned long long bits = (unsigned long long)__double_as_longlong((double)out[q]);
                __hip_atomic_store(dstw, tag | (bits & 0xffffffffull), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(dstw + (HW - 1), tag | (bits >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              } else {
                __hip_atomic_store(dstw, tag | (unsigned long long)__float_as_uint((float)out[q]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              }
            }
          }
        }
      }
      if (kLexProf && prof && t == 0) {
        const long long nw = (long long)wall_clock64();
        pf_ph[2] += nw - pf_m, pf_m = nw;
      }
#pragma unroll
      for (int q = 0; q < Q; q++)
        if (act[q]) xim[q] = out[q], pp[q] = xip[q], xip[q] = n_xip[q], xjp[q] = n_xjp[q], rh[q] = n_rh[q], mk[q] = n_mk[q], klo[q] = n_klo[q], khi[q] = n_khi[q];
      if (t == 0 && from_mem && st < g.ni) __hip_atomic_store(my_prog, (unsigned)(st + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // line st is taken
      if (R > 1) lds_barrier();  // (the group below reads NLINE at the top of the next step)
      if (kLexProf && prof && t == 0) pf_ph[3] += (long long)wall_clock64() - pf_m;
    }
    __syncthreads();
    const double sblk = block_sum_rt(acc, wsum, nwaves);
    if (t == 0) __hip_atomic_store(&partials[strip], sblk, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (kLexProf && prof && t == 0) {
      long long* q = prof + (size_t)8 * strip;
      q[0] = pf_start, q[1] = pf_first, q[2] = (long long)wall_clock64(), q[3] = pf_wait, q[4] = pf_nwait, q[5] = blockIdx.x, q[6] = (pf_ph[0] << 32) | pf_ph[1], q[7] = (pf_ph[2] << 32) | pf_ph[3];
    }
  }
  // ---- residual: the strips' partials in strip order, by the workgroup that arrives last (hand-off as in stencil_k)
  int* last_flag = sh + 4;
  const int nblk = gridDim.x;
  if (t == 0) *last_flag = arrive_and_test_last(counter, nblk);
  __syncthreads();
  if (*last_flag) {
    double xs = 0.0;
    for (int q = t; q < nstrips; q += NT * R) xs += __hip_atomic_load(&partials[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    const double tot = block_sum_rt(xs, wsum, nwaves);
    if (t == 0) {
      const bool bad = __hip_atomic_load(&ctl[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u;
      dst[0] = bad ? __builtin_nan("") : (accumulate ? dst[0] + tot : tot);
      *counter = 0u;
    }
  }
}

// imask_k (cz_blas.f90:24-104): 1 on the inner box, 0 elsewhere (whole padded array)
__global__ void __launch_bounds__(256)
imask_k(REAL* X, int nkp, int nip, int njp, int kk0, int kk1, int ii0, int ii1, int jj0, int jj1) {
  const size_t n = (size_t)nkp * nip * njp;
  for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (size_t)gridDim.x * 256) {
    const int kk = (int)(e % nkp);
    const size_t r = e / nkp;
    const int ii = (int)(r % nip), jj = (int)(r / nip);
    const bool in = kk >= kk0 && kk <= kk1 && ii >= ii0 && ii <= ii1 && jj >= jj0 && jj <= jj1;
    X[e] = in ? (REAL)1.0 : (REAL)0.0;
  }
}
