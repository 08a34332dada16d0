// cz_k_stencil.h -- part of cz_kernels.hip (ONE translation unit per precision; this file is included inside its anonymous
// namespace and is not a stand-alone header): stencil_k: one sweep (Jacobi / one RB colour / SpMV / residual).
// ------------------------------------------------------------------------------------------------------------
// The 7-point sweep.  MODE selects the point update:
//   JACOBI  cz_solver.f90:334-351   out = p + ((ss-b)/dd - p)*omg , acc += dp*dp
//   RB      cz_solver.f90:466-480   same, in place (OUT == P), only points of one colour
//   AX      cz_blas.f90:626-632     out = ss - dd*p
//   RK      cz_blas.f90:705-711     out = b - (ss - dd*p)
// with ss = c1*p(i+1) + c2*p(i-1) + c3*p(j+1) + c4*p(j-1) + c5*p(k+1) + c6*p(k-1), left to right.
// Elements outside the inner box are never written.
// ------------------------------------------------------------------------------------------------------------
// MAF = 1: the weights come from maf_weights() instead of c (cz_maf.f90:131-438, cz_blas.f90:738-934):
//   JACOBI/RB  dp = ((sum w*p_nb + b)/dd - p)*omg      AX  out = (sum w*p_nb - dd*p)*pvt      RK  out = (b + dd*p - sum w*p_nb)*pvt
template <int V, int TB, int M, int PF, int MODE, int MAF>
__global__ void __launch_bounds__(TB)
stencil_k(const REAL* P, const REAL* B, REAL* OUT, Coef c, Geom g, int par, double* partials,
          const int* __restrict__ skip, Fin fin, MafArgs ma) {
  if (skip != nullptr && *skip != 0) return;

  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int t = threadIdx.x;
  const int R = g.R;
  const int L = g.S + 2 * R;  // vectors per LDS buffer
  Vec<V>* ldsv = reinterpret_cast<Vec<V>*>(smem);
  REAL* ldsf = reinterpret_cast<REAL*>(smem);
  double* wsum = reinterpret_cast<double*>(smem + (size_t)2 * L * sizeof(Vec<V>));

  // XCD-aware remap: hardware deals consecutive workgroup ids round-robin over the 8 XCDs; give each XCD a
  // contiguous run of logical ids so that row-adjacent segments (which share halo rows) meet in one L2.
  int lb = blockIdx.x;
  const int nblk = gridDim.x;
  if ((nblk & 7) == 0) lb = (lb & 7) * (nblk >> 3) + (lb >> 3);
  const int seg = lb % g.nseg;
  const int chunk = lb / g.nseg;

  const long long fb = g.F0 + (long long)seg * g.S;
  const int ja = g.jj0 + chunk * g.TJ;
  int jb = ja + g.TJ - 1;
  if (jb > g.jj1) jb = g.jj1;

  double acc = 0.0, acc2 = 0.0;
  const bool ax_dots = (MODE == MODE_AX) && fin.ax_dots;
  const bool ldb = (MODE != MODE_AX) || ax_dots;           // does the step need the second input vector?
  const REAL* Bsrc = (MODE == MODE_AX) ? fin.doty : B;      // b of the sweep / y of the fused dot products

  if (ja <= jb && fb < g.Fend) {
    // per-vector constants of this thread
    long long f[M];
    unsigned mk[M];   // bit c set: component c is an inner point (k range, valid row)
    int pbase[M];     // (kk + ii + par) & 1 of component 0 (MODE_RB)
    REAL XG[MAF ? M : 1], XGG[MAF ? M : 1];   // MAF: metric terms of the row ...
    Vec<V> ZT[MAF ? M : 1], ZTT[MAF ? M : 1];  // ... and of each k component
    const long long lim_ld = g.Fend + R;  // vectors below this exist in the plane (row ii1+1 is a halo row)
    long long eo[M];  // element offset of the vector inside a plane in memory (rows of g.nkp elements; f counts R vectors per row)
    auto eoff = [&](long long ff) -> long long {
      const long long r = ff / R;
      return r * g.nkp + (ff - r * R) * V;
    };
    // (the array's last plane: Geom::last_eo)
    auto lim = [&](long long e, int plane) -> long long { return plane == g.jlast ? (e < g.last_eo ? e : g.last_eo) : e; };
#pragma unroll
    for (int m = 0; m < M; m++) {
      f[m] = fb + t + m * TB;
      const long long row = f[m] / R;
      const int kv = (int)(f[m] - row * R);
      eo[m] = row * g.nkp + (long long)kv * V;
      unsigned bits = 0;
      if (f[m] < g.Fend) {
#pragma unroll
        for (int cc = 0; cc < V; cc++) {
          const int kk = kv * V + cc;
          if (kk >= g.kk0 && kk <= g.kk1) bits |= 1u << cc;
        }
      }
      mk[m] = bits;
      pbase[m] = (kv * V + (int)row + par) & 1;
      if (MAF) {
        const int nkp = g.nkp;
        int ii = (int)row;  // padded row index == index into xc for g = 2
        if (ii < 1) ii = 1;
        if (ii > g.nip - 2) ii = g.nip - 2;
        const REAL xm = ma.xc[ii - 1], x0 = ma.xc[ii], xp = ma.xc[ii + 1];
        XG[m] = (REAL)0.5 * (xp - xm);
        XGG[m] = xp - (REAL)2.0 * x0 + xm;
#pragma unroll
        for (int cc = 0; cc < V; cc++) {
          int kk = kv * V + cc;
          if (kk < 1) kk = 1;
          if (kk > nkp - 2) kk = nkp - 2;
          const REAL zm = ma.zc[kk - 1], z0 = ma.zc[kk], zp = ma.zc[kk + 1];
          ZT[m].v[cc] = (REAL)0.5 * (zp - zm);
          ZTT[m].v[cc] = zp - (REAL)2.0 * z0 + zm;
        }
      }
    }

    Vec<V> pm[M], pc[M], pn[M], bb[M];
    Vec<V> pnn[PF ? M : 1], bbn[PF ? M : 1];

    const REAL* Pm = P + (long long)(ja - 1) * g.PSE;
    const REAL* Pc = P + (long long)ja * g.PSE;
#pragma unroll
    for (int m = 0; m < M; m++) {
      const bool ok = f[m] < lim_ld;
      pm[m] = ok ? ldve<V>(Pm, eo[m]) : zerov<V>();
      pc[m] = ok ? ldve<V>(Pc, eo[m]) : zerov<V>();
    }
    // stage plane ja (own vectors + halo rows) into LDS buffer 0
    {
      Vec<V>* buf = ldsv;
#pragma unroll
      for (int m = 0; m < M; m++) buf[R + t + m * TB] = pc[m];
      for (int h = t; h < R; h += TB) {
        buf[h] = ldve<V>(Pc, eoff(fb - R + h));
        const long long fh = fb + g.S + h;
        buf[R + g.S + h] = (fh < lim_ld) ? ldve<V>(Pc, eoff(fh)) : zerov<V>();
      }
    }
    if (PF) {
      const REAL* Pn = P + (long long)(ja + 1) * g.PSE;
      const REAL* Bc = Bsrc + (long long)ja * g.PSE;
#pragma unroll
      for (int m = 0; m < M; m++) {
        pn[m] = (f[m] < lim_ld) ? ldve<V>(Pn, lim(eo[m], ja + 1)) : zerov<V>();
        if (ldb) bb[m] = (f[m] < g.Fend) ? ldve<V>(Bc, eo[m]) : zerov<V>();
      }
    }
    // the halo rows this thread stages for the next centre plane (the same vectors of every plane)
    const long long eo_lo = (t < R) ? eoff(fb - R + t) : 0, eo_hi = (t < R && fb + g.S + t < lim_ld) ? eoff(fb + g.S + t) : 0;
    __syncthreads();

    int cur = 0;
    for (int jj = ja; jj <= jb; jj++) {
      const bool more = jj < jb;
      const REAL* Pn = P + (long long)(jj + 1) * g.PSE;
      // ---- issue the loads of the following step early
      if (PF) {
        if (more) {
          const REAL* Pnn = Pn + g.PSE;
          const REAL* Bn = Bsrc + (long long)(jj + 1) * g.PSE;
#pragma unroll
          for (int m = 0; m < M; m++) {
            pnn[m] = (f[m] < lim_ld) ? ldve<V>(Pnn, lim(eo[m], jj + 2)) : zerov<V>();
            if (ldb) bbn[m] = (f[m] < g.Fend) ? ldve<V>(Bn, eo[m]) : zerov<V>();
          }
        }
      } else {
        const REAL* Bc = Bsrc + (long long)jj * g.PSE;
#pragma unroll
        for (int m = 0; m < M; m++) {
          pn[m] = (f[m] < lim_ld) ? ldve<V>(Pn, lim(eo[m], jj + 1)) : zerov<V>();
          if (ldb) bb[m] = (f[m] < g.Fend) ? ldve<V>(Bc, eo[m]) : zerov<V>();
        }
      }
      // halo rows of the next centre plane (only the first R threads; R <= TB in the common case)
      Vec<V> hlo = zerov<V>(), hhi = zerov<V>();
      const bool halo_in_regs = (R <= TB);
      if (more && halo_in_regs && t < R) {
        hlo = ldve<V>(Pn, lim(eo_lo, jj + 1));
        const long long fh = fb + g.S + t;
        if (fh < lim_ld) hhi = ldve<V>(Pn, lim(eo_hi, jj + 1));
      }

      // ---- update plane jj
      const Vec<V>* buf = ldsv + (size_t)cur * L;
      const REAL* buff = ldsf + (size_t)cur * L * V;
      REAL* Oc = OUT + (long long)jj * g.PSE;
      REAL YE = (REAL)0, YEE = (REAL)0;
      if (MAF) {
        const REAL ym = ma.yc[jj - 1], y0 = ma.yc[jj], yp = ma.yc[jj + 1];
        YE = (REAL)0.5 * (yp - ym);
        YEE = yp - (REAL)2.0 * y0 + ym;
      }
#pragma unroll
      for (int m = 0; m < M; m++) {
        if (mk[m] == 0) continue;
        const int li = t + m * TB;
        const Vec<V> im = buf[li];
        const Vec<V> ip = buf[li + 2 * R];
        const REAL kl = buff[(R + li) * V - 1];
        const REAL kr = buff[(R + li) * V + V];
        Vec<V> o;
        unsigned wmask = mk[m];
        Vec<V> pv;
        if (MAF && (MODE == MODE_AX || MODE == MODE_RK)) pv = ldve<V>(ma.pvt + (long long)jj * g.PSE, eo[m]);
        if (MODE == MODE_RB) {
          // colour: (kk + ii + jj + par) even
          unsigned cm = 0;
#pragma unroll
          for (int cc = 0; cc < V; cc++)
            if (((pbase[m] + cc + jj) & 1) == 0) cm |= 1u << cc;
          wmask &= cm;
        }
#pragma unroll
        for (int cc = 0; cc < V; cc++) {
          const REAL pp = pc[m].v[cc];
          const REAL km1 = (cc == 0) ? kl : pc[m].v[cc > 0 ? cc - 1 : 0];
          const REAL kp1 = (cc == V - 1) ? kr : pc[m].v[cc < V - 1 ? cc + 1 : V - 1];
          if (MAF) {
            const MafW w = maf_weights(XG[m], XGG[m], YE, YEE, ZT[m].v[cc], ZTT[m].v[cc]);
            if (MODE == MODE_JACOBI || MODE == MODE_RB) {
              const REAL rp = w.w1 * ip.v[cc] + w.w2 * im.v[cc] + w.w3 * pn[m].v[cc] + w.w4 * pm[m].v[cc] + w.w5 * kp1 +
                              w.w6 * km1 + bb[m].v[cc];  // cz_maf.f90:219-225
              const REAL dp = (rp / w.dd - pp) * c.omg;
              o.v[cc] = pp + dp;
              const REAL d2 = dp * dp;
              if (wmask & (1u << cc)) acc += (double)d2;
            } else if (MODE == MODE_AX) {  // cz_blas.f90:916-924
              o.v[cc] = (w.w1 * ip.v[cc] + w.w2 * im.v[cc] + w.w3 * pn[m].v[cc] + w.w4 * pm[m].v[cc] + w.w5 * kp1 + w.w6 * km1 -
                         w.dd * pp) * pv.v[cc];
            } else {  // cz_blas.f90:811-820
              o.v[cc] = (bb[m].v[cc] + w.dd * pp - w.w1 * ip.v[cc] - w.w2 * im.v[cc] - w.w3 * pn[m].v[cc] - w.w4 * pm[m].v[cc] -
                         w.w5 * kp1 - w.w6 * km1) * pv.v[cc];
            }
            continue;
          }
          const REAL ss = c.c1 * ip.v[cc] + c.c2 * im.v[cc] + c.c3 * pn[m].v[cc] + c.c4 * pm[m].v[cc] + c.c5 * kp1 +
                          c.c6 * km1;
          if (MODE == MODE_JACOBI || MODE == MODE_RB) {
            const REAL dp = ((ss - bb[m].v[cc]) / c.dd - pp) * c.omg;
            o.v[cc] = pp + dp;
            const REAL d2 = dp * dp;
            if (wmask & (1u << cc)) acc += (double)d2;
          } else if (MODE == MODE_AX) {
            o.v[cc] = ss - c.dd * pp;
          } else {
            o.v[cc] = bb[m].v[cc] - (ss - c.dd * pp);
          }
        }
        if (ax_dots) {
#pragma unroll
          for (int cc = 0; cc < V; cc++) {
            const REAL oy = o.v[cc] * bb[m].v[cc];
            const REAL oo = o.v[cc] * o.v[cc];
            if (wmask & (1u << cc)) {
              acc += (double)oy;
              acc2 += (double)oo;
            }
          }
        }
        if (MODE == MODE_RB) {
          // in place: components of the other colour / outside the box keep their value; a full-vector store
          // of unchanged bits is harmless because every element is owned by exactly one thread.
          if (mk[m] == (1u << V) - 1) {
#pragma unroll
            for (int cc = 0; cc < V; cc++)
              if (!(wmask & (1u << cc))) o.v[cc] = pc[m].v[cc];
            stve<V>(Oc, eo[m], o);
          } else {
#pragma unroll
            for (int cc = 0; cc < V; cc++)
              if (wmask & (1u << cc)) Oc[eo[m] + cc] = o.v[cc];
          }
        } else {
          if (wmask == (1u << V) - 1) {
            stve<V>(Oc, eo[m], o);
          } else {
#pragma unroll
            for (int cc = 0; cc < V; cc++)
              if (wmask & (1u << cc)) Oc[eo[m] + cc] = o.v[cc];
          }
        }
      }

      // ---- stage plane jj+1 into the other LDS buffer, rotate the register queue
      if (more) {
        Vec<V>* nbuf = ldsv + (size_t)(cur ^ 1) * L;
#pragma unroll
        for (int m = 0; m < M; m++) nbuf[R + t + m * TB] = pn[m];
        if (halo_in_regs) {
          if (t < R) {
            nbuf[t] = hlo;
            nbuf[R + g.S + t] = hhi;
          }
        } else {
          for (int h = t; h < R; h += TB) {
            nbuf[h] = ldve<V>(Pn, lim(eoff(fb - R + h), jj + 1));
            const long long fh = fb + g.S + h;
            nbuf[R + g.S + h] = (fh < lim_ld) ? ldve<V>(Pn, lim(eoff(fh), jj + 1)) : zerov<V>();
          }
        }
      }
      __syncthreads();
#pragma unroll
      for (int m = 0; m < M; m++) {
        pm[m] = pc[m];
        pc[m] = pn[m];
        if (PF) {
          pn[m] = pnn[m];
          bb[m] = bbn[m];
        }
      }
      cur ^= 1;
    }
  }

  if (MODE == MODE_JACOBI || MODE == MODE_RB || ax_dots) {
    __syncthreads();
    const double s = block_sum<TB>(acc, wsum);
    double s2 = 0.0;
    if (ax_dots) {
      __syncthreads();
      s2 = block_sum<TB>(acc2, wsum);
    }
    if (fin.dst == nullptr) {
      if (t == 0) partials[lb] = s;
    } else {
      int* last_flag = reinterpret_cast<int*>(wsum + 16);
      if (t == 0) {
        // write-through (sc1) store of the partial, drained, then the ticket: no L2 write-back fence per workgroup
        // (a release fence here flushes the XCD's dirty p' lines and cost +27 % on the whole sweep, profiles/README.md)
        __hip_atomic_store(&partials[lb], s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (ax_dots) __hip_atomic_store(&partials[nblk + lb], s2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        *last_flag = arrive_and_test_last(fin.counter, nblk);
      }
      __syncthreads();
      if (*last_flag) {
        double x = 0.0, x2 = 0.0;
        // every load of the handed-off partials is an sc1 (agent-scope) load
        for (int i = t; i < nblk; i += TB) {
          x += __hip_atomic_load(&partials[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (ax_dots) x2 += __hip_atomic_load(&partials[nblk + i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
        const double tot = block_sum<TB>(x, wsum);
        double tot2 = 0.0;
        if (ax_dots) {
          __syncthreads();
          tot2 = block_sum<TB>(x2, wsum);
        }
        if (t == 0) {
          double r = fin.accumulate ? fin.dst[0] + tot : tot;
          fin.dst[0] = r;
          if (ax_dots) fin.dst2[0] = tot2;
          if (fin.do_check) {  // cz_Poisson.cpp:69-77
            r *= fin.res_normal;
            r = sqrt(r);
            fin.hist[fin.itr] = r;
            if (r < fin.eps) {
              *fin.flag = 1;
              *fin.conv_itr = fin.itr;
            }
          }
          *fin.counter = 0u;
        }
      }
    }
  }
}
