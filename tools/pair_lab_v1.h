// tools/pair_lab_v1.h -- round 1's two-stage pass kernel (jacobi2_k), kept ONLY as the A/B partner of tools/pair_lab.hip: the library
// runs jacobi2p_k (cubez_amd/csrc/cz_k_pair2.h), which must reproduce this kernel's output bit for bit.  Included inside pair_lab.hip's
// anonymous namespace after cz_k_pair.h (Geom2, Fin2, relax_vec, pair_finalize).  Loads are issued at the top of a plane step and waited
// for a few instructions later; j-1 operands ride in register queues; k neighbours across vectors are stride-4 ds_read_b32.
// RB = 0: two Jacobi sweeps.  RB = 1: one red-black SOR iteration (cz_solver.f90:466-480 for colour 0 then colour 1):
// stage 1 updates the points of colour 0, stage 2 those of colour 1 from the freshly updated colour-0 neighbours; the
// other colour passes through each stage unchanged.  Out of place (U -> W) like the Jacobi pair.
template <int V, int TB, int MV, int RB>
__global__ void __launch_bounds__(TB, (TB == 512 && MV <= 2) ? 4 : 1)
jacobi2_k(const REAL* __restrict__ U, const REAL* __restrict__ B, REAL* __restrict__ W, Coef c, Geom2 g, double* partials,
          const int* __restrict__ skip, Fin2 fin) {
  if (skip != nullptr && *skip != 0) return;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int t = threadIdx.x;
  const int R = g.R;
  const int LU = g.S + 4 * R, LV = g.S + 2 * R;
  Vec<V>* ldsU = reinterpret_cast<Vec<V>*>(smem);                 // 2 buffers of LU vectors
  Vec<V>* ldsV = ldsU + (size_t)2 * LU;                            // 2 buffers of LV vectors
  double* wsum = reinterpret_cast<double*>(ldsV + (size_t)2 * LV);  // 16 doubles + flag

  int lb = blockIdx.x;
  const int nblk = gridDim.x;
  int seg, chunk;
  if (g.band) {
    // XCD bands: the hardware deals workgroup ids round-robin over the 8 XCDs; XCD x owns a contiguous band of segments of EVERY
    // chunk (row-adjacent segments meet in one L2) and walks it chunk by chunk -- all XCDs carry the same load whatever the
    // number of chunks is.  Ids beyond a shorter band are idle.
    const int x = lb & 7, r = lb >> 3;
    const int base = g.nseg >> 3, rem = g.nseg & 7, bmax = base + (rem ? 1 : 0);
    const int blen = base + (x < rem ? 1 : 0);
    const int sl = r % bmax;
    chunk = r / bmax;
    seg = (sl < blen) ? x * base + min(x, rem) + sl : g.nseg;  // nseg = no work
  } else {
    if ((nblk & 7) == 0) lb = (lb & 7) * (nblk >> 3) + (lb >> 3);
    seg = lb % g.nseg;
    chunk = lb / g.nseg;
  }
  lb = blockIdx.x;  // slot of this workgroup's partial sums
  const long long fb = (seg < g.nseg) ? g.F0 + (long long)seg * g.S : g.Fend;
  const int ja = g.jj0 + chunk * g.TJ;
  int jb = ja + g.TJ - 1;
  if (jb > g.jj1) jb = g.jj1;

  double acc1 = 0.0, acc2 = 0.0;
  const PlainDiv dv{c.dd};

  if (ja <= jb && fb < g.Fend) {
    const long long e1_0 = fb - R;       // first vector of E1
    const long long e2_0 = fb - 2 * R;   // first vector of E2
    long long f[MV];
    unsigned ka[MV];     // stage-1 bits: components of the vector inside the stage-1 box (0 when the row is outside)
    unsigned own[MV];    // stage-2 bits if this workgroup owns the vector (stores, residual counts), else 0
    int pbase[MV];       // RB: (kk + ii + par) of component 0; component cc on plane jj has colour (pbase + cc + jj) & 1
    bool ld[MV];
#pragma unroll
    for (int m = 0; m < MV; m++) {
      const int e = t + m * TB;
      f[m] = e1_0 + e;
      ld[m] = (e < LV) && (f[m] < g.PSV) && !g.zero_u;
      const long long row = f[m] / R;
      const int kv = (int)(f[m] - row * R);
      unsigned bits1 = 0, bits2 = 0;
#pragma unroll
      for (int cc = 0; cc < V; cc++) {
        const int kk = kv * V + cc;
        if (kk >= g.kk0a && kk <= g.kk1a) bits1 |= 1u << cc;
        if (kk >= g.kk0 && kk <= g.kk1) bits2 |= 1u << cc;
      }
      pbase[m] = kv * V + (int)row + g.par;
      ka[m] = (e < LV && f[m] >= g.F0a && f[m] < g.Fenda) ? bits1 : 0u;
      own[m] = (e >= R && e < R + g.S && f[m] >= g.F0 && f[m] < g.Fend) ? bits2 : 0u;
    }

    Vec<V> ua[MV], ub[MV], uc[MV], b1[MV], b2[MV], va[MV], vb[MV], vc[MV];
    // prologue: u(ja-2), u(ja-1); LDS_U[0] = u(ja-1) on E2
    {
      const REAL* Ua = U + (long long)(ja - 2) * g.PSV * V;
      const REAL* Ub = U + (long long)(ja - 1) * g.PSV * V;
#pragma unroll
      for (int m = 0; m < MV; m++) {
        ua[m] = ld[m] ? ldv<V>(Ua, f[m]) : zerov<V>();
        ub[m] = ld[m] ? ldv<V>(Ub, f[m]) : zerov<V>();
        b2[m] = zerov<V>();
        va[m] = zerov<V>();
        vb[m] = zerov<V>();
      }
#pragma unroll
      for (int m = 0; m < MV; m++)
        if (t + m * TB < LV) ldsU[R + t + m * TB] = ub[m];
      if (t < R) {
        const long long fh = fb + g.S + R + t;
        ldsU[t] = g.zero_u ? zerov<V>() : ldv<V>(Ub, e2_0 + t);
        ldsU[R + LV + t] = (fh < g.PSV && !g.zero_u) ? ldv<V>(Ub, fh) : zerov<V>();
      }
    }
    __syncthreads();

    int cur = 0;
    for (int q = ja - 1; q <= jb + 1; q++) {
      const bool more = q <= jb;
      const bool plane_inner = (q >= g.jj0a && q <= g.jj1a);
      const bool count1 = (q >= ja && q <= jb);
      const bool do2 = (q - 1 >= ja);
      // ---- loads of this step: u(q+1) and b(q) on E1, outer halo rows of u(q+1)
      const REAL* Uc = U + (long long)(q + 1) * g.PSV * V;
      const REAL* Bq = B + (long long)q * g.PSV * V;
#pragma unroll
      for (int m = 0; m < MV; m++) {
        uc[m] = ld[m] ? ldv<V>(Uc, f[m]) : zerov<V>();
        b1[m] = (ka[m] != 0 && plane_inner) ? ldv<V>(Bq, f[m]) : zerov<V>();
      }
      Vec<V> hlo = zerov<V>(), hhi = zerov<V>();
      if (more && t < R && !g.zero_u) {
        hlo = ldv<V>(Uc, e2_0 + t);
        const long long fh = fb + g.S + R + t;
        if (fh < g.PSV) hhi = ldv<V>(Uc, fh);
      }

      // ---- stage 1: v(q) on E1
      const Vec<V>* bufU = ldsU + (size_t)cur * LU;
      const REAL* bufUf = reinterpret_cast<const REAL*>(bufU);
#pragma unroll
      for (int m = 0; m < MV; m++) {
        const int e = t + m * TB;
        if (e >= LV) continue;
        unsigned msk = plane_inner ? ka[m] : 0u;
        if (RB) msk &= colour_bits<V>(pbase[m] + q);  // colour 0 on plane q
        if (msk == 0) {
          vc[m] = ub[m];  // outside the inner box: the first sweep leaves the value alone
        } else {
          const int x = e + R;
          const Vec<V> im = bufU[x - R];
          const Vec<V> ip = bufU[x + R];
          const REAL kl = bufUf[x * V - 1];
          const REAL kr = bufUf[x * V + V];
          vc[m] = relax_vec<V>(ub[m], im, ip, ua[m], uc[m], kl, kr, b1[m], c, dv, msk, count1 ? (own[m] & msk) : 0u, acc1);
        }
      }
      // ---- publish v(q) for the next step's stage 2
      {
        Vec<V>* nV = ldsV + (size_t)(cur ^ 1) * LV;
#pragma unroll
        for (int m = 0; m < MV; m++)
          if (t + m * TB < LV) nV[t + m * TB] = vc[m];
      }
      // ---- stage 2: w(q-1) on the own segment
      if (do2) {
        const Vec<V>* bufV = ldsV + (size_t)cur * LV;
        const REAL* bufVf = reinterpret_cast<const REAL*>(bufV);
        REAL* Wq = W + (long long)(q - 1) * g.PSV * V;
#pragma unroll
        for (int m = 0; m < MV; m++) {
          if (own[m] == 0) continue;
          const int e = t + m * TB;
          const Vec<V> im = bufV[e - R];
          const Vec<V> ip = bufV[e + R];
          const REAL kl = bufVf[e * V - 1];
          const REAL kr = bufVf[e * V + V];
          unsigned m2 = own[m];
          if (RB) m2 &= colour_bits<V>(pbase[m] + (q - 1) + 1);  // colour 1 on plane q-1
          const Vec<V> o = relax_vec<V>(vb[m], im, ip, va[m], vc[m], kl, kr, b2[m], c, dv, m2, m2, acc2);
          if (own[m] == (1u << V) - 1) {
            stv<V>(Wq, f[m], o);
          } else {
#pragma unroll
            for (int cc = 0; cc < V; cc++)
              if (own[m] & (1u << cc)) Wq[f[m] * V + cc] = o.v[cc];
          }
        }
      }
      // ---- stage the next u centre plane, rotate
      if (more) {
        Vec<V>* nU = ldsU + (size_t)(cur ^ 1) * LU;
#pragma unroll
        for (int m = 0; m < MV; m++)
          if (t + m * TB < LV) nU[R + t + m * TB] = uc[m];
        if (t < R) {
          nU[t] = hlo;
          nU[R + LV + t] = hhi;
        }
      }
      __syncthreads();
#pragma unroll
      for (int m = 0; m < MV; m++) {
        ua[m] = ub[m];
        ub[m] = uc[m];
        b2[m] = b1[m];
        va[m] = vb[m];
        vb[m] = vc[m];
      }
      cur ^= 1;
    }
  }

  // ---- residuals: per-workgroup partials, finalised by the last workgroup (write-through hand-off, see stencil_k)
  __syncthreads();
  const double s1 = block_sum<TB>(acc1, wsum);
  __syncthreads();
  const double s2 = block_sum<TB>(acc2, wsum);
  int* last_flag = reinterpret_cast<int*>(wsum + 16);
  if (t == 0) {
    __hip_atomic_store(&partials[lb], s1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(&partials[nblk + lb], s2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned ticket = __hip_atomic_fetch_add(fin.counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    *last_flag = (ticket == (unsigned)nblk - 1u);
  }
  __syncthreads();
  if (*last_flag) pair_finalize<TB>(partials, nblk, fin, wsum);
}

