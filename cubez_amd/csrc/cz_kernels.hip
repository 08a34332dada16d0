// cz_kernels.hip -- hand-written gfx950 (CDNA4) kernels of the CubeZ hot path + the C-ABI of include/cz_hip.h
// parts 1-3.  One translation unit per precision (-DCZ_REAL_IS_DOUBLE for FP64), compiled with
// -ffp-contract=off so that every sweep reproduces the reference's un-fused FP32/FP64 arithmetic bit for bit
// (SURVEY.md section 7 "hard parts").
//
// Data layout (cz_solver.f90:29): arrays are dense (NK+2g, NI+2g, NJ+2g), K fastest.  A (k,i) PLANE of one j is
// therefore one contiguous run of (NK+2g)*(NI+2g) elements: row i+1 follows row i directly.  The stencil kernel
// exploits this by treating a plane as a 1-D array of 16-byte vectors (float4 / double2):
//     k+-1 neighbour = +-1 element, i+-1 neighbour = +-R vectors (R = (NK+2g)/V), j+-1 neighbour = +-1 plane.
// A workgroup owns a contiguous SEGMENT of S = TB*M vectors of the plane (about S/R rows) and marches it through
// a CHUNK of TJ consecutive planes (2.5-D blocking): plane j-1/j/j+1 values of its own vectors live in registers
// (a 3-deep queue that rotates as j advances), the centre plane j of the segment plus one row of halo on either
// side is staged in LDS (double-buffered, one barrier per plane) for the i+-1 and the vector-crossing k+-1
// neighbours.  Every global access is a 16-byte-per-lane fully coalesced load/store; each p and b element is read
// from HBM once per sweep (plus R-vector row halos shared with the neighbouring workgroup through L2 and two
// planes per chunk), each new element is written once.  No MFMA: 18 flop per 12 bytes is far below the ridge.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <vector>

#include "cz_hip.h"
#include "cz_internal.h"

typedef CZ_REAL REAL;
#ifdef CZ_REAL_IS_DOUBLE
constexpr int VW = 2;  // elements per 16-byte vector
#else
constexpr int VW = 4;
#endif

namespace {

template <int V>
struct alignas(sizeof(REAL) * V) Vec {
  REAL v[V];
};

struct Coef {
  REAL c1, c2, c3, c4, c5, c6, dd, omg;
};

// Geometry of one launch, all in PADDED 0-based indices (kk = k+g-1, ...).
struct Geom {
  int nip;          // padded rows per plane = NI+2g
  int R;            // vectors per k-row = (NK+2g)/V
  long long PSV;    // vectors per plane = R*(NI+2g)
  int kk0, kk1;     // inner k range (inclusive)
  int jj0, jj1;     // inner j range (inclusive)
  long long F0;     // first vector of the update range inside a plane = ii0*R
  long long Fend;   // one past the last vector of the update range    = (ii1+1)*R
  int nseg;         // segments per plane
  int TJ;           // planes per chunk
  int S;            // vectors per segment
};

enum { MODE_JACOBI = 0, MODE_RB = 1, MODE_AX = 2, MODE_RK = 3 };

// In-kernel finalisation of the residual: the workgroup that arrives last sums the per-workgroup partials in a fixed
// order (deterministic) and, if asked, performs the convergence bookkeeping of cz_Poisson.cpp:67-77 -- no extra
// launches per sweep.  Hand-off follows cdna_hip_programming.md Guideline 16 in its write-through form: sc1 store of
// the partial -> s_waitcnt vmcnt(0) -> agent-scope ticket add; the last arriver reads every partial with sc1 loads.
struct Fin {
  double* dst = nullptr;  // device double receiving sum dp^2 (nullptr: leave the partials for a separate reduce launch)
  int accumulate = 0;     // dst += instead of dst =
  int do_check = 0;       // also: res = sqrt(dst*res_normal); hist[itr] = res; eps test -> flag/conv_itr
  int itr = 0;
  double res_normal = 0.0, eps = 0.0;
  double* hist = nullptr;
  int* flag = nullptr;
  int* conv_itr = nullptr;
  unsigned* counter = nullptr;  // arrival ticket, zero before every launch (the last workgroup resets it)
  // MODE_AX only: fold the dot products that follow the SpMV in BiCGSTAB into it (cz_Poisson.cpp:421-427, 457-464):
  // dst[0] = sum out*y, dst2[0] = sum out*out over the inner box (per-point products rounded to REAL like blas_dot1/2)
  int ax_dots = 0;
  const REAL* doty = nullptr;
  double* dst2 = nullptr;
};

// 16-byte global accesses go through a native vector type so that hipcc emits one global_load/store_dwordx4
// (a struct copy was split into dwordx3 + dword stores).
template <int V>
struct NatVec {
  typedef REAL type __attribute__((ext_vector_type(V)));
};
template <>
struct NatVec<1> {
  typedef REAL type;
};
template <int V>
__device__ __forceinline__ Vec<V> ldv(const REAL* base, long long vec_index) {
  typedef typename NatVec<V>::type nv;
  const nv x = *reinterpret_cast<const nv*>(base + vec_index * V);
  Vec<V> r;
  __builtin_memcpy(&r, &x, sizeof(r));
  return r;
}
template <int V>
__device__ __forceinline__ void stv(REAL* base, long long vec_index, const Vec<V>& x) {
  typedef typename NatVec<V>::type nv;
  nv y;
  __builtin_memcpy(&y, &x, sizeof(y));
  *reinterpret_cast<nv*>(base + vec_index * V) = y;
}
template <int V>
__device__ __forceinline__ Vec<V> zerov() {
  Vec<V> z;
#pragma unroll
  for (int c = 0; c < V; c++) z.v[c] = (REAL)0;
  return z;
}

// deterministic block reduction of one double per thread: wave64 shuffle tree, then LDS across waves.
template <int TB>
__device__ __forceinline__ double block_sum(double x, double* wsum /* TB/64 doubles of LDS */) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off, 64);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (lane == 0) wsum[wave] = x;
  __syncthreads();
  double s = 0.0;
  if (threadIdx.x == 0) {
#pragma unroll
    for (int w = 0; w < TB / 64; w++) s += wsum[w];
  }
  return s;  // valid on thread 0
}

// MAF flavour (cz_maf.f90, cz_blas.f90:738-1039): the six neighbour weights and the diagonal are
// recomputed at every point from 1-D coordinate arrays (device copies of xc, yc, zc; X(i) of the Fortran is xc[i+1], which
// for g = 2 is xc[padded index]).  pvt: row scaling of calc_ax_maf / calc_rk_maf.
struct MafArgs {
  const REAL* xc;
  const REAL* yc;
  const REAL* zc;
  const REAL* pvt;
};

struct MafW {
  REAL w1, w2, w3, w4, w5, w6, dd;  // weights of p(i+1), p(i-1), p(j+1), p(j-1), p(k+1), p(k-1); dd = 2(C1+C2+C3)
};

// cz_maf.f90:193-221, operation for operation
__device__ __forceinline__ MafW maf_weights(REAL XG, REAL XGG, REAL YE, REAL YEE, REAL ZT, REAL ZTT) {
  const REAL YJA = XG * YE * ZT;
  const REAL YJAI = (REAL)1.0 / YJA;
  const REAL GX = YE * ZT * YJAI;
  const REAL EY = XG * ZT * YJAI;
  const REAL TZ = XG * YE * YJAI;
  const REAL C1 = GX * GX, C2 = EY * EY, C3 = TZ * TZ;
  const REAL C7 = -XGG * C1 * GX;
  const REAL C8 = -YEE * C2 * EY;
  const REAL C9 = -ZTT * C3 * TZ;
  MafW w;
  w.w1 = C1 + (REAL)0.5 * C7, w.w2 = C1 - (REAL)0.5 * C7;
  w.w3 = C2 + (REAL)0.5 * C8, w.w4 = C2 - (REAL)0.5 * C8;
  w.w5 = C3 + (REAL)0.5 * C9, w.w6 = C3 - (REAL)0.5 * C9;
  w.dd = (REAL)2.0 * (C1 + C2 + C3);
  return w;
}

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
#pragma unroll
    for (int m = 0; m < M; m++) {
      f[m] = fb + t + m * TB;
      const long long row = f[m] / R;
      const int kv = (int)(f[m] - row * R);
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
        const int nkp = R * V;
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

    const REAL* Pm = P + (long long)(ja - 1) * g.PSV * V;
    const REAL* Pc = P + (long long)ja * g.PSV * V;
#pragma unroll
    for (int m = 0; m < M; m++) {
      const bool ok = f[m] < lim_ld;
      pm[m] = ok ? ldv<V>(Pm, f[m]) : zerov<V>();
      pc[m] = ok ? ldv<V>(Pc, f[m]) : zerov<V>();
    }
    // stage plane ja (own vectors + halo rows) into LDS buffer 0
    {
      Vec<V>* buf = ldsv;
#pragma unroll
      for (int m = 0; m < M; m++) buf[R + t + m * TB] = pc[m];
      for (int h = t; h < R; h += TB) {
        buf[h] = ldv<V>(Pc, fb - R + h);
        const long long fh = fb + g.S + h;
        buf[R + g.S + h] = (fh < lim_ld) ? ldv<V>(Pc, fh) : zerov<V>();
      }
    }
    if (PF) {
      const REAL* Pn = P + (long long)(ja + 1) * g.PSV * V;
      const REAL* Bc = Bsrc + (long long)ja * g.PSV * V;
#pragma unroll
      for (int m = 0; m < M; m++) {
        pn[m] = (f[m] < lim_ld) ? ldv<V>(Pn, f[m]) : zerov<V>();
        if (ldb) bb[m] = (f[m] < g.Fend) ? ldv<V>(Bc, f[m]) : zerov<V>();
      }
    }
    __syncthreads();

    int cur = 0;
    for (int jj = ja; jj <= jb; jj++) {
      const bool more = jj < jb;
      const REAL* Pn = P + (long long)(jj + 1) * g.PSV * V;
      // ---- issue the loads of the following step early
      if (PF) {
        if (more) {
          const REAL* Pnn = Pn + g.PSV * V;
          const REAL* Bn = Bsrc + (long long)(jj + 1) * g.PSV * V;
#pragma unroll
          for (int m = 0; m < M; m++) {
            pnn[m] = (f[m] < lim_ld) ? ldv<V>(Pnn, f[m]) : zerov<V>();
            if (ldb) bbn[m] = (f[m] < g.Fend) ? ldv<V>(Bn, f[m]) : zerov<V>();
          }
        }
      } else {
        const REAL* Bc = Bsrc + (long long)jj * g.PSV * V;
#pragma unroll
        for (int m = 0; m < M; m++) {
          pn[m] = (f[m] < lim_ld) ? ldv<V>(Pn, f[m]) : zerov<V>();
          if (ldb) bb[m] = (f[m] < g.Fend) ? ldv<V>(Bc, f[m]) : zerov<V>();
        }
      }
      // halo rows of the next centre plane (only the first R threads; R <= TB in the common case)
      Vec<V> hlo = zerov<V>(), hhi = zerov<V>();
      const bool halo_in_regs = (R <= TB);
      if (more && halo_in_regs && t < R) {
        hlo = ldv<V>(Pn, fb - R + t);
        const long long fh = fb + g.S + t;
        if (fh < lim_ld) hhi = ldv<V>(Pn, fh);
      }

      // ---- update plane jj
      const Vec<V>* buf = ldsv + (size_t)cur * L;
      const REAL* buff = ldsf + (size_t)cur * L * V;
      REAL* Oc = OUT + (long long)jj * g.PSV * V;
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
        if (MAF && (MODE == MODE_AX || MODE == MODE_RK)) pv = ldv<V>(ma.pvt + (long long)jj * g.PSV * V, f[m]);
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
            stv<V>(Oc, f[m], o);
          } else {
#pragma unroll
            for (int cc = 0; cc < V; cc++)
              if (wmask & (1u << cc)) Oc[f[m] * V + cc] = o.v[cc];
          }
        } else {
          if (wmask == (1u << V) - 1) {
            stv<V>(Oc, f[m], o);
          } else {
#pragma unroll
            for (int cc = 0; cc < V; cc++)
              if (wmask & (1u << cc)) Oc[f[m] * V + cc] = o.v[cc];
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
            nbuf[h] = ldv<V>(Pn, fb - R + h);
            const long long fh = fb + g.S + h;
            nbuf[R + g.S + h] = (fh < lim_ld) ? ldv<V>(Pn, fh) : zerov<V>();
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
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned ticket = __hip_atomic_fetch_add(fin.counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        *last_flag = (ticket == (unsigned)nblk - 1u);
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


// ------------------------------------------------------------------------------------------------------------
// TWO relaxed-Jacobi sweeps per pass over memory (temporal blocking, single-domain runs).
//
// Each sweep of cz_solver.f90:334-351 is HBM bound at 12 B per update and stencil_k already moves within 5 % of the
// ideal bytes (profiles/r01), so the only way past the streaming ceiling is to apply sweep n+1 and sweep n+2 while
// the data are on chip.  Same 2.5-D march as stencil_k, two stages deep:
//     stage 1 at plane q   : v(q)   = relax(u(q-1), u(q), u(q+1))     on E1 = own segment +- one k-row (R vectors)
//     stage 2 at plane q-1 : w(q-1) = relax(v(q-2), v(q-1), v(q))     on the own segment
// u = input field (time n), v = time n+1 (never leaves the CU: registers + LDS), w = output (time n+2).
// Register queues hold u(q-1..q+1) and v(q-2..q) of the thread's vectors; LDS holds the centre planes u(q) (own
// segment +- 2 rows) and v(q-1) (own +- 1 row) for the i+-1 / k+-1 neighbours, double-buffered, one barrier per plane.
// The halo rows of v and the first/last plane of a chunk are recomputed by the neighbouring workgroups (redundant
// arithmetic, (S+2R)/S in i and (TJ+2)/TJ in j) instead of being exchanged.  Points outside the inner box pass
// through unchanged (v = u), exactly what a separate first sweep would have left in memory, and the per-point
// arithmetic is the same un-fused float sequence, so the result is bit-identical to two launches of stencil_k.
// Both residuals (sum dp^2 of sweep n+1 and of sweep n+2) are produced; each point is counted by the one workgroup
// that owns it.
// ------------------------------------------------------------------------------------------------------------
struct Geom2 {
  int R;
  long long PSV;
  int kk0, kk1, jj0, jj1;      // stage-2 (output) box = the inner box
  long long F0, Fend;
  // stage-1 box: the inner box, grown by one layer across rank-internal faces of a decomposed run (the first sweep
  // must also be applied to the ghost layer the second sweep reads; two ghost layers are exchanged per pair)
  int kk0a, kk1a, jj0a, jj1a;
  long long F0a, Fenda;
  int nseg, TJ, S;  // S = TB*MV - 2R
  int par;          // RB: colour 0 = points with (kk + ii + jj + par) even
  int zero_u;       // the input field is identically zero (a freshly cleared preconditioner vector): u is not read
};

struct Fin2 {
  double* dst = nullptr;   // [0] <- sum of sweep n+1, [1] <- sum of sweep n+2
  int do_check = 0, itr = 0;  // itr = iteration number of sweep n+1
  int single = 0;             // RB: both stages belong to ONE iteration: dst[0] = sum1 + sum2, one bookkeeping step
  const double* extra = nullptr;  // per-workgroup sums of the shell launch of a split pass (pair_shell_k): n_extra first-stage
  int n_extra = 0;                // sums followed by n_extra second-stage sums, added to this launch's own
  double res_normal = 0.0, eps = 0.0;
  double* hist = nullptr;
  int* flag = nullptr;
  int* conv_itr = nullptr;
  unsigned* counter = nullptr;
};

// bit cc set when (base + cc) is even
template <int V>
__device__ __forceinline__ unsigned colour_bits(int base) {
  const unsigned even = (V == 4) ? 0x5u : (V == 2) ? 0x1u : 0x1u;   // components 0,2 / 0 / 0
  const unsigned odd = (V == 4) ? 0xAu : (V == 2) ? 0x2u : 0x0u;    // components 1,3 / 1 / -
  return (base & 1) ? odd : even;
}

template <int V>
__device__ __forceinline__ Vec<V> relax_vec(const Vec<V>& pc, const Vec<V>& im, const Vec<V>& ip, const Vec<V>& pm,
                                            const Vec<V>& pn, REAL kl, REAL kr, const Vec<V>& bb, const Coef& c,
                                            unsigned mask, unsigned count_mask, double& acc) {
  Vec<V> o;
#pragma unroll
  for (int cc = 0; cc < V; cc++) {
    const REAL pp = pc.v[cc];
    const REAL km1 = (cc == 0) ? kl : pc.v[cc > 0 ? cc - 1 : 0];
    const REAL kp1 = (cc == V - 1) ? kr : pc.v[cc < V - 1 ? cc + 1 : V - 1];
    const REAL ss = c.c1 * ip.v[cc] + c.c2 * im.v[cc] + c.c3 * pn.v[cc] + c.c4 * pm.v[cc] + c.c5 * kp1 + c.c6 * km1;
    const REAL dp = ((ss - bb.v[cc]) / c.dd - pp) * c.omg;
    const REAL d2 = dp * dp;
    o.v[cc] = (mask & (1u << cc)) ? pp + dp : pp;
    if (count_mask & (1u << cc)) acc += (double)d2;
  }
  return o;
}

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
  if ((nblk & 7) == 0) lb = (lb & 7) * (nblk >> 3) + (lb >> 3);
  const int seg = lb % g.nseg;
  const int chunk = lb / g.nseg;
  const long long fb = g.F0 + (long long)seg * g.S;
  const int ja = g.jj0 + chunk * g.TJ;
  int jb = ja + g.TJ - 1;
  if (jb > g.jj1) jb = g.jj1;

  double acc1 = 0.0, acc2 = 0.0;

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
          vc[m] = relax_vec<V>(ub[m], im, ip, ua[m], uc[m], kl, kr, b1[m], c, msk, count1 ? (own[m] & msk) : 0u, acc1);
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
          const Vec<V> o = relax_vec<V>(vb[m], im, ip, va[m], vc[m], kl, kr, b2[m], c, m2, m2, acc2);
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
  if (*last_flag) {
    double x1 = 0.0, x2 = 0.0;
    for (int i = t; i < nblk; i += TB) {
      x1 += __hip_atomic_load(&partials[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      x2 += __hip_atomic_load(&partials[nblk + i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    for (int i = t; i < fin.n_extra; i += TB) {  // written by an earlier launch on this stream
      x1 += fin.extra[i];
      x2 += fin.extra[fin.n_extra + i];
    }
    __syncthreads();
    const double t1 = block_sum<TB>(x1, wsum);
    __syncthreads();
    const double t2 = block_sum<TB>(x2, wsum);
    if (t == 0 && fin.single) {
      const double tot = t1 + t2;  // colour 0 + colour 1 (cz_Poisson.cpp:205-209 accumulate into one res)
      fin.dst[0] = tot;
      if (fin.do_check) {
        const double r = sqrt(tot * fin.res_normal);
        fin.hist[fin.itr] = r;
        if (r < fin.eps) {
          *fin.flag = 1;
          *fin.conv_itr = fin.itr;
        }
      }
      *fin.counter = 0u;
    } else if (t == 0) {
      fin.dst[0] = t1;
      fin.dst[1] = t2;
      if (fin.do_check) {  // cz_Poisson.cpp:69-77 for iteration itr, then itr+1
        double r = sqrt(t1 * fin.res_normal);
        fin.hist[fin.itr] = r;
        if (r < fin.eps) {
          *fin.flag = 1;
          *fin.conv_itr = fin.itr;
        } else {
          r = sqrt(t2 * fin.res_normal);
          fin.hist[fin.itr + 1] = r;
          if (r < fin.eps) {
            *fin.flag = 1;
            *fin.conv_itr = fin.itr + 1;
          }
        }
      }
      *fin.counter = 0u;
    }
  }
}

// ------------------------------------------------------------------------------------------------------------
// The same two-stage update on thin boxes: the cells a decomposed brick owes its neighbours (two layers behind every
// rank-internal face).  The driver runs this first, starts the halo exchange on a second stream and lets jacobi2_k
// work on the interior meanwhile.  A slab two cells thick has no plane to march along, so it is cut into
// small 3-D tiles instead: a workgroup stages the tile of u with two halo layers in LDS, applies stage 1 to the tile
// plus one layer (LDS), then stage 2 to the tile.  Tile shapes follow the slab's orientation (long in k wherever k is
// not the thin axis, so that global accesses stay coalesced).  Same scalar operation sequence as relax_vec<1> => the
// fields are bit-identical to an unsplit jacobi2_k launch.
// ------------------------------------------------------------------------------------------------------------
struct ShellBox {
  int i0, j0, k0, ni, nj, nk;  // padded 0-based start, extent
  int kind;                    // tile shape: 0 = 64x4x2 (k,i,j; J slabs), 1 = 64x2x4 (I slabs), 2 = 2x16x16 (K slabs), 3 = 32x4x4
  int ntk, nti, ntj;           // tiles per axis
};
struct ShellTab {
  int n;
  ShellBox b[6];
  int ii0a, ii1a, jj0a, jj1a, kk0a, kk1a;  // stage-1 box of the brick (inner box grown across rank-internal faces)
  int nkp, nip, njp;
  int par;
};

// all tiles of one box that this workgroup takes; the tile shape is a compile-time constant (index arithmetic without divisions)
template <int RB, int TK, int TI, int TJ>
__device__ __forceinline__ void shell_tiles(const REAL* __restrict__ U, const REAL* __restrict__ B, REAL* __restrict__ W, const Coef& c,
                                            const ShellTab& s, const ShellBox& d, REAL* lu, double& acc1, double& acc2) {
  constexpr int UK = TK + 4, UI = TI + 4, UJ = TJ + 4;  // u tile: two halo layers
  constexpr int VK = TK + 2, VI = TI + 2, VJ = TJ + 2;  // v tile: one halo layer
  REAL* lv = lu + UK * UI * UJ;
  const int t = threadIdx.x;
  const int si = s.nkp, sj = s.nkp * s.nip;  // a halo'd tile spans < 2^31 elements: 32-bit offsets from the tile origin
  const int ntiles = d.ntk * d.nti * d.ntj;
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int tkx = tile % d.ntk, tr = tile / d.ntk;
    const int K0 = d.k0 + tkx * TK, I0 = d.i0 + (tr % d.nti) * TI, J0 = d.j0 + (tr / d.nti) * TJ;
    const int ck = min(TK, d.k0 + d.nk - K0), ci = min(TI, d.i0 + d.ni - I0), cj = min(TJ, d.j0 + d.nj - J0);  // clipped core
    const size_t org = (size_t)(K0 - 2) + (size_t)(I0 - 2) * s.nkp + (size_t)(J0 - 2) * s.nkp * s.nip;  // first cell of the u tile
    const REAL* __restrict__ Ut = U + org;
    const REAL* __restrict__ Bt = B + org;
    REAL* __restrict__ Wt = W + org;
    // ---- every global read of the tile is issued before the first use (one memory latency per tile, not one per element)
    constexpr int NU = (UK * UI * UJ + 255) / 256, NV = (VK * VI * VJ + 255) / 256, NO = (TK * TI * TJ + 255) / 256;
    REAL ru[NU], rb1[NV], rb2[NO];
#pragma unroll
    for (int n = 0; n < NU; n++) {
      const int e = t + n * 256;
      const int k = e % UK, r = e / UK, i = r % UI, j = r / UI;
      const int gk = K0 - 2 + k, gi = I0 - 2 + i, gj = J0 - 2 + j;
      ru[n] = (e < UK * UI * UJ && gk < s.nkp && gi < s.nip && gj < s.njp) ? Ut[k + i * si + j * sj] : (REAL)0;
    }
    unsigned in1 = 0;  // bit n: stage 1 applies to this thread's n-th point of the v tile
#pragma unroll
    for (int n = 0; n < NV; n++) {
      const int e = t + n * 256;
      const int k = e % VK, r = e / VK, i = r % VI, j = r / VI;
      const int gk = K0 - 1 + k, gi = I0 - 1 + i, gj = J0 - 1 + j;
      const bool inside = e < VK * VI * VJ && gi >= s.ii0a && gi <= s.ii1a && gj >= s.jj0a && gj <= s.jj1a && gk >= s.kk0a && gk <= s.kk1a &&
                          !(RB && ((gk + gi + gj + s.par) & 1));
      rb1[n] = inside ? Bt[(k + 1) + (i + 1) * si + (j + 1) * sj] : (REAL)0;
      if (inside) in1 |= 1u << n;
    }
#pragma unroll
    for (int n = 0; n < NO; n++) {
      const int e = t + n * 256;
      const int k = e % TK, r = e / TK, i = r % TI, j = r / TI;
      const bool live = e < TK * TI * TJ && k < ck && i < ci && j < cj;
      rb2[n] = live ? Bt[(k + 2) + (i + 2) * si + (j + 2) * sj] : (REAL)0;
    }
#pragma unroll
    for (int n = 0; n < NU; n++)
      if (t + n * 256 < UK * UI * UJ) lu[t + n * 256] = ru[n];
    __syncthreads();
    // ---- stage 1 on the core plus one layer
#pragma unroll
    for (int n = 0; n < NV; n++) {
      const int e = t + n * 256;
      if (e >= VK * VI * VJ) continue;
      const int k = e % VK, r = e / VK, i = r % VI, j = r / VI;
      const int cu = (k + 1) + UK * ((i + 1) + UI * (j + 1));
      REAL v = lu[cu];
      if (in1 & (1u << n)) {
        const bool core = k >= 1 && k <= ck && i >= 1 && i <= ci && j >= 1 && j <= cj;
        Vec<1> pc, im, ip, pm, pn, bb;
        pc.v[0] = v, im.v[0] = lu[cu - UK], ip.v[0] = lu[cu + UK], pm.v[0] = lu[cu - UK * UI], pn.v[0] = lu[cu + UK * UI];
        bb.v[0] = rb1[n];
        v = relax_vec<1>(pc, im, ip, pm, pn, lu[cu - 1], lu[cu + 1], bb, c, 1u, core ? 1u : 0u, acc1).v[0];
      }
      lv[e] = v;
    }
    __syncthreads();
    // ---- stage 2 on the core
#pragma unroll
    for (int n = 0; n < NO; n++) {
      const int e = t + n * 256;
      const int k = e % TK, r = e / TK, i = r % TI, j = r / TI;
      if (e >= TK * TI * TJ || k >= ck || i >= ci || j >= cj) continue;
      const int gk = K0 + k, gi = I0 + i, gj = J0 + j;
      const int cv = (k + 1) + VK * ((i + 1) + VI * (j + 1));
      REAL o = lv[cv];
      if (!(RB && !((gk + gi + gj + s.par) & 1))) {  // RB: colour 0 passes through stage 2
        Vec<1> pc, im, ip, pm, pn, bb;
        pc.v[0] = o, im.v[0] = lv[cv - VK], ip.v[0] = lv[cv + VK], pm.v[0] = lv[cv - VK * VI], pn.v[0] = lv[cv + VK * VI];
        bb.v[0] = rb2[n];
        o = relax_vec<1>(pc, im, ip, pm, pn, lv[cv - 1], lv[cv + 1], bb, c, 1u, 1u, acc2).v[0];
      }
      Wt[(k + 2) + (i + 2) * si + (j + 2) * sj] = o;
    }
    __syncthreads();
  }
}

template <int RB>
__global__ void __launch_bounds__(256)
pair_shell_k(const REAL* __restrict__ U, const REAL* __restrict__ B, REAL* __restrict__ W, Coef c, ShellTab s, double* partials,
             const int* __restrict__ skip) {
  if (skip != nullptr && *skip != 0) return;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  __shared__ double wsum[8];
  const int t = threadIdx.x;
  const ShellBox d = s.b[blockIdx.y];
  REAL* lu = reinterpret_cast<REAL*>(smem);
  double acc1 = 0.0, acc2 = 0.0;
  switch (d.kind) {  // uniform per workgroup
    case 0: shell_tiles<RB, 64, 4, 2>(U, B, W, c, s, d, lu, acc1, acc2); break;
    case 1: shell_tiles<RB, 64, 2, 4>(U, B, W, c, s, d, lu, acc1, acc2); break;
    case 2: shell_tiles<RB, 2, 16, 16>(U, B, W, c, s, d, lu, acc1, acc2); break;
    default: shell_tiles<RB, 32, 4, 4>(U, B, W, c, s, d, lu, acc1, acc2); break;
  }
  // residuals: one pair of sums per workgroup; the interior launch that follows on the stream adds them to its own
  const int nblk = gridDim.x * gridDim.y;
  const int lb = blockIdx.y * gridDim.x + blockIdx.x;
  const double s1 = block_sum<256>(acc1, wsum);
  __syncthreads();
  const double s2 = block_sum<256>(acc2, wsum);
  if (t == 0) {
    partials[lb] = s1;
    partials[nblk + lb] = s2;
  }
}

// ------------------------------------------------------------------------------------------------------------
// Line SOR by parallel cyclic reduction, pcr_rb (cz_solver.f90:497-662).
// One wave64 per (i,j) column of the active checkerboard colour, four columns per workgroup.  The column's tridiagonal
// system along k lives in LDS (a, c, d and their successors a1, c1, d1, ping-ponged instead of copied back); lanes
// stride over k, so every global access is coalesced along the unit-stride axis.  pn-1 reduction stages of stride
// 2^(p-1), then the 2x2 systems of stride 2^(pn-1), then the relaxation -- operation for operation the reference's
// arithmetic (serial build: entries outside kst..ked are zero, kept in two pad slots).  sum dp^2 is accumulated in double.
// ------------------------------------------------------------------------------------------------------------
struct PcrGeom {
  int nkp, nip;                 // padded extents
  int kk0, n;                   // padded index of kst, number of unknowns per column
  int ii0, ni, jj0, nj;         // inner (i,j) range in padded indices / counts
  int ist1, jst1;               // 1-based ist, jst (colour rule mod(i+j,2) == color uses 1-based indices)
  int pn, color;
  int nhalf;                    // columns of one colour per j row (upper bound)
};

template <int NW>
__global__ void __launch_bounds__(64 * NW)
pcr_rb_k(REAL* X, const REAL* __restrict__ MSK, const REAL* __restrict__ RHS, PcrGeom g, REAL omg, double* partials,
         double* dst, int accumulate, unsigned* counter) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int n = g.n, LD = n + 2;  // slot 0 and slot n+1 are the zero entries k = kst-1 / ked+1
  REAL* base = reinterpret_cast<REAL*>(smem) + (size_t)wave * 6 * LD;
  REAL* A[2] = {base, base + 3 * LD};          // [buf][a | c | d]
  double* wsum = reinterpret_cast<double*>(reinterpret_cast<REAL*>(smem) + (size_t)NW * 6 * LD + 4);
  wsum = reinterpret_cast<double*>((reinterpret_cast<size_t>(wsum) + 15) & ~(size_t)15);

  const long long col = (long long)blockIdx.x * NW + wave;  // column ordinal among the colour's columns
  const int jrow = (int)(col / g.nhalf), ih = (int)(col % g.nhalf);
  bool active = jrow < g.nj;
  int ii = 0, jj = 0;
  if (active) {
    const int j1 = g.jst1 + jrow;
    int i1 = g.ist1 + 2 * ih;
    if (((i1 + j1) & 1) != g.color) i1 += 1;   // first i of this colour in the row
    active = (i1 - g.ist1) < g.ni;
    ii = g.ii0 + (i1 - g.ist1);
    jj = g.jj0 + jrow;
  }
  const REAL r = (REAL)1.0 / (REAL)6.0;
  const size_t rowlen = (size_t)g.nkp, plane = (size_t)g.nkp * g.nip;
  const size_t c0 = (size_t)g.kk0 + (size_t)ii * rowlen + (size_t)jj * plane;  // element (kst, i, j)
  double acc = 0.0;

  // ---- set-up: coefficients and source term (:545-568)
  if (active) {
    REAL* a = A[0];
    REAL* c = a + LD;
    REAL* d = c + LD;
    if (lane == 0) {
      for (int b = 0; b < 2; b++)
        for (int v = 0; v < 3; v++) A[b][v * LD] = (REAL)0, A[b][v * LD + n + 1] = (REAL)0;
    }
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
  __syncthreads();
  // ---- PCR stages (:572-595)
  int cur = 0;
  for (int p = 1; p <= g.pn - 1; p++) {
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
    __syncthreads();
    cur ^= 1;
  }
  // ---- 2x2 systems of the last stage (:599-616), result into the d slot of the other buffer
  {
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
    __syncthreads();
  }
  // ---- relaxation (:626-633)
  if (active) {
    const REAL* d1 = A[cur ^ 1] + 2 * LD;
    for (int k = lane; k < n; k += 64) {
      const size_t e = c0 + k;
      const REAL pp = X[e];
      const REAL dp = (d1[k + 1] - pp) * omg * MSK[e];
      X[e] = pp + dp;
      const REAL d2 = dp * dp;
      acc += (double)d2;
    }
  }
  // ---- residual: partial per workgroup, fixed-order sum by the last one (write-through hand-off as in stencil_k)
  __syncthreads();
  const double sblk = block_sum<64 * NW>(acc, wsum);
  int* last_flag = reinterpret_cast<int*>(wsum + 8);
  const int nblk = gridDim.x;
  if (threadIdx.x == 0) {
    __hip_atomic_store(&partials[blockIdx.x], sblk, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned ticket = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    *last_flag = (ticket == (unsigned)nblk - 1u);
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
        lp[x] = relax_vec<1>(pc, im, ip, pm, pn, lp[x - 1], lp[x + 1], bv, c, 1u, 1u, acc).v[0];
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

__device__ __forceinline__ void wave_lds_sync() {
  // the lanes of ONE wave hand data to each other through LDS: order the accesses, no workgroup barrier
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// ORDER selects the columns of one launch (the line-SOR variants of cz_solver.f90 differ in the column order):
//   0  one checkerboard colour, in place          pcr_rb (:540), pcr_rb_esa (:1324)           g.color = colour
//   1  one diagonal (i-ist)+(j-jst) = g.color of the lexicographic order, in place: a column of pcr (:718-719) / pcr_esa sees
//      the new values of its i-1 and j-1 neighbours, which lie on the diagonal before => diagonals in sequence, columns of one in parallel
//   2  all columns from the old field, result into WOUT (pcr_j_esa :1553-1632; the caller copies back, :1655-1663)
template <int NW, int L, int FINAL4, int ORDER>
__global__ void __launch_bounds__(64 * NW)
pcr_rb2_k(REAL* X, REAL* WOUT, const REAL* __restrict__ MSK, const REAL* __restrict__ RHS, PcrGeom g, REAL omg,
          const REAL* __restrict__ tab, int tab_len, int nfin, double* partials, double* dst, int accumulate, unsigned* counter) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int n = g.n, LD = n + 2;  // slot 0 and slot n+1 are the zero entries k = kst-1 / ked+1
  REAL* T = reinterpret_cast<REAL*>(smem);
  REAL* D = T + tab_len + (size_t)wave * 2 * L * LD;  // [buf][line][LD]
  double* wsum = reinterpret_cast<double*>(T + tab_len + (size_t)NW * 2 * L * LD + 4);
  wsum = reinterpret_cast<double*>((reinterpret_cast<size_t>(wsum) + 15) & ~(size_t)15);

  for (int i = threadIdx.x; i < tab_len; i += 64 * NW) T[i] = tab[i];
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
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned ticket = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    *last_flag = (ticket == (unsigned)nblk - 1u);
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

template <int M, int NW, int L, int FINAL4, int ORDER>
__global__ void __launch_bounds__(64 * NW)
pcr_line_reg_k(REAL* X, REAL* WOUT, const REAL* __restrict__ MSK, const REAL* __restrict__ RHS, PcrGeom g, REAL omg,
               const REAL* __restrict__ tab, int tab_len, double* partials, double* dst, int accumulate, unsigned* counter) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int NE = 64 * M;
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
      REAL sol[L][M];
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
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned ticket = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    *last_flag = (ticket == (unsigned)nblk - 1u);
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

// cz_Poisson.cpp:67-77 on the device
__global__ void check_k(const double* res_dev, double res_normal, double eps, int itr, double* hist, int* flag,
                        int* conv_itr) {
  if (*flag != 0) return;
  double r = res_dev[0];
  r *= res_normal;
  r = sqrt(r);
  hist[itr] = r;
  if (r < eps) {
    *flag = 1;
    *conv_itr = itr;
  }
}

// the same bookkeeping for a fused pair (iterations itr, itr+1) whose two sums were all-reduced first
__global__ void check2_k(const double* res_dev, double res_normal, double eps, int itr, double* hist, int* flag,
                         int* conv_itr) {
  if (*flag != 0) return;
  double r = sqrt(res_dev[0] * res_normal);
  hist[itr] = r;
  if (r < eps) {
    *flag = 1;
    *conv_itr = itr;
    return;
  }
  r = sqrt(res_dev[1] * res_normal);
  hist[itr + 1] = r;
  if (r < eps) {
    *flag = 1;
    *conv_itr = itr + 1;
  }
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
};

template <int V, int OP>
__global__ void __launch_bounds__(256)
ewise_k(REAL* Z, const REAL* X, const REAL* Y, REAL a, REAL b, EGeom g) {
  const long long f = g.F0 + (long long)blockIdx.x * 256 + threadIdx.x;
  if (f >= g.Fend) return;
  const long long pv = (long long)(g.jj0 + blockIdx.y) * g.PSV + f;
  const int kv = (int)(f % g.R);
  unsigned mk = 0;
#pragma unroll
  for (int cc = 0; cc < V; cc++) {
    const int kk = kv * V + cc;
    if (kk >= g.kk0 && kk <= g.kk1) mk |= 1u << cc;
  }
  if (mk == 0) return;
  Vec<V> x = ldv<V>(X, pv), y, z, o;
  if (OP != OP_COPY) y = ldv<V>(Y, pv);
  if (OP == OP_BICG1 || OP == OP_BICG2) z = ldv<V>(Z, pv);
#pragma unroll
  for (int cc = 0; cc < V; cc++) {
    if (OP == OP_TRIAD) o.v[cc] = a * x.v[cc] + y.v[cc];                              // cz_blas.f90:297
    if (OP == OP_BICG1) o.v[cc] = x.v[cc] + a * (z.v[cc] - b * y.v[cc]);              // :490  p = r + beta*(p - omg*q)
    if (OP == OP_BICG2) o.v[cc] = a * x.v[cc] + b * y.v[cc] + z.v[cc];                // :554
    if (OP == OP_COPY) o.v[cc] = x.v[cc];
  }
  if (mk == (1u << V) - 1) {
    stv<V>(Z, pv, o);
  } else {
#pragma unroll
    for (int cc = 0; cc < V; cc++)
      if (mk & (1u << cc)) Z[pv * V + cc] = o.v[cc];
  }
}

// search_pivot (cz_blas.f90:947-1039): pvt = 1 / max(|row entries|) on the inner box
template <int V>
__global__ void __launch_bounds__(256)
pivot_k(REAL* PVT, EGeom g, MafArgs ma, int nkp, int nip) {
  const long long f = g.F0 + (long long)blockIdx.x * 256 + threadIdx.x;
  if (f >= g.Fend) return;
  const int jj = g.jj0 + blockIdx.y;
  const long long pv = (long long)jj * g.PSV + f;
  const long long row = f / g.R;
  const int kv = (int)(f - row * g.R);
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
    PVT[pv * V + cc] = (REAL)1.0 / ss;
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
    for (int pl = blockIdx.y; pl < nplanes; pl += gridDim.y) {
      const long long pv = (long long)(g.jj0 + pl) * g.PSV + f;
      const Vec<V> x = ldv<V>(X, pv);
      Vec<V> y = x;
      if (TWO) y = ldv<V>(Y, pv);
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
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned ticket = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    last_flag = (ticket == (unsigned)nblk - 1u);
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
      for (int pl = blockIdx.y; pl < nplanes; pl += gridDim.y) {
        const long long pv = (long long)(g.jj0 + pl) * g.PSV + f;
        const Vec<V> x = ldv<V>(X, pv), y = ldv<V>(Y, pv), w = ldv<V>(W, pv);
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
          stv<V>(Z, pv, o);
        } else {
#pragma unroll
          for (int cc = 0; cc < V; cc++)
            if (mk & (1u << cc)) Z[pv * V + cc] = o.v[cc];
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
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned ticket = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    last_flag = (ticket == (unsigned)nblk - 1u);
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

// ------------------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------------------
struct Tuning {
  int threads = 512, m = 2, tj = 16 /* 0 = auto */, pf = 0;  // best of tools/tune_jacobi.py at 512^3 FP32
  int fuse_fin = 1;
  int pcr_fast = 2, pcr_variant = 0;  // CZHIP_PCR=fast[,variant]: fast 0 = the per-line kernel (pcr_rb_k), 1 = table in LDS + d in LDS
                                      // (pcr_rb2_k; variant = NW*10+L), 2 = table in LDS + d in registers (pcr_line_reg_k)
  #ifdef CZ_REAL_IS_DOUBLE
  int t2_threads = 1024, t2_mv = 2, t2_tj = 64;  // best of tools/tune_jacobi2.py at 512^3 FP64 (profiles/r01)
#else
  int t2_threads = 512, t2_mv = 2, t2_tj = 16;   // best at 512^3 FP32
#endif  // two-sweep kernel: threads, vectors/thread, planes/chunk
  int use_t2 = 1;                                 // driver may fuse pairs of Jacobi sweeps (single-domain runs)  // 1: residual finalised by the last workgroup of the sweep; 0: separate reduce(+check) launches
};

struct Ctx {
  bool ready = false;
  int device = 0;
  hipStream_t stream = nullptr;
  double* partials = nullptr;   // device
  REAL* pcr_tab_perm = nullptr; // the same table in the [m][lane] order of pcr_line_reg_k, for pcr_perm_M entries per lane
  int pcr_perm_M = 0;
  size_t pcr_perm_cap = 0;
  REAL* pcr_tab = nullptr;      // pcr_coef_k's table for lines of pcr_tab_n unknowns (pcr_tab_pn stages)
  int pcr_tab_n = 0, pcr_tab_pn = 0, pcr_tab_final4 = -1;
  size_t pcr_tab_cap = 0;
  double* shell_partials = nullptr;  // per-workgroup sums of the last pair_shell_k launch, folded in by the interior launch
  int shell_pending = 0;
  size_t partials_cap = 0;
  unsigned* counter = nullptr;  // arrival ticket of the in-kernel finalisation
  REAL* xyz = nullptr;           // device copies of the host coordinate arrays X, Y, Z handed to the *_maf_ drop-in symbols
  size_t xyz_cap = 0;
  double* scal_dev = nullptr;   // a few device doubles for the synchronous entry points
  double* scal_host = nullptr;  // pinned
  Tuning tune;
  std::map<std::vector<double>, REAL*> bc_tabs;  // key: ix, jx, dh, org0, org1
  int num_cu = 256;
  // optional per-launch HIP-event timing of the labelled kernels (bench.py roofline leg)
  bool timing = false;
  struct Ev { hipEvent_t a, b; int label; };
  std::vector<Ev> ev_used, ev_free;
  double t_acc[16] = {0};       // per label: time [ms] and launches of events already folded (long runs)
  long long t_cnt[16] = {0};
};
thread_local Ctx ctx;  // one context per host thread (= per rank; LOCAL transport runs ranks as threads)

enum { LBL_JACOBI = 0, LBL_RBSOR, LBL_AX, LBL_RK, LBL_REDUCE, LBL_EWISE, LBL_DOT, LBL_JACOBI2, LBL_RBSOR2, LBL_PCR, LBL_SHELL, LBL_PSOR, LBL_COUNT };
static_assert(LBL_COUNT <= 16, "Ctx::t_acc / t_cnt hold 16 labels");
const char* const kLabelNames[LBL_COUNT] = {"jacobi", "rbsor", "calc_ax", "calc_rk", "reduce", "ewise", "dot", "jacobi2", "rbsor2", "pcr_rb", "pair_shell", "psor"};

struct ScopedTimer {
  bool on;
  Ctx::Ev ev;
  explicit ScopedTimer(int label) : on(ctx.timing) {
    if (!on) return;
    if (!ctx.ev_free.empty()) {
      ev = ctx.ev_free.back();
      ctx.ev_free.pop_back();
    } else {
      HIP_CHECK(hipEventCreate(&ev.a));
      HIP_CHECK(hipEventCreate(&ev.b));
    }
    ev.label = label;
    HIP_CHECK(hipEventRecord(ev.a, ctx.stream));
  }
  ~ScopedTimer() {
    if (!on) return;
    HIP_CHECK(hipEventRecord(ev.b, ctx.stream));
    ctx.ev_used.push_back(ev);
    if (ctx.ev_used.size() >= 4096) fold_events(2048);
  }
  // long runs: turn the oldest recorded pairs into per-label sums and recycle their events (they completed long ago)
  static void fold_events(size_t n) {
    for (size_t i = 0; i < n; i++) {
      Ctx::Ev& e = ctx.ev_used[i];
      HIP_CHECK(hipEventSynchronize(e.b));
      float ms = 0.f;
      HIP_CHECK(hipEventElapsedTime(&ms, e.a, e.b));
      ctx.t_acc[e.label] += ms;
      ctx.t_cnt[e.label]++;
      ctx.ev_free.push_back(e);
    }
    ctx.ev_used.erase(ctx.ev_used.begin(), ctx.ev_used.begin() + n);
  }
};

void ensure_init() {
  if (!ctx.ready) czhip_init(-1);
}

void ensure_partials(size_t n) {
  if (n <= ctx.partials_cap) return;
  if (ctx.partials) {
    HIP_CHECK(hipStreamSynchronize(ctx.stream));
    HIP_CHECK(hipFree(ctx.partials));
  }
  size_t cap = n < 65536 ? 65536 : n;
  HIP_CHECK(hipMalloc(&ctx.partials, cap * sizeof(double)));
  ctx.partials_cap = cap;
}

struct Box {
  int ni, nj, nk, g, nkp, nip, njp;
  int kk0, kk1, ii0, ii1, jj0, jj1;
  bool empty;
};

Box make_box(const int* sz, const int* idx, int g) {
  Box b;
  b.ni = sz[0], b.nj = sz[1], b.nk = sz[2], b.g = g;
  b.nkp = b.nk + 2 * g, b.nip = b.ni + 2 * g, b.njp = b.nj + 2 * g;
  // 1-based inclusive (ist,ied,jst,jed,kst,ked) -> padded 0-based
  b.ii0 = idx[0] + g - 1, b.ii1 = idx[1] + g - 1;
  b.jj0 = idx[2] + g - 1, b.jj1 = idx[3] + g - 1;
  b.kk0 = idx[4] + g - 1, b.kk1 = idx[5] + g - 1;
  b.empty = b.ii1 < b.ii0 || b.jj1 < b.jj0 || b.kk1 < b.kk0;
  // the 7-point stencil reads one layer around the box: it must exist inside the padded array
  if (!b.empty && (g < 1 || b.ii0 < 1 || b.jj0 < 1 || b.kk0 < 1 || b.ii1 > b.nip - 2 || b.jj1 > b.njp - 2 || b.kk1 > b.nkp - 2)) {
    fprintf(stderr, "czhip: index range (%d..%d, %d..%d, %d..%d) does not fit sz=(%d,%d,%d) g=%d\n", idx[0], idx[1], idx[2],
            idx[3], idx[4], idx[5], sz[0], sz[1], sz[2], g);
    exit(1);
  }
  return b;
}

inline bool vec_ok(const Box& b, std::initializer_list<const void*> ptrs) {
  if (b.nkp % VW != 0) return false;
  for (const void* p : ptrs)
    if (p && (reinterpret_cast<uintptr_t>(p) & 15u)) return false;
  return true;
}

template <int V>
EGeom make_egeom(const Box& b) {
  EGeom e;
  e.R = b.nkp / V;
  e.PSV = (long long)e.R * b.nip;
  e.kk0 = b.kk0, e.kk1 = b.kk1, e.jj0 = b.jj0;
  e.F0 = (long long)b.ii0 * e.R;
  e.Fend = (long long)(b.ii1 + 1) * e.R;
  return e;
}

template <int V, int TB, int M, int PF, int MODE, int MAF = 0>
void launch_stencil_inst(const REAL* P, const REAL* B, REAL* OUT, const Coef& c, const Box& b, int par, int tj_req,
                         const int* skip, int* nblk_out, const Fin& fin, const MafArgs& ma = MafArgs()) {
  Geom g;
  g.nip = b.nip;
  g.R = b.nkp / V;
  g.PSV = (long long)g.R * b.nip;
  g.kk0 = b.kk0, g.kk1 = b.kk1, g.jj0 = b.jj0, g.jj1 = b.jj1;
  g.F0 = (long long)b.ii0 * g.R;
  g.Fend = (long long)(b.ii1 + 1) * g.R;
  g.S = TB * M;
  const long long nf = g.Fend - g.F0;
  g.nseg = (int)((nf + g.S - 1) / g.S);
  const int nplanes = b.jj1 - b.jj0 + 1;
  int tj = tj_req;
  if (tj <= 0) {
    // auto: enough workgroups to fill the chip a few times over, chunk count a multiple of 8 (XCD remap)
    const int waves_per_wg = TB / 64;
    const long long want = (long long)ctx.num_cu * 32 / waves_per_wg;  // one full residency of waves
    int nchunk = (int)((want + g.nseg - 1) / g.nseg);
    nchunk = ((nchunk + 7) / 8) * 8;
    if (nchunk > nplanes) nchunk = nplanes;
    if (nchunk < 1) nchunk = 1;
    tj = (nplanes + nchunk - 1) / nchunk;
  }
  if (tj > nplanes) tj = nplanes;
  g.TJ = tj;
  int nchunk = (nplanes + tj - 1) / tj;
  // pad the chunk count to a multiple of 8 when that costs nothing but empty workgroups (keeps the remap on)
  if (((long long)nchunk * g.nseg) % 8 != 0 && nchunk >= 8) nchunk = ((nchunk + 7) / 8) * 8;
  const long long nblk = (long long)nchunk * g.nseg;
  const size_t lds = (size_t)2 * (g.S + 2 * g.R) * sizeof(Vec<V>) + 18 * sizeof(double);
  if (lds > 160 * 1024) {
    fprintf(stderr, "czhip: k-row of %d elements needs %zu bytes of LDS (>160 KiB)\n", b.nkp, lds);
    exit(1);
  }
  if (MODE == MODE_JACOBI || MODE == MODE_RB || MODE == MODE_AX) ensure_partials((size_t)2 * nblk);
  static bool attr_set = false;
  if (!attr_set) {
    HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&stencil_k<V, TB, M, PF, MODE, MAF>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_set = true;
  }
  {
    ScopedTimer tm(MODE == MODE_JACOBI ? LBL_JACOBI : MODE == MODE_RB ? LBL_RBSOR : MODE == MODE_AX ? LBL_AX : LBL_RK);
    hipLaunchKernelGGL((stencil_k<V, TB, M, PF, MODE, MAF>), dim3((unsigned)nblk), dim3(TB), lds, ctx.stream, P, B, OUT, c, g, par,
                       ctx.partials, skip, fin, ma);
  }
  HIP_CHECK(hipGetLastError());
  if (nblk_out) *nblk_out = (int)nblk;
}

template <int MODE>
void launch_stencil(const REAL* P, const REAL* B, REAL* OUT, const Coef& c, const Box& b, int par, const int* skip,
                    int* nblk_out, const Fin& fin = Fin()) {
  const Tuning& tu = ctx.tune;
  if (!vec_ok(b, {P, B, OUT})) {
    launch_stencil_inst<1, 256, 2, 0, MODE>(P, B, OUT, c, b, par, tu.tj, skip, nblk_out, fin);
    return;
  }
#define CZ_INST(TB_, M_, PF_)                                                                      \
  if (tu.threads == TB_ && tu.m == M_ && tu.pf == PF_) {                                           \
    launch_stencil_inst<VW, TB_, M_, PF_, MODE>(P, B, OUT, c, b, par, tu.tj, skip, nblk_out, fin); \
    return;                                                                                        \
  }
  if (MODE == MODE_JACOBI || MODE == MODE_RB) {
    CZ_INST(256, 1, 0) CZ_INST(256, 1, 1) CZ_INST(256, 2, 0) CZ_INST(256, 2, 1) CZ_INST(256, 4, 0) CZ_INST(256, 4, 1)
    CZ_INST(512, 1, 0) CZ_INST(512, 1, 1) CZ_INST(512, 2, 0) CZ_INST(512, 2, 1) CZ_INST(512, 4, 0) CZ_INST(512, 4, 1)
    CZ_INST(1024, 1, 0) CZ_INST(1024, 1, 1) CZ_INST(1024, 2, 0) CZ_INST(1024, 2, 1)
  }
#undef CZ_INST
  launch_stencil_inst<VW, 512, 2, 0, MODE>(P, B, OUT, c, b, par, tu.tj, skip, nblk_out, fin);
}

// MAF flavour: one tuned shape (and the scalar fallback); coordinates / pvt are device pointers
template <int MODE>
void launch_stencil_maf(const REAL* P, const REAL* B, REAL* OUT, REAL omg, const Box& b, int par, const int* skip, int* nblk_out,
                        const Fin& fin, const MafArgs& ma) {
  if (b.g != 2) {
    fprintf(stderr, "czhip: the MAF kernels assume GUIDE = 2 (X(-1:sz+2), cz_maf.f90:146-148)\n");
    exit(1);
  }
  Coef c;
  c.c1 = c.c2 = c.c3 = c.c4 = c.c5 = c.c6 = c.dd = (REAL)0;
  c.omg = omg;
  if (vec_ok(b, {P, B, OUT, ma.pvt}))
    launch_stencil_inst<VW, 512, 2, 0, MODE, 1>(P, B, OUT, c, b, par, ctx.tune.tj, skip, nblk_out, fin, ma);
  else
    launch_stencil_inst<1, 256, 2, 0, MODE, 1>(P, B, OUT, c, b, par, ctx.tune.tj, skip, nblk_out, fin, ma);
}

void reduce_partials(int n, double* dst, int accumulate, const int* skip) {
  ScopedTimer tm(LBL_REDUCE);
  hipLaunchKernelGGL(reduce_partials_k, dim3(1), dim3(1024), 0, ctx.stream, ctx.partials, n, dst, accumulate, skip);
  HIP_CHECK(hipGetLastError());
}


// two fused sweeps; returns false when the geometry does not suit the kernel (caller falls back to two stencil_k launches)
template <int TB, int MV, int RB>
bool launch_jacobi2_inst(const REAL* U, const REAL* B, REAL* W, const Coef& c, const Box& b, const Box& ba, int tj_req,
                         const int* skip, const Fin2& fin_in, int par, int zero_u, bool probe) {
  constexpr int V = VW;
  Geom2 g;
  g.R = b.nkp / V;
  if (2 * g.R >= TB * MV / 2 || g.R > TB) return false;  // halo rows would dominate / do not fit the loader
  g.PSV = (long long)g.R * b.nip;
  g.kk0 = b.kk0, g.kk1 = b.kk1, g.jj0 = b.jj0, g.jj1 = b.jj1;
  g.F0 = (long long)b.ii0 * g.R;
  g.Fend = (long long)(b.ii1 + 1) * g.R;
  g.kk0a = ba.kk0, g.kk1a = ba.kk1, g.jj0a = ba.jj0, g.jj1a = ba.jj1;
  g.F0a = (long long)ba.ii0 * g.R;
  g.Fenda = (long long)(ba.ii1 + 1) * g.R;
  g.S = TB * MV - 2 * g.R;
  g.par = par;
  g.zero_u = zero_u;
  const long long nf = g.Fend - g.F0;
  g.nseg = (int)((nf + g.S - 1) / g.S);
  const int nplanes = b.jj1 - b.jj0 + 1;
  int tj = tj_req;
  if (tj <= 0) tj = 16;
  if (tj > nplanes) tj = nplanes;
  g.TJ = tj;
  int nchunk = (nplanes + tj - 1) / tj;
  if (((long long)nchunk * g.nseg) % 8 != 0 && nchunk >= 8) nchunk = ((nchunk + 7) / 8) * 8;
  const long long nblk = (long long)nchunk * g.nseg;
  const size_t lds = (size_t)2 * ((g.S + 4 * g.R) + (g.S + 2 * g.R)) * sizeof(Vec<V>) + 18 * sizeof(double);
  if (lds > 160 * 1024) return false;
  if (probe) return true;
  ensure_partials((size_t)2 * nblk);
  static bool attr_set = false;
  if (!attr_set) {
    HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&jacobi2_k<V, TB, MV, RB>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  160 * 1024));
    attr_set = true;
  }
  Fin2 fin = fin_in;
  fin.counter = ctx.counter;
  {
    ScopedTimer tm(RB ? LBL_RBSOR2 : LBL_JACOBI2);
    hipLaunchKernelGGL((jacobi2_k<V, TB, MV, RB>), dim3((unsigned)nblk), dim3(TB), lds, ctx.stream, U, B, W, c, g, ctx.partials, skip, fin);
  }
  HIP_CHECK(hipGetLastError());
  return true;
}

template <int RB>
bool launch_jacobi2(const REAL* U, const REAL* B, REAL* W, const Coef& c, const Box& b, const Box& ba, const int* skip,
                    const Fin2& fin, int par = 0, int zero_u = 0, bool probe = false) {
  if (!vec_ok(b, {U, B, W})) return false;
  // the stage-1 box may exceed the output box by at most one layer per side
  if (ba.ii0 < b.ii0 - 1 || ba.ii0 > b.ii0 || ba.ii1 > b.ii1 + 1 || ba.ii1 < b.ii1 || ba.jj0 < b.jj0 - 1 || ba.jj0 > b.jj0 ||
      ba.jj1 > b.jj1 + 1 || ba.jj1 < b.jj1 || ba.kk0 < b.kk0 - 1 || ba.kk0 > b.kk0 || ba.kk1 > b.kk1 + 1 || ba.kk1 < b.kk1)
    return false;
  // the two-stage march reads two layers around the box
  if (b.ii0 < 2 || b.jj0 < 2 || b.ii1 > b.nip - 3 || b.jj1 > b.njp - 3) return false;
  const Tuning& tu = ctx.tune;
#define CZ_INST2(TB_, MV_) \
  if (tu.t2_threads == TB_ && tu.t2_mv == MV_) return launch_jacobi2_inst<TB_, MV_, RB>(U, B, W, c, b, ba, tu.t2_tj, skip, fin, par, zero_u, probe);
  CZ_INST2(256, 4) CZ_INST2(512, 2) CZ_INST2(512, 3) CZ_INST2(1024, 2)
#undef CZ_INST2
  return launch_jacobi2_inst<512, 2, RB>(U, B, W, c, b, ba, tu.t2_tj, skip, fin, par, zero_u, probe);
}

// the shell boxes of a decomposed brick, all in one launch (pair_shell_k); boxes: n x (ist,ied,jst,jed,kst,ked), 1-based
template <int RB>
void launch_pair_shell(const REAL* U, const REAL* B, REAL* W, const Coef& c, const int* sz, int g, const Box& ba, const int* boxes, int n,
                       int par, const int* skip) {
  ShellTab s;
  s.n = n;
  int most_tiles = 0;
  size_t lds = 0;
  for (int m = 0; m < n; m++) {
    const Box b = make_box(sz, boxes + 6 * m, g);
    if (b.empty || b.ii0 < 2 || b.jj0 < 2 || b.kk0 < 2 || b.ii1 > b.nip - 3 || b.jj1 > b.njp - 3 || b.kk1 > b.nkp - 3) {
      fprintf(stderr, "czhip: pair_shell: box %d is empty or closer than two cells to the array edge\n", m);
      exit(1);
    }
    ShellBox& d = s.b[m];
    d.i0 = b.ii0, d.j0 = b.jj0, d.k0 = b.kk0;
    d.ni = b.ii1 - b.ii0 + 1, d.nj = b.jj1 - b.jj0 + 1, d.nk = b.kk1 - b.kk0 + 1;
    // tile shape by orientation: long in k (coalesced rows) unless k is the thin axis
    int tk, ti, tj;
    if (d.nk == 2) d.kind = 2, tk = 2, ti = 16, tj = 16;
    else if (d.nj == 2) d.kind = 0, tk = 64, ti = 4, tj = 2;
    else if (d.ni == 2) d.kind = 1, tk = 64, ti = 2, tj = 4;
    else d.kind = 3, tk = 32, ti = 4, tj = 4;
    d.ntk = (d.nk + tk - 1) / tk, d.nti = (d.ni + ti - 1) / ti, d.ntj = (d.nj + tj - 1) / tj;
    most_tiles = std::max(most_tiles, d.ntk * d.nti * d.ntj);
    lds = std::max(lds, sizeof(REAL) * ((size_t)(tk + 4) * (ti + 4) * (tj + 4) + (size_t)(tk + 2) * (ti + 2) * (tj + 2)));
  }
  s.ii0a = ba.ii0, s.ii1a = ba.ii1, s.jj0a = ba.jj0, s.jj1a = ba.jj1, s.kk0a = ba.kk0, s.kk1a = ba.kk1;
  s.nkp = ba.nkp, s.nip = ba.nip, s.njp = ba.njp;
  s.par = par;
  const unsigned gx = (unsigned)std::min(most_tiles, 2048);
  {
    ScopedTimer tm(LBL_SHELL);
    hipLaunchKernelGGL((pair_shell_k<RB>), dim3(gx, (unsigned)n), dim3(256), lds, ctx.stream, U, B, W, c, s, ctx.shell_partials, skip);
  }
  HIP_CHECK(hipGetLastError());
  ctx.shell_pending = (int)(gx * n);
}

Coef make_coef_omg(REAL omg) {
  Coef c;
  c.c1 = c.c2 = c.c3 = c.c4 = c.c5 = c.c6 = c.dd = (REAL)0;
  c.omg = omg;
  return c;
}

Coef make_coef(const REAL* cf, REAL omg) {
  Coef c;
  c.c1 = cf[0], c.c2 = cf[1], c.c3 = cf[2], c.c4 = cf[3], c.c5 = cf[4], c.c6 = cf[5], c.dd = cf[6], c.omg = omg;
  return c;
}

template <int OP>
void launch_ewise(REAL* Z, const REAL* X, const REAL* Y, REAL a, REAL bcoef, const Box& b) {
  if (b.empty) return;
  ScopedTimer tm(LBL_EWISE);
  const int nplanes = b.jj1 - b.jj0 + 1;
  if (vec_ok(b, {Z, X, Y})) {
    EGeom e = make_egeom<VW>(b);
    dim3 grid((unsigned)((e.Fend - e.F0 + 255) / 256), (unsigned)nplanes);
    hipLaunchKernelGGL((ewise_k<VW, OP>), grid, dim3(256), 0, ctx.stream, Z, X, Y, a, bcoef, e);
  } else {
    EGeom e = make_egeom<1>(b);
    dim3 grid((unsigned)((e.Fend - e.F0 + 255) / 256), (unsigned)nplanes);
    hipLaunchKernelGGL((ewise_k<1, OP>), grid, dim3(256), 0, ctx.stream, Z, X, Y, a, bcoef, e);
  }
  HIP_CHECK(hipGetLastError());
}

// dot -> device double dst[0]
template <int TWO>
void launch_dot(const REAL* X, const REAL* Y, const Box& b, double* dst) {
  if (b.empty) {
    HIP_CHECK(hipMemsetAsync(dst, 0, sizeof(double), ctx.stream));
    return;
  }
  const int nplanes = b.jj1 - b.jj0 + 1;
  ScopedTimer tm(LBL_DOT);
  if (vec_ok(b, {X, Y})) {
    EGeom e = make_egeom<VW>(b);
    const unsigned gx = (unsigned)((e.Fend - e.F0 + 255) / 256);
    const unsigned gy = (unsigned)std::max(1, std::min(nplanes, (int)(4096 / gx)));
    ensure_partials((size_t)gx * gy);
    hipLaunchKernelGGL((dot_k<VW, TWO>), dim3(gx, gy), dim3(256), 0, ctx.stream, X, Y, e, nplanes, ctx.partials, dst, ctx.counter);
  } else {
    EGeom e = make_egeom<1>(b);
    const unsigned gx = (unsigned)((e.Fend - e.F0 + 255) / 256);
    const unsigned gy = (unsigned)std::max(1, std::min(nplanes, (int)(4096 / gx)));
    ensure_partials((size_t)gx * gy);
    hipLaunchKernelGGL((dot_k<1, TWO>), dim3(gx, gy), dim3(256), 0, ctx.stream, X, Y, e, nplanes, ctx.partials, dst, ctx.counter);
  }
  HIP_CHECK(hipGetLastError());
}

double read_scalar(int slot) {
  HIP_CHECK(hipMemcpyAsync(ctx.scal_host + slot, ctx.scal_dev + slot, sizeof(double), hipMemcpyDeviceToHost, ctx.stream));
  HIP_CHECK(hipStreamSynchronize(ctx.stream));
  return ctx.scal_host[slot];
}

inline double npts(const int* idx) {
  return (double)(idx[1] - idx[0] + 1) * (double)(idx[3] - idx[2] + 1) * (double)(idx[5] - idx[4] + 1);
}

// Host evaluation of the Dirichlet table sin(pi*x)*sin(pi*y), cz_solver.f90:36,52-58.
// ioff/joff: brick offset in global cells (head-1).  The reference evaluates x = org + dh*real(i-1) with the BRICK
// origin org = G_origin + (head-1)*dh (cz_Evaluate.cpp:136-138), which rounds differently from the single-domain
// x = G_origin + dh*real(i_global-1); the driver passes the global origin plus an integer offset instead so that a
// decomposed run carries bit-identical Dirichlet data (ioff = joff = 0 reproduces the reference expression exactly).
REAL* bc_table(int ix, int jx, REAL dh, const REAL* org, int ioff = 0, int joff = 0) {
  std::vector<double> key = {(double)ix, (double)jx, (double)dh, (double)org[0], (double)org[1], (double)ioff, (double)joff};
  auto it = ctx.bc_tabs.find(key);
  if (it != ctx.bc_tabs.end()) return it->second;
  std::vector<REAL> tab((size_t)ix * jx);
  volatile REAL one = (REAL)1.0;  // keep asin() a run-time libm call like the rest
#ifdef CZ_REAL_IS_DOUBLE
  const REAL pi = 2.0 * asin(one);
#else
  const REAL pi = 2.0f * asinf(one);
#endif
  for (int j = 1; j <= jx; j++)
    for (int i = 1; i <= ix; i++) {
      const REAL x = org[0] + dh * (REAL)(ioff + i - 1);
      const REAL y = org[1] + dh * (REAL)(joff + j - 1);
#ifdef CZ_REAL_IS_DOUBLE
      tab[(size_t)(j - 1) * ix + (i - 1)] = sin(pi * x) * sin(pi * y);
#else
      tab[(size_t)(j - 1) * ix + (i - 1)] = sinf(pi * x) * sinf(pi * y);
#endif
    }
  REAL* d = nullptr;
  HIP_CHECK(hipMalloc(&d, tab.size() * sizeof(REAL)));
  HIP_CHECK(hipMemcpy(d, tab.data(), tab.size() * sizeof(REAL), hipMemcpyHostToDevice));
  ctx.bc_tabs[key] = d;
  return d;
}

}  // namespace

namespace {
struct CheckArgs {
  int enabled = 0, itr = 0;
  double res_normal = 0.0, eps = 0.0;
  double* hist = nullptr;
  int* flag = nullptr;
  int* conv_itr = nullptr;
};
template <int MODE>
void sweep_async(const REAL* p_in, REAL* p_out, const REAL* b, const Box& bx, const Coef& cf, int par, double* res_dev,
                 int accumulate, const int* skip, const CheckArgs& ck, const MafArgs* ma = nullptr) {
  int nblk = 0;
  if (ctx.tune.fuse_fin) {
    Fin fin;
    fin.dst = res_dev, fin.accumulate = accumulate, fin.counter = ctx.counter;
    fin.do_check = ck.enabled, fin.itr = ck.itr, fin.res_normal = ck.res_normal, fin.eps = ck.eps;
    fin.hist = ck.hist, fin.flag = ck.flag, fin.conv_itr = ck.conv_itr;
    if (ma) launch_stencil_maf<MODE>(p_in, b, p_out, cf.omg, bx, par, skip, &nblk, fin, *ma);
    else launch_stencil<MODE>(p_in, b, p_out, cf, bx, par, skip, &nblk, fin);
  } else {
    if (ma) launch_stencil_maf<MODE>(p_in, b, p_out, cf.omg, bx, par, skip, &nblk, Fin(), *ma);
    else launch_stencil<MODE>(p_in, b, p_out, cf, bx, par, skip, &nblk);
    reduce_partials(nblk, res_dev, accumulate, skip);
    if (ck.enabled) czhip_check_async(res_dev, ck.res_normal, ck.eps, ck.itr, ck.hist, ck.flag, ck.conv_itr);
  }
}
}  // namespace

static inline int rb_parity(int g, const int* idx, int ofst, int color) {
  // cz_solver.f90:466  k = kst + mod(i+j+kp,2), step 2  <=>  (k + i + j + kst + kp) even  (1-based)
  // padded 0-based indices shift each of k,i,j by g-1
  return (3 * (g - 1) + idx[4] + ofst + color) & 1;
}

namespace {
// device copies of the reference's host-resident 1-D coordinate arrays (cz_Evaluate.cpp:342-363 fills them on the host)
MafArgs upload_xyz(const int* sz, int g, const REAL* X, const REAL* Y, const REAL* Z, const REAL* pvt) {
  const size_t nx = sz[0] + 2 * g, ny = sz[1] + 2 * g, nz = sz[2] + 2 * g, tot = nx + ny + nz;
  if (tot > ctx.xyz_cap) {
    if (ctx.xyz) {
      HIP_CHECK(hipStreamSynchronize(ctx.stream));
      HIP_CHECK(hipFree(ctx.xyz));
    }
    HIP_CHECK(hipMalloc(&ctx.xyz, tot * sizeof(REAL)));
    ctx.xyz_cap = tot;
  }
  HIP_CHECK(hipStreamSynchronize(ctx.stream));  // the previous call may still read the old copy
  HIP_CHECK(hipMemcpy(ctx.xyz, X, nx * sizeof(REAL), hipMemcpyHostToDevice));
  HIP_CHECK(hipMemcpy(ctx.xyz + nx, Y, ny * sizeof(REAL), hipMemcpyHostToDevice));
  HIP_CHECK(hipMemcpy(ctx.xyz + nx + ny, Z, nz * sizeof(REAL), hipMemcpyHostToDevice));
  MafArgs ma;
  ma.xc = ctx.xyz, ma.yc = ctx.xyz + nx, ma.zc = ctx.xyz + nx + ny, ma.pvt = pvt;
  return ma;
}

void launch_pivot(REAL* pvt, const Box& b, const MafArgs& ma) {
  if (b.empty) return;
  const int nplanes = b.jj1 - b.jj0 + 1;
  if (vec_ok(b, {pvt})) {
    EGeom e = make_egeom<VW>(b);
    dim3 grid((unsigned)((e.Fend - e.F0 + 255) / 256), (unsigned)nplanes);
    hipLaunchKernelGGL((pivot_k<VW>), grid, dim3(256), 0, ctx.stream, pvt, e, ma, b.nkp, b.nip);
  } else {
    EGeom e = make_egeom<1>(b);
    dim3 grid((unsigned)((e.Fend - e.F0 + 255) / 256), (unsigned)nplanes);
    hipLaunchKernelGGL((pivot_k<1>), grid, dim3(256), 0, ctx.stream, pvt, e, ma, b.nkp, b.nip);
  }
  HIP_CHECK(hipGetLastError());
}
}  // namespace

namespace {
int num_stage(int n) {  // cz.h:293-300
  int b = 1;
  for (int i = 1; i < 20; i++) {
    b *= 2;
    if (n < b) return i;
  }
  return -1;
}

template <int NW>
bool try_pcr_rb(REAL* x, const REAL* msk, const REAL* rhs, PcrGeom g, REAL omg, double* res_dev, int accumulate, size_t lds_cap) {
  const long long ncol = (long long)g.nhalf * g.nj;
  const unsigned nblk = (unsigned)((ncol + NW - 1) / NW);
  const size_t lds = (size_t)NW * 6 * (g.n + 2) * sizeof(REAL) + 64 + 16 * sizeof(double);
  if (lds > lds_cap) return false;
  ensure_partials(nblk);
  static bool attr_set = false;
  if (!attr_set) {
    HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&pcr_rb_k<NW>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_set = true;
  }
  ScopedTimer tm(LBL_PCR);
  hipLaunchKernelGGL((pcr_rb_k<NW>), dim3(nblk), dim3(64 * NW), lds, ctx.stream, x, msk, rhs, g, omg, ctx.partials, res_dev, accumulate,
                     ctx.counter);
  HIP_CHECK(hipGetLastError());
  return true;
}

template <int NW, int L, int FINAL4, int ORDER>
bool try_pcr_rb2_inst(REAL* x, REAL* wout, const REAL* msk, const REAL* rhs, const PcrGeom& g, REAL omg, double* res_dev, int accumulate,
                      int tab_len, int nfin, long long ncol) {
  const size_t lds = ((size_t)tab_len + (size_t)NW * 2 * L * (g.n + 2) + 8) * sizeof(REAL) + 32 * sizeof(double);
  if (lds > 160 * 1024) return false;
  const long long ngroups = (ncol + L - 1) / L;
  const int per_cu = (int)std::max<size_t>(1, std::min<size_t>((size_t)160 * 1024 / lds, (size_t)(32 / NW)));
  const unsigned nblk = (unsigned)std::max<long long>(1, std::min<long long>((ngroups + NW - 1) / NW, (long long)ctx.num_cu * per_cu));
  ensure_partials(nblk);
  static bool attr_set = false;
  if (!attr_set) {
    HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&pcr_rb2_k<NW, L, FINAL4, ORDER>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  160 * 1024));
    attr_set = true;
  }
  ScopedTimer tm(LBL_PCR);
  hipLaunchKernelGGL((pcr_rb2_k<NW, L, FINAL4, ORDER>), dim3(nblk), dim3(64 * NW), lds, ctx.stream, x, wout, msk, rhs, g, omg, ctx.pcr_tab,
                     tab_len, nfin, ctx.partials, res_dev, accumulate, ctx.counter);
  HIP_CHECK(hipGetLastError());
  return true;
}

// the line-independent coefficients of a line of n unknowns (pcr_coef_k), computed once per (n, pn, variant)
void ensure_pcr_table(int n, int pn, int final4, int nfin, int tab_len) {
  if (ctx.pcr_tab_n == n && ctx.pcr_tab_pn == pn && ctx.pcr_tab_final4 == final4) return;
  if ((size_t)tab_len > ctx.pcr_tab_cap) {
    if (ctx.pcr_tab) {
      HIP_CHECK(hipStreamSynchronize(ctx.stream));
      HIP_CHECK(hipFree(ctx.pcr_tab));
    }
    HIP_CHECK(hipMalloc(&ctx.pcr_tab, (size_t)tab_len * sizeof(REAL)));
    ctx.pcr_tab_cap = tab_len;
  }
  hipLaunchKernelGGL(pcr_coef_k, dim3(1), dim3(256), (size_t)4 * (n + 2) * sizeof(REAL), ctx.stream, ctx.pcr_tab, n, pn, nfin, final4);
  HIP_CHECK(hipGetLastError());
  ctx.pcr_tab_n = n, ctx.pcr_tab_pn = pn, ctx.pcr_tab_final4 = final4;
  ctx.pcr_perm_M = 0;  // the permuted copy is stale
}

template <int M, int NW, int L, int FINAL4, int ORDER>
bool try_pcr_reg_inst(REAL* x, REAL* wout, const REAL* msk, const REAL* rhs, const PcrGeom& g, REAL omg, double* res_dev, int accumulate,
                      int tab_len, long long ncol) {
  const size_t lds = ((size_t)tab_len + 8) * sizeof(REAL) + 32 * sizeof(double);
  if (lds > 160 * 1024) return false;
  const long long ngroups = (ncol + L - 1) / L;
  const int per_cu = (int)std::max<size_t>(1, std::min<size_t>((size_t)160 * 1024 / lds, (size_t)(32 / NW)));
  const unsigned nblk = (unsigned)std::max<long long>(1, std::min<long long>((ngroups + NW - 1) / NW, (long long)ctx.num_cu * per_cu));
  ensure_partials(nblk);
  static bool attr_set = false;
  if (!attr_set) {
    HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&pcr_line_reg_k<M, NW, L, FINAL4, ORDER>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_set = true;
  }
  ScopedTimer tm(LBL_PCR);
  hipLaunchKernelGGL((pcr_line_reg_k<M, NW, L, FINAL4, ORDER>), dim3(nblk), dim3(64 * NW), lds, ctx.stream, x, wout, msk, rhs, g, omg,
                     ctx.pcr_tab_perm, tab_len, ctx.partials, res_dev, accumulate, ctx.counter);
  HIP_CHECK(hipGetLastError());
  return true;
}

// register form (pcr_line_reg_k): lines of up to 1024 unknowns whose permuted table fits LDS
template <int FINAL4, int ORDER>
bool try_pcr_reg(REAL* x, REAL* wout, const REAL* msk, const REAL* rhs, const PcrGeom& g, REAL omg, double* res_dev, int accumulate,
                 long long ncol) {
  const int n = g.n, pn = g.pn;
  if (pn < (FINAL4 ? 3 : 2) || n > 1024) return false;
  const int nstage = FINAL4 ? pn - 2 : pn - 1;
  int M = 2;
  while (64 * M < n) M *= 2;
  if ((1 << nstage) < M) return false;  // the final stage must pair entries of different lanes
  const int NE = 64 * M;
  const int tab_len = (nstage * 3 + (FINAL4 ? 7 : 3)) * NE;
  if (((size_t)tab_len + 8) * sizeof(REAL) + 32 * sizeof(double) > 160 * 1024) return false;
  const int nfin = std::min(1 << nstage, n);
  ensure_pcr_table(n, pn, FINAL4, nfin, nstage * 3 * n + (FINAL4 ? 7 : 3) * nfin);
  if (ctx.pcr_perm_M != M) {
    if ((size_t)tab_len > ctx.pcr_perm_cap) {
      if (ctx.pcr_tab_perm) {
        HIP_CHECK(hipStreamSynchronize(ctx.stream));
        HIP_CHECK(hipFree(ctx.pcr_tab_perm));
      }
      HIP_CHECK(hipMalloc(&ctx.pcr_tab_perm, (size_t)tab_len * sizeof(REAL)));
      ctx.pcr_perm_cap = tab_len;
    }
    hipLaunchKernelGGL(pcr_coef_perm_k, dim3(1), dim3(256), 0, ctx.stream, ctx.pcr_tab, ctx.pcr_tab_perm, n, pn, nfin, FINAL4, M);
    HIP_CHECK(hipGetLastError());
    ctx.pcr_perm_M = M;
  }
  const int v = ctx.tune.pcr_variant;
#define CZ_PCR_REG(M_)                                                                                                                \
  if (M == M_) {                                                                                                                      \
    if (ORDER == 1) return try_pcr_reg_inst<M_, 4, 1, FINAL4, ORDER>(x, wout, msk, rhs, g, omg, res_dev, accumulate, tab_len, ncol);   \
    if (v == 161) return try_pcr_reg_inst<M_, 16, 1, FINAL4, ORDER>(x, wout, msk, rhs, g, omg, res_dev, accumulate, tab_len, ncol);   \
    if (v == 81) return try_pcr_reg_inst<M_, 8, 1, FINAL4, ORDER>(x, wout, msk, rhs, g, omg, res_dev, accumulate, tab_len, ncol);     \
    if (v == 82) return try_pcr_reg_inst<M_, 8, 2, FINAL4, ORDER>(x, wout, msk, rhs, g, omg, res_dev, accumulate, tab_len, ncol);     \
    if (v == 162) return try_pcr_reg_inst<M_, 16, 2, FINAL4, ORDER>(x, wout, msk, rhs, g, omg, res_dev, accumulate, tab_len, ncol);   \
    /* measured at 512^3 (profiles/r01/pcr_variants.txt): FP32 8 waves x 2 lines, FP64 16 waves x 1 line */                         \
    if (sizeof(REAL) == 4 && M_ <= 8)                                                                                                 \
      return try_pcr_reg_inst<M_, 8, 2, FINAL4, ORDER>(x, wout, msk, rhs, g, omg, res_dev, accumulate, tab_len, ncol);                \
    return try_pcr_reg_inst<M_, 16, 1, FINAL4, ORDER>(x, wout, msk, rhs, g, omg, res_dev, accumulate, tab_len, ncol);                 \
  }
  CZ_PCR_REG(2) CZ_PCR_REG(4) CZ_PCR_REG(8) CZ_PCR_REG(16)
#undef CZ_PCR_REG
  return false;
}

// fast form: coefficient table (computed once per line length and variant) + persistent right-hand-side-only kernel
template <int FINAL4, int ORDER>
bool try_pcr_rb2(REAL* x, REAL* wout, const REAL* msk, const REAL* rhs, const PcrGeom& g, REAL omg, double* res_dev, int accumulate) {
  const int n = g.n, pn = g.pn;
  if (pn < (FINAL4 ? 3 : 2) || pn > 20) return false;
  {
    long long nc;
    if (ORDER == 0) nc = (long long)g.nhalf * g.nj;
    else if (ORDER == 1) nc = std::min(g.ni - 1, g.color) - std::max(0, g.color - (g.nj - 1)) + 1;
    else nc = (long long)g.ni * g.nj;
    if (ctx.tune.pcr_fast >= 2 && try_pcr_reg<FINAL4, ORDER>(x, wout, msk, rhs, g, omg, res_dev, accumulate, nc)) return true;
  }
  const int nstage = FINAL4 ? pn - 2 : pn - 1;
  const int nfin = std::min(1 << nstage, n);
  const int tab_len = nstage * 3 * n + (FINAL4 ? 7 : 3) * nfin;
  const size_t fixed = ((size_t)tab_len + 8) * sizeof(REAL) + 32 * sizeof(double);
  const size_t per_line = (size_t)2 * (n + 2) * sizeof(REAL);
  if (fixed + 4 * per_line > 160 * 1024) return false;  // table + four lines must fit
  ensure_pcr_table(n, pn, FINAL4, nfin, tab_len);
  long long ncol;
  if (ORDER == 0) ncol = (long long)g.nhalf * g.nj;
  else if (ORDER == 1) ncol = std::min(g.ni - 1, g.color) - std::max(0, g.color - (g.nj - 1)) + 1;
  else ncol = (long long)g.ni * g.nj;
  if (ORDER == 1) {  // a diagonal holds few lines: small workgroups spread them over the chip
    if (try_pcr_rb2_inst<4, 1, FINAL4, ORDER>(x, wout, msk, rhs, g, omg, res_dev, accumulate, tab_len, nfin, ncol)) return true;
    return false;
  }
  const int v = ctx.tune.pcr_variant;
  // measured at 512^3 FP32 (profiles/r01/pcr_variants.txt): waves per CU matter most, 16 x 1 line beats 8 x 2 lines
  if (v == 0 || v == 161)
    if (try_pcr_rb2_inst<16, 1, FINAL4, ORDER>(x, wout, msk, rhs, g, omg, res_dev, accumulate, tab_len, nfin, ncol)) return true;
  if (v == 0 || v == 82)
    if (try_pcr_rb2_inst<8, 2, FINAL4, ORDER>(x, wout, msk, rhs, g, omg, res_dev, accumulate, tab_len, nfin, ncol)) return true;
  if (v == 0 || v == 81)
    if (try_pcr_rb2_inst<8, 1, FINAL4, ORDER>(x, wout, msk, rhs, g, omg, res_dev, accumulate, tab_len, nfin, ncol)) return true;
  return try_pcr_rb2_inst<4, 1, FINAL4, ORDER>(x, wout, msk, rhs, g, omg, res_dev, accumulate, tab_len, nfin, ncol);
}

PcrGeom make_pcr_geom(const Box& b, const int* idx, int pn, int sel) {
  PcrGeom g;
  g.nkp = b.nkp, g.nip = b.nip;
  g.kk0 = b.kk0, g.n = b.kk1 - b.kk0 + 1;
  g.ii0 = b.ii0, g.ni = b.ii1 - b.ii0 + 1, g.jj0 = b.jj0, g.nj = b.jj1 - b.jj0 + 1;
  g.ist1 = idx[0], g.jst1 = idx[2];
  g.pn = pn, g.color = sel;
  g.nhalf = (g.ni + 1) / 2 + 1;
  return g;
}

// The line-SOR variants that end in 4x4 systems or visit the columns in another order (pcr, pcr_esa, pcr_rb_esa, pcr_j_esa).
// order 0: colour `sel` in place; 1: lexicographic in place = one launch per diagonal; 2: all columns, x -> wout.
// They exist in the table form only: a line whose table does not fit LDS is refused.
void launch_pcr_variant(REAL* x, REAL* wout, const REAL* msk, const REAL* rhs, const Box& b, const int* idx, int pn, int order, int sel,
                        int final4, REAL omg, double* res_dev, int accumulate) {
  if (b.empty) {
    if (!accumulate) HIP_CHECK(hipMemsetAsync(res_dev, 0, sizeof(double), ctx.stream));
    return;
  }
  bool ok = true;
  if (order == 1) {
    const int ni = b.ii1 - b.ii0 + 1, nj = b.jj1 - b.jj0 + 1;
    for (int dgn = 0; dgn <= ni + nj - 2 && ok; dgn++) {
      const PcrGeom g = make_pcr_geom(b, idx, pn, dgn);
      ok = final4 ? try_pcr_rb2<1, 1>(x, wout, msk, rhs, g, omg, res_dev, accumulate || dgn > 0)
                  : try_pcr_rb2<0, 1>(x, wout, msk, rhs, g, omg, res_dev, accumulate || dgn > 0);
    }
  } else {
    const PcrGeom g = make_pcr_geom(b, idx, pn, sel);
    if (order == 0) ok = final4 ? try_pcr_rb2<1, 0>(x, wout, msk, rhs, g, omg, res_dev, accumulate) : try_pcr_rb2<0, 0>(x, wout, msk, rhs, g, omg, res_dev, accumulate);
    else ok = final4 ? try_pcr_rb2<1, 2>(x, wout, msk, rhs, g, omg, res_dev, accumulate) : try_pcr_rb2<0, 2>(x, wout, msk, rhs, g, omg, res_dev, accumulate);
  }
  if (!ok) {
    fprintf(stderr, "czhip: line SOR (4x4 / ordered variants): the coefficient table of a k-line of %d unknowns does not fit the 160 KiB of LDS\n",
            b.kk1 - b.kk0 + 1);
    exit(1);
  }
}

void launch_pcr_rb(REAL* x, const REAL* msk, const REAL* rhs, const Box& b, const int* idx, int pn, int color, REAL omg,
                   double* res_dev, int accumulate) {
  if (b.empty) {
    if (!accumulate) HIP_CHECK(hipMemsetAsync(res_dev, 0, sizeof(double), ctx.stream));
    return;
  }
  PcrGeom g;
  g.nkp = b.nkp, g.nip = b.nip;
  g.kk0 = b.kk0, g.n = b.kk1 - b.kk0 + 1;
  g.ii0 = b.ii0, g.ni = b.ii1 - b.ii0 + 1, g.jj0 = b.jj0, g.nj = b.jj1 - b.jj0 + 1;
  g.ist1 = idx[0], g.jst1 = idx[2];
  g.pn = pn, g.color = color;
  g.nhalf = (g.ni + 1) / 2 + 1;
  if (ctx.tune.pcr_fast && try_pcr_rb2<0, 0>(x, nullptr, msk, rhs, g, omg, res_dev, accumulate)) return;
  // one wave per k-line, NW lines per workgroup; each line keeps 2 x (a, c, d) of n+2 entries in LDS.  Prefer four
  // lines per group while two groups still fit a CU's 160 KiB, then fall back to fewer lines per group for long lines.
  if (try_pcr_rb<4>(x, msk, rhs, g, omg, res_dev, accumulate, 80 * 1024)) return;
  if (try_pcr_rb<2>(x, msk, rhs, g, omg, res_dev, accumulate, 80 * 1024)) return;
  if (try_pcr_rb<1>(x, msk, rhs, g, omg, res_dev, accumulate, 160 * 1024)) return;
  fprintf(stderr, "czhip: pcr_rb: a k-line of %d unknowns does not fit the 160 KiB of LDS\n", g.n);
  exit(1);
}

// one lexicographic SOR sweep (psor / psor_maf): a launch per tile hyperplane, then the fixed-order sum of the tile partials
void launch_psor(REAL* p, const REAL* b, const Coef& c, const Box& bx, double* res_dev, int accumulate, const int* skip,
                 const MafArgs* ma) {
  if (bx.empty) {
    if (!accumulate) HIP_CHECK(hipMemsetAsync(res_dev, 0, sizeof(double), ctx.stream));
    return;
  }
  constexpr int T = 16;
  PsorGeom g;
  g.nkp = bx.nkp, g.nip = bx.nip, g.njp = bx.njp;
  g.kk0 = bx.kk0, g.kk1 = bx.kk1, g.ii0 = bx.ii0, g.ii1 = bx.ii1, g.jj0 = bx.jj0, g.jj1 = bx.jj1;
  g.ntk = (bx.kk1 - bx.kk0 + T) / T, g.nti = (bx.ii1 - bx.ii0 + T) / T, g.ntj = (bx.jj1 - bx.jj0 + T) / T;
  const size_t ntiles = (size_t)g.ntk * g.nti * g.ntj;
  ensure_partials(ntiles);
  const size_t lds = ((size_t)(T + 2) * (T + 2) * (T + 2) + (size_t)T * T * T) * sizeof(REAL);
  static bool attr_set = false;
  if (!attr_set) {
    HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&psor_tile_k<T, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
    HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&psor_tile_k<T, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
    attr_set = true;
  }
  {
    ScopedTimer tm(LBL_PSOR);
    for (int H = 0; H <= g.ntk + g.nti + g.ntj - 3; H++) {
      if (ma) hipLaunchKernelGGL((psor_tile_k<T, 1>), dim3(g.nti, g.ntj), dim3(T * T), lds, ctx.stream, p, b, c, g, H, ctx.partials, skip, *ma);
      else hipLaunchKernelGGL((psor_tile_k<T, 0>), dim3(g.nti, g.ntj), dim3(T * T), lds, ctx.stream, p, b, c, g, H, ctx.partials, skip, MafArgs());
    }
  }
  HIP_CHECK(hipGetLastError());
  reduce_partials((int)ntiles, res_dev, accumulate, skip);
}

void launch_imask(REAL* x, const Box& b) {
  hipLaunchKernelGGL(imask_k, dim3(2048), dim3(256), 0, ctx.stream, x, b.nkp, b.nip, b.njp, b.kk0, b.kk1, b.ii0, b.ii1, b.jj0, b.jj1);
  HIP_CHECK(hipGetLastError());
}
}  // namespace

// ============================================================================================================
// Part 2: runtime
// ============================================================================================================
extern "C" {

int czhip_real_bytes(void) { return (int)sizeof(REAL); }
const char* czhip_arch(void) { return "gfx950"; }

int czhip_init(int device) {
  if (ctx.ready) return 0;
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev == 0) {
    fprintf(stderr, "czhip: no HIP device available (%s) -- this library has no CPU fallback\n", hipGetErrorString(e));
    exit(1);
  }
  if (device < 0) {
    const char* lr = getenv("LOCAL_RANK");
    device = lr ? atoi(lr) % ndev : 0;
  }
  HIP_CHECK(hipSetDevice(device));
  ctx.device = device;
  hipDeviceProp_t prop;
  HIP_CHECK(hipGetDeviceProperties(&prop, device));
  ctx.num_cu = prop.multiProcessorCount;
  HIP_CHECK(hipStreamCreateWithFlags(&ctx.stream, hipStreamNonBlocking));
  HIP_CHECK(hipMalloc(&ctx.scal_dev, 16 * sizeof(double)));
  HIP_CHECK(hipMemset(ctx.scal_dev, 0, 16 * sizeof(double)));
  HIP_CHECK(hipHostMalloc(&ctx.scal_host, 16 * sizeof(double), hipHostMallocDefault));
  HIP_CHECK(hipMalloc(&ctx.counter, 64));
  HIP_CHECK(hipMemset(ctx.counter, 0, 64));
  ctx.ready = true;
  ensure_partials(65536);
  HIP_CHECK(hipMalloc(&ctx.shell_partials, (size_t)2 * 2048 * 6 * sizeof(double)));
  if (const char* ff = getenv("CZHIP_FUSE_FIN")) ctx.tune.fuse_fin = atoi(ff);
  if (const char* t2 = getenv("CZHIP_T2")) {  // "enable[,threads,mv,tj]"
    int en = 1, a = 0, b2 = 0, c2 = -1;
    const int n = sscanf(t2, "%d,%d,%d,%d", &en, &a, &b2, &c2);
    if (n >= 1) czhip_set_tuning2(n >= 2 ? a : 0, n >= 3 ? b2 : 0, n >= 4 ? c2 : -1, en);
  }
  if (const char* pc = getenv("CZHIP_PCR")) {  // "fast[,variant]"
    int f = 1, v = 0;
    sscanf(pc, "%d,%d", &f, &v);
    ctx.tune.pcr_fast = f, ctx.tune.pcr_variant = v;
  }
  const char* tu = getenv("CZHIP_TUNING");  // "threads,m,tj,pf"
  if (tu) {
    int a = 0, b = 0, c = 0, d = -1;
    if (sscanf(tu, "%d,%d,%d,%d", &a, &b, &c, &d) >= 2) czhip_set_tuning(a, b, c < 0 ? 0 : c, d);
  }
  return 0;
}

void czhip_finalize(void) {
  if (!ctx.ready) return;
  HIP_CHECK(hipStreamSynchronize(ctx.stream));
  for (auto& kv : ctx.bc_tabs) (void)hipFree(kv.second);
  ctx.bc_tabs.clear();
  (void)hipFree(ctx.partials);
  (void)hipFree(ctx.shell_partials);
  if (ctx.pcr_tab) (void)hipFree(ctx.pcr_tab);
  if (ctx.pcr_tab_perm) (void)hipFree(ctx.pcr_tab_perm);
  (void)hipFree(ctx.scal_dev);
  (void)hipHostFree(ctx.scal_host);
  (void)hipStreamDestroy(ctx.stream);
  ctx = Ctx();
}

CZ_REAL* czhip_alloc_s3d(const int* sz) {
  ensure_init();
  const size_t n = (size_t)(sz[0] + 4) * (size_t)(sz[1] + 4) * (size_t)(sz[2] + 4);  // GUIDE = 2, cz_Define.h:40
  REAL* p = nullptr;
  HIP_CHECK(hipMalloc(&p, n * sizeof(REAL)));
  HIP_CHECK(hipMemsetAsync(p, 0, n * sizeof(REAL), ctx.stream));
  HIP_CHECK(hipStreamSynchronize(ctx.stream));
  return p;
}

void czhip_free(void* d) {
  if (!d) return;
  ensure_init();
  HIP_CHECK(hipStreamSynchronize(ctx.stream));
  HIP_CHECK(hipFree(d));
}

void czhip_h2d(void* dst, const void* src, size_t bytes) {
  ensure_init();
  HIP_CHECK(hipStreamSynchronize(ctx.stream));
  HIP_CHECK(hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice));
}

void czhip_d2h(void* dst, const void* src, size_t bytes) {
  ensure_init();
  HIP_CHECK(hipStreamSynchronize(ctx.stream));
  HIP_CHECK(hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost));
}

void czhip_sync(void) {
  ensure_init();
  HIP_CHECK(hipStreamSynchronize(ctx.stream));
}

void* czhip_stream(void) {
  ensure_init();
  return (void*)ctx.stream;
}

int czhip_set_tuning(int threads, int m, int tj, int pf) {
  Tuning t = ctx.tune;
  if (threads > 0) t.threads = threads;
  if (m > 0) t.m = m;
  if (tj >= 0) t.tj = tj;
  if (pf >= 0) t.pf = pf;
  const bool ok = (t.threads == 256 || t.threads == 512 || t.threads == 1024) && (t.m == 1 || t.m == 2 || (t.m == 4 && t.threads != 1024)) &&
                  (t.pf == 0 || t.pf == 1);
  if (!ok) return 1;
  ctx.tune = t;
  return 0;
}

void czhip_get_tuning(int* threads, int* m, int* tj, int* pf) {
  *threads = ctx.tune.threads, *m = ctx.tune.m, *tj = ctx.tune.tj, *pf = ctx.tune.pf;
}

void czhip_timing(int enable) {
  ensure_init();
  HIP_CHECK(hipStreamSynchronize(ctx.stream));
  for (auto& e : ctx.ev_used) ctx.ev_free.push_back(e);
  ctx.ev_used.clear();
  for (int l = 0; l < 16; l++) ctx.t_acc[l] = 0.0, ctx.t_cnt[l] = 0;
  ctx.timing = enable != 0;
}

int czhip_timing_read(const char* label, double* total_ms) {
  ensure_init();
  HIP_CHECK(hipStreamSynchronize(ctx.stream));
  int want = -1;
  for (int l = 0; l < LBL_COUNT; l++)
    if (!strcmp(kLabelNames[l], label)) want = l;
  if (want < 0) {
    if (total_ms) *total_ms = 0.0;
    return 0;
  }
  double tot = ctx.t_acc[want];
  int n = (int)ctx.t_cnt[want];
  for (auto& e : ctx.ev_used) {
    if (e.label != want) continue;
    float ms = 0.f;
    HIP_CHECK(hipEventElapsedTime(&ms, e.a, e.b));
    tot += ms;
    n++;
  }
  if (total_ms) *total_ms = tot;
  return n;
}

// ============================================================================================================
// Part 3: asynchronous operations
// ============================================================================================================
void czhip_jacobi_async(const CZ_REAL* p_in, CZ_REAL* p_out, const CZ_REAL* b, const int* sz, const int* idx, int g,
                        const CZ_REAL* cf, CZ_REAL omg, double* res_dev, int accumulate, const int* skip) {
  ensure_init();
  const Box bx = make_box(sz, idx, g);
  if (bx.empty) {
    if (!accumulate) HIP_CHECK(hipMemsetAsync(res_dev, 0, sizeof(double), ctx.stream));
    return;
  }
  sweep_async<MODE_JACOBI>(p_in, p_out, b, bx, make_coef(cf, omg), 0, res_dev, accumulate, skip, CheckArgs());
}

void czhip_jacobi_checked_async(const CZ_REAL* p_in, CZ_REAL* p_out, const CZ_REAL* b, const int* sz, const int* idx, int g,
                                const CZ_REAL* cf, CZ_REAL omg, double* res_dev, double res_normal, double eps, int itr,
                                double* hist_dev, int* flag_dev, int* conv_itr_dev) {
  ensure_init();
  const Box bx = make_box(sz, idx, g);
  CheckArgs ck;
  ck.enabled = 1, ck.itr = itr, ck.res_normal = res_normal, ck.eps = eps, ck.hist = hist_dev, ck.flag = flag_dev, ck.conv_itr = conv_itr_dev;
  if (bx.empty) {
    HIP_CHECK(hipMemsetAsync(res_dev, 0, sizeof(double), ctx.stream));
    czhip_check_async(res_dev, res_normal, eps, itr, hist_dev, flag_dev, conv_itr_dev);
    return;
  }
  sweep_async<MODE_JACOBI>(p_in, p_out, b, bx, make_coef(cf, omg), 0, res_dev, 0, flag_dev, ck);
}

void czhip_rbsor_async(CZ_REAL* p, const CZ_REAL* b, const int* sz, const int* idx, int g, const CZ_REAL* cf, int ofst,
                       int color, CZ_REAL omg, double* res_dev, int accumulate, const int* skip) {
  ensure_init();
  const Box bx = make_box(sz, idx, g);
  if (bx.empty) {
    if (!accumulate) HIP_CHECK(hipMemsetAsync(res_dev, 0, sizeof(double), ctx.stream));
    return;
  }
  sweep_async<MODE_RB>(p, p, b, bx, make_coef(cf, omg), rb_parity(g, idx, ofst, color), res_dev, accumulate, skip, CheckArgs());
}

void czhip_rbsor_checked_async(CZ_REAL* p, const CZ_REAL* b, const int* sz, const int* idx, int g, const CZ_REAL* cf, int ofst,
                               int color, CZ_REAL omg, double* res_dev, int accumulate, double res_normal, double eps, int itr,
                               double* hist_dev, int* flag_dev, int* conv_itr_dev) {
  ensure_init();
  const Box bx = make_box(sz, idx, g);
  CheckArgs ck;
  ck.enabled = 1, ck.itr = itr, ck.res_normal = res_normal, ck.eps = eps, ck.hist = hist_dev, ck.flag = flag_dev, ck.conv_itr = conv_itr_dev;
  if (bx.empty) {
    if (!accumulate) HIP_CHECK(hipMemsetAsync(res_dev, 0, sizeof(double), ctx.stream));
    czhip_check_async(res_dev, res_normal, eps, itr, hist_dev, flag_dev, conv_itr_dev);
    return;
  }
  sweep_async<MODE_RB>(p, p, b, bx, make_coef(cf, omg), rb_parity(g, idx, ofst, color), res_dev, accumulate, flag_dev, ck);
}

// Two fused Jacobi sweeps u -> w (time n -> n+2).  res_dev[0], res_dev[1] receive sum dp^2 of sweep n+1 / n+2.
// With check arguments (hist_dev != NULL) the last workgroup also performs the bookkeeping of cz_Poisson.cpp:67-77 for
// iterations itr and itr+1 in order; a converged first sweep leaves the flag set with conv_itr = itr (the caller then
// recomputes that single sweep from u, which this kernel never modifies).  Returns 1 if launched, 0 if the geometry
// is not supported (caller falls back to two single sweeps).
int czhip_jacobi2_async(const CZ_REAL* u, CZ_REAL* w, const CZ_REAL* b, const int* sz, const int* idx, const int* idx1, int g,
                        const CZ_REAL* cf, CZ_REAL omg, double* res_dev, double res_normal, double eps, int itr, double* hist_dev,
                        int* flag_dev, int* conv_itr_dev, const int* skip_flag_dev) {
  ensure_init();
  if (!ctx.tune.fuse_fin) return 0;
  const Box bx = make_box(sz, idx, g);
  if (bx.empty || g < 2) return 0;
  const Box ba = idx1 ? make_box(sz, idx1, g) : bx;
  Fin2 fin;
  fin.dst = res_dev;
  if (hist_dev) {
    fin.do_check = 1, fin.itr = itr, fin.res_normal = res_normal, fin.eps = eps;
    fin.hist = hist_dev, fin.flag = flag_dev, fin.conv_itr = conv_itr_dev;
  }
  return launch_jacobi2<0>(u, b, w, make_coef(cf, omg), bx, ba, hist_dev ? flag_dev : skip_flag_dev, fin) ? 1 : 0;
}

// The first pair of sweeps of a preconditioner solve, whose start vector was just cleared (cz_Poisson.cpp:405-409: blas_clear_
// then 8 sweeps): the input field is identically zero, so it is neither cleared in memory nor read -- the arithmetic is
// the same with literal zeros.  `u_shape` is only used for its alignment/geometry checks.  No convergence bookkeeping.
int czhip_jacobi2_from_zero_async(const CZ_REAL* u_shape, CZ_REAL* w, const CZ_REAL* b, const int* sz, const int* idx, const int* idx1,
                                  int g, const CZ_REAL* cf, CZ_REAL omg, double* res_dev) {
  ensure_init();
  if (!ctx.tune.fuse_fin) return 0;
  const Box bx = make_box(sz, idx, g);
  if (bx.empty || g < 2) return 0;
  const Box ba = idx1 ? make_box(sz, idx1, g) : bx;
  Fin2 fin;
  fin.dst = res_dev;
  return launch_jacobi2<0>(u_shape, b, w, make_coef(cf, omg), bx, ba, nullptr, fin, 0, 1) ? 1 : 0;
}

// One complete red-black SOR iteration (colour 0 then colour 1, cz_Poisson.cpp:205-209) in one pass over memory, u -> w.
// res_dev[0] receives the iteration's sum dp^2 (both colours).  Same conventions as czhip_jacobi2_async.
int czhip_rbsor2_async(const CZ_REAL* u, CZ_REAL* w, const CZ_REAL* b, const int* sz, const int* idx, const int* idx1, int g,
                       const CZ_REAL* cf, int ofst, CZ_REAL omg, double* res_dev, double res_normal, double eps, int itr,
                       double* hist_dev, int* flag_dev, int* conv_itr_dev, const int* skip_flag_dev) {
  ensure_init();
  if (!ctx.tune.fuse_fin) return 0;
  const Box bx = make_box(sz, idx, g);
  if (bx.empty || g < 2) return 0;
  const Box ba = idx1 ? make_box(sz, idx1, g) : bx;
  Fin2 fin;
  fin.dst = res_dev;
  fin.single = 1;
  if (hist_dev) {
    fin.do_check = 1, fin.itr = itr, fin.res_normal = res_normal, fin.eps = eps;
    fin.hist = hist_dev, fin.flag = flag_dev, fin.conv_itr = conv_itr_dev;
  }
  return launch_jacobi2<1>(u, b, w, make_coef(cf, omg), bx, ba, hist_dev ? flag_dev : skip_flag_dev, fin, rb_parity(g, idx, ofst, 0)) ? 1 : 0;
}

// The fused pass split the way a decomposed brick runs it: the slabs behind the faces with nID[f] >= 0 first
// (pair_shell_k), then the interior (jacobi2_k) -- same result as the unsplit launch.  rb_ofst < 0: two Jacobi sweeps,
// res_dev[0..1]; rb_ofst >= 0: one red-black iteration with that ofst, res_dev[0].  Returns 0 when nothing was launched.
int czhip_pair_split_async(const CZ_REAL* u, CZ_REAL* w, const CZ_REAL* b, const int* sz, const int* idx, const int* idx1,
                           const int* nID, int g, const CZ_REAL* cf, CZ_REAL omg, int rb_ofst, double* res_dev) {
  ensure_init();
  int boxes[36], in0[6], in1[6];
  const int n = czhip_internal::pair_plan(idx, nID, boxes, in0, in1);
  if (n == 0 || !czhip_internal::pair_probe(u, w, b, sz, in0, in1, g)) return 0;
  const int rb = rb_ofst >= 0 ? rb_parity(g, idx, rb_ofst, 0) : -1;
  czhip_internal::pair_shell_async(u, w, b, sz, idx1 ? idx1 : idx, boxes, n, g, cf, omg, rb, nullptr);
  return czhip_internal::pair_box_async(u, w, b, sz, in0, in1, g, cf, omg, rb, res_dev, 1, nullptr);
}

int czhip_set_tuning2(int threads, int vec_per_thread, int planes_per_chunk, int enable) {
  Tuning t = ctx.tune;
  if (threads > 0) t.t2_threads = threads;
  if (vec_per_thread > 0) t.t2_mv = vec_per_thread;
  if (planes_per_chunk >= 0) t.t2_tj = planes_per_chunk;
  if (enable >= 0) t.use_t2 = enable;
  const int k = t.t2_threads * 100 + t.t2_mv;
  if (k != 25604 && k != 51202 && k != 51203 && k != 102402) return 1;
  ctx.tune = t;
  return 0;
}

int czhip_use_t2(void) { return ctx.tune.use_t2; }

// line-SOR kernel choice: form 0 = pcr_rb_k (the reference's arithmetic literally, pcr_rb only), 1 = table + d in LDS, 2 = table +
// d in registers (default); variant = waves per workgroup * 10 + lines per wave, 0 = measured default.  Negative: keep.
int czhip_set_pcr_mode(int form, int variant) {
  ensure_init();
  if (form > 2) return 1;
  if (form >= 0) ctx.tune.pcr_fast = form;
  if (variant >= 0) ctx.tune.pcr_variant = variant;
  return 0;
}

void czhip_check2_async(const double* res_dev, double res_normal, double eps, int itr, double* hist_dev, int* flag_dev,
                        int* conv_itr_dev) {
  ensure_init();
  hipLaunchKernelGGL(check2_k, dim3(1), dim3(1), 0, ctx.stream, res_dev, res_normal, eps, itr, hist_dev, flag_dev, conv_itr_dev);
  HIP_CHECK(hipGetLastError());
}

void czhip_check_async(const double* res_dev, double res_normal, double eps, int itr, double* hist_dev, int* flag_dev,
                       int* conv_itr_dev) {
  ensure_init();
  hipLaunchKernelGGL(check_k, dim3(1), dim3(1), 0, ctx.stream, res_dev, res_normal, eps, itr, hist_dev, flag_dev, conv_itr_dev);
  HIP_CHECK(hipGetLastError());
}

// ============================================================================================================
// Part 1: drop-in kernels (synchronous, reference semantics)
// ============================================================================================================
void bc_k_(int* sz, int* gp, CZ_REAL* p, CZ_REAL* dh, CZ_REAL* org, int* nID) {
  ensure_init();
  const int g = *gp, ix = sz[0], jx = sz[1], kx = sz[2];
  const int nkp = kx + 2 * g, nip = ix + 2 * g;
  // nID order I-,I+,J-,J+,K-,K+ (cz_fparam.fi:10-16)
  if (nID[4] < 0 || nID[5] < 0) {
    const REAL* tab = bc_table(ix, jx, *dh, org);
    dim3 grid((ix + 127) / 128, jx);
    if (nID[4] < 0) hipLaunchKernelGGL(bc_kface_k, grid, dim3(128), 0, ctx.stream, p, tab, ix, jx, 1, g, nkp, nip);
    if (nID[5] < 0) hipLaunchKernelGGL(bc_kface_k, grid, dim3(128), 0, ctx.stream, p, tab, ix, jx, kx, g, nkp, nip);
  }
  {
    dim3 grid((kx + 127) / 128, jx);
    if (nID[0] < 0) hipLaunchKernelGGL(bc_iface_k, grid, dim3(128), 0, ctx.stream, p, jx, kx, 1, g, nkp, nip);
    if (nID[1] < 0) hipLaunchKernelGGL(bc_iface_k, grid, dim3(128), 0, ctx.stream, p, jx, kx, ix, g, nkp, nip);
  }
  {
    dim3 grid((kx + 127) / 128, ix);
    if (nID[2] < 0) hipLaunchKernelGGL(bc_jface_k, grid, dim3(128), 0, ctx.stream, p, ix, kx, 1, g, nkp, nip);
    if (nID[3] < 0) hipLaunchKernelGGL(bc_jface_k, grid, dim3(128), 0, ctx.stream, p, ix, kx, jx, g, nkp, nip);
  }
  HIP_CHECK(hipGetLastError());
  HIP_CHECK(hipStreamSynchronize(ctx.stream));
}

void jacobi_(CZ_REAL* p, int* sz, int* idx, int* g, CZ_REAL* cf, CZ_REAL* omg, CZ_REAL* b, double* res, CZ_REAL* wk2,
             double* flop) {
  ensure_init();
  *flop += 18.0 * npts(idx);  // cz_solver.f90:315-318
  const Box bx = make_box(sz, idx, *g);
  if (bx.empty) return;
  czhip_jacobi_async(p, wk2, b, sz, idx, *g, cf, *omg, ctx.scal_dev + 0, 0, nullptr);
  launch_ewise<OP_COPY>(p, wk2, nullptr, (REAL)0, (REAL)0, bx);  // p <- wk2 on the inner box (:369-375)
  *res += read_scalar(0);                                          // :384 (double accumulation)
}

void psor2sma_core_(CZ_REAL* p, int* sz, int* idx, int* g, CZ_REAL* cf, int* ip, int* color, CZ_REAL* omg, CZ_REAL* b,
                    double* res, double* flop) {
  ensure_init();
  *flop += 18.0 * 0.5 * npts(idx);  // :438-441
  const Box bx = make_box(sz, idx, *g);
  if (bx.empty) return;
  czhip_rbsor_async(p, b, sz, idx, *g, cf, *ip, *color, *omg, ctx.scal_dev + 0, 0, nullptr);
  *res += read_scalar(0);
}

void blas_clear_(CZ_REAL* x, int* sz, int* g) {
  ensure_init();
  const size_t n = (size_t)(sz[0] + 2 * *g) * (size_t)(sz[1] + 2 * *g) * (size_t)(sz[2] + 2 * *g);
  HIP_CHECK(hipMemsetAsync(x, 0, n * sizeof(REAL), ctx.stream));
  HIP_CHECK(hipStreamSynchronize(ctx.stream));
}

void blas_copy_(CZ_REAL* dst, CZ_REAL* src, int* sz, int* g) {
  ensure_init();
  const size_t n = (size_t)(sz[0] + 2 * *g) * (size_t)(sz[1] + 2 * *g) * (size_t)(sz[2] + 2 * *g);
  HIP_CHECK(hipMemcpyAsync(dst, src, n * sizeof(REAL), hipMemcpyDeviceToDevice, ctx.stream));
  HIP_CHECK(hipStreamSynchronize(ctx.stream));
}

void blas_triad_(CZ_REAL* z, CZ_REAL* x, CZ_REAL* y, CZ_REAL* a, int* sz, int* idx, int* g, double* flop) {
  ensure_init();
  *flop += 2.0 * npts(idx);
  launch_ewise<OP_TRIAD>(z, x, y, *a, (REAL)0, make_box(sz, idx, *g));
  HIP_CHECK(hipStreamSynchronize(ctx.stream));
}

void blas_dot1_(CZ_REAL* r, CZ_REAL* p, int* sz, int* idx, int* g, double* flop) {
  ensure_init();
  *flop += 2.0 * npts(idx);
  launch_dot<0>(p, p, make_box(sz, idx, *g), ctx.scal_dev + 1);
  *r = (REAL)read_scalar(1);
}

void blas_dot2_(CZ_REAL* r, CZ_REAL* p, CZ_REAL* q, int* sz, int* idx, int* g, double* flop) {
  ensure_init();
  *flop += 2.0 * npts(idx);
  launch_dot<1>(p, q, make_box(sz, idx, *g), ctx.scal_dev + 1);
  *r = (REAL)read_scalar(1);
}

void blas_bicg_1_(CZ_REAL* p, CZ_REAL* r, CZ_REAL* q, CZ_REAL* beta, CZ_REAL* omg, int* sz, int* idx, int* g, double* flop) {
  ensure_init();
  *flop += 4.0 * npts(idx);
  launch_ewise<OP_BICG1>(p, r, q, *beta, *omg, make_box(sz, idx, *g));
  HIP_CHECK(hipStreamSynchronize(ctx.stream));
}

void blas_bicg_2_(CZ_REAL* z, CZ_REAL* x, CZ_REAL* y, CZ_REAL* a, CZ_REAL* b, int* sz, int* idx, int* g, double* flop) {
  ensure_init();
  *flop += 4.0 * npts(idx);
  launch_ewise<OP_BICG2>(z, x, y, *a, *b, make_box(sz, idx, *g));
  HIP_CHECK(hipStreamSynchronize(ctx.stream));
}

void blas_calc_ax_(CZ_REAL* ap, CZ_REAL* p, int* sz, int* idx, int* g, CZ_REAL* cf, double* flop) {
  ensure_init();
  *flop += 13.0 * npts(idx);
  const Box bx = make_box(sz, idx, *g);
  if (bx.empty) return;
  launch_stencil<MODE_AX>(p, p, ap, make_coef(cf, (REAL)0), bx, 0, nullptr, nullptr);
  HIP_CHECK(hipStreamSynchronize(ctx.stream));
}

void blas_calc_rk_(CZ_REAL* r, CZ_REAL* p, CZ_REAL* b, int* sz, int* idx, int* g, CZ_REAL* cf, double* flop) {
  ensure_init();
  *flop += 14.0 * npts(idx);
  const Box bx = make_box(sz, idx, *g);
  if (bx.empty) return;
  launch_stencil<MODE_RK>(p, b, r, make_coef(cf, (REAL)0), bx, 0, nullptr, nullptr);
  HIP_CHECK(hipStreamSynchronize(ctx.stream));
}

// ---- MAF flavour, drop-in symbols (cz_Ffunc.h:170-208, 524-553).  X, Y, Z are HOST arrays as in the reference.
void jacobi_maf_(CZ_REAL* p, int* sz, int* idx, int* g, CZ_REAL* X, CZ_REAL* Y, CZ_REAL* Z, CZ_REAL* omg, CZ_REAL* b, double* res,
                 CZ_REAL* wk2, CZ_REAL* tmp, double* flop) {
  ensure_init();
  *flop += 66.0 * npts(idx);                                    // cz_maf.f90:157-160
  for (int k = 0; k < sz[2] + 2 * *g; k++) tmp[k] = (REAL)0;    // :155 (host work array; only used by the _SVR build)
  const Box bx = make_box(sz, idx, *g);
  if (bx.empty) return;
  const MafArgs ma = upload_xyz(sz, *g, X, Y, Z, nullptr);
  sweep_async<MODE_JACOBI>(p, wk2, b, bx, make_coef_omg(*omg), 0, ctx.scal_dev + 0, 0, nullptr, CheckArgs(), &ma);
  launch_ewise<OP_COPY>(p, wk2, nullptr, (REAL)0, (REAL)0, bx);
  *res += read_scalar(0);
}

void psor2sma_core_maf_(CZ_REAL* p, int* sz, int* idx, int* g, CZ_REAL* X, CZ_REAL* Y, CZ_REAL* Z, int* ip, int* color,
                        CZ_REAL* omg, CZ_REAL* b, double* res, CZ_REAL* tmp, double* flop) {
  ensure_init();
  (void)tmp;
  *flop += 66.0 * 0.5 * npts(idx);
  const Box bx = make_box(sz, idx, *g);
  if (bx.empty) return;
  const MafArgs ma = upload_xyz(sz, *g, X, Y, Z, nullptr);
  sweep_async<MODE_RB>(p, p, b, bx, make_coef_omg(*omg), rb_parity(*g, idx, *ip, *color), ctx.scal_dev + 0, 0, nullptr,
                       CheckArgs(), &ma);
  *res += read_scalar(0);
}

void psor_(CZ_REAL* p, int* sz, int* idx, int* g, CZ_REAL* cf, CZ_REAL* omg, CZ_REAL* b, double* res, double* flop) {
  ensure_init();
  *flop += 18.0 * npts(idx);  // cz_solver.f90:237-240
  const Box bx = make_box(sz, idx, *g);
  if (bx.empty) return;
  launch_psor(p, b, make_coef(cf, *omg), bx, ctx.scal_dev + 0, 0, nullptr, nullptr);
  *res += read_scalar(0);
}

void psor_maf_(CZ_REAL* p, int* sz, int* idx, int* g, CZ_REAL* X, CZ_REAL* Y, CZ_REAL* Z, CZ_REAL* omg, CZ_REAL* b, double* res,
               double* flop) {
  ensure_init();
  *flop += 66.0 * npts(idx);  // cz_maf.f90:50-53
  const Box bx = make_box(sz, idx, *g);
  if (bx.empty) return;
  if (bx.g != 2) {
    fprintf(stderr, "czhip: the MAF kernels assume GUIDE = 2 (X(-1:sz+2), cz_maf.f90:36-38)\n");
    exit(1);
  }
  const MafArgs ma = upload_xyz(sz, *g, X, Y, Z, nullptr);
  launch_psor(p, b, make_coef_omg(*omg), bx, ctx.scal_dev + 0, 0, nullptr, &ma);
  *res += read_scalar(0);
}

void calc_rk_maf_(CZ_REAL* r, CZ_REAL* p, CZ_REAL* b, int* sz, int* idx, int* g, CZ_REAL* X, CZ_REAL* Y, CZ_REAL* Z,
                  CZ_REAL* pvt, double* flop) {
  ensure_init();
  *flop += 63.0 * npts(idx);
  const Box bx = make_box(sz, idx, *g);
  if (bx.empty) return;
  const MafArgs ma = upload_xyz(sz, *g, X, Y, Z, pvt);
  launch_stencil_maf<MODE_RK>(p, b, r, (REAL)0, bx, 0, nullptr, nullptr, Fin(), ma);
  HIP_CHECK(hipStreamSynchronize(ctx.stream));
}

void calc_ax_maf_(CZ_REAL* ap, CZ_REAL* p, int* sz, int* idx, int* g, CZ_REAL* X, CZ_REAL* Y, CZ_REAL* Z, CZ_REAL* pvt,
                  double* flop) {
  ensure_init();
  *flop += 63.0 * npts(idx);
  const Box bx = make_box(sz, idx, *g);
  if (bx.empty) return;
  const MafArgs ma = upload_xyz(sz, *g, X, Y, Z, pvt);
  launch_stencil_maf<MODE_AX>(p, p, ap, (REAL)0, bx, 0, nullptr, nullptr, Fin(), ma);
  HIP_CHECK(hipStreamSynchronize(ctx.stream));
}

void search_pivot_(CZ_REAL* pvt, int* sz, int* idx, int* g, CZ_REAL* X, CZ_REAL* Y, CZ_REAL* Z) {
  ensure_init();
  const Box bx = make_box(sz, idx, *g);
  if (bx.empty) return;
  const MafArgs ma = upload_xyz(sz, *g, X, Y, Z, nullptr);
  launch_pivot(pvt, bx, ma);
  HIP_CHECK(hipStreamSynchronize(ctx.stream));
}

// ---- line SOR by PCR, drop-in symbols (cz_Ffunc.h:60-77 pcr_rb_, :440-443 imask_k_).  The six 1-D work arrays of the
// reference are host scratch of its CPU implementation; the GPU keeps the line systems in LDS and does not touch them.
void pcr_rb_(int* sz, int* idx, int* g, int* pn, int* ofst, int* color, CZ_REAL* x, CZ_REAL* msk, CZ_REAL* rhs, CZ_REAL* a,
             CZ_REAL* c, CZ_REAL* d, CZ_REAL* a1, CZ_REAL* c1, CZ_REAL* d1, CZ_REAL* omg, double* res, double* flop) {
  ensure_init();
  (void)ofst, (void)a, (void)c, (void)d, (void)a1, (void)c1, (void)d1;
  const double nk = idx[5] - idx[4] + 1;
  *flop += (double)((idx[3] - idx[2] + 1) * (idx[1] - idx[0] + 1)) *
           (nk * 6.0 + nk * (*pn - 1) * 14.0 + (double)(1 << (*pn - 1)) * 9.0 + nk * 6.0 + 6.0) * 0.5;  // cz_solver.f90:523-531
  const Box bx = make_box(sz, idx, *g);
  if (bx.empty) return;
  launch_pcr_rb(x, msk, rhs, bx, idx, *pn, *color, *omg, ctx.scal_dev + 0, 0);
  *res += read_scalar(0);
}

namespace {
double pcr_flop(const int* idx, int stages, double fin) {
  const double nk = idx[5] - idx[4] + 1;
  return (double)((idx[3] - idx[2] + 1) * (idx[1] - idx[0] + 1)) * (nk * 6.0 + nk * stages * 14.0 + fin + nk * 6.0 + 6.0);
}
}  // namespace

void pcr_(int* sz, int* idx, int* g, int* pn, CZ_REAL* x, CZ_REAL* msk, CZ_REAL* rhs, CZ_REAL* a, CZ_REAL* c, CZ_REAL* d, CZ_REAL* a1,
          CZ_REAL* c1, CZ_REAL* d1, CZ_REAL* omg, double* res, double* flop) {
  ensure_init();
  (void)a, (void)c, (void)d, (void)a1, (void)c1, (void)d1;
  *flop += pcr_flop(idx, *pn - 2, (double)(1 << (*pn - 2)) * 74.0);  // cz_solver.f90:689-696
  const Box bx = make_box(sz, idx, *g);
  if (bx.empty) return;
  launch_pcr_variant(x, nullptr, msk, rhs, bx, idx, *pn, 1, 0, 1, *omg, ctx.scal_dev + 0, 0);
  *res += read_scalar(0);
}

void pcr_esa_(int* sz, int* idx, int* g, int* pn, int* s, CZ_REAL* x, CZ_REAL* msk, CZ_REAL* rhs, CZ_REAL* a, CZ_REAL* c, CZ_REAL* d,
              CZ_REAL* a1, CZ_REAL* c1, CZ_REAL* d1, CZ_REAL* omg, double* res, double* flop) {
  ensure_init();
  (void)s, (void)a, (void)c, (void)d, (void)a1, (void)c1, (void)d1;
  *flop += pcr_flop(idx, *pn - 2, (double)(1 << (*pn - 2)) * 78.0);  // :1078-1085
  const Box bx = make_box(sz, idx, *g);
  if (bx.empty) return;
  launch_pcr_variant(x, nullptr, msk, rhs, bx, idx, *pn, 1, 0, 1, *omg, ctx.scal_dev + 0, 0);
  *res += read_scalar(0);
}

void pcr_rb_esa_(int* sz, int* idx, int* g, int* pn, int* ofst, int* color, int* s, CZ_REAL* x, CZ_REAL* msk, CZ_REAL* rhs, CZ_REAL* a,
                 CZ_REAL* c, CZ_REAL* d, CZ_REAL* a1, CZ_REAL* c1, CZ_REAL* d1, CZ_REAL* omg, double* res, double* flop) {
  ensure_init();
  (void)ofst, (void)s, (void)a, (void)c, (void)d, (void)a1, (void)c1, (void)d1;
  *flop += pcr_flop(idx, *pn - 2, (double)(1 << (*pn - 2)) * 78.0) * 0.5;  // :1291-1299
  const Box bx = make_box(sz, idx, *g);
  if (bx.empty) return;
  launch_pcr_variant(x, nullptr, msk, rhs, bx, idx, *pn, 0, *color, 1, *omg, ctx.scal_dev + 0, 0);
  *res += read_scalar(0);
}

void pcr_j_esa_(int* sz, int* idx, int* g, int* pn, int* s, CZ_REAL* x, CZ_REAL* msk, CZ_REAL* rhs, CZ_REAL* a, CZ_REAL* c, CZ_REAL* d,
                CZ_REAL* a1, CZ_REAL* c1, CZ_REAL* d1, CZ_REAL* src, CZ_REAL* wrk, CZ_REAL* omg, double* res, double* flop) {
  ensure_init();
  (void)s, (void)a, (void)c, (void)d, (void)a1, (void)c1, (void)d1, (void)src;
  *flop += pcr_flop(idx, *pn - 1, (double)(1 << (*pn - 1)) * 9.0);  // :1499-1506
  const Box bx = make_box(sz, idx, *g);
  if (bx.empty) return;
  launch_pcr_variant(x, wrk, msk, rhs, bx, idx, *pn, 2, 0, 0, *omg, ctx.scal_dev + 0, 0);
  czhip_internal::copy_inner_async(x, wrk, sz, idx, *g);  // :1655-1663
  *res += read_scalar(0);
}

void imask_k_(CZ_REAL* x, int* sz, int* idx, int* g) {
  ensure_init();
  Box bx = make_box(sz, idx, *g);
  launch_imask(x, bx);
  HIP_CHECK(hipStreamSynchronize(ctx.stream));
}

}  // extern "C"

// internal hooks for the driver (cz_driver.cpp): asynchronous forms of the blas kernels
namespace czhip_internal {
hipStream_t stream() {
  ensure_init();
  return ctx.stream;
}
void triad_async(REAL* z, const REAL* x, const REAL* y, REAL a, const int* sz, const int* idx, int g) {
  launch_ewise<OP_TRIAD>(z, x, y, a, (REAL)0, make_box(sz, idx, g));
}
void bicg1_async(REAL* p, const REAL* r, const REAL* q, REAL beta, REAL omg, const int* sz, const int* idx, int g) {
  launch_ewise<OP_BICG1>(p, r, q, beta, omg, make_box(sz, idx, g));
}
void bicg2_async(REAL* z, const REAL* x, const REAL* y, REAL a, REAL b, const int* sz, const int* idx, int g) {
  launch_ewise<OP_BICG2>(z, x, y, a, b, make_box(sz, idx, g));
}
void calc_ax_async(REAL* ap, const REAL* p, const int* sz, const int* idx, int g, const REAL* cf) {
  const Box bx = make_box(sz, idx, g);
  if (!bx.empty) launch_stencil<MODE_AX>(p, p, ap, make_coef(cf, (REAL)0), bx, 0, nullptr, nullptr);
}
void calc_rk_async(REAL* r, const REAL* p, const REAL* b, const int* sz, const int* idx, int g, const REAL* cf) {
  const Box bx = make_box(sz, idx, g);
  if (!bx.empty) launch_stencil<MODE_RK>(p, b, r, make_coef(cf, (REAL)0), bx, 0, nullptr, nullptr);
}
// SpMV with the two dot products of the result folded in: dots_dev[0] = ap.y, dots_dev[1] = ap.ap
void calc_ax_dots_async(REAL* ap, const REAL* p, const REAL* y, const int* sz, const int* idx, int g, const REAL* cf,
                        const MafPtrs* maf, double* dots_dev) {
  const Box bx = make_box(sz, idx, g);
  if (bx.empty) {
    HIP_CHECK(hipMemsetAsync(dots_dev, 0, 2 * sizeof(double), ctx.stream));
    return;
  }
  Fin fin;
  fin.dst = dots_dev, fin.dst2 = dots_dev + 1, fin.ax_dots = 1, fin.doty = y, fin.counter = ctx.counter;
  if (maf) {
    MafArgs ma{maf->xc, maf->yc, maf->zc, maf->pvt};
    launch_stencil_maf<MODE_AX>(p, p, ap, (REAL)0, bx, 0, nullptr, nullptr, fin, ma);
  } else {
    launch_stencil<MODE_AX>(p, p, ap, make_coef(cf, (REAL)0), bx, 0, nullptr, nullptr, fin);
  }
}
// z = a*x + y with dots_dev[0] = z.z, dots_dev[1] = z.w
void triad_dots_async(REAL* z, const REAL* x, const REAL* y, const REAL* w, REAL a, const int* sz, const int* idx, int g,
                      double* dots_dev) {
  const Box b = make_box(sz, idx, g);
  if (b.empty) {
    HIP_CHECK(hipMemsetAsync(dots_dev, 0, 2 * sizeof(double), ctx.stream));
    return;
  }
  const int nplanes = b.jj1 - b.jj0 + 1;
  ScopedTimer tm(LBL_EWISE);
  if (vec_ok(b, {z, x, y, w})) {
    EGeom e = make_egeom<VW>(b);
    const unsigned gx = (unsigned)((e.Fend - e.F0 + 255) / 256);
    const unsigned gy = (unsigned)std::max(1, std::min(nplanes, (int)(4096 / gx)));
    ensure_partials((size_t)2 * gx * gy);
    hipLaunchKernelGGL((triad_dots_k<VW>), dim3(gx, gy), dim3(256), 0, ctx.stream, z, x, y, w, a, e, nplanes, ctx.partials, dots_dev,
                       ctx.counter);
  } else {
    EGeom e = make_egeom<1>(b);
    const unsigned gx = (unsigned)((e.Fend - e.F0 + 255) / 256);
    const unsigned gy = (unsigned)std::max(1, std::min(nplanes, (int)(4096 / gx)));
    ensure_partials((size_t)2 * gx * gy);
    hipLaunchKernelGGL((triad_dots_k<1>), dim3(gx, gy), dim3(256), 0, ctx.stream, z, x, y, w, a, e, nplanes, ctx.partials, dots_dev,
                       ctx.counter);
  }
  HIP_CHECK(hipGetLastError());
}
void dot1_async(const REAL* p, const int* sz, const int* idx, int g, double* dst_dev) {
  launch_dot<0>(p, p, make_box(sz, idx, g), dst_dev);
}
void dot2_async(const REAL* p, const REAL* q, const int* sz, const int* idx, int g, double* dst_dev) {
  launch_dot<1>(p, q, make_box(sz, idx, g), dst_dev);
}
int pcr_num_stage(int n) { return num_stage(n); }
void pcr_rb_async(REAL* x, const REAL* msk, const REAL* rhs, const int* sz, const int* idx, int g, int pn, int color, REAL omg,
                  double* res_dev, int accumulate) {
  launch_pcr_rb(x, msk, rhs, make_box(sz, idx, g), idx, pn, color, omg, res_dev, accumulate);
}
// order 0: colour `sel` in place; 1: lexicographic in place (one launch per diagonal); 2: all columns x -> wout
void pcr_variant_async(REAL* x, REAL* wout, const REAL* msk, const REAL* rhs, const int* sz, const int* idx, int g, int pn, int order,
                       int sel, int final4, REAL omg, double* res_dev, int accumulate) {
  ensure_init();
  launch_pcr_variant(x, wout, msk, rhs, make_box(sz, idx, g), idx, pn, order, sel, final4, omg, res_dev, accumulate);
}
void imask_async(REAL* x, const int* sz, const int* idx, int g) { launch_imask(x, make_box(sz, idx, g)); }
// MAF flavour, device-resident coordinates (xc|yc|zc and pvt are device pointers)
void jacobi_maf_async(const REAL* p_in, REAL* p_out, const REAL* b, const int* sz, const int* idx, int g, const REAL* xc,
                      const REAL* yc, const REAL* zc, REAL omg, double* res_dev, const int* skip, int check, double res_normal,
                      double eps, int itr, double* hist, int* flag, int* conv_itr) {
  const Box bx = make_box(sz, idx, g);
  if (bx.empty) return;
  MafArgs ma{xc, yc, zc, nullptr};
  CheckArgs ck;
  if (check) ck.enabled = 1, ck.itr = itr, ck.res_normal = res_normal, ck.eps = eps, ck.hist = hist, ck.flag = flag, ck.conv_itr = conv_itr;
  sweep_async<MODE_JACOBI>(p_in, p_out, b, bx, make_coef_omg(omg), 0, res_dev, 0, check ? flag : skip, ck, &ma);
}
void rbsor_maf_async(REAL* p, const REAL* b, const int* sz, const int* idx, int g, const REAL* xc, const REAL* yc, const REAL* zc,
                     int ofst, int color, REAL omg, double* res_dev, int accumulate, const int* skip, int check, double res_normal,
                     double eps, int itr, double* hist, int* flag, int* conv_itr) {
  const Box bx = make_box(sz, idx, g);
  if (bx.empty) return;
  MafArgs ma{xc, yc, zc, nullptr};
  CheckArgs ck;
  if (check) ck.enabled = 1, ck.itr = itr, ck.res_normal = res_normal, ck.eps = eps, ck.hist = hist, ck.flag = flag, ck.conv_itr = conv_itr;
  sweep_async<MODE_RB>(p, p, b, bx, make_coef_omg(omg), rb_parity(g, idx, ofst, color), res_dev, accumulate, check ? flag : skip, ck,
                       &ma);
}
// one psor / psor_maf sweep (xc == nullptr: constant coefficients cf), res_dev[0] = or += sum dp^2
void psor_async(REAL* p, const REAL* b, const int* sz, const int* idx, int g, const REAL* cf, const REAL* xc, const REAL* yc,
                const REAL* zc, REAL omg, double* res_dev, int accumulate, const int* skip) {
  ensure_init();
  const Box bx = make_box(sz, idx, g);
  if (xc) {
    MafArgs ma{xc, yc, zc, nullptr};
    launch_psor(p, b, make_coef_omg(omg), bx, res_dev, accumulate, skip, &ma);
  } else {
    launch_psor(p, b, make_coef(cf, omg), bx, res_dev, accumulate, skip, nullptr);
  }
}
void calc_ax_maf_async(REAL* ap, const REAL* p, const int* sz, const int* idx, int g, const REAL* xc, const REAL* yc, const REAL* zc,
                       const REAL* pvt) {
  const Box bx = make_box(sz, idx, g);
  MafArgs ma{xc, yc, zc, pvt};
  if (!bx.empty) launch_stencil_maf<MODE_AX>(p, p, ap, (REAL)0, bx, 0, nullptr, nullptr, Fin(), ma);
}
void calc_rk_maf_async(REAL* r, const REAL* p, const REAL* b, const int* sz, const int* idx, int g, const REAL* xc, const REAL* yc,
                       const REAL* zc, const REAL* pvt) {
  const Box bx = make_box(sz, idx, g);
  MafArgs ma{xc, yc, zc, pvt};
  if (!bx.empty) launch_stencil_maf<MODE_RK>(p, b, r, (REAL)0, bx, 0, nullptr, nullptr, Fin(), ma);
}
void search_pivot_async(REAL* pvt, const int* sz, const int* idx, int g, const REAL* xc, const REAL* yc, const REAL* zc) {
  const Box bx = make_box(sz, idx, g);
  MafArgs ma{xc, yc, zc, nullptr};
  launch_pivot(pvt, bx, ma);
}
// ---- a fused pair of sweeps split into shell + interior launches (decomposed runs).  rb < 0: two Jacobi
// sweeps, res_dev[0..1]; rb >= 0: one red-black iteration with colour parity `rb` (rb_par), res_dev[0].
int rb_par(int g, const int* idx, int ofst) { return rb_parity(g, idx, ofst, 0); }

// Split of an inner box for overlapped exchanges: the cells within two layers of a rank-internal face (nID[f] >= 0; what
// the neighbours receive as their two ghost layers) form up to six disjoint slabs -- J faces first (contiguous planes), then
// I, then K -- and the rest is the interior.  Returns the number of slabs (0: nothing to split or the box is too thin);
// boxes: n x (ist,ied,jst,jed,kst,ked); interior1 = first-sweep range of the interior.
int pair_plan(const int* O, const int* nID, int* boxes, int* interior, int* interior1) {
  for (int a = 0; a < 3; a++) {
    interior[2 * a] = O[2 * a] + (nID[2 * a] >= 0 ? 2 : 0);
    interior[2 * a + 1] = O[2 * a + 1] - (nID[2 * a + 1] >= 0 ? 2 : 0);
    if (interior[2 * a + 1] - interior[2 * a] + 1 < 2) return 0;
  }
  for (int f = 0; f < 6; f++) interior1[f] = interior[f] + ((nID[f] >= 0) ? ((f & 1) ? 1 : -1) : 0);
  int n = 0;
  auto add = [&](int i0, int i1, int j0, int j1, int k0, int k1) {
    int* b = boxes + 6 * n++;
    b[0] = i0, b[1] = i1, b[2] = j0, b[3] = j1, b[4] = k0, b[5] = k1;
  };
  const int* I = interior;
  if (nID[2] >= 0) add(O[0], O[1], O[2], O[2] + 1, O[4], O[5]);
  if (nID[3] >= 0) add(O[0], O[1], O[3] - 1, O[3], O[4], O[5]);
  if (nID[0] >= 0) add(O[0], O[0] + 1, I[2], I[3], O[4], O[5]);
  if (nID[1] >= 0) add(O[1] - 1, O[1], I[2], I[3], O[4], O[5]);
  if (nID[4] >= 0) add(I[0], I[1], I[2], I[3], O[4], O[4] + 1);
  if (nID[5] >= 0) add(I[0], I[1], I[2], I[3], O[5] - 1, O[5]);
  return n;
}

int pair_probe(const REAL* u, REAL* w, const REAL* b, const int* sz, const int* idx, const int* idx1, int g) {
  ensure_init();
  if (!ctx.tune.fuse_fin || g < 2) return 0;
  const Box bx = make_box(sz, idx, g);
  if (bx.empty) return 0;
  const Box ba = make_box(sz, idx1, g);
  return launch_jacobi2<0>(u, b, w, Coef(), bx, ba, nullptr, Fin2(), 0, 0, true) ? 1 : 0;
}

void pair_shell_async(const REAL* u, REAL* w, const REAL* b, const int* sz, const int* idx1_brick, const int* boxes, int n, int g,
                      const REAL* cf, REAL omg, int rb, const int* skip) {
  ensure_init();
  const Box ba = make_box(sz, idx1_brick, g);
  if (rb >= 0) launch_pair_shell<1>(u, b, w, make_coef(cf, omg), sz, g, ba, boxes, n, rb, skip);
  else launch_pair_shell<0>(u, b, w, make_coef(cf, omg), sz, g, ba, boxes, n, 0, skip);
}

int pair_box_async(const REAL* u, REAL* w, const REAL* b, const int* sz, const int* idx, const int* idx1, int g, const REAL* cf,
                   REAL omg, int rb, double* res_dev, int with_shell, const int* skip) {
  ensure_init();
  const Box bx = make_box(sz, idx, g);
  const Box ba = make_box(sz, idx1, g);
  Fin2 fin;
  fin.dst = res_dev;
  fin.single = rb >= 0;
  if (with_shell) fin.extra = ctx.shell_partials, fin.n_extra = ctx.shell_pending;
  ctx.shell_pending = 0;
  if (rb >= 0) return launch_jacobi2<1>(u, b, w, make_coef(cf, omg), bx, ba, skip, fin, rb) ? 1 : 0;
  return launch_jacobi2<0>(u, b, w, make_coef(cf, omg), bx, ba, skip, fin) ? 1 : 0;
}

void copy_shell_async(REAL* dst, const REAL* src, const int* sz, const int* idx, int g) {
  ensure_init();
  const int nkp = sz[2] + 2 * g, nip = sz[0] + 2 * g, njp = sz[1] + 2 * g;
  const long long rows = (long long)nip * njp;
  hipLaunchKernelGGL(copy_shell_k, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, ctx.stream, dst, src, nkp, nip, njp,
                     idx[4] + g - 1, idx[5] + g - 1, idx[0] + g - 1, idx[1] + g - 1, idx[2] + g - 1, idx[3] + g - 1);
  HIP_CHECK(hipGetLastError());
}
void copy_inner_async(REAL* dst, const REAL* src, const int* sz, const int* idx, int g) {
  launch_ewise<OP_COPY>(dst, src, nullptr, (REAL)0, (REAL)0, make_box(sz, idx, g));
}
void bc_async(const int* sz, int g, REAL* p, REAL dh, const REAL* org, const int* nID, int ioff, int joff) {
  // same launches as bc_k_ without the trailing synchronisation
  ensure_init();
  const int ix = sz[0], jx = sz[1], kx = sz[2];
  const int nkp = kx + 2 * g, nip = ix + 2 * g;
  if (nID[4] < 0 || nID[5] < 0) {
    const REAL* tab = bc_table(ix, jx, dh, org, ioff, joff);
    dim3 grid((ix + 127) / 128, jx);
    if (nID[4] < 0) hipLaunchKernelGGL(bc_kface_k, grid, dim3(128), 0, ctx.stream, p, tab, ix, jx, 1, g, nkp, nip);
    if (nID[5] < 0) hipLaunchKernelGGL(bc_kface_k, grid, dim3(128), 0, ctx.stream, p, tab, ix, jx, kx, g, nkp, nip);
  }
  {
    dim3 grid((kx + 127) / 128, jx);
    if (nID[0] < 0) hipLaunchKernelGGL(bc_iface_k, grid, dim3(128), 0, ctx.stream, p, jx, kx, 1, g, nkp, nip);
    if (nID[1] < 0) hipLaunchKernelGGL(bc_iface_k, grid, dim3(128), 0, ctx.stream, p, jx, kx, ix, g, nkp, nip);
  }
  {
    dim3 grid((kx + 127) / 128, ix);
    if (nID[2] < 0) hipLaunchKernelGGL(bc_jface_k, grid, dim3(128), 0, ctx.stream, p, ix, kx, 1, g, nkp, nip);
    if (nID[3] < 0) hipLaunchKernelGGL(bc_jface_k, grid, dim3(128), 0, ctx.stream, p, ix, kx, jx, g, nkp, nip);
  }
  HIP_CHECK(hipGetLastError());
}
}  // namespace czhip_internal
