// cz_k_rb4.h -- part of cz_kernels.hip (ONE translation unit per precision; included inside its anonymous namespace after cz_k_pair2.h):
// rb4_k, TWO complete red-black SOR iterations (colour 0, colour 1, colour 0, colour 1; cz_solver.f90:404-493 four times,
// cz_Poisson.cpp:205-209 twice) per pass over memory.  Single-domain runs, constant coefficients.
// ------------------------------------------------------------------------------------------------------------
// Why: the fused red-black iteration (jacobi2p_k<RB = 1>) is bound by memory -- 5.1 TB/s of real traffic with the vector ALU 61 % busy
// (profiles/r03/pmc_jacobi2p_512_f32_rb_SQ.txt): a stage evaluates only the active colour.  Four stages per pass halve the bytes per
// iteration.  Round 3 modelled this at 1.16 x and did not build it because eight plane buffers with whole-row halos left 4.9 rows per
// segment; with the k windows of round 4 (Geom2) a row of the workgroup's view is ~34 vectors whatever the grid, and the budget closes:
//     fields      u on E4 = own segment +- 4 rows, f1 (after colour 0) on E3, f2 (after colour 1) on E2, f3 on E1, f4 = output on the segment
//     threads     one per vector of E3 (TB = LV = S + 6R: every thread evaluates stage 1; stages 2 and 3 on all of E3 as well -- what lies
//                 outside their sets is never read by a valid point -- stage 4 on the owned vectors)
//     LDS         u: 2 x (LV + 2R), f1..f3: 2 x LV each (E3 coordinates), one barrier per plane step   = 133 KB for R = 34
//     k windows   KT vectors + hv halo vectors per side; four stages reach 3 elements beyond a window: hv = 1 (FP32), 2 (FP64)
//     planes      stage s at step q works on plane q - s + 1: f1(q) from u(q-1..q+1), f2(q-1) from f1, f3(q-2), f4(q-3) -> W
// Same per-point operations on the same values as four psor2sma_core_ calls => the same bits (relax_vec_rb is the stage of jacobi2p_k).
// Threads -> vectors: which components of a vector a stage updates depends on the parity of its ROW (k of component 0 is even in every window);
// E3 is therefore dealt out by row parity -- threads 0 .. nA-1 take the first row of E3 and every second one after it, the others the rows in
// between -- so that all lanes of a wave (one wave of the sixteen excepted) agree on the colour offset and a stage BRANCHES on it instead of
// selecting every operand per lane (relax_vec_rb_branch): the selects were a quarter of the vector instructions of a kernel whose vector ALU
// is 93 % busy (profiles/r04/rb4_two_iterations_per_pass.txt).  LDS and memory are indexed by the vector (x), not by the thread.
// Both residuals are produced (iteration n+1 = stages 1 + 2, iteration n+2 = stages 3 + 4); if the FIRST of the two iterations converges
// the driver re-runs that single iteration from the untouched input (out of place, like the Jacobi pair).
// ------------------------------------------------------------------------------------------------------------
template <int V, int TB, int UNIT>
__global__ void __launch_bounds__(TB, 1)
rb4_k(const REAL* __restrict__ U, const REAL* __restrict__ B, REAL* __restrict__ W, Coef c, Geom2 g, double* partials,
      const int* __restrict__ skip, Fin2 fin) {
  if (skip != nullptr && *skip != 0) return;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int t = threadIdx.x;
  const int R = g.R;
  constexpr int LV = TB;      // E3: own segment +- three rows (S = LV - 6R); one vector per thread
  const int LU = LV + 2 * R;  // E4
  // (R vectors of padding in front of the u buffers and behind the last field buffer: stages 2 and 3 are evaluated on all of E3 and read
  // +-R beyond their sets -- inside the allocation, never used)
  Vec<V>* ldsU = reinterpret_cast<Vec<V>*>(smem) + R;   // 2 buffers of LU vectors: plane p in buffer p & 1
  Vec<V>* ldsF = ldsU + (size_t)2 * LU;                  // f1, f2, f3: [field][plane & 1][LV]
  double* wsum = reinterpret_cast<double*>(ldsF + (size_t)6 * LV + R);

  // ---- workgroup -> (window, segment, chunk): as jacobi2p_k
  const int lb = blockIdx.x;
  const int nblk = gridDim.x;
  int seg, chunk;
  if (g.map != nullptr) {
    seg = g.map[2 * lb];
    chunk = g.map[2 * lb + 1];
  } else {
    const int x = lb & 7, r = lb >> 3;
    const int base = g.nseg >> 3, rem = g.nseg & 7, bmax = base + (rem ? 1 : 0);
    const int blen = base + (x < rem ? 1 : 0);
    const int sl = r % bmax;
    chunk = r / bmax;
    seg = (sl < blen) ? x * base + min(x, rem) + sl : g.nseg;  // nseg = no work
  }
  int win = 0;
  if (seg < g.nseg) {
    win = seg / g.nsegw;
    seg -= win * g.nsegw;
  } else {
    seg = g.nsegw;
  }
  const int kw0 = win * g.KW - g.hv * V;
  const long long fb = (seg < g.nsegw) ? g.F0 + (long long)seg * g.S : g.Fend;
  const int ja = g.jj0 + chunk * g.TJ;
  int jb = ja + g.TJ - 1;
  if (jb > g.jj1) jb = g.jj1;

  double acc1 = 0.0, acc2 = 0.0;
  const HoistedDiv dv{fastdiv_init(c.dd)};

  if (ja <= jb && fb < g.Fend) {
    const long long e3_0 = fb - 3 * (long long)R;  // first vector of E3
    const long long vlast = g.PSV - 1;
    const size_t PB = (size_t)g.PSB;
    auto off_of = [&](long long f) -> unsigned {  // (see jacobi2p_k: clamped below the plane, not beyond it; `lim` for the array's last plane)
      if (f < 0) f = 0;
      if (f > vlast) f = vlast;
      const long long r = f / R;
      long long el = r * g.nkp + kw0 + (f - r * R) * V;
      el = el < 0 ? 0 : el;
      return (unsigned)(el * (long long)sizeof(REAL));
    };
    auto lim = [&](unsigned off, int plane) -> unsigned { return plane == g.jlast ? (off < g.last_off ? off : g.last_off) : off; };
    auto pl = [&](int p) -> int { return p < 0 ? 0 : (p > g.jlast ? g.jlast : p); };  // planes beyond the array are never used: clamped
    // this thread's vector: x = its index in E3, rows dealt by parity (see the head of the file)
    int x;
    {
      const int off = (int)(((e3_0 % R) + R) % R);
      const int L0 = off == 0 ? R : R - off;  // vectors of the first row of E3
      int nA = L0;                            // vectors in the first row and every second row after it
      for (int st = L0 + R; st < LV; st += 2 * R) nA += min(R, LV - st);
      if (t < nA) {
        const int m = t - L0, j = m / R;
        x = t < L0 ? t : L0 + R + 2 * R * j + (m - j * R);
      } else {
        const int n = t - nA, j = n / R;
        x = L0 + 2 * R * j + (n - j * R);
      }
    }
    const long long f = e3_0 + x;
    const unsigned bo = off_of(f);
    unsigned inbox = 0;  // components inside the inner box (every stage updates only those)
    unsigned own = 0;    // ... of a vector this workgroup owns (stores, residual counts)
    int pbase;
    {
      const long long fc = f < 0 ? 0 : f;
      const long long row = fc / R;
      const int kv = (int)(fc - row * R);
      const int kb = kw0 + kv * V;
      unsigned bits = 0;
#pragma unroll
      for (int cc = 0; cc < V; cc++) {
        const int kk = kb + cc;
        if (kk >= g.kk0 && kk <= g.kk1) bits |= 1u << cc;
      }
      pbase = kb + (int)row + g.par;
      const bool rows_in = f >= g.F0 && f < g.Fend;
      inbox = rows_in ? bits : 0u;
      const bool kown = kv >= g.hv && kv < g.hv + g.KT;
      own = (x >= 3 * R && x < LV - 3 * R && rows_in && kown) ? bits : 0u;
    }
    // the outer rows of E4: the first R threads stage the lower one, the last R threads the upper one
    const bool has_halo = (t < R) || (t >= TB - R);
    const int hl = (t < R) ? t : (LV + R + (t - (TB - R)));  // index inside an LDS u buffer (E4 coordinates)
    const unsigned hbo = off_of(has_halo ? (fb - 4 * (long long)R + hl) : f);
    const char* Ub = reinterpret_cast<const char*>(U);
    const char* Bb = reinterpret_cast<const char*>(B);
    char* Wb = reinterpret_cast<char*>(W);

    // ---- prologue: the first step is q0 = ja - 3 (stage 1 on plane ja - 3): LDS u(q0 - 1) own, u(q0) on E4; in flight u(q0 + 1), b(q0)
    const int q0 = ja - 3;
    Vec<V> uA, uB, bA, bB, hx;
    {
      const Vec<V> t2 = ld16<V>(Ub + (size_t)pl(q0 - 1) * PB, lim(bo, pl(q0 - 1)));
      const Vec<V> t1 = ld16<V>(Ub + (size_t)pl(q0) * PB, lim(bo, pl(q0)));
      const Vec<V> h1 = ld16<V>(Ub + (size_t)pl(q0) * PB, lim(hbo, pl(q0)));
      uA = ld16<V>(Ub + (size_t)pl(q0 + 1) * PB, lim(bo, pl(q0 + 1)));
      bA = ld16<V>(Bb + (size_t)pl(q0) * PB, lim(bo, pl(q0)));
      ldsU[(size_t)((q0 - 1) & 1) * LU + R + x] = t2;
      ldsU[(size_t)(q0 & 1) * LU + R + x] = t1;
      if (has_halo) ldsU[(size_t)(q0 & 1) * LU + hl] = h1;
    }
    Vec<V> bq1 = zerov<V>(), bq2 = zerov<V>(), bq3 = zerov<V>();  // b of the planes of stages 2, 3, 4
    __syncthreads();

    // the k neighbours beyond the wave's first and last vector come from LDS (one element each, the same address in all lanes)
    const int x_first = __builtin_amdgcn_readfirstlane(x), x_last = __builtin_amdgcn_readlane(x, 63);
    // One stage in two halves, so that the operands of the NEXT stage are requested from LDS before this stage's arithmetic starts (the stages of
    // a step depend on each other only through one register vector): `fetch` reads the field (centre plane p in LDS buffer `cur`, own vector of
    // plane p-1 in `prv`), `compute` makes the stage's result from them and plane p+1 (`nxt`).  A plane outside the inner box is left alone by an
    // empty mask (the result is then the centre value), not by a branch around the stage.
    struct Ops {
      Vec<V> pc, im, ip;  // (the own vector of plane p-1 is read by `compute`: 128 registers are all a thread of 1 024 has)
      REAL elo, ehi;
    };
    auto fetch = [&](const Vec<V>* cur) __attribute__((always_inline)) -> Ops {
      Ops o;
      o.pc = lds_ld<V>(cur + x);
      o.im = lds_ld<V>(cur + x - R);
      o.ip = lds_ld<V>(cur + x + R);
      o.elo = reinterpret_cast<const REAL*>(cur)[(long long)x_first * V - 1];
      o.ehi = reinterpret_cast<const REAL*>(cur)[(long long)(x_last + 1) * V];
      return o;
    };
    auto compute = [&](const Ops& o, const Vec<V>* cur, const Vec<V>* prv, const Vec<V>& nxt, const Vec<V>& bb, int p, int colour, unsigned msk, unsigned cnt, double& acc,
                       const bool first = false) __attribute__((always_inline)) -> Vec<V> {
      const Vec<V> pm = lds_ld<V>(prv + x);
      const REAL kl = lane_shr1(o.elo, o.pc.v[V - 1]);
      REAL kr = lane_shl1(o.ehi, o.pc.v[0]);
      // The last vector of E3 is not the last lane of a wave (rows are dealt by parity), and stage 1 needs its true neighbour -- the first
      // vector of E4's upper row: the corner of the dependence cone of the last owned vector runs through it.  (E3's first vector is thread 0,
      // lane 0: `elo`.  In the later stages both ends lie outside the stage's set.)
      if (first) kr = (x == LV - 1) ? reinterpret_cast<const REAL*>(cur)[(long long)LV * V] : kr;
      const bool sc = ((pbase + p + colour) & 1) != 0;
      return relax_vec_rb_branch<V>(o.pc, o.im, o.ip, pm, nxt, kl, kr, bb, sc, msk, cnt, acc,
                             [&](REAL pp, REAL ipv, REAL imv, REAL pnv, REAL pmv, REAL kp1, REAL km1, REAL bv, REAL, REAL) {
                               const REAL ss = offdiag_sum<UNIT>(c, ipv, imv, pnv, pmv, kp1, km1);
                               return (dv(ss - bv) - pp) * c.omg;
                             });
    };

    // One plane step q.  uc = u(q+1) and b1 = b(q) were requested one step ago; un / bn receive this step's requests.
    auto step = [&](const int q, Vec<V>& uc, Vec<V>& un, Vec<V>& b1, Vec<V>& bn) __attribute__((always_inline)) {
      {
        const int qu = pl(q + 2 <= jb + 4 ? q + 2 : jb + 4), qb = pl(q + 1 <= jb + 3 ? q + 1 : jb + 3);
        hx = ld16<V>(Ub + (size_t)pl(q + 1) * PB, lim(hbo, pl(q + 1)));
        un = ld16<V>(Ub + (size_t)qu * PB, lim(bo, qu));
        bn = ld16<V>(Bb + (size_t)qb * PB, lim(bo, qb));
      }
      // planes of the four stages; masks: inside the inner box a stage updates, inside the chunk the owner counts the residual
      const int p1 = q, p2 = q - 1, p3 = q - 2, p4 = q - 3;
      auto upd = [&](int p) -> unsigned { return (p >= g.jj0 && p <= g.jj1) ? inbox : 0u; };
      auto cnt = [&](int p) -> unsigned { return (p >= ja && p <= jb) ? own : 0u; };
      const Vec<V>* cU = ldsU + (size_t)(p1 & 1) * LU + R;  // u(p1) in E3 coordinates (index x)
      const Vec<V>* pU = ldsU + (size_t)((p1 - 1) & 1) * LU + R;
      Vec<V>* F1 = ldsF;
      Vec<V>* F2 = ldsF + (size_t)2 * LV;
      Vec<V>* F3 = ldsF + (size_t)4 * LV;
      const Ops o1 = fetch(cU);
      const Ops o2 = fetch(F1 + (size_t)(p2 & 1) * LV);
      // ---- stage 1: f1(p1), colour 0, every vector of E3
      const Vec<V> v1 = compute(o1, cU, pU, uc, b1, p1, 0, upd(p1), cnt(p1), acc1, true);
      const Ops o3 = fetch(F2 + (size_t)(p3 & 1) * LV);
      // ---- stage 2: f2(p2), colour 1, from f1(p2 - 1) [LDS], f1(p2) [LDS], f1(p1) [v1]
      const Vec<V> v2 = compute(o2, nullptr, F1 + (size_t)((p2 - 1) & 1) * LV, v1, bq1, p2, 1, upd(p2), cnt(p2), acc1);
      const Ops o4 = fetch(F3 + (size_t)(p4 & 1) * LV);
      // ---- stage 3: f3(p3), colour 0
      const Vec<V> v3 = compute(o3, nullptr, F2 + (size_t)((p3 - 1) & 1) * LV, v2, bq2, p3, 0, upd(p3), cnt(p3), acc2);
      // ---- stage 4: f4(p4) = the output, colour 1, owned vectors of the chunk's planes.  Whole waves: the k neighbours travel by lane
      // shifts, and the lane next to the first owned vector of a window holds a halo vector -- it owns nothing but must take part.
      if (p4 >= ja && __builtin_amdgcn_ballot_w64(own != 0) != 0ull) {
        const Vec<V> o = compute(o4, nullptr, F3 + (size_t)((p4 - 1) & 1) * LV, v3, bq3, p4, 1, own, own, acc2);
        char* Wq = Wb + (size_t)p4 * PB;
        if (own == (1u << V) - 1) {
          st16<V>(Wq, bo, o);
        } else if (own != 0) {
          REAL* wp = reinterpret_cast<REAL*>(Wq + bo);
#pragma unroll
          for (int cc = 0; cc < V; cc++)
            if (own & (1u << cc)) wp[cc] = o.v[cc];
        }
      }
      // ---- publish: f1(p1), f2(p2), f3(p3) and the next u centre plane u(q+1) with its outer rows
      F1[(size_t)(p1 & 1) * LV + x] = v1;
      F2[(size_t)(p2 & 1) * LV + x] = v2;
      F3[(size_t)(p3 & 1) * LV + x] = v3;
      Vec<V>* nU = ldsU + (size_t)((p1 + 1) & 1) * LU;
      nU[R + x] = uc;
      if (has_halo) nU[hl] = hx;
      bq3 = bq2, bq2 = bq1, bq1 = b1;  // (b1 is complete: stage 1 used it)
      __syncthreads();
    };
    // (the planes a stage publishes before its first needed one are pass-through copies or garbage that no later stage reads: stage s first
    // matters at plane ja - (4 - s), which it reaches at step ja - 3 + 2 (s - 1) ... every field buffer a valid point reads was written by
    // a step of this loop)
    for (int q = q0;; q += 2) {
      step(q, uA, uB, bA, bB);
      if (q + 1 > jb + 3) break;
      step(q + 1, uB, uA, bB, bA);
      if (q + 2 > jb + 3) break;
    }
  }

  // ---- residuals of the two iterations: per-workgroup partials, finalised by the last workgroup (see jacobi2p_k)
  __syncthreads();
  const double s1 = block_sum<TB>(acc1, wsum);
  __syncthreads();
  const double s2 = block_sum<TB>(acc2, wsum);
  int* last_flag = reinterpret_cast<int*>(wsum + 16);
  if (t == 0) {
    __hip_atomic_store(&partials[lb], s1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(&partials[nblk + lb], s2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    *last_flag = arrive_and_test_last(fin.counter, nblk);
  }
  __syncthreads();
  if (*last_flag) pair_finalize<TB>(partials, nblk, fin, wsum);
}
