// cz_k_pair2.h -- part of cz_kernels.hip (ONE translation unit per precision; included inside its anonymous namespace after
// cz_k_pair.h): jacobi2p_k, the software-pipelined form of the two-stage pass (two Jacobi sweeps / one red-black iteration
// per pass over memory; cz_solver.f90:334-351, 466-480).
// ------------------------------------------------------------------------------------------------------------
// Same tiling, same per-point arithmetic and the same results as jacobi2_k (cz_k_pair.h) -- what changes is WHEN things move:
//   * the loads of plane q+2 of u and of plane q+1 of b are issued at the top of step q and consumed in step q+1 (one plane
//     of prefetch).  jacobi2_k issued the loads of a step and waited for them a few instructions later, and the compiler's
//     s_waitcnt vmcnt(0) in front of the LDS publish drained the step's W stores as well: a memory round trip and a store
//     round trip on the critical path of every plane (profiles/r02/isa_notes.md).  Here every global access of the main loop
//     is unconditional (clamped addresses instead of predicates), so the compiler can count the younger loads and wait with
//     vmcnt(N > 0).
//   * the registers for that come from the j-1 operands: u(q-1) and v(q-2) of the thread's own vectors are re-read from the
//     LDS buffer that still holds them (same thread wrote them, same thread overwrites them later in program order) instead
//     of being carried in two register queues.
//   * the vector-crossing k neighbours come from the lane next door (DPP wave shift) with one broadcast LDS read per wave for
//     the lane at the end of the wave, instead of two stride-4 ds_read_b32 per vector (4-way bank conflicts, 32 % of all LDS
//     cycles in jacobi2_k, profiles/r01/pmc_jacobi2_512_f32.csv).
// LDS: u(q) on E2 = own segment +- 2 rows and u(q-1), v(q-1) on E1 = own +- 1 row and v(q-2); two buffers each, one barrier
// per plane.
// ------------------------------------------------------------------------------------------------------------
// global accesses of the pass: a vector is REAL-aligned in memory (16-byte aligned where the row length is a multiple of the vector width)
template <int V>
__device__ __forceinline__ Vec<V> ld16(const char* plane, unsigned byte_off) {
  typedef typename NatVec<V>::type nv;
  typedef nv unv __attribute__((aligned(sizeof(REAL))));
  const nv x = *reinterpret_cast<const unv*>(plane + byte_off);
  Vec<V> r;
  __builtin_memcpy(&r, &x, sizeof(r));
  return r;
}
template <int V>
__device__ __forceinline__ void st16(char* plane, unsigned byte_off, const Vec<V>& x) {
  typedef typename NatVec<V>::type nv;
  typedef nv unv __attribute__((aligned(sizeof(REAL))));
  nv y;
  __builtin_memcpy(&y, &x, sizeof(y));
  *reinterpret_cast<unv*>(plane + byte_off) = y;
}

// one ds_read_b128 per vector: left to itself the compiler re-reads overlapping pieces of a vector with ds_read_b32 / ds_read2_b32
// (operand pairs for packed FP32 math) -- stride-16-byte scalar reads, i.e. the 4-way bank conflicts this kernel set out to remove
template <int V>
__device__ __forceinline__ Vec<V> lds_ld(const Vec<V>* p) {
  typedef typename NatVec<V>::type nv;
  nv x = *reinterpret_cast<const nv*>(p);
  asm("" : "+v"(x));  // the value is needed whole, in consecutive registers: keeps the read one ds_read_b128
  Vec<V> r;
  __builtin_memcpy(&r, &x, sizeof(r));
  return r;
}

// value of `x` in the previous (SHR) / next (SHL) lane of the wave; lane 0 / lane 63 receive `edge`
__device__ __forceinline__ float lane_shr1(float edge, float x) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, edge), __builtin_bit_cast(int, x), 0x138, 0xf, 0xf, false));
}
__device__ __forceinline__ float lane_shl1(float edge, float x) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, edge), __builtin_bit_cast(int, x), 0x130, 0xf, 0xf, false));
}
__device__ __forceinline__ double lane_shr1(double edge, double x) {
  const long long e = __builtin_bit_cast(long long, edge), v = __builtin_bit_cast(long long, x);
  const int lo = __builtin_amdgcn_update_dpp((int)e, (int)v, 0x138, 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp((int)(e >> 32), (int)(v >> 32), 0x138, 0xf, 0xf, false);
  return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned)lo);
}
__device__ __forceinline__ double lane_shl1(double edge, double x) {
  const long long e = __builtin_bit_cast(long long, edge), v = __builtin_bit_cast(long long, x);
  const int lo = __builtin_amdgcn_update_dpp((int)e, (int)v, 0x130, 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp((int)(e >> 32), (int)(v >> 32), 0x130, 0xf, 0xf, false);
  return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned)lo);
}

// Red-black stage: only every other component of a vector belongs to the colour being updated -- component s, s+2 (FP32) / s (FP64), with
// s = (k + i + j + parity) & 1 of component 0, a per-lane value.  The operands of those components are selected first and the update is
// evaluated for V/2 points instead of V (the other colour passes through): same arithmetic on the updated points, hence the same bits,
// at roughly half the vector instructions of the stage.  `point(pp, ip, im, pn, pm, kp1, km1, bb, zt, ztt)` returns dp.
template <int V>
__device__ __forceinline__ typename NatVec<V>::type as_native(const Vec<V>& a) {
  typename NatVec<V>::type x;
  __builtin_memcpy(&x, &a, sizeof(x));
  return x;
}

template <int V, class F>
__device__ __forceinline__ Vec<V> relax_vec_rb(const Vec<V>& pc_, const Vec<V>& im_, const Vec<V>& ip_, const Vec<V>& pm_, const Vec<V>& pn_,
                                               REAL kl, REAL kr, const Vec<V>& bb_, const Vec<V>& zt_, const Vec<V>& ztt_, bool s, unsigned mask,
                                               unsigned count_mask, double& acc, const F& point) {
  static_assert(V == 2 || V == 4, "vector of two or four components");
  // (native vector VALUES: a select between two elements of an in-memory array is turned into an indexed load, which sends the
  // register arrays of the whole kernel to scratch)
  const auto pc = as_native<V>(pc_), im = as_native<V>(im_), ip = as_native<V>(ip_), pm = as_native<V>(pm_), pn = as_native<V>(pn_),
             bb = as_native<V>(bb_), zt = as_native<V>(zt_), ztt = as_native<V>(ztt_);
  auto o = pc;
  const unsigned ms = s ? (mask >> 1) : mask, cs = s ? (count_mask >> 1) : count_mask;  // bit 2a: slot a
#pragma unroll
  for (int a = 0; a < V / 2; a++) {
    const int c0 = 2 * a, c1 = 2 * a + 1;  // the slot's component when s = 0 / s = 1
    const REAL pp = s ? pc[c1] : pc[c0];
    const REAL km1 = s ? pc[c0] : (c0 == 0 ? kl : pc[c0 > 0 ? c0 - 1 : 0]);
    const REAL kp1 = s ? (c1 == V - 1 ? kr : pc[c1 < V - 1 ? c1 + 1 : V - 1]) : pc[c1];
    const REAL dp = point(pp, s ? ip[c1] : ip[c0], s ? im[c1] : im[c0], s ? pn[c1] : pn[c0], s ? pm[c1] : pm[c0], kp1, km1,
                          s ? bb[c1] : bb[c0], s ? zt[c1] : zt[c0], s ? ztt[c1] : ztt[c0]);
    const REAL d2 = dp * dp;
    const REAL nw = (ms & (1u << c0)) ? pp + dp : pp;
    o[c0] = s ? pc[c0] : nw;
    o[c1] = s ? nw : pc[c1];
    acc += (double)((cs & (1u << c0)) ? d2 : (REAL)0);
  }
  Vec<V> r;
  __builtin_memcpy(&r, &o, sizeof(r));
  return r;
}

// The same stage where the colour offset s is (nearly always) the same in all lanes of a wave -- rb4_k deals its vectors so that a wave holds rows
// of one parity: a branch on s instead of selects.  With s a constant of each side the operand selects of relax_vec_rb (8 per point, a quarter of
// the stage's vector instructions) fold away; a wave whose lanes disagree runs both sides under their masks.  Same operations on the same values.
template <int V, int S, class F>
__device__ __forceinline__ typename NatVec<V>::type relax_rb_side(const typename NatVec<V>::type& pc, const typename NatVec<V>::type& im,
                                                                  const typename NatVec<V>::type& ip, const typename NatVec<V>::type& pm,
                                                                  const typename NatVec<V>::type& pn, REAL kl, REAL kr,
                                                                  const typename NatVec<V>::type& bb, unsigned mask, unsigned count_mask, double& acc,
                                                                  const F& point) {
  auto o = pc;
#pragma unroll
  for (int a = 0; a < V / 2; a++) {
    const int cc = 2 * a + S;  // the slot's component
    const REAL pp = pc[cc];
    const REAL km1 = (cc == 0) ? kl : pc[cc > 0 ? cc - 1 : 0];
    const REAL kp1 = (cc == V - 1) ? kr : pc[cc < V - 1 ? cc + 1 : V - 1];
    const REAL dp = point(pp, ip[cc], im[cc], pn[cc], pm[cc], kp1, km1, bb[cc], (REAL)0, (REAL)0);
    const REAL d2 = dp * dp;
    o[cc] = (mask & (1u << cc)) ? pp + dp : pp;
    acc += (double)((count_mask & (1u << cc)) ? d2 : (REAL)0);
  }
  return o;
}
template <int V, class F>
__device__ __forceinline__ Vec<V> relax_vec_rb_branch(const Vec<V>& pc_, const Vec<V>& im_, const Vec<V>& ip_, const Vec<V>& pm_, const Vec<V>& pn_,
                                                      REAL kl, REAL kr, const Vec<V>& bb_, bool s, unsigned mask, unsigned count_mask, double& acc,
                                                      const F& point) {
  static_assert(V == 2 || V == 4, "vector of two or four components");
  const auto pc = as_native<V>(pc_), im = as_native<V>(im_), ip = as_native<V>(ip_), pm = as_native<V>(pm_), pn = as_native<V>(pn_),
             bb = as_native<V>(bb_);
  typename NatVec<V>::type o;
  if (s) o = relax_rb_side<V, 1>(pc, im, ip, pm, pn, kl, kr, bb, mask, count_mask, acc, point);
  else o = relax_rb_side<V, 0>(pc, im, ip, pm, pn, kl, kr, bb, mask, count_mask, acc, point);
  Vec<V> r;
  __builtin_memcpy(&r, &o, sizeof(r));
  return r;
}

// ZU = 1: the input field is identically zero and is not read (the first pair of a preconditioner solve).
// MAF = 1: weights from the 1-D coordinate arrays `ma` (device copies; index = padded index for g = 2) instead of c.
// BS = 1 | 2 (with ZU = 1): the right-hand side is not read but MADE, point by point, from the vectors the preceding element-wise update of
// BiCGSTAB would have made it from -- 1: b = a*x + y (blas_triad, cz_blas.f90:297; s = r - alpha q), 2: b = x + a*(z - b*y) (blas_bicg_1, :490;
// p = r + beta (p - omega q)) -- with the same operations on the same values, and the workgroup that owns a vector writes it to bs.out (the
// array the later passes of the solve read as b): the update kernel and one read of its result are saved (BSrc, cz_k_pair.h).
// PRE = n > 0 (small grids, round 4): chunks of at most n planes whose operands -- the n + 4 planes of u with their outer rows and the n + 2
// planes of b -- are ALL requested before the first plane step, into registers.  The pipelined form asks for a plane one step ahead, which
// hides the memory latency where a CU holds enough waves and the chunks are long; on a grid of 64^3 .. 128^3 cells every workgroup of the
// pass is resident at once, a chunk is two or three planes, and the launch lasts as long as ONE workgroup's chain of dependent round trips:
// three in the prologue and one per step, 16-24 us for 2-4 us of work (profiles/r03/small_grids_chunk_length.txt).  With everything in flight
// at once it is one round trip, then n + 2 steps of LDS and arithmetic.  Same operations on the same values: same bits.
template <int V, int TB, int MV, int RB, int ZU, int MAF = 0, int BS = 0, int PRE = 0, int UNIT = 0>
__global__ void __launch_bounds__(TB, (TB == 512 && !MAF && !PRE) ? 4 : 1)  // (MAF, 512 threads: 256 registers instead of 21-30 spilled)
jacobi2p_k(const REAL* __restrict__ U, const REAL* __restrict__ B, REAL* __restrict__ W, Coef c, Geom2 g, double* partials,
           const int* __restrict__ skip, Fin2 fin, MafArgs ma, BSrc bs) {
  static_assert(BS == 0 || (ZU == 1 && MAF == 0), "a made right-hand side belongs to the first pass of a preconditioner solve");
  static_assert(PRE == 0 || (ZU == 0 && MAF == 0 && BS == 0), "the preloaded form is the plain pass");
  static_assert(UNIT == 0 || MAF == 0, "unit coefficients are constant coefficients");
  if (skip != nullptr && *skip != 0) return;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int t = threadIdx.x;
  const int R = g.R;
  constexpr int LV = TB * MV;       // E1: own segment +- one row (S = LV - 2R)
  const int LU = LV + 2 * R;        // E2: own segment +- two rows
  Vec<V>* ldsU = reinterpret_cast<Vec<V>*>(smem);                   // 2 buffers of LU vectors
  Vec<V>* ldsV = ldsU + (size_t)2 * LU;                              // 2 buffers of LV vectors
  double* wsum = reinterpret_cast<double*>(ldsV + (size_t)2 * LV);   // 16 doubles + flag
  REAL* ztab = reinterpret_cast<REAL*>(wsum + 18);                   // MAF: ZT[k], then ZTT[k], k = 0 .. nkp-1

  int lb = blockIdx.x;
  const int nblk = gridDim.x;
  int seg, chunk;
  if (g.map != nullptr) {
    // balanced shares: the (segment, chunk) items in segment-major order are cut into eight equal runs, one per XCD (the hardware deals
    // workgroup ids round-robin over the XCDs), each walked chunk by chunk -- see pair_xcd_map
    seg = g.map[2 * lb];
    chunk = g.map[2 * lb + 1];
  } else {
    // XCD bands: XCD x owns a contiguous band of whole segments of every chunk (row-adjacent segments share their halo rows in one L2)
    // and walks it chunk by chunk
    const int x = lb & 7, r = lb >> 3;
    const int base = g.nseg >> 3, rem = g.nseg & 7, bmax = base + (rem ? 1 : 0);
    const int blen = base + (x < rem ? 1 : 0);
    const int sl = r % bmax;
    chunk = r / bmax;
    seg = (sl < blen) ? x * base + min(x, rem) + sl : g.nseg;  // nseg = no work
  }
  // k window of this segment (Geom2): ids are window-major, so the band / run of an XCD is a set of row-adjacent segments of ONE window
  int win = 0;
  if (g.nwin > 1 && seg < g.nseg) {
    win = seg / g.nsegw;
    seg -= win * g.nsegw;
  } else if (g.nwin > 1) {
    seg = g.nsegw;  // no work
  }
  const int nseg_w = (g.nwin > 1) ? g.nsegw : g.nseg;
  const int kw0 = win * g.KW - g.hv * V;  // element of the row that vector 0 of the window's view starts at (-V: the left halo of window 0)
  const long long fb = (seg < nseg_w) ? g.F0 + (long long)seg * g.S : g.Fend;
  const int ja = g.jj0 + chunk * g.TJ;
  int jb = ja + g.TJ - 1;
  if (jb > g.jj1) jb = g.jj1;

  double acc1 = 0.0, acc2 = 0.0;
#ifdef CZ_P2_PLAIN_DIV  // tools/pair_lab A/B only
  const PlainDiv dv{c.dd};
#elif defined(CZ_P2_SHORT_DIV)
  const ShortDiv dv{fastdiv_init(c.dd)};
#elif defined(CZ_P2_MEDIUM_DIV)
  const MediumDiv dv{fastdiv_init(c.dd)};
#else
  const HoistedDiv dv{fastdiv_init(c.dd)};  // exact IEEE quotients, the divisor's share of the work done once (cz_k_fastdiv.h)
#endif

  if (ja <= jb && fb < g.Fend) {
    const long long e1_0 = fb - R;      // first vector of E1
    const long long e2_0 = fb - 2 * R;  // first vector of E2
    const long long vlast = g.PSV - 1;
    const size_t PB = (size_t)g.PSB;  // bytes per plane
    // byte offset of vector f of the (window's) row view inside a plane in memory.  Below the plane (row 0's left halo vector of window 0:
    // masked, and no unmasked point reads it) it is clamped to 0.  Beyond the plane's end it is NOT clamped here: the last vector of the last
    // row of a plane whose rows are no multiple of the vector width hangs over by a few elements and IS read by the first stage of a
    // decomposed brick (row nip-1 is the second ghost layer); the elements behind the plane are the next plane's, masked -- only in the
    // array's last plane must the access stay inside, and there the vector is never used (lim, Geom2::last_off).
    auto off_of = [&](long long f) -> unsigned {
      const long long r = f / R;
      long long el = r * g.nkp + kw0 + (f - r * R) * V;
      el = el < 0 ? 0 : el;
      return (unsigned)(el * (long long)sizeof(REAL));
    };
    auto lim = [&](unsigned off, int plane) -> unsigned { return plane == g.jlast ? (off < g.last_off ? off : g.last_off) : off; };
    unsigned bo[MV];   // byte offset of the thread's m-th vector inside a plane (clamped into the plane: such lanes are masked)
    unsigned ka[MV];   // stage-1 bits: components of the vector inside the stage-1 box (0 when the row is outside)
    unsigned own[MV];  // stage-2 bits if this workgroup owns the vector (stores, residual counts), else 0
    int pbase[MV];     // RB: (kk + ii + par) of component 0; component cc on plane jj has colour (pbase + cc + jj) & 1
#pragma unroll
    for (int m = 0; m < MV; m++) {
      const int e = t + m * TB;
      const long long f = e1_0 + e;
      const long long fc = f < vlast ? f : vlast;
      bo[m] = off_of(fc);
      const long long row = f / R;
      const int kv = (int)(f - row * R);
      const int kb = kw0 + kv * V;  // k of component 0
      unsigned bits1 = 0, bits2 = 0;
#pragma unroll
      for (int cc = 0; cc < V; cc++) {
        const int kk = kb + cc;
        if (kk >= g.kk0a && kk <= g.kk1a) bits1 |= 1u << cc;
        if (kk >= g.kk0 && kk <= g.kk1) bits2 |= 1u << cc;
      }
      pbase[m] = kb + (int)row + g.par;
      ka[m] = (f >= g.F0a && f < g.Fenda) ? bits1 : 0u;
      const bool kown = kv >= g.hv && kv < g.hv + g.KT;  // the window that holds the vector owns it (its halo vectors belong to the windows next door)
      own[m] = (e >= R && e < LV - R && f >= g.F0 && f < g.Fend && kown) ? bits2 : 0u;
    }
    REAL XG[MAF ? MV : 1], XGG[MAF ? MV : 1];  // MAF: metric terms of the rows of the thread's vectors
    int kvo[MAF ? MV : 1];                      // MAF: first k of the vector (offset into ztab)
    if (MAF) {
      const int nkp = R * V;  // (the row view of the window: element kk of it is k = kw0 + kk of the row)
#pragma unroll
      for (int m = 0; m < MV; m++) {
        const long long f = e1_0 + t + m * TB;
        const long long row = f / R;
        kvo[m] = (int)(f - row * R) * V;
        int ii = (int)row;  // padded row index == index into xc for g = 2
        if (ii < 1) ii = 1;
        const int nip = (int)(g.PSV / R);
        if (ii > nip - 2) ii = nip - 2;
        const REAL xm = ma.xc[ii - 1], x0 = ma.xc[ii], xp = ma.xc[ii + 1];
        XG[m] = (REAL)0.5 * (xp - xm);
        XGG[m] = xp - (REAL)2.0 * x0 + xm;
      }
      for (int kk = t; kk < nkp; kk += TB) {
        int kc = kw0 + kk < 1 ? 1 : kw0 + kk;
        if (kc > g.nkp - 2) kc = g.nkp - 2;
        const REAL zm = ma.zc[kc - 1], z0 = ma.zc[kc], zp = ma.zc[kc + 1];
        ztab[kk] = (REAL)0.5 * (zp - zm);
        ztab[nkp + kk] = zp - (REAL)2.0 * z0 + zm;
      }
    }
    // the two outer rows of E2: the first R threads stage the lower one, the last R threads the upper one (2R <= TB)
    const bool has_halo = (t < R) || (t >= TB - R);
    const int hl = (t < R) ? t : (LV + R + (t - (TB - R)));  // index inside an LDS u buffer
    unsigned hbo;
    {
      long long fh = e2_0 + hl;
      if (!has_halo) fh = e1_0 + t;
      if (fh > vlast) fh = vlast;
      hbo = off_of(fh);
    }
    const char* Ub = reinterpret_cast<const char*>(U);
    const char* Bb = reinterpret_cast<const char*>(B);
    char* Wb = reinterpret_cast<char*>(W);
    const REAL bs_a = (BS && bs.pa) ? *bs.pa : bs.a;
    const char* SXb = reinterpret_cast<const char*>(bs.x);
    const char* SYb = reinterpret_cast<const char*>(bs.y);
    const char* SZb = reinterpret_cast<const char*>(bs.z);
    char* SOb = reinterpret_cast<char*>(bs.out);

    // Register sets.  Two of each, used alternately by even and odd steps, so that nothing in flight is ever copied (a v_mov of a
    // register with a pending load would wait for it): u(q+1) / u(q+2) and b(q) / b(q+1).
    Vec<V> uA[MV], uB[MV], bA[MV], bB[MV], b2[MV], vc[MV], hx;
    Vec<V> rx[BS ? MV : 1], ry[BS ? MV : 1], rz[BS == 2 ? MV : 1];  // BS: the operands of b(q), asked for one step ahead (ONE set: b(q) is made
                                                                     // from them at the top of step q, before the requests of that step)
    // PRE: every operand of the chunk.  UP[p] = u(ja-2+p), HP[p] its outer rows (the has_halo threads), BP[p] = b(ja-1+p); planes beyond
    // jb+2 / jb+1 (a last chunk shorter than PRE) are clamped and never used.
    Vec<V> UP[PRE ? PRE + 4 : 1][MV], HP[PRE ? PRE + 4 : 1], BP[PRE ? PRE + 2 : 1][MV];
    if (PRE) {
#pragma unroll
      for (int p = 0; p < PRE + 4; p++) {
        const int pl = (ja - 2 + p <= jb + 2) ? ja - 2 + p : jb + 2;
        const char* Pp = Ub + (size_t)pl * PB;
#pragma unroll
        for (int m = 0; m < MV; m++) UP[p][m] = ld16<V>(Pp, lim(bo[m], pl));
        HP[p] = ld16<V>(Pp, lim(hbo, pl));
      }
#pragma unroll
      for (int p = 0; p < PRE + 2; p++) {
        const int pl = (ja - 1 + p <= jb + 1) ? ja - 1 + p : jb + 1;
        const char* Qp = Bb + (size_t)pl * PB;
#pragma unroll
        for (int m = 0; m < MV; m++) BP[p][m] = ld16<V>(Qp, bo[m]);
      }
#pragma unroll
      for (int m = 0; m < MV; m++) {
        ldsU[LU + R + t + m * TB] = UP[0][m];
        ldsU[R + t + m * TB] = UP[1][m];
        b2[m] = zerov<V>();
      }
      if (has_halo) ldsU[hl] = HP[1];
    } else
    // prologue: LDS_U[1] <- u(ja-2) (own vectors), LDS_U[0] <- u(ja-1) on E2; in flight: u(ja) and b(ja-1)
    {
      const char* P2 = Ub + (size_t)(ja - 2) * PB;
      const char* P1 = Ub + (size_t)(ja - 1) * PB;
      const char* P0 = Ub + (size_t)ja * PB;
      const char* Q1 = Bb + (size_t)(ja - 1) * PB;
      Vec<V> t2[MV], t1[MV], h1;
#pragma unroll
      for (int m = 0; m < MV; m++) {
        t2[m] = ZU ? zerov<V>() : ld16<V>(P2, lim(bo[m], ja - 2));
        t1[m] = ZU ? zerov<V>() : ld16<V>(P1, lim(bo[m], ja - 1));
      }
      h1 = ZU ? zerov<V>() : ld16<V>(P1, lim(hbo, ja - 1));
#pragma unroll
      for (int m = 0; m < MV; m++) {
        uA[m] = ZU ? zerov<V>() : ld16<V>(P0, lim(bo[m], ja));
        if (BS) {
          rx[m] = ld16<V>(SXb + (size_t)(ja - 1) * PB, bo[m]);
          ry[m] = ld16<V>(SYb + (size_t)(ja - 1) * PB, bo[m]);
          if (BS == 2) rz[m] = ld16<V>(SZb + (size_t)(ja - 1) * PB, bo[m]);
          bA[m] = zerov<V>();
        } else {
          bA[m] = ld16<V>(Q1, bo[m]);
        }
        b2[m] = zerov<V>();
      }
#pragma unroll
      for (int m = 0; m < MV; m++) {
        ldsU[LU + R + t + m * TB] = t2[m];
        ldsU[R + t + m * TB] = t1[m];
      }
      if (has_halo) ldsU[hl] = h1;
    }
    __syncthreads();

    // One plane step.  cur: LDS buffer holding u(q) / v(q-1).  uc = u(q+1) and b1 = b(q) were requested one step ago; un, hn, bn
    // receive this step's requests (u(q+2), b(q+1)).
    auto step = [&](const int q, const int cur, Vec<V>* uc, Vec<V>* un, Vec<V>* b1, Vec<V>* bn, const Vec<V>* hpre) __attribute__((always_inline)) {
      const bool plane_inner = (q >= g.jj0a && q <= g.jj1a);
      const bool count1 = (q >= ja && q <= jb);
      const bool do2 = (q - 1 >= ja);
      if (BS) {
        // b(q) from its operands (requested one step ago; loads return in order, nothing younger is waited for), kept in b1
#pragma unroll
        for (int m = 0; m < MV; m++) {
#pragma unroll
          for (int cc = 0; cc < V; cc++) {
            if (BS == 1) b1[m].v[cc] = bs_a * rx[m].v[cc] + ry[m].v[cc];
            if (BS == 2) b1[m].v[cc] = rx[m].v[cc] + bs_a * (rz[m].v[cc] - bs.b * ry[m].v[cc]);
          }
        }
        if (count1) {  // the planes of this chunk, the vectors of this segment: every point of the inner box has exactly one owner
          char* Sq = SOb + (size_t)q * PB;
#pragma unroll
          for (int m = 0; m < MV; m++) {
            if (own[m] == (1u << V) - 1) {
              st16<V>(Sq, bo[m], b1[m]);
            } else if (own[m] != 0) {
              REAL* sp = reinterpret_cast<REAL*>(Sq + bo[m]);
#pragma unroll
              for (int cc = 0; cc < V; cc++)
                if (own[m] & (1u << cc)) sp[cc] = b1[m].v[cc];
            }
          }
        }
      }
      // ---- requests for the NEXT step (the last step re-reads a plane it already has: no branch around a load)
      if (PRE) {
        hx = *hpre;  // (everything was requested before the first step)
      } else {
        const int qu = (q + 2 <= jb + 2) ? q + 2 : jb + 2;
        const int qb = (q + 1 <= jb + 1) ? q + 1 : jb + 1;
        const char* Un = Ub + (size_t)qu * PB;
        const char* Bn = Bb + (size_t)qb * PB;
        // the outer rows of u(q+1) go to LDS at the end of THIS step (one register set; the oldest request of the step)
        hx = ZU ? zerov<V>() : ld16<V>(Ub + (size_t)(q + 1) * PB, lim(hbo, q + 1));
#pragma unroll
        for (int m = 0; m < MV; m++) un[m] = ZU ? zerov<V>() : ld16<V>(Un, lim(bo[m], qu));
        if (BS) {
          const size_t po = (size_t)qb * PB;
#pragma unroll
          for (int m = 0; m < MV; m++) {
            rx[m] = ld16<V>(SXb + po, bo[m]);
            ry[m] = ld16<V>(SYb + po, bo[m]);
            if (BS == 2) rz[m] = ld16<V>(SZb + po, bo[m]);
          }
        } else {
#pragma unroll
          for (int m = 0; m < MV; m++) bn[m] = ld16<V>(Bn, bo[m]);
        }
      }
      const Vec<V>* cU = ldsU + (size_t)cur * LU;        // u(q) on E2
      Vec<V>* pU = ldsU + (size_t)(cur ^ 1) * LU;        // u(q-1), own vectors; receives u(q+1)
      const Vec<V>* cV = ldsV + (size_t)cur * LV;        // v(q-1) on E1
      Vec<V>* pV = ldsV + (size_t)(cur ^ 1) * LV;        // v(q-2), own vectors; receives v(q)
      const int w0 = (t & ~63);                          // first lane of the wave
      REAL YE1 = 0, YEE1 = 0, YE2 = 0, YEE2 = 0;         // MAF: metric terms of plane q (stage 1) and q-1 (stage 2)
      if (MAF) {
        const REAL ya = ma.yc[q - 2 > 0 ? q - 2 : 0], yb = ma.yc[q - 1], y0 = ma.yc[q], yp = ma.yc[q + 1];
        YE1 = (REAL)0.5 * (yp - yb), YEE1 = yp - (REAL)2.0 * y0 + yb;
        YE2 = (REAL)0.5 * (y0 - ya), YEE2 = y0 - (REAL)2.0 * yb + ya;
      }

      // ---- stage 1: v(q) on E1
      if (plane_inner) {
#pragma unroll
        for (int m = 0; m < MV; m++) {
          const int x = R + t + m * TB;
          const int xw = R + w0 + m * TB;
          const Vec<V> pc = lds_ld<V>(cU + x);
          const Vec<V> im = lds_ld<V>(cU + x - R);
          const Vec<V> ip = lds_ld<V>(cU + x + R);
          const Vec<V> pm = lds_ld<V>(pU + x);
          // the k neighbours beyond the ends of the wave's run of vectors: one broadcast read per wave, the rest from the next lane
          const REAL elo = reinterpret_cast<const REAL*>(cU)[(size_t)xw * V - 1];
          const REAL ehi = reinterpret_cast<const REAL*>(cU)[(size_t)(xw + 64) * V];
          const REAL kl = lane_shr1(elo, pc.v[V - 1]);
          const REAL kr = lane_shl1(ehi, pc.v[0]);
          const unsigned msk = ka[m];
          const unsigned cnt = count1 ? own[m] : 0u;
          Vec<V> ZT = zerov<V>(), ZTT = zerov<V>();
          if (MAF) {
            ZT = lds_ld<V>(reinterpret_cast<const Vec<V>*>(ztab + kvo[m]));
            ZTT = lds_ld<V>(reinterpret_cast<const Vec<V>*>(ztab + R * V + kvo[m]));
          }
          if (RB) {  // colour 0 on plane q: components with (pbase + cc + q) even
            const bool sc = ((pbase[m] + q) & 1) != 0;
            if (MAF) {
              const REAL xg = XG[m], xgg = XGG[m], omg = c.omg;
              vc[m] = relax_vec_rb<V>(pc, im, ip, pm, uc[m], kl, kr, b1[m], ZT, ZTT, sc, msk, cnt & msk, acc1,
                                      [=](REAL pp, REAL ipv, REAL imv, REAL pnv, REAL pmv, REAL kp1, REAL km1, REAL bv, REAL z, REAL zz) {
                                        const MafW w = maf_weights(xg, xgg, YE1, YEE1, z, zz);
                                        const REAL rp = w.w1 * ipv + w.w2 * imv + w.w3 * pnv + w.w4 * pmv + w.w5 * kp1 + w.w6 * km1 + bv;
                                        return (rp / w.dd - pp) * omg;
                                      });
            } else {
              vc[m] = relax_vec_rb<V>(pc, im, ip, pm, uc[m], kl, kr, b1[m], ZT, ZTT, sc, msk, cnt & msk, acc1,
                                      [&](REAL pp, REAL ipv, REAL imv, REAL pnv, REAL pmv, REAL kp1, REAL km1, REAL bv, REAL, REAL) {
                                        const REAL ss = offdiag_sum<UNIT>(c, ipv, imv, pnv, pmv, kp1, km1);
                                        return (dv(ss - bv) - pp) * c.omg;
                                      });
            }
          } else if (MAF) {
            vc[m] = relax_vec_maf<V>(pc, im, ip, pm, uc[m], kl, kr, b1[m], XG[m], XGG[m], YE1, YEE1, ZT, ZTT, c.omg, msk, cnt & msk, acc1);
          } else {
            vc[m] = relax_vec<V, UNIT>(pc, im, ip, pm, uc[m], kl, kr, b1[m], c, dv, msk, cnt & msk, acc1);
          }
        }
      } else {
#pragma unroll
        for (int m = 0; m < MV; m++) vc[m] = lds_ld<V>(cU + R + t + m * TB);  // outside the stage-1 box: the first sweep leaves the plane alone
      }

      // ---- stage 2: w(q-1) on the own segment
      if (do2) {
        char* Wq = Wb + (size_t)(q - 1) * PB;
#pragma unroll
        for (int m = 0; m < MV; m++) {
          const int e = t + m * TB;
          const int ew = w0 + m * TB;
          // (all lanes: the lane shifts need every lane's own value)
          const REAL elo = reinterpret_cast<const REAL*>(cV)[ew > 0 ? (size_t)ew * V - 1 : 0];
          const REAL ehi = reinterpret_cast<const REAL*>(cV)[ew + 64 < LV ? (size_t)(ew + 64) * V : 0];
          const Vec<V> vb = lds_ld<V>(cV + e);  // v(q-1) of the own vector
          const REAL kl = lane_shr1(elo, vb.v[V - 1]);
          const REAL kr = lane_shl1(ehi, vb.v[0]);
          if (own[m] != 0) {
            const Vec<V> im = lds_ld<V>(cV + e - R);
            const Vec<V> ip = lds_ld<V>(cV + e + R);
            const Vec<V> pm = lds_ld<V>(pV + e);
            const unsigned m2 = own[m];
            Vec<V> ZT = zerov<V>(), ZTT = zerov<V>();
            if (MAF) {
              ZT = lds_ld<V>(reinterpret_cast<const Vec<V>*>(ztab + kvo[m]));
              ZTT = lds_ld<V>(reinterpret_cast<const Vec<V>*>(ztab + R * V + kvo[m]));
            }
            Vec<V> o;
            if (RB) {  // colour 1 on plane q-1: components with (pbase + cc + q - 1) odd
              const bool sc = ((pbase[m] + q) & 1) != 0;
              if (MAF) {
                const REAL xg = XG[m], xgg = XGG[m], omg = c.omg;
                o = relax_vec_rb<V>(vb, im, ip, pm, vc[m], kl, kr, b2[m], ZT, ZTT, sc, m2, m2, acc2,
                                    [=](REAL pp, REAL ipv, REAL imv, REAL pnv, REAL pmv, REAL kp1, REAL km1, REAL bv, REAL z, REAL zz) {
                                      const MafW w = maf_weights(xg, xgg, YE2, YEE2, z, zz);
                                      const REAL rp = w.w1 * ipv + w.w2 * imv + w.w3 * pnv + w.w4 * pmv + w.w5 * kp1 + w.w6 * km1 + bv;
                                      return (rp / w.dd - pp) * omg;
                                    });
              } else {
                o = relax_vec_rb<V>(vb, im, ip, pm, vc[m], kl, kr, b2[m], ZT, ZTT, sc, m2, m2, acc2,
                                    [&](REAL pp, REAL ipv, REAL imv, REAL pnv, REAL pmv, REAL kp1, REAL km1, REAL bv, REAL, REAL) {
                                      const REAL ss = offdiag_sum<UNIT>(c, ipv, imv, pnv, pmv, kp1, km1);
                                      return (dv(ss - bv) - pp) * c.omg;
                                    });
              }
            } else if (MAF) {
              o = relax_vec_maf<V>(vb, im, ip, pm, vc[m], kl, kr, b2[m], XG[m], XGG[m], YE2, YEE2, ZT, ZTT, c.omg, m2, m2, acc2);
            } else {
              o = relax_vec<V, UNIT>(vb, im, ip, pm, vc[m], kl, kr, b2[m], c, dv, m2, m2, acc2);
            }
            if (own[m] == (1u << V) - 1) {
              st16<V>(Wq, bo[m], o);
            } else {
              REAL* wp = reinterpret_cast<REAL*>(Wq + bo[m]);
#pragma unroll
              for (int cc = 0; cc < V; cc++)
                if (own[m] & (1u << cc)) wp[cc] = o.v[cc];
            }
          }
        }
      }

      // ---- publish v(q) and the next u centre plane (u(q+1), requested one step ago)
#pragma unroll
      for (int m = 0; m < MV; m++) pV[t + m * TB] = vc[m];
#pragma unroll
      for (int m = 0; m < MV; m++) pU[R + t + m * TB] = uc[m];
      if (has_halo) pU[hl] = hx;
#pragma unroll
      for (int m = 0; m < MV; m++) b2[m] = b1[m];  // b(q) for the next step's stage 2 (complete: stage 1 used it)
      __syncthreads();
    };

    if (PRE) {
      // step p handles plane q = ja-1+p: uc = u(q+1) = UP[p+2] with its outer rows HP[p+2], b1 = b(q) = BP[p]
#pragma unroll
      for (int p = 0; p < PRE + 2; p++) {
        if (ja - 1 + p > jb + 1) break;
        step(ja - 1 + p, p & 1, UP[p + 2], nullptr, BP[p], nullptr, &HP[p + 2]);
      }
    } else {
      for (int q = ja - 1;; q += 2) {
        step(q, 0, uA, uB, bA, bB, nullptr);
        if (q + 1 > jb + 1) break;
        step(q + 1, 1, uB, uA, bB, bA, nullptr);
        if (q + 2 > jb + 1) break;
      }
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
    *last_flag = arrive_and_test_last(fin.counter, nblk);
  }
  __syncthreads();
  if (*last_flag) pair_finalize<TB>(partials, nblk, fin, wsum);
}
