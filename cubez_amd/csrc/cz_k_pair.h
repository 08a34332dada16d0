// cz_k_pair.h -- part of cz_kernels.hip (ONE translation unit per precision; this file is included inside its anonymous
// namespace and is not a stand-alone header): what the two-stage pass kernels share (geometry, per-point update, finalisation) and
// pair_shell_k (the shell slabs of a decomposed brick).  The pass itself is jacobi2p_k, cz_k_pair2.h.
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
  // Rows of the (k, i) plane as the pass sees them: R vectors each, row i starting at a vector boundary.  In memory a row is nkp elements
  // long and rows follow one another without padding: where nkp is not a multiple of the vector width the last vector of a row is partial
  // (its tail belongs to the next row and is masked like every k outside the box) and a vector is only REAL-aligned in memory -- the global
  // accesses of the pass are dword-aligned dwordx4, which this hardware takes.  (Rounds 1-2 required nkp % V == 0 and sent every other
  // size to the one-sweep scalar kernel: 220 000 against 740 000 MLUPS at 511^3, profiles/r03/unaligned_k_extent.txt.)
  //
  // K WINDOWS (round 4).  The segment of a workgroup is a run of whole rows with one (stage 1) and two (u) halo rows on either side in LDS, so
  // its useful share is (TB MV - 2R) / (TB MV): rows beyond 2 044 (FP32) / 1 020 (FP64) elements did not fit at all (single sweeps at half
  // the rate until round 3), and from 700 elements up a segment was two or three rows of five to seven.  Now the k axis may be cut into
  // `nwin` windows of KT vectors: a workgroup sees its window as a plane of its own whose rows are R = KT + 2 vectors long -- the window plus
  // ONE halo vector on either side (a stage-2 point at the window's edge reads the stage-1 value next to it, which this workgroup computes
  // itself from the u values of that halo vector: V >= 2 elements reach far enough).  Nothing else changes in the kernel: +-R is still the i
  // neighbour, +-1 element the k neighbour, the lane next door holds it; only the map from (row, vector of the row) to memory gains the
  // window's origin `kw0`, and a vector is owned by the workgroup whose window holds it.  Same per-point arithmetic on the same values =>
  // the same bits (test_two_fused_sweeps_equal_two_oracle_sweeps with forced windows; k = 1 100 FP64 and k = 2 100 FP32 boxes).
  int R;
  long long PSV;               // R * nip: vectors per plane in that view
  int nkp = 0;                 // elements per row in memory
  long long PSB = 0;           // bytes per plane in memory
  int jlast = 0;               // index of the array's last plane, whose last vectors must not be read beyond the array:
  unsigned last_off = 0;       // ... offsets into that plane are clamped to this (the values clamped away are never used)
  int nwin = 1;                // k windows per row
  int hv = 0;                  // halo vectors on either side of a window (1 when nwin > 1)
  int KT = 1 << 30;            // vectors a window owns: vector kv of a virtual row is owned when hv <= kv < hv + KT
  int KW = 0;                  // elements from one window's origin to the next (= KT * V)
  int nsegw = 0;               // segments per window (nseg = nwin * nsegw; segment s of window w has the id w * nsegw + s)
  int kk0, kk1, jj0, jj1;      // stage-2 (output) box = the inner box
  long long F0, Fend;
  // stage-1 box: the inner box, grown by one layer across rank-internal faces of a decomposed run (the first sweep
  // must also be applied to the ghost layer the second sweep reads; two ghost layers are exchanged per pair)
  int kk0a, kk1a, jj0a, jj1a;
  long long F0a, Fenda;
  int nseg, TJ, S;  // S = TB*MV - 2R
  int par;          // RB: colour 0 = points with (kk + ii + jj + par) even
  int zero_u;       // the input field is identically zero (a freshly cleared preconditioner vector): u is not read
  int band;         // workgroup id -> (segment, chunk) by XCD bands (see jacobi2p_k)
  const int* map;   // or by a table: map[2 * id] = segment (nseg: no work), map[2 * id + 1] = chunk (pair_xcd_map, cz_h_launch.h)
};

// jacobi2p_k<..., BS>: the right-hand side made on the fly (see there).  x, y, z: operands; out: where the owner of a vector stores it.
struct BSrc {
  const REAL* x = nullptr;
  const REAL* y = nullptr;
  const REAL* z = nullptr;
  REAL* out = nullptr;
  REAL a = 0, b = 0;
  const REAL* pa = nullptr;  // where set: a is read from the device (bicg_scal_k)
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

// the ordinary division (the compiler's IEEE expansion at every point); see cz_k_fastdiv.h for the hoisted form
struct PlainDiv {
  REAL d;
  __device__ __forceinline__ REAL operator()(REAL n) const { return n / d; }
};
struct HoistedDiv {
  FastDiv<REAL> f;
  __device__ __forceinline__ REAL operator()(REAL n) const { return fastdiv(n, f); }
};

struct ShortDiv {  // cz_k_fastdiv.h: only for divisors that passed the exhaustive comparison
  FastDiv<REAL> f;
  __device__ __forceinline__ REAL operator()(REAL n) const { return shortdiv(n, f); }
};

struct MediumDiv {
  FastDiv<REAL> f;
  __device__ __forceinline__ REAL operator()(REAL n) const { return mediumdiv(n, f); }
};

template <int V, int UNIT = 0, class DIV>
__device__ __forceinline__ Vec<V> relax_vec(const Vec<V>& pc, const Vec<V>& im, const Vec<V>& ip, const Vec<V>& pm,
                                            const Vec<V>& pn, REAL kl, REAL kr, const Vec<V>& bb, const Coef& c, const DIV& dv,
                                            unsigned mask, unsigned count_mask, double& acc) {
  Vec<V> o;
#if defined(CZ_P2_RES_GROUP)  // tools/pair_lab A/B only: the vector's dp^2 summed in REAL, one conversion and one double add per vector
  REAL grp = (REAL)0;
#endif
#pragma unroll
  for (int cc = 0; cc < V; cc++) {
    const REAL pp = pc.v[cc];
    const REAL km1 = (cc == 0) ? kl : pc.v[cc > 0 ? cc - 1 : 0];
    const REAL kp1 = (cc == V - 1) ? kr : pc.v[cc < V - 1 ? cc + 1 : V - 1];
    const REAL ss = offdiag_sum<UNIT>(c, ip.v[cc], im.v[cc], pn.v[cc], pm.v[cc], kp1, km1);
    const REAL dp = (dv(ss - bb.v[cc]) - pp) * c.omg;
    const REAL d2 = dp * dp;
    o.v[cc] = (mask & (1u << cc)) ? pp + dp : pp;
#if defined(CZ_P2_NO_RES)  // tools/pair_lab A/B only: no residual at all (what the accumulation costs at most)
    (void)d2, (void)count_mask, (void)acc;
#elif defined(CZ_P2_RES_GROUP)
    grp += (count_mask & (1u << cc)) ? d2 : (REAL)0;
#else
    acc += (double)((count_mask & (1u << cc)) ? d2 : (REAL)0);  // (+0.0 leaves the sum as it is)
#endif
  }
#if defined(CZ_P2_RES_GROUP)
  acc += (double)grp;
#endif
  return o;
}

// The MAF flavour of the per-point update (cz_maf.f90:193-225, operation for operation as in stencil_k<..., MAF = 1>): the six weights and
// the diagonal are recomputed at every point from the metric terms of the 1-D grids -- XG, XGG of the row, YE, YEE of the plane, ZT, ZTT of
// the component.
template <int V>
__device__ __forceinline__ Vec<V> relax_vec_maf(const Vec<V>& pc, const Vec<V>& im, const Vec<V>& ip, const Vec<V>& pm, const Vec<V>& pn,
                                                REAL kl, REAL kr, const Vec<V>& bb, REAL XG, REAL XGG, REAL YE, REAL YEE, const Vec<V>& ZT,
                                                const Vec<V>& ZTT, REAL omg, unsigned mask, unsigned count_mask, double& acc) {
  Vec<V> o;
#pragma unroll
  for (int cc = 0; cc < V; cc++) {
    const REAL pp = pc.v[cc];
    const REAL km1 = (cc == 0) ? kl : pc.v[cc > 0 ? cc - 1 : 0];
    const REAL kp1 = (cc == V - 1) ? kr : pc.v[cc < V - 1 ? cc + 1 : V - 1];
    const MafW w = maf_weights(XG, XGG, YE, YEE, ZT.v[cc], ZTT.v[cc]);
    const REAL rp = w.w1 * ip.v[cc] + w.w2 * im.v[cc] + w.w3 * pn.v[cc] + w.w4 * pm.v[cc] + w.w5 * kp1 + w.w6 * km1 + bb.v[cc];  // :219-225
    const REAL dp = (rp / w.dd - pp) * omg;
    const REAL d2 = dp * dp;
    o.v[cc] = (mask & (1u << cc)) ? pp + dp : pp;
    acc += (double)((count_mask & (1u << cc)) ? d2 : (REAL)0);
  }
  return o;
}

// The workgroup whose ticket came last sums all per-workgroup partials in a fixed order (sc1 loads, see stencil_k) and does the
// bookkeeping of cz_Poisson.cpp:67-77 for the one or two iterations of the pass.  Called by every thread of that workgroup.
template <int TB>
__device__ __forceinline__ void pair_finalize(const double* partials, int nblk, const Fin2& fin, double* wsum) {
  const int t = threadIdx.x;
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
// MAF = 1: the weights of every point from the 1-D coordinate arrays (cz_maf.f90:193-225; padded index == index into xc / yc / zc for g = 2),
// through relax_vec_maf<1> like the pass itself.
template <int RB, int TK, int TI, int TJ, int MAF>
__device__ __forceinline__ void shell_tiles(const REAL* __restrict__ U, const REAL* __restrict__ B, REAL* __restrict__ W, const Coef& c,
                                            const ShellTab& s, const ShellBox& d, REAL* lu, double& acc1, double& acc2, const MafArgs& ma) {
  // metric terms of padded index x of a 1-D grid: XG-like first difference and XGG-like second difference
  auto met1 = [](const REAL* __restrict__ xc, int x) { return (REAL)0.5 * (xc[x + 1] - xc[x - 1]); };
  auto met2 = [](const REAL* __restrict__ xc, int x) { return xc[x + 1] - (REAL)2.0 * xc[x] + xc[x - 1]; };
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
        if (MAF) {
          const int gk = K0 - 1 + k, gi = I0 - 1 + i, gj = J0 - 1 + j;
          Vec<1> zt, ztt;
          zt.v[0] = met1(ma.zc, gk), ztt.v[0] = met2(ma.zc, gk);
          v = relax_vec_maf<1>(pc, im, ip, pm, pn, lu[cu - 1], lu[cu + 1], bb, met1(ma.xc, gi), met2(ma.xc, gi), met1(ma.yc, gj), met2(ma.yc, gj), zt, ztt,
                               c.omg, 1u, core ? 1u : 0u, acc1).v[0];
        } else {
          v = relax_vec<1>(pc, im, ip, pm, pn, lu[cu - 1], lu[cu + 1], bb, c, PlainDiv{c.dd}, 1u, core ? 1u : 0u, acc1).v[0];
        }
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
        if (MAF) {
          Vec<1> zt, ztt;
          zt.v[0] = met1(ma.zc, gk), ztt.v[0] = met2(ma.zc, gk);
          o = relax_vec_maf<1>(pc, im, ip, pm, pn, lv[cv - 1], lv[cv + 1], bb, met1(ma.xc, gi), met2(ma.xc, gi), met1(ma.yc, gj), met2(ma.yc, gj), zt, ztt,
                               c.omg, 1u, 1u, acc2).v[0];
        } else {
          o = relax_vec<1>(pc, im, ip, pm, pn, lv[cv - 1], lv[cv + 1], bb, c, PlainDiv{c.dd}, 1u, 1u, acc2).v[0];
        }
      }
      Wt[(k + 2) + (i + 2) * si + (j + 2) * sj] = o;
    }
    __syncthreads();
  }
}

template <int RB, int MAF = 0>
__global__ void __launch_bounds__(256)
pair_shell_k(const REAL* __restrict__ U, const REAL* __restrict__ B, REAL* __restrict__ W, Coef c, ShellTab s, double* partials,
             const int* __restrict__ skip, MafArgs ma) {
  if (skip != nullptr && *skip != 0) return;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  __shared__ double wsum[8];
  const int t = threadIdx.x;
  const ShellBox d = s.b[blockIdx.y];
  REAL* lu = reinterpret_cast<REAL*>(smem);
  double acc1 = 0.0, acc2 = 0.0;
  switch (d.kind) {  // uniform per workgroup
    case 0: shell_tiles<RB, 64, 4, 2, MAF>(U, B, W, c, s, d, lu, acc1, acc2, ma); break;
    case 1: shell_tiles<RB, 64, 2, 4, MAF>(U, B, W, c, s, d, lu, acc1, acc2, ma); break;
    case 2: shell_tiles<RB, 2, 16, 16, MAF>(U, B, W, c, s, d, lu, acc1, acc2, ma); break;
    default: shell_tiles<RB, 32, 4, 4, MAF>(U, B, W, c, s, d, lu, acc1, acc2, ma); break;
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

// The sums of a pair_shell_k launch added to the sums of the interior launch of the same pass (the driver runs the two concurrently on
// two streams and folds afterwards): res[0] += first-stage sums, res[1] += second-stage sums (single: both into res[0]).  One
// workgroup, fixed order.
__global__ void __launch_bounds__(256)
shell_fold_k(const double* __restrict__ partials, int n, double* __restrict__ res, int single, const int* __restrict__ skip) {
  if (skip != nullptr && *skip != 0) return;
  __shared__ double wsum[8];
  double x1 = 0.0, x2 = 0.0;
  for (int i = threadIdx.x; i < n; i += 256) x1 += partials[i], x2 += partials[n + i];
  const double s1 = block_sum<256>(x1, wsum);
  __syncthreads();
  const double s2 = block_sum<256>(x2, wsum);
  if (threadIdx.x == 0) {
    if (single) {
      res[0] += s1 + s2;
    } else {
      res[0] += s1;
      res[1] += s2;
    }
  }
}
