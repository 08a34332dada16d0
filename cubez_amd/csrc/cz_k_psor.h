// cz_k_psor.h -- part of cz_kernels.hip (ONE translation unit per precision; included inside its anonymous namespace after cz_k_linesor.h):
// psor_col_k, the lexicographic point SOR sweep (psor_ / psor_maf_, cz_solver.f90:207-269, cz_maf.f90:23-112) in ONE launch.
// ------------------------------------------------------------------------------------------------------------
// In the order (j outer, i, k inner) an update sees the NEW values of its k-1, i-1, j-1 neighbours and the OLD ones of k+1, i+1, j+1: all
// points of a hyperplane k + i + j = const are independent, the sweep is a wavefront of 3N hyperplanes.  psor_tile_k runs it as 3N/16 - 2
// launches of 16^3 tiles, 46 barrier-separated steps each: 94 launches x 22 us at 512^3, bound by the chain of launches, not by memory.
// Here a workgroup of 16 x 16 threads owns a COLUMN of the (i, j) plane and walks it along k: at step s thread (i, j) updates k = s - i - j
//     k-1         its own previous result (a register)
//     i-1, j-1    what the thread next door computed in the step before: through LDS
//     k+1         the old value of its own line, read ahead
//     i+1, j+1    old values the thread next door holds as its own coming centre value (it updates that point one step LATER): through LDS
// The LDS planes carry a halo (i = -1 and 16, j = -1 and 16) that the 64 lanes of wave 0 serve as virtual threads of the plane:
//     i = 16 / j = 16   old values of the column on the high side, which runs later, or of the boundary: from memory
//     i = -1 / j = -1   on the box's low faces the boundary value (memory, never written); else the new values of column (a-1, b) / (a, b-1),
//                       which that column hands over through memory, see below.
// One LDS-only barrier per step, planes double-buffered.  Every thread streams its own k-line (old values in, new values out, b in) with
// dword-aligned 16-byte accesses once every four steps, two groups ahead of their use; the threads beyond a partial column's edge stream
// the boundary line next to it, so the same LDS reads serve full and partial columns.
// Between columns: the high faces of a column go to memory as 64-bit words {sweep number | value bits} (FP64: two words), one row of 16 words
// per step, written write-through and read -- one step ahead -- with agent-scope loads until the word carries this sweep's number: value
// and flag in one single-copy-atomic store, the protocol of pcr_lex_wg_k (RCCL's LL protocol).  The face buffers hold the whole sweep (no
// ring, hence no back-pressure: a column never waits for a LATER one).  Columns are handed out by a ticket in the order of their diagonals
// a + b, so the columns a workgroup waits for were taken before its own, by workgroups that are running or done: progress with any number
// of resident workgroups.  Every wait is bounded (`spin_limit`); on expiry all workgroups leave and the residual is NaN.
// Why an old value read from memory is still old: the column on the high side cannot update (k, I0+16, j) before it has received
// new(k, I0+15, j), which this column computes FROM that old value -- the load has returned before the word is stored.
// Same per-point operations in the same order as psor_tile_k and the reference => same bits.
// ------------------------------------------------------------------------------------------------------------
constexpr int PC_T = 16;               // a column is PC_T x PC_T points of the (i, j) plane
constexpr int PC_L = PC_T + 2;         // LDS row: i = -1 .. PC_T
constexpr int kPsorColHW = sizeof(REAL) == 8 ? 2 : 1;  // hand-off words per value

struct PsorColGeom {
  int nkp, nip, njp;
  int kk0, nk, ii0, ii1, jj0, jj1;  // inner box: first padded k and count, padded i / j ranges (inclusive)
  int nti, ntj;                     // columns per axis
  long long face_words;             // 64-bit words of one face buffer = (nk + PC_T) rows x PC_T values x kPsorColHW
};

// a run of N consecutive elements from an address that is only element-aligned (dword-aligned 16-byte loads, as load_run)
template <int N>
__device__ __forceinline__ void pc_load(const REAL* __restrict__ p, REAL* o) {
  static_assert(N % kRunW == 0, "runs of whole vectors");
#pragma unroll
  for (int c = 0; c < N; c += kRunW) {
    const RunVec v = *reinterpret_cast<const RunVec*>(p + c);
#pragma unroll
    for (int w = 0; w < kRunW; w++) o[c + w] = v[w];
  }
}

template <int MAF>
__global__ void __launch_bounds__(PC_T * PC_T)
psor_col_k(REAL* __restrict__ P, const REAL* __restrict__ B, Coef c, PsorColGeom g, const int* __restrict__ order, unsigned* ctl,
           unsigned long long* faces, unsigned seq, long long spin_limit, double* partials, double* dst, int accumulate, unsigned* counter,
           const int* __restrict__ skip, MafArgs ma) {
  if (skip != nullptr && *skip != 0) return;
  constexpr int NT = PC_T * PC_T, G = 4;  // threads; steps per group (one 16-byte access per stream and group)
  constexpr int HW = kPsorColHW;
  __shared__ REAL sNEW[2][PC_L * PC_L], sOLD[2][PC_L * PC_L];
  __shared__ double wsum[NT / 64 + 2];
  __shared__ int sh[4];
  const int t = threadIdx.x, i = t & (PC_T - 1), j = t >> 4;
  const int ncols = g.nti * g.ntj;
  const size_t si = (size_t)g.nkp, sj = (size_t)g.nkp * g.nip;
  const int li = (i + 1) + PC_L * (j + 1);  // this thread's place in an LDS plane
  unsigned polls = 0;
  long long t0 = 0;
  if (t == 0) sh[2] = 0;

  for (;;) {
    __syncthreads();
    if (t == 0) {
      sh[0] = (int)__hip_atomic_fetch_add(&ctl[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      sh[1] = (int)__hip_atomic_load(&ctl[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    const int ticket = sh[0];
    if (ticket >= ncols || sh[1] != 0 || sh[2] != 0) break;
    const int col = order[ticket];
    const int a = col % g.nti, b = col / g.nti;
    const int I0 = g.ii0 + a * PC_T, J0 = g.jj0 + b * PC_T;
    const int gi = I0 + i, gj = J0 + j;
    const bool col_in = gi <= g.ii1 && gj <= g.jj1;
    // (threads beyond the box stream the boundary line next to it: their OLD values are what the last inner thread reads as i+1 / j+1)
    const int gic = min(gi, g.ii1 + 1), gjc = min(gj, g.jj1 + 1);
    REAL* line = P + (size_t)g.kk0 + (size_t)gic * si + (size_t)gjc * sj;  // element k = 0 of this thread's line
    const REAL* bline = B + (size_t)g.kk0 + (size_t)gic * si + (size_t)gjc * sj;
    // ---- the halo of the LDS planes, served by wave 0: four groups of 16 lanes (virtual threads (hi, hj) with the same rule k = s - hi - hj)
    //   lanes  0..15  OLD at (16, lane)        lanes 16..31  OLD at (lane - 16, 16)         old values on the high sides: memory
    //   lanes 32..47  NEW at (-1, lane - 32)   lanes 48..63  NEW at (lane - 48, -1)         new values on the low sides: face words or boundary
    const int hgrp = t >> 4;
    const bool is_halo = t < 64;
    int hi = 0, hj = 0;
    if (hgrp == 0) hi = PC_T, hj = t & 15;
    else if (hgrp == 1) hi = t & 15, hj = PC_T;
    else if (hgrp == 2) hi = -1, hj = t & 15;
    else if (hgrp == 3) hi = t & 15, hj = -1;
    const int hli = (hi + 1) + PC_L * (hj + 1);
    const bool h_new = is_halo && hgrp >= 2;
    // (a face word exists only for the rows of the box: the lanes of a partial column's missing rows have nothing to wait for)
    const bool h_ring = h_new && ((hgrp == 2) ? (a > 0 && J0 + (t & 15) <= g.jj1) : (b > 0 && I0 + (t & 15) <= g.ii1));
    const int hgi = is_halo ? min(max(I0 + hi, g.ii0 - 1), g.ii1 + 1) : g.ii0, hgj = is_halo ? min(max(J0 + hj, g.jj0 - 1), g.jj1 + 1) : g.jj0;
    const REAL* hline = P + (size_t)g.kk0 + (size_t)hgi * si + (size_t)hgj * sj;  // (threads of the other waves: one common line, never used)
    // face words: rows are indexed by r = k + (the coordinate inside the face); one row = PC_T values
    unsigned long long* faceI = faces + (size_t)(2 * col) * g.face_words;       // this column's high-i face, for column (a+1, b)
    unsigned long long* faceJ = faces + (size_t)(2 * col + 1) * g.face_words;   // high-j face, for column (a, b+1)
    const unsigned long long* rin = faces;                                      // the face a halo lane reads (others: a valid dummy)
    if (h_ring) rin = faces + (size_t)(hgrp == 2 ? 2 * (col - 1) : 2 * (col - g.nti) + 1) * g.face_words;
    const int hc = h_ring ? ((hgrp == 2) ? hj : hi) : 0;  // the halo lane's coordinate inside the face
    const bool feedI = a + 1 < g.nti && i == PC_T - 1, feedJ = b + 1 < g.ntj && j == PC_T - 1;
    const int nrows = g.nk + PC_T;

    REAL XG = 0, XGG = 0, YE = 0, YEE = 0;
    if (MAF) {  // padded index == index into xc / yc / zc for g = 2 (see MafArgs)
      const int xi = min(gi, g.ii1), yj = min(gj, g.jj1);
      const REAL xm = ma.xc[xi - 1], x0 = ma.xc[xi], xp = ma.xc[xi + 1];
      const REAL ym = ma.yc[yj - 1], y0 = ma.yc[yj], yp = ma.yc[yj + 1];
      XG = (REAL)0.5 * (xp - xm), XGG = xp - (REAL)2.0 * x0 + xm;
      YE = (REAL)0.5 * (yp - ym), YEE = yp - (REAL)2.0 * y0 + ym;
    }

    // ---- line streams.  Group `grp` covers the steps s = G grp .. G grp + G - 1, i.e. this thread's k = kb .. kb + G - 1 with
    // kb = G grp - i - j.  pb[m] = p_old(kb + m), m = 0 .. 2G - 1 (the second half arrives while the first is used); bb[m] = b(kb + m);
    // hb[m]: the halo lane's line from hkb, where hb[u + 2] is what it publishes at sub-step u.  Every load is unconditional: elements before
    // k = -1 or behind k = nk belong to the neighbouring rows of the padded array (the launcher makes sure there is one on either side) and
    // are never used; k = -1 and k = nk ARE the boundary values the first / last point needs.
    const int nsteps = g.nk + 2 * (PC_T - 1);
    const int ngroups = (nsteps + G - 1) / G;
    REAL pb[2 * G], bb[2 * G], hb[2 * G], ob[G];
    const int hoff = h_new ? 2 : 0;
    {
      const int kb = -i - j;
      pc_load<G>(line + kb, &pb[0]);
      pc_load<G>(line + kb + G, &pb[G]);
      pc_load<G>(bline + kb, &bb[0]);
      pc_load<G>(bline + kb + G, &bb[G]);
      const int hks = -hi - hj - hoff;
      pc_load<G>(hline + hks, &hb[0]);
      pc_load<G>(hline + hks + G, &hb[G]);
    }
    REAL prev_new = line[-1];  // new value of k - 1 of the first point: the low boundary
    unsigned long long rw[HW];
    double acc = 0.0;

    // What the planes hold for step s is published in step s - 1: the new value of this thread's k (read as i-1 / j-1 next door), its old
    // value two points ahead (read as i+1 / j+1), and the halo.  `nv`: the thread's new value of this step; `uo`: pb / hb index of sub-step 0.
    // The face word of the halo lane was asked for one step ago (`rw`); the one for the next step is asked for here.
    auto publish = [&](int nxt, int s, REAL nv, REAL old2, REAL hv) __attribute__((always_inline)) {
      sNEW[nxt][li] = nv;
      sOLD[nxt][li] = old2;
      const int hk = s - hi - hj;  // the virtual thread's k at step s
      if (is_halo) {
        if (!h_new) {
          sOLD[nxt][hli] = hv;  // old value two points ahead of hk
        } else if (!h_ring) {
          sNEW[nxt][hli] = hv;  // boundary value at hk
        } else if (hk >= 0 && hk < g.nk) {
          const unsigned long long* q = rin + ((size_t)(hk + hc) * PC_T + hc) * HW;
          bool ok = true;
#pragma unroll
          for (int w = 0; w < HW; w++) ok = ok && (unsigned)(rw[w] >> 32) == seq;
          while (!ok) {
#pragma unroll
            for (int w = 0; w < HW; w++) rw[w] = __hip_atomic_load(q + w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            ok = true;
#pragma unroll
            for (int w = 0; w < HW; w++) ok = ok && (unsigned)(rw[w] >> 32) == seq;
            if (!ok && pipe_give_up(polls, t0, spin_limit, ctl)) {
              sh[2] = 1;
              break;
            }
          }
          t0 = 0;
          REAL v;
          if (sizeof(REAL) == 8) v = (REAL)__longlong_as_double((long long)((rw[0] & 0xffffffffull) | (rw[HW - 1] << 32)));
          else v = (REAL)__uint_as_float((unsigned)(rw[0] & 0xffffffffull));
          sNEW[nxt][hli] = v;
        }
      }
      // the word of the next step (all threads: no branch around a load; rows clamped into the face, threads without a face read word 0)
      {
        const int r = min(max(hk + 1 + hc, 0), nrows - 1);
        const unsigned long long* q = rin + (h_ring ? ((size_t)r * PC_T + hc) * HW : 0);
#pragma unroll
        for (int w = 0; w < HW; w++) rw[w] = __hip_atomic_load(q + w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    };
    // step -1: nothing is computed, the planes of step 0 are published
    {
      const int r = min(max(-1 - hi - hj + hc, 0), nrows - 1);
      const unsigned long long* q = rin + (h_ring ? ((size_t)r * PC_T + hc) * HW : 0);
#pragma unroll
      for (int w = 0; w < HW; w++) rw[w] = __hip_atomic_load(q + w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    publish(0, -1, (REAL)0, pb[1], hb[1]);
    lds_barrier();

    for (int grp = 0; grp < ngroups; grp++) {
      const int kb = G * grp - i - j;  // this thread's k at sub-step 0
#pragma unroll
      for (int u = 0; u < G; u++) {
        const int s = G * grp + u, cur = s & 1, nxt = cur ^ 1;
        const int k = kb + u;
        const bool active = col_in && k >= 0 && k < g.nk;
        // ---- operands
        const REAL pp = pb[u], kp1 = pb[u + 1];
        const REAL im1 = sNEW[cur][li - 1], jm1 = sNEW[cur][li - PC_L];
        const REAL ip1 = sOLD[cur][li + 1], jp1 = sOLD[cur][li + PC_L];
        REAL nv = pp;
        if (active) {
          REAL dp;
          if (MAF) {
            const int gk = g.kk0 + k;
            const REAL zm = ma.zc[gk - 1], z0 = ma.zc[gk], zp = ma.zc[gk + 1];
            const MafW w = maf_weights(XG, XGG, YE, YEE, (REAL)0.5 * (zp - zm), zp - (REAL)2.0 * z0 + zm);
            const REAL rp = w.w1 * ip1 + w.w2 * im1 + w.w3 * jp1 + w.w4 * jm1 + w.w5 * kp1 + w.w6 * prev_new + bb[u];  // cz_maf.f90:93-99
            dp = (rp / w.dd - pp) * c.omg;
          } else {
            const REAL ss = c.c1 * ip1 + c.c2 * im1 + c.c3 * jp1 + c.c4 * jm1 + c.c5 * kp1 + c.c6 * prev_new;  // cz_solver.f90:250-255
            dp = ((ss - bb[u]) / c.dd - pp) * c.omg;
          }
          nv = pp + dp;
          const REAL d2 = dp * dp;
          acc += (double)d2;
          prev_new = nv;
        }
        ob[u] = nv;
        // ---- hand the new value to the columns on the high sides: row r = k + (coordinate inside the face) of the face buffer
        if (active && (feedI || feedJ)) {
          const unsigned long long tag = (unsigned long long)seq << 32;
          unsigned long long w0, w1 = 0ull;
          if (sizeof(REAL) == 8) {
            const unsigned long long bits = (unsigned long long)__double_as_longlong((double)nv);
            w0 = tag | (bits & 0xffffffffull), w1 = tag | (bits >> 32);
          } else {
            w0 = tag | (unsigned long long)__float_as_uint((float)nv);
          }
          if (feedI) {
            unsigned long long* q = faceI + ((size_t)(k + j) * PC_T + j) * HW;
            __hip_atomic_store(q, w0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (HW == 2) __hip_atomic_store(q + (HW - 1), w1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
          if (feedJ) {
            unsigned long long* q = faceJ + ((size_t)(k + i) * PC_T + i) * HW;
            __hip_atomic_store(q, w0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (HW == 2) __hip_atomic_store(q + (HW - 1), w1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
        }
        publish(nxt, s, nv, pb[u + 2], hb[u + 2]);
        lds_barrier();
      }
      if (sh[2] != 0) break;  // a wait was given up (written before a barrier every thread has passed)
      // ---- the group's new values back to the line: one 16-byte store where all four exist
      if (col_in) {
        if (kb >= 0 && kb + G <= g.nk) {
#pragma unroll
          for (int cc = 0; cc < G; cc += kRunW) {
            RunVec v;
#pragma unroll
            for (int w = 0; w < kRunW; w++) v[w] = ob[cc + w];
            *reinterpret_cast<RunVec*>(line + kb + cc) = v;
          }
        } else {
#pragma unroll
          for (int u = 0; u < G; u++)
            if (kb + u >= 0 && kb + u < g.nk) line[kb + u] = ob[u];
        }
      }
      // ---- rotate the streams and ask for the runs after the next
#pragma unroll
      for (int m = 0; m < G; m++) pb[m] = pb[G + m], bb[m] = bb[G + m], hb[m] = hb[G + m];
      {
        const int kn = kb + 2 * G;
        pc_load<G>(line + kn, &pb[G]);
        pc_load<G>(bline + kn, &bb[G]);
        pc_load<G>(hline + (G * grp - hi - hj - hoff) + 2 * G, &hb[G]);
      }
    }
    __syncthreads();
    const double sblk = block_sum<NT>(acc, wsum);
    if (t == 0) __hip_atomic_store(&partials[col], sblk, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  // ---- residual: the columns' partials in column order, by the workgroup that arrives last (hand-off as in stencil_k)
  const int nblk = gridDim.x;
  __syncthreads();
  if (t == 0) sh[3] = arrive_and_test_last(counter, nblk);
  __syncthreads();
  if (sh[3]) {
    double xs = 0.0;
    for (int q = t; q < ncols; q += NT) xs += __hip_atomic_load(&partials[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    const double tot = block_sum<NT>(xs, wsum);
    if (t == 0) {
      const bool bad = __hip_atomic_load(&ctl[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u;
      dst[0] = bad ? __builtin_nan("") : (accumulate ? dst[0] + tot : tot);
      *counter = 0u;
    }
  }
}
