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
// steps per group (one run of G elements per line stream and group) and steps per loop body of psor_col_k; the launcher's read-extent guard
// (try_psor_col) is derived from the same constants
constexpr int kPsorG = sizeof(REAL) == 4 ? 8 : 4;
constexpr int kPsorNG = 4, kPsorNS = kPsorNG * kPsorG;

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

// threads of a workgroup that walks NC columns at once: NC x 256 computing threads + the wave that feeds the faces + the wave that takes
// them (32 lanes of each per column).  Two columns (of ONE diagonal, hence independent) per workgroup double the columns in flight where a
// CU holds one workgroup: ten waves spread over the four SIMDs as 3, 3, 2, 2 and fit the 3 waves per SIMD that ~150 registers allow, two
// workgroups of six waves (2, 2, 1, 1 each) do not.
constexpr int psor_col_threads(int nc) { return nc * PC_T * PC_T + 128; }
// Round 4: TWO workgroups per CU.  The timeline of round 3 (profiles/r03/psor_col_per_column_timeline_512_f32.txt) shows what bounded the sweep
// besides the chain: one column per CU = 256 columns in flight of 1 024, four rounds of ~245 us (540 steps x 0.45 us) = 0.98 ms whatever
// the chain does.  A step is latency (one barrier, one LDS round trip, one division), so two independent columns share a CU almost for
// free -- if their registers fit: 12 waves per CU = 3 per SIMD = 168 registers per thread.  The kernel had 203: the ring of the OLD-halo
// line (32 registers in every computing thread, used by 32 of 256) moved to the idle half of the taking wave.
#ifndef PSOR_MIN_WAVES
#define PSOR_MIN_WAVES 3
#endif

template <int MAF, int NC, int AH_>
__global__ void __launch_bounds__(psor_col_threads(NC), MAF ? 2 : PSOR_MIN_WAVES)  // (MAF: 30-50 registers more; one column per CU as before)
psor_col_k(REAL* __restrict__ P, const REAL* __restrict__ B, Coef c, PsorColGeom g, const int* __restrict__ order, int ntickets, unsigned* ctl,
           unsigned long long* faces, unsigned seq, long long spin_limit, double* partials, double* dst, int accumulate, unsigned* counter,
           const int* __restrict__ skip, MafArgs ma, long long* prof) {
  // prof (development aid, tools/psor_lab): per column {start, end} in ticks of the 100 MHz wall clock and the workgroup that ran it
  if (skip != nullptr && *skip != 0) return;
  constexpr int NT = psor_col_threads(NC);
  // steps per group: one run of G elements per stream and group.  FP32: 32-byte runs (two 16-byte loads back to back) -- a CU sustains only so
  // many line requests in flight, and with 16-byte runs every one of them fetched a line for 16 bytes: 1.42 -> 1.23 ms per 512^3 sweep
  // (profiles/r03/psor_one_launch_vs_tile_hyperplanes.txt); FP64 stays at 16 elements per loop body = one line (the flush period)
  constexpr int G = kPsorG;
  constexpr int NCW = 4 * NC;                       // computing waves
  constexpr int HW = kPsorColHW;
  constexpr int NG = kPsorNG, NS = kPsorNS;  // the loop bodies cover NG groups: register rings with compile-time indices
  static_assert(NS <= 128 / (int)sizeof(REAL), "a loop body must not outrun the two-line output ring");
  __shared__ REAL sNEW[NC][2][PC_L * PC_L], sOLD[NC][2][PC_L * PC_L];
  // The new values go back to memory as WHOLE 128-byte lines: written 16 bytes at a time as they are produced, a line left the XCD's L2 before
  // its other seven pieces arrived -- 2.1 GB of partial-line writes per 512^3 sweep where 0.53 GB are due (WRITE_SIZE, profiles/r03).  Every
  // thread therefore collects its values in a ring of two lines in LDS ([entry][thread]: conflict-free) and stores a line in one burst
  // once it is complete.
  constexpr int EL = 128 / (int)sizeof(REAL);  // elements per line
  __shared__ REAL sOUT[NC][2 * EL][PC_T * PC_T];
  __shared__ double wsum[NT / 64 + 2];
  __shared__ int sh[4 + NC];
  const int t = threadIdx.x;
  // Three kinds of wave.  Memory operations of a wave complete in the order they were issued (one counter, vmcnt, for loads AND stores): a
  // wave that stores write-through face words and waits for loads would wait for the acknowledgement of its stores at every load -- the
  // first version of this kernel did, and a column that fed another took 0.55 us per step instead of 0.36.  So:
  //   waves 0..4NC-1  compute (four per column): stream their lines (loads, whole-line plain stores); the first of a column also serves its OLD halo
  //   the next wave   feeds: reads the new values of the columns' high faces from the LDS planes and stores the face words -- stores only
  //   the last wave   takes: reads the face words (or the boundary) of the low faces and publishes them in the NEW halos -- loads only
  const int wv = __builtin_amdgcn_readfirstlane(t >> 6);
  const int lane = t & 63;
  const int ncols = g.nti * g.ntj;
  const size_t si = (size_t)g.nkp, sj = (size_t)g.nkp * g.nip;
  const int nsteps = g.nk + 2 * (PC_T - 1);
  // whole loop bodies; one step more than the sweep has: the feeding wave stores what step s - 1 computed in step s
  const int ngroups = ((nsteps + 1 + G - 1) / G + NG - 1) / NG * NG;
  const int nrows = g.nk + PC_T;
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
    if (ticket >= ntickets || sh[1] != 0 || sh[2] != 0) break;
    // this thread's column: computing threads by their block of 256, the lanes of the two other waves by their half
    const int cs = (wv < NCW) ? (wv >> 2) : ((NC > 1) ? ((t >> 5) & 1) : 0);
    const int colx = order[NC * ticket + cs];  // (-1: an odd column count on the diagonal leaves this place empty)
    const bool has_col = colx >= 0;
    const int col = has_col ? colx : order[NC * ticket];  // (an empty place shadows the ticket's first column: loads only, nothing stored)
    const int a = col % g.nti, b = col / g.nti;
    const int I0 = g.ii0 + a * PC_T, J0 = g.jj0 + b * PC_T;
    // face words: rows are indexed by r = k + (the coordinate inside the face); one row = PC_T values
    unsigned long long* faceI = faces + (size_t)(2 * col) * g.face_words;       // this column's high-i face, for column (a+1, b)
    unsigned long long* faceJ = faces + (size_t)(2 * col + 1) * g.face_words;   // high-j face, for column (a, b+1)
    double acc = 0.0;
    if (prof && has_col && (t & 255) == 0 && wv < NCW) prof[4 * col] = (long long)wall_clock64(), prof[4 * col + 2] = blockIdx.x, prof[4 * col + 3] = -(long long)__builtin_readcyclecounter();

    if (wv < NCW) {
      // ================================================================ compute waves
      const int tc = t & 255;  // thread of the column
      const int i = tc & (PC_T - 1), j = tc >> 4;
      const int li = (i + 1) + PC_L * (j + 1);  // this thread's place in an LDS plane
      const int gi = I0 + i, gj = J0 + j;
      const bool col_in = has_col && gi <= g.ii1 && gj <= g.jj1;
      // (threads beyond the box stream the boundary line next to it: their OLD values are what the last inner thread reads as i+1 / j+1)
      const int gic = min(gi, g.ii1 + 1), gjc = min(gj, g.jj1 + 1);
      REAL* line = P + (size_t)g.kk0 + (size_t)gic * si + (size_t)gjc * sj;  // element k = 0 of this thread's line
      const REAL* bline = B + (size_t)g.kk0 + (size_t)gic * si + (size_t)gjc * sj;
      // this line's place in the grid of 128-byte memory lines: element k sits at position phi + k (P is 256-byte aligned)
      const int phi = (int)((((size_t)g.kk0 + (size_t)gic * si + (size_t)gjc * sj) + (reinterpret_cast<size_t>(P) / sizeof(REAL))) % EL);
      int next_line = 0;  // lines of this thread's row stored so far (line n holds the positions n EL .. n EL + EL - 1)
      // store every complete line that has not been stored yet; `kdone`: the last k this thread has produced (all = the row is finished)
      auto flush_lines = [&](int kdone, bool all) __attribute__((always_inline)) {
#if defined(PSOR_LAB_NO_STORE)  // tools/psor_lab only, TIMING only: nothing is stored (what the stores cost)
        return;
#endif
        if (!col_in) return;
        const int kd = min(kdone, g.nk - 1);
        while (kd >= 0 && (next_line + 1) * EL - 1 <= phi + kd + ((all && kd == g.nk - 1) ? EL : 0) && next_line * EL <= phi + g.nk - 1) {
          const int c0 = next_line * EL;  // first position of the line
#pragma unroll
          for (int v = 0; v < EL; v += kRunW) {
            RunVec x;
#pragma unroll
            for (int w = 0; w < kRunW; w++) x[w] = sOUT[cs][(c0 + v + w) % (2 * EL)][tc];
            const int k0 = c0 + v - phi;  // k of the vector's first element
            if (k0 >= 0 && k0 + kRunW <= g.nk) {
              *reinterpret_cast<RunVec*>(line + k0) = x;  // (16-byte aligned: c0 + v is a multiple of the vector width)
            } else {
#pragma unroll
              for (int w = 0; w < kRunW; w++)
                if (k0 + w >= 0 && k0 + w < g.nk) line[k0 + w] = x[w];
            }
          }
          next_line++;
        }
      };

      REAL XG = 0, XGG = 0, YE = 0, YEE = 0;
      if (MAF) {  // padded index == index into xc / yc / zc for g = 2 (see MafArgs)
        const int xi = min(gi, g.ii1), yj = min(gj, g.jj1);
        const REAL xm = ma.xc[xi - 1], x0 = ma.xc[xi], xp = ma.xc[xi + 1];
        const REAL ym = ma.yc[yj - 1], y0 = ma.yc[yj], yp = ma.yc[yj + 1];
        XG = (REAL)0.5 * (xp - xm), XGG = xp - (REAL)2.0 * x0 + xm;
        YE = (REAL)0.5 * (yp - ym), YEE = yp - (REAL)2.0 * y0 + ym;
      }
      // ---- line streams.  Group `grp` covers the steps s = G grp .. G grp + G - 1, i.e. this thread's k = kb .. kb + G - 1 with
      // kb = G grp - i - j.  The loop body is unrolled over NG = 4 groups, so that the streams live in register rings with compile-time
      // indices and nothing in flight is ever copied: pbuf[G q + u] = p_old(kb + u) of group slot q, bbuf likewise b.  A group uses its own slot and the first two entries of the next; when it is
      // done its slot is asked for again, for the group NG groups ahead -- three groups (12 steps) before the first use.  Every load is
      // unconditional: elements before k = -1 or behind k = nk belong to the neighbouring rows of the padded array (the launcher makes sure
      // they exist) and are never used; k = -1 and k = nk ARE the boundary values the first / last point needs.
      REAL pbuf[NS], bbuf[NS];
#if defined(PSOR_LAB_WHOLE_LINES)
      RunVec wl[2 * (EL / kRunW)] = {};
#endif
#pragma unroll
      for (int q = 0; q < NG; q++) {
        pc_load<G>(line + (G * q - i - j), &pbuf[G * q]);
#if defined(PSOR_LAB_NO_B)  // tools/psor_lab only: the right-hand side is not streamed (what a third less read traffic is worth; the results differ)
        for (int u = 0; u < G; u++) bbuf[G * q + u] = (REAL)0.25;
#else
        pc_load<G>(bline + (G * q - i - j), &bbuf[G * q]);
#endif
      }
      REAL prev_new = line[-1];  // new value of k - 1 of the first point: the low boundary
      // (consumed here, once: a loop-carried register that starts life as a load makes the compiler wait for ALL loads in flight -- the
      // streams asked for twelve steps ahead -- at every use inside the loop)
      asm volatile("" : "+v"(prev_new));
      // What the planes hold for step s + 1 is published in step s: the new value of this thread's k (read as i-1 / j-1 next door) and its old
      // value two points ahead (read as i+1 / j+1); the halos of both planes come from the taking wave.
      auto publish = [&](int nxt, REAL nv, REAL old2) __attribute__((always_inline)) {
        sNEW[cs][nxt][li] = nv;
        sOLD[cs][nxt][li] = old2;
      };
      publish(0, (REAL)0, pbuf[1]);  // step -1: nothing is computed, the planes of step 0 are published
      lds_barrier();

      for (int sg = 0; sg < ngroups; sg += NG) {
#pragma unroll
        for (int q = 0; q < NG; q++) {
          const int grp = sg + q;
          const int kb = G * grp - i - j;  // this thread's k at sub-step 0
#pragma unroll
          for (int u = 0; u < G; u++) {
            const int m = G * q + u;                 // position in the register rings (compile time)
            const int cur = m & 1, nxt = cur ^ 1;   // (step and m have the same parity: NS is even)
            const int k = kb + u;
            const bool active = col_in && k >= 0 && k < g.nk;
            // ---- operands
            const REAL pp = pbuf[m], kp1 = pbuf[(m + 1) % NS];
            const REAL im1 = sNEW[cs][cur][li - 1], jm1 = sNEW[cs][cur][li - PC_L];
            const REAL ip1 = sOLD[cs][cur][li + 1], jp1 = sOLD[cs][cur][li + PC_L];
            REAL nv = pp;
            if (active) {
              REAL dp;
              if (MAF) {
                const int gk = g.kk0 + k;
                const REAL zm = ma.zc[gk - 1], z0 = ma.zc[gk], zp = ma.zc[gk + 1];
                const MafW w = maf_weights(XG, XGG, YE, YEE, (REAL)0.5 * (zp - zm), zp - (REAL)2.0 * z0 + zm);
                const REAL rp = w.w1 * ip1 + w.w2 * im1 + w.w3 * jp1 + w.w4 * jm1 + w.w5 * kp1 + w.w6 * prev_new + bbuf[m];  // cz_maf.f90:93-99
                dp = (rp / w.dd - pp) * c.omg;
              } else {
                const REAL ss = c.c1 * ip1 + c.c2 * im1 + c.c3 * jp1 + c.c4 * jm1 + c.c5 * kp1 + c.c6 * prev_new;  // cz_solver.f90:250-255
                dp = ((ss - bbuf[m]) / c.dd - pp) * c.omg;
              }
              nv = pp + dp;
              const REAL d2 = dp * dp;
              acc += (double)d2;
              prev_new = nv;
            }
            if (active) sOUT[cs][(phi + k) % (2 * EL)][tc] = nv;
            publish(nxt, nv, pbuf[(m + 2) % NS]);
            lds_barrier();
          }
          // ---- this group's slot of the rings is free: ask for the runs NG groups ahead
#if !defined(PSOR_LAB_WHOLE_LINES)
          pc_load<G>(line + kb + NS, &pbuf[G * q]);
#if !defined(PSOR_LAB_NO_B)
          pc_load<G>(bline + kb + NS, &bbuf[G * q]);
#endif
#endif
        }
#if defined(PSOR_LAB_WHOLE_LINES)  // tools/psor_lab only, TIMING only (the values are not used, the results are wrong): what the memory system makes of
        // one ALIGNED whole 128-byte line of p and of b per thread and loop body, requested in one burst, instead of four unaligned 32-byte runs
        {
#pragma unroll
          for (int v = 0; v < 2 * (EL / kRunW); v++) asm volatile("" ::"v"(wl[v]));  // (the burst of the body before has to be there by now)
          const size_t ea = (reinterpret_cast<size_t>(line + (G * (sg + NG) - i - j) + NS)) & ~(size_t)127;
          const size_t eb = (reinterpret_cast<size_t>(bline + (G * (sg + NG) - i - j) + NS)) & ~(size_t)127;
#pragma unroll
          for (int v = 0; v < EL / kRunW; v++) {
            wl[v] = *reinterpret_cast<const RunVec*>(ea + (size_t)v * sizeof(RunVec));
            wl[EL / kRunW + v] = *reinterpret_cast<const RunVec*>(eb + (size_t)v * sizeof(RunVec));
          }
        }
#endif

        // ---- the lines completed in this loop body (NS steps <= EL: at most one per thread, two entries of the ring are never in doubt)
        flush_lines(G * (sg + NG) - 1 - i - j, false);
        if (sh[2] != 0) break;  // a wait was given up (published by the taking wave before the LAST barrier of this body, see there)
      }
      flush_lines(g.nk - 1, true);  // what is left: the last, incomplete line
    } else if (wv == NCW) {
      // ================================================================ the wave that feeds the faces (stores only)
      // At step s the plane `cur` holds what the computing threads published in step s - 1: thread (15, j) its new value of k = s - 16 - j,
      // thread (i, 15) of k = s - 16 - i -- row s - 16 of both faces.  Of each half of the wave (one per column): lanes 0..15 face I (j = lane),
      // lanes 16..31 face J (i = lane - 16).
      const int fc = lane & 15;                       // coordinate inside the face
      const bool toI = (lane & 16) == 0;
#if defined(PSOR_LAB_FORCE_FEED)  // tools/psor_lab only: every column stores both faces (what the stores cost a column that nobody follows)
      const bool feeds = has_col && (NC > 1 || lane < 32);
#elif defined(PSOR_LAB_NO_FEED)   // tools/psor_lab only: nobody stores (timing of a column's own steps; the sweep is wrong and gives up)
      const bool feeds = false;
#else
      const bool feeds = has_col && (NC > 1 || lane < 32) && (toI ? (a + 1 < g.nti && J0 + fc <= g.jj1) : (b + 1 < g.ntj && I0 + fc <= g.ii1));
#endif
      const int src = toI ? (PC_T + PC_L * (fc + 1)) : ((fc + 1) + PC_L * PC_T);  // LDS place of thread (15, fc) / (fc, 15)
      unsigned long long* face = toI ? faceI : faceJ;
      lds_barrier();  // (step -1)
      for (int s = 0; s < G * ngroups; s++) {
        const int cur = s & 1;
        const int k = s - PC_T - fc;  // what that thread computed in the step before
        if (feeds && k >= 0 && k < g.nk) {
          const REAL nv = sNEW[cs][cur][src];
          const unsigned long long tag = (unsigned long long)seq << 32;
          unsigned long long* q = face + ((size_t)(k + fc) * PC_T + fc) * HW;
          if (sizeof(REAL) == 8) {
            const unsigned long long bits = (unsigned long long)__double_as_longlong((double)nv);
            __hip_atomic_store(q, tag | (bits & 0xffffffffull), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(q + (HW - 1), tag | (bits >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          } else {
            __hip_atomic_store(q, tag | (unsigned long long)__float_as_uint((float)nv), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
        }
        lds_barrier();
        if ((s & (NS - 1)) == NS - 1 && sh[2] != 0) break;  // (where the computing waves look)
      }
    } else {
      // ================================================================ the wave that takes the low faces (loads only)
      // Lanes 0..15: NEW halo at (-1, lane), from face I of column (a-1, b); lanes 16..31: NEW halo at (lane - 16, -1), from face J of column
      // (a, b-1); on the box's low faces the boundary line (memory, never written).  Virtual threads of the plane: at step s lane (hi, hj)
      // publishes, for step s + 1, the new value of k = s - hi - hj = s + 1 - (coordinate): row s + 1 of the face.  Words and boundary
      // values are asked for eight steps ahead (rq / bq[s & 7]); a word that has not arrived is read again until it carries this sweep's number.
      const int hc = lane & 15;
      const bool fromI = (lane & 16) == 0;
      const bool takes = NC > 1 || lane < 32;  // (of each half of the wave, one per column: lanes 0..15 the i side, 16..31 the j side)
      const int hi = fromI ? -1 : hc, hj = fromI ? hc : -1;
      const int hli = (hi + 1) + PC_L * (hj + 1);
      // (a face word exists only for the rows of the box: the lanes of a partial column's missing rows have nothing to wait for)
      const bool h_ring = has_col && takes && (fromI ? (a > 0 && J0 + hc <= g.jj1) : (b > 0 && I0 + hc <= g.ii1));
      const unsigned long long* rin = faces;  // (lanes without a face: a valid dummy)
      if (h_ring) rin = faces + (size_t)(fromI ? 2 * (col - 1) : 2 * (col - g.nti) + 1) * g.face_words;
      const int hgi = takes ? min(max(I0 + hi, g.ii0 - 1), g.ii1 + 1) : g.ii0, hgj = takes ? min(max(J0 + hj, g.jj0 - 1), g.jj1 + 1) : g.jj0;
      const REAL* hline = P + (size_t)g.kk0 + (size_t)hgi * si + (size_t)hgj * sj;
      const int hk0 = -hi - hj;  // = 1 - hc: the virtual thread's k at step 0
      // Lanes 32..63 (NC = 1: otherwise idle): the OLD halo, i = 16 / j = 16 of the planes -- old values of the column on the high side or of
      // the boundary, which the last row / column of computing threads reads as i+1 / j+1.  Virtual threads (16, v) and (v - 16, 16), v = lane
      // - 32; at step s such a thread's own k is s - hi - hj and it publishes, like every thread, the old value two points ahead.
      static_assert(NC == 1, "the OLD halo rides on the idle half of the taking wave");
      const int ov = lane & 31;
      const bool olds = lane >= 32;
      const int ohi = (ov < 16) ? PC_T : (ov & 15), ohj = (ov < 16) ? (ov & 15) : PC_T;
      const int ohli = (ohi + 1) + PC_L * (ohj + 1);
      const REAL* oline = P + (size_t)g.kk0 + (size_t)min(I0 + ohi, g.ii1 + 1) * si + (size_t)min(J0 + ohj, g.jj1 + 1) * sj;
      const int ok0 = 2 - ohi - ohj;
      // AH: how many steps ahead of their use the face words (and the halo values) are asked for.  A word can only be there when the column on the
      // low side is 16 steps (the skew of a column) + 1 (its feeding wave) + AH steps ahead, plus the time a write-through store takes to
      // become visible: a request that comes back without this sweep's number is repeated at the step of its use, with the whole memory
      // latency exposed, so a column settles that far behind the one it follows -- the ask-ahead distance is part of EVERY hop of the chain
      // (62 hops at 512^3).  Round 3 asked 8 steps ahead (3.6 us at 0.45 us per step).  Measured (profiles/r04/psor_what_bounds_it.txt): 4 steps
      // ahead shorten the chain -- 256^3 FP32 0.408 -> 0.366 ms, 512^3 FP64 2.14 -> 2.01, MAF 1.88 -> 1.82 -- but at 512^3 FP32, where most
      // columns run long after the ones they depend on, 1.8 us no longer covers a load under full memory traffic: 1.22 -> 1.27 ms.  The launcher
      // picks 8 for FP32 boxes of more than 300 points along k and 4 otherwise.
      constexpr int AH = AH_;
      static_assert(AH == 2 || AH == 4 || AH == 8 || AH == 16, "ring of request slots with compile-time indices");  // (16 was measured: 1.38 ms at 512^3 FP32 -- the hops dominate)
      constexpr int UB = AH > 8 ? AH : 8;  // steps per unrolled loop body (a multiple of the ring, a divisor of NS)
      static_assert(NS % UB == 0, "the give-up flag is published at the last step of a loop body of the computing waves");
      unsigned long long rq[AH][HW];
      REAL bq[AH], oq[AH];
      // A given-up wait is a private matter of this wave until the last step of the loop body it happened in: sh[2] is written only in front
      // of that step's barrier, so every wave of the workgroup reads the same value behind it (written in mid-body, waves that had passed
      // the body's last barrier but not yet looked could disagree with those that had, and the barrier counts would part -- ADVICE r3).
      bool gave_up = false;
      auto ask = [&](int slot, int s) __attribute__((always_inline)) {  // what step s publishes: row hk(s) + hc of the face / element hk(s) of the line
        const int hk = s + hk0;
        const int r = min(max(hk + hc, 0), nrows - 1);
        const unsigned long long* q = rin + (h_ring ? ((size_t)r * PC_T + hc) * HW : 0);
#pragma unroll
        for (int w = 0; w < HW; w++) rq[slot][w] = __hip_atomic_load(q + w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        bq[slot] = hline[min(max(hk, -1), g.nk)];
        oq[slot] = oline[min(max(s + ok0, -1), g.nk)];
      };
      auto take = [&](int nxt, int slot, int s) __attribute__((always_inline)) {
        const int hk = s + hk0;
        const bool need = h_ring && hk >= 0 && hk < g.nk;
        bool ok = true;
#pragma unroll
        for (int w = 0; w < HW; w++) ok = ok && (unsigned)(rq[slot][w] >> 32) == seq;
        if (need && !ok && !gave_up) {
          const unsigned long long* q = rin + ((size_t)(hk + hc) * PC_T + hc) * HW;
          do {
#pragma unroll
            for (int w = 0; w < HW; w++) rq[slot][w] = __hip_atomic_load(q + w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            ok = true;
#pragma unroll
            for (int w = 0; w < HW; w++) ok = ok && (unsigned)(rq[slot][w] >> 32) == seq;
            if (!ok && pipe_give_up(polls, t0, spin_limit, ctl)) {
              gave_up = true;  // (the rest of this body runs on whatever the words hold: the sweep is void, its residual NaN)
              break;
            }
          } while (!ok);
          t0 = 0;
        }
        REAL rv;
        if (sizeof(REAL) == 8) rv = (REAL)__longlong_as_double((long long)((rq[slot][0] & 0xffffffffull) | (rq[slot][HW - 1] << 32)));
        else rv = (REAL)__uint_as_float((unsigned)(rq[slot][0] & 0xffffffffull));
        if (takes) sNEW[cs][nxt][hli] = need ? rv : bq[slot];  // (a face lane outside its range publishes something nobody reads)
        if (olds) sOLD[cs][nxt][ohli] = oq[slot];
        ask(slot, s + AH);
      };
#pragma unroll
      for (int m = -1; m < AH - 1; m++) ask(m & (AH - 1), m);
      take(0, AH - 1, -1);  // step -1
      lds_barrier();
      for (int s0 = 0; s0 < G * ngroups; s0 += UB) {
#pragma unroll
        for (int m = 0; m < UB; m++) {
          take((m & 1) ^ 1, m & (AH - 1), s0 + m);
          if (m == UB - 1 && (s0 & (NS - 1)) == NS - UB && __builtin_amdgcn_ballot_w64(gave_up) != 0ull && lane == 0) sh[2] = 1;
          lds_barrier();
        }
        if ((s0 & (NS - 1)) == NS - UB && sh[2] != 0) break;  // (where the computing waves look)
      }
    }
    __syncthreads();
    if (prof && has_col && (t & 255) == 0 && wv < NCW) prof[4 * col + 1] = (long long)wall_clock64(), prof[4 * col + 3] += (long long)__builtin_readcyclecounter();  // (shader-clock cycles of the column)
    // one partial sum per column (in the order of the columns whatever NC is: the same bits)
#pragma unroll
    for (int n = 0; n < NC; n++) {
      const double sblk = block_sum<NT>((wv < NCW && cs == n) ? acc : 0.0, wsum);
      if (t == 0 && order[NC * ticket + n] >= 0) __hip_atomic_store(&partials[order[NC * ticket + n]], sblk, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __syncthreads();
    }
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
      if (bad) ctl[2] = 1u;  // (sticky: the launcher clears ctl[0..1] only; the host asks with psor_failed)
      *counter = 0u;
    }
  }
}
