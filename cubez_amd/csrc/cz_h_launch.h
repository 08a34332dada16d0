// cz_h_launch.h -- part of cz_kernels.hip (ONE translation unit per precision; this file is included inside its anonymous
// namespace and is not a stand-alone header): host side: launch geometry of the sweep, pair, shell, element-wise and dot kernels.
template <int V, int TB, int M, int PF, int MODE, int MAF = 0>
void launch_stencil_inst(const REAL* P, const REAL* B, REAL* OUT, const Coef& c, const Box& b, int par, int tj_req,
                         const int* skip, int* nblk_out, const Fin& fin, const MafArgs& ma = MafArgs()) {
  Geom g;
  g.nip = b.nip;
  g.R = (b.nkp + V - 1) / V;  // rows from a vector boundary each; the last vector of a row partial where nkp % V != 0 (Geom)
  g.PSV = (long long)g.R * b.nip;
  g.nkp = b.nkp;
  g.PSE = (long long)b.nkp * b.nip;
  g.jlast = b.njp - 1;
  g.last_eo = g.PSE - V;
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
    cz_fatal(1, "czhip: k-row of %d elements needs %zu bytes of LDS (>160 KiB)\n", b.nkp, lds);
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
  if (!rows_ok(b, {P, B, OUT})) {
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
    cz_fatal(1, "czhip: the MAF kernels assume GUIDE = 2 (X(-1:sz+2), cz_maf.f90:146-148)\n");
  }
  Coef c;
  c.c1 = c.c2 = c.c3 = c.c4 = c.c5 = c.c6 = c.dd = (REAL)0;
  c.omg = omg;
  if (rows_ok(b, {P, B, OUT, ma.pvt}))
    launch_stencil_inst<VW, 512, 2, 0, MODE, 1>(P, B, OUT, c, b, par, ctx.tune.tj, skip, nblk_out, fin, ma);
  else
    launch_stencil_inst<1, 256, 2, 0, MODE, 1>(P, B, OUT, c, b, par, ctx.tune.tj, skip, nblk_out, fin, ma);
}

void reduce_partials(int n, double* dst, int accumulate, const int* skip) {
  ScopedTimer tm(LBL_REDUCE);
  hipLaunchKernelGGL(reduce_partials_k, dim3(1), dim3(1024), 0, ctx.stream, ctx.partials, n, dst, accumulate, skip);
  HIP_CHECK(hipGetLastError());
}


// Planes per chunk of the two-stage pass.  A workgroup of chunk length tj costs about tj + 3.5 plane steps (two redundant planes and
// the un-overlapped prologue); each XCD serves its band of segments with num_cu/8 * wg_per_cu resident workgroups, so a launch takes
// ceil(band * nchunk / slots) rounds of that (tools/pair_lab sweeps, profiles/r02/pair_lab_sweep_*.txt: the model ranks the measured
// times of both shapes and both precisions).  Returns the cost in units of plane steps; *tj_out the best chunk length.
inline double pair_tj_model(int nseg, int nplanes, int wg_per_cu, bool balanced, int* tj_out, double extra = 3.5) {
  // (decomposed runs: cu_reserved CUs of every XCD stay free for the exchange stream; reserve_comm_cus)
  const int slots = std::max(1, ctx.num_cu / 8 - ctx.cu_reserved) * wg_per_cu;
  double best = 1e300;
  int best_tj = std::min(16, nplanes);
  // (chunks from two planes up: on small grids, where one round of short chunks holds every item, the chain of plane steps of a workgroup is
  // the whole launch -- 64^3 FP64: 16 us per pass with chunks of 2 planes against 32 us with 12, the floor of rounds 1-2;
  // profiles/r03/small_grids_chunk_length.txt)
  for (int tj = std::min(2, nplanes); tj <= std::min(nplanes, 128); tj++) {
    const int nchunk = (nplanes + tj - 1) / tj;
    if (tj > 2 && (nplanes + tj - 2) / (tj - 1) == nchunk) continue;  // a shorter chunk gives the same count: not a candidate
    // items of the busiest XCD: a band of whole segments, or an eighth of all items with the balanced table (pair_xcd_map)
    const long long items = balanced ? ((long long)nseg * nchunk + 7) / 8 : (long long)((nseg + 7) / 8) * nchunk;
    const double cost = (double)((items + slots - 1) / slots) * (tj + extra);
    if (cost < best) best = cost, best_tj = tj;
  }
  *tj_out = best_tj;
  return best;
}

// Workgroup id -> (segment, chunk) table with equal shares per XCD.  With whole-segment bands an XCD serves ceil(nseg/8) or floor(nseg/8)
// segments: 5 against 4 at 512^3 with the 1024-thread shape, i.e. three XCDs idle for a fifth of the launch.  Here the nseg*nchunk items,
// taken in segment-major order, are cut into eight runs of equal length; an XCD's run covers whole segments except at its two ends, where
// a segment is shared with the neighbouring XCD by plane range.  Inside its run an XCD walks chunk by chunk (row-adjacent segments are then
// in the same planes at the same time and share their halo rows in the XCD's L2).  Cached per (nseg, nchunk); 8 bytes per workgroup.
// Measured (profiles/r02/ab_xcd_map.txt): +6 % at 256^3 and 384^3, where the bands are 3 against 2 segments; -1 % at 512^3 and -4 % at
// 640^3, where an XCD that finishes early only hands its share of the HBM bandwidth to the others and shared segments cost L2 hits.  So
// the table is used where the bands would leave more than a tenth of the XCD slots idle (pair_use_map).
inline const int* pair_xcd_map(int nseg, int nchunk, long long* nblk_out) {
  const long long key = ((long long)nseg << 32) | (unsigned)nchunk;
  auto it = ctx.pair_maps.find(key);
  if (it == ctx.pair_maps.end()) {
    const long long T = (long long)nseg * nchunk;
    std::vector<std::vector<std::pair<int, int>>> own(8);  // (chunk, segment) for the sort
    for (long long e = 0; e < T; e++) own[(size_t)(e * 8 / T)].push_back({(int)(e % nchunk), (int)(e / nchunk)});
    size_t most = 0;
    for (auto& v : own) {
      std::sort(v.begin(), v.end());
      most = std::max(most, v.size());
    }
    std::vector<int> tab(2 * 8 * most);
    for (size_t r = 0; r < most; r++)
      for (int x = 0; x < 8; x++) {
        const bool has = r < own[x].size();
        tab[2 * (8 * r + x)] = has ? own[x][r].second : nseg;
        tab[2 * (8 * r + x) + 1] = has ? own[x][r].first : 0;
      }
    Ctx::PairMap pm;
    pm.nblk = (long long)(8 * most);
    HIP_CHECK(hipMalloc(&pm.dev, tab.size() * sizeof(int)));
    HIP_CHECK(hipMemcpy(pm.dev, tab.data(), tab.size() * sizeof(int), hipMemcpyHostToDevice));
    it = ctx.pair_maps.emplace(key, pm).first;
  }
  *nblk_out = it->second.nblk;
  return it->second.dev;
}

// (fewer than eight segments -- small planes, boxes long in j -- leave whole XCDs without a band: 40 x 2000 x 40 ran on ONE XCD, at 0.6 of the
// single-sweep rate, until round 3; profiles/r03/non_cubic_boxes.txt)
inline bool pair_use_map(int nseg) { return ctx.tune.t2_map && 10 * nseg < 9 * 8 * ((nseg + 7) / 8); }

// k windows of the two-stage pass (Geom2).  A window of KT vectors costs a workgroup (KT + 2) / KT in loads and first-stage work and leaves it
// rows of R = KT + 2 vectors, i.e. a useful share of (LV - 2R) / LV of its LV = TB MV vectors.  That product peaks near KT = sqrt(LV) = 45 --
// and the measurements say otherwise (profiles/r04/k_windows_sweep.txt): pieces of rows shorter than about 100 vectors (1.6 KB) cost more in
// memory efficiency than they save in redundant work (512^3 FP32, R = 129: two windows of 65 are 8 % SLOWER than whole rows), pieces of 129 to
// 176 vectors are the best everywhere they were tried: 512^3 FP64 (R = 258) 0.80 -> 0.655 ms per pass with two windows, 1024^3 FP32 (R = 257)
// 598 000 -> 820 000 MLUPS, rows of 2 104 FP32 / 1 104 FP64 elements (which took single sweeps until round 3) 790 000 / 410 000 MLUPS.  Rule:
// whole rows up to 192 vectors, else the fewest windows of at most 176.
constexpr int kPairWin = 176;
inline bool pair_whole_rows_ok(int Rfull, int /*LV*/) { return Rfull <= 192; }

// c1 .. c6 all exactly 1: the kernels may leave the six multiplications out (offdiag_sum<UNIT>, cz_k_common.h)
inline bool coef_is_unit(const Coef& c) {
  return c.c1 == (REAL)1 && c.c2 == (REAL)1 && c.c3 == (REAL)1 && c.c4 == (REAL)1 && c.c5 == (REAL)1 && c.c6 == (REAL)1;
}

// two fused sweeps (jacobi2p_k); returns false when the geometry does not suit the kernel (caller falls back to two stencil_k launches)
template <int TB, int MV, int RB, int ZU, int MAF = 0, int BS = 0, int PRE = 0, int UNIT = 0>
bool launch_jacobi2_inst(const REAL* U, const REAL* B, REAL* W, const Coef& c, const Box& b, const Box& ba, int tj_req,
                         const int* skip, const Fin2& fin_in, int par, bool probe, double* model_cost, const MafArgs& ma = MafArgs(),
                         const BSrc& bs = BSrc()) {
  constexpr int V = VW;
  Geom2 g;
  const int Rfull = (b.nkp + V - 1) / V;  // rows as vectors from a vector boundary each; the last one partial where nkp % V != 0 (Geom2)
  // k windows (Geom2): whole rows where a segment of them is a decent share of the workgroup's vectors, else windows of about kPairWin vectors;
  // CZHIP_T2_KWIN / ctx.tune.t2_kwin: > 0 = vectors per window, 0 = whole rows wherever they fit, -1 = this rule
  const bool whole_fits = 2 * Rfull <= TB && 4 * Rfull < TB * MV;  // the outer rows are staged by 2R threads / halo rows would dominate
  int want = ctx.tune.t2_kwin;
  if (want < 0) want = (whole_fits && (pair_whole_rows_ok(Rfull, TB * MV) || BS)) ? 0 : kPairWin;  // (BS: the pass that makes its right-hand side reads
                                                                                                     // three or four arrays; the halo vectors of two windows cost it 6-8 % at 512^3 FP64)
  if (want == 0 && !whole_fits) want = kPairWin;
  g.R = Rfull;
  if (want > 0 && want < Rfull) {
    g.nwin = (Rfull + want - 1) / want;
    g.KT = (Rfull + g.nwin - 1) / g.nwin;
    g.hv = 1, g.KW = g.KT * V, g.R = g.KT + 2;
  }
  if (2 * g.R > TB || 4 * g.R >= TB * MV) return false;
  g.PSV = (long long)g.R * b.nip;
  g.nkp = b.nkp;
  g.PSB = (long long)b.nkp * b.nip * (long long)sizeof(REAL);
  if (g.PSB >= (1LL << 32)) return false;  // 32-bit byte offsets inside a plane
  g.jlast = b.njp - 1;
  g.last_off = (unsigned)(g.PSB - (long long)sizeof(Vec<V>));
  g.kk0 = b.kk0, g.kk1 = b.kk1, g.jj0 = b.jj0, g.jj1 = b.jj1;
  g.F0 = (long long)b.ii0 * g.R;
  g.Fend = (long long)(b.ii1 + 1) * g.R;
  g.kk0a = ba.kk0, g.kk1a = ba.kk1, g.jj0a = ba.jj0, g.jj1a = ba.jj1;
  g.F0a = (long long)ba.ii0 * g.R;
  g.Fenda = (long long)(ba.ii1 + 1) * g.R;
  g.S = TB * MV - 2 * g.R;
  g.par = par;
  g.zero_u = ZU;
  const long long nf = g.Fend - g.F0;
  g.nsegw = (int)((nf + g.S - 1) / g.S);
  g.nseg = g.nwin * g.nsegw;
  const int nplanes = b.jj1 - b.jj0 + 1;
  const size_t lds = (size_t)2 * ((g.S + 4 * g.R) + (g.S + 2 * g.R)) * sizeof(Vec<V>) + 18 * sizeof(double) +
                     (MAF ? (size_t)2 * g.R * V * sizeof(REAL) : 0);  // MAF: the table of the k metric terms
  if (lds > 160 * 1024) return false;
  const int wg_per_cu = (MAF && TB == 512) ? 1 : std::max(1, std::min((int)(160 * 1024 / lds), 2048 / TB));  // (MAF: a 512-thread workgroup of up to 256 registers per thread fills a CU's register file)
  int tj = tj_req;
  if (PRE) {
    // the preloaded form (jacobi2p_k<PRE>): chunks of PRE planes, and only where every workgroup of the pass is resident at once -- one
    // workgroup per CU (its operands live in ~190 registers per thread) less the CUs left to the exchange stream of a decomposed run
    tj = std::min(PRE, nplanes);
    const long long slots = (long long)std::max(1, ctx.num_cu / 8 - ctx.cu_reserved) * 8;  // (one per CU also for the 256 x 1 shape: measured, profiles/r04/small_grids_preloaded_shapes.txt)
    if ((long long)g.nseg * ((nplanes + tj - 1) / tj) > slots) return false;
  } else {
    const double cost = pair_tj_model(g.nseg, nplanes, wg_per_cu, pair_use_map(g.nseg), tj > 0 ? &g.TJ : &tj);
    if (model_cost) *model_cost = cost * wg_per_cu * (double)(TB * MV + g.S);  // plane steps x work per CU and step
  }
  if (tj > nplanes) tj = nplanes;
  g.TJ = tj;
  const int nchunk = (nplanes + tj - 1) / tj;
  g.band = 1;
  g.map = nullptr;
  long long nblk = 8LL * ((g.nseg + 7) / 8) * nchunk;
  if (probe) return true;
  if (pair_use_map(g.nseg)) g.map = pair_xcd_map(g.nseg, nchunk, &nblk);
  ensure_partials((size_t)2 * nblk);
  static bool attr_set = false;
  if (!attr_set) {
    HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&jacobi2p_k<V, TB, MV, RB, ZU, MAF, BS, PRE, UNIT>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  160 * 1024));
    attr_set = true;
  }
  Fin2 fin = fin_in;
  fin.counter = ctx.counter;
  {
    ScopedTimer tm(RB ? LBL_RBSOR2 : LBL_JACOBI2);
    hipLaunchKernelGGL((jacobi2p_k<V, TB, MV, RB, ZU, MAF, BS, PRE, UNIT>), dim3((unsigned)nblk), dim3(TB), lds, ctx.stream, U, B, W, c, g, ctx.partials, skip, fin, ma, bs);
  }
  HIP_CHECK(hipGetLastError());
  return true;
}

template <int RB>
bool launch_jacobi2(const REAL* U, const REAL* B, REAL* W, const Coef& c, const Box& b, const Box& ba, const int* skip,
                    const Fin2& fin, int par = 0, int zero_u = 0, bool probe = false, const MafArgs* ma = nullptr, const BSrc* bs = nullptr, int bs_op = 0) {
  if (!rows_ok(b, {U, B, W})) return false;
  if (!ma && !fastdiv_ok(c.dd)) return false;  // jacobi2p_k divides by dd with the hoisted form (cz_k_fastdiv.h); odd magnitudes take single sweeps
  if (ma && b.g != 2) return false;             // the MAF kernels index the coordinate arrays with the padded index (GUIDE = 2)
  // the stage-1 box may exceed the output box by at most one layer per side
  if (ba.ii0 < b.ii0 - 1 || ba.ii0 > b.ii0 || ba.ii1 > b.ii1 + 1 || ba.ii1 < b.ii1 || ba.jj0 < b.jj0 - 1 || ba.jj0 > b.jj0 ||
      ba.jj1 > b.jj1 + 1 || ba.jj1 < b.jj1 || ba.kk0 < b.kk0 - 1 || ba.kk0 > b.kk0 || ba.kk1 > b.kk1 + 1 || ba.kk1 < b.kk1)
    return false;
  // the two-stage march reads two layers around the box
  if (b.ii0 < 2 || b.jj0 < 2 || b.ii1 > b.nip - 3 || b.jj1 > b.njp - 3) return false;
  const Tuning& tu = ctx.tune;
  // shape: 512 threads (two workgroups per CU) or 1024 threads (one, less redundant first-stage work); fixed by CZHIP_T2 or chosen by the
  // cost model of pair_tj_model
  int tb = tu.t2_threads;
  if (tb == 0) {
    double c512 = 0.0, c1024 = 0.0;
    const bool ok512 = launch_jacobi2_inst<512, 2, RB, 0>(U, B, W, c, b, ba, tu.t2_tj, skip, fin, par, true, &c512);
    const bool ok1024 = launch_jacobi2_inst<1024, 2, RB, 0>(U, B, W, c, b, ba, tu.t2_tj, skip, fin, par, true, &c1024);
    if (!ok512 && !ok1024) return false;
    tb = (ok512 && (!ok1024 || c512 <= c1024)) ? 512 : 1024;
  }
  if (ma) {  // MAF flavour (cz_maf.f90): weights recomputed per point from the 1-D grids.  Shape by measurement (profiles/r04/
             // register_spills_priced.txt; 512^3): the 512-thread form may use 256 registers per thread and spills nothing, the 1024-thread
             // form keeps 6-28 registers in scratch.  FP64 and the red-black pass: 512 threads (FP64 Jacobi 197 000 against 176 500 MLUPS,
             // red-black 115 600 against 93 200; FP32 red-black 181 800 / 182 500); FP32 Jacobi: 1024 threads, 6 registers in scratch and
             // all the same 352 600 against 273 900 MLUPS -- the spill is priced, the shape without it is slower.
    const bool first512 = tu.t2_threads == 512 || (tu.t2_threads == 0 && (sizeof(REAL) == 8 || RB));
    if (first512 && launch_jacobi2_inst<512, 2, RB, 0, 1>(U, B, W, c, b, ba, tu.t2_tj, skip, fin, par, probe, nullptr, *ma)) return true;
    return launch_jacobi2_inst<1024, 2, RB, 0, 1>(U, B, W, c, b, ba, tu.t2_tj, skip, fin, par, probe, nullptr, *ma);
  }
  if (zero_u && bs_op != 0) {  // the right-hand side made from the operands of the vector update before the solve (jacobi2p_k<BS>)
    if (!bs || !rows_ok(b, {bs->x, bs->y, bs->z, bs->out})) return false;
    if (bs_op == 1) {
      if (tb == 512) return launch_jacobi2_inst<512, 2, RB, 1, 0, 1>(U, B, W, c, b, ba, tu.t2_tj, skip, fin, par, probe, nullptr, MafArgs(), *bs);
      return launch_jacobi2_inst<1024, 2, RB, 1, 0, 1>(U, B, W, c, b, ba, tu.t2_tj, skip, fin, par, probe, nullptr, MafArgs(), *bs);
    }
    if (tb == 512) return launch_jacobi2_inst<512, 2, RB, 1, 0, 2>(U, B, W, c, b, ba, tu.t2_tj, skip, fin, par, probe, nullptr, MafArgs(), *bs);
    return launch_jacobi2_inst<1024, 2, RB, 1, 0, 2>(U, B, W, c, b, ba, tu.t2_tj, skip, fin, par, probe, nullptr, MafArgs(), *bs);
  }
  if (zero_u) {
    if (tb == 512) return launch_jacobi2_inst<512, 2, RB, 1>(U, B, W, c, b, ba, tu.t2_tj, skip, fin, par, probe, nullptr);
    return launch_jacobi2_inst<1024, 2, RB, 1>(U, B, W, c, b, ba, tu.t2_tj, skip, fin, par, probe, nullptr);
  }
  // small grids: the preloaded form (jacobi2p_k<PRE>) where the pass is at most one workgroup per CU.  Two shapes: 256 threads x 1 vector -- a
  // 64^3 pass cut into 512 x 2 pieces is 62 workgroups, three quarters of the CUs idle while the others grind through 2.4 us plane steps --
  // and 512 x 2 with its smaller share of halo rows for the grids that fill the chip either way; the first form of the list that fits is
  // the fastest at every size measured (32^3 .. 128^3, FP32 and FP64: 1.5x at 32^3 .. 64^3, 1.05-1.1x at 96^3 and 128^3; beyond that no form
  // fits and the pipelined one stays).  t2_pre: 1 = this rule,
  // 0 = never, TB * 10 + PRE = that form or none (measurements).  A fixed shape (CZHIP_T2=1,threads,2,tj with tj > 0) keeps the pipelined form.
  if (!probe && tu.t2_pre && tu.t2_tj == 0 && tu.t2_threads != 1024) {
#define CZ_PRE(TB_, MV_, PRE_) \
  if ((tu.t2_pre == 1 || tu.t2_pre == TB_ * 10 + PRE_) && launch_jacobi2_inst<TB_, MV_, RB, 0, 0, 0, PRE_>(U, B, W, c, b, ba, 0, skip, fin, par, false, nullptr)) return true;
    CZ_PRE(256, 1, 2) CZ_PRE(256, 1, 4) CZ_PRE(512, 2, 2) CZ_PRE(512, 2, 3) CZ_PRE(512, 2, 4)
#undef CZ_PRE
  }
  // unit coefficients (what CZ sets: cz.h:169-172): the Jacobi pair without the six multiplications per point -- 2-3 % at 512^3 FP32, where the
  // vector ALU is 77 % busy; the red-black pair is bound by memory and keeps one form (profiles/r04/unit_coefficients.txt)
  if (RB == 0 && tu.unit_coef && coef_is_unit(c)) {
    if (tb == 512) return launch_jacobi2_inst<512, 2, RB, 0, 0, 0, 0, 1>(U, B, W, c, b, ba, tu.t2_tj, skip, fin, par, probe, nullptr);
    return launch_jacobi2_inst<1024, 2, RB, 0, 0, 0, 0, 1>(U, B, W, c, b, ba, tu.t2_tj, skip, fin, par, probe, nullptr);
  }
  if (tb == 512) return launch_jacobi2_inst<512, 2, RB, 0>(U, B, W, c, b, ba, tu.t2_tj, skip, fin, par, probe, nullptr);
  return launch_jacobi2_inst<1024, 2, RB, 0>(U, B, W, c, b, ba, tu.t2_tj, skip, fin, par, probe, nullptr);
}

// TWO red-black iterations per pass (rb4_k, cz_k_rb4.h): single-domain boxes, constant coefficients.  Returns false when the geometry does not suit
// the kernel (the caller then runs two fused iterations, jacobi2p_k<RB = 1>).  The k axis is cut into windows of about kRb4Win vectors whatever
// the row length: the six halo rows of a segment must stay a small share of the 1 024 vectors a workgroup holds.
// Window length by measurement (profiles/r04/rb4_two_iterations_per_pass.txt; 128^3 .. 1024^3): FP32 rows up to 50 vectors whole (128^3, 192^3:
// 1.12 x / 1.25 x the one-iteration pass), rows of 65 / 97 vectors in windows of 22 / 25 (1.14 x / 1.28 x; windows of 33 give 1.02 x / 1.09 x), rows of
// 129 / 257 in windows of 43 (1.15-1.19 x; 26 and 33 give 1.10 x); FP64 anything from 20 to 33 (1.25-1.32 x).
inline int rb4_window(int Rfull) { return VW == 4 ? (Rfull < 115 ? 26 : 48) : 28; }
constexpr int kRb4Win = 48;
// do the 256-thread preloaded forms of the one-iteration pass (launch_jacobi2: small grids) take this box?  Where they do they beat rb4_k
// (32^3 .. 80^3: rb4_k 0.67-0.85 x; from 96^3 on, where they no longer fit, 1.00-1.16 x: profiles/r04/rb4_two_iterations_per_pass.txt)
inline bool pair_small_form_fits(const Box& b) {
  if (!ctx.tune.t2_pre || ctx.tune.t2_tj != 0 || ctx.tune.t2_threads == 1024) return false;
  const int R = (b.nkp + VW - 1) / VW;
  if (4 * R >= 256) return false;
  const long long S = 256 - 2 * R, nf = (long long)(b.ii1 - b.ii0 + 1) * R, nseg = (nf + S - 1) / S;
  const int nplanes = b.jj1 - b.jj0 + 1;
  const long long slots = (long long)std::max(1, ctx.num_cu / 8 - ctx.cu_reserved) * 8;
  return nseg * ((nplanes + 1) / 2) <= slots || nseg * ((nplanes + 3) / 4) <= slots;
}
bool launch_rb4(const REAL* U, const REAL* B, REAL* W, const Coef& c, const Box& b, const int* skip, const Fin2& fin_in, int par, bool probe) {
  constexpr int V = VW, TB = 1024;
  if (!ctx.tune.rb4 || !ctx.tune.fuse_fin) return false;
  if (!rows_ok(b, {U, B, W}) || !fastdiv_ok(c.dd)) return false;
  if (b.ii0 < 2 || b.jj0 < 2 || b.ii1 > b.nip - 3 || b.jj1 > b.njp - 3) return false;
  if (ctx.tune.rb4 == 1 && pair_small_form_fits(b)) return false;  // (rb4 = 2: also there -- measurements and tests)
  Geom2 g;
  const int Rfull = (b.nkp + V - 1) / V;
  const int hv = V == 4 ? 1 : 2;  // four stages reach three elements beyond a window
  int want = ctx.tune.rb4_kwin > 0 ? ctx.tune.rb4_kwin : rb4_window(Rfull);
  g.R = Rfull;
  if (Rfull > (ctx.tune.rb4_kwin > 0 ? want : kRb4Win) + 2 * hv) {
    g.nwin = (Rfull + want - 1) / want;
    g.KT = (Rfull + g.nwin - 1) / g.nwin;
    g.hv = hv, g.KW = g.KT * V, g.R = g.KT + 2 * hv;
  }
  if (2 * g.R > TB || 8 * g.R > TB) return false;  // at least a quarter of the workgroup's vectors must be its own
  g.PSV = (long long)g.R * b.nip;
  g.nkp = b.nkp;
  g.PSB = (long long)b.nkp * b.nip * (long long)sizeof(REAL);
  if (g.PSB >= (1LL << 32)) return false;
  g.jlast = b.njp - 1;
  g.last_off = (unsigned)(g.PSB - (long long)sizeof(Vec<V>));
  g.kk0 = b.kk0, g.kk1 = b.kk1, g.jj0 = b.jj0, g.jj1 = b.jj1;
  g.F0 = (long long)b.ii0 * g.R;
  g.Fend = (long long)(b.ii1 + 1) * g.R;
  g.kk0a = b.kk0, g.kk1a = b.kk1, g.jj0a = b.jj0, g.jj1a = b.jj1, g.F0a = g.F0, g.Fenda = g.Fend;
  g.S = TB - 6 * g.R;
  g.par = par;
  g.zero_u = 0;
  const long long nf = g.Fend - g.F0;
  g.nsegw = (int)((nf + g.S - 1) / g.S);
  g.nseg = g.nwin * g.nsegw;
  const int nplanes = b.jj1 - b.jj0 + 1;
  const size_t lds = ((size_t)2 * g.R + (size_t)2 * (TB + 2 * g.R) + (size_t)6 * TB) * sizeof(Vec<V>) + 18 * sizeof(double);
  if (lds > 160 * 1024) return false;
  int tj = 0;
  pair_tj_model(g.nseg, nplanes, 1, pair_use_map(g.nseg), &tj, 7.5);  // (six redundant planes and a longer prologue per chunk)
  if (ctx.tune.rb4_tj > 0) tj = ctx.tune.rb4_tj;
  if (tj > nplanes) tj = nplanes;
  g.TJ = tj;
  const int nchunk = (nplanes + tj - 1) / tj;
  g.band = 1;
  g.map = nullptr;
  long long nblk = 8LL * ((g.nseg + 7) / 8) * nchunk;
  if (probe) return true;
  if (pair_use_map(g.nseg)) g.map = pair_xcd_map(g.nseg, nchunk, &nblk);
  ensure_partials((size_t)2 * nblk);
  static bool attr_set = false;
  if (!attr_set) {
    HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&rb4_k<V, TB, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&rb4_k<V, TB, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_set = true;
  }
  Fin2 fin = fin_in;
  fin.counter = ctx.counter;
  fin.single = 0;
  {
    // (unit coefficients: rb4_k is an arithmetic kernel -- vector ALU 93 % busy -- and the six multiplications are a fifth of a point's
    // instructions: 6-7 % at 512^3 FP32, profiles/r04/unit_coefficients.txt)
    ScopedTimer tm(LBL_RBSOR4);
    if (ctx.tune.unit_coef && coef_is_unit(c))
      hipLaunchKernelGGL((rb4_k<V, TB, 1>), dim3((unsigned)nblk), dim3(TB), lds, ctx.stream, U, B, W, c, g, ctx.partials, skip, fin);
    else
      hipLaunchKernelGGL((rb4_k<V, TB, 0>), dim3((unsigned)nblk), dim3(TB), lds, ctx.stream, U, B, W, c, g, ctx.partials, skip, fin);
  }
  HIP_CHECK(hipGetLastError());
  return true;
}

// the shell boxes of a decomposed brick, all in one launch (pair_shell_k); boxes: n x (ist,ied,jst,jed,kst,ked), 1-based
template <int RB, int MAF = 0>
void launch_pair_shell(const REAL* U, const REAL* B, REAL* W, const Coef& c, const int* sz, int g, const Box& ba, const int* boxes, int n,
                       int par, const int* skip, hipStream_t st, const MafArgs& ma = MafArgs()) {
  ShellTab s;
  s.n = n;
  int most_tiles = 0;
  size_t lds = 0;
  for (int m = 0; m < n; m++) {
    const Box b = make_box(sz, boxes + 6 * m, g);
    if (b.empty || b.ii0 < 2 || b.jj0 < 2 || b.kk0 < 2 || b.ii1 > b.nip - 3 || b.jj1 > b.njp - 3 || b.kk1 > b.nkp - 3) {
      cz_fatal(1, "czhip: pair_shell: box %d is empty or closer than two cells to the array edge\n", m);
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
    ScopedTimer tm(LBL_SHELL, st);
    hipLaunchKernelGGL((pair_shell_k<RB, MAF>), dim3(gx, (unsigned)n), dim3(256), lds, st, U, B, W, c, s, ctx.shell_partials, skip, ma);
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
void launch_ewise(REAL* Z, const REAL* X, const REAL* Y, REAL a, REAL bcoef, const Box& b, const REAL* a_dev = nullptr, const REAL* b_dev = nullptr) {
  if (b.empty) return;
  ScopedTimer tm(LBL_EWISE);
  const int nplanes = b.jj1 - b.jj0 + 1;
  if (rows_ok(b, {Z, X, Y})) {
    EGeom e = make_egeom<VW>(b);
    e.pa = a_dev, e.pb = b_dev;
    dim3 grid((unsigned)((e.Fend - e.F0 + 255) / 256), (unsigned)nplanes);
    hipLaunchKernelGGL((ewise_k<VW, OP>), grid, dim3(256), 0, ctx.stream, Z, X, Y, a, bcoef, e);
  } else {
    EGeom e = make_egeom<1>(b);
    e.pa = a_dev, e.pb = b_dev;
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
  if (rows_ok(b, {X, Y})) {
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
