// cz_h_linesor.h -- part of cz_kernels.hip (ONE translation unit per precision; this file is included inside its anonymous
// namespace and is not a stand-alone header): host side: launches of the line-SOR and psor kernels.
int num_stage(int n) {  // cz.h:293-300
  int b = 1;
  for (int i = 1; i < 20; i++) {
    b *= 2;
    if (n < b) return i;
  }
  return -1;
}

// the literal per-line kernel; wout: ORDER 2 only.  GS = 1: work arrays in global scratch, any line length (persistent workgroups).
template <int NW, int ORDER = 0, int MAF = 0, int FINAL4 = 0, int GS = 0>
bool try_pcr_rb(REAL* x, const REAL* msk, const REAL* rhs, PcrGeom g, REAL omg, double* res_dev, int accumulate, size_t lds_cap,
                const MafArgs& ma = MafArgs(), REAL* wout = nullptr) {
  if (FINAL4 && g.pn < 2) return false;  // (a line of one unknown has no 4x4 form)
  const long long ncol = (ORDER == 0) ? (long long)g.nhalf * g.nj
                         : (ORDER == 1) ? (long long)(std::min(g.ni - 1, g.color) - std::max(0, g.color - (g.nj - 1)) + 1)
                                        : (long long)g.ni * g.nj;
  unsigned nblk = (unsigned)((ncol + NW - 1) / NW);
  size_t lds = (size_t)NW * 6 * (g.n + 2) * sizeof(REAL) + 64 + 16 * sizeof(double);
  if (GS) {
    lds = 64 + 16 * sizeof(double);
    nblk = (unsigned)std::min<long long>(nblk, (long long)ctx.num_cu * (16 / NW));  // 16 waves per CU: latency-bound either way
    const size_t need = (size_t)nblk * NW * 6 * (g.n + 2);
    if (need > ctx.pcr_scratch_cap) {
      if (ctx.pcr_scratch) {
        HIP_CHECK(hipStreamSynchronize(ctx.stream));
        HIP_CHECK(hipFree(ctx.pcr_scratch));
      }
      HIP_CHECK(hipMalloc(&ctx.pcr_scratch, need * sizeof(REAL)));
      ctx.pcr_scratch_cap = need;
    }
  } else if (lds > lds_cap) {
    return false;
  }
  ensure_partials(nblk);
  static bool attr_set = false;
  if (!attr_set) {
    HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&pcr_rb_k<NW, ORDER, MAF, FINAL4, GS>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_set = true;
  }
  ScopedTimer tm(LBL_PCR);
  hipLaunchKernelGGL((pcr_rb_k<NW, ORDER, MAF, FINAL4, GS>), dim3(nblk), dim3(64 * NW), lds, ctx.stream, x, wout, msk, rhs, g, omg, ctx.partials, res_dev,
                     accumulate, ctx.counter, ma, ctx.pcr_scratch, ncol);
  HIP_CHECK(hipGetLastError());
  return true;
}

template <int NW, int L, int FINAL4, int ORDER, int TG = 0>
bool try_pcr_rb2_inst(REAL* x, REAL* wout, const REAL* msk, const REAL* rhs, const PcrGeom& g, REAL omg, double* res_dev, int accumulate,
                      int tab_len, int nfin, long long ncol) {
  const size_t lds = ((size_t)(TG ? 0 : tab_len) + (size_t)NW * 2 * L * (g.n + 2) + 8) * sizeof(REAL) + 32 * sizeof(double);
  if (lds > 160 * 1024) return false;
  const long long ngroups = (ncol + L - 1) / L;
  const int per_cu = (int)std::max<size_t>(1, std::min<size_t>((size_t)160 * 1024 / lds, (size_t)(32 / NW)));
  const unsigned nblk = (unsigned)std::max<long long>(1, std::min<long long>((ngroups + NW - 1) / NW, (long long)ctx.num_cu * per_cu));
  ensure_partials(nblk);
  static bool attr_set = false;
  if (!attr_set) {
    HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&pcr_rb2_k<NW, L, FINAL4, ORDER, TG>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  160 * 1024));
    attr_set = true;
  }
  ScopedTimer tm(LBL_PCR);
  hipLaunchKernelGGL((pcr_rb2_k<NW, L, FINAL4, ORDER, TG>), dim3(nblk), dim3(64 * NW), lds, ctx.stream, x, wout, msk, rhs, g, omg, ctx.pcr_tab,
                     tab_len, nfin, ctx.partials, res_dev, accumulate, ctx.counter);
  HIP_CHECK(hipGetLastError());
  return true;
}

// the line-independent coefficients of a line of n unknowns (pcr_coef_k), computed once per (n, pn, variant)
// (pcr_coef_k reduces a and c of one line in LDS: 4 (n + 2) words -- lines of up to ~10 000 FP32 / ~5 000 FP64 unknowns; false beyond)
bool ensure_pcr_table(int n, int pn, int final4, int nfin, int tab_len) {
  if (ctx.pcr_tab_n == n && ctx.pcr_tab_pn == pn && ctx.pcr_tab_final4 == final4) return true;
  const size_t coef_lds = (size_t)4 * (n + 2) * sizeof(REAL);
  if (coef_lds > 160 * 1024) return false;
  static bool attr_set = false;
  if (!attr_set) {
    HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&pcr_coef_k), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_set = true;
  }
  if ((size_t)tab_len > ctx.pcr_tab_cap) {
    if (ctx.pcr_tab) {
      HIP_CHECK(hipStreamSynchronize(ctx.stream));
      HIP_CHECK(hipFree(ctx.pcr_tab));
    }
    HIP_CHECK(hipMalloc(&ctx.pcr_tab, (size_t)tab_len * sizeof(REAL)));
    ctx.pcr_tab_cap = tab_len;
  }
  hipLaunchKernelGGL(pcr_coef_k, dim3(1), dim3(256), coef_lds, ctx.stream, ctx.pcr_tab, n, pn, nfin, final4);
  HIP_CHECK(hipGetLastError());
  ctx.pcr_tab_n = n, ctx.pcr_tab_pn = pn, ctx.pcr_tab_final4 = final4;
  ctx.pcr_perm_M = 0;  // the permuted copy is stale
  return true;
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

// the permuted coefficient table of the register form (pcr_line_reg_k): lines of up to 1024 unknowns whose table fits LDS
template <int FINAL4>
bool prep_pcr_reg_table(const PcrGeom& g, int* M_out, int* tab_len_out) {
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
  *M_out = M, *tab_len_out = tab_len;
  return true;
}

// register form (pcr_line_reg_k)
template <int FINAL4, int ORDER>
bool try_pcr_reg(REAL* x, REAL* wout, const REAL* msk, const REAL* rhs, const PcrGeom& g, REAL omg, double* res_dev, int accumulate,
                 long long ncol) {
  int M = 0, tab_len = 0;
  if (!prep_pcr_reg_table<FINAL4>(g, &M, &tab_len)) return false;
  // ONE launch shape per (entries per lane, final systems, order), the measured one (profiles/r01/pcr_variants.txt at 512^3: FP32 8 waves x 2
  // lines, FP64 and the longest lines 16 waves x 1 line; the lines of one diagonal: 4 waves).  Rounds 1-2 compiled four more shapes per case
  // for CZHIP_PCR=2,variant -- 96 kernels per precision that nothing but that variable reached, 35 % of the library's code and build time.
#define CZ_PCR_REG(M_)                                                                                                                \
  if (M == M_) {                                                                                                                      \
    if constexpr (ORDER == 1)                                                                                                         \
      return try_pcr_reg_inst<M_, 4, 1, FINAL4, ORDER>(x, wout, msk, rhs, g, omg, res_dev, accumulate, tab_len, ncol);                \
    else if constexpr (sizeof(REAL) == 4 && M_ <= 8)                                                                                  \
      return try_pcr_reg_inst<M_, 8, 2, FINAL4, ORDER>(x, wout, msk, rhs, g, omg, res_dev, accumulate, tab_len, ncol);                \
    else if constexpr (M_ == 16 && sizeof(REAL) == 8)                                                                                 \
      /* FP64 lines of 513 .. 1 024 unknowns: 4 waves per workgroup.  With 16 (128 registers per thread) the kernel kept 105-192 registers \
         in scratch; measured at 256 x 256 x 1024 the two shapes run at the same rate (27 490 / 27 465 MLUPS pcr_rb), so the one without   \
         scratch traffic stays.  The other spilling shapes were priced the same way and KEPT because the shape without spills is slower:  \
         FP32 16 entries per lane (8-32 registers spilled: 136 200 against 125 600 MLUPS with 8 waves), FP64 8 entries per lane with 4x4   \
         final systems (15-17 spilled: 83 700 against 74 700 at 512^3) -- profiles/r04/register_spills_priced.txt */                       \
      return try_pcr_reg_inst<M_, 4, 1, FINAL4, ORDER>(x, wout, msk, rhs, g, omg, res_dev, accumulate, tab_len, ncol);                \
    else                                                                                                                              \
      return try_pcr_reg_inst<M_, 16, 1, FINAL4, ORDER>(x, wout, msk, rhs, g, omg, res_dev, accumulate, tab_len, ncol);               \
  }
  CZ_PCR_REG(2) CZ_PCR_REG(4) CZ_PCR_REG(8) CZ_PCR_REG(16)
#undef CZ_PCR_REG
  return false;
}

// pcr / pcr_esa in one launch per sweep (pcr_lex_wg_k): NT threads per line, R groups of them per workgroup, Q rows per thread.
// CZHIP_PCR_PIPE=0 turns it off (one launch per diagonal instead).
template <int FINAL4, int NT, int Q, int MAF = 0>
bool try_pcr_lex_wg_inst(REAL* x, const REAL* msk, const REAL* rhs, const PcrGeom& g, REAL omg, double* res_dev, int accumulate, int nstage, int nfin,
                         const MafArgs& ma = MafArgs()) {
  // measured at 512^3 FP32 (profiles/r02/pcr_lex_*): a stage costs in proportion to the waves behind its barrier, so one group of NT
  // threads per workgroup unless the lines are short
  constexpr int kMaxT = lex_max_threads(MAF);
  if (NT > kMaxT) return false;
  int R = std::max(1, std::min(512 / NT, (g.nj + Q - 1) / Q));
  if (ctx.tune.pcr_rows > 0) R = std::max(1, std::min(kMaxT / NT, ctx.tune.pcr_rows));
  const int RS = R * Q;
  const int nstrips = (g.nj + RS - 1) / RS;
  const int ntab = MAF ? 0 : 3 * nstage + (FINAL4 ? 7 : 3);
  const size_t lds = ((size_t)(MAF ? 9 : 3) * RS * (NT + 2) + (size_t)R * NT + (size_t)ntab * NT) * sizeof(REAL) + 8 * sizeof(int) + 16 + 20 * sizeof(double);
  if (lds > 160 * 1024) return false;  // (the coefficients of every entry sit in LDS)
  static bool attr_set = false;
  if (!attr_set) {
    HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&pcr_lex_wg_k<FINAL4, NT, Q, MAF>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_set = true;
  }
  const size_t ctl_words = (size_t)kPipeCtlStride * (nstrips + 2);
  if (ctl_words > ctx.pipe_ctl_cap) {
    if (ctx.pipe_ctl) {
      HIP_CHECK(hipStreamSynchronize(ctx.stream));
      HIP_CHECK(hipFree(ctx.pipe_ctl));
    }
    HIP_CHECK(hipMalloc(&ctx.pipe_ctl, ctl_words * sizeof(unsigned)));
    ctx.pipe_ctl_cap = ctl_words;
  }
  ensure_partials((size_t)nstrips);
  // Workgroups: as many as fit the chip at once (measured at 512^3: one per CU steps no faster than two, and the strips beyond the
  // resident window then wait for a workgroup: 4.21 against 4.08 ms); CZHIP_PCR_WG_PER_CU overrides.
  const int fit = (int)std::max<size_t>(1, std::min<size_t>((size_t)160 * 1024 / lds, (size_t)(kMaxT == 512 ? 512 : 1024) / ((size_t)NT * R)));  // (128 registers per thread: 1 024 threads per CU; 256: 512)
  const int per_cu = ctx.tune.pcr_wg_per_cu > 0 ? std::min(ctx.tune.pcr_wg_per_cu, fit) : fit;
  unsigned nblk = (unsigned)std::min(nstrips, ctx.num_cu * per_cu);
  if (ctx.tune.pcr_max_wg > 0) nblk = std::min(nblk, (unsigned)ctx.tune.pcr_max_wg);
  // Hand-off buffer: per strip `nslots` lines of {sequence number | value} words.  A strip may run at most nslots lines ahead of the strip
  // below; with W workgroups resident the strip at the head of the resident window can then reach line W x nslots, which must cover its
  // whole row so that it ends and frees a workgroup for the first strip that is not resident yet.  How many workgroups ARE resident is not
  // the launcher's to know: other queues of this process (rank threads of the LOCAL transport), other processes on the device or a CU mask
  // can hold most of the chip.  The ring therefore counts on an EIGHTH of the workgroups that one per CU would give (and on one when fewer
  // than eight are launched): 32 lines per ring at 512^3, 64 MB in all.  Should even that fail, every wait inside the kernel is bounded
  // and the sweep ends with a NaN residual, which the solver loops treat as a hard error (CZ::lex_failed, cz_driver.cpp).
  int nslots = 8;  // (what a strip knows of the progress of the strip below is a step or two old: 4 slots make it wait in most steps)
  if (ctx.tune.pcr_slots > 0) {
    while (nslots < ctx.tune.pcr_slots) nslots *= 2;  // a power of two (the kernel masks the line number)
    if (ctx.tune.pcr_slots < 8) nslots = ctx.tune.pcr_slots >= 4 ? 4 : 2;
  } else {
    const int sure = std::max(1, std::min((int)nblk, ctx.num_cu) / 8);
    while (nslots < (g.ni + sure - 1) / sure + 2) nslots *= 2;
  }
  constexpr size_t HW = sizeof(REAL) == 8 ? 2 : 1;
  const size_t hb_words = (size_t)nstrips * nslots * NT * HW;
  if (hb_words > ctx.pipe_hb_cap) {
    if (ctx.pipe_hb) {
      HIP_CHECK(hipStreamSynchronize(ctx.stream));
      HIP_CHECK(hipFree(ctx.pipe_hb));
    }
    HIP_CHECK(hipMalloc(&ctx.pipe_hb, hb_words * sizeof(unsigned long long)));
    HIP_CHECK(hipMemsetAsync(ctx.pipe_hb, 0, hb_words * sizeof(unsigned long long), ctx.stream));
    ctx.pipe_hb_cap = hb_words;
    ctx.pipe_seq = 0;
  }
  if (ctx.pipe_seq > 0xffffffffu - (unsigned)(g.ni + 2)) {  // the sequence numbers would wrap: start over on a clean buffer
    HIP_CHECK(hipMemsetAsync(ctx.pipe_hb, 0, ctx.pipe_hb_cap * sizeof(unsigned long long), ctx.stream));
    ctx.pipe_seq = 0;
  }
  const unsigned seq_base = ctx.pipe_seq;  // line i of this sweep carries seq_base + i + 1: larger than anything the buffer holds
  ctx.pipe_seq += (unsigned)g.ni + 1u;
  ScopedTimer tm(LBL_PCR);
  HIP_CHECK(hipMemsetAsync(ctx.pipe_ctl, 0, ctl_words * sizeof(unsigned), ctx.stream));
  long long* prof = nullptr;
  const char* prof_env = (kLexProf && !ctx.pipe_prof_file.empty()) ? ctx.pipe_prof_file.c_str() : nullptr;  // development aid (build with -DCZ_LEX_PROF): when each strip started / ended and how long it waited
  if (prof_env) {
    HIP_CHECK(hipMalloc(&prof, (size_t)8 * nstrips * sizeof(long long)));
    HIP_CHECK(hipMemsetAsync(prof, 0, (size_t)8 * nstrips * sizeof(long long), ctx.stream));
  }
  hipLaunchKernelGGL((pcr_lex_wg_k<FINAL4, NT, Q, MAF>), dim3(nblk), dim3(NT * R), lds, ctx.stream, x, msk, rhs, g, omg, ctx.pcr_tab, nfin, R, ctx.pipe_ctl,
                     ctx.pipe_hb, nslots, seq_base, nstrips, ctx.tune.pipe_spin_ticks, ctx.partials, res_dev, accumulate, ctx.counter, prof, ma);
  HIP_CHECK(hipGetLastError());
  if (prof) {
    std::vector<long long> h((size_t)8 * nstrips);
    HIP_CHECK(hipStreamSynchronize(ctx.stream));
    HIP_CHECK(hipMemcpy(h.data(), prof, h.size() * sizeof(long long), hipMemcpyDeviceToHost));
    HIP_CHECK(hipFree(prof));
    const long long t0 = h[0];
    fprintf(stderr, "pcr_lex_wg_k<%d,%d,%d%s> n %d ni %d nj %d R %d, %u workgroups: strip: start first_line_ready end | waited (us) in n waits | wg\n", FINAL4, NT, Q, MAF ? ",maf" : "", g.n, g.ni, g.nj, R, nblk);
    const int every = std::max(1, atoi(prof_env));
    for (int sidx = 0; sidx < nstrips; sidx += every) {
      const long long* q = &h[(size_t)8 * sidx];
      fprintf(stderr, "  strip %4d: %9.2f %9.2f %9.2f | %8.2f %5lld | %3lld | phases (us): to first barrier %.1f, stages %.1f, final+relax %.1f, rotation+barrier %.1f\n", sidx,
              (q[0] - t0) * 0.01, (q[1] - t0) * 0.01, (q[2] - t0) * 0.01, q[3] * 0.01, q[4], q[5], (q[6] >> 32) * 0.01, (q[6] & 0xffffffffLL) * 0.01,
              (q[7] >> 32) * 0.01, (q[7] & 0xffffffffLL) * 0.01);
    }
  }
  return true;
}

template <int FINAL4>
bool try_pcr_lex_wg(REAL* x, const REAL* msk, const REAL* rhs, const PcrGeom& g, REAL omg, double* res_dev, int accumulate) {
  const int n = g.n, pn = g.pn;
  if (pn < (FINAL4 ? 3 : 2) || n > 1024) return false;
  const int nstage = FINAL4 ? pn - 2 : pn - 1;
  if (nstage < 2 || nstage > 10) return false;
  const int nfin = std::min(1 << nstage, n);
  ensure_pcr_table(n, pn, FINAL4, nfin, nstage * 3 * n + (FINAL4 ? 7 : 3) * nfin);
#define CZ_LEX_WG(NT_)                                                                                                               \
  if (n <= NT_)                                                                                                                     \
    return ctx.tune.pcr_q == 1 ? try_pcr_lex_wg_inst<FINAL4, NT_, 1>(x, msk, rhs, g, omg, res_dev, accumulate, nstage, nfin)          \
                               : try_pcr_lex_wg_inst<FINAL4, NT_, 2>(x, msk, rhs, g, omg, res_dev, accumulate, nstage, nfin);
  CZ_LEX_WG(64) CZ_LEX_WG(128) CZ_LEX_WG(256) CZ_LEX_WG(512) CZ_LEX_WG(1024)
#undef CZ_LEX_WG
  return false;
}

// the MAF forms of the lexicographic line SOR (pcr_maf, pcr_eda_maf, pcr_esa_maf) in one launch: a, c and d of every line reduced in LDS
bool try_pcr_lex_wg_maf(REAL* x, const REAL* msk, const REAL* rhs, const PcrGeom& g, REAL omg, double* res_dev, int accumulate, const MafArgs& ma) {
  const int n = g.n, nstage = g.pn - 1;
  if (ctx.tune.pcr_fast < 2 || ctx.tune.pcr_pipe == 0 || n > 1024 || nstage < 2 || nstage > 10) return false;
#define CZ_LEX_WG(NT_) \
  if (n <= NT_) return try_pcr_lex_wg_inst<0, NT_, 1, 1>(x, msk, rhs, g, omg, res_dev, accumulate, nstage, 0, ma);
  CZ_LEX_WG(64) CZ_LEX_WG(128) CZ_LEX_WG(256) CZ_LEX_WG(512) CZ_LEX_WG(1024)
#undef CZ_LEX_WG
  return false;
}

// fast form: coefficient table (computed once per line length and variant) + persistent right-hand-side-only kernel
template <int FINAL4, int ORDER>
bool try_pcr_rb2(REAL* x, REAL* wout, const REAL* msk, const REAL* rhs, const PcrGeom& g, REAL omg, double* res_dev, int accumulate) {
  const int n = g.n, pn = g.pn;
  if (pn < (FINAL4 ? 2 : 1) || pn > 20) return false;  // (a line of ONE unknown has no 4x4 form: the reference's 2**(pn-2) is 0 there)
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
  long long ncol;
  if (ORDER == 0) ncol = (long long)g.nhalf * g.nj;
  else if (ORDER == 1) ncol = std::min(g.ni - 1, g.color) - std::max(0, g.color - (g.nj - 1)) + 1;
  else ncol = (long long)g.ni * g.nj;
  if (fixed + 4 * per_line > 160 * 1024) {
    // the table and four lines do not fit LDS together: keep the table in global memory (one copy, read by every workgroup through the caches),
    // the right-hand sides in LDS -- as many waves per workgroup as still give two workgroups per CU, down to one line per workgroup
    if (!ensure_pcr_table(n, pn, FINAL4, nfin, tab_len)) return false;
    if (ORDER != 1) {  // (a diagonal holds few lines: small workgroups there)
      if (8 * per_line <= 72 * 1024 && try_pcr_rb2_inst<8, 1, FINAL4, ORDER, 1>(x, wout, msk, rhs, g, omg, res_dev, accumulate, tab_len, nfin, ncol)) return true;
    }
    if ((4 * per_line <= 72 * 1024 || ORDER == 1) && try_pcr_rb2_inst<4, 1, FINAL4, ORDER, 1>(x, wout, msk, rhs, g, omg, res_dev, accumulate, tab_len, nfin, ncol)) return true;
    if (try_pcr_rb2_inst<2, 1, FINAL4, ORDER, 1>(x, wout, msk, rhs, g, omg, res_dev, accumulate, tab_len, nfin, ncol)) return true;
    return try_pcr_rb2_inst<1, 1, FINAL4, ORDER, 1>(x, wout, msk, rhs, g, omg, res_dev, accumulate, tab_len, nfin, ncol);
  }
  if (!ensure_pcr_table(n, pn, FINAL4, nfin, tab_len)) return false;
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
// Forms, in the order they are tried: coefficient table + right-hand sides in registers / in LDS; table in global memory + right-hand sides in
// LDS; a, c, d of the line in LDS; a, c, d in global scratch -- every line length the reference accepts runs in one of them.
void launch_pcr_variant(REAL* x, REAL* wout, const REAL* msk, const REAL* rhs, const Box& b, const int* idx, int pn, int order, int sel,
                        int final4, REAL omg, double* res_dev, int accumulate) {
  if (b.empty) {
    if (!accumulate) HIP_CHECK(hipMemsetAsync(res_dev, 0, sizeof(double), ctx.stream));
    return;
  }
  // the literal per-line kernel (a, c and d of every line reduced in LDS, no table): lines too long for the table forms, or czhip_set_pcr_mode(0, .);
  // beyond the ~6 800 FP32 / ~3 400 FP64 unknowns whose six work rows fill LDS, the same kernel on global scratch (no limit)
  auto literal = [&](const PcrGeom& g, int acc) -> bool {
    const MafArgs no = MafArgs();
    if (order == 2) {  // pcr_j_esa: 2x2 final systems, old field in, wout out
      return try_pcr_rb<4, 2, 0, 0>(x, msk, rhs, g, omg, res_dev, acc, 80 * 1024, no, wout) || try_pcr_rb<2, 2, 0, 0>(x, msk, rhs, g, omg, res_dev, acc, 80 * 1024, no, wout) ||
             try_pcr_rb<1, 2, 0, 0>(x, msk, rhs, g, omg, res_dev, acc, 160 * 1024, no, wout) || try_pcr_rb<4, 2, 0, 0, 1>(x, msk, rhs, g, omg, res_dev, acc, 0, no, wout);
    }
    if (order == 0) {
      if (final4) return try_pcr_rb<4, 0, 0, 1>(x, msk, rhs, g, omg, res_dev, acc, 80 * 1024) || try_pcr_rb<2, 0, 0, 1>(x, msk, rhs, g, omg, res_dev, acc, 80 * 1024) ||
                         try_pcr_rb<1, 0, 0, 1>(x, msk, rhs, g, omg, res_dev, acc, 160 * 1024) || try_pcr_rb<4, 0, 0, 1, 1>(x, msk, rhs, g, omg, res_dev, acc, 0);
      return try_pcr_rb<4, 0, 0, 0>(x, msk, rhs, g, omg, res_dev, acc, 80 * 1024) || try_pcr_rb<2, 0, 0, 0>(x, msk, rhs, g, omg, res_dev, acc, 80 * 1024) ||
             try_pcr_rb<1, 0, 0, 0>(x, msk, rhs, g, omg, res_dev, acc, 160 * 1024) || try_pcr_rb<4, 0, 0, 0, 1>(x, msk, rhs, g, omg, res_dev, acc, 0);
    }
    return final4 ? (try_pcr_rb<1, 1, 0, 1>(x, msk, rhs, g, omg, res_dev, acc, 160 * 1024) || try_pcr_rb<4, 1, 0, 1, 1>(x, msk, rhs, g, omg, res_dev, acc, 0))
                  : (try_pcr_rb<1, 1, 0, 0>(x, msk, rhs, g, omg, res_dev, acc, 160 * 1024) || try_pcr_rb<4, 1, 0, 0, 1>(x, msk, rhs, g, omg, res_dev, acc, 0));
  };
  const bool table_forms = ctx.tune.pcr_fast > 0 || order == 2;
  bool ok = true;
  if (order == 1) {
    const int ni = b.ii1 - b.ii0 + 1, nj = b.jj1 - b.jj0 + 1;
    if (table_forms) {
      const PcrGeom g = make_pcr_geom(b, idx, pn, 0);
      if (ctx.tune.pcr_fast >= 2 && ctx.tune.pcr_pipe != 0 &&
          (final4 ? try_pcr_lex_wg<1>(x, msk, rhs, g, omg, res_dev, accumulate) : try_pcr_lex_wg<0>(x, msk, rhs, g, omg, res_dev, accumulate)))
        return;
    }
    for (int dgn = 0; dgn <= ni + nj - 2 && ok; dgn++) {
      const PcrGeom g = make_pcr_geom(b, idx, pn, dgn);
      ok = table_forms && (final4 ? try_pcr_rb2<1, 1>(x, wout, msk, rhs, g, omg, res_dev, accumulate || dgn > 0)
                                  : try_pcr_rb2<0, 1>(x, wout, msk, rhs, g, omg, res_dev, accumulate || dgn > 0));
      if (!ok) ok = literal(g, accumulate || dgn > 0);
    }
  } else {
    const PcrGeom g = make_pcr_geom(b, idx, pn, sel);
    ok = false;
    if (table_forms) {
      if (order == 0) ok = final4 ? try_pcr_rb2<1, 0>(x, wout, msk, rhs, g, omg, res_dev, accumulate) : try_pcr_rb2<0, 0>(x, wout, msk, rhs, g, omg, res_dev, accumulate);
      else ok = final4 ? try_pcr_rb2<1, 2>(x, wout, msk, rhs, g, omg, res_dev, accumulate) : try_pcr_rb2<0, 2>(x, wout, msk, rhs, g, omg, res_dev, accumulate);
    }
    if (!ok) ok = literal(g, accumulate);
  }
  if (!ok) {
    cz_fatal(1, "czhip: line SOR: no kernel form took a k-line of %d unknowns (pn = %d)\n", b.kk1 - b.kk0 + 1, pn);
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
  if (try_pcr_rb<4, 0, 0, 0, 1>(x, msk, rhs, g, omg, res_dev, accumulate, 0)) return;  // work arrays in global scratch: any length
  cz_fatal(1, "czhip: pcr_rb: no kernel form took a k-line of %d unknowns\n", g.n);
}

// the MAF line solvers (pcr_rb_maf, pcr_maf and their _eda / _esa forms): literal kernel, order 0 = colour `sel`, 1 = lexicographic
void launch_pcr_maf(REAL* x, const REAL* msk, const REAL* rhs, const Box& b, const int* idx, int pn, int order, int sel, REAL omg,
                    double* res_dev, int accumulate, const MafArgs& ma) {
  if (b.empty) {
    if (!accumulate) HIP_CHECK(hipMemsetAsync(res_dev, 0, sizeof(double), ctx.stream));
    return;
  }
  bool ok = true;
  auto reg = [&](const PcrGeom& g, int acc) -> bool {  // register form: lines of up to 1024 unknowns
    const int n = g.n, nstage = g.pn - 1;
    if (!ctx.tune.pcr_fast || g.pn < 1 || n > 1024) return false;
    int M = 2;
    while (64 * M < n) M *= 2;
    if ((1 << nstage) < M) return false;
    const long long ncol = (order == 0) ? (long long)g.nhalf * g.nj
                                        : (long long)(std::min(g.ni - 1, g.color) - std::max(0, g.color - (g.nj - 1)) + 1);
    auto go = [&](auto kern, int NW, int L) {
      const long long ngroups = (ncol + L - 1) / L;
      const unsigned nblk = (unsigned)std::max<long long>(1, std::min<long long>((ngroups + NW - 1) / NW, (long long)ctx.num_cu * (32 / NW)));
      ensure_partials(nblk);
      ScopedTimer tm(LBL_PCR);
      hipLaunchKernelGGL(kern, dim3(nblk), dim3(64 * NW), 0, ctx.stream, x, msk, rhs, g, omg, ma, ctx.partials, res_dev, acc, ctx.counter);
      HIP_CHECK(hipGetLastError());
      return true;
    };
#define CZ_PCR_MAF(M_)                                                                 \
  if (M == M_) {                                                                       \
    if (order == 1) return go(pcr_line_reg_maf_k<M_, 1, 1, 1>, 1, 1);                   \
    /* measured at 512^3 FP32: 4 x 1 0.749 ms, 16 x 1 0.761, 8 x 1 0.885, 8 x 2 0.876 (the other shapes are no longer compiled) */  \
    return go(pcr_line_reg_maf_k<M_, 4, 1, 0>, 4, 1);                                   \
  }
    CZ_PCR_MAF(2) CZ_PCR_MAF(4) CZ_PCR_MAF(8) CZ_PCR_MAF(16)
#undef CZ_PCR_MAF
    return false;
  };
  auto one = [&](const PcrGeom& g, int acc) {
    if (reg(g, acc)) return true;
    if (order == 0) {
      return try_pcr_rb<4, 0, 1>(x, msk, rhs, g, omg, res_dev, acc, 80 * 1024, ma) || try_pcr_rb<2, 0, 1>(x, msk, rhs, g, omg, res_dev, acc, 80 * 1024, ma) ||
             try_pcr_rb<1, 0, 1>(x, msk, rhs, g, omg, res_dev, acc, 160 * 1024, ma) || try_pcr_rb<4, 0, 1, 0, 1>(x, msk, rhs, g, omg, res_dev, acc, 0, ma);
    }
    return try_pcr_rb<1, 1, 1>(x, msk, rhs, g, omg, res_dev, acc, 160 * 1024, ma) || try_pcr_rb<4, 1, 1, 0, 1>(x, msk, rhs, g, omg, res_dev, acc, 0, ma);
  };
  if (order == 1) {
    if (try_pcr_lex_wg_maf(x, msk, rhs, make_pcr_geom(b, idx, pn, 0), omg, res_dev, accumulate, ma)) return;
    const int ni = b.ii1 - b.ii0 + 1, nj = b.jj1 - b.jj0 + 1;
    for (int dgn = 0; dgn <= ni + nj - 2 && ok; dgn++) ok = one(make_pcr_geom(b, idx, pn, dgn), accumulate || dgn > 0);
  } else {
    ok = one(make_pcr_geom(b, idx, pn, sel), accumulate);
  }
  if (!ok) {
    cz_fatal(1, "czhip: pcr_*_maf: a k-line of %d unknowns does not fit the 160 KiB of LDS\n", b.kk1 - b.kk0 + 1);
  }
}

// psor / psor_maf in one launch (psor_col_k); false when the geometry does not suit it (the caller then launches per tile hyperplane)
bool try_psor_col(REAL* p, const REAL* b, const Coef& c, const Box& bx, double* res_dev, int accumulate, const int* skip, const MafArgs* ma) {
  if (!ctx.tune.psor_col) return false;
  PsorColGeom g;
  g.nkp = bx.nkp, g.nip = bx.nip, g.njp = bx.njp;
  g.kk0 = bx.kk0, g.nk = bx.kk1 - bx.kk0 + 1, g.ii0 = bx.ii0, g.ii1 = bx.ii1, g.jj0 = bx.jj0, g.jj1 = bx.jj1;
  g.nti = (bx.ii1 - bx.ii0 + PC_T) / PC_T, g.ntj = (bx.jj1 - bx.jj0 + PC_T) / PC_T;
  g.face_words = (long long)(g.nk + PC_T) * PC_T * kPsorColHW;
  // the line streams read runs that start up to 2 (PC_T - 1) + 2 elements in front of a line and end up to 2 (PC_T - 1) + 2 NS behind it (the
  // last prefetch of a thread with i + j = 0: element G (ngroups - 1) + NS + G - 1 <= nk + 2 (PC_T - 1) + 2 NS - 1, psor_col_k), on the
  // lines ii0-1 .. ii1+1 x jj0-1 .. jj1+1: all of it must lie inside the padded array
  const long long plane = (long long)bx.nkp * bx.nip, total = plane * bx.njp;
  const long long lo = (long long)bx.kk0 + (long long)(bx.ii0 - 1) * bx.nkp + (long long)(bx.jj0 - 1) * plane - (2 * (PC_T - 1) + 4);
  const long long hi = (long long)bx.kk0 + (long long)(bx.ii1 + 1) * bx.nkp + (long long)(bx.jj1 + 1) * plane + g.nk + 2 * (PC_T - 1) + 2 * kPsorNS;
  if (bx.ii0 < 1 || bx.jj0 < 1 || bx.kk0 < 1 || lo < 0 || hi >= total) return false;
  const int ncols = g.nti * g.ntj;
  constexpr int NC = 1;  // (two columns per workgroup were measured: one barrier for ten waves makes every step twice as long -- profiles/r03)
  if (ctx.psor_order_nti != g.nti || ctx.psor_order_ntj != g.ntj) {
    // tickets in the order of the diagonals a + b; a ticket = NC columns of ONE diagonal (independent of each other), -1 = none
    std::vector<int> order;
    for (int d = 0; d <= g.nti + g.ntj - 2; d++) {
      int n = 0;
      for (int a = std::max(0, d - (g.ntj - 1)); a <= std::min(g.nti - 1, d); a++, n++) order.push_back(a + g.nti * (d - a));
      while (n % NC) order.push_back(-1), n++;
    }
    if (ctx.psor_order) {
      HIP_CHECK(hipStreamSynchronize(ctx.stream));
      HIP_CHECK(hipFree(ctx.psor_order));
    }
    HIP_CHECK(hipMalloc(&ctx.psor_order, order.size() * sizeof(int)));
    HIP_CHECK(hipMemcpy(ctx.psor_order, order.data(), order.size() * sizeof(int), hipMemcpyHostToDevice));
    ctx.psor_order_nti = g.nti, ctx.psor_order_ntj = g.ntj, ctx.psor_ntickets = (int)(order.size() / NC);
  }
  const size_t words = (size_t)2 * ncols * g.face_words;
  if (words > ctx.psor_faces_cap || ctx.psor_seq == 0xffffffffu) {
    if (ctx.psor_faces && words > ctx.psor_faces_cap) {
      HIP_CHECK(hipStreamSynchronize(ctx.stream));
      HIP_CHECK(hipFree(ctx.psor_faces));
      ctx.psor_faces = nullptr;
    }
    if (!ctx.psor_faces) {
      HIP_CHECK(hipMalloc(&ctx.psor_faces, words * sizeof(unsigned long long)));
      ctx.psor_faces_cap = words;
    }
    HIP_CHECK(hipMemsetAsync(ctx.psor_faces, 0, ctx.psor_faces_cap * sizeof(unsigned long long), ctx.stream));  // no word carries a sweep number yet
    ctx.psor_seq = 0;
  }
  if (!ctx.psor_ctl) {
    HIP_CHECK(hipMalloc(&ctx.psor_ctl, 256));
    HIP_CHECK(hipMemsetAsync(ctx.psor_ctl, 0, 256, ctx.stream));
  }
  ensure_partials((size_t)ncols);
  const int per_cu = ctx.tune.psor_wg_per_cu > 0 ? std::min(ctx.tune.psor_wg_per_cu, 8) : (ma ? 1 : 2);  // (what a CU holds: 12 waves of <= 168 registers and 2 x 71 KB of LDS)
  const int ntickets = ctx.psor_ntickets;
  const unsigned nblk = (unsigned)std::min(ntickets, ctx.num_cu * per_cu);
  const unsigned seq = ++ctx.psor_seq;
  ScopedTimer tm(LBL_PSOR);
  HIP_CHECK(hipMemsetAsync(ctx.psor_ctl, 0, 2 * sizeof(unsigned), ctx.stream));  // ticket and error word ([2]: sticky "a sweep gave up")
  // (steps ahead of their use at which the face words are asked for: psor_col_k, AH)
  if (ma)
    hipLaunchKernelGGL((psor_col_k<1, NC, 4>), dim3(nblk), dim3(psor_col_threads(NC)), 0, ctx.stream, p, b, c, g, ctx.psor_order, ntickets, ctx.psor_ctl,
                       ctx.psor_faces, seq, ctx.tune.pipe_spin_ticks, ctx.partials, res_dev, accumulate, ctx.counter, skip, *ma, nullptr);
  else if ((sizeof(REAL) == 4 && g.nk > 300 && ctx.tune.psor_ahead == 0) || ctx.tune.psor_ahead == 8)
    hipLaunchKernelGGL((psor_col_k<0, NC, 8>), dim3(nblk), dim3(psor_col_threads(NC)), 0, ctx.stream, p, b, c, g, ctx.psor_order, ntickets, ctx.psor_ctl,
                       ctx.psor_faces, seq, ctx.tune.pipe_spin_ticks, ctx.partials, res_dev, accumulate, ctx.counter, skip, MafArgs(), nullptr);
  else
    hipLaunchKernelGGL((psor_col_k<0, NC, 4>), dim3(nblk), dim3(psor_col_threads(NC)), 0, ctx.stream, p, b, c, g, ctx.psor_order, ntickets, ctx.psor_ctl,
                       ctx.psor_faces, seq, ctx.tune.pipe_spin_ticks, ctx.partials, res_dev, accumulate, ctx.counter, skip, MafArgs(), nullptr);
  HIP_CHECK(hipGetLastError());
  return true;
}

// one lexicographic SOR sweep (psor / psor_maf): one launch (psor_col_k), or a launch per tile hyperplane and the fixed-order sum of the tile partials
void launch_psor(REAL* p, const REAL* b, const Coef& c, const Box& bx, double* res_dev, int accumulate, const int* skip,
                 const MafArgs* ma) {
  if (bx.empty) {
    if (!accumulate) HIP_CHECK(hipMemsetAsync(res_dev, 0, sizeof(double), ctx.stream));
    return;
  }
  if (try_psor_col(p, b, c, bx, res_dev, accumulate, skip, ma)) return;
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
