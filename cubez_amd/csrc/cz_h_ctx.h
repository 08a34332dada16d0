// cz_h_ctx.h -- part of cz_kernels.hip (ONE translation unit per precision; this file is included inside its anonymous
// namespace and is not a stand-alone header): host side: tuning, per-thread context, timers, index boxes.
// ------------------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------------------
struct Tuning {
  int threads = 512, m = 2, tj = 16 /* 0 = auto */, pf = 0;  // best of tools/tune_jacobi.py at 512^3 FP32
  int fuse_fin = 1;
  int pcr_pipe = 1;                  // CZHIP_PCR_PIPE: 1 = the lexicographic line SOR (pcr, pcr_esa) in one launch per sweep (pcr_lex_wg_k), 0 = a launch per diagonal
  int pcr_rows = 0, pcr_q = 1;       // groups of NT threads per workgroup (0: chosen by the launcher), rows per thread (1, 2)
  long long pipe_spin_ticks = 200000000;  // bound of every wait inside it, in ticks of the 100 MHz wall clock (2 s)
  int pcr_wg_per_cu = 0, pcr_max_wg = 0, pcr_slots = 0;  // pcr_lex_wg_k: workgroups per CU / in all, lines per hand-off ring (0: the launcher's choice);
                                                         // CZHIP_PCR_WG_PER_CU, CZHIP_PCR_MAX_WG, CZHIP_PCR_SLOTS, czhip_set_pcr_lex_limits
  int t2_any_rows = 1;               // the two-stage pass takes row lengths that are no multiple of the vector width (CZHIP_T2_ROWS)
  int pcr_fast = 2, pcr_variant = 0;  // CZHIP_PCR=fast[,variant]: fast 0 = the per-line kernel (pcr_rb_k), 1 = table in LDS + d in LDS
                                      // (pcr_rb2_k; variant = NW*10+L), 2 = table in LDS + d in registers (pcr_line_reg_k)
  int psor_ahead = 0;                    // psor_col_k: steps ahead at which face words are asked for (0: the launcher's rule; 4, 8: measurements)
  int psor_col = 1, psor_wg_per_cu = 0;  // psor / psor_maf: 1 = the sweep in one launch (psor_col_k), 0 = a launch per tile hyperplane (psor_tile_k);
                                         // workgroups per CU of the former (0: four); CZHIP_PSOR=one_launch[,wg_per_cu], czhip_set_psor
  int t2_threads = 0, t2_mv = 2, t2_tj = 0;  // two-stage pass: threads per workgroup and planes per chunk, 0 = chosen per launch by
                                              // the balance model (pair_tj_model, cz_h_launch.h); CZHIP_T2=enable,threads,2,tj fixes them
  int t2_kwin = -1;                               // two-stage pass: vectors per k window; -1 = chosen per launch, 0 = whole rows where they fit (CZHIP_T2_KWIN)
  int rb4 = 1, rb4_kwin = 0, rb4_tj = 0;          // two red-black iterations per pass (rb4_k) in single-domain runs; vectors per k window / planes per chunk (0: the launcher's rule); CZHIP_RB4
  int unit_coef = 1;                              // the kernels' form for coefficients that are all exactly 1 (offdiag_sum<UNIT>; CZHIP_UNIT_COEF)
  int t2_pre = 1;                                 // two-stage pass on small grids: every operand of a chunk requested before its first step (jacobi2p_k<PRE>; CZHIP_T2_PRE)
  int t2_map = 1;                                 // two-stage pass: equal shares of (segment, chunk) items per XCD (CZHIP_T2_MAP=0: whole-segment bands)
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
  unsigned* pipe_ctl = nullptr;  // pcr_lex_wg_k: strip ticket, error word, one counter per strip (zeroed before every sweep)
  size_t pipe_ctl_cap = 0;
  unsigned long long* pipe_hb = nullptr;  // pcr_lex_wg_k: hand-off lines between strips ({sequence number | value} words)
  size_t pipe_hb_cap = 0;
  std::string pipe_prof_file;    // CZHIP_PCR_PIPE_PROF (development aid, -DCZ_LEX_PROF builds)
  unsigned pipe_seq = 0;         // sequence numbers handed out so far (monotonic: a stale word never matches)
  REAL* pcr_tab = nullptr;      // pcr_coef_k's table for lines of pcr_tab_n unknowns (pcr_tab_pn stages)
  int pcr_tab_n = 0, pcr_tab_pn = 0, pcr_tab_final4 = -1;
  size_t pcr_tab_cap = 0;
  unsigned long long* psor_faces = nullptr;  // psor_col_k: the face words the columns hand to their high-side neighbours (one sweep)
  size_t psor_faces_cap = 0;
  int* psor_order = nullptr;                 // ... columns in the order of their diagonals (ticket -> column)
  int psor_order_nti = 0, psor_order_ntj = 0, psor_ntickets = 0;
  unsigned* psor_ctl = nullptr;              // ... ticket and error word
  unsigned psor_seq = 0;                     // ... sweeps so far: the number a valid face word carries
  REAL* pcr_scratch = nullptr;  // pcr_rb_k<GS = 1>: a, c, d of the lines in flight (lines too long for LDS)
  size_t pcr_scratch_cap = 0;
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
  struct PairMap { int* dev = nullptr; long long nblk = 0; };
  std::map<long long, PairMap> pair_maps;        // workgroup id -> (segment, chunk) tables of the two-stage pass, key nseg << 32 | nchunk
  int num_cu = 256;
  int cu_reserved = 0;          // CUs per XCD the sweeps leave to the exchange stream (decomposed runs; reserve_comm_cus): the launch geometry counts them out
  // optional per-launch HIP-event timing of the labelled kernels (bench.py roofline leg)
  bool timing = false;
  struct Ev { hipEvent_t a, b; int label; };
  std::vector<Ev> ev_used, ev_free;
  double t_acc[16] = {0};       // per label: time [ms] and launches of events already folded (long runs)
  long long t_cnt[16] = {0};
};
thread_local Ctx ctx;  // one context per host thread (= per rank; LOCAL transport runs ranks as threads)

enum { LBL_JACOBI = 0, LBL_RBSOR, LBL_AX, LBL_RK, LBL_REDUCE, LBL_EWISE, LBL_DOT, LBL_JACOBI2, LBL_RBSOR2, LBL_PCR, LBL_SHELL, LBL_PSOR, LBL_RBSOR4, LBL_COUNT };
static_assert(LBL_COUNT <= 16, "Ctx::t_acc / t_cnt hold 16 labels");
const char* const kLabelNames[LBL_COUNT] = {"jacobi", "rbsor", "calc_ax", "calc_rk", "reduce", "ewise", "dot", "jacobi2", "rbsor2", "pcr_rb", "pair_shell", "psor", "rbsor4"};

struct ScopedTimer {
  bool on;
  Ctx::Ev ev;
  hipStream_t st;
  explicit ScopedTimer(int label, hipStream_t stream = nullptr) : on(ctx.timing), st(stream ? stream : ctx.stream) {
    if (!on) return;
    if (!ctx.ev_free.empty()) {
      ev = ctx.ev_free.back();
      ctx.ev_free.pop_back();
    } else {
      HIP_CHECK(hipEventCreate(&ev.a));
      HIP_CHECK(hipEventCreate(&ev.b));
    }
    ev.label = label;
    HIP_CHECK(hipEventRecord(ev.a, st));
  }
  ~ScopedTimer() {
    if (!on) return;
    HIP_CHECK(hipEventRecord(ev.b, st));
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
    cz_fatal(1, "czhip: index range (%d..%d, %d..%d, %d..%d) does not fit sz=(%d,%d,%d) g=%d\n", idx[0], idx[1], idx[2],
            idx[3], idx[4], idx[5], sz[0], sz[1], sz[2], g);
  }
  return b;
}

inline bool vec_ok(const Box& b, std::initializer_list<const void*> ptrs) {
  if (b.nkp % VW != 0) return false;
  for (const void* p : ptrs)
    if (p && (reinterpret_cast<uintptr_t>(p) & 15u)) return false;
  return true;
}

// the two-stage pass (jacobi2p_k) takes rows of any length and arrays of any REAL alignment (Geom2); CZHIP_T2_ROWS=0 restores the rule of
// rounds 1-2 (A/B measurements)
inline bool rows_ok(const Box& b, std::initializer_list<const void*> ptrs) {
  if (ctx.tune.t2_any_rows) {
    for (const void* p : ptrs)
      if (p && (reinterpret_cast<uintptr_t>(p) & (sizeof(REAL) - 1))) return false;
    return true;
  }
  return vec_ok(b, ptrs);
}

template <int V>
EGeom make_egeom(const Box& b) {
  EGeom e;
  e.R = (b.nkp + V - 1) / V;
  e.PSV = (long long)e.R * b.nip;
  e.nkp = b.nkp;
  e.PSE = (long long)b.nkp * b.nip;
  e.kk0 = b.kk0, e.kk1 = b.kk1, e.jj0 = b.jj0;
  e.F0 = (long long)b.ii0 * e.R;
  e.Fend = (long long)(b.ii1 + 1) * e.R;
  return e;
}
