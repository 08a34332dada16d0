// cz_driver.cpp -- restatement of the reference's host driver for a GPU-resident solve.
//
//   CZ::Setup / Solve / Evaluate   <- src/cz_cpp/cz_Evaluate.cpp:21-567   (argv parsing, solver select, allocation,
//                                      boundary conditions, dispatch, "Iter = .. Res = .." print, history file)
//   CZ::JACOBI / RBSOR / PBiCGSTAB <- src/cz_cpp/cz_Poisson.cpp:30-82, 159-235, 332-504
//   CZ::Fdot1/2, Preconditioner    <- cz_Poisson.cpp:239-270, 273-322
//   CZ::range_inner_index          <- src/cz_cpp/cz_miscel.cpp:20-52
//
// What is different from the reference, and why (details in DESIGN.md):
//   * all 3-D arrays live in HBM; the Fortran kernels are the HIP kernels of cz_kernels.hip.
//   * JACOBI ping-pongs between X and WRK instead of sweeping into WRK and copying back (12 instead of 20 B/LUP).
//   * the per-iteration "all-reduce, sqrt, history line, eps test" of cz_Poisson.cpp:67-77 runs on the device
//     (czhip_check_async); sweeps queued after convergence turn into no-ops through a device flag, so the host never
//     waits for the GPU inside the loop yet the iteration count, history and final field are exactly those of the
//     sequential loop.
//   * bc_k_ after each checked iteration (cz_Poisson.cpp:74) is skipped: sweeps write the inner box only, the
//     Dirichlet faces set at start-up are never touched, so the call is an identity.
//   * the CBrick/MPI domain decomposition is replaced by cell-ownership decomposition + RCCL (cz_comm.cpp).
#include "cz_driver.h"

#include <strings.h>
#include <ctype.h>
#include <unistd.h>

#include <cfloat>
#include <chrono>
#include <cstdint>
#include <cmath>
#include <cstring>
#include <ctime>
#include <algorithm>
#include <vector>

#include "cz_comm.h"

using namespace czhip_internal;

#define Hostonly_ if (myRank == 0)

namespace {
const char* printMethod(int t) {
  switch (t) {
    case LS_JACOBI: return "JACOBI";
    case LS_SOR2SMA: return "SOR2SMA";
    case LS_BICGSTAB: return "PBiCGSTAB";
    case LS_PSOR: return "PSOR";
    case LS_PCR_RB: return "PCR_RB";
    case LS_PCR: return "PCR";
    case LS_PCR_ESA: return "PCR_ESA";
    case LS_PCR_EDA: return "PCR_EDA";
    case LS_PCR_RB_ESA: return "PCR_RB_ESA";
    case LS_PCR_J_ESA: return "PCR_J_ESA";
    case LS_PSOR_MAF: return "PSOR_MAF";
    case LS_PCR_MAF: return "PCR_MAF";
    case LS_PCR_EDA_MAF: return "PCR_EDA_MAF";
    case LS_PCR_ESA_MAF: return "PCR_ESA_MAF";
    case LS_PCR_RB_MAF: return "PCR_RB_MAF";
    case LS_PCR_RB_ESA_MAF: return "PCR_RB_ESA_MAF";
    case LS_JACOBI_MAF: return "JACOBI_MAF";
    case LS_SOR2SMA_MAF: return "SOR2SMA_MAF";
    case LS_BICGSTAB_MAF: return "PBiCGSTAB_MAF";
    default: return "NONE";
  }
}
double now_s() {
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
constexpr int POLL_EVERY = 32;   // iterations between convergence polls of the host
constexpr int POLL_SLOTS = 4;
}  // namespace

CZ::CZ() {
  czhip_init(-1);
  HIP_CHECK(hipMalloc(&d_res, 16 * sizeof(double)));
  HIP_CHECK(hipMemset(d_res, 0, 16 * sizeof(double)));
  HIP_CHECK(hipMalloc(&d_flag, (2 + 2 * POLL_SLOTS) * sizeof(int)));
  HIP_CHECK(hipMemset(d_flag, 0, (2 + 2 * POLL_SLOTS) * sizeof(int)));
  HIP_CHECK(hipHostMalloc(&h_scal, 16 * sizeof(double), hipHostMallocDefault));
  HIP_CHECK(hipHostMalloc(&h_flag, 2 * POLL_SLOTS * sizeof(int), hipHostMallocDefault));
  cfg = CzConfig::from_env();  // the environment as this driver was created in (cz_config.h); nothing below asks it again
  overlap = cfg.num(CZV_OVERLAP, overlap);
  lag_reduce = cfg.num(CZV_LAG_REDUCE, lag_reduce);
  if (const char* sk = cfg.str(CZV_TEST_SKEW)) sscanf(sk, "%d,%d", &skew_rank, &skew_ms);  // tests: "rank,milliseconds"
}

CZ::~CZ() {
  czhip_sync();
  if (comm_cus > 0) reserve_comm_cus(0);  // the library context outlives this object
  REAL_TYPE* arrs[] = {WRK, WRK2, P, RHS, pcg_p, pcg_p_, pcg_r, pcg_r0, pcg_q, pcg_s, pcg_s_, pcg_t_, pvt, MSK};
  if (d_xc) (void)hipFree(d_xc);
  if (d_yc) (void)hipFree(d_yc);
  if (d_zc) (void)hipFree(d_zc);
  for (REAL_TYPE* a : arrs)
    if (a) czhip_free(a);
  if (d_hist) (void)hipFree(d_hist);
  (void)hipFree(d_res);
  (void)hipFree(d_flag);
  (void)hipHostFree(h_scal);
  (void)hipHostFree(h_flag);
  if (comm) comm_destroy(comm);
  if (ev_shell) (void)hipEventDestroy(ev_shell);
  if (ev_src) (void)hipEventDestroy(ev_src);
  if (ev_comm) (void)hipEventDestroy(ev_comm);
  if (ev_int) (void)hipEventDestroy(ev_int);
  for (hipEvent_t e : ev_chk)
    if (e) (void)hipEventDestroy(e);
  if (comm_stream) (void)hipStreamDestroy(comm_stream);
  if (fph) fclose(fph);
}

double CZ::npts() const {
  return (double)(innerFidx[I_plus] - innerFidx[I_minus] + 1) * (double)(innerFidx[J_plus] - innerFidx[J_minus] + 1) *
         (double)(innerFidx[K_plus] - innerFidx[K_minus] + 1);
}

// cz_miscel.cpp:20-52.  The reference always starts at 2 because its CBrick "node" bricks share one layer with the
// lower neighbour; this build's bricks own disjoint cells, so a face that borders another rank starts at 1 / ends at
// size, a physical face at 2 / size-1 (identical to the reference when numProc == 1).
double CZ::range_inner_index() {
  int ist = (nID[I_minus] < 0) ? 2 : 1, jst = (nID[J_minus] < 0) ? 2 : 1, kst = (nID[K_minus] < 0) ? 2 : 1;
  int ied = size[0], jed = size[1], ked = size[2];
  if (nID[I_plus] < 0) ied = size[0] - 1;
  if (nID[J_plus] < 0) jed = size[1] - 1;
  if (nID[K_plus] < 0) ked = size[2] - 1;
  innerFidx[I_minus] = ist, innerFidx[I_plus] = ied;
  innerFidx[J_minus] = jst, innerFidx[J_plus] = jed;
  innerFidx[K_minus] = kst, innerFidx[K_plus] = ked;
  return (double)(ied - ist + 1) * (double)(jed - jst + 1) * (double)(ked - kst + 1);
}

// cz_Evaluate.cpp:571-581
void CZ::setStrPre() {
  if (!strcasecmp(precon.c_str(), "jacobi")) pc_type = LS_JACOBI;
  else if (!strcasecmp(precon.c_str(), "sor2sma")) pc_type = LS_SOR2SMA;
  else if (!strcasecmp(precon.c_str(), "jacobi_maf")) pc_type = LS_JACOBI_MAF, SW_maf = 1;
  else if (!strcasecmp(precon.c_str(), "sor2sma_maf")) pc_type = LS_SOR2SMA_MAF, SW_maf = 1;
  else if (!strcasecmp(precon.c_str(), "pcr_rb")) pc_type = LS_PCR_RB;
  else if (!strcasecmp(precon.c_str(), "pcr_rb_esa")) pc_type = LS_PCR_RB_ESA;  // :585-587
  else if (!strcasecmp(precon.c_str(), "pcr_j_esa")) pc_type = LS_PCR_J_ESA;    // :588-590 (CZ::Preconditioner has no case for it: acts as none)
  else if (!strcasecmp(precon.c_str(), "pcr")) pc_type = LS_PCR;                // :591-593
  else if (!strcasecmp(precon.c_str(), "pcr_eda")) pc_type = LS_PCR_EDA;        // :594-596
  else if (!strcasecmp(precon.c_str(), "psor")) pc_type = LS_PSOR;
  else if (!strcasecmp(precon.c_str(), "psor_maf")) pc_type = LS_PSOR_MAF, SW_maf = 1;
  else if (!strcasecmp(precon.c_str(), "pcr_rb_maf")) pc_type = LS_PCR_RB_MAF, SW_maf = 1;          // :606-617
  else if (!strcasecmp(precon.c_str(), "pcr_rb_esa_maf")) pc_type = LS_PCR_RB_ESA_MAF, SW_maf = 1;
  else if (!strcasecmp(precon.c_str(), "pcr_maf")) pc_type = LS_PCR_MAF, SW_maf = 1;
  else if (!strcasecmp(precon.c_str(), "pcr_eda_maf")) pc_type = LS_PCR_EDA_MAF, SW_maf = 1;
  else if (!strcasecmp(precon.c_str(), "none")) pc_type = LS_NONE;
  else {
    Hostonly_ printf("Invalid preconditioner '%s' (this build: none | jacobi | psor | sor2sma | pcr | pcr_eda | pcr_rb | pcr_rb_esa | pcr_j_esa | jacobi_maf | psor_maf | sor2sma_maf | pcr_maf | pcr_eda_maf | pcr_rb_maf | pcr_rb_esa_maf)\n", precon.c_str());
    exit(0);
  }
}

// cz_Evaluate.cpp:684-803 (hot-path solvers only; see DESIGN.md "out of scope")
void CZ::setLS(const char* q) {
  if (!strcasecmp(q, "jacobi")) {
    ls_type = LS_JACOBI;
    hist_name = "jacobi.txt";
  } else if (!strcasecmp(q, "sor2sma")) {
    ls_type = LS_SOR2SMA;
    hist_name = "sor2sma.txt";
  } else if (!strcasecmp(q, "pbicgstab")) {
    ls_type = LS_BICGSTAB;
    hist_name = "pbicgstab.txt";
    setStrPre();
  } else if (!strcasecmp(q, "psor")) {  // :691-694, lexicographic point SOR
    ls_type = LS_PSOR;
    hist_name = "psor.txt";
  } else if (!strcasecmp(q, "psor_maf")) {  // :748-751
    ls_type = LS_PSOR_MAF;
    hist_name = "psor_maf.txt";
    SW_maf = 1;
  } else if (!strcasecmp(q, "pcr_rb_maf") || !strcasecmp(q, "pcr_rb_esa_maf") || !strcasecmp(q, "pcr_maf") || !strcasecmp(q, "pcr_eda_maf") ||
             !strcasecmp(q, "pcr_esa_maf")) {  // :767-797, the MAF line solvers
    ls_type = !strcasecmp(q, "pcr_rb_maf") ? LS_PCR_RB_MAF : !strcasecmp(q, "pcr_rb_esa_maf") ? LS_PCR_RB_ESA_MAF
              : !strcasecmp(q, "pcr_maf") ? LS_PCR_MAF : !strcasecmp(q, "pcr_eda_maf") ? LS_PCR_EDA_MAF : LS_PCR_ESA_MAF;
    hist_name = std::string(q) + ".txt";
    for (char& ch : hist_name) ch = (char)tolower((unsigned char)ch);
    SW_maf = 1;
  } else if (!strcasecmp(q, "pcr_rb_esa")) {  // :712-716
    ls_type = LS_PCR_RB_ESA;
    hist_name = "pcr_rb_esa.txt";
  } else if (!strcasecmp(q, "pcr_j_esa")) {  // :718-722
    ls_type = LS_PCR_J_ESA;
    hist_name = "pcr_j_esa.txt";
  } else if (!strcasecmp(q, "pcr")) {  // :724-727
    ls_type = LS_PCR;
    hist_name = "pcr.txt";
  } else if (!strcasecmp(q, "pcr_eda")) {  // :729-732
    ls_type = LS_PCR_EDA;
    hist_name = "pcr_eda.txt";
  } else if (!strcasecmp(q, "pcr_esa")) {  // :734-737
    ls_type = LS_PCR_ESA;
    hist_name = "pcr_esa.txt";
  } else if (!strcasecmp(q, "pcr_rb")) {  // :707-710, line SOR by parallel cyclic reduction
    ls_type = LS_PCR_RB;
    hist_name = "pcr_rb.txt";
  } else if (!strcasecmp(q, "jacobi_maf")) {  // :738-760, the MAF flavours
    ls_type = LS_JACOBI_MAF;
    hist_name = "jacobi_maf.txt";
    SW_maf = 1;
  } else if (!strcasecmp(q, "sor2sma_maf")) {
    ls_type = LS_SOR2SMA_MAF;
    hist_name = "sor2sma_maf.txt";
    SW_maf = 1;
  } else if (!strcasecmp(q, "pbicgstab_maf")) {
    ls_type = LS_BICGSTAB_MAF;
    hist_name = "pbicgstab_maf.txt";
    setStrPre();
    SW_maf = 1;
  } else {
    printf("Invalid solver\n");  // :799-802
    exit(0);
  }
}

// Replaces CBrick's SubDomain (cz_Evaluate.cpp:103-159): Cartesian split of the G_size nodes into G_div bricks of
// disjoint cells, rank = ri + div_i*(rj + div_j*rk).
bool CZ::decompose(int div_type) {
  if (numProc == 1) {
    G_div[0] = G_div[1] = G_div[2] = 1;  // :162-176
    for (int a = 0; a < 3; a++) size[a] = G_size[a], head[a] = 1, origin[a] = G_origin[a];
    for (int f = 0; f < 6; f++) nID[f] = -1;
    return true;
  }
  if (div_type == 0) comm_auto_division(numProc, G_size, G_div);
  int sz3[3], hd3[3], nid6[6];
  if (!comm_decompose(G_size, G_div, numProc, myRank, sz3, hd3, nid6)) return false;
  for (int a = 0; a < 3; a++) {
    size[a] = sz3[a], head[a] = hd3[a];
    origin[a] = G_origin[a] + (REAL_TYPE)(head[a] - 1) * pitch[a];  // :136-138
  }
  for (int f = 0; f < 6; f++) nID[f] = nid6[f];
  return true;
}

void CZ::ensure_hist(int n) {
  if (n <= hist_cap) return;
  if (d_hist) {
    czhip_sync();
    HIP_CHECK(hipFree(d_hist));
  }
  hist_cap = n + 1024;
  HIP_CHECK(hipMalloc(&d_hist, (size_t)hist_cap * sizeof(double)));
  HIP_CHECK(hipMemset(d_hist, 0, (size_t)hist_cap * sizeof(double)));
}

// ------------------------------------------------------------------------------------------------------------
int CZ::Setup(int argc, char** argv) {
  int div_type = 0;
  const int gc = GUIDE;
  if (argc != 7 && argc != 8 && argc != 10 && argc != 11) return 0;  // main.cpp:19

  comm_world(&myRank, &numProc);  // rank/size from the launcher environment (replaces MPI_Comm_rank/size, :44-50)

  G_size[0] = atoi(argv[1]);
  G_size[1] = atoi(argv[2]);
  G_size[2] = atoi(argv[3]);
  if (G_size[0] < 3 || G_size[1] < 3 || G_size[2] < 3) {
    Hostonly_ printf("command line error : grid size must be >= 3\n");
    return 0;
  }

  const char* q = argv[4];
  if (!strcasecmp(q, "pbicgstab") || !strcasecmp(q, "pbicgstab_maf")) {  // :63-70
    if (argc != 8 && argc != 11) {
      Hostonly_ printf("command line error : pbicgstab\n");
      exit(0);
    }
    precon = argv[7];
  }
  if (argc == 10) {  // :73-78
    div_type = 1;
    G_div[0] = atoi(argv[7]), G_div[1] = atoi(argv[8]), G_div[2] = atoi(argv[9]);
  }
  if (argc == 11) {  // :80-85
    div_type = 1;
    G_div[0] = atoi(argv[8]), G_div[1] = atoi(argv[9]), G_div[2] = atoi(argv[10]);
  }

  pitch[0] = pitch[1] = pitch[2] = 1.0 / (REAL_TYPE)(G_size[2] - 1);  // :88

  if (div_type == 1 && G_div[0] * G_div[1] * G_div[2] != numProc) {  // :93-96
    printf("\tThe number of proceees does not agree with the division size.\n");
    return 0;
  }
  ac1 = atof(argv[6]);  // :99

  if (!decompose(div_type)) return 0;
  if (numProc > 1) {
    comm = comm_create(myRank, numProc, size, nID, sizeof(REAL_TYPE), G_div);
    if (!comm) return 0;
  }

  setLS(q);
  if (!quiet) Hostonly_ {
    printf("Iterative Mehtod = %s\n", printMethod(ls_type));  // :194 (sic)
    if (ls_type == LS_BICGSTAB || ls_type == LS_BICGSTAB_MAF) printf("Preconditioner = %s\n", printMethod(pc_type));
  }

  if (!quiet) Hostonly_ {  // :210-218
    if (!(fph = fopen(hist_name.c_str(), "w"))) {
      printf("\tSorry, can't open file.\n");
      exit(0);
    }
    fprintf(fph, "Itration      Residual\n");
  }

  double sum_r = range_inner_index();  // :222-224
  plan_overlap();
  if (!Comm_SUM_1(&sum_r)) return 0;
  res_normal = 1.0 / (double)sum_r;

  // :239-288 (only the arrays the hot path touches)
  RHS = czhip_alloc_s3d(size);
  P = czhip_alloc_s3d(size);
  WRK = czhip_alloc_s3d(size);
  if (numProc > 1) {
    // The fused pass needs the two-layer exchange, single sweeps the one-layer exchange: all bricks must take the same path.
    // Bricks of an uneven division can differ (k-extent multiple of the vector width or not): agree on the weakest.
    int idx1[6];
    for (int f = 0; f < 6; f++) idx1[f] = innerFidx[f] + ((nID[f] >= 0) ? ((f & 1) ? 1 : -1) : 0);
    // (the MAF flavour of the pass needs a little more LDS -- the table of the k metric terms -- and is probed as well where it will run)
    const double mine = (pair_probe(P, WRK, RHS, size, innerFidx, idx1, GUIDE, cf[6], 0) && (!SW_maf || pair_probe(P, WRK, RHS, size, innerFidx, idx1, GUIDE, cf[6], 1))) ? 0.0 : 1.0;
    pairs_ok = comm_allreduce_max_host(comm, mine) == 0.0;
    if (cfg.has(CZV_COMM_DEBUG)) {  // one line per rank on stderr: what a multi-GPU run decided
      int dev = -1;
      (void)hipGetDevice(&dev);
      fprintf(stderr,
              "cz rank %d/%d device %d: div %dx%dx%d size %dx%dx%d head %d,%d,%d nID %d %d %d %d %d %d fused_pass=%d shell_slabs=%d overlap=%d "
              "lagged_reduce=%d comm_cus_per_xcd=%d\n",
              myRank, numProc, dev, G_div[0], G_div[1], G_div[2], size[0], size[1], size[2], head[0], head[1], head[2], nID[0], nID[1], nID[2],
              nID[3], nID[4], nID[5], (int)pairs_ok, n_shell, overlap, lag_reduce, comm_cus);
      if (myRank == 0) {  // the switches in force (cz_config.h), once per job
        std::string sw = cfg.describe(true);
        for (char& ch : sw) if (ch == '\n') ch = ' ';
        fprintf(stderr, "cz config: %s\n", sw.c_str());
      }
    }
  }
  const bool bicg = ls_type == LS_BICGSTAB || ls_type == LS_BICGSTAB_MAF;
  if (bicg) {
    pcg_p = czhip_alloc_s3d(size), pcg_p_ = czhip_alloc_s3d(size), pcg_r = czhip_alloc_s3d(size);
    pcg_r0 = czhip_alloc_s3d(size), pcg_q = czhip_alloc_s3d(size), pcg_s = czhip_alloc_s3d(size);
    pcg_s_ = czhip_alloc_s3d(size), pcg_t_ = czhip_alloc_s3d(size);
  }
  if (!quiet) Hostonly_ {
    const double arr = (double)(size[0] + 2 * gc) * (size[1] + 2 * gc) * (size[2] + 2 * gc) * sizeof(REAL_TYPE);
    printf("\n----------\n\n\tDevice memory per rank : %.1f MiB in %d arrays of (%d+4)x(%d+4)x(%d+4) %s\n", arr *
           (bicg ? 11 : 3) / 1048576.0, bicg ? 11 : 3, size[0], size[1], size[2],
           sizeof(REAL_TYPE) == 4 ? "float" : "double");
  }

  ItrMax = atoi(argv[5]);  // :330

  auto is_line = [](int t) {
    return (t >= LS_PCR && t <= LS_PCR_J_ESA) || (t >= LS_PCR_MAF && t <= LS_PCR_RB_ESA_MAF);
  };
  if (is_line(ls_type) || is_line(pc_type)) {
    // Decomposed line SOR.  The colour and Jacobi orders exchange ghost columns after each colour / iteration: with whole k-lines per
    // brick (gdv_z = 1) they reproduce the single-domain run bit for bit.  What cannot: (a) a cut along k -- every brick then solves ITS
    // piece of a line with the neighbour's last values beyond its ends, exactly what the reference's MPI path does with CBrick bricks;
    // (b) the lexicographic orders (pcr, pcr_eda, pcr_esa, and psor), one wavefront through the whole grid -- every brick sweeps its own
    // cells in that order with the ghost values of the last exchange, one Comm_S per iteration (cz_Poisson.cpp:124, 794).  Both are
    // block-local iterations, NOT the single-domain iterate; tests/test_gpu_decomp.py pins them against the same loops restated with the
    // oracle's kernels (tests/blocklocal.py).
    MSK = czhip_alloc_s3d(size);            // :242
    imask_async(MSK, size, innerFidx, gc);  // :389
  }
  if (SW_maf) {
    // :342-363 one-dimensional grid xc[i] = (i-1)*pitch (local index; the reference adds no brick origin), uploaded once;
    // :369 search_pivot_
    REAL_TYPE** dst[3] = {&d_xc, &d_yc, &d_zc};
    for (int a = 0; a < 3; a++) {
      std::vector<REAL_TYPE> c(size[a] + 2 * gc);
      for (int i = 0; i < size[a] + 2 * gc; i++) c[i] = (REAL_TYPE)(i - 1) * pitch[a];
      HIP_CHECK(hipMalloc(dst[a], c.size() * sizeof(REAL_TYPE)));
      HIP_CHECK(hipMemcpy(*dst[a], c.data(), c.size() * sizeof(REAL_TYPE), hipMemcpyHostToDevice));
    }
    pvt = czhip_alloc_s3d(size);
    search_pivot_async(pvt, size, innerFidx, gc, d_xc, d_yc, d_zc);
  }

  // :375-386  boundary values on P and RHS, ghost layers filled from the neighbours
  // (global origin + integer brick offset instead of the brick origin: bit-identical faces on every decomposition)
  // (two ghost layers incl. edges: what a fused pair of Jacobi sweeps reads; a superset of Comm_S(X, 1))
  bc_async(size, gc, P, pitch[0], G_origin, nID, head[0] - 1, head[1] - 1);
  if (!Comm_S2(P)) return 0;
  bc_async(size, gc, RHS, pitch[0], G_origin, nID, head[0] - 1, head[1] - 1);
  if (!Comm_S2(RHS)) return 0;
  czhip_sync();
  set_up = true;
  sweeps_done = 0;
  wrk_shell_tag = 0;
  return 1;
}

int CZ::Solve() {
  if (!set_up) return 0;
  double res = 0.0, flop = 0.0;
  int itr = 0;
  history.clear();
  if (profile) czhip_timing(1);  // restart the section timers (the reference's PM.start/stop around the kernels)
  czhip_sync();
  const double t0 = now_s();
  switch (ls_type) {  // :415-488
    case LS_JACOBI:
    case LS_JACOBI_MAF:
      if (0 == (itr = JACOBI(res, P, RHS, ItrMax, flop, ls_type))) return 0;
      break;
    case LS_SOR2SMA:
    case LS_SOR2SMA_MAF:
      if (0 == (itr = RBSOR(res, P, RHS, ItrMax, flop, ls_type))) return 0;
      break;
    case LS_BICGSTAB:
    case LS_BICGSTAB_MAF:
      if (0 == (itr = PBiCGSTAB(res, P, RHS, flop, ls_type))) return 0;
      break;
    case LS_PCR_RB:
      if (0 == (itr = LSOR_PCR_RB(res, P, RHS, ItrMax, flop, ls_type))) return 0;
      break;
    case LS_PSOR:
    case LS_PSOR_MAF:
      if (0 == (itr = PSOR(res, P, RHS, ItrMax, flop, ls_type))) return 0;
      break;
    case LS_PCR:
    case LS_PCR_EDA:
    case LS_PCR_ESA:
    case LS_PCR_RB_ESA:
    case LS_PCR_J_ESA:
      if (0 == (itr = LSOR_PCR_VARIANT(res, P, RHS, ItrMax, flop, ls_type))) return 0;
      break;
    case LS_PCR_MAF:
    case LS_PCR_EDA_MAF:
    case LS_PCR_ESA_MAF:
    case LS_PCR_RB_MAF:
    case LS_PCR_RB_ESA_MAF:
      if (0 == (itr = LSOR_PCR_MAF(res, P, RHS, ItrMax, flop, ls_type))) return 0;
      break;
    default:
      break;
  }
  czhip_sync();
  solve_seconds = now_s() - t0;
  result_itr = itr;
  result_res = res;

  if (fph) {
    for (size_t i = 0; i < history.size(); i++) fprintf(fph, "%6d, %13.6e\n", (int)i + 1, history[i]);  // cz_Poisson.cpp:71
    fflush(fph);
  }
  if (!quiet) Hostonly_ {  // :492-496
    printf("\n=================================\n");
    printf("Iter = %d  Res = %e\n", itr, res);
    printf("=================================\n");
  }
  return itr;
}

int CZ::Evaluate(int argc, char** argv) {
  if (!Setup(argc, argv)) return 0;
  if (!Solve()) return 0;
  if (!quiet) Hostonly_ {
    const double lups = 1.0 / res_normal * (double)(result_itr > ItrMax ? ItrMax : result_itr);
    if (ls_type != LS_BICGSTAB && ls_type != LS_BICGSTAB_MAF)
      printf("\n\tGPU time = %.6f s   %.1f MLUPS\n", solve_seconds, lups / solve_seconds * 1e-6);
    else
      printf("\n\tGPU time = %.6f s\n", solve_seconds);
  }
  if (profile) {  // :506-545 (PMlib report; rank 0 writes, the others' sections are assumed alike)
    Hostonly_ {
      FILE* fp = fopen("profiling.txt", "w");
      if (!fp) {
        printf("\tSorry, can't open 'profiling.txt' file. Write failed.\n");
        return 0;
      }
      WriteProfile(fp);
      fclose(fp);
    }
  }
  if (debug_mode == 1) {  // :550-563
    int loc[3];
    // fileout_t_ has a body only in the reference's -D_aurora_=1 build (cz_utility.f90:33-44); here CZ_SPH=1 asks for the files
    const bool dump = cfg.on(CZV_SPH, false);
    char fname[32];
    if (dump) {
      snprintf(fname, sizeof(fname), "p_%05d.sph", myRank);  // :553-554
      std::vector<REAL_TYPE> p((size_t)(size[0] + 2 * GUIDE) * (size[1] + 2 * GUIDE) * (size[2] + 2 * GUIDE));
      Field(p.data());
      if (!WriteSph(fname, p.data())) return 0;
    }
    double errmax = ErrorMax(loc);
    if (!quiet) Hostonly_ printf("\nError max = %e at (%d %d %d)\n\n", errmax, loc[0], loc[1], loc[2]);
    if (dump) {
      snprintf(fname, sizeof(fname), "e_%05d.sph", myRank);  // :560-561
      std::vector<REAL_TYPE> e;
      Exact(e);
      if (!WriteSph(fname, e.data())) return 0;
    }
  }
  return 1;
}

// Bench leg: n more iterations of the stationary solver, with the complete per-iteration work of the checked loop
// (sweep, residual reduction, convergence bookkeeping) but eps disabled so that nothing is skipped.
int CZ::Sweeps(int n) {
  if (!set_up || (ls_type != LS_JACOBI && ls_type != LS_SOR2SMA && ls_type != LS_JACOBI_MAF && ls_type != LS_SOR2SMA_MAF &&
                  ls_type != LS_PCR_RB && ls_type != LS_PSOR && ls_type != LS_PSOR_MAF && ls_type != LS_PCR && ls_type != LS_PCR_EDA && ls_type != LS_PCR_ESA &&
                  ls_type != LS_PCR_RB_ESA && ls_type != LS_PCR_J_ESA && !(ls_type >= LS_PCR_MAF && ls_type <= LS_PCR_RB_ESA_MAF)))
    return 0;
  const double keep = eps;
  eps = -1.0;
  double res = 0.0, flop = 0.0;
  history.clear();
  if (ls_type == LS_PCR || ls_type == LS_PCR_EDA || ls_type == LS_PCR_ESA || ls_type == LS_PCR_RB_ESA || ls_type == LS_PCR_J_ESA)
    LSOR_PCR_VARIANT(res, P, RHS, n, flop, ls_type);
  else if (ls_type >= LS_PCR_MAF && ls_type <= LS_PCR_RB_ESA_MAF) LSOR_PCR_MAF(res, P, RHS, n, flop, ls_type);
  else if (ls_type == LS_PSOR || ls_type == LS_PSOR_MAF) PSOR(res, P, RHS, n, flop, ls_type);
  else if (ls_type == LS_PCR_RB) LSOR_PCR_RB(res, P, RHS, n, flop, ls_type);
  else if (ls_type == LS_JACOBI || ls_type == LS_JACOBI_MAF) JACOBI(res, P, RHS, n, flop, ls_type);
  else RBSOR(res, P, RHS, n, flop, ls_type);
  eps = keep;
  sweeps_done += n;
  result_res = res;
  return n;
}

// ------------------------------------------------------------------------------------------------------------
// communication wrappers (cz_comm.cpp:23-38, 102-120 of the reference)
bool CZ::Comm_S(REAL_TYPE* X, const int* skip_flag) {
  if (numProc == 1) return true;
  return comm_halo(comm, X, skip_flag, stream());
}
bool CZ::Comm_S2(REAL_TYPE* X, const int* skip_flag) {
  if (numProc == 1) return true;
  return comm_halo2(comm, X, skip_flag, stream());
}
bool CZ::Comm_SUM_dev(double* d_val, int count, const int* skip_flag) {
  if (numProc == 1) return true;
  // INVARIANT: collectives are issued unconditionally -- never gated by a device flag the host has not read, never by host timing.
  // What a rank issues depends only on its argv, on the agreed launch count and on `stop` (see the poll in CZ::JACOBI).
  (void)skip_flag;
  return comm_allreduce_sum(comm, d_val, count, stream());
}
bool CZ::Comm_SUM_1(double* host_val) {
  if (numProc == 1) return true;
  HIP_CHECK(hipMemcpyAsync(d_res + 8, host_val, sizeof(double), hipMemcpyHostToDevice, stream()));
  if (!comm_allreduce_sum(comm, d_res + 8, 1, stream())) return false;
  HIP_CHECK(hipMemcpyAsync(h_scal + 8, d_res + 8, sizeof(double), hipMemcpyDeviceToHost, stream()));
  HIP_CHECK(hipStreamSynchronize(stream()));
  *host_val = h_scal[8];
  return true;
}

// WRK must carry the guide cells / faces of the array it ping-pongs with.  Single-domain runs know two kinds of shells that never change
// after set-up -- P's Dirichlet faces and the all-zero shell of the preconditioner's work vectors -- so the copy is made only when the
// kind changes (8 preconditioner solves per BiCGSTAB iteration otherwise pay 76 us each at 512^3).  Decomposed runs always copy: ghost
// cells change with every exchange.
void CZ::sync_wrk_shell(const REAL_TYPE* X) {
  const int tag = (numProc == 1) ? (X == P ? 1 : xx_shell_is_zero(X) ? 2 : 0) : 0;
  if (tag == 0 || tag != wrk_shell_tag) copy_shell_async(WRK, X, size, innerFidx, GUIDE);
  wrk_shell_tag = tag;
}

void CZ::skew_wait() const {
  if (skew_ms > 0 && myRank == skew_rank) usleep((useconds_t)skew_ms * 1000u);
}

// Split of the inner box for overlapped exchanges (pair_plan, cz_kernels.hip); n_shell = 0 when there is nothing to overlap.
void CZ::plan_overlap() {
  n_shell = 0;
  comm_cus = 0;
  if (numProc > 1 && overlap) n_shell = pair_plan(innerFidx, nID, shell_boxes, interior, interior1);
  // CUs per XCD the sweeps leave to the exchange stream while an interior launch fills the chip (RCCL's send/recv kernels need CUs of their
  // own for as long as a message is in flight; reserve_comm_cus, cz_kernels.hip): CZ_COMM_CUS, default 2, through the launch geometry --
  // at 512^3 FP32 the interior launch of the two-stage pass uses 30 of an XCD's 32 CUs anyway (profiles/r03/cu_reserve_cost.txt).  Every rank
  // reserves alike (argv and environment are the job's), also a brick without a rank-internal face: the launch geometry depends on it.
  if (numProc > 1 && overlap) {
    comm_cus = reserve_comm_cus(cfg.num(CZV_COMM_CUS, 2));
  } else {
    reserve_comm_cus(0);
  }
  if (n_shell == 0) return;
  if (!comm_stream) {
    // highest priority: pack / send-recv / unpack must get workgroup slots while the interior sweep (thousands of queued
    // workgroups on the compute stream) keeps the GPU full, or the overlap would turn into a tail
    int prio_least = 0, prio_greatest = 0;
    HIP_CHECK(hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest));
    HIP_CHECK(hipStreamCreateWithPriority(&comm_stream, hipStreamNonBlocking, prio_greatest));
    HIP_CHECK(hipEventCreateWithFlags(&ev_shell, hipEventDisableTiming));
    HIP_CHECK(hipEventCreateWithFlags(&ev_src, hipEventDisableTiming));
    HIP_CHECK(hipEventCreateWithFlags(&ev_comm, hipEventDisableTiming));
    HIP_CHECK(hipEventCreateWithFlags(&ev_int, hipEventDisableTiming));
    HIP_CHECK(hipEventCreateWithFlags(&ev_chk[0], hipEventDisableTiming));
    HIP_CHECK(hipEventCreateWithFlags(&ev_chk[1], hipEventDisableTiming));
  }
}

// One fused pair of sweeps (rb < 0) or one red-black iteration (rb = colour parity) of a decomposed run, src -> dst, with
// the two-layer exchange of dst hidden behind the interior:
//   stream      : [ev_src] -> interior ------------------------------------------> wait ev_comm -> fold slab sums -> (all-reduce, test)
//   comm_stream : wait ev_src -> shell slabs -> pack, send/recv, unpack -> [ev_comm]
// The slabs and the interior write disjoint cells of dst and read only src; the unpack writes ghost cells of dst.
// Returns false (nothing launched) when the split does not apply; the caller then takes the unsplit path.
bool CZ::pair_overlapped(REAL_TYPE* src, REAL_TYPE* dst, REAL_TYPE* B, const int* idx1, int rb, const int* skip, double* res_slot, const MafPtrs* maf) {
  if (n_shell == 0) return false;
  double* rs = res_slot ? res_slot : d_res;
  const int gc = GUIDE;
  hipStream_t st = stream();
  if (!pair_probe(src, dst, B, size, interior, interior1, gc, cf[6], maf ? 1 : 0)) return false;
  // The shell slabs and the interior read src and write disjoint cells of dst: they run side by side (round 2; round 1 ran the slabs first
  // on the compute stream, 49 us per pass of a corner brick during which the GPU was mostly idle).  The slabs go first on the exchange
  // stream, which has the higher priority, so the exchange starts as early as before.
  HIP_CHECK(hipEventRecord(ev_src, st));  // src is complete (and nobody reads dst any more) once everything issued so far on st is done
  HIP_CHECK(hipStreamWaitEvent(comm_stream, ev_src, 0));
  pair_shell_async(src, dst, B, size, idx1, shell_boxes, n_shell, gc, cf, ac1, rb, skip, comm_stream, maf);
  if (!comm_halo2(comm, dst, skip, comm_stream)) return false;
  HIP_CHECK(hipEventRecord(ev_comm, comm_stream));
  if (!pair_box_async(src, dst, B, size, interior, interior1, gc, cf, ac1, rb, rs, 0, skip, maf)) {
    cz_fatal(1, "error : interior launch refused after a successful probe\n");
  }
  HIP_CHECK(hipStreamWaitEvent(st, ev_comm, 0));
  pair_shell_fold_async(rs, rb >= 0 ? 1 : 0, skip, st);  // (behind ev_comm: the slabs have finished)
  return true;
}

// Drain the queue and turn the device-side bookkeeping into the loop's return values.
int CZ::finish_stationary(int itr_max, int first_itr, bool converge_check, double& res) {
  (void)first_itr;
  if (!converge_check) {
    // (a preconditioner solve inside BiCGSTAB returns without draining: 20-30 us of idle GPU per solve otherwise, see
    // profiles/r03/bicgstab_iteration_timeline_512_f64.txt)
    if (!in_precond) czhip_sync();
    return itr_max + 1;
  }
  HIP_CHECK(hipMemcpyAsync(h_flag, d_flag, 2 * sizeof(int), hipMemcpyDeviceToHost, stream()));
  HIP_CHECK(hipStreamSynchronize(stream()));
  const bool conv = h_flag[0] != 0;
  const int n_exec = conv ? h_flag[1] : itr_max;
  const size_t base = history.size();
  history.resize(base + n_exec);
  if (n_exec > 0) {
    HIP_CHECK(hipMemcpy(history.data() + base, d_hist + 1, (size_t)n_exec * sizeof(double), hipMemcpyDeviceToHost));
    res = history.back();
  }
  return conv ? n_exec : itr_max + 1;
}

// d_res[0] of the sweep(s) just issued, read back behind them: NaN = the sweep gave up a wait between its workgroups (pcr_lex_wg_k)
bool CZ::sweep_failed(const char* solver) {
  HIP_CHECK(hipMemcpyAsync(h_scal + 9, d_res, sizeof(double), hipMemcpyDeviceToHost, stream()));
  HIP_CHECK(hipStreamSynchronize(stream()));
  if (!std::isnan(h_scal[9])) return false;
  fprintf(stderr, "cz rank %d: %s: the residual of a sweep is NaN -- a hand-off between the workgroups of the one-launch lexicographic sweep did "
                  "not arrive within its bound (czhip_set_pcr_lex_timeout), or the data hold NaN.  The iterate is void.  CZHIP_PCR_PIPE=0 selects "
                  "the launch-per-diagonal form.\n", myRank, solver);
  line_error = true;
  return true;
}

// ------------------------------------------------------------------------------------------------------------
// How the sweeps of a stationary solve (or of a preconditioner solve) are executed.  Decided ONCE, before the loop, from what is known
// then -- solver flavour, iteration budget, whether convergence is tested, the geometry probes of the two-stage pass, what the ranks
// agreed on at set-up -- and then carried out by the loop without further choices.  (Round 2 chose among five launch paths inside the loop,
// iteration by iteration, with fallbacks from one to the next.)
//   kind     SINGLE  one sweep (one colour) per launch, one-layer exchange after each                         cz_Poisson.cpp:58-63, 205-215
//            WHOLE   two Jacobi sweeps / one red-black iteration per pass over memory (jacobi2p_k), two-layer exchange after the pass
//            SPLIT   the same pass as shell slabs + interior; the exchange runs on the exchange stream beside the interior (decomposed runs)
//   lag      SPLIT + convergence test: residual all-reduce and test one pass behind, on the exchange stream; three rotating buffers
//   zero_start  the start vector is identically zero and the first pass takes it as a literal (first pair of a preconditioner solve)
// What the ranks of a decomposed run must agree on is the sequence of collectives: the exchange depth (SINGLE against the fused kinds --
// `pairs_ok`, all-reduced at set-up) and the iteration at which they stop (the poll below).  WHOLE against SPLIT and lag against no lag may
// differ from brick to brick (a brick too thin to split): the same exchanges and all-reduces in the same order either way.
CZ::PassPlan CZ::plan_pass(REAL_TYPE* X, REAL_TYPE* B, int s_type, int itr_max, bool converge_check, bool x_is_zero, bool rb, bool probe_only) {
  PassPlan p;
  p.maf = (s_type == LS_JACOBI_MAF || s_type == LS_SOR2SMA_MAF) ? 1 : 0;
  p.rb = rb ? 1 : 0;
  p.comm_cus = comm_cus;
  int idx1[6];
  for (int f = 0; f < 6; f++) idx1[f] = innerFidx[f] + ((nID[f] >= 0) ? ((f & 1) ? 1 : -1) : 0);
  bool fused = czhip_use_t2() != 0 && (rb || itr_max >= 2);
  if (fused) {
    const bool mine = pair_probe(X, WRK, B, size, innerFidx, idx1, GUIDE, cf[6], p.maf) != 0;
    if (numProc > 1 && pairs_ok && !mine) {  // (pairs_ok was probed on P / WRK / RHS: same geometry, same alignment)
      cz_fatal(1, "cz rank %d: the fused pass the ranks agreed on at set-up is refused for this solve\n", myRank);
    }
    fused = numProc > 1 ? pairs_ok : mine;
  }
  if (fused) {
    p.kind = PassPlan::WHOLE, p.depth = 2;
    if (numProc > 1 && n_shell > 0 && pair_probe(X, WRK, B, size, interior, interior1, GUIDE, cf[6], p.maf)) p.kind = PassPlan::SPLIT;
    p.lag = (p.kind == PassPlan::SPLIT && converge_check && lag_reduce != 0) ? 1 : 0;
    p.buffers = p.lag ? 3 : 2;
    p.zero_start = (x_is_zero && numProc == 1 && !converge_check && !p.maf && (rb || itr_max >= 2)) ? 1 : 0;
  }
  if (probe_only) return p;  // a question (bicg_fusable), not a solve: cz_info 7..9 and the debug line keep describing solves that ran
  if (cfg.has(CZV_COMM_DEBUG) && numProc > 1 && (p.kind != last_plan.kind || p.lag != last_plan.lag || p.maf != last_plan.maf || p.rb != last_plan.rb || !plan_printed)) {
    static const char* const kinds[] = {"single sweeps", "fused pass, whole box", "fused pass, shell + interior (exchange overlapped)"};
    fprintf(stderr, "cz rank %d: %s plan: %s, exchange depth %d, lagged reduce %d, buffers %d, zero start %d, maf %d, comm CUs per XCD %d\n", myRank,
            rb ? "RBSOR" : "JACOBI", kinds[p.kind], p.depth, p.lag, p.buffers, p.zero_start, p.maf, p.comm_cus);
    plan_printed = true;
  }
  last_plan = p;
  return p;
}

// ------------------------------------------------------------------------------------------------------------
// cz_Poisson.cpp:30-82
int CZ::JACOBI(double& res, REAL_TYPE* X, REAL_TYPE* B, const int itr_max, double& flop, int s_type, bool converge_check,
               bool x_is_zero, const BMade* made) {
  const int gc = GUIDE;
  hipStream_t st = stream();
  const PassPlan plan = plan_pass(X, B, s_type, itr_max, converge_check, x_is_zero, false);
  if (made && !(plan.kind == PassPlan::WHOLE && plan.zero_start)) {  // (bicg_fusable asked the same questions before the update was withheld)
    cz_fatal(1, "error : the solve that was to make its right-hand side does not start with a whole fused pass from zero\n");
  }
  const bool maf = plan.maf != 0;  // cz_Poisson.cpp:45-53
  const MafPtrs mp{d_xc, d_yc, d_zc, nullptr};
  const MafPtrs* mpp = maf ? &mp : nullptr;
  // ping-pong partner: same guide cells / Dirichlet faces as X
  sync_wrk_shell(X);
  REAL_TYPE* buf[3] = {X, WRK, nullptr};
  const int nbuf = plan.buffers;
  const int* skip = nullptr;
  reset_ticket();
  if (converge_check) {
    ensure_hist(itr_max + 3);
    HIP_CHECK(hipMemsetAsync(d_flag, 0, 4 * sizeof(int), st));  // flag, iteration, the two snapshots of the lagged mode
    skip = d_flag;
  }
  // Fused passes apply the sweeps two at a time (temporal blocking, the intermediate field stays on chip).  Every launch is remembered so
  // that the state at the converged iteration can be produced exactly.
  struct Launch {
    int first_itr, nsweep, src;
  };
  std::vector<Launch> launches;
  int idx1[6];  // index range of the first sweep of a pair: one layer into the ghost cells across rank-internal faces
  for (int f = 0; f < 6; f++) idx1[f] = innerFidx[f] + ((nID[f] >= 0) ? ((f & 1) ? 1 : -1) : 0);
  if (plan.depth == 2 && numProc > 1) {
    // the pair reads two ghost layers (and the edge cells) of X and one of B
    if (!Comm_S2(X) || !Comm_S2(B)) return 0;
    sync_wrk_shell(X);
  }
  // Lagged mode (decomposed, checked runs): the residual all-reduce and the convergence test of pass n run on the exchange stream while
  // pass n+1 is being swept (pass n+2 waits for the test of pass n).  A pass may therefore run beyond convergence once; with THREE rotating
  // buffers it cannot touch the source or the destination of the converged pass, so the exact-iteration fix-up below still finds both.
  if (plan.lag) {
    if (!WRK2) WRK2 = czhip_alloc_s3d(size);
    copy_shell_async(WRK2, X, size, innerFidx, gc);
    buf[2] = WRK2;
  }
  if (x_is_zero && !plan.zero_start) {
    // the caller skipped its blas_clear_ (cz_Poisson.cpp:405, 441) and this solve does not take the zero as a literal: clear now
    // (guide cells / faces are zero already)
    const size_t nbytes = (size_t)(size[0] + 2 * gc) * (size[1] + 2 * gc) * (size[2] + 2 * gc) * sizeof(REAL_TYPE);
    HIP_CHECK(hipMemsetAsync(X, 0, nbytes, st));
  }
  hipEvent_t ev[POLL_SLOTS];
  int npoll = 0, cur = 0;
  bool stop = false;
  int itr = 1;
  while (itr <= itr_max && !stop) {
    REAL_TYPE* src = buf[cur];
    REAL_TYPE* dst = buf[(cur + 1) % nbuf];
    int done = 0;
    const bool pass = plan.kind != PassPlan::SINGLE && itr + 1 <= itr_max;
    if (pass && plan.kind == PassPlan::SPLIT && plan.lag) {
      const int p = (int)launches.size();
      if (p >= 2) HIP_CHECK(hipStreamWaitEvent(st, ev_chk[p & 1], 0));  // the test of pass p-2
      double* rs = d_res + ((p & 1) ? 10 : 0);
      // Pass p looks at the flag as the test of pass p-2 left it (d_flag[2 + (p & 1)], written by that test and by nothing else until
      // pass p is over): every workgroup of the pass, its pack and its unpack take the same decision.  The live flag d_flag[0] may be
      // set by the test of pass p-1 while pass p is running.
      int* snap = d_flag + 2 + (p & 1);
      if (!pair_overlapped(src, dst, B, idx1, -1, snap, rs, mpp)) return 0;  // :58 twice, :63 hidden behind the interior
      HIP_CHECK(hipEventRecord(ev_int, st));
      HIP_CHECK(hipStreamWaitEvent(comm_stream, ev_int, 0));
      if (!comm_allreduce_sum(comm, rs, 2, comm_stream)) return 0;                                            // :67, both sweeps
      check2_on_stream(comm_stream, rs, res_normal, eps, itr, d_hist, d_flag, d_flag + 1, snap);               // :69-77
      HIP_CHECK(hipEventRecord(ev_chk[p & 1], comm_stream));
      done = 2;
    } else if (pass && plan.kind == PassPlan::SPLIT) {
      if (!pair_overlapped(src, dst, B, idx1, -1, skip, nullptr, mpp)) return 0;
      if (converge_check) {
        if (!Comm_SUM_dev(d_res, 2, skip)) return 0;
        czhip_check2_async(d_res, res_normal, eps, itr, d_hist, d_flag, d_flag + 1);
      }
      done = 2;
    } else if (pass) {  // the whole inner box in one launch
      const bool in_kernel_check = converge_check && numProc == 1;
      int launched;
      if (plan.zero_start && itr == 1 && made)  // ... and the right-hand side made on the way (BiCGSTAB's vector update folded in)
        launched = pass_from_zero_made(src, dst, B, made->op, made->x, made->y, made->z, made->a, made->a_dev, made->b, size, innerFidx, idx1, gc,
                                       cf, ac1, -1, d_res, 0);
      else if (plan.zero_start && itr == 1)  // start vector identically zero (preconditioner): neither cleared in memory nor read
        launched = czhip_jacobi2_from_zero_async(src, dst, B, size, innerFidx, idx1, gc, cf, ac1, d_res);
      else if (maf)
        launched = pair_maf_async(src, dst, B, size, innerFidx, idx1, gc, d_xc, d_yc, d_zc, ac1, -1, d_res, res_normal, eps, itr,
                                  in_kernel_check ? d_hist : nullptr, d_flag, d_flag + 1, skip);  // :45-53 twice
      else
        launched = czhip_jacobi2_async(src, dst, B, size, innerFidx, idx1, gc, cf, ac1, d_res, res_normal, eps, itr,
                                       in_kernel_check ? d_hist : nullptr, d_flag, d_flag + 1, skip);  // :58 + :67-77, twice
      if (!launched) {
        cz_fatal(1, "error : fused pass refused after a successful probe\n");
      }
      if (numProc > 1) {
        if (!Comm_S2(dst, skip)) return 0;  // :63, two layers once per pair
        if (converge_check) {
          if (!Comm_SUM_dev(d_res, 2, skip)) return 0;  // :67 for both sweeps in one all-reduce
          czhip_check2_async(d_res, res_normal, eps, itr, d_hist, d_flag, d_flag + 1);
        }
      }
      done = 2;
    } else {  // one sweep: the SINGLE plan, or the odd last sweep of a fused one
      if (plan.lag) {  // the tests still in flight on the exchange stream come first
        HIP_CHECK(hipStreamWaitEvent(st, ev_chk[0], 0));
        HIP_CHECK(hipStreamWaitEvent(st, ev_chk[1], 0));
      }
      const bool fused_check = converge_check && numProc == 1;  // no all-reduce between sweep and test: one launch
      if (maf)
        jacobi_maf_async(src, dst, B, size, innerFidx, gc, d_xc, d_yc, d_zc, ac1, d_res, skip, fused_check ? 1 : 0, res_normal, eps,
                         itr, d_hist, d_flag, d_flag + 1);
      else if (fused_check)
        czhip_jacobi_checked_async(src, dst, B, size, innerFidx, gc, cf, ac1, d_res, res_normal, eps, itr, d_hist, d_flag,
                                   d_flag + 1);  // :58 + :67-77
      else
        czhip_jacobi_async(src, dst, B, size, innerFidx, gc, cf, ac1, d_res, 0, skip);  // :58
      if (!Comm_S(dst, skip)) return 0;  // :63
      if (converge_check && !fused_check) {
        if (!Comm_SUM_dev(d_res, 1, skip)) return 0;                                 // :67
        czhip_check_async(d_res, res_normal, eps, itr, d_hist, d_flag, d_flag + 1);  // :69-77
      }
      done = 1;
    }
    flop += (maf ? 66.0 : 18.0) * npts() * done;
    launches.push_back({itr, done, cur});
    itr += done;
    cur = (cur + 1) % nbuf;
    if (converge_check && launches.size() % (POLL_EVERY / 2) == 0 && itr <= itr_max) {
      // lagging, non-blocking view of the flag: look at the copy issued two polls ago
      const int slot = npoll % POLL_SLOTS;
      if (npoll >= POLL_SLOTS) HIP_CHECK(hipEventDestroy(ev[slot]));
      if (plan.lag) {
        // every rank must read the same flag here (they all stop issuing passes at the same one): the copy follows the tests of
        // all passes issued so far, which run on the other stream
        HIP_CHECK(hipStreamWaitEvent(st, ev_chk[0], 0));
        HIP_CHECK(hipStreamWaitEvent(st, ev_chk[1], 0));
      }
      HIP_CHECK(hipMemcpyAsync(h_flag + 2 * slot + 0, d_flag, sizeof(int), hipMemcpyDeviceToHost, st));
      HIP_CHECK(hipEventCreateWithFlags(&ev[slot], hipEventDisableTiming));
      HIP_CHECK(hipEventRecord(ev[slot], st));
      npoll++;
      if (npoll >= 3) {
        // INVARIANT (rank lock step): `stop` decides whether this rank issues further passes, i.e. further collectives.  It is a
        // function of h_flag[2 * old] only: a copy of the device flag taken at a fixed position of the stream (behind the tests of all
        // passes issued up to poll npoll-3), and the flag is computed from all-reduced sums -- the same bits on every rank.  When the
        // host gets to look at the copy (skew_wait: a test delays one rank here) cannot change what it reads.
        const int old = (npoll - 3) % POLL_SLOTS;
        skew_wait();
        HIP_CHECK(hipEventSynchronize(ev[old]));
        if (h_flag[2 * old] != 0) stop = true;
      }
    }
  }
  for (int i = 0; i < (npoll < POLL_SLOTS ? npoll : POLL_SLOTS); i++) HIP_CHECK(hipEventDestroy(ev[i]));
  if (plan.lag) HIP_CHECK(hipStreamSynchronize(comm_stream));  // the last tests
  last_lag = plan.lag;
  const int ret = finish_stationary(itr_max, 1, converge_check, res);

  // which buffer holds the iterate of the last executed sweep?
  int final_buf = 0;
  if (!launches.empty()) {
    const Launch* last = &launches.back();
    if (converge_check && ret <= itr_max) {  // converged at iteration `ret`: find the launch that contains it
      for (const Launch& l : launches)
        if (ret >= l.first_itr && ret < l.first_itr + l.nsweep) {
          last = &l;
          break;
        }
      if (last->nsweep == 2 && ret == last->first_itr) {
        // the first sweep of a fused pair converged: the pair wrote time n+2 into its destination; its source is
        // untouched, so one plain sweep reproduces the converged iterate (exactly what the sequential loop holds)
        if (maf)
          jacobi_maf_async(buf[last->src], buf[(last->src + 1) % nbuf], B, size, innerFidx, gc, d_xc, d_yc, d_zc, ac1, d_res + 4, nullptr, 0, 0.0, 0.0,
                           0, nullptr, nullptr, nullptr);
        else
          czhip_jacobi_async(buf[last->src], buf[(last->src + 1) % nbuf], B, size, innerFidx, gc, cf, ac1, d_res + 4, 0, nullptr);
      }
    }
    final_buf = (last->src + 1) % nbuf;
  }
  if (final_buf != 0) {
    // the result is in WRK (or WRK2).  The arrays are ours: swap the roles instead of copying back.
    if (X == P) {
      REAL_TYPE* t = P;
      P = buf[final_buf];
      if (final_buf == 1) WRK = t;
      else WRK2 = t;
    } else {
      copy_inner_async(X, buf[final_buf], size, innerFidx, gc);
    }
  }
  return ret;
}

// cz_Poisson.cpp:159-235
int CZ::RBSOR(double& res, REAL_TYPE* X, REAL_TYPE* B, const int itr_max, double& flop, int s_type, bool converge_check, bool x_is_zero,
              const BMade* made) {
  const int gc = GUIDE;
  hipStream_t st = stream();
  const PassPlan plan = plan_pass(X, B, s_type, itr_max, converge_check, x_is_zero, true);
  if (made && !(plan.kind == PassPlan::WHOLE && plan.zero_start)) {  // (see CZ::JACOBI)
    cz_fatal(1, "error : the solve that was to make its right-hand side does not start with a whole fused pass from zero\n");
  }
  if (x_is_zero && !plan.zero_start) {  // the caller skipped its blas_clear_ and this solve does not take the zero as a literal: clear now
    const size_t nb = (size_t)(size[0] + 2 * gc) * (size[1] + 2 * gc) * (size[2] + 2 * gc) * sizeof(REAL_TYPE);
    HIP_CHECK(hipMemsetAsync(X, 0, nb, st));
  }
  const bool maf = plan.maf != 0;  // cz_Poisson.cpp:190-200
  const MafPtrs mp{d_xc, d_yc, d_zc, nullptr};
  const MafPtrs* mpp = maf ? &mp : nullptr;
  const int* skip = nullptr;
  reset_ticket();
  if (converge_check) {
    ensure_hist(itr_max + 2);
    HIP_CHECK(hipMemsetAsync(d_flag, 0, 4 * sizeof(int), st));
    skip = d_flag;
  }
  // :178-186.  ip makes colour 0 the points of even GLOBAL i+j+k; the kernel's parity is relative to kst
  // (cz_solver.f90:466), which is 1 instead of 2 on a face that borders another rank.
  int ip = 0;
  if (numProc > 1) ip = (head[0] + head[1] + head[2] + 1 + innerFidx[K_minus]) % 2;

  // Fused plans: the whole iteration (colour 0, then colour 1) in ONE pass over memory, out of place X <-> WRK; decomposed runs then
  // exchange two ghost layers once per iteration.  SINGLE: the reference's two in-place colour launches with an exchange after each colour.
  const bool fused = plan.kind != PassPlan::SINGLE;
  int idx1[6];
  for (int f = 0; f < 6; f++) idx1[f] = innerFidx[f] + ((nID[f] >= 0) ? ((f & 1) ? 1 : -1) : 0);
  REAL_TYPE* buf[3] = {X, WRK, nullptr};
  const int nbuf = plan.buffers;
  int cur = 0, n_fused = 0;
  if (fused) {
    if (numProc > 1 && (!Comm_S2(X) || !Comm_S2(B))) return 0;
    sync_wrk_shell(X);
  }
  // lagged mode: residual all-reduce + test one iteration behind on the exchange stream, three rotating buffers (see CZ::JACOBI); the
  // iterate of iteration k is in buf[k % nbuf]
  if (plan.lag) {
    if (!WRK2) WRK2 = czhip_alloc_s3d(size);
    copy_shell_async(WRK2, X, size, innerFidx, gc);
    buf[2] = WRK2;
  }
  // Round 4: TWO iterations per pass over memory (rb4_k) where the whole inner box is one launch of one rank with constant coefficients.
  // Every fused launch is remembered (first iteration, iterations, source buffer) so that the state at the converged iteration can be
  // produced exactly, as in CZ::JACOBI: a converged FIRST iteration of such a pass is re-run alone from the pass's untouched input.
  struct Launch {
    int first_itr, niter, src;
  };
  std::vector<Launch> launches;
  const bool rb4 = plan.kind == PassPlan::WHOLE && numProc == 1 && !maf &&
                   czhip_rbsor4_async(X, WRK, B, size, innerFidx, gc, cf, ip, ac1, d_res, 0.0, 0.0, 0, nullptr, nullptr, nullptr, nullptr, 1) != 0;
  rb4_passes = 0;
  hipEvent_t ev[POLL_SLOTS];
  int npoll = 0;
  bool stop = false;
  int itr = 1;
  while (itr <= itr_max && !stop) {
    int done = 1;
    const bool in_kernel_check = converge_check && numProc == 1;
    REAL_TYPE* src = buf[cur];
    REAL_TYPE* dst = buf[(cur + 1) % nbuf];
    if (plan.kind == PassPlan::SPLIT && plan.lag) {
      const int p = itr - 1;
      if (p >= 2) HIP_CHECK(hipStreamWaitEvent(st, ev_chk[p & 1], 0));  // the test of iteration itr-2
      double* rs = d_res + ((p & 1) ? 10 : 0);
      int* snap = d_flag + 2 + (p & 1);  // the flag as the test of iteration itr-2 left it (see CZ::JACOBI)
      if (!pair_overlapped(src, dst, B, idx1, rb_par(gc, innerFidx, ip), snap, rs, mpp)) return 0;
      HIP_CHECK(hipEventRecord(ev_int, st));
      HIP_CHECK(hipStreamWaitEvent(comm_stream, ev_int, 0));
      if (!comm_allreduce_sum(comm, rs, 1, comm_stream)) return 0;
      check_on_stream(comm_stream, rs, res_normal, eps, itr, d_hist, d_flag, d_flag + 1, snap);
      HIP_CHECK(hipEventRecord(ev_chk[p & 1], comm_stream));
    } else if (plan.kind == PassPlan::SPLIT) {
      if (!pair_overlapped(src, dst, B, idx1, rb_par(gc, innerFidx, ip), skip, nullptr, mpp)) return 0;
      if (converge_check) {
        if (!Comm_SUM_dev(d_res, 1, skip)) return 0;
        czhip_check_async(d_res, res_normal, eps, itr, d_hist, d_flag, d_flag + 1);
      }
    } else if (plan.kind == PassPlan::WHOLE && rb4 && itr + 1 <= itr_max && !(plan.zero_start && itr == 1)) {
      if (!czhip_rbsor4_async(src, dst, B, size, innerFidx, gc, cf, ip, ac1, d_res, res_normal, eps, itr, in_kernel_check ? d_hist : nullptr, d_flag,
                              d_flag + 1, skip, 0)) {  // :205-209 twice (+ :218-230 for both iterations)
        cz_fatal(1, "error : the two-iteration red-black pass refused after a successful probe\n");
      }
      done = 2;
      rb4_passes++;
    } else if (plan.kind == PassPlan::WHOLE) {
      const int launched = (plan.zero_start && itr == 1)  // start vector identically zero (preconditioner), the right-hand side made on the way or read
                               ? pass_from_zero_made(src, dst, B, made ? made->op : 0, made ? made->x : nullptr, made ? made->y : nullptr,
                                                     made ? made->z : nullptr, made ? made->a : (REAL_TYPE)0, made ? made->a_dev : nullptr,
                                                     made ? made->b : (REAL_TYPE)0, size, innerFidx, idx1, gc, cf, ac1, ip, d_res, 0)
                           : maf ? pair_maf_async(src, dst, B, size, innerFidx, idx1, gc, d_xc, d_yc, d_zc, ac1, ip, d_res, res_normal, eps, itr,
                                                in_kernel_check ? d_hist : nullptr, d_flag, d_flag + 1, skip)  // :190-200
                               : czhip_rbsor2_async(src, dst, B, size, innerFidx, idx1, gc, cf, ip, ac1, d_res, res_normal, eps, itr,
                                                    in_kernel_check ? d_hist : nullptr, d_flag, d_flag + 1, skip);  // :205-209 (+ :218-230)
      if (!launched) {
        cz_fatal(1, "error : fused red-black iteration refused after a successful probe\n");
      }
      if (numProc > 1) {
        if (!Comm_S2(dst, skip)) return 0;  // :215
        if (converge_check) {
          if (!Comm_SUM_dev(d_res, 1, skip)) return 0;
          czhip_check_async(d_res, res_normal, eps, itr, d_hist, d_flag, d_flag + 1);
        }
      }
    } else {
      for (int color = 0; color < 2; color++) {  // :205-209
        if (maf)
          rbsor_maf_async(X, B, size, innerFidx, gc, d_xc, d_yc, d_zc, ip, color, ac1, d_res, color, skip,
                          (in_kernel_check && color == 1) ? 1 : 0, res_normal, eps, itr, d_hist, d_flag, d_flag + 1);
        else if (in_kernel_check && color == 1)
          czhip_rbsor_checked_async(X, B, size, innerFidx, gc, cf, ip, color, ac1, d_res, 1, res_normal, eps, itr, d_hist,
                                    d_flag, d_flag + 1);
        else
          czhip_rbsor_async(X, B, size, innerFidx, gc, cf, ip, color, ac1, d_res, color, skip);
        // the reference exchanges once per iteration (:215); exchanging after each colour makes the decomposed run
        // identical to the single-domain one
        if (!Comm_S(X, skip)) return 0;
      }
      if (converge_check && !in_kernel_check) {
        if (!Comm_SUM_dev(d_res, 1, skip)) return 0;
        czhip_check_async(d_res, res_normal, eps, itr, d_hist, d_flag, d_flag + 1);
      }
    }
    flop += (maf ? 66.0 : 18.0) * npts() * done;
    if (fused) {
      launches.push_back({itr, done, cur});
      cur = (cur + 1) % nbuf;
      n_fused++;
    }
    const int last_done = itr + done - 1;
    const bool poll_now = last_done / POLL_EVERY > (itr - 1) / POLL_EVERY;  // a multiple of POLL_EVERY iterations was completed by this launch
    itr += done;
    if (converge_check && poll_now && last_done < itr_max) {
      const int slot = npoll % POLL_SLOTS;
      if (npoll >= POLL_SLOTS) HIP_CHECK(hipEventDestroy(ev[slot]));
      if (plan.lag) {  // the same flag on every rank: after the tests of all iterations issued so far (see CZ::JACOBI)
        HIP_CHECK(hipStreamWaitEvent(st, ev_chk[0], 0));
        HIP_CHECK(hipStreamWaitEvent(st, ev_chk[1], 0));
      }
      HIP_CHECK(hipMemcpyAsync(h_flag + 2 * slot + 0, d_flag, sizeof(int), hipMemcpyDeviceToHost, st));
      HIP_CHECK(hipEventCreateWithFlags(&ev[slot], hipEventDisableTiming));
      HIP_CHECK(hipEventRecord(ev[slot], st));
      npoll++;
      if (npoll >= 3) {
        const int old = (npoll - 3) % POLL_SLOTS;  // same invariant as in CZ::JACOBI: a stream-ordered copy of all-reduced state
        skew_wait();
        HIP_CHECK(hipEventSynchronize(ev[old]));
        if (h_flag[2 * old] != 0) stop = true;
      }
    }
  }
  for (int i = 0; i < (npoll < POLL_SLOTS ? npoll : POLL_SLOTS); i++) HIP_CHECK(hipEventDestroy(ev[i]));
  if (plan.lag) HIP_CHECK(hipStreamSynchronize(comm_stream));  // the last tests
  last_lag = plan.lag;
  const int ret = finish_stationary(itr_max, 1, converge_check, res);
  if (n_fused > 0) {
    // out-of-place iterations: which buffer holds the iterate of the last executed iteration?  (launches after convergence were no-ops,
    // or -- lagged mode -- wrote the third buffer)
    const Launch* last = &launches.back();
    if (converge_check && ret <= itr_max) {  // converged at iteration `ret`: the launch that contains it
      for (const Launch& l : launches)
        if (ret >= l.first_itr && ret < l.first_itr + l.niter) {
          last = &l;
          break;
        }
      if (last->niter == 2 && ret == last->first_itr) {
        // the first iteration of a two-iteration pass converged: the pass wrote iteration n+2 into its destination; its source is untouched,
        // so one fused iteration from it reproduces the converged iterate (exactly what the sequential loop holds)
        if (!czhip_rbsor2_async(buf[last->src], buf[(last->src + 1) % nbuf], B, size, innerFidx, idx1, gc, cf, ip, ac1, d_res + 4, 0.0, 0.0, 0, nullptr,
                                nullptr, nullptr, nullptr)) {
          cz_fatal(1, "error : fused red-black iteration refused after a successful probe\n");
        }
      }
    }
    const int fb = (last->src + 1) % nbuf;
    if (fb != 0) {
      if (X == P) {
        REAL_TYPE* t = P;
        P = buf[fb];
        if (fb == 1) WRK = t;
        else WRK2 = t;
      } else {
        copy_inner_async(X, buf[fb], size, innerFidx, gc);
      }
    }
  }
  return ret;
}

// cz_Poisson.cpp:95-146.  Lexicographic point SOR, in place; one sweep = the launches of psor_async (tile hyperplanes).
int CZ::PSOR(double& res, REAL_TYPE* X, REAL_TYPE* B, const int itr_max, double& flop, int s_type, bool converge_check) {
  const bool maf = (s_type == LS_PSOR_MAF);  // :108-114
  const int gc = GUIDE;
  hipStream_t st = stream();
  reset_ticket();
  const int* skip = nullptr;
  if (converge_check) {
    ensure_hist(itr_max + 2);
    HIP_CHECK(hipMemsetAsync(d_flag, 0, 2 * sizeof(int), st));
    skip = d_flag;
  }
  hipEvent_t ev[POLL_SLOTS];
  int npoll = 0;
  bool stop = false;
  int itr;
  for (itr = 1; itr <= itr_max && !stop; itr++) {
    psor_async(X, B, size, innerFidx, gc, cf, maf ? d_xc : nullptr, d_yc, d_zc, ac1, d_res, 0, skip);  // :108-122
    flop += (maf ? 66.0 : 18.0) * npts();
    if (!Comm_S(X, skip)) return 0;  // :124 (decomposed: block-local sweeps, ghosts of the last exchange)
    if (converge_check) {
      if (!Comm_SUM_dev(d_res, 1, skip)) return 0;                                 // :127
      czhip_check_async(d_res, res_normal, eps, itr, d_hist, d_flag, d_flag + 1);  // :128-141 on the device
      if (itr % POLL_EVERY == 0 && itr < itr_max) {  // lagging, non-blocking view of the flag (as in RBSOR)
        const int slot = npoll % POLL_SLOTS;
        if (npoll >= POLL_SLOTS) HIP_CHECK(hipEventDestroy(ev[slot]));
        HIP_CHECK(hipMemcpyAsync(h_flag + 2 * slot + 0, d_flag, sizeof(int), hipMemcpyDeviceToHost, st));
        HIP_CHECK(hipEventCreateWithFlags(&ev[slot], hipEventDisableTiming));
        HIP_CHECK(hipEventRecord(ev[slot], st));
        npoll++;
        if (npoll >= 3) {
          const int old = (npoll - 3) % POLL_SLOTS;
          HIP_CHECK(hipEventSynchronize(ev[old]));
          if (h_flag[2 * old] != 0) stop = true;
        }
      }
    }
  }
  for (int i = 0; i < (npoll < POLL_SLOTS ? npoll : POLL_SLOTS); i++) HIP_CHECK(hipEventDestroy(ev[i]));
  const int ret = finish_stationary(itr_max, 1, converge_check, res);
  if (psor_failed()) {  // a column of the one-launch sweep gave up waiting for the columns before it: the iterate is void
    fprintf(stderr, "cz rank %d: %s: a sweep gave up a hand-off between its workgroups (bound: czhip_set_pcr_lex_timeout); the iterate is void.  "
                    "CZHIP_PSOR=0 selects the launch-per-tile-hyperplane form.\n", myRank, printMethod(s_type));
    line_error = true;
    return 0;
  }
  return ret;
}

// cz_Poisson.cpp:621-742 (pcr_rb_esa), :745-826 (pcr), :910-1005 (pcr_esa), :1008-1095 (pcr_j_esa): the line-SOR variants that end
// in 4x4 systems and / or visit the columns in another order; one loop, the variant picks order and final stage.
int CZ::LSOR_PCR_VARIANT(double& res, REAL_TYPE* X, REAL_TYPE* B, const int itr_max, double& flop, int s_type, bool converge_check) {
  const int gc = GUIDE;
  hipStream_t st = stream();
  reset_ticket();
  const int n = innerFidx[K_plus] - innerFidx[K_minus] + 1;
  const int pn = pcr_num_stage(n);
  if (pn < 0) {
    printf("error : number of stage\n");
    exit(0);
  }
  const int final4 = (s_type == LS_PCR_J_ESA || s_type == LS_PCR_EDA) ? 0 : 1;
  const int order = (s_type == LS_PCR_RB_ESA) ? 0 : (s_type == LS_PCR_J_ESA) ? 2 : 1;
  const int stages = final4 ? pn - 2 : pn - 1;
  const double fin = final4 ? (double)(1 << (pn - 2)) * (s_type == LS_PCR ? 74.0 : 78.0) : (double)(1 << (pn - 1)) * 9.0;
  const int* skip = nullptr;
  if (converge_check) {
    ensure_hist(itr_max + 2);
    HIP_CHECK(hipMemsetAsync(d_flag, 0, 2 * sizeof(int), st));
    skip = d_flag;
  }
  (void)skip;
  int itr;
  for (itr = 1; itr <= itr_max; itr++) {
    if (order == 0) {
      for (int color = 0; color < 2; color++) {  // :685-690; global colouring and an exchange per colour as in LSOR_PCR_RB
        pcr_variant_async(X, nullptr, MSK, B, size, innerFidx, gc, pn, 0, (color + head[0] + head[1]) & 1, final4, ac1, d_res, color);
        if (!Comm_S(X)) return 0;
      }
    } else if (order == 1) {
      pcr_variant_async(X, nullptr, MSK, B, size, innerFidx, gc, pn, 1, 0, final4, ac1, d_res, 0);  // :783-786, :966-969
      if (!Comm_S(X)) return 0;  // :794 (decomposed: block-local, see CZ::Setup)
    } else {
      pcr_variant_async(X, WRK, MSK, B, size, innerFidx, gc, pn, 2, 0, final4, ac1, d_res, 0);  // :1061-1064
      copy_inner_async(X, WRK, size, innerFidx, gc);
      if (!Comm_S(X)) return 0;  // :1068
    }
    flop += (npts() / n) * (n * 6.0 + n * (double)stages * 14.0 + fin + n * 6.0 + 6.0);
    if (converge_check) {
      // long launches: a host round trip per iteration is negligible, the reference's sequential test is kept as is
      if (!Comm_SUM_dev(d_res, 1)) return 0;
      czhip_check_async(d_res, res_normal, eps, itr, d_hist, d_flag, d_flag + 1);
      HIP_CHECK(hipMemcpyAsync(h_flag, d_flag, sizeof(int), hipMemcpyDeviceToHost, st));
      if (sweep_failed(printMethod(s_type))) return 0;  // (synchronises)
      if (h_flag[0]) break;
    }
  }
  if (converge_check) {
    const int n_exec = itr > itr_max ? itr_max : itr;
    const size_t base = history.size();
    history.resize(base + n_exec);
    HIP_CHECK(hipMemcpy(history.data() + base, d_hist + 1, (size_t)n_exec * sizeof(double), hipMemcpyDeviceToHost));
    res = history.back();
  } else {
    if (sweep_failed(printMethod(s_type))) return 0;  // (synchronises)
  }
  return itr;
}

// The line solvers of the MAF flavour (cz_Poisson.cpp:549-557, 665-683, 770-776, 853-859, 940-946 call pcr_rb_maf_, pcr_rb_esa_maf_,
// pcr_maf_, pcr_eda_maf_, pcr_esa_maf_): coefficients from the metrics of the 1-D grids, pn-1 stages + 2x2 systems.
int CZ::LSOR_PCR_MAF(double& res, REAL_TYPE* X, REAL_TYPE* B, const int itr_max, double& flop, int s_type, bool converge_check) {
  const int gc = GUIDE;
  hipStream_t st = stream();
  reset_ticket();
  const int n = innerFidx[K_plus] - innerFidx[K_minus] + 1;
  const int pn = pcr_num_stage(n);
  if (pn < 0) {
    printf("error : number of stage\n");
    exit(0);
  }
  const bool rb = (s_type == LS_PCR_RB_MAF || s_type == LS_PCR_RB_ESA_MAF);
  const double fin = (s_type == LS_PCR_EDA_MAF || s_type == LS_PCR_ESA_MAF) ? 9.0 : 11.0;
  if (converge_check) {
    ensure_hist(itr_max + 2);
    HIP_CHECK(hipMemsetAsync(d_flag, 0, 2 * sizeof(int), st));
  }
  int itr;
  for (itr = 1; itr <= itr_max; itr++) {
    if (rb) {
      for (int color = 0; color < 2; color++) {  // global colouring, exchange per colour (as LSOR_PCR_RB)
        pcr_maf_async(X, MSK, B, size, innerFidx, gc, pn, 0, (color + head[0] + head[1]) & 1, d_xc, d_yc, d_zc, ac1, d_res, color);
        if (!Comm_S(X)) return 0;
      }
    } else {
      pcr_maf_async(X, MSK, B, size, innerFidx, gc, pn, 1, 0, d_xc, d_yc, d_zc, ac1, d_res, 0);
      if (!Comm_S(X)) return 0;
    }
    flop += (npts() / n) * ((24.0 + 6.0 + 12.0) + n * 21.0 + (n - 2.0) * 6.0 + n * (double)(pn - 1) * 16.0 + (double)(1 << (pn - 1)) * fin + n * 6.0);
    if (converge_check) {
      if (!Comm_SUM_dev(d_res, 1)) return 0;
      czhip_check_async(d_res, res_normal, eps, itr, d_hist, d_flag, d_flag + 1);
      HIP_CHECK(hipMemcpyAsync(h_flag, d_flag, sizeof(int), hipMemcpyDeviceToHost, st));
      if (sweep_failed(printMethod(s_type))) return 0;  // (synchronises)
      if (h_flag[0]) break;
    }
  }
  if (converge_check) {
    const int n_exec = itr > itr_max ? itr_max : itr;
    const size_t base = history.size();
    history.resize(base + n_exec);
    HIP_CHECK(hipMemcpy(history.data() + base, d_hist + 1, (size_t)n_exec * sizeof(double), hipMemcpyDeviceToHost));
    res = history.back();
  } else {
    if (sweep_failed(printMethod(s_type))) return 0;  // (synchronises)
  }
  return itr;
}

// cz_Poisson.cpp:518-611.  Line SOR: every (i,j) column of one checkerboard colour is solved along k by parallel cyclic
// reduction (pcr_rb_k), colour 0 then colour 1, in place.  Single-domain.
int CZ::LSOR_PCR_RB(double& res, REAL_TYPE* X, REAL_TYPE* B, const int itr_max, double& flop, int s_type, bool converge_check) {
  const int gc = GUIDE;
  hipStream_t st = stream();
  reset_ticket();
  const int n = innerFidx[K_plus] - innerFidx[K_minus] + 1;
  const int pn = pcr_num_stage(n);  // :535-538
  if (pn < 0) {
    printf("error : number of stage\n");
    exit(0);
  }
  if (converge_check) {
    ensure_hist(itr_max + 2);
    HIP_CHECK(hipMemsetAsync(d_flag, 0, 2 * sizeof(int), st));
  }
  int itr;
  for (itr = 1; itr <= itr_max; itr++) {
    for (int color = 0; color < 2; color++) {  // :573-578
      // mod(i+j, 2) == color is meant in GLOBAL indices: the brick's local rule is shifted by its head (the reference's
      // MPI path ignores this, cz_solver.f90:534 "unused variable"; here decomposed == single domain)
      pcr_rb_async(X, MSK, B, size, innerFidx, gc, pn, (color + head[0] + head[1]) & 1, ac1, d_res, color);
      if (!Comm_S(X)) return 0;  // after each colour, like the two-colour RB-SOR path
    }
    flop += npts() * (12.0 + (pn - 1) * 14.0);
    if (converge_check) {
      // the line solves are long launches: a host round trip per iteration is negligible here, so the reference's
      // sequential test (:584-601) is kept as is
      if (!Comm_SUM_dev(d_res, 1)) return 0;
      czhip_check_async(d_res, res_normal, eps, itr, d_hist, d_flag, d_flag + 1);
      HIP_CHECK(hipMemcpyAsync(h_flag, d_flag, sizeof(int), hipMemcpyDeviceToHost, st));
      if (sweep_failed(printMethod(s_type))) return 0;  // (synchronises)
      if (h_flag[0]) break;
    }
  }
  if (converge_check) {
    const int n_exec = itr > itr_max ? itr_max : itr;
    const size_t base = history.size();
    history.resize(base + n_exec);
    HIP_CHECK(hipMemcpy(history.data() + base, d_hist + 1, (size_t)n_exec * sizeof(double), hipMemcpyDeviceToHost));
    res = history.back();
  } else {
    if (sweep_failed(printMethod(s_type))) return 0;  // (synchronises)
  }
  return itr;
}

// cz_Poisson.cpp:239-270.  The reference reduces in REAL on every rank and all-reduces the REAL; here the double
// partial sums are all-reduced and rounded to REAL once.
REAL_TYPE CZ::Fdot1(REAL_TYPE* x, double& flop) {
  dot1_async(x, size, innerFidx, GUIDE, d_res + 1);
  flop += 2.0 * npts();
  if (!Comm_SUM_dev(d_res + 1, 1)) exit(0);
  HIP_CHECK(hipMemcpyAsync(h_scal + 1, d_res + 1, sizeof(double), hipMemcpyDeviceToHost, stream()));
  HIP_CHECK(hipStreamSynchronize(stream()));
  return (REAL_TYPE)h_scal[1];
}

REAL_TYPE CZ::Fdot2(REAL_TYPE* x, REAL_TYPE* y, double& flop) {
  dot2_async(x, y, size, innerFidx, GUIDE, d_res + 1);
  flop += 2.0 * npts();
  if (!Comm_SUM_dev(d_res + 1, 1)) exit(0);
  HIP_CHECK(hipMemcpyAsync(h_scal + 1, d_res + 1, sizeof(double), hipMemcpyDeviceToHost, stream()));
  HIP_CHECK(hipStreamSynchronize(stream()));
  return (REAL_TYPE)h_scal[1];
}

// The work vectors the preconditioner solves into are allocated zero-filled and afterwards only written on the inner box,
// so their guide cells and faces are zero for the whole run.
bool CZ::xx_shell_is_zero(const REAL_TYPE* xx) const { return xx == pcg_p_ || xx == pcg_s_; }

// cz_Poisson.cpp:273-322
// May PBiCGSTAB withhold `p = r + beta (p - omega q)` and `s = r - alpha q` and let the first pair of the preconditioner solve that follows
// make them (jacobi2p_k<BS>)?  Plain Jacobi or red-black SOR preconditioner on the whole-box fused pass with the literal zero start, one rank (a decomposed
// solve reads the right-hand side in its ghost layer, where the operands are not valid), and the launcher takes it.  CZ_BICG_FUSE=0: never.
bool CZ::bicg_fusable(int pc_type) {
  if (!cfg.on(CZV_BICG_FUSE, true)) return false;
  if ((pc_type != LS_JACOBI && pc_type != LS_SOR2SMA) || numProc != 1 || czhip_use_t2() == 0) return false;
  const bool rb = pc_type == LS_SOR2SMA;
  const PassPlan plan = plan_pass(pcg_p_, pcg_p, pc_type, 8, false, true, rb, true);
  if (!(plan.kind == PassPlan::WHOLE && plan.zero_start)) return false;
  int idx1[6];
  for (int f = 0; f < 6; f++) idx1[f] = innerFidx[f];
  return czhip_jacobi2_from_zero_made_async(pcg_p_, WRK, pcg_s, 2, pcg_r, pcg_q, pcg_p, (REAL_TYPE)0, (REAL_TYPE)0, size, innerFidx, idx1, GUIDE, cf, ac1,
                                            rb ? 0 : -1, d_res, 1) != 0;
}

void CZ::Preconditioner(REAL_TYPE* xx, REAL_TYPE* bb, double& flop, int s_type, const BMade* made) {
  double res = 0.0;
  const int lc_max = 8;  // :280
  const size_t nbytes = (size_t)(size[0] + 2 * GUIDE) * (size[1] + 2 * GUIDE) * (size[2] + 2 * GUIDE) * sizeof(REAL_TYPE);
  // blas_clear_(xx) of the caller (cz_Poisson.cpp:405, 441) is folded in here.  With the plain Jacobi preconditioner on
  // the fused-pair path the clear is not even executed: xx's guide cells / faces are zero from allocation on (sweeps only
  // ever write its inner box) and the first pair takes "u == 0" as a literal instead of reading it.
  // (single-domain only: a decomposed run leaves the neighbours' values in xx's ghost layers)
  const bool zero_start = (s_type == LS_JACOBI || s_type == LS_SOR2SMA) && numProc == 1 && czhip_use_t2() != 0 && xx_shell_is_zero(xx);
  if (!zero_start) HIP_CHECK(hipMemsetAsync(xx, 0, nbytes, stream()));
  struct Scope {
    bool& f;
    explicit Scope(bool& x) : f(x) { f = true; }
    ~Scope() { f = false; }
  } scope(in_precond);
  switch (s_type) {
    case LS_JACOBI:
    case LS_JACOBI_MAF:
      JACOBI(res, xx, bb, lc_max, flop, s_type, false, zero_start, made);
      break;
    case LS_SOR2SMA:
    case LS_SOR2SMA_MAF:
      RBSOR(res, xx, bb, lc_max, flop, s_type, false, zero_start, made);
      break;
    case LS_PCR_RB:
      LSOR_PCR_RB(res, xx, bb, lc_max, flop, s_type, false);
      break;
    case LS_PSOR:
    case LS_PSOR_MAF:
      PSOR(res, xx, bb, lc_max, flop, s_type, false);
      break;
    case LS_PCR:
    case LS_PCR_EDA:
    case LS_PCR_RB_ESA:  // (LS_PCR_J_ESA has no case in the reference either, cz_Poisson.cpp:282-321: it falls to the copy)
      LSOR_PCR_VARIANT(res, xx, bb, lc_max, flop, s_type, false);
      break;
    case LS_PCR_MAF:
    case LS_PCR_EDA_MAF:
    case LS_PCR_RB_MAF:
    case LS_PCR_RB_ESA_MAF:  // :300-316
      LSOR_PCR_MAF(res, xx, bb, lc_max, flop, s_type, false);
      break;
    default: {
      const size_t n = (size_t)(size[0] + 2 * GUIDE) * (size[1] + 2 * GUIDE) * (size[2] + 2 * GUIDE);
      HIP_CHECK(hipMemcpyAsync(xx, bb, n * sizeof(REAL_TYPE), hipMemcpyDeviceToDevice, stream()));  // blas_copy_
    }
  }
}

// cz_Poisson.cpp:332-504
int CZ::PBiCGSTAB(double& res, REAL_TYPE* X, REAL_TYPE* B, double& flop, int s_type) {
  const bool maf = (s_type == LS_BICGSTAB_MAF);
  const int gc = GUIDE;
  hipStream_t st = stream();
  const size_t nbytes = (size_t)(size[0] + 2 * gc) * (size[1] + 2 * gc) * (size[2] + 2 * gc) * sizeof(REAL_TYPE);
  int itr;
  double flop_count = 0.0;
  res = 0.0;

  HIP_CHECK(hipMemsetAsync(pcg_q, 0, nbytes, st));                       // :344 blas_clear_
  if (maf) calc_rk_maf_async(pcg_r, X, B, size, innerFidx, gc, d_xc, d_yc, d_zc, pvt);  // :352
  else calc_rk_async(pcg_r, X, B, size, innerFidx, gc, cf);                             // :356
  flop += (maf ? 63.0 : 14.0) * npts();
  if (!Comm_S(pcg_r)) return 0;                                         // :362
  HIP_CHECK(hipMemcpyAsync(pcg_r0, pcg_r, nbytes, hipMemcpyDeviceToDevice, st));  // :365 blas_copy_

  REAL_TYPE rho_old = 1.0, alpha = 0.0, omega = 1.0, r_omega = -omega;  // :368-371

  // Dot products that directly follow the kernel producing their operand are folded into that kernel (same per-point
  // REAL products, double accumulation, rounded once to REAL like Fdot1/Fdot2): q.r0 into q = A p_, t_.s and t_.t_ into
  // t_ = A s_, r.r and the NEXT iteration's rho = r.r0 into r = s - omega t_.
  MafPtrs mp{d_xc, d_yc, d_zc, pvt};
  auto fetch2 = [&](double* d_two, REAL_TYPE& a, REAL_TYPE& b2) -> bool {  // all-reduce + D2H of two device doubles
    if (!Comm_SUM_dev(d_two, 2)) return false;
    HIP_CHECK(hipMemcpyAsync(h_scal + 2, d_two, 2 * sizeof(double), hipMemcpyDeviceToHost, st));
    HIP_CHECK(hipStreamSynchronize(st));
    a = (REAL_TYPE)h_scal[2], b2 = (REAL_TYPE)h_scal[3];
    return true;
  };
  REAL_TYPE rho_next = 0.0;
  // The two vector updates that make the right-hand side of a preconditioner solve (:398, :434) are folded into the first pair of that solve
  // where it is the whole-box fused pass from a literal zero: one launch and one read of an array less per solve.
  const bool fuse = !maf && bicg_fusable(pc_type);
  bicg_fused = 0;
  // alpha / omega on the device between their dot products and their users (bicg_scalar_async): d_res[12..15] holds alpha, omega, -alpha, -omega
  // as REALs.  CZ_BICG_DEVSC=0: the host computes them from read-back dot products as in rounds 1-2 (and as the reference does).
  REAL_TYPE* const d_bs = reinterpret_cast<REAL_TYPE*>(d_res + 12);
  const bool devsc = cfg.on(CZV_BICG_DEVSC, true);
  bool pc_copy;  // Preconditioner() has no case for pc_type: it copies (cz_Poisson.cpp:282-321)
  switch (pc_type) {
    case LS_JACOBI: case LS_JACOBI_MAF: case LS_SOR2SMA: case LS_SOR2SMA_MAF: case LS_PCR_RB: case LS_PSOR: case LS_PSOR_MAF: case LS_PCR:
    case LS_PCR_EDA: case LS_PCR_RB_ESA: case LS_PCR_MAF: case LS_PCR_EDA_MAF: case LS_PCR_RB_MAF: case LS_PCR_RB_ESA_MAF:
      pc_copy = false;
      break;
    default:
      pc_copy = true;
  }
  if (!cfg.on(CZV_BICG_ALIAS, true)) pc_copy = false;  // (the copy is made: A/B and the bit-equality test)

  for (itr = 1; itr < ItrMax; itr++) {  // :373
    REAL_TYPE rho;
    if (itr == 1) {
      flop_count = 0.0;
      rho = Fdot2(pcg_r, pcg_r0, flop_count);  // :376
      flop += flop_count;
    } else {
      rho = rho_next;  // r.r0 was accumulated while r was written (below)
      flop += 2.0 * npts();
    }
    if (fabs(rho) < FLT_MIN) {  // :379-383
      itr = 0;
      break;
    }
    BMade made_p{0, nullptr, nullptr, nullptr, (REAL_TYPE)0, (REAL_TYPE)0, nullptr};
    if (itr == 1) {
      HIP_CHECK(hipMemcpyAsync(pcg_p, pcg_r, nbytes, hipMemcpyDeviceToDevice, st));  // :387
    } else {
      REAL_TYPE beta = rho / rho_old * alpha / omega;  // :394
      if (fuse) {
        // :398 withheld: the first pair of the solve below makes p = r + beta (p - omega q) on its way and writes it to the array of s
        // (dead until :434), which then IS p -- the pass cannot update p in place: neighbouring workgroups read each other's rows
        made_p = BMade{2, pcg_r, pcg_q, pcg_p, beta, omega, nullptr};
        std::swap(pcg_p, pcg_s);
        bicg_fused++;
      } else {
        bicg1_async(pcg_p, pcg_r, pcg_q, beta, omega, size, innerFidx, gc);  // :398
      }
      flop += 4.0 * npts();
    }
    if (!Comm_S(pcg_p)) return 0;                    // :402
    flop_count = 0.0;                                // :405 blas_clear_(pcg_p_) happens inside Preconditioner
    // "no preconditioner" is a copy in the reference (:318-320 blas_copy_ after :405 blas_clear_): p_ IS p then -- nothing writes p before
    // the last reader of p_ (:470) is through -- and three passes over an array per solve are not made (6.0 -> 4.7 ms per iteration at 512^3 FP64)
    REAL_TYPE* const p_ = pc_copy ? pcg_p : pcg_p_;
    if (!pc_copy) Preconditioner(pcg_p_, pcg_p, flop_count, pc_type, made_p.op ? &made_p : nullptr);  // :409
    flop += flop_count;
    if (line_error) return 0;

    // :417/:421 q = A p_  and  :427 q.r0
    calc_ax_dots_async(pcg_q, p_, pcg_r0, size, innerFidx, gc, cf, maf ? &mp : nullptr, d_res + 2);
    flop += (maf ? 63.0 : 13.0) * npts() + 2.0 * npts();
    REAL_TYPE r_alpha = (REAL_TYPE)0;
    if (devsc) {
      // alpha = rho / (q . r0) (:427) is made on the device and read there by the launches that need it; the host reads it back with omega and
      // the residual at the end of the iteration (one wait instead of three: 2 x 25-35 us of idle GPU per iteration, a fifth of an iteration at 128^3)
      if (!Comm_SUM_dev(d_res + 2, 2)) return 0;
      bicg_scalar_async(1, d_res + 2, rho, d_bs);
    } else {
      REAL_TYPE q_r0, q_q;
      if (!fetch2(d_res + 2, q_r0, q_q)) return 0;
      alpha = rho / q_r0;  // :427
      r_alpha = -alpha;
    }
    BMade made_s{1, pcg_q, pcg_r, nullptr, r_alpha, (REAL_TYPE)0, devsc ? d_bs + 2 : nullptr};
    if (fuse) bicg_fused++;  // :434 withheld likewise: s = r - alpha q
    else triad_async(pcg_s, pcg_q, pcg_r, r_alpha, size, innerFidx, gc, devsc ? d_bs + 2 : nullptr);  // :434
    flop += 2.0 * npts();
    if (!Comm_S(pcg_s)) return 0;  // :438

    flop_count = 0.0;  // :441 blas_clear_(pcg_s_) happens inside Preconditioner
    REAL_TYPE* const s_ = pc_copy ? pcg_s : pcg_s_;
    if (!pc_copy) Preconditioner(pcg_s_, pcg_s, flop_count, pc_type, fuse ? &made_s : nullptr);  // :445
    flop += flop_count;
    if (line_error) return 0;

    // :453/:457 t_ = A s_  and  :464 t_.s, t_.t_
    calc_ax_dots_async(pcg_t_, s_, pcg_s, size, innerFidx, gc, cf, maf ? &mp : nullptr, d_res + 4);
    flop += (maf ? 63.0 : 13.0) * npts() + 4.0 * npts();
    if (devsc) {
      if (!Comm_SUM_dev(d_res + 4, 2)) return 0;
      bicg_scalar_async(2, d_res + 4, (REAL_TYPE)0, d_bs);  // omega = (t . s) / (t . t)  :464
    } else {
      REAL_TYPE ts, tt;
      if (!fetch2(d_res + 4, ts, tt)) return 0;
      omega = ts / tt;  // :464
      r_omega = -omega;
    }

    bicg2_async(X, p_, s_, alpha, omega, size, innerFidx, gc, devsc ? d_bs : nullptr, devsc ? d_bs + 1 : nullptr);  // :470
    flop += 4.0 * npts();
    // :476 r = s - omega t_  with  :481 res = r.r  and the next :376 rho = r.r0
    triad_dots_async(pcg_r, pcg_t_, pcg_s, pcg_r0, r_omega, size, innerFidx, gc, d_res + 6, devsc ? d_bs + 3 : nullptr);
    flop += 2.0 * npts() + 2.0 * npts();
    REAL_TYPE rr;
    if (devsc) {  // the one wait of the iteration: r.r, r.r0 and the two scalars the next beta needs
      if (!Comm_SUM_dev(d_res + 6, 2)) return 0;
      HIP_CHECK(hipMemcpyAsync(h_scal + 2, d_res + 6, 2 * sizeof(double), hipMemcpyDeviceToHost, st));
      HIP_CHECK(hipMemcpyAsync(h_scal + 12, d_res + 12, 2 * sizeof(REAL_TYPE), hipMemcpyDeviceToHost, st));
      HIP_CHECK(hipStreamSynchronize(st));
      rr = (REAL_TYPE)h_scal[2], rho_next = (REAL_TYPE)h_scal[3];
      const REAL_TYPE* hs = reinterpret_cast<const REAL_TYPE*>(h_scal + 12);
      alpha = hs[0], omega = hs[1];
      r_omega = -omega;
    } else if (!fetch2(d_res + 6, rr, rho_next)) {
      return 0;
    }
    res = rr;

    if (!Comm_S(X)) return 0;  // :486
    // :488 all-reduces `res` a second time although Fdot1 already did (an MPI-only double count in the reference,
    // a no-op in its serial build); dropped here.
    res *= res_normal;  // :490
    res = sqrt(res);
    history.push_back(res);  // :492
    // :495 bc_k_(X): identity, X's faces are never written (see file header)
    if (res < eps) break;  // :498
    rho_old = rho;
  }
  return itr;
}

// ------------------------------------------------------------------------------------------------------------
void CZ::Field(REAL_TYPE* host) const {
  const size_t n = (size_t)(size[0] + 2 * GUIDE) * (size[1] + 2 * GUIDE) * (size[2] + 2 * GUIDE);
  czhip_d2h(host, P, n * sizeof(REAL_TYPE));
}

// profiling.txt (cz_Evaluate.cpp:506-545).  The reference prints PMlib's "Basic Report" (PMlib 6.4.x is a third-party
// library that is not part of the reference tree); this is the same table -- one line per measured section with call
// count, accumulated time, share, time per call, operation count and rate -- filled from the library's HIP-event timing
// of its launches (czhip_timing), under the reference's section labels (cz_miscel.cpp:177-262) where a launch maps to one.
void CZ::WriteProfile(FILE* fp) const {
  struct Sec {
    const char* label;   // reference label (or the nearest description)
    const char* lib;     // czhip_timing label
    double flop_per_call;
  };
  const double n = npts();
  const bool maf = SW_maf != 0;
  const int kn = innerFidx[K_plus] - innerFidx[K_minus] + 1;
  const int pn = pcr_num_stage(kn);
  const double pcr_flop = (n / kn) * (kn * 6.0 + kn * (pn - 1) * 14.0 + (double)(1 << (pn > 0 ? pn - 1 : 0)) * 9.0 + kn * 6.0 + 6.0) * 0.5;
  const Sec secs[] = {
      {maf ? "JACOBI_MAF_kernel" : "JACOBI_kernel", "jacobi", (maf ? 66.0 : 18.0) * n},
      {"JACOBI_kernel x2 (fused pair)", "jacobi2", 36.0 * n},
      {maf ? "SOR2SMA_MAF_kernel" : "SOR2SMA_kernel", "rbsor", (maf ? 33.0 : 9.0) * n},
      {"SOR2SMA_kernel x2 (both colours)", "rbsor2", 18.0 * n},
      {"Shell slabs of a fused pass", "pair_shell", 0.0},
      {"PCR_RB", "pcr_rb", pcr_flop},
      {maf ? "SOR_MAF_kernel" : "SOR_kernel", "psor", (maf ? 66.0 : 18.0) * n},
      {"Blas_AX", "calc_ax", (maf ? 63.0 : 13.0) * n},
      {"Blas_Residual", "calc_rk", (maf ? 63.0 : 14.0) * n},
      {"Dot1 / Dot2", "dot", 2.0 * n},
      {"Blas_TRIAD / BiCG_1 / BiCG_2", "ewise", 0.0},
      {"Residual reduction", "reduce", 0.0},
  };
  char host[256] = "unknown";
  gethostname(host, sizeof(host) - 1);
  time_t now = time(nullptr);
  char date[64];
  strftime(date, sizeof(date), "%Y/%m/%d : %H:%M:%S", localtime(&now));
  double tot_ms = 0.0;
  struct Row {
    const Sec* s;
    int calls;
    double ms;
  };
  std::vector<Row> rows;
  for (const Sec& s : secs) {
    double ms = 0.0;
    const int c = czhip_timing_read(s.lib, &ms);
    if (c == 0) continue;
    rows.push_back({&s, c, ms});
    tot_ms += ms;
  }
  std::sort(rows.begin(), rows.end(), [](const Row& a, const Row& b) { return a.ms > b.ms; });  // time cost order
  fprintf(fp, "\n# PMlib Basic Report -------------------------------------------------------\n\n");
  fprintf(fp, "\tTiming Statistics Report (PMlib layout; sections timed with HIP events on the GPU stream)\n");
  fprintf(fp, "\tHost name : %s\n\tDate      : %s\n\n\tCubeZ hot path on %s\n\n", host, date, czhip_arch());
  fprintf(fp, "\tParallel Mode:   %d process%s x 1 GPU\n\n", numProc, numProc > 1 ? "es" : "");
  fprintf(fp, "\tTotal execution time            = %e [sec]\n", solve_seconds);
  fprintf(fp, "\tTotal time of measured sections = %e [sec]\n\n", tot_ms * 1e-3);
  fprintf(fp, "\tExclusive sections statistics per process and total job.\n\n");
  fprintf(fp, "\tSection                          |  call  |        accumulated time[sec]           | [flop counts]\n");
  fprintf(fp, "\tLabel                            |        |      avr   avr[%%]     sdv    avr/call  |      avr       sdv   speed\n");
  fprintf(fp, "\t---------------------------------+--------+----------------------------------------+----------------------------\n");
  double tot_flop = 0.0;
  for (const Row& r : rows) {
    const double sec = r.ms * 1e-3, flop = r.s->flop_per_call * r.calls;
    tot_flop += flop;
    fprintf(fp, "\t%-33s: %8d   %9.3e %6.2f  %8.2e  %9.3e    %9.3e  %8.2e  %7.2f %s\n", r.s->label, r.calls, sec,
            tot_ms > 0 ? 100.0 * r.ms / tot_ms : 0.0, 0.0, sec / r.calls, flop, 0.0, sec > 0 ? flop / sec * 1e-12 : 0.0, "Tflops");
  }
  fprintf(fp, "\t---------------------------------+--------+----------------------------------------+----------------------------\n");
  fprintf(fp, "\t%-33s  %8s   %9.3e %37s %9.3e  %8s  %7.2f %s\n", "Sections per process", "", tot_ms * 1e-3, "", tot_flop, "",
          tot_ms > 0 ? tot_flop / (tot_ms * 1e-3) * 1e-12 : 0.0, "Tflops");
  fprintf(fp, "\t---------------------------------+--------+----------------------------------------+----------------------------\n");
  fprintf(fp, "\t%-33s  %8s   %9.3e %37s %9.3e  %8s  %7.2f %s\n\n", "Sections total job", "", tot_ms * 1e-3, "", tot_flop * numProc, "",
          tot_ms > 0 ? tot_flop * numProc / (tot_ms * 1e-3) * 1e-12 : 0.0, "Tflops");
  fprintf(fp, "\tInclusive section: %s  = %e [sec] (host wall clock around the solver loop)\n", printMethod(ls_type), solve_seconds);
}

// fileout_t (cz_utility.f90:17-47, the -D_aurora_=1 body): a Fortran sequential unformatted file -- every record framed
// by its byte length (4-byte integer) -- holding (1,1) | (ix,jx,kx) | org | (dh,dh,dh) | (0, 0.0) | s(1:kx,1:ix,1:jx) with
// i fastest, then j, then k.  `s` is the padded host copy of a field.
bool CZ::WriteSph(const char* fname, const REAL_TYPE* s) const {
  FILE* fp = fopen(fname, "wb");
  if (!fp) {
    printf("\tSorry, can't open '%s' file. Write failed.\n", fname);
    return false;
  }
  const int g = GUIDE, ix = size[0], jx = size[1], kx = size[2];
  const size_t nk = kx + 2 * g, ni = ix + 2 * g;
  auto rec = [&](const void* a, size_t na, const void* b = nullptr, size_t nb = 0) {
    const int32_t len = (int32_t)(na + nb);
    fwrite(&len, 4, 1, fp);
    fwrite(a, 1, na, fp);
    if (b) fwrite(b, 1, nb, fp);
    fwrite(&len, 4, 1, fp);
  };
  const int32_t one[2] = {1, 1}, dims[3] = {ix, jx, kx}, nn = 0;
  const REAL_TYPE dh3[3] = {pitch[0], pitch[0], pitch[0]}, rtime = 0;
  rec(one, sizeof(one));
  rec(dims, sizeof(dims));
  rec(origin, 3 * sizeof(REAL_TYPE));
  rec(dh3, sizeof(dh3));
  rec(&nn, 4, &rtime, sizeof(REAL_TYPE));
  const size_t nbytes = (size_t)ix * jx * kx * sizeof(REAL_TYPE);
  if (nbytes > 0x7ffffff7u) {
    printf("\t'%s': the field record exceeds the 2 GiB a 4-byte record length can frame. Write failed.\n", fname);
    fclose(fp);
    return false;
  }
  const int32_t len = (int32_t)nbytes;
  fwrite(&len, 4, 1, fp);
  std::vector<REAL_TYPE> slab((size_t)ix * jx);
  for (int k = 1; k <= kx; k++) {
    for (int j = 1; j <= jx; j++)
      for (int i = 1; i <= ix; i++)
        slab[(size_t)(j - 1) * ix + (i - 1)] = s[(size_t)(k + g - 1) + (size_t)(i + g - 1) * nk + (size_t)(j + g - 1) * nk * ni];
    fwrite(slab.data(), sizeof(REAL_TYPE), slab.size(), fp);
  }
  fwrite(&len, 4, 1, fp);
  const bool ok = !ferror(fp);
  fclose(fp);
  return ok;
}

// exact_t (cz_utility.f90:52-82) on the host: the analytic solution on the whole brick (1..ix, 1..jx, 1..kx), padded layout
void CZ::Exact(std::vector<REAL_TYPE>& e) const {
  const int g = GUIDE;
  const size_t nk = size[2] + 2 * g, ni = size[0] + 2 * g, nj = size[1] + 2 * g;
  e.assign(nk * ni * nj, (REAL_TYPE)0);
  volatile REAL_TYPE one = 1.0, two = 2.0;
#ifdef CZ_REAL_IS_DOUBLE
  const REAL_TYPE r2 = sqrt(two), pi = 2.0 * asin(one);
#else
  const REAL_TYPE r2 = sqrtf(two), pi = 2.0f * asinf(one);
#endif
  const REAL_TYPE dh = pitch[0];
  for (int j = 1; j <= size[1]; j++)
    for (int i = 1; i <= size[0]; i++)
      for (int k = 1; k <= size[2]; k++) {
        const REAL_TYPE x = G_origin[0] + dh * (REAL_TYPE)(head[0] - 1 + i - 1);
        const REAL_TYPE y = G_origin[1] + dh * (REAL_TYPE)(head[1] - 1 + j - 1);
        const REAL_TYPE z = G_origin[2] + dh * (REAL_TYPE)(head[2] - 1 + k - 1);
#ifdef CZ_REAL_IS_DOUBLE
        e[(size_t)(k + g - 1) + (size_t)(i + g - 1) * nk + (size_t)(j + g - 1) * nk * ni] =
            sin(pi * x) * sin(pi * y) / sinh(r2 * pi) * (sinh(r2 * pi * z) - sinh(r2 * pi * (z - (REAL_TYPE)1.0)));
#else
        e[(size_t)(k + g - 1) + (size_t)(i + g - 1) * nk + (size_t)(j + g - 1) * nk * ni] =
            sinf(pi * x) * sinf(pi * y) / sinhf(r2 * pi) * (sinhf(r2 * pi * z) - sinhf(r2 * pi * (z - (REAL_TYPE)1.0)));
#endif
      }
}

// Debug epilogue (cz_Evaluate.cpp:550-563): exact_t_ + err_t_ of cz_utility.f90:52-129 restated on the host
// (out of the hot path; the field comes back over PCIe once).
double CZ::ErrorMax(int loc[3]) {
  const int g = GUIDE;
  const size_t nk = size[2] + 2 * g, ni = size[0] + 2 * g, nj = size[1] + 2 * g;
  std::vector<REAL_TYPE> p(nk * ni * nj);
  Field(p.data());
  volatile REAL_TYPE one = 1.0, two = 2.0;
#ifdef CZ_REAL_IS_DOUBLE
  const REAL_TYPE r2 = sqrt(two), pi = 2.0 * asin(one);
#define CZ_SIN sin
#define CZ_SINH sinh
#else
  const REAL_TYPE r2 = sqrtf(two), pi = 2.0f * asinf(one);
#define CZ_SIN sinf
#define CZ_SINH sinhf
#endif
  const REAL_TYPE dh = pitch[0];
  double d = 0.0;
  loc[0] = loc[1] = loc[2] = -1;
  for (int j = innerFidx[J_minus]; j <= innerFidx[J_plus]; j++)
    for (int i = innerFidx[I_minus]; i <= innerFidx[I_plus]; i++)
      for (int k = innerFidx[K_minus]; k <= innerFidx[K_plus]; k++) {
        const REAL_TYPE x = G_origin[0] + dh * (REAL_TYPE)(head[0] - 1 + i - 1);
        const REAL_TYPE y = G_origin[1] + dh * (REAL_TYPE)(head[1] - 1 + j - 1);
        const REAL_TYPE z = G_origin[2] + dh * (REAL_TYPE)(head[2] - 1 + k - 1);
        const REAL_TYPE e = CZ_SIN(pi * x) * CZ_SIN(pi * y) / CZ_SINH(r2 * pi) *
                            (CZ_SINH(r2 * pi * z) - CZ_SINH(r2 * pi * (z - (REAL_TYPE)1.0)));  // cz_utility.f90:75
        const REAL_TYPE r = p[(size_t)(k + g - 1) + (size_t)(i + g - 1) * nk + (size_t)(j + g - 1) * nk * ni] - e;
        const double qq = fabs((double)r);
        if (d < qq) {
          d = qq;
          loc[0] = i, loc[1] = j, loc[2] = k;
        }
      }
  if (numProc > 1) {
    // Comm_MAX_1 (cz_Evaluate.cpp:558): the maximum over ranks; location stays the local one as in the reference
    d = comm_allreduce_max_host(comm, d);
  }
  return d;
}

// ============================================================================================================
// Part 4 of include/cz_hip.h
// ============================================================================================================
struct cz_handle {
  CZ cz;
};

extern "C" {
cz_handle* cz_create(void) { return new cz_handle(); }
void cz_destroy(cz_handle* h) { delete h; }
int cz_evaluate(cz_handle* h, int argc, char** argv) { return h->cz.Evaluate(argc, argv); }
int cz_setup(cz_handle* h, int argc, char** argv) { return h->cz.Setup(argc, argv); }
int cz_solve(cz_handle* h) { return h->cz.Solve(); }
int cz_sweeps(cz_handle* h, int n) { return h->cz.Sweeps(n); }
int cz_result_iter(const cz_handle* h) { return h->cz.result_itr; }
double cz_result_res(const cz_handle* h) { return h->cz.result_res; }
int cz_history(const cz_handle* h, double* out, int cap) {
  const int n = (int)h->cz.history.size();
  for (int i = 0; i < n && i < cap; i++) out[i] = h->cz.history[i];
  return n;
}
void cz_field(const cz_handle* h, CZ_REAL* host_out) { h->cz.Field(host_out); }
void cz_local_size(const cz_handle* h, int* size3, int* head3, int* nID6, int* inner6) {
  for (int a = 0; a < 3; a++) size3[a] = h->cz.size[a], head3[a] = h->cz.head[a];
  for (int f = 0; f < 6; f++) nID6[f] = h->cz.nID[f], inner6[f] = h->cz.innerFidx[f];
}
double cz_error_max(cz_handle* h, int* loc3) { return h->cz.ErrorMax(loc3); }
void cz_set_quiet(cz_handle* h, int q) { h->cz.quiet = q != 0; }
void cz_set_debug(cz_handle* h, int m) { h->cz.debug_mode = m; }
void cz_set_profile(cz_handle* h, int on) { h->cz.profile = on != 0; }
double cz_last_solve_seconds(const cz_handle* h) { return h->cz.solve_seconds; }
int cz_info(const cz_handle* h, int what) {
  const CZ& c = h->cz;
  switch (what) {
    case 0: return c.numProc;
    case 1: return c.pairs_ok ? 1 : 0;
    case 2: return c.n_shell;
    case 3: return c.overlap;
    case 4: return c.last_lag;
    case 10: return c.bicg_fused;
    case 11: return c.rb4_passes;
    case 5: return comm_transport_ranks(c.comm);
    case 6: return c.comm_cus;
    case 7: return c.last_plan.kind;
    case 8: return c.last_plan.depth;
    case 9: return c.last_plan.buffers;
    default: return -1;
  }
}
double cz_kernel_ms(const cz_handle* h, const char* label) {
  (void)h;
  double tot = 0.0;
  const int n = czhip_timing_read(label, &tot);
  return n > 0 ? tot / n : 0.0;
}
}  // extern "C"
