// cz_driver.h -- the restated host driver: class CZ of the reference (src/cz_cpp/cz.h:84-181, DomainInfo.h:30-49)
// with every 3-D array resident in HBM and the solver loops (cz_Poisson.cpp) rebuilt around asynchronous launches.
#ifndef CZ_DRIVER_H_
#define CZ_DRIVER_H_

#include <cstdio>
#include <string>
#include <vector>

#include "cz_config.h"
#include "cz_internal.h"

typedef CZ_REAL REAL_TYPE;  // cz_Define.h:28-37
#define GUIDE 2             // cz_Define.h:40

// cz_Define.h:68-89 (only the solvers of the hot path are accepted; the others are rejected by setLS)
enum LinearSolver { LS_NONE = 0, LS_PSOR = 1, LS_SOR2SMA, LS_BICGSTAB, LS_JACOBI, LS_PCR = 5, LS_PCR_EDA, LS_PCR_ESA, LS_PCR_RB, LS_PCR_RB_ESA, LS_PCR_J_ESA, LS_PSOR_MAF = 11, LS_SOR2SMA_MAF, LS_BICGSTAB_MAF, LS_JACOBI_MAF, LS_PCR_MAF, LS_PCR_EDA_MAF, LS_PCR_ESA_MAF, LS_PCR_RB_MAF, LS_PCR_RB_ESA_MAF };

// CB_Define_stub.h:64-70 / cz_fparam.fi:10-16
enum { I_minus = 0, I_plus, J_minus, J_plus, K_minus, K_plus };

struct CommCtx;  // cz_comm.cpp (RCCL halo exchange / all-reduce); nullptr when numProc == 1

class CZ {
 public:
  // ---- DomainInfo (DomainInfo.h:30-49)
  int myRank = 0, numProc = 1;
  int nID[6] = {-1, -1, -1, -1, -1, -1};
  int head[3] = {1, 1, 1};
  int G_div[3] = {1, 1, 1};
  REAL_TYPE pitch[3] = {0, 0, 0};
  int size[3] = {0, 0, 0};
  REAL_TYPE origin[3] = {0, 0, 0};
  int G_size[3] = {0, 0, 0};
  REAL_TYPE G_origin[3] = {0, 0, 0};
  int innerFidx[6] = {0, 0, 0, 0, 0, 0};

  // ---- CZ (cz.h:84-133, 154-181)
  int debug_mode = 0;
  int ItrMax = 0;
  int ls_type = LS_NONE, pc_type = LS_NONE;
  double eps = 1.0e-5;
  REAL_TYPE ac1 = 0;
  double res_normal = 0.0;
  REAL_TYPE cf[7] = {1, 1, 1, 1, 1, 1, 6};
  std::string precon;
  FILE* fph = nullptr;
  std::string hist_name;

  REAL_TYPE *WRK = nullptr, *P = nullptr, *RHS = nullptr;
  // MAF flavour (cz.h:113-118, cz_Evaluate.cpp:245-253, 342-369): 1-D grid on the device, pivot array
  int SW_maf = 0;
  REAL_TYPE* MSK = nullptr;  // cz.h:95, cz_Evaluate.cpp:389 (line-SOR solvers)
  REAL_TYPE *d_xc = nullptr, *d_yc = nullptr, *d_zc = nullptr, *pvt = nullptr;
  REAL_TYPE *pcg_p = nullptr, *pcg_p_ = nullptr, *pcg_r = nullptr, *pcg_r0 = nullptr, *pcg_q = nullptr, *pcg_s = nullptr,
            *pcg_s_ = nullptr, *pcg_t_ = nullptr;

  // ---- build-specific state
  bool quiet = false;
  bool profile = false;          // write profiling.txt after the solve (the cz command line turns it on; CZ_PROFILE=0 off)
  bool set_up = false;
  int result_itr = 0;
  double result_res = 0.0;
  std::vector<double> history;   // residual of iteration 1..n (index 0 = iteration 1)
  double solve_seconds = 0.0;
  int sweeps_done = 0;           // stationary-solver iterations executed so far (bench leg)
  CommCtx* comm = nullptr;
  // decomposed runs: the exchange of a fused pair runs on comm_stream while the interior is being swept
  hipStream_t comm_stream = nullptr;
  hipEvent_t ev_shell = nullptr, ev_src = nullptr, ev_comm = nullptr, ev_int = nullptr, ev_chk[2] = {nullptr, nullptr};
  bool pairs_ok = true;          // decomposed runs: EVERY brick can run the fused pass (agreed at set-up; the exchange pattern depends on it)
  int rb4_passes = 0;            // two-iteration red-black passes (rb4_k) of the last RBSOR solve (cz_info 11)
  int bicg_fused = 0;            // vector updates of the last BiCGSTAB solve that were made inside the first pair of a preconditioner solve (cz_info 10)
  bool in_precond = false;       // inside Preconditioner: an unchecked solve does not drain the queue (the caller's next launch follows in stream order)
  int last_lag = 0;              // the last stationary solve ran its all-reduce + test one pass behind (cz_info)
  CzConfig cfg;  // the environment as read when this object was created (cz_config.h)
  int skew_rank = -1, skew_ms = 0;  // CZ_TEST_SKEW=rank,ms: that rank sleeps before each look at the convergence flag (tests)
  void skew_wait() const;
  int lag_reduce = 1;            // CZ_LAG_REDUCE=0: residual all-reduce + test on the compute stream after every pass (no lag)
  REAL_TYPE* WRK2 = nullptr;     // third rotation buffer of the lagged mode
  int wrk_shell_tag = 0;         // whose shell WRK carries: 0 unknown, 1 P's (Dirichlet faces), 2 all zero (sync_wrk_shell)
  void sync_wrk_shell(const REAL_TYPE* X);
  int overlap = 1;               // CZ_OVERLAP=0 turns it off (exchange after the whole sweep, same results)
  // how the sweeps of the current / last stationary solve are executed (CZ::plan_pass; cz_info 7-9)
  struct PassPlan {
    enum Kind { SINGLE = 0, WHOLE = 1, SPLIT = 2 };
    int kind = SINGLE;  // SINGLE: one sweep / colour per launch; WHOLE: fused pass over the whole inner box; SPLIT: shell slabs + interior, exchange overlapped
    int depth = 1;      // ghost layers exchanged per pass
    int lag = 0;        // residual all-reduce + test one pass behind, on the exchange stream
    int buffers = 2;    // rotating field buffers
    int zero_start = 0; // the first pass takes the start vector as a literal zero
    int maf = 0, rb = 0;
    int comm_cus = 0;
  };
  PassPlan last_plan;
  bool plan_printed = false;
  int comm_cus = 0;              // CUs per XCD the sweeps leave to the exchange stream (CZ_COMM_CUS; 0 in single-domain runs)
  int n_shell = 0;               // shell boxes (cells within two layers of a rank-internal face), 1-based index ranges
  int shell_boxes[36];
  int interior[6], interior1[6]; // the rest of the inner box / its first-sweep range

  // device-side convergence bookkeeping (cz_Poisson.cpp:67-77 moved to the GPU)
  double* d_res = nullptr;       // [0] sum dp^2 of the current iteration, [1..7] dot products
  double* d_hist = nullptr;      // residual history, index = iteration
  int hist_cap = 0;
  int* d_flag = nullptr;         // [0] converged flag, [1] iteration at which it was set
  double* h_scal = nullptr;      // pinned: mirrors of d_res
  int* h_flag = nullptr;         // pinned

  CZ();
  ~CZ();

  int Evaluate(int argc, char** argv);  // cz_Evaluate.cpp:21-567
  int Setup(int argc, char** argv);     //   :21-391
  int Solve();                          //   :397-496
  int Sweeps(int n);
  double ErrorMax(int loc[3]);          //   :550-563
  void Field(REAL_TYPE* host) const;
  void WriteProfile(FILE* fp) const;                                             // cz_Evaluate.cpp:506-545
  bool WriteSph(const char* fname, const REAL_TYPE* padded_host_field) const;  // cz_utility.f90:17-47
  void Exact(std::vector<REAL_TYPE>& e) const;                                   // cz_utility.f90:52-82

 private:
  void setLS(const char* q);                                   // cz_Evaluate.cpp:684-803
  void setStrPre();                                            // :571-681
  double range_inner_index();                                  // cz_miscel.cpp:20-52
  bool decompose(int div_type);                                // replaces CBrick SubDomain (cz_Evaluate.cpp:103-159)
  void ensure_hist(int n);

  // cz_Poisson.cpp
  // the right-hand side of a preconditioner solve as the vector update that makes it (PBiCGSTAB): op 1: B = a*x + y, op 2: B = x + a*(z - b*y);
  // the first pair of the solve then makes B on its way instead of reading it (czhip_jacobi2_from_zero_made_async)
  struct BMade {
    int op;
    const REAL_TYPE* x;
    const REAL_TYPE* y;
    const REAL_TYPE* z;
    REAL_TYPE a, b;
    const REAL_TYPE* a_dev;  // where set: a lives on the device (bicg_scalar_async)
  };
  int JACOBI(double& res, REAL_TYPE* X, REAL_TYPE* B, int itr_max, double& flop, int s_type, bool converge_check = true,
             bool x_is_zero = false, const BMade* made = nullptr);
  bool bicg_fusable(int pc_type);
  bool xx_shell_is_zero(const REAL_TYPE* xx) const;
  int RBSOR(double& res, REAL_TYPE* X, REAL_TYPE* B, int itr_max, double& flop, int s_type, bool converge_check = true, bool x_is_zero = false,
            const BMade* made = nullptr);
  int PSOR(double& res, REAL_TYPE* X, REAL_TYPE* B, int itr_max, double& flop, int s_type, bool converge_check = true);
  int LSOR_PCR_VARIANT(double& res, REAL_TYPE* X, REAL_TYPE* B, int itr_max, double& flop, int s_type, bool converge_check = true);
  int LSOR_PCR_MAF(double& res, REAL_TYPE* X, REAL_TYPE* B, int itr_max, double& flop, int s_type, bool converge_check = true);
  int LSOR_PCR_RB(double& res, REAL_TYPE* X, REAL_TYPE* B, int itr_max, double& flop, int s_type, bool converge_check = true);
  REAL_TYPE Fdot1(REAL_TYPE* x, double& flop);
  REAL_TYPE Fdot2(REAL_TYPE* x, REAL_TYPE* y, double& flop);
  void Preconditioner(REAL_TYPE* xx, REAL_TYPE* bb, double& flop, int s_type, const BMade* made = nullptr);
  int PBiCGSTAB(double& res, REAL_TYPE* X, REAL_TYPE* B, double& flop, int s_type);

  // cz_comm.cpp replacements (no-ops when numProc == 1, like cz_comm.cpp:25,76,104)
  bool Comm_S(REAL_TYPE* X, const int* skip_flag = nullptr);
  bool Comm_S2(REAL_TYPE* X, const int* skip_flag = nullptr);  // two layers + edges (fused Jacobi pairs)
  bool Comm_SUM_dev(double* d_val, int count, const int* skip_flag = nullptr);
  void plan_overlap();
  bool pair_overlapped(REAL_TYPE* src, REAL_TYPE* dst, REAL_TYPE* B, const int* idx1, int rb, const int* skip, double* res_slot = nullptr,
                       const czhip_internal::MafPtrs* maf = nullptr);
  PassPlan plan_pass(REAL_TYPE* X, REAL_TYPE* B, int s_type, int itr_max, bool converge_check, bool x_is_zero, bool rb, bool probe_only = false);
  bool Comm_SUM_1(double* host_val);

  int finish_stationary(int itr_max, int first_itr, bool converge_check, double& res);
  // line solvers: the one-launch lexicographic sweep reports a lost hand-off between its workgroups as a NaN residual (every wait inside
  // it is bounded); the iterate is then void and the solve ends with "Solver error" instead of sweeping on (true = failed; message printed)
  bool sweep_failed(const char* solver);
  bool line_error = false;       // set by sweep_failed; PBiCGSTAB gives up when a line-solver preconditioner set it
  double npts() const;
};

#endif
