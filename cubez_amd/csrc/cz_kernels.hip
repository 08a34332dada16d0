// cz_kernels.hip -- hand-written gfx950 (CDNA4) kernels of the CubeZ hot path + the C-ABI of include/cz_hip.h
// parts 1-3.  One translation unit per precision (-DCZ_REAL_IS_DOUBLE for FP64), compiled with
// -ffp-contract=off so that every sweep reproduces the reference's un-fused FP32/FP64 arithmetic bit for bit
// (SURVEY.md section 7 "hard parts").
//
// Data layout (cz_solver.f90:29): arrays are dense (NK+2g, NI+2g, NJ+2g), K fastest.  A (k,i) PLANE of one j is
// therefore one contiguous run of (NK+2g)*(NI+2g) elements: row i+1 follows row i directly.  The stencil kernel
// exploits this by treating a plane as a 1-D array of 16-byte vectors (float4 / double2):
//     k+-1 neighbour = +-1 element, i+-1 neighbour = +-R vectors (R = (NK+2g)/V), j+-1 neighbour = +-1 plane.
// A workgroup owns a contiguous SEGMENT of S = TB*M vectors of the plane (about S/R rows) and marches it through
// a CHUNK of TJ consecutive planes (2.5-D blocking): plane j-1/j/j+1 values of its own vectors live in registers
// (a 3-deep queue that rotates as j advances), the centre plane j of the segment plus one row of halo on either
// side is staged in LDS (double-buffered, one barrier per plane) for the i+-1 and the vector-crossing k+-1
// neighbours.  Every global access is a 16-byte-per-lane fully coalesced load/store; each p and b element is read
// from HBM once per sweep (plus R-vector row halos shared with the neighbouring workgroup through L2 and two
// planes per chunk), each new element is written once.  No MFMA: 18 flop per 12 bytes is far below the ridge.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "cz_config.h"
#include "cz_hip.h"
#include "cz_internal.h"

typedef CZ_REAL REAL;
#ifdef CZ_REAL_IS_DOUBLE
constexpr int VW = 2;  // elements per 16-byte vector
#else
constexpr int VW = 4;
#endif

namespace {
#include "cz_k_common.h"
#include "cz_k_fastdiv.h"
#include "cz_k_stencil.h"
#include "cz_k_pair.h"
#include "cz_k_pair2.h"
#include "cz_k_rb4.h"
#include "cz_k_linesor.h"
#include "cz_k_psor.h"
#include "cz_k_blas.h"
#include "cz_h_ctx.h"
#include "cz_h_launch.h"

}  // namespace

namespace {
struct CheckArgs {
  int enabled = 0, itr = 0;
  double res_normal = 0.0, eps = 0.0;
  double* hist = nullptr;
  int* flag = nullptr;
  int* conv_itr = nullptr;
};
template <int MODE>
void sweep_async(const REAL* p_in, REAL* p_out, const REAL* b, const Box& bx, const Coef& cf, int par, double* res_dev,
                 int accumulate, const int* skip, const CheckArgs& ck, const MafArgs* ma = nullptr) {
  int nblk = 0;
  if (ctx.tune.fuse_fin) {
    Fin fin;
    fin.dst = res_dev, fin.accumulate = accumulate, fin.counter = ctx.counter;
    fin.do_check = ck.enabled, fin.itr = ck.itr, fin.res_normal = ck.res_normal, fin.eps = ck.eps;
    fin.hist = ck.hist, fin.flag = ck.flag, fin.conv_itr = ck.conv_itr;
    if (ma) launch_stencil_maf<MODE>(p_in, b, p_out, cf.omg, bx, par, skip, &nblk, fin, *ma);
    else launch_stencil<MODE>(p_in, b, p_out, cf, bx, par, skip, &nblk, fin);
  } else {
    if (ma) launch_stencil_maf<MODE>(p_in, b, p_out, cf.omg, bx, par, skip, &nblk, Fin(), *ma);
    else launch_stencil<MODE>(p_in, b, p_out, cf, bx, par, skip, &nblk);
    reduce_partials(nblk, res_dev, accumulate, skip);
    if (ck.enabled) czhip_check_async(res_dev, ck.res_normal, ck.eps, ck.itr, ck.hist, ck.flag, ck.conv_itr);
  }
}
}  // namespace

static inline int rb_parity(int g, const int* idx, int ofst, int color) {
  // cz_solver.f90:466  k = kst + mod(i+j+kp,2), step 2  <=>  (k + i + j + kst + kp) even  (1-based)
  // padded 0-based indices shift each of k,i,j by g-1
  return (3 * (g - 1) + idx[4] + ofst + color) & 1;
}

namespace {
// device copies of the reference's host-resident 1-D coordinate arrays (cz_Evaluate.cpp:342-363 fills them on the host)
MafArgs upload_xyz(const int* sz, int g, const REAL* X, const REAL* Y, const REAL* Z, const REAL* pvt) {
  const size_t nx = sz[0] + 2 * g, ny = sz[1] + 2 * g, nz = sz[2] + 2 * g, tot = nx + ny + nz;
  if (tot > ctx.xyz_cap) {
    if (ctx.xyz) {
      HIP_CHECK(hipStreamSynchronize(ctx.stream));
      HIP_CHECK(hipFree(ctx.xyz));
    }
    HIP_CHECK(hipMalloc(&ctx.xyz, tot * sizeof(REAL)));
    ctx.xyz_cap = tot;
  }
  HIP_CHECK(hipStreamSynchronize(ctx.stream));  // the previous call may still read the old copy
  HIP_CHECK(hipMemcpy(ctx.xyz, X, nx * sizeof(REAL), hipMemcpyHostToDevice));
  HIP_CHECK(hipMemcpy(ctx.xyz + nx, Y, ny * sizeof(REAL), hipMemcpyHostToDevice));
  HIP_CHECK(hipMemcpy(ctx.xyz + nx + ny, Z, nz * sizeof(REAL), hipMemcpyHostToDevice));
  MafArgs ma;
  ma.xc = ctx.xyz, ma.yc = ctx.xyz + nx, ma.zc = ctx.xyz + nx + ny, ma.pvt = pvt;
  return ma;
}

void launch_pivot(REAL* pvt, const Box& b, const MafArgs& ma) {
  if (b.empty) return;
  const int nplanes = b.jj1 - b.jj0 + 1;
  if (rows_ok(b, {pvt})) {
    EGeom e = make_egeom<VW>(b);
    dim3 grid((unsigned)((e.Fend - e.F0 + 255) / 256), (unsigned)nplanes);
    hipLaunchKernelGGL((pivot_k<VW>), grid, dim3(256), 0, ctx.stream, pvt, e, ma, b.nkp, b.nip);
  } else {
    EGeom e = make_egeom<1>(b);
    dim3 grid((unsigned)((e.Fend - e.F0 + 255) / 256), (unsigned)nplanes);
    hipLaunchKernelGGL((pivot_k<1>), grid, dim3(256), 0, ctx.stream, pvt, e, ma, b.nkp, b.nip);
  }
  HIP_CHECK(hipGetLastError());
}
}  // namespace

namespace {
#include "cz_h_linesor.h"
}  // namespace

// ============================================================================================================
// Part 2: runtime
// ============================================================================================================
extern "C" {

int czhip_real_bytes(void) { return (int)sizeof(REAL); }
// every environment variable the library reads, with the value in force now (cz_config.h); the string lives until the next call on this thread
const char* czhip_config_describe(int only_set) {
  static thread_local std::string text;
  text = CzConfig::from_env().describe(only_set != 0);
  return text.c_str();
}
const char* czhip_arch(void) { return "gfx950"; }

int czhip_init(int device) {
  if (ctx.ready) return 0;
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev == 0) {
    cz_fatal(1, "czhip: no HIP device available (%s) -- this library has no CPU fallback\n", hipGetErrorString(e));
  }
  const CzConfig cfg = CzConfig::from_env();  // read once per context (cz_config.h)
  if (device < 0) device = cfg.num(CZV_LOCAL_RANK, 0) % ndev;
  HIP_CHECK(hipSetDevice(device));
  ctx.device = device;
  hipDeviceProp_t prop;
  HIP_CHECK(hipGetDeviceProperties(&prop, device));
  ctx.num_cu = prop.multiProcessorCount;
  ctx.cu_reserved = 0;
  HIP_CHECK(hipStreamCreateWithFlags(&ctx.stream, hipStreamNonBlocking));
  HIP_CHECK(hipMalloc(&ctx.scal_dev, 16 * sizeof(double)));
  HIP_CHECK(hipMemset(ctx.scal_dev, 0, 16 * sizeof(double)));
  HIP_CHECK(hipHostMalloc(&ctx.scal_host, 16 * sizeof(double), hipHostMallocDefault));
  HIP_CHECK(hipMalloc(&ctx.counter, 64));
  HIP_CHECK(hipMemset(ctx.counter, 0, 64));
  ctx.ready = true;
  ensure_partials(65536);
  HIP_CHECK(hipMalloc(&ctx.shell_partials, (size_t)2 * 2048 * 6 * sizeof(double)));
  ctx.tune.fuse_fin = cfg.num(CZV_FUSE_FIN, ctx.tune.fuse_fin);
  if (const char* t2 = cfg.str(CZV_T2)) {  // "enable[,threads,mv,tj]"
    int en = 1, a = 0, b2 = 0, c2 = -1;
    const int n = sscanf(t2, "%d,%d,%d,%d", &en, &a, &b2, &c2);
    if (n >= 1) czhip_set_tuning2(n >= 2 ? a : 0, n >= 3 ? b2 : 0, n >= 4 ? c2 : -1, en);
  }
  ctx.tune.t2_map = cfg.num(CZV_T2_MAP, ctx.tune.t2_map);
  if (const char* pc = cfg.str(CZV_PCR)) {  // "fast[,variant]"
    int f = 1, v = 0;
    sscanf(pc, "%d,%d", &f, &v);
    ctx.tune.pcr_fast = f, ctx.tune.pcr_variant = v;
  }
  ctx.tune.t2_any_rows = cfg.on(CZV_T2_ROWS, ctx.tune.t2_any_rows != 0) ? 1 : 0;
  ctx.tune.t2_kwin = cfg.num(CZV_T2_KWIN, ctx.tune.t2_kwin);
  ctx.tune.t2_pre = cfg.num(CZV_T2_PRE, ctx.tune.t2_pre);
  ctx.tune.unit_coef = cfg.num(CZV_UNIT_COEF, ctx.tune.unit_coef);
  if (const char* v = cfg.str(CZV_RB4)) {  // "enable[,vectors per window[,planes per chunk]]"
    int en = 1, kw = 0, tj = 0;
    sscanf(v, "%d,%d,%d", &en, &kw, &tj);
    ctx.tune.rb4 = en, ctx.tune.rb4_kwin = kw, ctx.tune.rb4_tj = tj;
  }
  if (const char* pp = cfg.str(CZV_PCR_PIPE)) {  // "form[,seconds[,groups[,rows per thread]]]": form as Tuning::pcr_pipe; bound of the waits inside the kernel
    int w = 1, rows = 0, q = 1;
    double sec = 2.0;
    sscanf(pp, "%d,%lf,%d,%d", &w, &sec, &rows, &q);
    ctx.tune.pcr_pipe = w, ctx.tune.pipe_spin_ticks = (long long)(sec * 1e8), ctx.tune.pcr_rows = rows, ctx.tune.pcr_q = q;
  }
  if (const char* v = cfg.str(CZV_PSOR)) {  // "one_launch[,workgroups per CU]"
    int one = 1, wg = 0;
    sscanf(v, "%d,%d", &one, &wg);
    ctx.tune.psor_col = one, ctx.tune.psor_wg_per_cu = wg;
  }
  ctx.tune.pcr_wg_per_cu = cfg.num(CZV_PCR_WG_PER_CU, ctx.tune.pcr_wg_per_cu);
  ctx.tune.pcr_max_wg = cfg.num(CZV_PCR_MAX_WG, ctx.tune.pcr_max_wg);
  ctx.tune.pcr_slots = cfg.num(CZV_PCR_SLOTS, ctx.tune.pcr_slots);
  if (const char* pf = cfg.str(CZV_PCR_PIPE_PROF)) ctx.pipe_prof_file = pf;
  if (const char* tu = cfg.str(CZV_TUNING)) {  // "threads,m,tj,pf"
    int a = 0, b = 0, c = 0, d = -1;
    if (sscanf(tu, "%d,%d,%d,%d", &a, &b, &c, &d) >= 2) czhip_set_tuning(a, b, c < 0 ? 0 : c, d);
  }
  return 0;
}

void czhip_finalize(void) {
  if (!ctx.ready) return;
  HIP_CHECK(hipStreamSynchronize(ctx.stream));
  for (auto& kv : ctx.bc_tabs) (void)hipFree(kv.second);
  ctx.bc_tabs.clear();
  for (auto& kv : ctx.pair_maps) (void)hipFree(kv.second.dev);
  ctx.pair_maps.clear();
  (void)hipFree(ctx.partials);
  (void)hipFree(ctx.shell_partials);
  if (ctx.pcr_tab) (void)hipFree(ctx.pcr_tab);
  if (ctx.pcr_tab_perm) (void)hipFree(ctx.pcr_tab_perm);
  if (ctx.pcr_scratch) (void)hipFree(ctx.pcr_scratch);
  if (ctx.psor_faces) (void)hipFree(ctx.psor_faces);
  if (ctx.psor_order) (void)hipFree(ctx.psor_order);
  if (ctx.psor_ctl) (void)hipFree(ctx.psor_ctl);
  if (ctx.pipe_ctl) (void)hipFree(ctx.pipe_ctl);
  ctx.pipe_ctl = nullptr, ctx.pipe_ctl_cap = 0;
  if (ctx.pipe_hb) (void)hipFree(ctx.pipe_hb);
  ctx.pipe_hb = nullptr, ctx.pipe_hb_cap = 0, ctx.pipe_seq = 0;
  (void)hipFree(ctx.scal_dev);
  (void)hipHostFree(ctx.scal_host);
  if (ctx.counter) (void)hipFree(ctx.counter);
  if (ctx.xyz) (void)hipFree(ctx.xyz);
  for (auto* list : {&ctx.ev_used, &ctx.ev_free})
    for (auto& e : *list) {
      (void)hipEventDestroy(e.a);
      (void)hipEventDestroy(e.b);
    }
  (void)hipStreamDestroy(ctx.stream);
  ctx = Ctx();
}

CZ_REAL* czhip_alloc_s3d(const int* sz) {
  ensure_init();
  const size_t n = (size_t)(sz[0] + 4) * (size_t)(sz[1] + 4) * (size_t)(sz[2] + 4);  // GUIDE = 2, cz_Define.h:40
  REAL* p = nullptr;
  HIP_CHECK(hipMalloc(&p, n * sizeof(REAL)));
  HIP_CHECK(hipMemsetAsync(p, 0, n * sizeof(REAL), ctx.stream));
  HIP_CHECK(hipStreamSynchronize(ctx.stream));
  return p;
}

void czhip_free(void* d) {
  if (!d) return;
  ensure_init();
  HIP_CHECK(hipStreamSynchronize(ctx.stream));
  HIP_CHECK(hipFree(d));
}

void czhip_h2d(void* dst, const void* src, size_t bytes) {
  ensure_init();
  HIP_CHECK(hipStreamSynchronize(ctx.stream));
  HIP_CHECK(hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice));
}

void czhip_d2h(void* dst, const void* src, size_t bytes) {
  ensure_init();
  HIP_CHECK(hipStreamSynchronize(ctx.stream));
  HIP_CHECK(hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost));
}

void czhip_sync(void) {
  ensure_init();
  HIP_CHECK(hipStreamSynchronize(ctx.stream));
}

void* czhip_stream(void) {
  ensure_init();
  return (void*)ctx.stream;
}

int czhip_set_tuning(int threads, int m, int tj, int pf) {
  Tuning t = ctx.tune;
  if (threads > 0) t.threads = threads;
  if (m > 0) t.m = m;
  if (tj >= 0) t.tj = tj;
  if (pf >= 0) t.pf = pf;
  const bool ok = (t.threads == 256 || t.threads == 512 || t.threads == 1024) && (t.m == 1 || t.m == 2 || (t.m == 4 && t.threads != 1024)) &&
                  (t.pf == 0 || t.pf == 1);
  if (!ok) return 1;
  ctx.tune = t;
  return 0;
}

void czhip_get_tuning(int* threads, int* m, int* tj, int* pf) {
  *threads = ctx.tune.threads, *m = ctx.tune.m, *tj = ctx.tune.tj, *pf = ctx.tune.pf;
}

void czhip_timing(int enable) {
  ensure_init();
  HIP_CHECK(hipStreamSynchronize(ctx.stream));
  for (auto& e : ctx.ev_used) ctx.ev_free.push_back(e);
  ctx.ev_used.clear();
  for (int l = 0; l < 16; l++) ctx.t_acc[l] = 0.0, ctx.t_cnt[l] = 0;
  ctx.timing = enable != 0;
}

int czhip_timing_read(const char* label, double* total_ms) {
  ensure_init();
  HIP_CHECK(hipStreamSynchronize(ctx.stream));
  int want = -1;
  for (int l = 0; l < LBL_COUNT; l++)
    if (!strcmp(kLabelNames[l], label)) want = l;
  if (want < 0) {
    if (total_ms) *total_ms = 0.0;
    return 0;
  }
  double tot = ctx.t_acc[want];
  int n = (int)ctx.t_cnt[want];
  for (auto& e : ctx.ev_used) {
    if (e.label != want) continue;
    float ms = 0.f;
    HIP_CHECK(hipEventElapsedTime(&ms, e.a, e.b));
    tot += ms;
    n++;
  }
  if (total_ms) *total_ms = tot;
  return n;
}

// ============================================================================================================
// Part 3: asynchronous operations
// ============================================================================================================
void czhip_jacobi_async(const CZ_REAL* p_in, CZ_REAL* p_out, const CZ_REAL* b, const int* sz, const int* idx, int g,
                        const CZ_REAL* cf, CZ_REAL omg, double* res_dev, int accumulate, const int* skip) {
  ensure_init();
  const Box bx = make_box(sz, idx, g);
  if (bx.empty) {
    if (!accumulate) HIP_CHECK(hipMemsetAsync(res_dev, 0, sizeof(double), ctx.stream));
    return;
  }
  sweep_async<MODE_JACOBI>(p_in, p_out, b, bx, make_coef(cf, omg), 0, res_dev, accumulate, skip, CheckArgs());
}

void czhip_jacobi_checked_async(const CZ_REAL* p_in, CZ_REAL* p_out, const CZ_REAL* b, const int* sz, const int* idx, int g,
                                const CZ_REAL* cf, CZ_REAL omg, double* res_dev, double res_normal, double eps, int itr,
                                double* hist_dev, int* flag_dev, int* conv_itr_dev) {
  ensure_init();
  const Box bx = make_box(sz, idx, g);
  CheckArgs ck;
  ck.enabled = 1, ck.itr = itr, ck.res_normal = res_normal, ck.eps = eps, ck.hist = hist_dev, ck.flag = flag_dev, ck.conv_itr = conv_itr_dev;
  if (bx.empty) {
    HIP_CHECK(hipMemsetAsync(res_dev, 0, sizeof(double), ctx.stream));
    czhip_check_async(res_dev, res_normal, eps, itr, hist_dev, flag_dev, conv_itr_dev);
    return;
  }
  sweep_async<MODE_JACOBI>(p_in, p_out, b, bx, make_coef(cf, omg), 0, res_dev, 0, flag_dev, ck);
}

void czhip_rbsor_async(CZ_REAL* p, const CZ_REAL* b, const int* sz, const int* idx, int g, const CZ_REAL* cf, int ofst,
                       int color, CZ_REAL omg, double* res_dev, int accumulate, const int* skip) {
  ensure_init();
  const Box bx = make_box(sz, idx, g);
  if (bx.empty) {
    if (!accumulate) HIP_CHECK(hipMemsetAsync(res_dev, 0, sizeof(double), ctx.stream));
    return;
  }
  sweep_async<MODE_RB>(p, p, b, bx, make_coef(cf, omg), rb_parity(g, idx, ofst, color), res_dev, accumulate, skip, CheckArgs());
}

void czhip_rbsor_checked_async(CZ_REAL* p, const CZ_REAL* b, const int* sz, const int* idx, int g, const CZ_REAL* cf, int ofst,
                               int color, CZ_REAL omg, double* res_dev, int accumulate, double res_normal, double eps, int itr,
                               double* hist_dev, int* flag_dev, int* conv_itr_dev) {
  ensure_init();
  const Box bx = make_box(sz, idx, g);
  CheckArgs ck;
  ck.enabled = 1, ck.itr = itr, ck.res_normal = res_normal, ck.eps = eps, ck.hist = hist_dev, ck.flag = flag_dev, ck.conv_itr = conv_itr_dev;
  if (bx.empty) {
    if (!accumulate) HIP_CHECK(hipMemsetAsync(res_dev, 0, sizeof(double), ctx.stream));
    czhip_check_async(res_dev, res_normal, eps, itr, hist_dev, flag_dev, conv_itr_dev);
    return;
  }
  sweep_async<MODE_RB>(p, p, b, bx, make_coef(cf, omg), rb_parity(g, idx, ofst, color), res_dev, accumulate, flag_dev, ck);
}

// Two fused Jacobi sweeps u -> w (time n -> n+2).  res_dev[0], res_dev[1] receive sum dp^2 of sweep n+1 / n+2.
// With check arguments (hist_dev != NULL) the last workgroup also performs the bookkeeping of cz_Poisson.cpp:67-77 for
// iterations itr and itr+1 in order; a converged first sweep leaves the flag set with conv_itr = itr (the caller then
// recomputes that single sweep from u, which this kernel never modifies).  Returns 1 if launched, 0 if the geometry
// is not supported (caller falls back to two single sweeps).
int czhip_jacobi2_async(const CZ_REAL* u, CZ_REAL* w, const CZ_REAL* b, const int* sz, const int* idx, const int* idx1, int g,
                        const CZ_REAL* cf, CZ_REAL omg, double* res_dev, double res_normal, double eps, int itr, double* hist_dev,
                        int* flag_dev, int* conv_itr_dev, const int* skip_flag_dev) {
  ensure_init();
  if (!ctx.tune.fuse_fin) return 0;
  const Box bx = make_box(sz, idx, g);
  if (bx.empty || g < 2) return 0;
  const Box ba = idx1 ? make_box(sz, idx1, g) : bx;
  Fin2 fin;
  fin.dst = res_dev;
  if (hist_dev) {
    fin.do_check = 1, fin.itr = itr, fin.res_normal = res_normal, fin.eps = eps;
    fin.hist = hist_dev, fin.flag = flag_dev, fin.conv_itr = conv_itr_dev;
  }
  return launch_jacobi2<0>(u, b, w, make_coef(cf, omg), bx, ba, hist_dev ? flag_dev : skip_flag_dev, fin) ? 1 : 0;
}

// The first pair of sweeps of a preconditioner solve, whose start vector was just cleared (cz_Poisson.cpp:405-409: blas_clear_
// then 8 sweeps): the input field is identically zero, so it is neither cleared in memory nor read -- the arithmetic is
// the same with literal zeros.  `u_shape` is only used for its alignment/geometry checks.  No convergence bookkeeping.
int czhip_jacobi2_from_zero_async(const CZ_REAL* u_shape, CZ_REAL* w, const CZ_REAL* b, const int* sz, const int* idx, const int* idx1,
                                  int g, const CZ_REAL* cf, CZ_REAL omg, double* res_dev) {
  ensure_init();
  if (!ctx.tune.fuse_fin) return 0;
  const Box bx = make_box(sz, idx, g);
  if (bx.empty || g < 2) return 0;
  const Box ba = idx1 ? make_box(sz, idx1, g) : bx;
  Fin2 fin;
  fin.dst = res_dev;
  return launch_jacobi2<0>(u_shape, b, w, make_coef(cf, omg), bx, ba, nullptr, fin, 0, 1) ? 1 : 0;
}

// The same with the right-hand side of the solve MADE on the way: op 1: b = a*x + y (blas_triad_: s = r - alpha q), op 2: b = x + a*(z - bb*y)
// (blas_bicg_1_: p = r + beta (p - omega q)), written to `b_out` by the pass (jacobi2p_k<BS>) -- the vector update before a preconditioner
// solve of BiCGSTAB (cz_Poisson.cpp:398, 434) and the first pass of that solve in one launch; b_out must not be one of x, y, z.  op 0: b_out
// is read as the right-hand side (the plain start from zero).  rb_ofst < 0: two Jacobi sweeps; >= 0: one red-black iteration with that offset.
int czhip_jacobi2_from_zero_made_async(const CZ_REAL* u_shape, CZ_REAL* w, CZ_REAL* b_out, int op, const CZ_REAL* x, const CZ_REAL* y,
                                       const CZ_REAL* z, CZ_REAL a, CZ_REAL bb, const int* sz, const int* idx, const int* idx1, int g,
                                       const CZ_REAL* cf, CZ_REAL omg, int rb_ofst, double* res_dev, int probe) {
  return czhip_internal::pass_from_zero_made(u_shape, w, b_out, op, x, y, z, a, nullptr, bb, sz, idx, idx1, g, cf, omg, rb_ofst, res_dev, probe);
}
}  // extern "C"
namespace czhip_internal {
// (a_dev: the coefficient a read from the device instead, see BSrc::pa)
int pass_from_zero_made(const CZ_REAL* u_shape, CZ_REAL* w, CZ_REAL* b_out, int op, const CZ_REAL* x, const CZ_REAL* y, const CZ_REAL* z, CZ_REAL a,
                        const CZ_REAL* a_dev, CZ_REAL bb, const int* sz, const int* idx, const int* idx1, int g, const CZ_REAL* cf, CZ_REAL omg,
                        int rb_ofst, double* res_dev, int probe) {
  ensure_init();
  if (!ctx.tune.fuse_fin || op < 0 || op > 2) return 0;
  if (op != 0 && (b_out == x || b_out == y || (op == 2 && b_out == z))) return 0;
  const Box bx = make_box(sz, idx, g);
  if (bx.empty || g < 2) return 0;
  const Box ba = idx1 ? make_box(sz, idx1, g) : bx;
  Fin2 fin;
  fin.dst = res_dev;
  BSrc bs;
  if (op != 0) bs.x = x, bs.y = y, bs.z = (op == 2) ? z : x, bs.out = b_out, bs.a = a, bs.b = bb, bs.pa = a_dev;
  if (rb_ofst >= 0) {
    fin.single = 1;
    return launch_jacobi2<1>(u_shape, b_out, w, make_coef(cf, omg), bx, ba, nullptr, fin, rb_parity(g, idx, rb_ofst, 0), 1, probe != 0, nullptr, op ? &bs : nullptr, op) ? 1 : 0;
  }
  return launch_jacobi2<0>(u_shape, b_out, w, make_coef(cf, omg), bx, ba, nullptr, fin, 0, 1, probe != 0, nullptr, op ? &bs : nullptr, op) ? 1 : 0;
}
}  // namespace czhip_internal
extern "C" {

// One complete red-black SOR iteration (colour 0 then colour 1, cz_Poisson.cpp:205-209) in one pass over memory, u -> w.
// res_dev[0] receives the iteration's sum dp^2 (both colours).  Same conventions as czhip_jacobi2_async.
int czhip_rbsor2_async(const CZ_REAL* u, CZ_REAL* w, const CZ_REAL* b, const int* sz, const int* idx, const int* idx1, int g,
                       const CZ_REAL* cf, int ofst, CZ_REAL omg, double* res_dev, double res_normal, double eps, int itr,
                       double* hist_dev, int* flag_dev, int* conv_itr_dev, const int* skip_flag_dev) {
  ensure_init();
  if (!ctx.tune.fuse_fin) return 0;
  const Box bx = make_box(sz, idx, g);
  if (bx.empty || g < 2) return 0;
  const Box ba = idx1 ? make_box(sz, idx1, g) : bx;
  Fin2 fin;
  fin.dst = res_dev;
  fin.single = 1;
  if (hist_dev) {
    fin.do_check = 1, fin.itr = itr, fin.res_normal = res_normal, fin.eps = eps;
    fin.hist = hist_dev, fin.flag = flag_dev, fin.conv_itr = conv_itr_dev;
  }
  return launch_jacobi2<1>(u, b, w, make_coef(cf, omg), bx, ba, hist_dev ? flag_dev : skip_flag_dev, fin, rb_parity(g, idx, ofst, 0)) ? 1 : 0;
}

// TWO red-black SOR iterations (four colour sweeps) in one pass over memory, u -> w (rb4_k; single-domain boxes).  res_dev[0], res_dev[1] receive
// the sums dp^2 of iteration itr and itr + 1; with hist_dev the last workgroup does the bookkeeping of both (a converged FIRST iteration leaves
// the flag set with conv_itr = itr: the caller recomputes that one iteration from u, which this kernel never modifies).  probe != 0: only
// answer whether the launcher takes the geometry.  Returns 1 if launched (or launchable), 0 otherwise.
int czhip_rbsor4_async(const CZ_REAL* u, CZ_REAL* w, const CZ_REAL* b, const int* sz, const int* idx, int g, const CZ_REAL* cf, int ofst, CZ_REAL omg,
                       double* res_dev, double res_normal, double eps, int itr, double* hist_dev, int* flag_dev, int* conv_itr_dev,
                       const int* skip_flag_dev, int probe) {
  ensure_init();
  const Box bx = make_box(sz, idx, g);
  if (bx.empty || g < 2) return 0;
  Fin2 fin;
  fin.dst = res_dev;
  if (hist_dev) {
    fin.do_check = 1, fin.itr = itr, fin.res_normal = res_normal, fin.eps = eps;
    fin.hist = hist_dev, fin.flag = flag_dev, fin.conv_itr = conv_itr_dev;
  }
  return launch_rb4(u, b, w, make_coef(cf, omg), bx, hist_dev ? flag_dev : skip_flag_dev, fin, rb_parity(g, idx, ofst, 0), probe != 0) ? 1 : 0;
}

// rb4_k switches (measurements): enable 0 | 1, vectors per k window, planes per chunk (0: the launcher's rule); negative: keep.
int czhip_set_rb4(int enable, int window, int planes) {
  ensure_init();
  if (enable >= 0) ctx.tune.rb4 = enable;  // (2: also on the small grids where the preloaded one-iteration pass is faster -- tests)
  if (window >= 0) ctx.tune.rb4_kwin = window;
  if (planes >= 0) ctx.tune.rb4_tj = planes;
  return 0;
}

// The fused pass split the way a decomposed brick runs it: the slabs behind the faces with nID[f] >= 0 first
// (pair_shell_k), then the interior (jacobi2_k) -- same result as the unsplit launch.  rb_ofst < 0: two Jacobi sweeps,
// res_dev[0..1]; rb_ofst >= 0: one red-black iteration with that ofst, res_dev[0].  Returns 0 when nothing was launched.
int czhip_pair_split_async(const CZ_REAL* u, CZ_REAL* w, const CZ_REAL* b, const int* sz, const int* idx, const int* idx1,
                           const int* nID, int g, const CZ_REAL* cf, CZ_REAL omg, int rb_ofst, double* res_dev) {
  ensure_init();
  int boxes[36], in0[6], in1[6];
  const int n = czhip_internal::pair_plan(idx, nID, boxes, in0, in1);
  if (n == 0 || !czhip_internal::pair_probe(u, w, b, sz, in0, in1, g, cf[6], 0)) return 0;
  const int rb = rb_ofst >= 0 ? rb_parity(g, idx, rb_ofst, 0) : -1;
  czhip_internal::pair_shell_async(u, w, b, sz, idx1 ? idx1 : idx, boxes, n, g, cf, omg, rb, nullptr, nullptr, nullptr);
  return czhip_internal::pair_box_async(u, w, b, sz, in0, in1, g, cf, omg, rb, res_dev, 1, nullptr, nullptr);
}

// MAF flavour of czhip_jacobi2_async / czhip_rbsor2_async (rb_ofst < 0: two jacobi_maf sweeps, res_dev[0..1]; rb_ofst >= 0: one red-black
// iteration with that ofst, res_dev[0]).  X, Y, Z are HOST coordinate arrays like those of the drop-in *_maf_ symbols.
int czhip_pair_maf_async(const CZ_REAL* u, CZ_REAL* w, const CZ_REAL* b, const int* sz, const int* idx, int g, const CZ_REAL* X,
                         const CZ_REAL* Y, const CZ_REAL* Z, CZ_REAL omg, int rb_ofst, double* res_dev) {
  ensure_init();
  const MafArgs ma = upload_xyz(sz, g, X, Y, Z, nullptr);
  return czhip_internal::pair_maf_async(u, w, b, sz, idx, idx, g, ma.xc, ma.yc, ma.zc, omg, rb_ofst, res_dev, 0.0, 0.0, 0, nullptr, nullptr,
                                        nullptr, nullptr);
}

int czhip_set_tuning2(int threads, int vec_per_thread, int planes_per_chunk, int enable) {
  Tuning t = ctx.tune;
  if (threads > 0) t.t2_threads = threads;
  if (threads == -2) t.t2_threads = 0;  // chosen per launch
  if (vec_per_thread > 0) t.t2_mv = vec_per_thread;
  if (planes_per_chunk >= 0) t.t2_tj = planes_per_chunk;  // 0: chosen per launch
  if (enable >= 0) t.use_t2 = enable;
  if ((t.t2_threads != 0 && t.t2_threads != 512 && t.t2_threads != 1024) || t.t2_mv != 2) return 1;
  ctx.tune = t;
  return 0;
}

int czhip_use_t2(void) { return ctx.tune.use_t2; }

// the preloaded form of the pass on small grids (jacobi2p_k<PRE>): 1 = where every workgroup is resident at once (default), 0 = never; negative:
// keep.  Returns the setting that was in force.  Same bits either way.
int czhip_set_pair_preload(int enable) {
  ensure_init();
  const int before = ctx.tune.t2_pre;
  if (enable >= 0) ctx.tune.t2_pre = enable;  // (TB * 10 + PRE: that form only -- measurements)
  return before;
}

// the kernels' form for unit coefficients (offdiag_sum<UNIT>: c1 .. c6 all exactly 1, as CZ sets them): 1 = taken where the coefficients allow
// (default), 0 = never; negative: keep.  Returns the setting that was in force.  Same bits either way.
int czhip_set_unit_coef(int enable) {
  ensure_init();
  const int before = ctx.tune.unit_coef;
  if (enable >= 0) ctx.tune.unit_coef = enable;
  return before;
}

// k windows of the two-stage pass (Geom2, cz_k_pair.h): vectors per window; 0 = whole rows wherever they fit, -1 = the launcher's rule
// (pair_whole_rows_ok), <= -2 = keep.  Returns the setting that was in force.  Same bits whatever the windows.
int czhip_set_pair_window(int vectors) {
  ensure_init();
  const int before = ctx.tune.t2_kwin;
  if (vectors >= -1) ctx.tune.t2_kwin = vectors;
  return before;
}

// Self-test of the hoisted division of the two-stage pass (cz_k_fastdiv.h): number of numerators (of 2^32: every float / a structured
// sample of doubles) whose quotient by d differs in any bit from the ordinary IEEE division; -1 when the launchers would not use the
// hoisted form for this divisor at all.
long long czhip_selftest_fastdiv(CZ_REAL d) {
  ensure_init();
  if (!fastdiv_ok(d)) return -1;
  unsigned long long* bad = nullptr;
  HIP_CHECK(hipMalloc(&bad, sizeof(*bad)));
  HIP_CHECK(hipMemsetAsync(bad, 0, sizeof(*bad), ctx.stream));
  hipLaunchKernelGGL(fastdiv_check_k<0>, dim3(4096), dim3(256), 0, ctx.stream, d, bad);
  HIP_CHECK(hipGetLastError());
  unsigned long long h = 0;
  HIP_CHECK(hipMemcpyAsync(&h, bad, sizeof(h), hipMemcpyDeviceToHost, ctx.stream));
  HIP_CHECK(hipStreamSynchronize(ctx.stream));
  HIP_CHECK(hipFree(bad));
  return (long long)h;
}

// Measurement aid (tools/cu_reserve_cost.py): put the CU reservation of decomposed runs in force on this context by hand; returns what is in force.
int czhip_set_comm_cus(int k) {
  ensure_init();
  return czhip_internal::reserve_comm_cus(k);
}

// line-SOR kernel choice: form 0 = pcr_rb_k (the reference's arithmetic literally, pcr_rb only), 1 = table + d in LDS, 2 = table +
// d in registers (default); variant = waves per workgroup * 10 + lines per wave, 0 = measured default.  Negative: keep.
int czhip_set_pcr_mode(int form, int variant) {
  ensure_init();
  if (form > 2) return 1;
  if (form >= 0) ctx.tune.pcr_fast = form;
  if (variant >= 0) ctx.tune.pcr_variant = variant;
  return 0;
}

// the lexicographic line SOR (pcr_, pcr_esa_, pcr_eda_ and the solvers of those names): one_launch 1 = pcr_lex_wg_k, 0 = a launch per
// diagonal; groups of threads per workgroup (0 = the launcher's choice); rows per thread (1 | 2).  Negative: keep.  Same bits in every shape.
int czhip_set_pcr_lex(int one_launch, int groups, int rows_per_thread) {
  ensure_init();
  if (rows_per_thread > 2 || rows_per_thread == 0) return 1;
  if (one_launch >= 0) ctx.tune.pcr_pipe = one_launch ? 1 : 0;
  if (groups >= 0) ctx.tune.pcr_rows = groups;
  if (rows_per_thread > 0) ctx.tune.pcr_q = rows_per_thread;
  return 0;
}

// bound of every wait inside pcr_lex_wg_k, in seconds (default 2; negative: keep).  Returns the bound in force.  When a wait runs out the
// sweep's residual is NaN: a lost hand-off must not pass for a result.  (0 makes every wait longer than a few hundred polls give up: test aid.)
double czhip_set_pcr_lex_timeout(double seconds) {
  ensure_init();
  if (seconds >= 0.0) ctx.tune.pipe_spin_ticks = (long long)(seconds * 1e8);
  return (double)ctx.tune.pipe_spin_ticks * 1e-8;
}

// psor_ / psor_maf_ and the solvers of those names: one_launch 1 = the whole sweep in one launch (psor_col_k: columns of workgroups walking k,
// faces handed on through memory), 0 = a launch per tile hyperplane (psor_tile_k); workgroups per CU of the former (0 = four).  Negative: keep.
// Same bits either way.
int czhip_set_psor(int one_launch, int wg_per_cu) {
  ensure_init();
  if (one_launch >= 0) ctx.tune.psor_col = one_launch ? 1 : 0;
  if (wg_per_cu >= 0) ctx.tune.psor_wg_per_cu = wg_per_cu;
  return 0;
}

// psor_col_k: how many steps ahead of their use the face words of the columns before are asked for (4 | 8; 0 = the launcher's rule: 8 for FP32
// boxes of more than 300 points along k, else 4; profiles/r04/psor_what_bounds_it.txt).  Returns the previous setting.  Same bits either way.
int czhip_set_psor_ahead(int steps) {
  ensure_init();
  const int before = ctx.tune.psor_ahead;
  if (steps == 0 || steps == 4 || steps == 8) ctx.tune.psor_ahead = steps;
  return before;
}

// pcr_lex_wg_k launch limits (test aid; negative: keep, 0: the launcher's choice): workgroups per CU, workgroups in all, lines per hand-off
// ring (rounded up to a power of two).  The launcher's own choice of the ring always lets the sweep finish; a forced small ring with few
// workgroups does not -- the waits inside the kernel then run out and the sweep reports a NaN residual (tests/test_gpu_kernels.py).
int czhip_set_pcr_lex_limits(int wg_per_cu, int max_wg, int slots) {
  ensure_init();
  if (wg_per_cu >= 0) ctx.tune.pcr_wg_per_cu = wg_per_cu;
  if (max_wg >= 0) ctx.tune.pcr_max_wg = max_wg;
  if (slots >= 0) ctx.tune.pcr_slots = slots;
  return 0;
}

void czhip_check2_async(const double* res_dev, double res_normal, double eps, int itr, double* hist_dev, int* flag_dev,
                        int* conv_itr_dev) {
  ensure_init();
  hipLaunchKernelGGL(check2_k, dim3(1), dim3(1), 0, ctx.stream, res_dev, res_normal, eps, itr, hist_dev, flag_dev, conv_itr_dev, (int*)nullptr);
  HIP_CHECK(hipGetLastError());
}

void czhip_check_async(const double* res_dev, double res_normal, double eps, int itr, double* hist_dev, int* flag_dev,
                       int* conv_itr_dev) {
  ensure_init();
  hipLaunchKernelGGL(check_k, dim3(1), dim3(1), 0, ctx.stream, res_dev, res_normal, eps, itr, hist_dev, flag_dev, conv_itr_dev, (int*)nullptr);
  HIP_CHECK(hipGetLastError());
}

// ============================================================================================================
// Part 1: drop-in kernels (synchronous, reference semantics)
// ============================================================================================================
void bc_k_(int* sz, int* gp, CZ_REAL* p, CZ_REAL* dh, CZ_REAL* org, int* nID) {
  ensure_init();
  const int g = *gp, ix = sz[0], jx = sz[1], kx = sz[2];
  const int nkp = kx + 2 * g, nip = ix + 2 * g;
  // nID order I-,I+,J-,J+,K-,K+ (cz_fparam.fi:10-16)
  if (nID[4] < 0 || nID[5] < 0) {
    const REAL* tab = bc_table(ix, jx, *dh, org);
    dim3 grid((ix + 127) / 128, jx);
    if (nID[4] < 0) hipLaunchKernelGGL(bc_kface_k, grid, dim3(128), 0, ctx.stream, p, tab, ix, jx, 1, g, nkp, nip);
    if (nID[5] < 0) hipLaunchKernelGGL(bc_kface_k, grid, dim3(128), 0, ctx.stream, p, tab, ix, jx, kx, g, nkp, nip);
  }
  {
    dim3 grid((kx + 127) / 128, jx);
    if (nID[0] < 0) hipLaunchKernelGGL(bc_iface_k, grid, dim3(128), 0, ctx.stream, p, jx, kx, 1, g, nkp, nip);
    if (nID[1] < 0) hipLaunchKernelGGL(bc_iface_k, grid, dim3(128), 0, ctx.stream, p, jx, kx, ix, g, nkp, nip);
  }
  {
    dim3 grid((kx + 127) / 128, ix);
    if (nID[2] < 0) hipLaunchKernelGGL(bc_jface_k, grid, dim3(128), 0, ctx.stream, p, ix, kx, 1, g, nkp, nip);
    if (nID[3] < 0) hipLaunchKernelGGL(bc_jface_k, grid, dim3(128), 0, ctx.stream, p, ix, kx, jx, g, nkp, nip);
  }
  HIP_CHECK(hipGetLastError());
  HIP_CHECK(hipStreamSynchronize(ctx.stream));
}

void jacobi_(CZ_REAL* p, int* sz, int* idx, int* g, CZ_REAL* cf, CZ_REAL* omg, CZ_REAL* b, double* res, CZ_REAL* wk2,
             double* flop) {
  ensure_init();
  *flop += 18.0 * npts(idx);  // cz_solver.f90:315-318
  const Box bx = make_box(sz, idx, *g);
  if (bx.empty) return;
  czhip_jacobi_async(p, wk2, b, sz, idx, *g, cf, *omg, ctx.scal_dev + 0, 0, nullptr);
  launch_ewise<OP_COPY>(p, wk2, nullptr, (REAL)0, (REAL)0, bx);  // p <- wk2 on the inner box (:369-375)
  *res += read_scalar(0);                                          // :384 (double accumulation)
}

void psor2sma_core_(CZ_REAL* p, int* sz, int* idx, int* g, CZ_REAL* cf, int* ip, int* color, CZ_REAL* omg, CZ_REAL* b,
                    double* res, double* flop) {
  ensure_init();
  *flop += 18.0 * 0.5 * npts(idx);  // :438-441
  const Box bx = make_box(sz, idx, *g);
  if (bx.empty) return;
  czhip_rbsor_async(p, b, sz, idx, *g, cf, *ip, *color, *omg, ctx.scal_dev + 0, 0, nullptr);
  *res += read_scalar(0);
}

void blas_clear_(CZ_REAL* x, int* sz, int* g) {
  ensure_init();
  const size_t n = (size_t)(sz[0] + 2 * *g) * (size_t)(sz[1] + 2 * *g) * (size_t)(sz[2] + 2 * *g);
  HIP_CHECK(hipMemsetAsync(x, 0, n * sizeof(REAL), ctx.stream));
  HIP_CHECK(hipStreamSynchronize(ctx.stream));
}

void blas_copy_(CZ_REAL* dst, CZ_REAL* src, int* sz, int* g) {
  ensure_init();
  const size_t n = (size_t)(sz[0] + 2 * *g) * (size_t)(sz[1] + 2 * *g) * (size_t)(sz[2] + 2 * *g);
  HIP_CHECK(hipMemcpyAsync(dst, src, n * sizeof(REAL), hipMemcpyDeviceToDevice, ctx.stream));
  HIP_CHECK(hipStreamSynchronize(ctx.stream));
}

void blas_triad_(CZ_REAL* z, CZ_REAL* x, CZ_REAL* y, CZ_REAL* a, int* sz, int* idx, int* g, double* flop) {
  ensure_init();
  *flop += 2.0 * npts(idx);
  launch_ewise<OP_TRIAD>(z, x, y, *a, (REAL)0, make_box(sz, idx, *g));
  HIP_CHECK(hipStreamSynchronize(ctx.stream));
}

void blas_dot1_(CZ_REAL* r, CZ_REAL* p, int* sz, int* idx, int* g, double* flop) {
  ensure_init();
  *flop += 2.0 * npts(idx);
  launch_dot<0>(p, p, make_box(sz, idx, *g), ctx.scal_dev + 1);
  *r = (REAL)read_scalar(1);
}

void blas_dot2_(CZ_REAL* r, CZ_REAL* p, CZ_REAL* q, int* sz, int* idx, int* g, double* flop) {
  ensure_init();
  *flop += 2.0 * npts(idx);
  launch_dot<1>(p, q, make_box(sz, idx, *g), ctx.scal_dev + 1);
  *r = (REAL)read_scalar(1);
}

void blas_bicg_1_(CZ_REAL* p, CZ_REAL* r, CZ_REAL* q, CZ_REAL* beta, CZ_REAL* omg, int* sz, int* idx, int* g, double* flop) {
  ensure_init();
  *flop += 4.0 * npts(idx);
  launch_ewise<OP_BICG1>(p, r, q, *beta, *omg, make_box(sz, idx, *g));
  HIP_CHECK(hipStreamSynchronize(ctx.stream));
}

void blas_bicg_2_(CZ_REAL* z, CZ_REAL* x, CZ_REAL* y, CZ_REAL* a, CZ_REAL* b, int* sz, int* idx, int* g, double* flop) {
  ensure_init();
  *flop += 4.0 * npts(idx);
  launch_ewise<OP_BICG2>(z, x, y, *a, *b, make_box(sz, idx, *g));
  HIP_CHECK(hipStreamSynchronize(ctx.stream));
}

void blas_calc_ax_(CZ_REAL* ap, CZ_REAL* p, int* sz, int* idx, int* g, CZ_REAL* cf, double* flop) {
  ensure_init();
  *flop += 13.0 * npts(idx);
  const Box bx = make_box(sz, idx, *g);
  if (bx.empty) return;
  launch_stencil<MODE_AX>(p, p, ap, make_coef(cf, (REAL)0), bx, 0, nullptr, nullptr);
  HIP_CHECK(hipStreamSynchronize(ctx.stream));
}

void blas_calc_rk_(CZ_REAL* r, CZ_REAL* p, CZ_REAL* b, int* sz, int* idx, int* g, CZ_REAL* cf, double* flop) {
  ensure_init();
  *flop += 14.0 * npts(idx);
  const Box bx = make_box(sz, idx, *g);
  if (bx.empty) return;
  launch_stencil<MODE_RK>(p, b, r, make_coef(cf, (REAL)0), bx, 0, nullptr, nullptr);
  HIP_CHECK(hipStreamSynchronize(ctx.stream));
}

// ---- MAF flavour, drop-in symbols (cz_Ffunc.h:170-208, 524-553).  X, Y, Z are HOST arrays as in the reference.
void jacobi_maf_(CZ_REAL* p, int* sz, int* idx, int* g, CZ_REAL* X, CZ_REAL* Y, CZ_REAL* Z, CZ_REAL* omg, CZ_REAL* b, double* res,
                 CZ_REAL* wk2, CZ_REAL* tmp, double* flop) {
  ensure_init();
  *flop += 66.0 * npts(idx);                                    // cz_maf.f90:157-160
  for (int k = 0; k < sz[2] + 2 * *g; k++) tmp[k] = (REAL)0;    // :155 (host work array; only used by the _SVR build)
  const Box bx = make_box(sz, idx, *g);
  if (bx.empty) return;
  const MafArgs ma = upload_xyz(sz, *g, X, Y, Z, nullptr);
  sweep_async<MODE_JACOBI>(p, wk2, b, bx, make_coef_omg(*omg), 0, ctx.scal_dev + 0, 0, nullptr, CheckArgs(), &ma);
  launch_ewise<OP_COPY>(p, wk2, nullptr, (REAL)0, (REAL)0, bx);
  *res += read_scalar(0);
}

void psor2sma_core_maf_(CZ_REAL* p, int* sz, int* idx, int* g, CZ_REAL* X, CZ_REAL* Y, CZ_REAL* Z, int* ip, int* color,
                        CZ_REAL* omg, CZ_REAL* b, double* res, CZ_REAL* tmp, double* flop) {
  ensure_init();
  (void)tmp;
  *flop += 66.0 * 0.5 * npts(idx);
  const Box bx = make_box(sz, idx, *g);
  if (bx.empty) return;
  const MafArgs ma = upload_xyz(sz, *g, X, Y, Z, nullptr);
  sweep_async<MODE_RB>(p, p, b, bx, make_coef_omg(*omg), rb_parity(*g, idx, *ip, *color), ctx.scal_dev + 0, 0, nullptr,
                       CheckArgs(), &ma);
  *res += read_scalar(0);
}

void psor_(CZ_REAL* p, int* sz, int* idx, int* g, CZ_REAL* cf, CZ_REAL* omg, CZ_REAL* b, double* res, double* flop) {
  ensure_init();
  *flop += 18.0 * npts(idx);  // cz_solver.f90:237-240
  const Box bx = make_box(sz, idx, *g);
  if (bx.empty) return;
  launch_psor(p, b, make_coef(cf, *omg), bx, ctx.scal_dev + 0, 0, nullptr, nullptr);
  *res += read_scalar(0);
}

void psor_maf_(CZ_REAL* p, int* sz, int* idx, int* g, CZ_REAL* X, CZ_REAL* Y, CZ_REAL* Z, CZ_REAL* omg, CZ_REAL* b, double* res,
               double* flop) {
  ensure_init();
  *flop += 66.0 * npts(idx);  // cz_maf.f90:50-53
  const Box bx = make_box(sz, idx, *g);
  if (bx.empty) return;
  if (bx.g != 2) {
    cz_fatal(1, "czhip: the MAF kernels assume GUIDE = 2 (X(-1:sz+2), cz_maf.f90:36-38)\n");
  }
  const MafArgs ma = upload_xyz(sz, *g, X, Y, Z, nullptr);
  launch_psor(p, b, make_coef_omg(*omg), bx, ctx.scal_dev + 0, 0, nullptr, &ma);
  *res += read_scalar(0);
}

void calc_rk_maf_(CZ_REAL* r, CZ_REAL* p, CZ_REAL* b, int* sz, int* idx, int* g, CZ_REAL* X, CZ_REAL* Y, CZ_REAL* Z,
                  CZ_REAL* pvt, double* flop) {
  ensure_init();
  *flop += 63.0 * npts(idx);
  const Box bx = make_box(sz, idx, *g);
  if (bx.empty) return;
  const MafArgs ma = upload_xyz(sz, *g, X, Y, Z, pvt);
  launch_stencil_maf<MODE_RK>(p, b, r, (REAL)0, bx, 0, nullptr, nullptr, Fin(), ma);
  HIP_CHECK(hipStreamSynchronize(ctx.stream));
}

void calc_ax_maf_(CZ_REAL* ap, CZ_REAL* p, int* sz, int* idx, int* g, CZ_REAL* X, CZ_REAL* Y, CZ_REAL* Z, CZ_REAL* pvt,
                  double* flop) {
  ensure_init();
  *flop += 63.0 * npts(idx);
  const Box bx = make_box(sz, idx, *g);
  if (bx.empty) return;
  const MafArgs ma = upload_xyz(sz, *g, X, Y, Z, pvt);
  launch_stencil_maf<MODE_AX>(p, p, ap, (REAL)0, bx, 0, nullptr, nullptr, Fin(), ma);
  HIP_CHECK(hipStreamSynchronize(ctx.stream));
}

void search_pivot_(CZ_REAL* pvt, int* sz, int* idx, int* g, CZ_REAL* X, CZ_REAL* Y, CZ_REAL* Z) {
  ensure_init();
  const Box bx = make_box(sz, idx, *g);
  if (bx.empty) return;
  const MafArgs ma = upload_xyz(sz, *g, X, Y, Z, nullptr);
  launch_pivot(pvt, bx, ma);
  HIP_CHECK(hipStreamSynchronize(ctx.stream));
}

// ---- line SOR by PCR, drop-in symbols (cz_Ffunc.h:60-77 pcr_rb_, :440-443 imask_k_).  The six 1-D work arrays of the
// reference are host scratch of its CPU implementation; the GPU keeps the line systems in LDS and does not touch them.
void pcr_rb_(int* sz, int* idx, int* g, int* pn, int* ofst, int* color, CZ_REAL* x, CZ_REAL* msk, CZ_REAL* rhs, CZ_REAL* a,
             CZ_REAL* c, CZ_REAL* d, CZ_REAL* a1, CZ_REAL* c1, CZ_REAL* d1, CZ_REAL* omg, double* res, double* flop) {
  ensure_init();
  (void)ofst, (void)a, (void)c, (void)d, (void)a1, (void)c1, (void)d1;
  const double nk = idx[5] - idx[4] + 1;
  *flop += (double)((idx[3] - idx[2] + 1) * (idx[1] - idx[0] + 1)) *
           (nk * 6.0 + nk * (*pn - 1) * 14.0 + (double)(1 << (*pn - 1)) * 9.0 + nk * 6.0 + 6.0) * 0.5;  // cz_solver.f90:523-531
  const Box bx = make_box(sz, idx, *g);
  if (bx.empty) return;
  launch_pcr_rb(x, msk, rhs, bx, idx, *pn, *color, *omg, ctx.scal_dev + 0, 0);
  *res += read_scalar(0);
}

namespace {
double pcr_flop(const int* idx, int stages, double fin) {
  const double nk = idx[5] - idx[4] + 1;
  return (double)((idx[3] - idx[2] + 1) * (idx[1] - idx[0] + 1)) * (nk * 6.0 + nk * stages * 14.0 + fin + nk * 6.0 + 6.0);
}
}  // namespace

namespace {
// cz_maf.f90:470-483 etc.
double pcr_maf_flop(const int* idx, int pn, double fin) {
  const double nk = idx[5] - idx[4] + 1;
  return (double)((idx[3] - idx[2] + 1) * (idx[1] - idx[0] + 1)) *
         ((24.0 + 3.0 * 2.0 + 12.0) + nk * (11.0 + 10.0) + (nk - 2.0) * 6.0 + nk * (double)(pn - 1) * 16.0 + (double)(1 << (pn - 1)) * fin + nk * 6.0);
}
void pcr_maf_dropin(int* sz, int* idx, int g, int pn, int order, int color, CZ_REAL* x, CZ_REAL* msk, CZ_REAL* rhs, CZ_REAL* XX, CZ_REAL* YY,
                    CZ_REAL* ZZ, CZ_REAL omg, double* res) {
  const Box bx = make_box(sz, idx, g);
  if (bx.empty) return;
  if (bx.g != 2) {
    cz_fatal(1, "czhip: the MAF kernels assume GUIDE = 2 (X(-1:sz+2))\n");
  }
  const MafArgs ma = upload_xyz(sz, g, XX, YY, ZZ, nullptr);
  launch_pcr_maf(x, msk, rhs, bx, idx, pn, order, color, omg, ctx.scal_dev + 0, 0, ma);
  *res += read_scalar(0);
}
}  // namespace

// ---- the MAF line solvers, drop-in symbols (cz_Ffunc.h:211-315 <- cz_maf.f90:442-1560).  XX, YY, ZZ are host arrays like in
// jacobi_maf_; the one-dimensional work arrays and tmp are ignored.
void pcr_rb_maf_(int* sz, int* idx, int* g, int* pn, int* ofst, int* color, CZ_REAL* x, CZ_REAL* msk, CZ_REAL* rhs, CZ_REAL* XX, CZ_REAL* YY,
                 CZ_REAL* ZZ, CZ_REAL* a, CZ_REAL* c, CZ_REAL* d, CZ_REAL* aw, CZ_REAL* cw, CZ_REAL* dw, CZ_REAL* omg, double* res, CZ_REAL* tmp,
                 double* flop) {
  ensure_init();
  (void)ofst, (void)a, (void)c, (void)d, (void)aw, (void)cw, (void)dw, (void)tmp;
  *flop += pcr_maf_flop(idx, *pn, 11.0) * 0.5;
  pcr_maf_dropin(sz, idx, *g, *pn, 0, *color, x, msk, rhs, XX, YY, ZZ, *omg, res);
}
void pcr_rb_esa_maf_(int* sz, int* idx, int* g, int* pn, int* ofst, int* color, int* s, CZ_REAL* x, CZ_REAL* msk, CZ_REAL* rhs, CZ_REAL* XX,
                     CZ_REAL* YY, CZ_REAL* ZZ, CZ_REAL* a, CZ_REAL* c, CZ_REAL* d, CZ_REAL* aw, CZ_REAL* cw, CZ_REAL* dw, CZ_REAL* omg, double* res,
                     CZ_REAL* tmp, double* flop) {
  ensure_init();
  (void)ofst, (void)s, (void)a, (void)c, (void)d, (void)aw, (void)cw, (void)dw, (void)tmp;
  *flop += pcr_maf_flop(idx, *pn, 11.0) * 0.5;
  pcr_maf_dropin(sz, idx, *g, *pn, 0, *color, x, msk, rhs, XX, YY, ZZ, *omg, res);
}
void pcr_maf_(int* sz, int* idx, int* g, int* pn, CZ_REAL* x, CZ_REAL* msk, CZ_REAL* rhs, CZ_REAL* XX, CZ_REAL* YY, CZ_REAL* ZZ, CZ_REAL* a,
              CZ_REAL* c, CZ_REAL* d, CZ_REAL* aw, CZ_REAL* cw, CZ_REAL* dw, CZ_REAL* omg, double* res, CZ_REAL* tmp, double* flop) {
  ensure_init();
  (void)a, (void)c, (void)d, (void)aw, (void)cw, (void)dw, (void)tmp;
  *flop += pcr_maf_flop(idx, *pn, 11.0);
  pcr_maf_dropin(sz, idx, *g, *pn, 1, 0, x, msk, rhs, XX, YY, ZZ, *omg, res);
}
void pcr_eda_maf_(int* sz, int* idx, int* g, int* pn, CZ_REAL* x, CZ_REAL* msk, CZ_REAL* rhs, CZ_REAL* XX, CZ_REAL* YY, CZ_REAL* ZZ, CZ_REAL* aw,
                  CZ_REAL* cw, CZ_REAL* dw, CZ_REAL* omg, double* res, CZ_REAL* tmp, double* flop) {
  ensure_init();
  (void)aw, (void)cw, (void)dw, (void)tmp;
  *flop += pcr_maf_flop(idx, *pn, 9.0);
  pcr_maf_dropin(sz, idx, *g, *pn, 1, 0, x, msk, rhs, XX, YY, ZZ, *omg, res);
}
void pcr_esa_maf_(int* sz, int* idx, int* g, int* pn, int* s, CZ_REAL* x, CZ_REAL* msk, CZ_REAL* rhs, CZ_REAL* XX, CZ_REAL* YY, CZ_REAL* ZZ,
                  CZ_REAL* a, CZ_REAL* c, CZ_REAL* d, CZ_REAL* aw, CZ_REAL* cw, CZ_REAL* dw, CZ_REAL* omg, double* res, CZ_REAL* tmp, double* flop) {
  ensure_init();
  (void)s, (void)a, (void)c, (void)d, (void)aw, (void)cw, (void)dw, (void)tmp;
  *flop += pcr_maf_flop(idx, *pn, 9.0);
  pcr_maf_dropin(sz, idx, *g, *pn, 1, 0, x, msk, rhs, XX, YY, ZZ, *omg, res);
}

void pcr_(int* sz, int* idx, int* g, int* pn, CZ_REAL* x, CZ_REAL* msk, CZ_REAL* rhs, CZ_REAL* a, CZ_REAL* c, CZ_REAL* d, CZ_REAL* a1,
          CZ_REAL* c1, CZ_REAL* d1, CZ_REAL* omg, double* res, double* flop) {
  ensure_init();
  (void)a, (void)c, (void)d, (void)a1, (void)c1, (void)d1;
  *flop += pcr_flop(idx, *pn - 2, (double)(1 << (*pn - 2)) * 74.0);  // cz_solver.f90:689-696
  const Box bx = make_box(sz, idx, *g);
  if (bx.empty) return;
  launch_pcr_variant(x, nullptr, msk, rhs, bx, idx, *pn, 1, 0, 1, *omg, ctx.scal_dev + 0, 0);
  *res += read_scalar(0);
}

void pcr_eda_(int* sz, int* idx, int* g, int* pn, CZ_REAL* x, CZ_REAL* msk, CZ_REAL* rhs, CZ_REAL* a1, CZ_REAL* c1, CZ_REAL* d1, CZ_REAL* omg,
              double* res, double* flop) {
  ensure_init();
  (void)a1, (void)c1, (void)d1;
  *flop += pcr_flop(idx, *pn - 1, (double)(1 << (*pn - 1)) * 9.0);  // :908-915
  const Box bx = make_box(sz, idx, *g);
  if (bx.empty) return;
  launch_pcr_variant(x, nullptr, msk, rhs, bx, idx, *pn, 1, 0, 0, *omg, ctx.scal_dev + 0, 0);
  *res += read_scalar(0);
}

void pcr_esa_(int* sz, int* idx, int* g, int* pn, int* s, CZ_REAL* x, CZ_REAL* msk, CZ_REAL* rhs, CZ_REAL* a, CZ_REAL* c, CZ_REAL* d,
              CZ_REAL* a1, CZ_REAL* c1, CZ_REAL* d1, CZ_REAL* omg, double* res, double* flop) {
  ensure_init();
  (void)s, (void)a, (void)c, (void)d, (void)a1, (void)c1, (void)d1;
  *flop += pcr_flop(idx, *pn - 2, (double)(1 << (*pn - 2)) * 78.0);  // :1078-1085
  const Box bx = make_box(sz, idx, *g);
  if (bx.empty) return;
  launch_pcr_variant(x, nullptr, msk, rhs, bx, idx, *pn, 1, 0, 1, *omg, ctx.scal_dev + 0, 0);
  *res += read_scalar(0);
}

void pcr_rb_esa_(int* sz, int* idx, int* g, int* pn, int* ofst, int* color, int* s, CZ_REAL* x, CZ_REAL* msk, CZ_REAL* rhs, CZ_REAL* a,
                 CZ_REAL* c, CZ_REAL* d, CZ_REAL* a1, CZ_REAL* c1, CZ_REAL* d1, CZ_REAL* omg, double* res, double* flop) {
  ensure_init();
  (void)ofst, (void)s, (void)a, (void)c, (void)d, (void)a1, (void)c1, (void)d1;
  *flop += pcr_flop(idx, *pn - 2, (double)(1 << (*pn - 2)) * 78.0) * 0.5;  // :1291-1299
  const Box bx = make_box(sz, idx, *g);
  if (bx.empty) return;
  launch_pcr_variant(x, nullptr, msk, rhs, bx, idx, *pn, 0, *color, 1, *omg, ctx.scal_dev + 0, 0);
  *res += read_scalar(0);
}

void pcr_j_esa_(int* sz, int* idx, int* g, int* pn, int* s, CZ_REAL* x, CZ_REAL* msk, CZ_REAL* rhs, CZ_REAL* a, CZ_REAL* c, CZ_REAL* d,
                CZ_REAL* a1, CZ_REAL* c1, CZ_REAL* d1, CZ_REAL* src, CZ_REAL* wrk, CZ_REAL* omg, double* res, double* flop) {
  ensure_init();
  (void)s, (void)a, (void)c, (void)d, (void)a1, (void)c1, (void)d1, (void)src;
  *flop += pcr_flop(idx, *pn - 1, (double)(1 << (*pn - 1)) * 9.0);  // :1499-1506
  const Box bx = make_box(sz, idx, *g);
  if (bx.empty) return;
  launch_pcr_variant(x, wrk, msk, rhs, bx, idx, *pn, 2, 0, 0, *omg, ctx.scal_dev + 0, 0);
  czhip_internal::copy_inner_async(x, wrk, sz, idx, *g);  // :1655-1663
  *res += read_scalar(0);
}

void imask_k_(CZ_REAL* x, int* sz, int* idx, int* g) {
  ensure_init();
  Box bx = make_box(sz, idx, *g);
  launch_imask(x, bx);
  HIP_CHECK(hipStreamSynchronize(ctx.stream));
}

}  // extern "C"

// internal hooks for the driver (cz_driver.cpp): asynchronous forms of the blas kernels
namespace czhip_internal {
hipStream_t stream() {
  ensure_init();
  return ctx.stream;
}

// Decomposed runs: the sweeps leave k CUs of every XCD to what the exchange stream launches while an interior sweep fills the chip -- shell
// slabs, pack / unpack and above all RCCL's send/recv kernels, which need CUs of their own for as long as a message is in flight.  The
// interior launch of the two-stage pass is sized to fill its slots in ONE round with workgroups that live as long as the launch
// (pair_tj_model), so a stream priority alone frees nothing before the end.  The reservation is made through the launch geometry:
// pair_tj_model counts num_cu/8 - k slots per XCD, so a one-round launch leaves k CUs of every XCD without a workgroup (a 1024-thread
// workgroup with its ~130 KB of LDS takes a CU for itself), and launches of several rounds free slots all the time anyway.  Costs nothing at
// 512^3 FP32, where the launch uses 30 of the 32 slots as it is (profiles/r03/cu_reserve_cost.txt).
// (Rounds 3: a hardware-enforced form -- a CU mask on the compute stream's queue, hipExtStreamCreateWithCUMask -- was measured and removed in
// round 4: one-round launches took 1.8x as long on a masked queue, and swapping the context's stream under live users (timer events, the
// driver's cached stream, RCCL's enqueue state) ended two measurement runs in a hang.  tools/cumask_lab.hip keeps the experiment.)
// Returns the reservation in force.
int reserve_comm_cus(int k) {
  ensure_init();
  const int per_xcd = ctx.num_cu / 8;
  if (ctx.num_cu % 8 != 0 || per_xcd < 4) k = 0;  // not the 8-XCD part this was measured on
  k = std::max(0, std::min(k, per_xcd / 2));
  ctx.cu_reserved = k;
  return k;
}
int comm_cus_reserved() { return ctx.cu_reserved; }
void triad_async(REAL* z, const REAL* x, const REAL* y, REAL a, const int* sz, const int* idx, int g, const REAL* a_dev) {
  launch_ewise<OP_TRIAD>(z, x, y, a, (REAL)0, make_box(sz, idx, g), a_dev);
}
void bicg1_async(REAL* p, const REAL* r, const REAL* q, REAL beta, REAL omg, const int* sz, const int* idx, int g) {
  launch_ewise<OP_BICG1>(p, r, q, beta, omg, make_box(sz, idx, g));
}
void bicg2_async(REAL* z, const REAL* x, const REAL* y, REAL a, REAL b, const int* sz, const int* idx, int g, const REAL* a_dev, const REAL* b_dev) {
  launch_ewise<OP_BICG2>(z, x, y, a, b, make_box(sz, idx, g), a_dev, b_dev);
}
void calc_ax_async(REAL* ap, const REAL* p, const int* sz, const int* idx, int g, const REAL* cf) {
  const Box bx = make_box(sz, idx, g);
  if (!bx.empty) launch_stencil<MODE_AX>(p, p, ap, make_coef(cf, (REAL)0), bx, 0, nullptr, nullptr);
}
void calc_rk_async(REAL* r, const REAL* p, const REAL* b, const int* sz, const int* idx, int g, const REAL* cf) {
  const Box bx = make_box(sz, idx, g);
  if (!bx.empty) launch_stencil<MODE_RK>(p, b, r, make_coef(cf, (REAL)0), bx, 0, nullptr, nullptr);
}
// SpMV with the two dot products of the result folded in: dots_dev[0] = ap.y, dots_dev[1] = ap.ap
void calc_ax_dots_async(REAL* ap, const REAL* p, const REAL* y, const int* sz, const int* idx, int g, const REAL* cf,
                        const MafPtrs* maf, double* dots_dev) {
  const Box bx = make_box(sz, idx, g);
  if (bx.empty) {
    HIP_CHECK(hipMemsetAsync(dots_dev, 0, 2 * sizeof(double), ctx.stream));
    return;
  }
  Fin fin;
  fin.dst = dots_dev, fin.dst2 = dots_dev + 1, fin.ax_dots = 1, fin.doty = y, fin.counter = ctx.counter;
  if (maf) {
    MafArgs ma{maf->xc, maf->yc, maf->zc, maf->pvt};
    launch_stencil_maf<MODE_AX>(p, p, ap, (REAL)0, bx, 0, nullptr, nullptr, fin, ma);
  } else {
    launch_stencil<MODE_AX>(p, p, ap, make_coef(cf, (REAL)0), bx, 0, nullptr, nullptr, fin);
  }
}
// z = a*x + y with dots_dev[0] = z.z, dots_dev[1] = z.w
void triad_dots_async(REAL* z, const REAL* x, const REAL* y, const REAL* w, REAL a, const int* sz, const int* idx, int g,
                      double* dots_dev, const REAL* a_dev) {
  const Box b = make_box(sz, idx, g);
  if (b.empty) {
    HIP_CHECK(hipMemsetAsync(dots_dev, 0, 2 * sizeof(double), ctx.stream));
    return;
  }
  const int nplanes = b.jj1 - b.jj0 + 1;
  ScopedTimer tm(LBL_EWISE);
  if (rows_ok(b, {z, x, y, w})) {
    EGeom e = make_egeom<VW>(b);
    e.pa = a_dev;
    const unsigned gx = (unsigned)((e.Fend - e.F0 + 255) / 256);
    const unsigned gy = (unsigned)std::max(1, std::min(nplanes, (int)(4096 / gx)));
    ensure_partials((size_t)2 * gx * gy);
    hipLaunchKernelGGL((triad_dots_k<VW>), dim3(gx, gy), dim3(256), 0, ctx.stream, z, x, y, w, a, e, nplanes, ctx.partials, dots_dev,
                       ctx.counter);
  } else {
    EGeom e = make_egeom<1>(b);
    e.pa = a_dev;
    const unsigned gx = (unsigned)((e.Fend - e.F0 + 255) / 256);
    const unsigned gy = (unsigned)std::max(1, std::min(nplanes, (int)(4096 / gx)));
    ensure_partials((size_t)2 * gx * gy);
    hipLaunchKernelGGL((triad_dots_k<1>), dim3(gx, gy), dim3(256), 0, ctx.stream, z, x, y, w, a, e, nplanes, ctx.partials, dots_dev,
                       ctx.counter);
  }
  HIP_CHECK(hipGetLastError());
}
// alpha (step 1) / omega (step 2) of BiCGSTAB from the dot products in dots_dev, into sc_dev[0..3] = alpha, omega, -alpha, -omega (bicg_scal_k)
void bicg_scalar_async(int step, const double* dots_dev, REAL rho, REAL* sc_dev) {
  if (step == 1) hipLaunchKernelGGL(bicg_scal_k<1>, dim3(1), dim3(1), 0, ctx.stream, dots_dev, rho, sc_dev);
  else hipLaunchKernelGGL(bicg_scal_k<2>, dim3(1), dim3(1), 0, ctx.stream, dots_dev, rho, sc_dev);
  HIP_CHECK(hipGetLastError());
}
void dot1_async(const REAL* p, const int* sz, const int* idx, int g, double* dst_dev) {
  launch_dot<0>(p, p, make_box(sz, idx, g), dst_dev);
}
void dot2_async(const REAL* p, const REAL* q, const int* sz, const int* idx, int g, double* dst_dev) {
  launch_dot<1>(p, q, make_box(sz, idx, g), dst_dev);
}
int pcr_num_stage(int n) { return num_stage(n); }
void pcr_rb_async(REAL* x, const REAL* msk, const REAL* rhs, const int* sz, const int* idx, int g, int pn, int color, REAL omg,
                  double* res_dev, int accumulate) {
  launch_pcr_rb(x, msk, rhs, make_box(sz, idx, g), idx, pn, color, omg, res_dev, accumulate);
}
// order 0: colour `sel` in place; 1: lexicographic in place (one launch per diagonal); 2: all columns x -> wout
void pcr_variant_async(REAL* x, REAL* wout, const REAL* msk, const REAL* rhs, const int* sz, const int* idx, int g, int pn, int order,
                       int sel, int final4, REAL omg, double* res_dev, int accumulate) {
  ensure_init();
  launch_pcr_variant(x, wout, msk, rhs, make_box(sz, idx, g), idx, pn, order, sel, final4, omg, res_dev, accumulate);
}
// the bookkeeping of a fused pair on another stream of the caller (decomposed runs: all-reduce + test on the exchange stream);
// snap_dev: see check_k
void check2_on_stream(hipStream_t st, const double* res_dev, double res_normal, double eps, int itr, double* hist_dev, int* flag_dev,
                      int* conv_itr_dev, int* snap_dev) {
  ensure_init();
  hipLaunchKernelGGL(check2_k, dim3(1), dim3(1), 0, st, res_dev, res_normal, eps, itr, hist_dev, flag_dev, conv_itr_dev, snap_dev);
  HIP_CHECK(hipGetLastError());
}
void check_on_stream(hipStream_t st, const double* res_dev, double res_normal, double eps, int itr, double* hist_dev, int* flag_dev,
                     int* conv_itr_dev, int* snap_dev) {
  ensure_init();
  hipLaunchKernelGGL(check_k, dim3(1), dim3(1), 0, st, res_dev, res_normal, eps, itr, hist_dev, flag_dev, conv_itr_dev, snap_dev);
  HIP_CHECK(hipGetLastError());
}
// Start of a solve: the arrival ticket of the in-kernel finalisations back to zero.  A launch that was cut short (a lagged pass
// overtaken by convergence in an older build, an aborted solve) must not leave a count behind for the next solve on this context.
void reset_ticket() {
  ensure_init();
  HIP_CHECK(hipMemsetAsync(ctx.counter, 0, 64, ctx.stream));
  if (ctx.psor_ctl) HIP_CHECK(hipMemsetAsync(ctx.psor_ctl, 0, 256, ctx.stream));  // (and the sticky "a psor sweep gave up" word of an earlier solve)
}
// MAF line solvers: order 0 = colour `sel` in place, 1 = lexicographic in place; xc, yc, zc device arrays
void pcr_maf_async(REAL* x, const REAL* msk, const REAL* rhs, const int* sz, const int* idx, int g, int pn, int order, int sel,
                   const REAL* xc, const REAL* yc, const REAL* zc, REAL omg, double* res_dev, int accumulate) {
  ensure_init();
  MafArgs ma{xc, yc, zc, nullptr};
  launch_pcr_maf(x, msk, rhs, make_box(sz, idx, g), idx, pn, order, sel, omg, res_dev, accumulate, ma);
}
void imask_async(REAL* x, const int* sz, const int* idx, int g) { launch_imask(x, make_box(sz, idx, g)); }
// MAF flavour, device-resident coordinates (xc|yc|zc and pvt are device pointers)
void jacobi_maf_async(const REAL* p_in, REAL* p_out, const REAL* b, const int* sz, const int* idx, int g, const REAL* xc,
                      const REAL* yc, const REAL* zc, REAL omg, double* res_dev, const int* skip, int check, double res_normal,
                      double eps, int itr, double* hist, int* flag, int* conv_itr) {
  const Box bx = make_box(sz, idx, g);
  if (bx.empty) return;
  MafArgs ma{xc, yc, zc, nullptr};
  CheckArgs ck;
  if (check) ck.enabled = 1, ck.itr = itr, ck.res_normal = res_normal, ck.eps = eps, ck.hist = hist, ck.flag = flag, ck.conv_itr = conv_itr;
  sweep_async<MODE_JACOBI>(p_in, p_out, b, bx, make_coef_omg(omg), 0, res_dev, 0, check ? flag : skip, ck, &ma);
}
void rbsor_maf_async(REAL* p, const REAL* b, const int* sz, const int* idx, int g, const REAL* xc, const REAL* yc, const REAL* zc,
                     int ofst, int color, REAL omg, double* res_dev, int accumulate, const int* skip, int check, double res_normal,
                     double eps, int itr, double* hist, int* flag, int* conv_itr) {
  const Box bx = make_box(sz, idx, g);
  if (bx.empty) return;
  MafArgs ma{xc, yc, zc, nullptr};
  CheckArgs ck;
  if (check) ck.enabled = 1, ck.itr = itr, ck.res_normal = res_normal, ck.eps = eps, ck.hist = hist, ck.flag = flag, ck.conv_itr = conv_itr;
  sweep_async<MODE_RB>(p, p, b, bx, make_coef_omg(omg), rb_parity(g, idx, ofst, color), res_dev, accumulate, check ? flag : skip, ck,
                       &ma);
}
// MAF flavour of the two-stage pass (cz_maf.f90:131-285 twice / both colours of :287-438): two jacobi_maf sweeps (rb_ofst < 0; res_dev[0..1])
// or one red-black iteration (rb_ofst = the reference's ofst; res_dev[0]) per pass over memory, u -> w.  Same conventions as
// czhip_jacobi2_async; returns 0 when the geometry does not suit the kernel (nothing launched).
int pair_maf_async(const REAL* u, REAL* w, const REAL* b, const int* sz, const int* idx, const int* idx1, int g, const REAL* xc,
                   const REAL* yc, const REAL* zc, REAL omg, int rb_ofst, double* res_dev, double res_normal, double eps, int itr,
                   double* hist_dev, int* flag_dev, int* conv_itr_dev, const int* skip_flag_dev) {
  ensure_init();
  if (!ctx.tune.fuse_fin) return 0;
  const Box bx = make_box(sz, idx, g);
  if (bx.empty || g < 2) return 0;
  const Box ba = idx1 ? make_box(sz, idx1, g) : bx;
  MafArgs ma{xc, yc, zc, nullptr};
  Fin2 fin;
  fin.dst = res_dev;
  fin.single = rb_ofst >= 0;
  if (hist_dev) {
    fin.do_check = 1, fin.itr = itr, fin.res_normal = res_normal, fin.eps = eps;
    fin.hist = hist_dev, fin.flag = flag_dev, fin.conv_itr = conv_itr_dev;
  }
  const int* skip = hist_dev ? flag_dev : skip_flag_dev;
  if (rb_ofst >= 0) return launch_jacobi2<1>(u, b, w, make_coef_omg(omg), bx, ba, skip, fin, rb_parity(g, idx, rb_ofst, 0), 0, false, &ma) ? 1 : 0;
  return launch_jacobi2<0>(u, b, w, make_coef_omg(omg), bx, ba, skip, fin, 0, 0, false, &ma) ? 1 : 0;
}
// has a one-launch psor sweep on this context given up a wait since the last call? (synchronises; the sweep's residual was NaN, the field is void)
int psor_failed() {
  ensure_init();
  if (!ctx.psor_ctl) return 0;
  unsigned h = 0;
  HIP_CHECK(hipMemcpyAsync(&h, ctx.psor_ctl + 2, sizeof(h), hipMemcpyDeviceToHost, ctx.stream));
  HIP_CHECK(hipStreamSynchronize(ctx.stream));
  if (h) HIP_CHECK(hipMemsetAsync(ctx.psor_ctl + 2, 0, sizeof(unsigned), ctx.stream));
  return h != 0;
}
// one psor / psor_maf sweep (xc == nullptr: constant coefficients cf), res_dev[0] = or += sum dp^2
void psor_async(REAL* p, const REAL* b, const int* sz, const int* idx, int g, const REAL* cf, const REAL* xc, const REAL* yc,
                const REAL* zc, REAL omg, double* res_dev, int accumulate, const int* skip) {
  ensure_init();
  const Box bx = make_box(sz, idx, g);
  if (xc) {
    MafArgs ma{xc, yc, zc, nullptr};
    launch_psor(p, b, make_coef_omg(omg), bx, res_dev, accumulate, skip, &ma);
  } else {
    launch_psor(p, b, make_coef(cf, omg), bx, res_dev, accumulate, skip, nullptr);
  }
}
void calc_ax_maf_async(REAL* ap, const REAL* p, const int* sz, const int* idx, int g, const REAL* xc, const REAL* yc, const REAL* zc,
                       const REAL* pvt) {
  const Box bx = make_box(sz, idx, g);
  MafArgs ma{xc, yc, zc, pvt};
  if (!bx.empty) launch_stencil_maf<MODE_AX>(p, p, ap, (REAL)0, bx, 0, nullptr, nullptr, Fin(), ma);
}
void calc_rk_maf_async(REAL* r, const REAL* p, const REAL* b, const int* sz, const int* idx, int g, const REAL* xc, const REAL* yc,
                       const REAL* zc, const REAL* pvt) {
  const Box bx = make_box(sz, idx, g);
  MafArgs ma{xc, yc, zc, pvt};
  if (!bx.empty) launch_stencil_maf<MODE_RK>(p, b, r, (REAL)0, bx, 0, nullptr, nullptr, Fin(), ma);
}
void search_pivot_async(REAL* pvt, const int* sz, const int* idx, int g, const REAL* xc, const REAL* yc, const REAL* zc) {
  const Box bx = make_box(sz, idx, g);
  MafArgs ma{xc, yc, zc, nullptr};
  launch_pivot(pvt, bx, ma);
}
// ---- a fused pair of sweeps split into shell + interior launches (decomposed runs).  rb < 0: two Jacobi
// sweeps, res_dev[0..1]; rb >= 0: one red-black iteration with colour parity `rb` (rb_par), res_dev[0].
int rb_par(int g, const int* idx, int ofst) { return rb_parity(g, idx, ofst, 0); }

// Split of an inner box for overlapped exchanges: the cells within two layers of a rank-internal face (nID[f] >= 0; what
// the neighbours receive as their two ghost layers) form up to six disjoint slabs -- J faces first (contiguous planes), then
// I, then K -- and the rest is the interior.  Returns the number of slabs (0: nothing to split or the box is too thin);
// boxes: n x (ist,ied,jst,jed,kst,ked); interior1 = first-sweep range of the interior.
int pair_plan(const int* O, const int* nID, int* boxes, int* interior, int* interior1) {
  for (int a = 0; a < 3; a++) {
    interior[2 * a] = O[2 * a] + (nID[2 * a] >= 0 ? 2 : 0);
    interior[2 * a + 1] = O[2 * a + 1] - (nID[2 * a + 1] >= 0 ? 2 : 0);
    if (interior[2 * a + 1] - interior[2 * a] + 1 < 2) return 0;
  }
  for (int f = 0; f < 6; f++) interior1[f] = interior[f] + ((nID[f] >= 0) ? ((f & 1) ? 1 : -1) : 0);
  int n = 0;
  auto add = [&](int i0, int i1, int j0, int j1, int k0, int k1) {
    int* b = boxes + 6 * n++;
    b[0] = i0, b[1] = i1, b[2] = j0, b[3] = j1, b[4] = k0, b[5] = k1;
  };
  const int* I = interior;
  if (nID[2] >= 0) add(O[0], O[1], O[2], O[2] + 1, O[4], O[5]);
  if (nID[3] >= 0) add(O[0], O[1], O[3] - 1, O[3], O[4], O[5]);
  if (nID[0] >= 0) add(O[0], O[0] + 1, I[2], I[3], O[4], O[5]);
  if (nID[1] >= 0) add(O[1] - 1, O[1], I[2], I[3], O[4], O[5]);
  if (nID[4] >= 0) add(I[0], I[1], I[2], I[3], O[4], O[4] + 1);
  if (nID[5] >= 0) add(I[0], I[1], I[2], I[3], O[5] - 1, O[5]);
  return n;
}

int pair_probe(const REAL* u, REAL* w, const REAL* b, const int* sz, const int* idx, const int* idx1, int g, REAL dd, int maf) {
  ensure_init();
  if (!ctx.tune.fuse_fin || g < 2) return 0;
  const Box bx = make_box(sz, idx, g);
  if (bx.empty) return 0;
  const Box ba = make_box(sz, idx1, g);
  Coef c = make_coef_omg((REAL)1);
  c.dd = dd;  // the divisor decides too: the pass divides with the hoisted form (cz_k_fastdiv.h)
  const MafArgs ma = MafArgs();
  return launch_jacobi2<0>(u, b, w, c, bx, ba, nullptr, Fin2(), 0, 0, true, maf ? &ma : nullptr) ? 1 : 0;
}

// maf: the MAF flavour (weights from the device coordinate arrays; cf is then not used)
void pair_shell_async(const REAL* u, REAL* w, const REAL* b, const int* sz, const int* idx1_brick, const int* boxes, int n, int g,
                      const REAL* cf, REAL omg, int rb, const int* skip, hipStream_t st, const MafPtrs* maf) {
  ensure_init();
  const Box ba = make_box(sz, idx1_brick, g);
  if (!st) st = ctx.stream;
  if (maf) {
    const MafArgs ma{maf->xc, maf->yc, maf->zc, nullptr};
    if (rb >= 0) launch_pair_shell<1, 1>(u, b, w, make_coef_omg(omg), sz, g, ba, boxes, n, rb, skip, st, ma);
    else launch_pair_shell<0, 1>(u, b, w, make_coef_omg(omg), sz, g, ba, boxes, n, 0, skip, st, ma);
    return;
  }
  if (rb >= 0) launch_pair_shell<1>(u, b, w, make_coef(cf, omg), sz, g, ba, boxes, n, rb, skip, st);
  else launch_pair_shell<0>(u, b, w, make_coef(cf, omg), sz, g, ba, boxes, n, 0, skip, st);
}

// res_dev += the sums of the last pair_shell_async launch (which ran on another stream beside the interior launch); stream-ordered on `st`
void pair_shell_fold_async(double* res_dev, int single, const int* skip, hipStream_t st) {
  ensure_init();
  if (ctx.shell_pending <= 0) return;
  hipLaunchKernelGGL(shell_fold_k, dim3(1), dim3(256), 0, st ? st : ctx.stream, ctx.shell_partials, ctx.shell_pending, res_dev, single, skip);
  HIP_CHECK(hipGetLastError());
  ctx.shell_pending = 0;
}

int pair_box_async(const REAL* u, REAL* w, const REAL* b, const int* sz, const int* idx, const int* idx1, int g, const REAL* cf,
                   REAL omg, int rb, double* res_dev, int with_shell, const int* skip, const MafPtrs* maf) {
  ensure_init();
  const Box bx = make_box(sz, idx, g);
  const Box ba = make_box(sz, idx1, g);
  Fin2 fin;
  fin.dst = res_dev;
  fin.single = rb >= 0;
  if (with_shell) {  // same stream as the shell launch: its sums join this launch's in the finaliser
    fin.extra = ctx.shell_partials, fin.n_extra = ctx.shell_pending;
    ctx.shell_pending = 0;
  }
  if (maf) {
    const MafArgs ma{maf->xc, maf->yc, maf->zc, nullptr};
    if (rb >= 0) return launch_jacobi2<1>(u, b, w, make_coef_omg(omg), bx, ba, skip, fin, rb, 0, false, &ma) ? 1 : 0;
    return launch_jacobi2<0>(u, b, w, make_coef_omg(omg), bx, ba, skip, fin, 0, 0, false, &ma) ? 1 : 0;
  }
  if (rb >= 0) return launch_jacobi2<1>(u, b, w, make_coef(cf, omg), bx, ba, skip, fin, rb) ? 1 : 0;
  return launch_jacobi2<0>(u, b, w, make_coef(cf, omg), bx, ba, skip, fin) ? 1 : 0;
}

void copy_shell_async(REAL* dst, const REAL* src, const int* sz, const int* idx, int g) {
  ensure_init();
  const int nkp = sz[2] + 2 * g, nip = sz[0] + 2 * g, njp = sz[1] + 2 * g;
  const long long rows = (long long)nip * njp;
  hipLaunchKernelGGL(copy_shell_k, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, ctx.stream, dst, src, nkp, nip, njp,
                     idx[4] + g - 1, idx[5] + g - 1, idx[0] + g - 1, idx[1] + g - 1, idx[2] + g - 1, idx[3] + g - 1);
  HIP_CHECK(hipGetLastError());
}
void copy_inner_async(REAL* dst, const REAL* src, const int* sz, const int* idx, int g) {
  launch_ewise<OP_COPY>(dst, src, nullptr, (REAL)0, (REAL)0, make_box(sz, idx, g));
}
void bc_async(const int* sz, int g, REAL* p, REAL dh, const REAL* org, const int* nID, int ioff, int joff) {
  // same launches as bc_k_ without the trailing synchronisation
  ensure_init();
  const int ix = sz[0], jx = sz[1], kx = sz[2];
  const int nkp = kx + 2 * g, nip = ix + 2 * g;
  if (nID[4] < 0 || nID[5] < 0) {
    const REAL* tab = bc_table(ix, jx, dh, org, ioff, joff);
    dim3 grid((ix + 127) / 128, jx);
    if (nID[4] < 0) hipLaunchKernelGGL(bc_kface_k, grid, dim3(128), 0, ctx.stream, p, tab, ix, jx, 1, g, nkp, nip);
    if (nID[5] < 0) hipLaunchKernelGGL(bc_kface_k, grid, dim3(128), 0, ctx.stream, p, tab, ix, jx, kx, g, nkp, nip);
  }
  {
    dim3 grid((kx + 127) / 128, jx);
    if (nID[0] < 0) hipLaunchKernelGGL(bc_iface_k, grid, dim3(128), 0, ctx.stream, p, jx, kx, 1, g, nkp, nip);
    if (nID[1] < 0) hipLaunchKernelGGL(bc_iface_k, grid, dim3(128), 0, ctx.stream, p, jx, kx, ix, g, nkp, nip);
  }
  {
    dim3 grid((kx + 127) / 128, ix);
    if (nID[2] < 0) hipLaunchKernelGGL(bc_jface_k, grid, dim3(128), 0, ctx.stream, p, ix, kx, 1, g, nkp, nip);
    if (nID[3] < 0) hipLaunchKernelGGL(bc_jface_k, grid, dim3(128), 0, ctx.stream, p, ix, kx, jx, g, nkp, nip);
  }
  HIP_CHECK(hipGetLastError());
}
}  // namespace czhip_internal
