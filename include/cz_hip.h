/*
 * cz_hip.h -- C-ABI of the MI355X-native CubeZ hot path (libczhip_f32.so / libczhip_f64.so).
 *
 * Part 1 is the DROP-IN BOUNDARY: the same `extern "C" void name_(...)` symbols, argument order and
 * by-pointer conventions as the reference's Fortran interface, /root/reference/src/cz_cpp/cz_Ffunc.h:16-578
 * (the lines are cited per prototype).  The only change of contract: every 3-D array argument is a
 * DEVICE pointer obtained from czhip_alloc_s3d() (the replacement of czAllocR_S3D, cz.h:209-232);
 * sz/idx/g/cf/nID and all scalars stay HOST pointers, `res`/`flop` stay host in/out accumulators,
 * the dot results stay host outputs.  Every part-1 call is complete (results visible on the host and
 * in device memory) when it returns, like the Fortran it replaces.
 *
 * Part 2 is the runtime the reference does not need on a CPU (device selection, allocation, copies).
 *
 * Part 3 is the asynchronous, device-resident form of the same operations that the restated solver
 * loops (part 4, replacing CZ::JACOBI / RBSOR / PBiCGSTAB of cz_Poisson.cpp) are built from: no host
 * round trip per sweep, residual history and convergence flag kept on the device.
 *
 * Precision is a build-time switch exactly as in the reference (cz_Define.h:28-37,
 * -D_REAL_IS_DOUBLE_): the f32 and f64 libraries export the same symbol names.
 *
 * Memory layout of every 3-D array (cz_solver.f90:29): dense (NK+2g, NI+2g, NJ+2g), K fastest,
 * lower bound 1-g; linear index of 1-based (k,i,j) = (k+g-1) + (i+g-1)*(NK+2g) + (j+g-1)*(NK+2g)*(NI+2g).
 */
#ifndef CZ_HIP_H_
#define CZ_HIP_H_

#include <stddef.h>

#ifdef CZ_REAL_IS_DOUBLE
typedef double CZ_REAL;
#else
typedef float CZ_REAL;
#endif

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------------------------------------
 * Part 1 -- drop-in kernels (replace cz_solver.f90 / cz_blas.f90 behind cz_Ffunc.h)
 * ---------------------------------------------------------------------------------------------- */

/* cz_Ffunc.h:20-25  <- cz_solver.f90:22-191.  Dirichlet faces where nID[face] < 0. */
void bc_k_(int* sz, int* g, CZ_REAL* p, CZ_REAL* dh, CZ_REAL* org, int* nID);

/* cz_Ffunc.h:27-36  <- cz_solver.f90:284-387.  One relaxed-Jacobi sweep; result in p AND wk2 (inner box),
 * *res += sum dp^2, *flop += 18*npts. */
void jacobi_(CZ_REAL* p, int* sz, int* idx, int* g, CZ_REAL* cf, CZ_REAL* omg, CZ_REAL* b, double* res,
             CZ_REAL* wk2, double* flop);

/* cz_Ffunc.h:48-58  <- cz_solver.f90:404-493.  One colour of red-black SOR, in place. */
void psor2sma_core_(CZ_REAL* p, int* sz, int* idx, int* g, CZ_REAL* cf, int* ip, int* color, CZ_REAL* omg,
                    CZ_REAL* b, double* res, double* flop);

/* cz_Ffunc.h:448-450 <- cz_blas.f90:112-149 (whole array incl. guide cells) */
void blas_clear_(CZ_REAL* x, int* sz, int* g);
/* cz_Ffunc.h:452-455 <- cz_blas.f90:159-195 */
void blas_copy_(CZ_REAL* dst, CZ_REAL* src, int* sz, int* g);
/* cz_Ffunc.h:463-470 <- cz_blas.f90:255-308   z = a*x + y */
void blas_triad_(CZ_REAL* z, CZ_REAL* x, CZ_REAL* y, CZ_REAL* a, int* sz, int* idx, int* g, double* flop);
/* cz_Ffunc.h:472-477 <- cz_blas.f90:320-373   *r = sum p*p  (r overwritten) */
void blas_dot1_(CZ_REAL* r, CZ_REAL* p, int* sz, int* idx, int* g, double* flop);
/* cz_Ffunc.h:479-485 <- cz_blas.f90:386-437   *r = sum p*q */
void blas_dot2_(CZ_REAL* r, CZ_REAL* p, CZ_REAL* q, int* sz, int* idx, int* g, double* flop);
/* cz_Ffunc.h:487-495 <- cz_blas.f90:452-502   p = r + beta*(p - omg*q) */
void blas_bicg_1_(CZ_REAL* p, CZ_REAL* r, CZ_REAL* q, CZ_REAL* beta, CZ_REAL* omg, int* sz, int* idx, int* g,
                  double* flop);
/* cz_Ffunc.h:497-505 <- cz_blas.f90:517-566   z = a*x + b*y + z */
void blas_bicg_2_(CZ_REAL* z, CZ_REAL* x, CZ_REAL* y, CZ_REAL* a, CZ_REAL* b, int* sz, int* idx, int* g,
                  double* flop);
/* cz_Ffunc.h:507-513 <- cz_blas.f90:579-644   ap = ss - dd*p */
void blas_calc_ax_(CZ_REAL* ap, CZ_REAL* p, int* sz, int* idx, int* g, CZ_REAL* cf, double* flop);
/* cz_Ffunc.h:515-522 <- cz_blas.f90:658-723   r = b - (ss - dd*p) */
void blas_calc_rk_(CZ_REAL* r, CZ_REAL* p, CZ_REAL* b, int* sz, int* idx, int* g, CZ_REAL* cf, double* flop);

/* MAF flavour: the same solvers on the metric-form Laplacian whose weights are recomputed per
 * point from 1-D coordinate arrays.  X, Y, Z (length N+4, Fortran X(-1:sz+2)) and tmp are HOST arrays exactly as in
 * the reference (cz_Evaluate.cpp:342-363 fills them on the host); p, b, wk2, pvt, r, ap are device arrays. */
/* cz_Ffunc.h:170-182 <- cz_maf.f90:131-285 */
void jacobi_maf_(CZ_REAL* p, int* sz, int* idx, int* g, CZ_REAL* X, CZ_REAL* Y, CZ_REAL* Z, CZ_REAL* omg, CZ_REAL* b, double* res,
                 CZ_REAL* wk2, CZ_REAL* tmp, double* flop);
/* cz_Ffunc.h:196-209 <- cz_maf.f90:301-438 */
void psor2sma_core_maf_(CZ_REAL* p, int* sz, int* idx, int* g, CZ_REAL* X, CZ_REAL* Y, CZ_REAL* Z, int* ip, int* color,
                        CZ_REAL* omg, CZ_REAL* b, double* res, CZ_REAL* tmp, double* flop);
/* cz_Ffunc.h:524-534 <- cz_blas.f90:738-832   r = (b + dd*p - sum w*p_nb) * pvt */
void calc_rk_maf_(CZ_REAL* r, CZ_REAL* p, CZ_REAL* b, int* sz, int* idx, int* g, CZ_REAL* X, CZ_REAL* Y, CZ_REAL* Z,
                  CZ_REAL* pvt, double* flop);
/* cz_Ffunc.h:536-545 <- cz_blas.f90:845-934   ap = (sum w*p_nb - dd*p) * pvt */
void calc_ax_maf_(CZ_REAL* ap, CZ_REAL* p, int* sz, int* idx, int* g, CZ_REAL* X, CZ_REAL* Y, CZ_REAL* Z, CZ_REAL* pvt,
                  double* flop);
/* Lexicographic point SOR.  Semantics = ONE thread of the reference (its PARALLEL DO makes the result
 * depend on the thread count); computed as a two-level hyperplane wavefront, bit-identical to the sequential loop. */
/* cz_Ffunc.h:38-46 <- cz_solver.f90:207-269 */
void psor_(CZ_REAL* p, int* sz, int* idx, int* g, CZ_REAL* cf, CZ_REAL* omg, CZ_REAL* b, double* res, double* flop);
/* cz_Ffunc.h:184-194 <- cz_maf.f90:23-112 */
void psor_maf_(CZ_REAL* p, int* sz, int* idx, int* g, CZ_REAL* X, CZ_REAL* Y, CZ_REAL* Z, CZ_REAL* omg, CZ_REAL* b, double* res,
               double* flop);
/* cz_Ffunc.h:547-553 <- cz_blas.f90:947-1039  pvt = 1 / max |row entries| */
void search_pivot_(CZ_REAL* pvt, int* sz, int* idx, int* g, CZ_REAL* X, CZ_REAL* Y, CZ_REAL* Z);

/* Line SOR by parallel cyclic reduction.  x, msk, rhs are device arrays; the six work arrays are
 * the reference's host scratch and are ignored (the line systems live in LDS).  Semantics = the reference's serial build
 * (its OpenMP form reads uninitialised private work arrays, see oracle/Makefile). */
/* cz_Ffunc.h:60-77 <- cz_solver.f90:497-662 */
void pcr_rb_(int* sz, int* idx, int* g, int* pn, int* ofst, int* color, CZ_REAL* x, CZ_REAL* msk, CZ_REAL* rhs, CZ_REAL* a,
             CZ_REAL* c, CZ_REAL* d, CZ_REAL* a1, CZ_REAL* c1, CZ_REAL* d1, CZ_REAL* omg, double* res, double* flop);
/* The other line-SOR variants: pn-2 PCR stages + 4x4 systems by Cramer's rule (pcr, pcr_esa, pcr_rb_esa) or pn-1 stages + 2x2
 * systems (pcr_j_esa); columns in lexicographic order in place (pcr, pcr_esa: computed diagonal by diagonal, bit-identical to
 * the sequential loop), one colour in place (pcr_rb_esa), or all from the old field through wrk (pcr_j_esa).  The work arrays
 * (a..d1, src) are the reference's scratch and are ignored; entries beyond a line are zeros ("ESA" semantics, also where the
 * reference indexes past its arrays, i.e. when n < 3/4 * 2^pn). */
/* cz_Ffunc.h:99-114 <- cz_solver.f90:666-878 */
void pcr_(int* sz, int* idx, int* g, int* pn, CZ_REAL* x, CZ_REAL* msk, CZ_REAL* rhs, CZ_REAL* a, CZ_REAL* c, CZ_REAL* d, CZ_REAL* a1,
          CZ_REAL* c1, CZ_REAL* d1, CZ_REAL* omg, double* res, double* flop);
/* cz_Ffunc.h:116-128 <- cz_solver.f90:883-1045 (lexicographic, pn-1 stages + 2x2 systems; its un-refreshed a(kst), c(ked) are +-0.0: fresh
 * zeros here, see oracle/cz_oracle.c) */
void pcr_eda_(int* sz, int* idx, int* g, int* pn, CZ_REAL* x, CZ_REAL* msk, CZ_REAL* rhs, CZ_REAL* a1, CZ_REAL* c1, CZ_REAL* d1, CZ_REAL* omg,
              double* res, double* flop);
/* cz_Ffunc.h:130-146 <- cz_solver.f90:1050-1257 */
void pcr_esa_(int* sz, int* idx, int* g, int* pn, int* s, CZ_REAL* x, CZ_REAL* msk, CZ_REAL* rhs, CZ_REAL* a, CZ_REAL* c, CZ_REAL* d,
              CZ_REAL* a1, CZ_REAL* c1, CZ_REAL* d1, CZ_REAL* omg, double* res, double* flop);
/* cz_Ffunc.h:79-97 <- cz_solver.f90:1261-1469 */
void pcr_rb_esa_(int* sz, int* idx, int* g, int* pn, int* ofst, int* color, int* s, CZ_REAL* x, CZ_REAL* msk, CZ_REAL* rhs, CZ_REAL* a,
                 CZ_REAL* c, CZ_REAL* d, CZ_REAL* a1, CZ_REAL* c1, CZ_REAL* d1, CZ_REAL* omg, double* res, double* flop);
/* cz_Ffunc.h:148-166 <- cz_solver.f90:1473-1676 */
void pcr_j_esa_(int* sz, int* idx, int* g, int* pn, int* s, CZ_REAL* x, CZ_REAL* msk, CZ_REAL* rhs, CZ_REAL* a, CZ_REAL* c, CZ_REAL* d,
                CZ_REAL* a1, CZ_REAL* c1, CZ_REAL* d1, CZ_REAL* src, CZ_REAL* wrk, CZ_REAL* omg, double* res, double* flop);
/* The MAF line solvers (cz_Ffunc.h:211-315 <- cz_maf.f90:442-1560): the tridiagonal coefficients come from the metrics of the 1-D
 * grids XX, YY, ZZ (host arrays as in jacobi_maf_), pn-1 PCR stages + 2x2 systems; one colour (pcr_rb_maf_, pcr_rb_esa_maf_) or
 * lexicographic order (pcr_maf_, pcr_eda_maf_, pcr_esa_maf_; computed diagonal by diagonal).  Work arrays and tmp are ignored. */
void pcr_rb_maf_(int* sz, int* idx, int* g, int* pn, int* ofst, int* color, CZ_REAL* x, CZ_REAL* msk, CZ_REAL* rhs, CZ_REAL* XX, CZ_REAL* YY,
                 CZ_REAL* ZZ, CZ_REAL* a, CZ_REAL* c, CZ_REAL* d, CZ_REAL* aw, CZ_REAL* cw, CZ_REAL* dw, CZ_REAL* omg, double* res, CZ_REAL* tmp,
                 double* flop);
void pcr_rb_esa_maf_(int* sz, int* idx, int* g, int* pn, int* ofst, int* color, int* s, CZ_REAL* x, CZ_REAL* msk, CZ_REAL* rhs, CZ_REAL* XX,
                     CZ_REAL* YY, CZ_REAL* ZZ, CZ_REAL* a, CZ_REAL* c, CZ_REAL* d, CZ_REAL* aw, CZ_REAL* cw, CZ_REAL* dw, CZ_REAL* omg, double* res,
                     CZ_REAL* tmp, double* flop);
void pcr_maf_(int* sz, int* idx, int* g, int* pn, CZ_REAL* x, CZ_REAL* msk, CZ_REAL* rhs, CZ_REAL* XX, CZ_REAL* YY, CZ_REAL* ZZ, CZ_REAL* a,
              CZ_REAL* c, CZ_REAL* d, CZ_REAL* aw, CZ_REAL* cw, CZ_REAL* dw, CZ_REAL* omg, double* res, CZ_REAL* tmp, double* flop);
void pcr_eda_maf_(int* sz, int* idx, int* g, int* pn, CZ_REAL* x, CZ_REAL* msk, CZ_REAL* rhs, CZ_REAL* XX, CZ_REAL* YY, CZ_REAL* ZZ, CZ_REAL* aw,
                  CZ_REAL* cw, CZ_REAL* dw, CZ_REAL* omg, double* res, CZ_REAL* tmp, double* flop);
void pcr_esa_maf_(int* sz, int* idx, int* g, int* pn, int* s, CZ_REAL* x, CZ_REAL* msk, CZ_REAL* rhs, CZ_REAL* XX, CZ_REAL* YY, CZ_REAL* ZZ,
                  CZ_REAL* a, CZ_REAL* c, CZ_REAL* d, CZ_REAL* aw, CZ_REAL* cw, CZ_REAL* dw, CZ_REAL* omg, double* res, CZ_REAL* tmp, double* flop);
/* cz_Ffunc.h:440-443 <- cz_blas.f90:24-104 */
void imask_k_(CZ_REAL* x, int* sz, int* idx, int* g);

/* ------------------------------------------------------------------------------------------------
 * Part 2 -- runtime
 * ---------------------------------------------------------------------------------------------- */
int czhip_real_bytes(void);              /* 4 or 8: which precision this library was built for */
const char* czhip_arch(void);            /* "gfx950" */
int czhip_init(int device);              /* bind the calling process to one GPU, create streams/workspace; 0 = ok.
                                            Any HIP failure anywhere prints a message and exit(1)s: the
                                            reference ABI has no error channel. */
void czhip_finalize(void);
CZ_REAL* czhip_alloc_s3d(const int* sz); /* czAllocR_S3D (cz.h:209-232): (NI+4)(NJ+4)(NK+4) zero-filled, on the device */
void czhip_free(void* dptr);
void czhip_h2d(void* dst_dev, const void* src_host, size_t bytes); /* synchronous */
void czhip_d2h(void* dst_host, const void* src_dev, size_t bytes); /* synchronous (drains the compute stream first) */
void czhip_sync(void);                   /* drain all library streams */
void* czhip_stream(void);                /* the hipStream_t every kernel of this library is launched on */

/* Tuning of the stencil kernels (threads per block, vectors per thread, planes per j-chunk, prefetch
 * depth).  0 keeps the current value.  Returns 0 if that instantiation exists. */
int czhip_set_tuning(int threads, int vec_per_thread, int planes_per_chunk, int prefetch);
void czhip_get_tuning(int* threads, int* vec_per_thread, int* planes_per_chunk, int* prefetch);

/* Per-launch HIP-event timing on the library's stream (bench.py's roofline leg): enable(1) starts a fresh
 * collection, enable(0) stops it; czhip_timing_read returns the number of launches recorded under `label`
 * ("jacobi", "rbsor", "calc_ax", "calc_rk", "reduce", "ewise", "dot") and their summed duration in ms. */
void czhip_timing(int enable);
int czhip_timing_read(const char* label, double* total_ms);

/* ------------------------------------------------------------------------------------------------
 * Part 3 -- asynchronous device-resident operations (stream-ordered, no host synchronisation)
 * ---------------------------------------------------------------------------------------------- */

/* One Jacobi sweep p_in -> p_out (ping-pong; p_out's non-inner elements are left untouched), sum dp^2
 * accumulated in double into res_dev[0] (device) when `accumulate` != 0, else stored.  If skip_flag_dev
 * is non-NULL and *skip_flag_dev != 0 on the device the sweep is a no-op (convergence reached earlier). */
void czhip_jacobi_async(const CZ_REAL* p_in, CZ_REAL* p_out, const CZ_REAL* b, const int* sz, const int* idx, int g,
                        const CZ_REAL* cf, CZ_REAL omg, double* res_dev, int accumulate, const int* skip_flag_dev);

/* One colour of RB-SOR in place; parity: points with (i+j+k+ofst+color) even relative to kst as in
 * cz_solver.f90:466. */
void czhip_rbsor_async(CZ_REAL* p, const CZ_REAL* b, const int* sz, const int* idx, int g, const CZ_REAL* cf,
                       int ofst, int color, CZ_REAL omg, double* res_dev, int accumulate, const int* skip_flag_dev);

/* The same sweeps with the convergence bookkeeping of czhip_check_async folded into the sweep kernel (performed by
 * its last workgroup): one launch per Jacobi iteration, two per RB-SOR iteration (pass the check arguments with the
 * second colour, accumulate = 1).  flag_dev doubles as the skip flag. */
void czhip_jacobi_checked_async(const CZ_REAL* p_in, CZ_REAL* p_out, const CZ_REAL* b, const int* sz, const int* idx, int g,
                                const CZ_REAL* cf, CZ_REAL omg, double* res_dev, double res_normal, double eps, int itr,
                                double* hist_dev, int* flag_dev, int* conv_itr_dev);
void czhip_rbsor_checked_async(CZ_REAL* p, const CZ_REAL* b, const int* sz, const int* idx, int g, const CZ_REAL* cf,
                               int ofst, int color, CZ_REAL omg, double* res_dev, int accumulate, double res_normal,
                               double eps, int itr, double* hist_dev, int* flag_dev, int* conv_itr_dev);

/* TWO Jacobi sweeps per pass over memory (temporal blocking; single-domain, g >= 2): u -> w = two applications of
 * cz_solver.f90:334-351, bit-identical to two czhip_jacobi_async calls, the intermediate field never leaves the chip.
 * res_dev[0] / res_dev[1] = sum dp^2 of the first / second sweep.  hist_dev != NULL adds the convergence bookkeeping for
 * iterations itr and itr+1 (in order; flag_dev doubles as skip flag).  If the FIRST sweep converges, flag is set with
 * conv_itr = itr and w holds time n+2: the caller re-runs one single sweep from u (never modified).  Returns 1 when
 * launched, 0 when the geometry is unsupported (caller falls back to single sweeps).
 * idx1 (NULL = idx): index range of the FIRST sweep; a decomposed run grows it by one layer across rank-internal faces
 * (the second sweep reads the first one's result on the ghost layer; two ghost layers are exchanged per pair).
 * skip_flag_dev is used when hist_dev is NULL (multi-rank runs do the bookkeeping after the all-reduce). */
int czhip_jacobi2_async(const CZ_REAL* u, CZ_REAL* w, const CZ_REAL* b, const int* sz, const int* idx, const int* idx1, int g,
                        const CZ_REAL* cf, CZ_REAL omg, double* res_dev, double res_normal, double eps, int itr, double* hist_dev,
                        int* flag_dev, int* conv_itr_dev, const int* skip_flag_dev);
/* One complete red-black SOR iteration (colour 0 then colour 1, cz_Poisson.cpp:205-209) in ONE pass over memory,
 * u -> w out of place (the caller ping-pongs like Jacobi): bit-identical to psor2sma_core_ colour 0 + colour 1.
 * res_dev[0] = the iteration's sum dp^2 over both colours; check arguments as above (one iteration). */
int czhip_rbsor2_async(const CZ_REAL* u, CZ_REAL* w, const CZ_REAL* b, const int* sz, const int* idx, const int* idx1, int g,
                       const CZ_REAL* cf, int ofst, CZ_REAL omg, double* res_dev, double res_normal, double eps, int itr,
                       double* hist_dev, int* flag_dev, int* conv_itr_dev, const int* skip_flag_dev);
/* TWO red-black SOR iterations (colour 0, 1, 0, 1: cz_Poisson.cpp:205-209 twice) in ONE pass over memory, u -> w out of place (single-domain
 * boxes, constant coefficients): bit-identical to four psor2sma_core_ calls.  res_dev[0], res_dev[1] = the sums dp^2 of iteration itr and
 * itr + 1; check arguments as for czhip_jacobi2_async (two iterations; a converged first iteration leaves conv_itr = itr and the caller
 * recomputes that iteration from u).  probe != 0: only says whether the launch would be taken.  Returns 0 when it is not (the caller then runs
 * czhip_rbsor2_async twice). */
int czhip_rbsor4_async(const CZ_REAL* u, CZ_REAL* w, const CZ_REAL* b, const int* sz, const int* idx, int g, const CZ_REAL* cf, int ofst,
                       CZ_REAL omg, double* res_dev, double res_normal, double eps, int itr, double* hist_dev, int* flag_dev,
                       int* conv_itr_dev, const int* skip_flag_dev, int probe);
/* ... its switches (measurements; negative = keep): 0 off / 1 on / 2 on also for small grids (where the preloaded one-iteration pass is faster
 * and 1 leaves the box to it), vectors per k window, planes per chunk (0 = chosen per launch) */
int czhip_set_rb4(int enable, int window, int planes);
/* The fused pass split the way a decomposed brick runs it (replaces the reference's "sweep, then Comm_S",
 * cz_Poisson.cpp:58-63): first the slabs two cells thick behind every face with nID[f] >= 0 (the cells the neighbours
 * receive), then the interior, so that the exchange can start after the first launch.  Same result as the unsplit op.
 * rb_ofst < 0: two Jacobi sweeps, res_dev[0..1]; rb_ofst >= 0: one red-black iteration (ofst), res_dev[0].
 * Returns 0 (nothing launched) when there is no internal face, the box is too thin or the geometry is unsupported. */
int czhip_pair_split_async(const CZ_REAL* u, CZ_REAL* w, const CZ_REAL* b, const int* sz, const int* idx, const int* idx1,
                           const int* nID, int g, const CZ_REAL* cf, CZ_REAL omg, int rb_ofst, double* res_dev);
/* First pair of a preconditioner solve whose start vector is identically zero (blas_clear_ + 8 sweeps, cz_Poisson.cpp:405-409):
 * u is neither cleared in memory nor read; u_shape only provides the array geometry/alignment.  Bit-identical to clearing u and
 * calling czhip_jacobi2_async. */
int czhip_jacobi2_from_zero_async(const CZ_REAL* u_shape, CZ_REAL* w, const CZ_REAL* b, const int* sz, const int* idx, const int* idx1,
                                  int g, const CZ_REAL* cf, CZ_REAL omg, double* res_dev);
/* The same with the right-hand side of the solve MADE on the way from the operands of the vector update that precedes a preconditioner
 * solve in BiCGSTAB -- op 1: b = a*x + y (blas_triad_, cz_blas.f90:297), op 2: b = x + a*(z - bb*y) (blas_bicg_1_, :490) -- and stored
 * to b_out (not one of x, y, z) for the later passes of the solve: update and first pass in one launch, same bits as the two calls.
 * op 0: b_out is read as the right-hand side.  rb_ofst < 0: two Jacobi sweeps (czhip_jacobi2_async); >= 0: one red-black iteration with
 * that colour offset (czhip_rbsor2_async).  probe != 0: only says whether the launch would be taken. */
int czhip_jacobi2_from_zero_made_async(const CZ_REAL* u_shape, CZ_REAL* w, CZ_REAL* b_out, int op, const CZ_REAL* x, const CZ_REAL* y,
                                       const CZ_REAL* z, CZ_REAL a, CZ_REAL bb, const int* sz, const int* idx, const int* idx1, int g,
                                       const CZ_REAL* cf, CZ_REAL omg, int rb_ofst, double* res_dev, int probe);
/* The same bookkeeping for a pair whose two sums were all-reduced first (decomposed runs). */
void czhip_check2_async(const double* res_dev, double res_normal, double eps, int itr, double* hist_dev, int* flag_dev,
                        int* conv_itr_dev);
/* The MAF flavour of the two calls above (cz_maf.f90:131-438): weights recomputed at every point from the host coordinate arrays X, Y, Z
 * (as in jacobi_maf_ / psor2sma_core_maf_).  rb_ofst < 0: two jacobi_maf sweeps, res_dev[0..1]; rb_ofst >= 0: one red-black iteration with
 * that ofst, res_dev[0].  Returns 1 if launched. */
int czhip_pair_maf_async(const CZ_REAL* u, CZ_REAL* w, const CZ_REAL* b, const int* sz, const int* idx, int g, const CZ_REAL* X,
                         const CZ_REAL* Y, const CZ_REAL* Z, CZ_REAL omg, int rb_ofst, double* res_dev);
/* Shape of the two-stage pass: threads per workgroup (512 | 1024; 0 / -1 keep, -2 = chosen per launch by the balance model), vectors per
 * thread (2), planes per chunk (0 = chosen per launch, -1 keep), enable (-1 keep).  Returns 0 if ok.  Every shape gives the same bits. */
int czhip_set_tuning2(int threads, int vec_per_thread, int planes_per_chunk, int enable);
/* line-SOR kernels (pcr*_): form 0 = one wave per line with the reference's arithmetic literally (a, c and d of the line reduced in LDS; every
 * variant but pcr_j_esa_; also the fall-back for lines whose coefficient table does not fit LDS), 1 = coefficient
 * table + right-hand side in LDS, 2 = table + right-hand side in registers (default); variant (form 1 only; form 2 has one measured shape per
 * case since round 3) = waves*10 + lines per wave, 0 = default; negative = keep.  All forms give the same bits. */
int czhip_set_pcr_mode(int form, int variant);
/* the lexicographic line SOR (pcr_, pcr_esa_, pcr_eda_; reference: cz_solver.f90:666-878, one thread walking j, i): one_launch 1 = the whole
 * sweep in one launch, rows of lines handed from workgroup to workgroup (default), 0 = one launch per diagonal i+j; groups of threads per
 * workgroup (0 = chosen per launch); rows per thread (1 | 2).  Negative = keep.  Every shape gives the same bits. */
int czhip_set_pcr_lex(int one_launch, int groups, int rows_per_thread);
/* Bound, in seconds, of every wait of one workgroup for another inside the one-launch sweep (default 2; negative = keep); returns the bound in
 * force.  If a wait runs out, every workgroup leaves and the residual of that sweep is NaN. */
double czhip_set_pcr_lex_timeout(double seconds);
/* the lexicographic point SOR (psor_, psor_maf_; reference: cz_solver.f90:207-269, one thread walking j, i, k): one_launch 1 = the whole sweep
 * in one launch (default), 0 = one launch per tile hyperplane; workgroups per CU of the former (0 = chosen by the launcher).  Negative = keep.
 * Same bits either way; the waits of the one-launch form are bounded by czhip_set_pcr_lex_timeout, a lost hand-off gives a NaN residual. */
int czhip_set_psor(int one_launch, int wg_per_cu);
/* ... and how many steps ahead of their use a column asks for the face values of the columns before it (4 | 8; 0 = chosen per launch).  Returns
 * the previous setting.  Measurement aid; same bits either way. */
int czhip_set_psor_ahead(int steps);
/* Launch limits of the one-launch sweep (test aid; negative = keep, 0 = chosen per launch): workgroups per CU, workgroups in all, lines per
 * hand-off ring between two rows (rounded up to a power of two).  The launcher's own ring size lets the sweep finish however few of its
 * workgroups the device keeps resident; a ring forced small with few workgroups cannot, and the sweep then ends as described above. */
int czhip_set_pcr_lex_limits(int wg_per_cu, int max_wg, int slots);
int czhip_use_t2(void);
/* The two-stage pass cuts long k rows into windows (a workgroup's LDS holds whole rows of its window only; cz_solver.f90:284-387 takes any
 * extent, and so does the pass since round 4): vectors (16 bytes) per window; 0 = whole rows wherever they fit, -1 = chosen per launch (default),
 * <= -2 = keep.  Returns the previous setting.  Results do not depend on it. */
int czhip_set_pair_window(int vectors);
/* Small grids (every workgroup of a pass resident at once): the pass requests all operands of a chunk before its first plane step instead of one
 * plane ahead; 1 = on (default), 0 = off, negative = keep.  Returns the previous setting.  Results do not depend on it. */
int czhip_set_pair_preload(int enable);
/* The reference's coefficients are c1 .. c6 = 1, dd = 6 (cz.h:169-172): where the six are exactly 1 the Jacobi pass and the two-iteration red-black
 * pass take a form without the six multiplications (x * 1 is x: same sum, same order).  1 = on (default), 0 = always the general form, negative =
 * keep.  Returns the previous setting.  Results do not depend on it. */
int czhip_set_unit_coef(int enable);
/* Every environment variable the library and the cz command line read (one table, cubez_amd/csrc/cz_config.h), one per line: NAME=value where set,
 * NAME (unset; default ...) otherwise; only_set != 0 lists the former only.  The string lives until the next call on the calling thread. */
const char* czhip_config_describe(int only_set);
/* Decomposed runs keep k CUs of every XCD free of the sweeps (CZ_COMM_CUS, default 2) so that RCCL's send/recv kernels run while an interior
 * sweep fills the chip -- through the launch geometry (the launches count those CUs out).  Measurement aid: put that reservation in force by
 * hand (the driver does it itself in decomposed runs and undoes it in single-domain ones at set-up); returns the reservation in force. */
int czhip_set_comm_cus(int k);
/* self-test: numerators (of 2^32) whose quotient by d in the two-stage pass differs from the IEEE division (expected 0); -1 = divisor not eligible */
long long czhip_selftest_fastdiv(CZ_REAL d);

/* Convergence bookkeeping on the device (cz_Poisson.cpp:67-77): res = sqrt(res_dev[0]*res_normal);
 * hist_dev[itr] = res; if (res < eps && !*flag) { *flag = 1; conv_itr_dev[0] = itr; }.  No-op when
 * already converged. */
void czhip_check_async(const double* res_dev, double res_normal, double eps, int itr, double* hist_dev,
                       int* flag_dev, int* conv_itr_dev);

/* ------------------------------------------------------------------------------------------------
 * Part 4 -- the restated driver (class CZ of cz.h / cz_Evaluate.cpp / cz_Poisson.cpp) behind a handle
 * ---------------------------------------------------------------------------------------------- */
typedef struct cz_handle cz_handle;

cz_handle* cz_create(void);
void cz_destroy(cz_handle*);
/* main.cpp:15-60 + CZ::Evaluate (cz_Evaluate.cpp:21-567): same argv as the reference CLI
 *   cz gsz_x gsz_y gsz_z solver ItrMax coef [precond] [gdv_x gdv_y gdv_z]
 * returns 1 on success, 0 on "Solver error" exactly like CZ::Evaluate. */
int cz_evaluate(cz_handle*, int argc, char** argv);
/* Split form used by bench.py / tests: setup (parse + allocate + boundary conditions), then solve. */
int cz_setup(cz_handle*, int argc, char** argv);
int cz_solve(cz_handle*);                      /* runs the selected solver to ItrMax / eps; returns Iter (0 = error) */
int cz_sweeps(cz_handle*, int n);              /* bench leg: n more iterations of the selected stationary solver with the
                                                  full per-iteration work (sweep + residual + convergence bookkeeping),
                                                  never stopping early; returns n */
int cz_result_iter(const cz_handle*);
double cz_result_res(const cz_handle*);
int cz_history(const cz_handle*, double* out, int cap); /* copies min(cap, n) residuals, returns n */
void cz_field(const cz_handle*, CZ_REAL* host_out);    /* D2H of the solution P, dense (NK+4,NI+4,NJ+4) */
void cz_local_size(const cz_handle*, int* size3, int* head3, int* nID6, int* inner6);
double cz_error_max(cz_handle*, int* loc3);   /* debug epilogue, cz_Evaluate.cpp:550-563 (host-side restatement) */
void cz_set_quiet(cz_handle*, int quiet);     /* suppress stdout / history file (tests, bench) */
double cz_last_solve_seconds(const cz_handle*);
/* What a (multi-GPU) run decided: what = 0 ranks, 1 every brick takes the fused pass, 2 shell slabs of this brick, 3 overlapped exchange,
 * 4 the last stationary solve ran its residual all-reduce + test one pass behind, 5 ranks of the RCCL communicator (ncclCommCount; 0 = LOCAL
 * test transport or single process), 6 CUs per XCD the sweeps leave to the exchange stream (CZ_COMM_CUS); the plan of the last stationary
 * solve: 7 kind of pass (0 single sweeps, 1 fused pass over the whole box, 2 fused pass as shell slabs + interior with the exchange
 * overlapped), 8 ghost layers exchanged per pass, 9 rotating field buffers; 10 vector updates of the last BiCGSTAB solve that were made inside
 * the first pair of the preconditioner solve they feed (czhip_jacobi2_from_zero_made_async); 11 passes of the last red-black SOR solve that made
 * two iterations each (czhip_rbsor4_async). */
int cz_info(const cz_handle*, int what);
double cz_kernel_ms(const cz_handle*, const char* label); /* HIP-event time of a labelled section, ms (avg per launch) */

void cz_set_debug(cz_handle*, int mode);      /* main.cpp:38-42: 1 = run the analytic-error epilogue in cz_evaluate */
void cz_set_profile(cz_handle*, int on);      /* cz_Evaluate.cpp:506-545: 1 = cz_evaluate writes profiling.txt (PMlib-style section report) */

/* ------------------------------------------------------------------------------------------------
 * Part 5 -- multi-GPU bootstrap (replaces MPI_Init / CBrick set-up, main.cpp:33-35, cz_Evaluate.cpp:103-159).
 * One process per GPU.  Rank 0 creates the RCCL unique id, the launcher (torch.distributed in bench.py, a shared
 * file for the `cz` binary) hands the bytes to every rank, every rank joins BEFORE cz_setup(); cz_setup then takes
 * rank / size from the communicator and decomposes the cube.  Nothing to call for a single-GPU run.
 * ---------------------------------------------------------------------------------------------- */
int cz_comm_unique_id_bytes(void);
int cz_comm_get_unique_id(char* out_bytes);
int cz_comm_bootstrap(int rank, int nranks, const char* id_bytes);
void cz_comm_shutdown(void);
int cz_comm_selftest(void); /* one-rank RCCL smoke test: init, all-reduce, grouped send/recv to self; 0 = ok */
/* Host-only decomposition helpers (no GPU needed): automatic division and the brick of one rank
 * (local size, 1-based global head index, neighbour table I-,I+,J-,J+,K-,K+ with -1 = physical boundary). */
void cz_comm_auto_division(int nproc, const int* G_size, int* G_div);
int cz_comm_decompose(const int* G_size, const int* G_div, int nproc, int rank, int* size, int* head, int* nID);
/* Test transport: n ranks as n host threads of one process on one GPU (device-to-device copies instead of RCCL). */
void* cz_comm_local_world(int n);
void cz_comm_local_world_free(void* world);
int cz_comm_bootstrap_local(void* world, int rank);

#ifdef __cplusplus
}
#endif
#endif /* CZ_HIP_H_ */
