/*
 * cz_oracle.c -- CPU restatement of the CubeZ hot-path grid kernels.
 *
 * TEST INFRASTRUCTURE ONLY.  This file is the parity oracle for the HIP path:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load it.  The product (cubez_amd/) never links, imports or calls it.
 *
 * Every function restates one Fortran subroutine of the reference
 * (/root/reference/src/cz_f90) with the same argument list, the same
 * operation order and the same floating-point association, so that -- built
 * without FP contraction and run with one thread -- its outputs are bit-for-bit
 * those of the reference (checked by tests/test_oracle_vs_ref.py against
 * oracle/_ref, the reference's own Fortran compiled with amdflang, and against
 * the fixtures in tests/golden/ that were generated from it).
 *
 * Parity status: PINNED (oracle/_ref + tests/golden).
 *
 * Layout (cz_solver.f90:29): every 3-D array is
 *   real p(1-g:sz(3)+g, 1-g:sz(1)+g, 1-g:sz(2)+g)      -- K fastest, then I, then J
 * sz = (NI, NJ, NK) without guide cells, g = guide width (2 in the reference).
 *
 * Precision is a compile-time switch exactly as in the reference
 * (cz_Define.h:28-37): -DCZ_REAL_IS_DOUBLE builds the FP64 variant.
 *
 * Extra output not in the reference: the *_w entry points additionally return
 * the residual / dot accumulated in double over the SAME per-point terms
 * (each term still rounded to REAL first).  The reference accumulates in REAL
 * sequentially (cz_solver.f90:294,348,384); that value moves with summation
 * order (SURVEY.md finding 3), so GPU-vs-oracle residual parity is stated on
 * the wide accumulation while the REAL one pins the oracle to the reference.
 */
#include <math.h>
#include <stdlib.h>
#include <stddef.h>

#ifdef CZ_REAL_IS_DOUBLE
typedef double REAL;
#define R_SIN sin
#define R_ASIN asin
#else
typedef float REAL;
#define R_SIN sinf
#define R_ASIN asinf
#endif

/* 1-based (k,i,j) -> linear index, lower bound 1-g on every axis */
#define IDX(k, i, j) ((size_t)((k) + g - 1) + (size_t)((i) + g - 1) * nk + (size_t)((j) + g - 1) * nk * ni)

#define UNPACK_SZ                          \
  const int g = *gp;                       \
  const size_t nk = (size_t)(sz[2] + 2 * g); \
  const size_t ni = (size_t)(sz[0] + 2 * g); \
  (void)ni

#define UNPACK_IDX                                   \
  const int ist = idx[0], ied = idx[1];              \
  const int jst = idx[2], jed = idx[3];              \
  const int kst = idx[4], ked = idx[5]

#define NPTS ((double)(ied - ist + 1) * (double)(jed - jst + 1) * (double)(ked - kst + 1))

int oracle_real_bytes(void) { return (int)sizeof(REAL); }

/* ---- bc_k : cz_solver.f90:22-191 ------------------------------------------------
 * Dirichlet faces where nID(face) < 0.  K faces first (sin*sin), then I faces,
 * then J faces (zero) -- so edges end up 0.  nID order: I-,I+,J-,J+,K-,K+
 * (cz_fparam.fi:10-16). */
void oracle_bc_k(const int* sz, const int* gp, REAL* p, const REAL* dh_p, const REAL* org, const int* nID) {
  UNPACK_SZ;
  const int ix = sz[0], jx = sz[1], kx = sz[2];
  const REAL dh = *dh_p;
  const REAL pi = (REAL)2.0 * R_ASIN((REAL)1.0); /* :36 */
  if (nID[4] < 0) {                              /* K_MINUS :41-64 */
    for (int j = 1; j <= jx; j++)
      for (int i = 1; i <= ix; i++) {
        REAL x = org[0] + dh * (REAL)(i - 1);
        REAL y = org[1] + dh * (REAL)(j - 1);
        p[IDX(1, i, j)] = R_SIN(pi * x) * R_SIN(pi * y);
      }
  }
  if (nID[5] < 0) { /* K_PLUS :67-90 */
    for (int j = 1; j <= jx; j++)
      for (int i = 1; i <= ix; i++) {
        REAL x = org[0] + dh * (REAL)(i - 1);
        REAL y = org[1] + dh * (REAL)(j - 1);
        p[IDX(kx, i, j)] = R_SIN(pi * x) * R_SIN(pi * y);
      }
  }
  if (nID[0] < 0) /* I_MINUS :93-113 */
    for (int k = 1; k <= kx; k++)
      for (int j = 1; j <= jx; j++) p[IDX(k, 1, j)] = (REAL)0.0;
  if (nID[1] < 0) /* I_PLUS :116-137 */
    for (int k = 1; k <= kx; k++)
      for (int j = 1; j <= jx; j++) p[IDX(k, ix, j)] = (REAL)0.0;
  if (nID[2] < 0) /* J_MINUS :140-161 */
    for (int k = 1; k <= kx; k++)
      for (int i = 1; i <= ix; i++) p[IDX(k, i, 1)] = (REAL)0.0;
  if (nID[3] < 0) /* J_PLUS :164-185 */
    for (int k = 1; k <= kx; k++)
      for (int i = 1; i <= ix; i++) p[IDX(k, i, jx)] = (REAL)0.0;
}

/* ---- jacobi : cz_solver.f90:284-387 ---------------------------------------------
 * wk2 = p + ((ss-b)/dd - p)*omg on the inner box, res1 += dp*dp (REAL), then
 * p <- wk2 on the inner box, res += dble(res1), flop += 18*npts. */
void oracle_jacobi_w(REAL* p, const int* sz, const int* idx, const int* gp, const REAL* cf, const REAL* omg_p,
                     const REAL* b, double* res, REAL* wk2, double* flop, double* res_wide) {
  UNPACK_SZ;
  UNPACK_IDX;
  const REAL c1 = cf[0], c2 = cf[1], c3 = cf[2], c4 = cf[3], c5 = cf[4], c6 = cf[5], dd = cf[6];
  const REAL omg = *omg_p;
  REAL res1 = (REAL)0.0;
  double resw = 0.0;
  *flop += 18.0 * NPTS; /* :315-318 */
#pragma omp parallel
  {
#pragma omp for schedule(static) collapse(2) reduction(+ : res1, resw)
    for (int j = jst; j <= jed; j++)
      for (int i = ist; i <= ied; i++)
        for (int k = kst; k <= ked; k++) { /* :334-351 */
          REAL pp = p[IDX(k, i, j)];
          REAL bb = b[IDX(k, i, j)];
          REAL ss = c1 * p[IDX(k, i + 1, j)] + c2 * p[IDX(k, i - 1, j)] + c3 * p[IDX(k, i, j + 1)] +
                    c4 * p[IDX(k, i, j - 1)] + c5 * p[IDX(k + 1, i, j)] + c6 * p[IDX(k - 1, i, j)];
          REAL dp = ((ss - bb) / dd - pp) * omg;
          REAL pn = pp + dp;
          wk2[IDX(k, i, j)] = pn;
          REAL d2 = dp * dp;
          res1 = res1 + d2;
          resw += (double)d2;
        }
    /* The reference has END DO NOWAIT here (:355), i.e. a data race with >1 thread
     * (SURVEY.md finding 2).  The oracle keeps the implicit barrier: it is the
     * single-thread semantics that are the contract. */
#pragma omp for schedule(static) collapse(2)
    for (int j = jst; j <= jed; j++)
      for (int i = ist; i <= ied; i++)
        for (int k = kst; k <= ked; k++) p[IDX(k, i, j)] = wk2[IDX(k, i, j)]; /* :369-375 */
  }
  *res = *res + (double)res1; /* :384 */
  if (res_wide) *res_wide += resw;
}

void oracle_jacobi(REAL* p, const int* sz, const int* idx, const int* gp, const REAL* cf, const REAL* omg,
                   const REAL* b, double* res, REAL* wk2, double* flop) {
  oracle_jacobi_w(p, sz, idx, gp, cf, omg, b, res, wk2, flop, NULL);
}

/* ---- psor2sma_core : cz_solver.f90:404-493 --------------------------------------
 * in-place update of the points k = kst+mod(i+j+ofst+color,2), ked, 2 (:419,466). */
void oracle_psor2sma_core_w(REAL* p, const int* sz, const int* idx, const int* gp, const REAL* cf,
                            const int* ofst, const int* color, const REAL* omg_p, const REAL* b, double* res,
                            double* flop, double* res_wide) {
  UNPACK_SZ;
  UNPACK_IDX;
  const int kp = *ofst + *color;
  const REAL c1 = cf[0], c2 = cf[1], c3 = cf[2], c4 = cf[3], c5 = cf[4], c6 = cf[5], dd = cf[6];
  const REAL omg = *omg_p;
  REAL res1 = (REAL)0.0;
  double resw = 0.0;
  *flop += 18.0 * 0.5 * NPTS; /* :438-441 */
#pragma omp parallel for schedule(static) collapse(2) reduction(+ : res1, resw)
  for (int j = jst; j <= jed; j++)
    for (int i = ist; i <= ied; i++)
      for (int k = kst + ((i + j + kp) % 2); k <= ked; k += 2) { /* :466 */
        REAL pp = p[IDX(k, i, j)];
        REAL bb = b[IDX(k, i, j)];
        REAL ss = c1 * p[IDX(k, i + 1, j)] + c2 * p[IDX(k, i - 1, j)] + c3 * p[IDX(k, i, j + 1)] +
                  c4 * p[IDX(k, i, j - 1)] + c5 * p[IDX(k + 1, i, j)] + c6 * p[IDX(k - 1, i, j)];
        REAL dp = ((ss - bb) / dd - pp) * omg;
        REAL pn = pp + dp;
        p[IDX(k, i, j)] = pn;
        REAL d2 = dp * dp;
        res1 = res1 + d2;
        resw += (double)d2;
      }
  *res = *res + (double)res1; /* :490 */
  if (res_wide) *res_wide += resw;
}

void oracle_psor2sma_core(REAL* p, const int* sz, const int* idx, const int* gp, const REAL* cf, const int* ofst,
                          const int* color, const REAL* omg, const REAL* b, double* res, double* flop) {
  oracle_psor2sma_core_w(p, sz, idx, gp, cf, ofst, color, omg, b, res, flop, NULL);
}

/* ---- blas_clear / blas_copy : cz_blas.f90:112-149 / 159-195 (full array incl. guide cells) */
void oracle_blas_clear(REAL* x, const int* sz, const int* gp) {
  const int g = *gp;
  const size_t n = (size_t)(sz[0] + 2 * g) * (size_t)(sz[1] + 2 * g) * (size_t)(sz[2] + 2 * g);
  for (size_t m = 0; m < n; m++) x[m] = (REAL)0.0;
}

void oracle_blas_copy(REAL* y, const REAL* x, const int* sz, const int* gp) {
  const int g = *gp;
  const size_t n = (size_t)(sz[0] + 2 * g) * (size_t)(sz[1] + 2 * g) * (size_t)(sz[2] + 2 * g);
  for (size_t m = 0; m < n; m++) y[m] = x[m];
}

/* ---- blas_triad : cz_blas.f90:255-308   z = a*x + y */
void oracle_blas_triad(REAL* z, const REAL* x, const REAL* y, const REAL* a_p, const int* sz, const int* idx,
                       const int* gp, double* flop) {
  UNPACK_SZ;
  UNPACK_IDX;
  const REAL a = *a_p;
  *flop += 2.0 * NPTS;
#pragma omp parallel for schedule(static) collapse(2)
  for (int j = jst; j <= jed; j++)
    for (int i = ist; i <= ied; i++)
      for (int k = kst; k <= ked; k++) z[IDX(k, i, j)] = a * x[IDX(k, i, j)] + y[IDX(k, i, j)]; /* :297 */
}

/* Summation order of the two dot products (TEST INFRASTRUCTURE for the BiCGSTAB tolerance, VERDICT r1 "weak" 1): 0 = the reference's
 * order (j ascending, cz_blas.f90:355-366), 1 = the same products summed with j DEscending.  Nothing else changes, so a solve run with
 * order 1 shows how far the reference's own Krylov path moves when only the rounding of its REAL-accumulated dots is permuted. */
static int g_dot_order = 0;
void oracle_set_dot_order(int order) { g_dot_order = order; }
int oracle_get_dot_order(void) { return g_dot_order; }

/* ---- blas_dot1 : cz_blas.f90:320-373   r = sum p*p (REAL accumulator, overwritten) */
void oracle_blas_dot1_w(REAL* r, const REAL* p, const int* sz, const int* idx, const int* gp, double* flop,
                        double* r_wide) {
  UNPACK_SZ;
  UNPACK_IDX;
  REAL acc = (REAL)0.0;
  double accw = 0.0;
  *flop += 2.0 * NPTS;
  const int rev = g_dot_order;
#pragma omp parallel for schedule(static) collapse(2) reduction(+ : acc, accw)
  for (int jj = jst; jj <= jed; jj++)
    for (int i = ist; i <= ied; i++)
      for (int k = kst; k <= ked; k++) {
        const int j = rev ? jst + jed - jj : jj;
        REAL q = p[IDX(k, i, j)];
        REAL t = q * q;
        acc = acc + t; /* :361-362 */
        accw += (double)t;
      }
  *r = acc;
  if (r_wide) *r_wide = accw;
}

void oracle_blas_dot1(REAL* r, const REAL* p, const int* sz, const int* idx, const int* gp, double* flop) {
  oracle_blas_dot1_w(r, p, sz, idx, gp, flop, NULL);
}

/* ---- blas_dot2 : cz_blas.f90:386-437   r = sum p*q */
void oracle_blas_dot2_w(REAL* r, const REAL* p, const REAL* q, const int* sz, const int* idx, const int* gp,
                        double* flop, double* r_wide) {
  UNPACK_SZ;
  UNPACK_IDX;
  REAL acc = (REAL)0.0;
  double accw = 0.0;
  *flop += 2.0 * NPTS;
  const int rev = g_dot_order;
#pragma omp parallel for schedule(static) collapse(2) reduction(+ : acc, accw)
  for (int jj = jst; jj <= jed; jj++)
    for (int i = ist; i <= ied; i++)
      for (int k = kst; k <= ked; k++) {
        const int j = rev ? jst + jed - jj : jj;
        REAL t = p[IDX(k, i, j)] * q[IDX(k, i, j)];
        acc = acc + t; /* :426 */
        accw += (double)t;
      }
  *r = acc;
  if (r_wide) *r_wide = accw;
}

void oracle_blas_dot2(REAL* r, const REAL* p, const REAL* q, const int* sz, const int* idx, const int* gp,
                      double* flop) {
  oracle_blas_dot2_w(r, p, q, sz, idx, gp, flop, NULL);
}

/* ---- blas_bicg_1 : cz_blas.f90:452-502   p = r + beta*(p - omg*q) */
void oracle_blas_bicg_1(REAL* p, const REAL* r, const REAL* q, const REAL* beta_p, const REAL* omg_p,
                        const int* sz, const int* idx, const int* gp, double* flop) {
  UNPACK_SZ;
  UNPACK_IDX;
  const REAL beta = *beta_p, omg = *omg_p;
  *flop += 4.0 * NPTS;
#pragma omp parallel for schedule(static) collapse(2)
  for (int j = jst; j <= jed; j++)
    for (int i = ist; i <= ied; i++)
      for (int k = kst; k <= ked; k++)
        p[IDX(k, i, j)] = r[IDX(k, i, j)] + beta * (p[IDX(k, i, j)] - omg * q[IDX(k, i, j)]); /* :490 */
}

/* ---- blas_bicg_2 : cz_blas.f90:517-566   z = a*x + b*y + z */
void oracle_blas_bicg_2(REAL* z, const REAL* x, const REAL* y, const REAL* a_p, const REAL* b_p, const int* sz,
                        const int* idx, const int* gp, double* flop) {
  UNPACK_SZ;
  UNPACK_IDX;
  const REAL a = *a_p, b = *b_p;
  *flop += 4.0 * NPTS;
#pragma omp parallel for schedule(static) collapse(2)
  for (int j = jst; j <= jed; j++)
    for (int i = ist; i <= ied; i++)
      for (int k = kst; k <= ked; k++)
        z[IDX(k, i, j)] = a * x[IDX(k, i, j)] + b * y[IDX(k, i, j)] + z[IDX(k, i, j)]; /* :554 */
}

/* ---- blas_calc_ax : cz_blas.f90:579-644   ap = ss - dd*p */
void oracle_blas_calc_ax(REAL* ap, const REAL* p, const int* sz, const int* idx, const int* gp, const REAL* cf,
                         double* flop) {
  UNPACK_SZ;
  UNPACK_IDX;
  const REAL c1 = cf[0], c2 = cf[1], c3 = cf[2], c4 = cf[3], c5 = cf[4], c6 = cf[5], dd = cf[6];
  *flop += 13.0 * NPTS;
#pragma omp parallel for schedule(static) collapse(2)
  for (int j = jst; j <= jed; j++)
    for (int i = ist; i <= ied; i++)
      for (int k = kst; k <= ked; k++) {
        REAL ss = c1 * p[IDX(k, i + 1, j)] + c2 * p[IDX(k, i - 1, j)] + c3 * p[IDX(k, i, j + 1)] +
                  c4 * p[IDX(k, i, j - 1)] + c5 * p[IDX(k + 1, i, j)] + c6 * p[IDX(k - 1, i, j)];
        ap[IDX(k, i, j)] = (ss - dd * p[IDX(k, i, j)]); /* :626-632 */
      }
}

/* ---- blas_calc_rk : cz_blas.f90:658-723   r = b - (ss - dd*p) */
void oracle_blas_calc_rk(REAL* r, const REAL* p, const REAL* b, const int* sz, const int* idx, const int* gp,
                         const REAL* cf, double* flop) {
  UNPACK_SZ;
  UNPACK_IDX;
  const REAL c1 = cf[0], c2 = cf[1], c3 = cf[2], c4 = cf[3], c5 = cf[4], c6 = cf[5], dd = cf[6];
  *flop += 14.0 * NPTS;
#pragma omp parallel for schedule(static) collapse(2)
  for (int j = jst; j <= jed; j++)
    for (int i = ist; i <= ied; i++)
      for (int k = kst; k <= ked; k++) {
        REAL ss = c1 * p[IDX(k, i + 1, j)] + c2 * p[IDX(k, i - 1, j)] + c3 * p[IDX(k, i, j + 1)] +
                  c4 * p[IDX(k, i, j - 1)] + c5 * p[IDX(k + 1, i, j)] + c6 * p[IDX(k - 1, i, j)];
        r[IDX(k, i, j)] = (b[IDX(k, i, j)] - (ss - dd * p[IDX(k, i, j)])); /* :705-711 */
      }
}

/* ---- exact_t / err_t : cz_utility.f90:52-82 / 86-129 (the reference's only known-answer check) */
void oracle_exact_t(const int* sz, const int* gp, REAL* e, const REAL* dh_p, const REAL* org) {
  UNPACK_SZ;
  const int ix = sz[0], jx = sz[1], kx = sz[2];
  const REAL dh = *dh_p;
#ifdef CZ_REAL_IS_DOUBLE
  const REAL r2 = sqrt(2.0);
#else
  const REAL r2 = sqrtf(2.0f);
#endif
  const REAL pi = (REAL)2.0 * R_ASIN((REAL)1.0);
  for (int j = 1; j <= jx; j++)
    for (int i = 1; i <= ix; i++)
      for (int k = 1; k <= kx; k++) {
        REAL x = org[0] + dh * (REAL)(i - 1);
        REAL y = org[1] + dh * (REAL)(j - 1);
        REAL z = org[2] + dh * (REAL)(k - 1);
#ifdef CZ_REAL_IS_DOUBLE
        e[IDX(k, i, j)] = sin(pi * x) * sin(pi * y) / sinh(r2 * pi) * (sinh(r2 * pi * z) - sinh(r2 * pi * (z - 1.0)));
#else
        e[IDX(k, i, j)] =
            sinf(pi * x) * sinf(pi * y) / sinhf(r2 * pi) * (sinhf(r2 * pi * z) - sinhf(r2 * pi * (z - 1.0f)));
#endif
      }
}

void oracle_err_t(const int* sz, const int* idx, const int* gp, double* d, const REAL* p, REAL* e, int* loc) {
  UNPACK_SZ;
  UNPACK_IDX;
  *d = 0.0;
  loc[0] = loc[1] = loc[2] = -1;
  for (int j = jst; j <= jed; j++)
    for (int i = ist; i <= ied; i++)
      for (int k = kst; k <= ked; k++) {
        REAL r = p[IDX(k, i, j)] - e[IDX(k, i, j)];
        e[IDX(k, i, j)] = r;
        double q = fabs((double)r);
        if (*d < q) {
          *d = q;
          loc[0] = i;
          loc[1] = j;
          loc[2] = k;
        }
      }
}

/* ==================================================================================================
 * MAF flavour: the same solvers on a metric-form Laplacian whose six neighbour
 * weights and diagonal are recomputed at every point from the 1-D coordinate arrays X, Y, Z
 * (cz_maf.f90, cz_blas.f90:738-1039).  X(-1:sz(1)+2) etc.: X(i) is x[i+1] in C.
 * ================================================================================================== */
#define XC(i) x[(i) + 1]
#define YC(j) y[(j) + 1]
#define ZC(k) z[(k) + 1]

/* cz_maf.f90:193-221 (identical block in every MAF kernel) */
#define MAF_COEF                                                 \
  const REAL XG = (REAL)0.5 * (XC(i + 1) - XC(i - 1));           \
  const REAL YE = (REAL)0.5 * (YC(j + 1) - YC(j - 1));           \
  const REAL ZT = (REAL)0.5 * (ZC(k + 1) - ZC(k - 1));           \
  const REAL XGG = XC(i + 1) - (REAL)2.0 * XC(i) + XC(i - 1);    \
  const REAL YEE = YC(j + 1) - (REAL)2.0 * YC(j) + YC(j - 1);    \
  const REAL ZTT = ZC(k + 1) - (REAL)2.0 * ZC(k) + ZC(k - 1);    \
  const REAL YJA = XG * YE * ZT;                                 \
  const REAL YJAI = (REAL)1.0 / YJA;                             \
  const REAL GX = YE * ZT * YJAI;                                \
  const REAL EY = XG * ZT * YJAI;                                \
  const REAL TZ = XG * YE * YJAI;                                \
  const REAL C1 = GX * GX, C2 = EY * EY, C3 = TZ * TZ;           \
  const REAL C7 = -XGG * C1 * GX;                                \
  const REAL C8 = -YEE * C2 * EY;                                \
  const REAL C9 = -ZTT * C3 * TZ

/* ---- jacobi_maf : cz_maf.f90:131-285 */
void oracle_jacobi_maf_w(REAL* p, const int* sz, const int* idx, const int* gp, const REAL* x, const REAL* y, const REAL* z,
                         const REAL* omg_p, const REAL* b, double* res, REAL* wk2, REAL* tmp, double* flop, double* res_wide) {
  UNPACK_SZ;
  UNPACK_IDX;
  const REAL omg = *omg_p;
  REAL res1 = (REAL)0.0;
  double resw = 0.0;
  for (int k = -1; k <= sz[2] + 2; k++) tmp[k + 1] = (REAL)0.0; /* :155 */
  *flop += 66.0 * NPTS;
#pragma omp parallel
  {
#pragma omp for schedule(static) collapse(2) reduction(+ : res1, resw)
    for (int j = jst; j <= jed; j++)
      for (int i = ist; i <= ied; i++)
        for (int k = kst; k <= ked; k++) {
          const REAL bb = b[IDX(k, i, j)];
          const REAL pp = p[IDX(k, i, j)];
          MAF_COEF;
          const REAL dd = (REAL)2.0 * (C1 + C2 + C3);
          const REAL rp = (C1 + (REAL)0.5 * C7) * p[IDX(k, i + 1, j)] + (C1 - (REAL)0.5 * C7) * p[IDX(k, i - 1, j)] +
                          (C2 + (REAL)0.5 * C8) * p[IDX(k, i, j + 1)] + (C2 - (REAL)0.5 * C8) * p[IDX(k, i, j - 1)] +
                          (C3 + (REAL)0.5 * C9) * p[IDX(k + 1, i, j)] + (C3 - (REAL)0.5 * C9) * p[IDX(k - 1, i, j)] + bb;
          const REAL dp = (rp / dd - pp) * omg;
          wk2[IDX(k, i, j)] = pp + dp;
          const REAL d2 = dp * dp;
          res1 = res1 + d2; /* non-_SVR build, :233 */
          resw += (double)d2;
        }
#pragma omp for schedule(static) collapse(2)
    for (int j = jst; j <= jed; j++)
      for (int i = ist; i <= ied; i++)
        for (int k = kst; k <= ked; k++) p[IDX(k, i, j)] = wk2[IDX(k, i, j)];
  }
  *res = *res + (double)res1;
  if (res_wide) *res_wide += resw;
}

void oracle_jacobi_maf(REAL* p, const int* sz, const int* idx, const int* gp, const REAL* x, const REAL* y, const REAL* z,
                       const REAL* omg, const REAL* b, double* res, REAL* wk2, REAL* tmp, double* flop) {
  oracle_jacobi_maf_w(p, sz, idx, gp, x, y, z, omg, b, res, wk2, tmp, flop, NULL);
}

/* ---- psor2sma_core_maf : cz_maf.f90:301-438 */
void oracle_psor2sma_core_maf_w(REAL* p, const int* sz, const int* idx, const int* gp, const REAL* x, const REAL* y,
                                const REAL* z, const int* ofst, const int* color, const REAL* omg_p, const REAL* b,
                                double* res, REAL* tmp, double* flop, double* res_wide) {
  UNPACK_SZ;
  UNPACK_IDX;
  (void)tmp;
  const int kp = *ofst + *color;
  const REAL omg = *omg_p;
  REAL res1 = (REAL)0.0;
  double resw = 0.0;
  *flop += 66.0 * 0.5 * NPTS;
#pragma omp parallel for schedule(static) collapse(2) reduction(+ : res1, resw)
  for (int j = jst; j <= jed; j++)
    for (int i = ist; i <= ied; i++)
      for (int k = kst + ((i + j + kp) % 2); k <= ked; k += 2) {
        const REAL pp = p[IDX(k, i, j)];
        const REAL bb = b[IDX(k, i, j)];
        MAF_COEF;
        const REAL dd = (REAL)2.0 * (C1 + C2 + C3);
        const REAL rp = (C1 + (REAL)0.5 * C7) * p[IDX(k, i + 1, j)] + (C1 - (REAL)0.5 * C7) * p[IDX(k, i - 1, j)] +
                        (C2 + (REAL)0.5 * C8) * p[IDX(k, i, j + 1)] + (C2 - (REAL)0.5 * C8) * p[IDX(k, i, j - 1)] +
                        (C3 + (REAL)0.5 * C9) * p[IDX(k + 1, i, j)] + (C3 - (REAL)0.5 * C9) * p[IDX(k - 1, i, j)] + bb;
        const REAL dp = (rp / dd - pp) * omg;
        p[IDX(k, i, j)] = pp + dp;
        const REAL d2 = dp * dp;
        res1 = res1 + d2;
        resw += (double)d2;
      }
  *res = *res + (double)res1;
  if (res_wide) *res_wide += resw;
}

void oracle_psor2sma_core_maf(REAL* p, const int* sz, const int* idx, const int* gp, const REAL* x, const REAL* y,
                              const REAL* z, const int* ofst, const int* color, const REAL* omg, const REAL* b, double* res,
                              REAL* tmp, double* flop) {
  oracle_psor2sma_core_maf_w(p, sz, idx, gp, x, y, z, ofst, color, omg, b, res, tmp, flop, NULL);
}

/* ---- the other line-SOR variants of cz_solver.f90.  They share one column solver and differ in
 *   - the number of PCR stages and the direct solve that ends them: pn-1 stages + 2x2 systems, or pn-2 stages + 4x4 systems
 *     by Cramer's rule;
 *   - how entries beyond the line are reached: clamped to the entries kst-1 / ked+1 (pcr) or read from arrays extended
 *     by s zero entries on both sides ("ESA").  Both read +0.0 there, so one restatement with wide zero pads serves both;
 *   - the order of the columns: lexicographic in place (pcr, pcr_esa: a column sees the new values of its i-1 and j-1
 *     neighbours), one checkerboard colour in place (pcr_rb_esa), or all columns from the old field (pcr_j_esa).
 * The ESA variants index a, c, d beyond their declared extent unless n >= 3/4 * 2^pn (e.g. d(k+3*sq) at :1388); this
 * restatement reads zeros there, which is what the reference reads when the memory behind its arrays is zero. */
static void pcr_column(const REAL* xs, REAL* xd, const REAL* msk, const REAL* rhs, size_t c0, size_t rowlen, size_t plane, int n, int pn,
                       int final4, REAL omg, REAL* W /* 6 arrays of n + 2P */, int P, REAL* res1, double* resw) {
  const REAL r = (REAL)1.0 / (REAL)6.0;
  const int LD = n + 2 * P;
  REAL *a = W + P, *c = a + LD, *d = c + LD, *a1 = d + LD, *c1 = a1 + LD, *d1 = c1 + LD; /* element k = 0..n-1 <-> kst+k */
  for (int k = -P; k < n + P; k++) a[k] = c[k] = d[k] = a1[k] = c1[k] = d1[k] = (REAL)0.0;
  for (int k = 1; k < n; k++) a[k] = -r;     /* :725-739, :1309-1329, :1101-1118 */
  for (int k = 0; k < n - 1; k++) c[k] = -r;
  for (int k = 0; k < n; k++) {              /* :741-757 */
    const size_t e = c0 + (size_t)k;
    d[k] = ((xs[e - plane] + xs[e + plane] + xs[e - rowlen] + xs[e + rowlen] - rhs[e]) * r) * msk[e];
  }
  d[0] = (d[0] + xs[c0 - 1] * r) * msk[c0];
  d[n - 1] = (d[n - 1] + xs[c0 + (size_t)n] * r) * msk[c0 + (size_t)n - 1];
  const int nstage = final4 ? pn - 2 : pn - 1;
  for (int p = 1; p <= nstage; p++) {        /* :759-783 */
    const int s = 1 << (p - 1);
    for (int k = 0; k < n; k++) {
      const REAL ap = a[k], cp = c[k];
      const REAL e = (REAL)1.0 / ((REAL)1.0 - ap * c[k - s] - cp * a[k + s]);
      a1[k] = -e * ap * a[k - s];
      c1[k] = -e * cp * c[k + s];
      d1[k] = e * (d[k] - ap * d[k - s] - cp * d[k + s]);
    }
    for (int k = 0; k < n; k++) a[k] = a1[k], c[k] = c1[k], d[k] = d1[k];
  }
  if (final4) {                              /* :787-842 */
    const int s = 1 << (pn - 2);
    for (int k = 0; k < s; k++) {
      const int kl = k + s, km = k + 2 * s, kr = k + 3 * s;
      const REAL cc1 = c[k], cc2 = c[kl], cc3 = c[km], aa2 = a[kl], aa3 = a[km], aa4 = a[kr];
      const REAL dd1 = d[k], dd2 = d[kl], dd3 = d[km], dd4 = d[kr];
      const REAL inv_detA = (REAL)1.0 / ((REAL)1.0 - aa4 * cc3 - aa3 * cc2 - aa2 * cc1 * ((REAL)1.0 - cc3 * aa4));
      const REAL detA1 = -cc3 * (aa4 * dd1 + cc1 * cc2 * dd4 - aa4 * cc1 * dd2) + dd1 + cc1 * cc2 * dd3 - aa3 * cc2 * dd1 - cc1 * dd2;
      const REAL detA2 = dd2 + cc2 * cc3 * dd4 - aa4 * cc3 * dd2 - cc2 * dd3 - aa2 * (dd1 - aa4 * cc3 * dd1);
      const REAL detA3 = dd3 - cc3 * dd4 - aa3 * dd2 - aa2 * (cc1 * dd3 - cc1 * cc3 * dd4 - aa3 * dd1);
      const REAL detA4 = dd4 + aa3 * aa4 * dd2 - aa4 * dd3 - aa3 * cc2 * dd4 - aa2 * (cc1 * dd4 + aa3 * aa4 * dd1 - aa4 * cc1 * dd3);
      d1[k] = detA1 * inv_detA;
      d1[kl] = detA2 * inv_detA;
      d1[km] = detA3 * inv_detA;
      d1[kr] = detA4 * inv_detA;
    }
  } else {                                   /* :1602-1617 */
    const int s = 1 << (pn - 1);
    for (int k = 0; k < s; k++) {
      const REAL cc1 = c[k], aa2 = a[k + s], f1 = d[k], f2 = d[k + s];
      const REAL jj = (REAL)1.0 / ((REAL)1.0 - aa2 * cc1);
      d1[k] = (f1 - cc1 * f2) * jj;
      d1[k + s] = (f2 - aa2 * f1) * jj;
    }
  }
  for (int k = 0; k < n; k++) {              /* :854-863 */
    const size_t e = c0 + (size_t)k;
    const REAL pp = xs[e];
    const REAL dp = (d1[k] - pp) * omg * msk[e];
    xd[e] = pp + dp;
    const REAL d2 = dp * dp;
    *res1 = *res1 + d2;
    *resw += (double)d2;
  }
}

/* order: 0 = lexicographic in place, 1 = one colour in place, 2 = all columns from the old field (result through wrk) */
static void pcr_sweep(const int* sz, const int* idx, const int* gp, int pn, int order, int color, int final4, REAL* x, const REAL* msk,
                      const REAL* rhs, REAL* wrk, REAL omg, double* res, double* res_wide) {
  UNPACK_SZ;
  UNPACK_IDX;
  const int n = ked - kst + 1, P = 1 << (pn > 1 ? pn : 1);
  REAL* W = (REAL*)malloc((size_t)6 * (n + 2 * P) * sizeof(REAL));
  REAL res1 = (REAL)0.0;
  double resw = 0.0;
  for (int j = jst; j <= jed; j++)
    for (int i = ist; i <= ied; i++) {
      if (order == 1 && (i + j) % 2 != color) continue;
      pcr_column(x, order == 2 ? wrk : x, msk, rhs, IDX(kst, i, j), nk, nk * ni, n, pn, final4, omg, W, P, &res1, &resw);
    }
  if (order == 2)
    for (int j = jst; j <= jed; j++)
      for (int i = ist; i <= ied; i++)
        for (int k = kst; k <= ked; k++) x[IDX(k, i, j)] = wrk[IDX(k, i, j)]; /* :1655-1663 */
  free(W);
  *res = *res + (double)res1;
  if (res_wide) *res_wide += resw;
}

/* the same sweeps with the additional double-precision accumulation of sum dp^2 (what the GPU's residual is compared with) */
void oracle_pcr_sweep_w(const int* sz, const int* idx, const int* gp, const int* pn, const int* order, const int* color, const int* final4,
                        REAL* x, const REAL* msk, const REAL* rhs, REAL* wrk, const REAL* omg, double* res, double* res_wide) {
  pcr_sweep(sz, idx, gp, *pn, *order, *color, *final4, x, msk, rhs, wrk, *omg, res, res_wide);
}

#define PCR_FLOP(stages, fin) \
  ((double)((jed - jst + 1) * (ied - ist + 1)) * ((ked - kst + 1) * 6.0 + (ked - kst + 1) * (double)(stages)*14.0 + (fin) + (ked - kst + 1) * 6.0 + 6.0))

/* pcr : cz_solver.f90:666-878 */
void oracle_pcr(const int* sz, const int* idx, const int* gp, const int* pn, REAL* x, const REAL* msk, const REAL* rhs, REAL* a, REAL* c,
                REAL* d, REAL* a1, REAL* c1, REAL* d1, const REAL* omg, double* res, double* flop) {
  UNPACK_IDX;
  (void)a, (void)c, (void)d, (void)a1, (void)c1, (void)d1;
  *flop += PCR_FLOP(*pn - 2, (double)(1 << (*pn - 2)) * 74.0); /* :689-696 */
  pcr_sweep(sz, idx, gp, *pn, 0, 0, 1, x, msk, rhs, NULL, *omg, res, NULL);
}

/* pcr_eda : cz_solver.f90:883-1045.  Lexicographic order, pn-1 stages + 2x2 systems, arrays extended by zeros.  The routine does
 * not refresh a(kst) and c(ked) (:932,:936 are commented out): they keep the previous column's last-stage values, which are
 * always +-0.0 -- the result can only differ from this restatement (fresh +0.0) in the sign of a zero when d(kst) or d(ked)
 * is exactly -0.0.  It also indexes past its allocation unless n >= 3/4 * 2^pn (:985-995); zeros are read here. */
void oracle_pcr_eda(const int* sz, const int* idx, const int* gp, const int* pn, REAL* x, const REAL* msk, const REAL* rhs, REAL* a1, REAL* c1,
                    REAL* d1, const REAL* omg, double* res, double* flop) {
  UNPACK_IDX;
  (void)a1, (void)c1, (void)d1;
  *flop += PCR_FLOP(*pn - 1, (double)(1 << (*pn - 1)) * 9.0); /* :908-915 */
  pcr_sweep(sz, idx, gp, *pn, 0, 0, 0, x, msk, rhs, NULL, *omg, res, NULL);
}

/* pcr_esa : cz_solver.f90:1050-1257 */
void oracle_pcr_esa(const int* sz, const int* idx, const int* gp, const int* pn, const int* s, REAL* x, const REAL* msk, const REAL* rhs,
                    REAL* a, REAL* c, REAL* d, REAL* a1, REAL* c1, REAL* d1, const REAL* omg, double* res, double* flop) {
  UNPACK_IDX;
  (void)s, (void)a, (void)c, (void)d, (void)a1, (void)c1, (void)d1;
  *flop += PCR_FLOP(*pn - 2, (double)(1 << (*pn - 2)) * 78.0); /* :1078-1085 */
  pcr_sweep(sz, idx, gp, *pn, 0, 0, 1, x, msk, rhs, NULL, *omg, res, NULL);
}

/* pcr_rb_esa : cz_solver.f90:1261-1469 */
void oracle_pcr_rb_esa(const int* sz, const int* idx, const int* gp, const int* pn, const int* ofst, const int* color, const int* s, REAL* x,
                       const REAL* msk, const REAL* rhs, REAL* a, REAL* c, REAL* d, REAL* a1, REAL* c1, REAL* d1, const REAL* omg,
                       double* res, double* flop) {
  UNPACK_IDX;
  (void)ofst, (void)s, (void)a, (void)c, (void)d, (void)a1, (void)c1, (void)d1;
  *flop += PCR_FLOP(*pn - 2, (double)(1 << (*pn - 2)) * 78.0) * 0.5; /* :1291-1299 */
  pcr_sweep(sz, idx, gp, *pn, 1, *color, 1, x, msk, rhs, NULL, *omg, res, NULL);
}

/* pcr_j_esa : cz_solver.f90:1473-1676 (src is the reference's scratch for the source term) */
void oracle_pcr_j_esa(const int* sz, const int* idx, const int* gp, const int* pn, const int* s, REAL* x, const REAL* msk, const REAL* rhs,
                      REAL* a, REAL* c, REAL* d, REAL* a1, REAL* c1, REAL* d1, REAL* src, REAL* wrk, const REAL* omg, double* res,
                      double* flop) {
  UNPACK_IDX;
  (void)s, (void)a, (void)c, (void)d, (void)a1, (void)c1, (void)d1, (void)src;
  *flop += PCR_FLOP(*pn - 1, (double)(1 << (*pn - 1)) * 9.0); /* :1499-1506 */
  pcr_sweep(sz, idx, gp, *pn, 2, 0, 0, x, msk, rhs, wrk, *omg, res, NULL);
}

/* ---- the line-SOR kernels of the MAF flavour, cz_maf.f90:442-1560.  The five routines share one column solver: the
 * tridiagonal coefficients come from the metrics of the 1-D grids (they differ from line to line through C1(i) + C2(j)),
 * pn-1 PCR stages, 2x2 systems, relaxation; entries beyond a line are +0.0 (clamped index in pcr_rb_maf / pcr_maf, zero
 * pads in the _eda / _esa forms).  Columns: one colour (pcr_rb_maf, pcr_rb_esa_maf) or lexicographic (pcr_maf, pcr_eda_maf,
 * pcr_esa_maf), in place.  (tmp is the per-k partial-sum array of the reference's _SVR build, unused otherwise.) */
static void pcr_maf_column(REAL* x, const REAL* msk, const REAL* rhs, const REAL* XX, const REAL* YY, const REAL* ZZ, int i, int j, int kst,
                           size_t c0, size_t rowlen, size_t plane, int n, int pn, REAL omg, REAL* W, int P, REAL* res1, double* resw) {
  const int LD = n + 2 * P;
  REAL *a = W + P, *c = a + LD, *d = c + LD, *aw = d + LD, *cw = aw + LD, *dw = cw + LD; /* element k = 0..n-1 <-> kst+k */
  for (int k = -P; k < n + P; k++) a[k] = c[k] = d[k] = aw[k] = cw[k] = dw[k] = (REAL)0.0;
#define XG(ii) XX[(ii) + 1] /* X(-1:sz+2) */
  const REAL GX = (REAL)2.0 / (XG(i + 1) - XG(i - 1));
  const REAL EY = (REAL)2.0 / (YY[j + 1 + 1] - YY[j - 1 + 1]);
  const REAL C1 = GX * GX, C2 = EY * EY;
  const REAL C7 = -(XG(i + 1) - (REAL)2.0 * XG(i) + XG(i - 1)) * C1 * GX;
  const REAL C8 = -(YY[j + 1 + 1] - (REAL)2.0 * YY[j + 1] + YY[j - 1 + 1]) * C2 * EY;
  const REAL dd1 = C1 + (REAL)0.5 * C7, dd2 = C1 - (REAL)0.5 * C7, cc1 = C2 + (REAL)0.5 * C8, cc2 = C2 - (REAL)0.5 * C8;
#undef XG
  for (int k = 0; k < n; k++) { /* :501-510 */
    const int kz = kst + k;
    const REAL f1 = ZZ[kz + 1 + 1], f2 = ZZ[kz - 1 + 1];
    const REAL TZ = (REAL)2.0 / (f1 - f2);
    const REAL ZTT = f1 - (REAL)2.0 * ZZ[kz + 1] + f2;
    const REAL f3 = TZ * TZ;
    aw[k] = f3;
    cw[k] = -ZTT * f3 * TZ;
    dw[k] = (REAL)0.5 / (C1 + C2 + f3);
  }
  a[0] = (REAL)0.0; /* :513-527 */
  c[0] = -(aw[0] + (REAL)0.5 * cw[0]) * dw[0];
  for (int k = 1; k < n - 1; k++) {
    const REAL f1 = aw[k], f2 = cw[k], aa3 = dw[k];
    a[k] = -(f1 - (REAL)0.5 * f2) * aa3;
    c[k] = -(f1 + (REAL)0.5 * f2) * aa3;
  }
  a[n - 1] = -(aw[n - 1] - (REAL)0.5 * cw[n - 1]) * dw[n - 1];
  c[n - 1] = (REAL)0.0;
  for (int k = 0; k < n; k++) { /* :530-538 */
    const size_t e = c0 + (size_t)k;
    d[k] = (dd1 * x[e + rowlen] + dd2 * x[e - rowlen] + cc1 * x[e + plane] + cc2 * x[e - plane] - rhs[e]) * dw[k] * msk[e];
  }
  d[0] = (d[0] + (aw[0] - (REAL)0.5 * cw[0]) * dw[0] * x[c0 - 1]) * msk[c0]; /* :545-546 */
  d[n - 1] = (d[n - 1] + (aw[n - 1] + (REAL)0.5 * cw[n - 1]) * dw[n - 1] * x[c0 + (size_t)n]) * msk[c0 + (size_t)n - 1];
  for (int p = 1; p <= pn - 1; p++) { /* :551-574 */
    const int s = 1 << (p - 1);
    for (int k = 0; k < n; k++) {
      const REAL ap = a[k], cp = c[k];
      const REAL e = (REAL)1.0 / ((REAL)1.0 - ap * c[k - s] - cp * a[k + s]);
      aw[k] = -e * ap * a[k - s];
      cw[k] = -e * cp * c[k + s];
      dw[k] = e * (d[k] - ap * d[k - s] - cp * d[k + s]);
    }
    for (int k = 0; k < n; k++) a[k] = aw[k], c[k] = cw[k], d[k] = dw[k];
  }
  {
    const int s = 1 << (pn - 1); /* :578-596 */
    for (int k = 0; k < s; k++) {
      const REAL c1_ = c[k], aa2 = a[k + s], f1 = d[k], f2 = d[k + s];
      const REAL jj = (REAL)1.0 / ((REAL)1.0 - aa2 * c1_);
      dw[k] = (f1 - c1_ * f2) * jj;
      dw[k + s] = (f2 - aa2 * f1) * jj;
    }
  }
  for (int k = 0; k < n; k++) { /* :626-637 */
    const size_t e = c0 + (size_t)k;
    const REAL pp = x[e];
    const REAL dp = (dw[k] - pp) * omg * msk[e];
    x[e] = pp + dp;
    const REAL d2 = dp * dp;
    *res1 = *res1 + d2;
    *resw += (double)d2;
  }
}

/* order 0: lexicographic, 1: one colour */
static void pcr_maf_sweep(const int* sz, const int* idx, const int* gp, int pn, int order, int color, REAL* x, const REAL* msk, const REAL* rhs,
                          const REAL* XX, const REAL* YY, const REAL* ZZ, REAL omg, double* res, double* res_wide) {
  UNPACK_SZ;
  UNPACK_IDX;
  const int n = ked - kst + 1, P = 1 << (pn > 1 ? pn : 1);
  REAL* W = (REAL*)malloc((size_t)6 * (n + 2 * P) * sizeof(REAL));
  REAL res1 = (REAL)0.0;
  double resw = 0.0;
  for (int j = jst; j <= jed; j++)
    for (int i = ist; i <= ied; i++) {
      if (order == 1 && (i + j) % 2 != color) continue;
      pcr_maf_column(x, msk, rhs, XX, YY, ZZ, i, j, kst, IDX(kst, i, j), nk, nk * ni, n, pn, omg, W, P, &res1, &resw);
    }
  free(W);
  *res = *res + (double)res1;
  if (res_wide) *res_wide += resw;
}

#define PCR_MAF_FLOP(fin) \
  ((double)((jed - jst + 1) * (ied - ist + 1)) * ((24.0 + 3.0 * 2.0 + 12.0) + (ked - kst + 1) * (11.0 + 10.0) + (ked - kst - 1) * 6.0 + \
                                                   (ked - kst + 1) * (double)(*pn - 1) * 16.0 + (double)(1 << (*pn - 1)) * (fin) + (ked - kst + 1) * 6.0))

void oracle_pcr_maf_sweep_w(const int* sz, const int* idx, const int* gp, const int* pn, const int* order, const int* color, REAL* x,
                            const REAL* msk, const REAL* rhs, const REAL* XX, const REAL* YY, const REAL* ZZ, const REAL* omg, double* res,
                            double* res_wide) {
  pcr_maf_sweep(sz, idx, gp, *pn, *order, *color, x, msk, rhs, XX, YY, ZZ, *omg, res, res_wide);
}

/* pcr_rb_maf : cz_maf.f90:442-668 */
void oracle_pcr_rb_maf(const int* sz, const int* idx, const int* gp, const int* pn, const int* ofst, const int* color, REAL* x, const REAL* msk,
                       const REAL* rhs, const REAL* XX, const REAL* YY, const REAL* ZZ, REAL* a, REAL* c, REAL* d, REAL* aw, REAL* cw, REAL* dw,
                       const REAL* omg, double* res, REAL* tmp, double* flop) {
  UNPACK_IDX;
  (void)ofst, (void)a, (void)c, (void)d, (void)aw, (void)cw, (void)dw, (void)tmp;
  *flop += PCR_MAF_FLOP(11.0) * 0.5;
  pcr_maf_sweep(sz, idx, gp, *pn, 1, *color, x, msk, rhs, XX, YY, ZZ, *omg, res, NULL);
}
/* pcr_rb_esa_maf : cz_maf.f90:1339-1560 */
void oracle_pcr_rb_esa_maf(const int* sz, const int* idx, const int* gp, const int* pn, const int* ofst, const int* color, const int* s, REAL* x,
                           const REAL* msk, const REAL* rhs, const REAL* XX, const REAL* YY, const REAL* ZZ, REAL* a, REAL* c, REAL* d, REAL* aw,
                           REAL* cw, REAL* dw, const REAL* omg, double* res, REAL* tmp, double* flop) {
  UNPACK_IDX;
  (void)ofst, (void)s, (void)a, (void)c, (void)d, (void)aw, (void)cw, (void)dw, (void)tmp;
  *flop += PCR_MAF_FLOP(11.0) * 0.5;
  pcr_maf_sweep(sz, idx, gp, *pn, 1, *color, x, msk, rhs, XX, YY, ZZ, *omg, res, NULL);
}
/* pcr_maf : cz_maf.f90:672-892 */
void oracle_pcr_maf(const int* sz, const int* idx, const int* gp, const int* pn, REAL* x, const REAL* msk, const REAL* rhs, const REAL* XX,
                    const REAL* YY, const REAL* ZZ, REAL* a, REAL* c, REAL* d, REAL* aw, REAL* cw, REAL* dw, const REAL* omg, double* res,
                    REAL* tmp, double* flop) {
  UNPACK_IDX;
  (void)a, (void)c, (void)d, (void)aw, (void)cw, (void)dw, (void)tmp;
  *flop += PCR_MAF_FLOP(11.0);
  pcr_maf_sweep(sz, idx, gp, *pn, 0, 0, x, msk, rhs, XX, YY, ZZ, *omg, res, NULL);
}
/* pcr_eda_maf : cz_maf.f90:896-1113 */
void oracle_pcr_eda_maf(const int* sz, const int* idx, const int* gp, const int* pn, REAL* x, const REAL* msk, const REAL* rhs, const REAL* XX,
                        const REAL* YY, const REAL* ZZ, REAL* aw, REAL* cw, REAL* dw, const REAL* omg, double* res, REAL* tmp, double* flop) {
  UNPACK_IDX;
  (void)aw, (void)cw, (void)dw, (void)tmp;
  *flop += PCR_MAF_FLOP(9.0);
  pcr_maf_sweep(sz, idx, gp, *pn, 0, 0, x, msk, rhs, XX, YY, ZZ, *omg, res, NULL);
}
/* pcr_esa_maf : cz_maf.f90:1117-1335 */
void oracle_pcr_esa_maf(const int* sz, const int* idx, const int* gp, const int* pn, const int* s, REAL* x, const REAL* msk, const REAL* rhs,
                        const REAL* XX, const REAL* YY, const REAL* ZZ, REAL* a, REAL* c, REAL* d, REAL* aw, REAL* cw, REAL* dw, const REAL* omg,
                        double* res, REAL* tmp, double* flop) {
  UNPACK_IDX;
  (void)s, (void)a, (void)c, (void)d, (void)aw, (void)cw, (void)dw, (void)tmp;
  *flop += PCR_MAF_FLOP(9.0);
  pcr_maf_sweep(sz, idx, gp, *pn, 0, 0, x, msk, rhs, XX, YY, ZZ, *omg, res, NULL);
}

/* ---- psor : cz_solver.f90:207-269.  Lexicographic in-place SOR (j outer, i, k inner): every update sees the new values of
 * its k-1, i-1, j-1 neighbours and the old ones of k+1, i+1, j+1.  SERIAL semantics: the reference's PARALLEL DO makes the
 * result depend on the thread count; this is what one thread computes. */
void oracle_psor_w(REAL* p, const int* sz, const int* idx, const int* gp, const REAL* cf, const REAL* omg_p, const REAL* b,
                   double* res, double* flop, double* res_wide) {
  UNPACK_SZ;
  UNPACK_IDX;
  const REAL c1 = cf[0], c2 = cf[1], c3 = cf[2], c4 = cf[3], c5 = cf[4], c6 = cf[5], dd = cf[6];
  const REAL omg = *omg_p;
  REAL res1 = (REAL)0.0;
  double resw = 0.0;
  *flop += 18.0 * NPTS;
  for (int j = jst; j <= jed; j++)
    for (int i = ist; i <= ied; i++)
      for (int k = kst; k <= ked; k++) {
        const REAL pp = p[IDX(k, i, j)];
        const REAL bb = b[IDX(k, i, j)];
        const REAL ss = c1 * p[IDX(k, i + 1, j)] + c2 * p[IDX(k, i - 1, j)] + c3 * p[IDX(k, i, j + 1)] + c4 * p[IDX(k, i, j - 1)] +
                        c5 * p[IDX(k + 1, i, j)] + c6 * p[IDX(k - 1, i, j)];
        const REAL dp = ((ss - bb) / dd - pp) * omg;
        p[IDX(k, i, j)] = pp + dp;
        const REAL d2 = dp * dp;
        res1 = res1 + d2;
        resw += (double)d2;
      }
  *res = *res + (double)res1;
  if (res_wide) *res_wide += resw;
}

void oracle_psor(REAL* p, const int* sz, const int* idx, const int* gp, const REAL* cf, const REAL* omg, const REAL* b, double* res,
                 double* flop) {
  oracle_psor_w(p, sz, idx, gp, cf, omg, b, res, flop, NULL);
}

/* ---- psor_maf : cz_maf.f90:23-112, same ordering */
void oracle_psor_maf_w(REAL* p, const int* sz, const int* idx, const int* gp, const REAL* x, const REAL* y, const REAL* z,
                       const REAL* omg_p, const REAL* b, double* res, double* flop, double* res_wide) {
  UNPACK_SZ;
  UNPACK_IDX;
  const REAL omg = *omg_p;
  REAL res1 = (REAL)0.0;
  double resw = 0.0;
  *flop += 66.0 * NPTS;
  for (int j = jst; j <= jed; j++)
    for (int i = ist; i <= ied; i++)
      for (int k = kst; k <= ked; k++) {
        const REAL bb = b[IDX(k, i, j)];
        const REAL pp = p[IDX(k, i, j)];
        MAF_COEF;
        const REAL dd = (REAL)2.0 * (C1 + C2 + C3);
        const REAL rp = (C1 + (REAL)0.5 * C7) * p[IDX(k, i + 1, j)] + (C1 - (REAL)0.5 * C7) * p[IDX(k, i - 1, j)] +
                        (C2 + (REAL)0.5 * C8) * p[IDX(k, i, j + 1)] + (C2 - (REAL)0.5 * C8) * p[IDX(k, i, j - 1)] +
                        (C3 + (REAL)0.5 * C9) * p[IDX(k + 1, i, j)] + (C3 - (REAL)0.5 * C9) * p[IDX(k - 1, i, j)] + bb;
        const REAL dp = (rp / dd - pp) * omg;
        p[IDX(k, i, j)] = pp + dp;
        const REAL d2 = dp * dp;
        res1 = res1 + d2;
        resw += (double)d2;
      }
  *res = *res + (double)res1;
  if (res_wide) *res_wide += resw;
}

void oracle_psor_maf(REAL* p, const int* sz, const int* idx, const int* gp, const REAL* x, const REAL* y, const REAL* z,
                     const REAL* omg, const REAL* b, double* res, double* flop) {
  oracle_psor_maf_w(p, sz, idx, gp, x, y, z, omg, b, res, flop, NULL);
}

/* ---- calc_rk_maf : cz_blas.f90:738-832   r = (b + dd*p - sum w*p_nb) * pvt */
void oracle_calc_rk_maf(REAL* r, const REAL* p, const REAL* b, const int* sz, const int* idx, const int* gp, const REAL* x,
                        const REAL* y, const REAL* z, const REAL* pvt, double* flop) {
  UNPACK_SZ;
  UNPACK_IDX;
  *flop += 63.0 * NPTS; /* cz_blas.f90:762-765 */
#pragma omp parallel for schedule(static) collapse(2)
  for (int j = jst; j <= jed; j++)
    for (int i = ist; i <= ied; i++)
      for (int k = kst; k <= ked; k++) {
        MAF_COEF;
        r[IDX(k, i, j)] = (b[IDX(k, i, j)] + (REAL)2.0 * (C1 + C2 + C3) * p[IDX(k, i, j)] -
                           (C1 + (REAL)0.5 * C7) * p[IDX(k, i + 1, j)] - (C1 - (REAL)0.5 * C7) * p[IDX(k, i - 1, j)] -
                           (C2 + (REAL)0.5 * C8) * p[IDX(k, i, j + 1)] - (C2 - (REAL)0.5 * C8) * p[IDX(k, i, j - 1)] -
                           (C3 + (REAL)0.5 * C9) * p[IDX(k + 1, i, j)] - (C3 - (REAL)0.5 * C9) * p[IDX(k - 1, i, j)]) *
                          pvt[IDX(k, i, j)];
      }
}

/* ---- calc_ax_maf : cz_blas.f90:845-934   ap = (sum w*p_nb - dd*p) * pvt */
void oracle_calc_ax_maf(REAL* ap, const REAL* p, const int* sz, const int* idx, const int* gp, const REAL* x, const REAL* y,
                        const REAL* z, const REAL* pvt, double* flop) {
  UNPACK_SZ;
  UNPACK_IDX;
  *flop += 63.0 * NPTS; /* :868-871 */
#pragma omp parallel for schedule(static) collapse(2)
  for (int j = jst; j <= jed; j++)
    for (int i = ist; i <= ied; i++)
      for (int k = kst; k <= ked; k++) {
        MAF_COEF;
        ap[IDX(k, i, j)] = ((C1 + (REAL)0.5 * C7) * p[IDX(k, i + 1, j)] + (C1 - (REAL)0.5 * C7) * p[IDX(k, i - 1, j)] +
                            (C2 + (REAL)0.5 * C8) * p[IDX(k, i, j + 1)] + (C2 - (REAL)0.5 * C8) * p[IDX(k, i, j - 1)] +
                            (C3 + (REAL)0.5 * C9) * p[IDX(k + 1, i, j)] + (C3 - (REAL)0.5 * C9) * p[IDX(k - 1, i, j)] -
                            (REAL)2.0 * (C1 + C2 + C3) * p[IDX(k, i, j)]) *
                           pvt[IDX(k, i, j)];
      }
}

/* ---- search_pivot : cz_blas.f90:947-1039   pvt = 1 / max |row entries| */
void oracle_search_pivot(REAL* pvt, const int* sz, const int* idx, const int* gp, const REAL* x, const REAL* y, const REAL* z) {
  UNPACK_SZ;
  UNPACK_IDX;
#ifdef CZ_REAL_IS_DOUBLE
#define R_ABS fabs
#define R_MAX fmax
#else
#define R_ABS fabsf
#define R_MAX fmaxf
#endif
  for (int j = jst; j <= jed; j++)
    for (int i = ist; i <= ied; i++)
      for (int k = kst; k <= ked; k++) {
        MAF_COEF;
        const REAL s1 = R_ABS(C1 + (REAL)0.5 * C7), s2 = R_ABS(C1 - (REAL)0.5 * C7);
        const REAL s3 = R_ABS(C2 + (REAL)0.5 * C8), s4 = R_ABS(C2 - (REAL)0.5 * C8);
        const REAL s5 = R_ABS(C3 + (REAL)0.5 * C9), s6 = R_ABS(C3 - (REAL)0.5 * C9);
        const REAL s7 = R_ABS((REAL)2.0 * (C1 + C2 + C3));
        const REAL ss = R_MAX(R_MAX(R_MAX(R_MAX(R_MAX(R_MAX(s1, s2), s3), s4), s5), s6), s7);
        pvt[IDX(k, i, j)] = (REAL)1.0 / ss;
      }
}

/* ==================================================================================================
 * Line SOR by parallel cyclic reduction: pcr_rb, cz_solver.f90:497-662.
 * For every (i,j) column of one checkerboard colour (mod(i+j,2) == color; `ofst` is not used by the reference) the
 * tridiagonal system along k (a = c = -1/6, unit diagonal) is reduced by pn-1 PCR stages of stride 2^(p-1), the
 * remaining 2x2 systems of stride 2^(pn-1) are inverted directly, and x is relaxed in place.
 * Work arrays a, c, d, a1, c1, d1 are (-1:sz(3)+2); entries outside kst..ked are never written and stay 0 (the driver
 * allocates them zero-filled, cz_Evaluate.cpp:257-262).
 * ================================================================================================== */
#define W1(arr, k) arr[(k) + 1]

/* imask_k : cz_blas.f90:24-104   1 on the inner box, 0 elsewhere */
void oracle_imask_k(REAL* x, const int* sz, const int* idx, const int* gp) {
  UNPACK_SZ;
  UNPACK_IDX;
  const size_t n = (size_t)(sz[0] + 2 * g) * (size_t)(sz[1] + 2 * g) * (size_t)(sz[2] + 2 * g);
  for (size_t m = 0; m < n; m++) x[m] = (REAL)0.0;
  for (int j = jst; j <= jed; j++)
    for (int i = ist; i <= ied; i++)
      for (int k = kst; k <= ked; k++) x[IDX(k, i, j)] = (REAL)1.0;
}

void oracle_pcr_rb(const int* sz, const int* idx, const int* gp, const int* pn_p, const int* ofst, const int* color_p, REAL* x,
                   const REAL* msk, const REAL* rhs, REAL* a, REAL* c, REAL* d, REAL* a1, REAL* c1, REAL* d1,
                   const REAL* omg_p, double* res, double* flop) {
  UNPACK_SZ;
  UNPACK_IDX;
  (void)ofst;
  const int pn = *pn_p, color = *color_p;
  const REAL omg = *omg_p;
  const REAL r = (REAL)1.0 / (REAL)6.0;
  *flop += (double)((jed - jst + 1) * (ied - ist + 1)) *
           ((ked - kst + 1) * 6.0 + (ked - kst + 1) * (pn - 1) * 14.0 + (double)(1 << (pn - 1)) * 9.0 + (ked - kst + 1) * 6.0 + 6.0) *
           0.5; /* :523-531 */
  for (int j = jst; j <= jed; j++)
    for (int i = ist; i <= ied; i++) {
      if ((i + j) % 2 != color) continue; /* :540 */
      W1(a, kst) = (REAL)0.0;
      for (int k = kst + 1; k <= ked; k++) W1(a, k) = -r;
      for (int k = kst; k <= ked - 1; k++) W1(c, k) = -r;
      W1(c, ked) = (REAL)0.0;
      for (int k = kst; k <= ked; k++) /* :558-564 */
        W1(d, k) = ((x[IDX(k, i, j - 1)] + x[IDX(k, i, j + 1)] + x[IDX(k, i - 1, j)] + x[IDX(k, i + 1, j)] - rhs[IDX(k, i, j)]) * r) *
                   msk[IDX(k, i, j)];
      W1(d, kst) = (W1(d, kst) + x[IDX(kst - 1, i, j)] * r) * msk[IDX(kst, i, j)]; /* :567-568 */
      W1(d, ked) = (W1(d, ked) + x[IDX(ked + 1, i, j)] * r) * msk[IDX(ked, i, j)];
      for (int p = 1; p <= pn - 1; p++) { /* :572-595 */
        const int s = 1 << (p - 1);
        for (int k = kst; k <= ked; k++) {
          const int kl = (k - s > kst - 1) ? k - s : kst - 1;
          const int kr = (k + s < ked + 1) ? k + s : ked + 1;
          const REAL ap = W1(a, k), cp = W1(c, k);
          const REAL e = (REAL)1.0 / ((REAL)1.0 - ap * W1(c, kl) - cp * W1(a, kr));
          W1(a1, k) = -e * ap * W1(a, kl);
          W1(c1, k) = -e * cp * W1(c, kr);
          W1(d1, k) = e * (W1(d, k) - ap * W1(d, kl) - cp * W1(d, kr));
        }
        for (int k = kst; k <= ked; k++) {
          W1(a, k) = W1(a1, k);
          W1(c, k) = W1(c1, k);
          W1(d, k) = W1(d1, k);
        }
      }
      { /* :599-616 final 2x2 systems */
        const int s = 1 << (pn - 1);
        for (int k = kst; k <= kst + s - 1; k++) {
          const int kr = (k + s < ked + 1) ? k + s : ked + 1;
          const REAL cc1 = W1(c, k), aa2 = W1(a, kr), f1 = W1(d, k), f2 = W1(d, kr);
          const REAL jj = (REAL)1.0 / ((REAL)1.0 - aa2 * cc1);
          const REAL dd1 = (f1 - cc1 * f2) * jj;
          const REAL dd2 = (f2 - aa2 * f1) * jj;
          W1(d1, k) = dd1;
          W1(d1, kr) = dd2;
        }
      }
      for (int k = kst; k <= ked; k++) { /* :626-633 */
        const REAL pp = x[IDX(k, i, j)];
        const REAL dp = (W1(d1, k) - pp) * omg * msk[IDX(k, i, j)];
        x[IDX(k, i, j)] = pp + dp;
        *res = *res + (double)(dp * dp);
      }
    }
}
