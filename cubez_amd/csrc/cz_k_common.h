// cz_k_common.h -- part of cz_kernels.hip (ONE translation unit per precision; this file is included inside its anonymous
// namespace and is not a stand-alone header): vector types, loads/stores, block reduction, MAF weights.
template <int V>
struct alignas(sizeof(REAL) * V) Vec {
  REAL v[V];
};

struct Coef {
  REAL c1, c2, c3, c4, c5, c6, dd, omg;
};
// The six off-diagonal terms of a point (cz_solver.f90:335-341, 436-441).  UNIT: c1 .. c6 are all exactly 1 -- what CZ::CZ sets and never
// changes (cz.h:169-172) -- and x * 1 is x: the same sum in the same order without the six multiplications, hence the same bits; the
// launchers take this form only after comparing the six coefficients with 1 (coef_is_unit, cz_h_launch.h).
template <int UNIT>
__device__ __forceinline__ REAL offdiag_sum(const Coef& c, REAL ip, REAL im, REAL pn, REAL pm, REAL kp1, REAL km1) {
  if (UNIT) return ip + im + pn + pm + kp1 + km1;
  return c.c1 * ip + c.c2 * im + c.c3 * pn + c.c4 * pm + c.c5 * kp1 + c.c6 * km1;
}

// Geometry of one launch, all in PADDED 0-based indices (kk = k+g-1, ...).
struct Geom {
  int nip;          // padded rows per plane = NI+2g
  int R;            // vectors per k-row = (NK+2g)/V
  long long PSV;    // vectors per plane = R*(NI+2g)
  int kk0, kk1;     // inner k range (inclusive)
  int jj0, jj1;     // inner j range (inclusive)
  long long F0;     // first vector of the update range inside a plane = ii0*R
  long long Fend;   // one past the last vector of the update range    = (ii1+1)*R
  int nseg;         // segments per plane
  int TJ;           // planes per chunk
  int S;            // vectors per segment
  // rows as R vectors from a vector boundary each (R = ceil(nkp / V)): in memory a row is nkp elements, the last vector of a row partial where
  // nkp % V != 0 and a vector only REAL-aligned (see Geom2, cz_k_pair.h)
  int nkp = 0;            // elements per row in memory
  long long PSE = 0;      // elements per plane in memory
  int jlast = 0;          // the array's last plane: element offsets into it are clamped to
  long long last_eo = 0;  // ... this, so that its last vector is not read beyond the array (values clamped away are never used)
};

enum { MODE_JACOBI = 0, MODE_RB = 1, MODE_AX = 2, MODE_RK = 3 };

// In-kernel finalisation of the residual: the workgroup that arrives last sums the per-workgroup partials in a fixed
// order (deterministic) and, if asked, performs the convergence bookkeeping of cz_Poisson.cpp:67-77 -- no extra
// launches per sweep.  Hand-off follows cdna_hip_programming.md Guideline 16 in its write-through form: sc1 store of
// the partial -> s_waitcnt vmcnt(0) -> agent-scope ticket add; the last arriver reads every partial with sc1 loads.
struct Fin {
  double* dst = nullptr;  // device double receiving sum dp^2 (nullptr: leave the partials for a separate reduce launch)
  int accumulate = 0;     // dst += instead of dst =
  int do_check = 0;       // also: res = sqrt(dst*res_normal); hist[itr] = res; eps test -> flag/conv_itr
  int itr = 0;
  double res_normal = 0.0, eps = 0.0;
  double* hist = nullptr;
  int* flag = nullptr;
  int* conv_itr = nullptr;
  unsigned* counter = nullptr;  // arrival ticket, zero before every launch (the last workgroup resets it)
  // MODE_AX only: fold the dot products that follow the SpMV in BiCGSTAB into it (cz_Poisson.cpp:421-427, 457-464):
  // dst[0] = sum out*y, dst2[0] = sum out*out over the inner box (per-point products rounded to REAL like blas_dot1/2)
  int ax_dots = 0;
  const REAL* doty = nullptr;
  double* dst2 = nullptr;
};

// 16-byte global accesses go through a native vector type so that hipcc emits one global_load/store_dwordx4
// (a struct copy was split into dwordx3 + dword stores).
template <int V>
struct NatVec {
  typedef REAL type __attribute__((ext_vector_type(V)));
};
template <>
struct NatVec<1> {
  typedef REAL type;
};
template <int V>
__device__ __forceinline__ Vec<V> ldv(const REAL* base, long long vec_index) {
  typedef typename NatVec<V>::type nv;
  const nv x = *reinterpret_cast<const nv*>(base + vec_index * V);
  Vec<V> r;
  __builtin_memcpy(&r, &x, sizeof(r));
  return r;
}
template <int V>
__device__ __forceinline__ void stv(REAL* base, long long vec_index, const Vec<V>& x) {
  typedef typename NatVec<V>::type nv;
  nv y;
  __builtin_memcpy(&y, &x, sizeof(y));
  *reinterpret_cast<nv*>(base + vec_index * V) = y;
}
// vector access at an ELEMENT offset that is only REAL-aligned (rows whose length is no multiple of the vector width)
template <int V>
__device__ __forceinline__ Vec<V> ldve(const REAL* base, long long elem) {
  typedef typename NatVec<V>::type nv;
  typedef nv unv __attribute__((aligned(sizeof(REAL))));
  const nv x = *reinterpret_cast<const unv*>(base + elem);
  Vec<V> r;
  __builtin_memcpy(&r, &x, sizeof(r));
  return r;
}
template <int V>
__device__ __forceinline__ void stve(REAL* base, long long elem, const Vec<V>& x) {
  typedef typename NatVec<V>::type nv;
  typedef nv unv __attribute__((aligned(sizeof(REAL))));
  nv y;
  __builtin_memcpy(&y, &x, sizeof(y));
  *reinterpret_cast<unv*>(base + elem) = y;
}
template <int V>
__device__ __forceinline__ Vec<V> zerov() {
  Vec<V> z;
#pragma unroll
  for (int c = 0; c < V; c++) z.v[c] = (REAL)0;
  return z;
}

// deterministic block reduction of one double per thread: wave64 shuffle tree, then LDS across waves.
template <int TB>
__device__ __forceinline__ double block_sum(double x, double* wsum /* TB/64 doubles of LDS */) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off, 64);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (lane == 0) wsum[wave] = x;
  __syncthreads();
  double s = 0.0;
  if (threadIdx.x == 0) {
#pragma unroll
    for (int w = 0; w < TB / 64; w++) s += wsum[w];
  }
  return s;  // valid on thread 0
}


// Hand-off of the per-workgroup partial sums to the workgroup that arrives last (in-kernel finalisation; no extra launch per sweep).
// Why it is sound (MI355X_MICROARCH.md, "Workgroup dispatch, XCD placement & inter-workgroup visibility"; cdna_hip_programming.md G16):
//   producer  lane 0 of every workgroup stores its partials with agent-scope relaxed atomic stores = `global_store ... sc1`: write-through,
//             the bytes leave the XCD's L2 for memory.  `s_waitcnt vmcnt(0)` (inline asm, so that no compiler pass can drop or move it)
//             holds the lane until those stores are acknowledged -- on gfx9 stores are counted in vmcnt -- and only then the lane adds to
//             the agent-scope ticket.  One lane signals for all of its workgroup's handed-off bytes because that lane stored all of them.
//   consumer  the workgroup whose add returned nblk-1 knows every other add, hence every other drain, came before its own.  Its lane 0
//             then executes ONE agent-scope acquire (`buffer_inv sc1`: no line of this CU's L1 survives) and waits for it; the
//             workgroup barrier that follows releases the other waves; all of them read the partials with agent-scope relaxed loads
//             (`sc1`: served from L2 / memory, never from L1).  The acquire costs about 1.7 us once per launch (one workgroup); a
//             release fence in every producer instead cost 27 % of a whole sweep (it writes back the XCD's dirty output lines).
//   The ticket is reset by the finishing workgroup and again at the start of every solve (reset_ticket).
// Called by lane 0 after its partial stores; returns true in the workgroup that must finalise.
__device__ __forceinline__ bool arrive_and_test_last(unsigned* counter, int nblk) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const unsigned ticket = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const bool last = (ticket == (unsigned)nblk - 1u);
  if (last) {
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  return last;
}

// MAF flavour (cz_maf.f90, cz_blas.f90:738-1039): the six neighbour weights and the diagonal are
// recomputed at every point from 1-D coordinate arrays (device copies of xc, yc, zc; X(i) of the Fortran is xc[i+1], which
// for g = 2 is xc[padded index]).  pvt: row scaling of calc_ax_maf / calc_rk_maf.
struct MafArgs {
  const REAL* xc;
  const REAL* yc;
  const REAL* zc;
  const REAL* pvt;
};

struct MafW {
  REAL w1, w2, w3, w4, w5, w6, dd;  // weights of p(i+1), p(i-1), p(j+1), p(j-1), p(k+1), p(k-1); dd = 2(C1+C2+C3)
};

// cz_maf.f90:193-221, operation for operation
__device__ __forceinline__ MafW maf_weights(REAL XG, REAL XGG, REAL YE, REAL YEE, REAL ZT, REAL ZTT) {
  const REAL YJA = XG * YE * ZT;
  const REAL YJAI = (REAL)1.0 / YJA;
  const REAL GX = YE * ZT * YJAI;
  const REAL EY = XG * ZT * YJAI;
  const REAL TZ = XG * YE * YJAI;
  const REAL C1 = GX * GX, C2 = EY * EY, C3 = TZ * TZ;
  const REAL C7 = -XGG * C1 * GX;
  const REAL C8 = -YEE * C2 * EY;
  const REAL C9 = -ZTT * C3 * TZ;
  MafW w;
  w.w1 = C1 + (REAL)0.5 * C7, w.w2 = C1 - (REAL)0.5 * C7;
  w.w3 = C2 + (REAL)0.5 * C8, w.w4 = C2 - (REAL)0.5 * C8;
  w.w5 = C3 + (REAL)0.5 * C9, w.w6 = C3 - (REAL)0.5 * C9;
  w.dd = (REAL)2.0 * (C1 + C2 + C3);
  return w;
}
