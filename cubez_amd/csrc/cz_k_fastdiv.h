// cz_k_fastdiv.h -- part of cz_kernels.hip (included inside its anonymous namespace): IEEE-exact division by a loop-invariant divisor.
//
// The sweeps divide by the same diagonal coefficient dd at every point (cz_solver.f90:345, `dp = ((ss - bb) / dd - pp) * omg`), and the
// division has to stay the correctly rounded IEEE one: fields are compared bit for bit with the reference.  hipcc expands `n / d` into
//     den_s = div_scale(d, d, n)   r0 = rcp(den_s)   r1 = Newton step(s) on r0            <- depends on n only through den_s
//     num_s = div_scale(n, d, n)   q = refinement of num_s * r1 against den_s, div_fmas, div_fixup
// (AMDGPU LowerFDIV32 / LowerFDIV64) -- 11 instructions for FP32, one of them the quarter-rate v_rcp_f32; about 40 % of the vector
// instructions of a sweep.  den_s takes only two numeric values for a divisor of ordinary magnitude: d itself, or d * 2^64 (FP32; 2^128
// in FP64) when the numerator is tiny or the quotient close to overflow (and NaN for a zero numerator, where div_fixup supplies the
// result).  So the reciprocal refinement is done ONCE per thread for both values with the very same instructions, and a point only
// selects between the two: bit-identical to `n / d` by construction, 3 instructions (FP32; 5 in FP64) and the transcendental cheaper.
// czhip_selftest_fastdiv (tests/test_gpu_kernels.py) compares the result with `n / d` for all 2^32 float numerators (2^32 sampled double
// numerators: every sign/exponent pattern x 2^20 mantissas) per divisor;
// fastdiv_ok() is the host-side gate: divisors near the ends of the exponent range make div_scale scale differently, the launchers then
// take the kernels that divide the ordinary way.
template <typename R>
struct FastDiv {
  R d;    // the divisor
  R r1;   // refined reciprocal of d
  R r1s;  // refined reciprocal of d * 2^64 (2^128)
};

inline bool fastdiv_ok(float d) {
  const float a = d < 0 ? -d : d;
  return a >= 0x1p-60f && a <= 0x1p60f;
}
inline bool fastdiv_ok(double d) {
  const double a = d < 0 ? -d : d;
  return a >= 0x1p-400 && a <= 0x1p400;
}

__device__ __forceinline__ FastDiv<float> fastdiv_init(float d) {
  FastDiv<float> f;
  f.d = d;
  {
    const float r0 = __builtin_amdgcn_rcpf(d);
    const float e = __builtin_fmaf(-d, r0, 1.0f);
    f.r1 = __builtin_fmaf(e, r0, r0);
  }
  {
    const float ds = __builtin_amdgcn_ldexpf(d, 64);
    const float r0 = __builtin_amdgcn_rcpf(ds);
    const float e = __builtin_fmaf(-ds, r0, 1.0f);
    f.r1s = __builtin_fmaf(e, r0, r0);
  }
  return f;
}

__device__ __forceinline__ float fastdiv(float n, const FastDiv<float>& f) {
  bool vcc, unused;
  const float den_s = __builtin_amdgcn_div_scalef(n, f.d, false, &unused);
  const float num_s = __builtin_amdgcn_div_scalef(n, f.d, true, &vcc);
  const float r1 = (den_s == f.d) ? f.r1 : f.r1s;
  const float q0 = num_s * r1;
  const float e2 = __builtin_fmaf(-den_s, q0, num_s);
  const float q1 = __builtin_fmaf(e2, r1, q0);
  const float e3 = __builtin_fmaf(-den_s, q1, num_s);
  const float q = __builtin_amdgcn_div_fmasf(e3, r1, q1, vcc);
  return __builtin_amdgcn_div_fixupf(q, f.d, n);
}

__device__ __forceinline__ FastDiv<double> fastdiv_init(double d) {
  FastDiv<double> f;
  f.d = d;
  {
    const double r0 = __builtin_amdgcn_rcp(d);
    const double e0 = __builtin_fma(-d, r0, 1.0);
    const double ra = __builtin_fma(r0, e0, r0);
    const double e1 = __builtin_fma(-d, ra, 1.0);
    f.r1 = __builtin_fma(ra, e1, ra);
  }
  {
    const double ds = __builtin_amdgcn_ldexp(d, 128);
    const double r0 = __builtin_amdgcn_rcp(ds);
    const double e0 = __builtin_fma(-ds, r0, 1.0);
    const double ra = __builtin_fma(r0, e0, r0);
    const double e1 = __builtin_fma(-ds, ra, 1.0);
    f.r1s = __builtin_fma(ra, e1, ra);
  }
  return f;
}

__device__ __forceinline__ double fastdiv(double n, const FastDiv<double>& f) {
  bool vcc, unused;
  const double den_s = __builtin_amdgcn_div_scale(n, f.d, false, &unused);
  const double num_s = __builtin_amdgcn_div_scale(n, f.d, true, &vcc);
  const double r1 = (den_s == f.d) ? f.r1 : f.r1s;
  const double q0 = num_s * r1;
  const double e2 = __builtin_fma(-den_s, q0, num_s);
  const double q = __builtin_amdgcn_div_fmas(e2, r1, q0, vcc);
  return __builtin_amdgcn_div_fixup(q, f.d, n);
}

// The short form (FP32): with r = RN(1/d), q0 = n*r is within an ulp of the quotient, e = n - d*q0 is exact in one fma, and
// q1 = q0 + e*r rounds to the correctly rounded quotient -- where nothing underflows on the way.  div_fixup supplies zeros with their
// sign, infinities and NaNs.  4 instructions against 10.  "Where nothing underflows" is not argued but CHECKED: a divisor takes this
// form only after shortdiv_check_k has compared it with `n / d` for all 2^32 numerators (czhip fastdiv gate, cz_kernels.hip); any
// difference and the divisor keeps the form above.
__device__ __forceinline__ float shortdiv(float n, const FastDiv<float>& f) {
  const float q0 = n * f.r1;
  const float e = __builtin_fmaf(-f.d, q0, n);
  const float q1 = __builtin_fmaf(e, f.r1, q0);
  return __builtin_amdgcn_div_fixupf(q1, f.d, n);
}
__device__ __forceinline__ double shortdiv(double n, const FastDiv<double>& f) { return fastdiv(n, f); }  // (FP64: no short form)

// The hoisted form with ONE correction step instead of two (FP32; the FP64 form above has one already): q0 = num_s * r1 is within an ulp
// when r1 is the correctly rounded reciprocal, and div_fmas rounds once, also into the subnormal range.  8 instructions against 10.
// Checked per divisor like the short form.
__device__ __forceinline__ float mediumdiv(float n, const FastDiv<float>& f) {
  bool vcc, unused;
  const float den_s = __builtin_amdgcn_div_scalef(n, f.d, false, &unused);
  const float num_s = __builtin_amdgcn_div_scalef(n, f.d, true, &vcc);
  const float r1 = (den_s == f.d) ? f.r1 : f.r1s;
  const float q0 = num_s * r1;
  const float e2 = __builtin_fmaf(-den_s, q0, num_s);
  const float q = __builtin_amdgcn_div_fmasf(e2, r1, q0, vcc);
  return __builtin_amdgcn_div_fixupf(q, f.d, n);
}
__device__ __forceinline__ double mediumdiv(double n, const FastDiv<double>& f) { return fastdiv(n, f); }

// every numerator of the self-test (see above); counts results whose bits differ from the compiler's n / d (two NaNs count as equal)
template <int SHORT>
__global__ void fastdiv_check_k(REAL d, unsigned long long* bad) {
  const FastDiv<REAL> f = fastdiv_init(d);
  unsigned long long c = 0;
  for (unsigned long long t = blockIdx.x * (unsigned long long)blockDim.x + threadIdx.x; t < (1ull << 32); t += (unsigned long long)gridDim.x * blockDim.x) {
    REAL n;
    if (sizeof(REAL) == 4) {
      const unsigned b = (unsigned)t;
      __builtin_memcpy(&n, &b, 4);
    } else {
      const unsigned long long se = t >> 20, mi = t & 0xfffff;
      unsigned long long h = t * 0x9E3779B97F4A7C15ull;
      h ^= h >> 29;
      unsigned long long mant = (mi << 32) | (h & 0xffffffffull);
      if (mi == 0) mant = 0;
      if (mi == 1) mant = 0xfffffffffffffull;
      if (mi == 2) mant = 1;
      const unsigned long long b = (se << 52) | (mant & 0xfffffffffffffull);
      __builtin_memcpy(&n, &b, sizeof(REAL));
    }
    const REAL q = SHORT == 1 ? shortdiv(n, f) : SHORT == 2 ? mediumdiv(n, f) : fastdiv(n, f), r = n / d;
    if ((sizeof(REAL) == 4 ? (__builtin_bit_cast(unsigned, (float)q) != __builtin_bit_cast(unsigned, (float)r)) : (__builtin_bit_cast(unsigned long long, (double)q) != __builtin_bit_cast(unsigned long long, (double)r))) && !(q != q && r != r)) c++;
  }
  if (c) atomicAdd(bad, c);
}
