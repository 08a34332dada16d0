// cumask_lab.hip -- what a CU mask on a HIP stream does on MI355X (gfx950), measured:
//   (1) which CU (XCC, SE, SH, CU id) each bit of the mask of hipExtStreamCreateWithCUMask enables,
//   (2) the CUs a kernel on a masked stream really runs on when another stream keeps the whole chip busy,
//   (3) what a streaming kernel loses when its stream is confined to (32 - k) CUs per XCD.
// The driver uses the answer to keep k CUs per XCD free for RCCL's send/recv kernels while the interior sweep runs.
//   hipcc --offload-arch=gfx950 -O2 tools/cumask_lab.hip -o tools/bin/cumask_lab
//   tools/bin/cumask_lab map | reserve <k> | stream <k>
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <set>
#include <vector>

#define CK(x)                                                                                  \
  do {                                                                                         \
    hipError_t e_ = (x);                                                                       \
    if (e_ != hipSuccess) {                                                                    \
      fprintf(stderr, "HIP error %s at %s:%d: %s\n", hipGetErrorString(e_), __FILE__, __LINE__, #x); \
      exit(1);                                                                                 \
    }                                                                                          \
  } while (0)

// one record per workgroup: XCC id and the HW_ID register (gfx9: cu_id [11:8], sh_id [12], se_id [15:13])
__global__ void where_k(unsigned* out, long long spin_ticks) {
  if (threadIdx.x == 0) {
    unsigned xcc, hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    out[2 * blockIdx.x] = xcc & 0xf;
    out[2 * blockIdx.x + 1] = hw;
  }
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < spin_ticks) {
  }
}

__global__ void __launch_bounds__(256) stream_k(const float4* __restrict__ a, const float4* __restrict__ b, float4* __restrict__ c, size_t n) {
  for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const float4 x = a[i], y = b[i];
    c[i] = make_float4(x.x + y.x, x.y + y.y, x.z + y.z, x.w + y.w);
  }
}

static unsigned cu_key(unsigned xcc, unsigned hw) { return (xcc << 16) | (hw & 0xff00); }  // XCC | se, sh, cu

static hipStream_t masked_stream(const std::vector<uint32_t>& mask) {
  hipStream_t s;
  CK(hipExtStreamCreateWithCUMask(&s, (uint32_t)mask.size(), mask.data()));
  return s;
}

// the CUs a kernel of `nwg` one-wave workgroups (each spinning `us` microseconds) runs on
static std::set<unsigned> cus_used(hipStream_t s, int nwg, double us, unsigned* d_out, std::vector<unsigned>& h) {
  CK(hipMemsetAsync(d_out, 0xff, (size_t)2 * nwg * sizeof(unsigned), s));
  hipLaunchKernelGGL(where_k, dim3(nwg), dim3(64), 0, s, d_out, (long long)(us * 100.0));
  CK(hipGetLastError());
  CK(hipStreamSynchronize(s));
  h.resize((size_t)2 * nwg);
  CK(hipMemcpy(h.data(), d_out, h.size() * sizeof(unsigned), hipMemcpyDeviceToHost));
  std::set<unsigned> u;
  for (int i = 0; i < nwg; i++) u.insert(cu_key(h[2 * i], h[2 * i + 1]));
  return u;
}

// mask with k CUs per XCD left out, given the bit -> XCC map found by `map` (bit i -> XCC i % nxcc on this part)
static std::vector<uint32_t> reserve_mask(int ncu, int nxcc, int k, bool complement) {
  std::vector<uint32_t> m((ncu + 31) / 32, 0u);
  const int per = ncu / nxcc;
  for (int i = 0; i < ncu; i++) {
    const int slot = i / nxcc;  // the slot-th CU of XCC i % nxcc
    const bool reserved = slot >= per - k;
    if (reserved == complement) m[i / 32] |= 1u << (i % 32);
  }
  return m;
}

int main(int argc, char** argv) {
  const char* what = argc > 1 ? argv[1] : "map";
  hipDeviceProp_t prop;
  CK(hipSetDevice(0));
  CK(hipGetDeviceProperties(&prop, 0));
  const int ncu = prop.multiProcessorCount, nxcc = 8;
  printf("device %s: %d CUs\n", prop.name, ncu);
  unsigned* d_out = nullptr;
  const int NWG = 8192;
  CK(hipMalloc(&d_out, (size_t)2 * NWG * sizeof(unsigned)));
  std::vector<unsigned> h;

  if (!strcmp(what, "map")) {
    // every single-bit mask: the one CU it enables
    std::map<unsigned, int> seen;
    printf("bit -> xcc se sh cu   (HW_ID decode: cu [11:8], sh [12], se [15:13])\n");
    for (int bit = 0; bit < ncu; bit++) {
      std::vector<uint32_t> m((ncu + 31) / 32, 0u);
      m[bit / 32] = 1u << (bit % 32);
      hipStream_t s = masked_stream(m);
      const std::set<unsigned> u = cus_used(s, 64, 5.0, d_out, h);
      CK(hipStreamDestroy(s));
      printf("%3d ->", bit);
      for (unsigned key : u) {
        printf("  xcc %u se %u sh %u cu %2u", key >> 16, (key >> 13) & 7, (key >> 12) & 1, (key >> 8) & 15);
        seen[key]++;
      }
      printf("%s\n", u.size() == 1 ? "" : "   <-- not exactly one CU");
    }
    printf("distinct CUs reached by single-bit masks: %zu of %d\n", seen.size(), ncu);
    // the unmasked stream for comparison
    hipStream_t s0;
    CK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking));
    const std::set<unsigned> all = cus_used(s0, NWG, 20.0, d_out, h);
    int per_xcc[16] = {0};
    for (unsigned key : all) per_xcc[key >> 16]++;
    printf("unmasked stream, %d workgroups: %zu CUs; per XCC:", NWG, all.size());
    for (int x = 0; x < nxcc; x++) printf(" %d", per_xcc[x]);
    printf("\n");
    return 0;
  }

  const int k = argc > 2 ? atoi(argv[2]) : 2;
  if (!strcmp(what, "reserve")) {
    // compute stream without k CUs per XCD, exchange stream on exactly those CUs: disjoint?  complete?
    hipStream_t sc = masked_stream(reserve_mask(ncu, nxcc, k, false));
    hipStream_t sx = masked_stream(reserve_mask(ncu, nxcc, k, true));
    const std::set<unsigned> uc = cus_used(sc, NWG, 20.0, d_out, h);
    const std::set<unsigned> ux = cus_used(sx, 1024, 20.0, d_out, h);
    int pc[16] = {0}, px[16] = {0}, both = 0;
    for (unsigned key : uc) pc[key >> 16]++;
    for (unsigned key : ux) px[key >> 16]++, both += (int)uc.count(key);
    printf("k = %d: compute stream on %zu CUs (per XCC:", k, uc.size());
    for (int x = 0; x < nxcc; x++) printf(" %d", pc[x]);
    printf("), exchange stream on %zu CUs (per XCC:", ux.size());
    for (int x = 0; x < nxcc; x++) printf(" %d", px[x]);
    printf("), shared: %d\n", both);
    return 0;
  }

  if (!strcmp(what, "stream")) {
    // c = a + b over 3 x 512 MiB on the whole chip and on (32 - k) CUs per XCD, k = 0 .. argv[2]
    const size_t n = (size_t)32 << 20;  // float4 elements: 512 MiB per array
    float4 *a, *b, *c;
    CK(hipMalloc(&a, n * sizeof(float4)));
    CK(hipMalloc(&b, n * sizeof(float4)));
    CK(hipMalloc(&c, n * sizeof(float4)));
    CK(hipMemset(a, 0, n * sizeof(float4)));
    CK(hipMemset(b, 0, n * sizeof(float4)));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int kk = 0; kk <= k; kk++) {
      hipStream_t s = masked_stream(reserve_mask(ncu, nxcc, kk, false));
      const int cus = ncu - kk * nxcc;
      float best = 1e30f;
      for (int rep = 0; rep < 12; rep++) {
        CK(hipEventRecord(e0, s));
        hipLaunchKernelGGL(stream_k, dim3(cus * 8), dim3(256), 0, s, a, b, c, n);
        CK(hipEventRecord(e1, s));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep >= 2) best = std::min(best, ms);
      }
      printf("k = %d (%3d CUs): c = a + b over 3 x 512 MiB  %.3f ms  %.0f GB/s\n", kk, cus, best, 3.0 * n * 16 / best * 1e-6);
      CK(hipStreamDestroy(s));
    }
    return 0;
  }
  fprintf(stderr, "usage: cumask_lab map | reserve <k> | stream <kmax>\n");
  return 1;
}
