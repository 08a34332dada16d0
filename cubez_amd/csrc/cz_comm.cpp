// cz_comm.cpp -- see cz_comm.h.  Face geometry (K-fastest layout, cz_solver.f90:29):
//   J faces  are contiguous: rows i=1..NI of plane j  -> sent/received in place, no pack kernel
//   I faces  are NJ runs of NK elements (one k-row per j)        -> packed [j][k]
//   K faces  are fully strided (one element per (i,j))            -> packed [j][i]
// Only owned cells travel (no edges/corners): the 7-point stencil never reads them.
#include "cz_comm.h"

#include <rccl/rccl.h>

#include <algorithm>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <vector>

#include "cz_internal.h"

#define NCCL_CHECK(expr)                                                                              \
  do {                                                                                                \
    ncclResult_t r_ = (expr);                                                                         \
    if (r_ != ncclSuccess) {                                                                          \
      fprintf(stderr, "czhip: RCCL error %d (%s) at %s:%d: %s\n", (int)r_, ncclGetErrorString(r_), __FILE__, \
              __LINE__, #expr);                                                                       \
      exit(1);                                                                                        \
    }                                                                                                 \
  } while (0)

namespace {

// ---- in-process world (LOCAL transport)
struct LocalWorld {
  int n = 0;
  std::mutex mu;
  std::condition_variable cv;
  int arrived = 0;
  long generation = 0;
  std::vector<CommCtx*> ranks;
  std::vector<double> red;
  void barrier() {
    std::unique_lock<std::mutex> lk(mu);
    const long gen = generation;
    if (++arrived == n) {
      arrived = 0;
      generation++;
      cv.notify_all();
    } else {
      cv.wait(lk, [&] { return generation != gen; });
    }
  }
};

enum Transport { T_NONE = 0, T_RCCL = 1, T_LOCAL = 2 };

struct Boot {
  Transport tr = T_NONE;
  int rank = 0, nproc = 1;
  ncclComm_t nccl = nullptr;
  LocalWorld* world = nullptr;
};
thread_local Boot boot;

template <typename T>
__global__ void pack_iface_k(T* __restrict__ buf, const T* __restrict__ X, int NK, int NJ, int nkp, int nip, int ii, int g,
                             const int* __restrict__ skip) {
  if (skip && *skip) return;
  const int k = blockIdx.x * blockDim.x + threadIdx.x;  // 0..NK-1
  const int j = blockIdx.y;                             // 0..NJ-1
  if (k >= NK) return;
  buf[(size_t)j * NK + k] = X[(size_t)(k + g) + (size_t)ii * nkp + (size_t)(j + g) * nkp * nip];
}
template <typename T>
__global__ void unpack_iface_k(T* __restrict__ X, const T* __restrict__ buf, int NK, int NJ, int nkp, int nip, int ii, int g,
                               const int* __restrict__ skip) {
  if (skip && *skip) return;
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  const int j = blockIdx.y;
  if (k >= NK) return;
  X[(size_t)(k + g) + (size_t)ii * nkp + (size_t)(j + g) * nkp * nip] = buf[(size_t)j * NK + k];
}
template <typename T>
__global__ void pack_kface_k(T* __restrict__ buf, const T* __restrict__ X, int NI, int NJ, int nkp, int nip, int kk, int g,
                             const int* __restrict__ skip) {
  if (skip && *skip) return;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int j = blockIdx.y;
  if (i >= NI) return;
  buf[(size_t)j * NI + i] = X[(size_t)kk + (size_t)(i + g) * nkp + (size_t)(j + g) * nkp * nip];
}
template <typename T>
__global__ void unpack_kface_k(T* __restrict__ X, const T* __restrict__ buf, int NI, int NJ, int nkp, int nip, int kk, int g,
                               const int* __restrict__ skip) {
  if (skip && *skip) return;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int j = blockIdx.y;
  if (i >= NI) return;
  X[(size_t)kk + (size_t)(i + g) * nkp + (size_t)(j + g) * nkp * nip] = buf[(size_t)j * NI + i];
}

// ---- depth-2 exchange (two ghost layers, edges included) for the two-sweep kernel --------------------------------
template <typename T>
__global__ void pack_i2_k(T* __restrict__ buf, const T* __restrict__ X, int NK, int NJ, int nkp, int nip, int ii0, int g,
                          const int* __restrict__ skip) {
  if (skip && *skip) return;
  const int k = blockIdx.x * blockDim.x + threadIdx.x, j = blockIdx.y, l = blockIdx.z;
  if (k >= NK) return;
  buf[((size_t)l * NJ + j) * NK + k] = X[(size_t)(k + g) + (size_t)(ii0 + l) * nkp + (size_t)(j + g) * nkp * nip];
}
template <typename T>
__global__ void unpack_i2_k(T* __restrict__ X, const T* __restrict__ buf, int NK, int NJ, int nkp, int nip, int ii0, int g,
                            const int* __restrict__ skip) {
  if (skip && *skip) return;
  const int k = blockIdx.x * blockDim.x + threadIdx.x, j = blockIdx.y, l = blockIdx.z;
  if (k >= NK) return;
  X[(size_t)(k + g) + (size_t)(ii0 + l) * nkp + (size_t)(j + g) * nkp * nip] = buf[((size_t)l * NJ + j) * NK + k];
}
// K layers over the WHOLE padded (i,j) extent: carries the i/j ghost values received in the earlier phases (edges)
template <typename T>
__global__ void pack_k2_k(T* __restrict__ buf, const T* __restrict__ X, int nkp, int nip, int njp, int kk0,
                          const int* __restrict__ skip) {
  if (skip && *skip) return;
  const int ii = blockIdx.x * blockDim.x + threadIdx.x, jj = blockIdx.y, l = blockIdx.z;
  if (ii >= nip) return;
  buf[((size_t)l * njp + jj) * nip + ii] = X[(size_t)(kk0 + l) + (size_t)ii * nkp + (size_t)jj * nkp * nip];
}
template <typename T>
__global__ void unpack_k2_k(T* __restrict__ X, const T* __restrict__ buf, int nkp, int nip, int njp, int kk0,
                            const int* __restrict__ skip) {
  if (skip && *skip) return;
  const int ii = blockIdx.x * blockDim.x + threadIdx.x, jj = blockIdx.y, l = blockIdx.z;
  if (ii >= nip) return;
  X[(size_t)(kk0 + l) + (size_t)ii * nkp + (size_t)jj * nkp * nip] = buf[((size_t)l * njp + jj) * nip + ii];
}

}  // namespace

struct CommCtx {
  Transport tr;
  int rank, nproc, eb;
  int size[3], nID[6];
  int g = 2;
  size_t face_elems[6];
  void* sendbuf[6] = {nullptr};
  void* recvbuf[6] = {nullptr};
  ncclComm_t nccl = nullptr;
  LocalWorld* world = nullptr;
  void* cur_X = nullptr;  // LOCAL: array being exchanged, published for the neighbours' J-face copies
  double* h_red = nullptr;
  // depth-2 exchange buffers (allocated on first use): [face] for I-,I+,K-,K+ (J faces travel in place)
  void* send2[6] = {nullptr};
  void* recv2[6] = {nullptr};
  size_t elems2[6] = {0};
};

// ------------------------------------------------------------------------------------------------------------
void comm_world(int* rank, int* nproc) {
  *rank = boot.rank;
  *nproc = boot.nproc;
}

void comm_auto_division(int nproc, const int G[3], int D[3]) {
  double best = -1.0;
  int bd[3] = {1, 1, nproc};
  for (int di = 1; di <= nproc; di++) {
    if (nproc % di) continue;
    for (int dj = 1; dj <= nproc / di; dj++) {
      if ((nproc / di) % dj) continue;
      const int dk = nproc / di / dj;
      if (di > G[0] / 2 || dj > G[1] / 2 || dk > G[2] / 2) continue;
      // exchanged elements per brick (both directions folded): cut faces only; packed faces cost a little more
      const double li = (double)G[0] / di, lj = (double)G[1] / dj, lk = (double)G[2] / dk;
      double cost = 0.0;
      // the busiest brick has min(2, d-1) cut faces per axis
      cost += std::min(2, dj - 1) * li * lk * 1.00;  // J face: contiguous
      cost += std::min(2, di - 1) * lj * lk * 1.02;  // I face: k-rows
      cost += std::min(2, dk - 1) * li * lj * 1.05;  // K face: strided
      if (best < 0 || cost < best) {
        best = cost;
        bd[0] = di, bd[1] = dj, bd[2] = dk;
      }
    }
  }
  D[0] = bd[0], D[1] = bd[1], D[2] = bd[2];
}

bool comm_decompose(const int G[3], const int D[3], int nproc, int rank, int size[3], int head[3], int nID[6]) {
  if (D[0] < 1 || D[1] < 1 || D[2] < 1 || D[0] * D[1] * D[2] != nproc || rank < 0 || rank >= nproc) return false;
  int r[3] = {rank % D[0], (rank / D[0]) % D[1], rank / (D[0] * D[1])};
  for (int a = 0; a < 3; a++) {
    const int base = G[a] / D[a], rem = G[a] % D[a];
    if (base < 2) return false;  // every brick needs an inner point next to each face
    size[a] = base + (r[a] < rem ? 1 : 0);
    head[a] = r[a] * base + std::min(r[a], rem) + 1;
  }
  auto rk = [&](int a, int b, int c) { return a + D[0] * (b + D[1] * c); };
  nID[0] = r[0] > 0 ? rk(r[0] - 1, r[1], r[2]) : -1;
  nID[1] = r[0] < D[0] - 1 ? rk(r[0] + 1, r[1], r[2]) : -1;
  nID[2] = r[1] > 0 ? rk(r[0], r[1] - 1, r[2]) : -1;
  nID[3] = r[1] < D[1] - 1 ? rk(r[0], r[1] + 1, r[2]) : -1;
  nID[4] = r[2] > 0 ? rk(r[0], r[1], r[2] - 1) : -1;
  nID[5] = r[2] < D[2] - 1 ? rk(r[0], r[1], r[2] + 1) : -1;
  return true;
}

CommCtx* comm_create(int rank, int nproc, const int size[3], const int nID[6], int elem_bytes) {
  if (boot.tr == T_NONE || boot.nproc != nproc) {
    fprintf(stderr, "czhip: %d ranks requested but no communicator was bootstrapped (cz_comm_bootstrap*)\n", nproc);
    return nullptr;
  }
  CommCtx* c = new CommCtx();
  c->tr = boot.tr, c->rank = rank, c->nproc = nproc, c->eb = elem_bytes;
  c->nccl = boot.nccl, c->world = boot.world;
  for (int a = 0; a < 3; a++) c->size[a] = size[a];
  for (int f = 0; f < 6; f++) c->nID[f] = nID[f];
  const size_t NI = size[0], NJ = size[1], NK = size[2];
  c->face_elems[0] = c->face_elems[1] = NJ * NK;
  c->face_elems[2] = c->face_elems[3] = NI * (NK + 2 * c->g);  // in-place rows incl. k guide cells
  c->face_elems[4] = c->face_elems[5] = NI * NJ;
  for (int f = 0; f < 6; f++) {
    if (nID[f] < 0 || f == 2 || f == 3) continue;
    HIP_CHECK(hipMalloc(&c->sendbuf[f], c->face_elems[f] * elem_bytes));
    HIP_CHECK(hipMalloc(&c->recvbuf[f], c->face_elems[f] * elem_bytes));
  }
  HIP_CHECK(hipHostMalloc(&c->h_red, 16 * sizeof(double), hipHostMallocDefault));
  if (c->tr == T_LOCAL) {
    std::lock_guard<std::mutex> lk(c->world->mu);
    c->world->ranks[rank] = c;
  }
  return c;
}

void comm_destroy(CommCtx* c) {
  if (!c) return;
  for (int f = 0; f < 6; f++) {
    if (c->sendbuf[f]) (void)hipFree(c->sendbuf[f]);
    if (c->recvbuf[f]) (void)hipFree(c->recvbuf[f]);
  }
  for (int f = 0; f < 6; f++) {
    if (c->send2[f]) (void)hipFree(c->send2[f]);
    if (c->recv2[f]) (void)hipFree(c->recv2[f]);
  }
  (void)hipHostFree(c->h_red);
  delete c;
}

namespace {
template <typename T>
void pack_faces(CommCtx* c, const T* X, const int* skip, hipStream_t st) {
  const int NI = c->size[0], NJ = c->size[1], NK = c->size[2], g = c->g;
  const int nkp = NK + 2 * g, nip = NI + 2 * g;
  // owned boundary layers: i = 1 / NI, k = 1 / NK (1-based) -> padded index +g-1
  if (c->nID[0] >= 0) hipLaunchKernelGGL(pack_iface_k<T>, dim3((NK + 127) / 128, NJ), dim3(128), 0, st, (T*)c->sendbuf[0], X, NK, NJ, nkp, nip, g, g, skip);
  if (c->nID[1] >= 0) hipLaunchKernelGGL(pack_iface_k<T>, dim3((NK + 127) / 128, NJ), dim3(128), 0, st, (T*)c->sendbuf[1], X, NK, NJ, nkp, nip, NI + g - 1, g, skip);
  if (c->nID[4] >= 0) hipLaunchKernelGGL(pack_kface_k<T>, dim3((NI + 127) / 128, NJ), dim3(128), 0, st, (T*)c->sendbuf[4], X, NI, NJ, nkp, nip, g, g, skip);
  if (c->nID[5] >= 0) hipLaunchKernelGGL(pack_kface_k<T>, dim3((NI + 127) / 128, NJ), dim3(128), 0, st, (T*)c->sendbuf[5], X, NI, NJ, nkp, nip, NK + g - 1, g, skip);
  HIP_CHECK(hipGetLastError());
}
template <typename T>
void unpack_faces(CommCtx* c, T* X, const int* skip, hipStream_t st) {
  const int NI = c->size[0], NJ = c->size[1], NK = c->size[2], g = c->g;
  const int nkp = NK + 2 * g, nip = NI + 2 * g;
  // ghost layers: i = 0 / NI+1, k = 0 / NK+1 (1-based) -> padded index g-1 / N+g
  if (c->nID[0] >= 0) hipLaunchKernelGGL(unpack_iface_k<T>, dim3((NK + 127) / 128, NJ), dim3(128), 0, st, X, (const T*)c->recvbuf[0], NK, NJ, nkp, nip, g - 1, g, skip);
  if (c->nID[1] >= 0) hipLaunchKernelGGL(unpack_iface_k<T>, dim3((NK + 127) / 128, NJ), dim3(128), 0, st, X, (const T*)c->recvbuf[1], NK, NJ, nkp, nip, NI + g, g, skip);
  if (c->nID[4] >= 0) hipLaunchKernelGGL(unpack_kface_k<T>, dim3((NI + 127) / 128, NJ), dim3(128), 0, st, X, (const T*)c->recvbuf[4], NI, NJ, nkp, nip, g - 1, g, skip);
  if (c->nID[5] >= 0) hipLaunchKernelGGL(unpack_kface_k<T>, dim3((NI + 127) / 128, NJ), dim3(128), 0, st, X, (const T*)c->recvbuf[5], NI, NJ, nkp, nip, NK + g, g, skip);
  HIP_CHECK(hipGetLastError());
}
// J faces in place: element offset of row i=1 (1-based) of plane j (1-based), k from the first guide cell
inline size_t jface_off(const CommCtx* c, int j1) {
  const size_t nkp = c->size[2] + 2 * c->g, nip = c->size[0] + 2 * c->g;
  return (size_t)(j1 + c->g - 1) * nkp * nip + (size_t)c->g * nkp;
}
}  // namespace

bool comm_halo(CommCtx* c, void* X, const int* skip, hipStream_t st) {
  if (!c) return true;
  char* Xb = (char*)X;
  const int NJ = c->size[1];
  const int opp[6] = {1, 0, 3, 2, 5, 4};
  if (c->eb == 4) pack_faces<float>(c, (const float*)X, skip, st);
  else pack_faces<double>(c, (const double*)X, skip, st);

  if (c->tr == T_RCCL) {
    const ncclDataType_t dt = c->eb == 4 ? ncclFloat : ncclDouble;
    NCCL_CHECK(ncclGroupStart());
    for (int f = 0; f < 6; f++) {
      if (c->nID[f] < 0) continue;
      const void* sb;
      void* rb;
      if (f == 2) sb = Xb + jface_off(c, 1) * c->eb, rb = Xb + jface_off(c, 0) * c->eb;
      else if (f == 3) sb = Xb + jface_off(c, NJ) * c->eb, rb = Xb + jface_off(c, NJ + 1) * c->eb;
      else sb = c->sendbuf[f], rb = c->recvbuf[f];
      NCCL_CHECK(ncclSend(sb, c->face_elems[f], dt, c->nID[f], c->nccl, st));
      NCCL_CHECK(ncclRecv(rb, c->face_elems[f], dt, c->nID[f], c->nccl, st));
    }
    NCCL_CHECK(ncclGroupEnd());
  } else {  // LOCAL
    c->cur_X = X;
    HIP_CHECK(hipStreamSynchronize(st));
    c->world->barrier();  // every rank has packed and published
    for (int f = 0; f < 6; f++) {
      if (c->nID[f] < 0) continue;
      CommCtx* nb = c->world->ranks[c->nID[f]];
      if (f == 2 || f == 3) {
        // my ghost plane (j=0 / NJ+1) <- neighbour's owned plane (j=NJnb / 1)
        char* nbX = (char*)nb->cur_X;
        const size_t src = (f == 2) ? jface_off(nb, nb->size[1]) : jface_off(nb, 1);
        const size_t dst = (f == 2) ? jface_off(c, 0) : jface_off(c, NJ + 1);
        HIP_CHECK(hipMemcpyAsync(Xb + dst * c->eb, nbX + src * c->eb, c->face_elems[f] * c->eb, hipMemcpyDeviceToDevice, st));
      } else {
        HIP_CHECK(hipMemcpyAsync(c->recvbuf[f], nb->sendbuf[opp[f]], c->face_elems[f] * c->eb, hipMemcpyDeviceToDevice, st));
      }
    }
    HIP_CHECK(hipStreamSynchronize(st));
    c->world->barrier();  // nobody repacks before everyone has copied
  }

  if (c->eb == 4) unpack_faces<float>(c, (float*)X, skip, st);
  else unpack_faces<double>(c, (double*)X, skip, st);
  return true;
}


// ------------------------------------------------------------------------------------------------------------
// Depth-2 exchange in three dependent phases I -> J -> K.  Each later phase sends the ghost cells the earlier ones
// received, so the edge cells the first sweep of a fused pair needs (ghost layer 1 in two directions) arrive after
// two hops; no diagonal messages.
//   I: layers i = 1,2 / NI-1,NI of the owned (j,k) extent, packed [layer][j][k]
//   J: planes j = 1,2 / NJ-1,NJ, whole padded planes, in place (contiguous)
//   K: layers k = 1,2 / NK-1,NK of the whole padded (i,j) extent, packed [layer][jj][ii]
// ------------------------------------------------------------------------------------------------------------
namespace {
template <typename T>
bool halo2_impl(CommCtx* c, T* X, const int* skip, hipStream_t st) {
  const int NI = c->size[0], NJ = c->size[1], NK = c->size[2], g = c->g;
  const int nkp = NK + 2 * g, nip = NI + 2 * g, njp = NJ + 2 * g;
  const size_t PS = (size_t)nkp * nip;
  const int opp[6] = {1, 0, 3, 2, 5, 4};
  if (!c->elems2[0]) {
    c->elems2[0] = c->elems2[1] = (size_t)2 * NJ * NK;
    c->elems2[2] = c->elems2[3] = (size_t)2 * PS;
    c->elems2[4] = c->elems2[5] = (size_t)2 * nip * njp;
    for (int f : {0, 1, 4, 5}) {
      if (c->nID[f] < 0) continue;
      HIP_CHECK(hipMalloc(&c->send2[f], c->elems2[f] * sizeof(T)));
      HIP_CHECK(hipMalloc(&c->recv2[f], c->elems2[f] * sizeof(T)));
    }
  }
  const ncclDataType_t dt = sizeof(T) == 4 ? ncclFloat : ncclDouble;
  auto exchange = [&](int f0, const void* sb0, void* rb0, const void* sb1, void* rb1) {
    // faces f0 (minus) and f0+1 (plus) of one axis
    const void* sb[2] = {sb0, sb1};
    void* rb[2] = {rb0, rb1};
    if (c->tr == T_RCCL) {
      if (c->nID[f0] < 0 && c->nID[f0 + 1] < 0) return;
      NCCL_CHECK(ncclGroupStart());
      for (int s = 0; s < 2; s++) {
        const int f = f0 + s;
        if (c->nID[f] < 0) continue;
        NCCL_CHECK(ncclSend(sb[s], c->elems2[f], dt, c->nID[f], c->nccl, st));
        NCCL_CHECK(ncclRecv(rb[s], c->elems2[f], dt, c->nID[f], c->nccl, st));
      }
      NCCL_CHECK(ncclGroupEnd());
    } else {
      // LOCAL: publish my send pointers, then copy from the neighbours'
      c->send2[f0 + 0] = const_cast<void*>(sb0);  // (for J these are array regions, for I/K the pack buffers)
      c->send2[f0 + 1] = const_cast<void*>(sb1);
      HIP_CHECK(hipStreamSynchronize(st));
      c->world->barrier();
      for (int s = 0; s < 2; s++) {
        const int f = f0 + s;
        if (c->nID[f] < 0) continue;
        CommCtx* nb = c->world->ranks[c->nID[f]];
        HIP_CHECK(hipMemcpyAsync(rb[s], nb->send2[opp[f]], c->elems2[f] * sizeof(T), hipMemcpyDeviceToDevice, st));
      }
      HIP_CHECK(hipStreamSynchronize(st));
      c->world->barrier();
    }
  };

  // ---- phase I
  {
    dim3 grid((NK + 127) / 128, NJ, 2);
    void* s0 = c->send2[0];
    void* s1 = c->send2[1];
    if (c->nID[0] >= 0) hipLaunchKernelGGL(pack_i2_k<T>, grid, dim3(128), 0, st, (T*)s0, X, NK, NJ, nkp, nip, g, g, skip);
    if (c->nID[1] >= 0) hipLaunchKernelGGL(pack_i2_k<T>, grid, dim3(128), 0, st, (T*)s1, X, NK, NJ, nkp, nip, NI + g - 2, g, skip);
    exchange(0, s0, c->recv2[0], s1, c->recv2[1]);
    c->send2[0] = s0, c->send2[1] = s1;
    if (c->nID[0] >= 0) hipLaunchKernelGGL(unpack_i2_k<T>, grid, dim3(128), 0, st, X, (const T*)c->recv2[0], NK, NJ, nkp, nip, 0, g, skip);
    if (c->nID[1] >= 0) hipLaunchKernelGGL(unpack_i2_k<T>, grid, dim3(128), 0, st, X, (const T*)c->recv2[1], NK, NJ, nkp, nip, NI + g, g, skip);
  }
  // ---- phase J (in place: two whole padded planes)
  {
    void* keep0 = c->send2[2];
    void* keep1 = c->send2[3];
    exchange(2, X + (size_t)g * PS, X, X + (size_t)(NJ + g - 2) * PS, X + (size_t)(NJ + g) * PS);
    c->send2[2] = keep0, c->send2[3] = keep1;
  }
  // ---- phase K
  {
    dim3 grid((nip + 127) / 128, njp, 2);
    void* s0 = c->send2[4];
    void* s1 = c->send2[5];
    if (c->nID[4] >= 0) hipLaunchKernelGGL(pack_k2_k<T>, grid, dim3(128), 0, st, (T*)s0, X, nkp, nip, njp, g, skip);
    if (c->nID[5] >= 0) hipLaunchKernelGGL(pack_k2_k<T>, grid, dim3(128), 0, st, (T*)s1, X, nkp, nip, njp, NK + g - 2, skip);
    exchange(4, s0, c->recv2[4], s1, c->recv2[5]);
    c->send2[4] = s0, c->send2[5] = s1;
    if (c->nID[4] >= 0) hipLaunchKernelGGL(unpack_k2_k<T>, grid, dim3(128), 0, st, X, (const T*)c->recv2[4], nkp, nip, njp, 0, skip);
    if (c->nID[5] >= 0) hipLaunchKernelGGL(unpack_k2_k<T>, grid, dim3(128), 0, st, X, (const T*)c->recv2[5], nkp, nip, njp, NK + g, skip);
  }
  HIP_CHECK(hipGetLastError());
  return true;
}
}  // namespace

bool comm_halo2(CommCtx* c, void* X, const int* skip, hipStream_t st) {
  if (!c) return true;
  if (c->g != 2) return false;
  return c->eb == 4 ? halo2_impl<float>(c, (float*)X, skip, st) : halo2_impl<double>(c, (double*)X, skip, st);
}

bool comm_allreduce_sum(CommCtx* c, double* d_val, int count, hipStream_t st) {
  if (!c) return true;
  if (c->tr == T_RCCL) {
    NCCL_CHECK(ncclAllReduce(d_val, d_val, count, ncclDouble, ncclSum, c->nccl, st));
    return true;
  }
  LocalWorld* w = c->world;
  HIP_CHECK(hipMemcpyAsync(c->h_red, d_val, count * sizeof(double), hipMemcpyDeviceToHost, st));
  HIP_CHECK(hipStreamSynchronize(st));
  for (int i = 0; i < count; i++) w->red[(size_t)c->rank * 16 + i] = c->h_red[i];
  w->barrier();
  for (int i = 0; i < count; i++) {
    double s = 0.0;
    for (int r = 0; r < w->n; r++) s += w->red[(size_t)r * 16 + i];
    c->h_red[i] = s;
  }
  w->barrier();
  HIP_CHECK(hipMemcpyAsync(d_val, c->h_red, count * sizeof(double), hipMemcpyHostToDevice, st));
  HIP_CHECK(hipStreamSynchronize(st));
  return true;
}

double comm_allreduce_max_host(CommCtx* c, double v) {
  if (!c) return v;
  if (c->tr == T_RCCL) {
    double* d = nullptr;
    HIP_CHECK(hipMalloc(&d, sizeof(double)));
    HIP_CHECK(hipMemcpy(d, &v, sizeof(double), hipMemcpyHostToDevice));
    NCCL_CHECK(ncclAllReduce(d, d, 1, ncclDouble, ncclMax, c->nccl, czhip_internal::stream()));
    HIP_CHECK(hipStreamSynchronize(czhip_internal::stream()));
    HIP_CHECK(hipMemcpy(&v, d, sizeof(double), hipMemcpyDeviceToHost));
    (void)hipFree(d);
    return v;
  }
  LocalWorld* w = c->world;
  w->red[(size_t)c->rank * 16] = v;
  w->barrier();
  double m = v;
  for (int r = 0; r < w->n; r++) m = std::max(m, w->red[(size_t)r * 16]);
  w->barrier();
  return m;
}

// ============================================================================================================
// C-ABI: bootstrap + host-only decomposition helpers (declared in include/cz_hip.h part 5)
// ============================================================================================================
extern "C" {

int cz_comm_unique_id_bytes(void) { return (int)sizeof(ncclUniqueId); }

// rank 0 of a multi-process job: create the RCCL id that the launcher broadcasts (torch.distributed / a file)
int cz_comm_get_unique_id(char* out) {
  ncclUniqueId id;
  NCCL_CHECK(ncclGetUniqueId(&id));
  memcpy(out, &id, sizeof(id));
  return 0;
}

// every rank: join the communicator (one process per GPU; the device was bound by czhip_init)
int cz_comm_bootstrap(int rank, int nranks, const char* id_bytes) {
  if (nranks <= 1) {
    boot = Boot();
    return 0;
  }
  ncclUniqueId id;
  memcpy(&id, id_bytes, sizeof(id));
  ncclComm_t comm;
  NCCL_CHECK(ncclCommInitRank(&comm, nranks, id, rank));
  boot.tr = T_RCCL, boot.rank = rank, boot.nproc = nranks, boot.nccl = comm, boot.world = nullptr;
  return 0;
}

void cz_comm_shutdown(void) {
  if (boot.tr == T_RCCL && boot.nccl) NCCL_CHECK(ncclCommDestroy(boot.nccl));
  boot = Boot();
}

// LOCAL transport: a world of n ranks living in n threads of this process
void* cz_comm_local_world(int n) {
  LocalWorld* w = new LocalWorld();
  w->n = n;
  w->ranks.assign(n, nullptr);
  w->red.assign((size_t)n * 16, 0.0);
  return w;
}
void cz_comm_local_world_free(void* w) { delete (LocalWorld*)w; }
int cz_comm_bootstrap_local(void* world, int rank) {
  LocalWorld* w = (LocalWorld*)world;
  boot.tr = T_LOCAL, boot.rank = rank, boot.nproc = w->n, boot.nccl = nullptr, boot.world = w;
  return 0;
}

// One-rank RCCL smoke test (a one-GPU box cannot host two ranks): communicator creation from a unique id, all-reduce of
// a device double and a grouped send/recv to self on the library stream.  Returns 0 when every result is right.
int cz_comm_selftest(void) {
  ncclUniqueId id;
  NCCL_CHECK(ncclGetUniqueId(&id));
  ncclComm_t comm;
  NCCL_CHECK(ncclCommInitRank(&comm, 1, id, 0));
  hipStream_t st = czhip_internal::stream();
  double* d = nullptr;
  HIP_CHECK(hipMalloc(&d, 64 * sizeof(double)));
  double h[64];
  for (int i = 0; i < 64; i++) h[i] = 1.5 * i;
  HIP_CHECK(hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice));
  NCCL_CHECK(ncclAllReduce(d, d, 2, ncclDouble, ncclSum, comm, st));
  NCCL_CHECK(ncclGroupStart());
  NCCL_CHECK(ncclSend(d + 8, 8, ncclDouble, 0, comm, st));
  NCCL_CHECK(ncclRecv(d + 32, 8, ncclDouble, 0, comm, st));
  NCCL_CHECK(ncclGroupEnd());
  HIP_CHECK(hipStreamSynchronize(st));
  double r[64];
  HIP_CHECK(hipMemcpy(r, d, sizeof(r), hipMemcpyDeviceToHost));
  int bad = 0;
  if (r[0] != 0.0 || r[1] != 1.5) bad++;
  for (int i = 0; i < 8; i++)
    if (r[32 + i] != 1.5 * (8 + i)) bad++;
  (void)hipFree(d);
  NCCL_CHECK(ncclCommDestroy(comm));
  return bad;
}

void cz_comm_auto_division(int nproc, const int* G_size, int* G_div) { comm_auto_division(nproc, G_size, G_div); }
int cz_comm_decompose(const int* G_size, const int* G_div, int nproc, int rank, int* size, int* head, int* nID) {
  return comm_decompose(G_size, G_div, nproc, rank, size, head, nID) ? 1 : 0;
}

}  // extern "C"
