// cz_comm.cpp -- see cz_comm.h.
//
// One exchange = ONE pack launch, ONE grouped ncclSend/ncclRecv over all neighbours, ONE unpack launch.
// A message is the box of owned cells next to a face or an edge of the brick; the receiver stores it in the mirror-image
// ghost box.  How a box travels follows the K-fastest layout:
//   J face   whole padded planes are contiguous: sent from / received into the array itself, no pack, no unpack (a plane carries the
//            i / k guide cells along: +1.6 % bytes; what lands on the receiver's edge cells is overwritten by the edge messages' unpack,
//            which runs after the group on the same stream)
//   I face   whole padded k-rows: 16-byte vectors along k, one integer division per ROW; the unpack leaves the k guide cells alone
//   K face   one or two cells per row: one thread per (i,j) row, 8-byte access where the pair is aligned
//   edge     a few thousand cells: element by element
// Two shapes of exchange:
//   depth 1, faces only          Comm_S(X, 1) of the reference (cz_comm.cpp:23-38): what one sweep reads
//   depth 2, faces + 12 edges    what a fused pair of sweeps reads: the first sweep is also applied to ghost layer 1, which
//                                needs ghost layer 2 behind it and the edge cells (ghost in two directions) beside it;
//                                edges travel directly to the diagonal neighbours, so the exchange stays single-phase
// Corners are never read by a 7-point stencil at this depth.
#include "cz_comm.h"

#include <rccl/rccl.h>

#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <mutex>
#include <thread>
#include <vector>

#include "cz_config.h"
#include "cz_internal.h"

#define NCCL_CHECK(expr)                                                                                                          \
  do {                                                                                                                            \
    ncclResult_t r_ = (expr);                                                                                                     \
    if (r_ != ncclSuccess) cz_fatal(1, "czhip: RCCL error %d (%s) at %s:%d: %s\n", (int)r_, ncclGetErrorString(r_), __FILE__, __LINE__, #expr); \
  } while (0)

namespace {

double wall_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// Seconds a collective may stay incomplete before the job is ended with a diagnostic: CZ_COMM_TIMEOUT, default 300 when CZ_COMM_DEBUG
// is set (bench.py sets it for N > 1) and 120 for the LOCAL test transport, whose waits are host-side; 0 (or less) = wait for ever.
double comm_timeout_s(const CzConfig& cfg, bool local) {
  if (cfg.has(CZV_COMM_TIMEOUT)) return cfg.real(CZV_COMM_TIMEOUT, 0.0);
  if (local) return 120.0;
  return cfg.has(CZV_COMM_DEBUG) ? 300.0 : 0.0;
}

// ---- in-process world (LOCAL transport)
struct LocalWorld {
  int n = 0;
  std::mutex mu;
  std::condition_variable cv;
  int arrived = 0;
  long generation = 0;
  std::vector<CommCtx*> ranks;
  std::vector<double> red;
  double wait_limit = 120.0;  // seconds a rank waits at a host-side barrier (comm_timeout_s at creation of the world)
  // every wait is bounded: ranks that issue different sequences of collectives (the failure of commit 24dd087's parent: bricks disagreeing
  // on the pass they run) end the process with the rank and the barrier number instead of blocking for ever
  void barrier(int rank = -1) {
    std::unique_lock<std::mutex> lk(mu);
    const long gen = generation;
    if (++arrived == n) {
      arrived = 0;
      generation++;
      cv.notify_all();
    } else {
      const double lim = wait_limit;
      if (lim <= 0.0) {  // CZ_COMM_TIMEOUT=0: no bound
        cv.wait(lk, [&] { return generation != gen; });
      } else if (!cv.wait_for(lk, std::chrono::duration<double>(lim), [&] { return generation != gen; })) {
        cz_fatal_quick(3, "cz rank %d: LOCAL barrier #%ld: only %d of %d ranks arrived within %.0f s -- the ranks issue different collectives\n", rank,
                gen, arrived, n, lim);
      }
    }
  }
};

// ---- watchdog of the stream-ordered collectives (RCCL): every exchange / all-reduce gets a sequence number and an event behind it;
// a helper thread ends the process when the oldest incomplete one is older than the limit, naming rank, number and kind.  With RCCL a
// rank that issues a different sequence than its peers blocks the whole job silently; this turns it into an exit code and one line.
struct Watch {
  struct Item {
    hipEvent_t ev;
    long seq;
    const char* what;
    double t;
  };
  bool on = false;
  double limit = 0.0;
  int rank = 0, device = 0, verbose = 0;
  long seq = 0, done = 0;
  std::mutex mu;
  std::deque<Item> q;
  std::vector<hipEvent_t> pool;
  std::thread th;
  std::atomic<bool> quit{false};

  void start(int rank_, double limit_, int verbose_) {
    rank = rank_, limit = limit_, verbose = verbose_;
    if (limit <= 0.0) return;
    HIP_CHECK(hipGetDevice(&device));
    on = true;
    th = std::thread([this] { run(); });
  }
  void note(const char* what, hipStream_t st) {
    seq++;
    if (verbose >= 2) fprintf(stderr, "cz rank %d: collective #%ld %s\n", rank, seq, what);
    if (!on) return;
    hipEvent_t ev;
    {
      std::lock_guard<std::mutex> lk(mu);
      if (!pool.empty()) {
        ev = pool.back();
        pool.pop_back();
      } else {
        HIP_CHECK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
      }
    }
    HIP_CHECK(hipEventRecord(ev, st));
    std::lock_guard<std::mutex> lk(mu);
    q.push_back({ev, seq, what, wall_s()});
  }
  void run() {
    (void)hipSetDevice(device);
    while (!quit.load()) {
      usleep(100000);
      std::lock_guard<std::mutex> lk(mu);
      while (!q.empty() && hipEventQuery(q.front().ev) == hipSuccess) {
        done = q.front().seq;
        pool.push_back(q.front().ev);
        q.pop_front();
      }
      if (!q.empty() && wall_s() - q.front().t > limit) {
        cz_fatal_quick(3, "cz rank %d: collective #%ld (%s) has not completed %.0f s after it was issued (last completed: #%ld, issued so far: #%ld) -- "
                        "the ranks are out of step or a peer is gone\n", rank, q.front().seq, q.front().what, limit, done, seq);
      }
    }
  }
  void stop() {
    if (!on) return;
    quit.store(true);
    th.join();
    for (auto& i : q) (void)hipEventDestroy(i.ev);
    for (auto& e : pool) (void)hipEventDestroy(e);
    on = false;
  }
};

enum Transport { T_NONE = 0, T_RCCL = 1, T_LOCAL = 2 };

struct Boot {
  Transport tr = T_NONE;
  int rank = 0, nproc = 1;
  ncclComm_t nccl = nullptr;
  ncclComm_t nccl_red = nullptr;  // a second communicator over the same ranks for the all-reduces (see cz_comm_bootstrap)
  LocalWorld* world = nullptr;
};
thread_local Boot boot;


constexpr int MAX_BOX = 18;  // 6 faces + 12 edges

enum BoxKind { BOX_GENERIC = 0, BOX_ROWS = 1 /* I face: whole padded k-rows */, BOX_KPAIR = 2 /* K face */, BOX_DIRECT = 3 /* J face: not packed */ };

struct BoxDesc {
  int i0, j0, k0;   // padded 0-based start
  int ni, nj, nk;   // extent
  long long off;    // element offset in the packed buffer
  int kind;
};
struct BoxTable {
  int n;
  BoxDesc b[MAX_BOX];
};

template <typename T>
struct Vec16 {
  typedef T type __attribute__((ext_vector_type(16 / sizeof(T))));
};
template <typename T>
struct Vec8 {
  typedef T type __attribute__((ext_vector_type(8 / sizeof(T) > 1 ? 8 / sizeof(T) : 2)));  // float2; (unused for double)
};

// gather the boxes of X into the packed buffer (DIR = 0) or scatter the packed buffer into the boxes (DIR = 1); blockIdx.y = box
template <typename T, int DIR>
__global__ void __launch_bounds__(256)
box_copy_k(T* __restrict__ buf, T* __restrict__ X, BoxTable tab, int nkp, int nip, const int* __restrict__ skip) {
  if (skip && *skip) return;
  const BoxDesc d = tab.b[blockIdx.y];
  const size_t plane = (size_t)nkp * nip;
  if (d.kind == BOX_DIRECT) return;
  if (d.kind == BOX_ROWS) {
    // packed layout: the ni*nj padded rows of the box one after the other (nkp elements each, 16-byte aligned on both sides)
    constexpr int VW = 16 / sizeof(T);
    typedef typename Vec16<T>::type V;
    const int nv = nkp / VW, nrows = d.ni * d.nj;
    for (int row = blockIdx.x; row < nrows; row += gridDim.x) {
      const int i = row % d.ni, j = row / d.ni;
      T* xr = X + (size_t)(d.i0 + i) * nkp + (size_t)(d.j0 + j) * plane;
      T* br = buf + d.off + (size_t)row * nkp;
      for (int v = threadIdx.x; v < nv; v += 256) {
        if (DIR == 0) {
          reinterpret_cast<V*>(br)[v] = reinterpret_cast<const V*>(xr)[v];
        } else {
          const int kk = v * VW;
          if (kk >= d.k0 && kk + VW <= d.k0 + d.nk) {
            reinterpret_cast<V*>(xr)[v] = reinterpret_cast<const V*>(br)[v];
          } else {  // the vectors holding the k guide cells: those cells belong to the edge messages
#pragma unroll
            for (int c = 0; c < VW; c++)
              if (kk + c >= d.k0 && kk + c < d.k0 + d.nk) xr[kk + c] = br[kk + c];
          }
        }
      }
    }
    return;
  }
  if (d.kind == BOX_KPAIR) {
    // nk <= 2 cells per (i,j) row; packed layout [j][i][k].  Consecutive lanes take consecutive i: every lane its own 128-byte line of X
    // (nothing to coalesce on that side), but the packed side is one contiguous run per wave.
    const int nrows = d.ni * d.nj;
    for (int row = blockIdx.x * 256 + threadIdx.x; row < nrows; row += gridDim.x * 256) {
      const int i = row % d.ni, j = row / d.ni;
      T* xr = X + (size_t)d.k0 + (size_t)(d.i0 + i) * nkp + (size_t)(d.j0 + j) * plane;
      T* br = buf + d.off + (size_t)row * d.nk;
      // (8- / 16-byte accesses need every row start aligned: k0 even AND an even row length -- with nkp odd every second row is not)
      if (sizeof(T) == 4 && d.nk == 2 && (d.k0 & 1) == 0 && (d.off & 1) == 0 && (nkp & 1) == 0) {
        typedef typename Vec8<T>::type V2;
        if (DIR == 0) *reinterpret_cast<V2*>(br) = *reinterpret_cast<const V2*>(xr);
        else *reinterpret_cast<V2*>(xr) = *reinterpret_cast<const V2*>(br);
      } else if (sizeof(T) == 8 && d.nk == 2 && (d.k0 & 1) == 0 && (d.off & 1) == 0 && (nkp & 1) == 0) {
        typedef typename Vec16<T>::type V2;
        if (DIR == 0) *reinterpret_cast<V2*>(br) = *reinterpret_cast<const V2*>(xr);
        else *reinterpret_cast<V2*>(xr) = *reinterpret_cast<const V2*>(br);
      } else {
        for (int k = 0; k < d.nk; k++) {
          if (DIR == 0) br[k] = xr[k];
          else xr[k] = br[k];
        }
      }
    }
    return;
  }
  const long long n = (long long)d.ni * d.nj * d.nk;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long long)gridDim.x * 256) {
    const int k = (int)(e % d.nk);
    const long long r = e / d.nk;
    const int i = (int)(r % d.ni), j = (int)(r / d.ni);
    const size_t lin = (size_t)(d.k0 + k) + (size_t)(d.i0 + i) * nkp + (size_t)(d.j0 + j) * plane;
    if (DIR == 0) buf[d.off + e] = X[lin];
    else X[lin] = buf[d.off + e];
  }
}

}  // namespace

// one exchange pattern (depth + edges) of one brick
struct Pattern {
  int nmsg = 0;
  int peer[MAX_BOX];
  int dir[MAX_BOX][3];
  size_t count[MAX_BOX], off[MAX_BOX];
  int direct[MAX_BOX];                                // the message is a run of whole planes of the array (J face): no buffer
  size_t direct_send[MAX_BOX], direct_recv[MAX_BOX];  // element offsets of those runs in the array
  BoxTable send, recv;
  size_t total = 0;
  void* sendbuf = nullptr;
  void* recvbuf = nullptr;
};

struct CommCtx {
  Transport tr;
  int rank, nproc, eb;
  int size[3], nID[6];
  int coord[3], div[3];
  int g = 2;
  ncclComm_t nccl = nullptr;      // halo exchanges
  ncclComm_t nccl_red = nullptr;  // all-reduces (== nccl when the job asked for one communicator)
  LocalWorld* world = nullptr;
  CzConfig cfg;  // the environment as read when this communicator context was created (cz_config.h)
  double* h_red = nullptr;
  Pattern shallow, deep;   // depth 1 faces / depth 2 faces + edges
  const Pattern* cur = nullptr;  // LOCAL: pattern being exchanged (published for the neighbours)
  const void* cur_X = nullptr;   // LOCAL: the array being exchanged (source of the neighbours' direct messages)
  Watch watch;
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


namespace {
void build_pattern(CommCtx* c, Pattern& p, int depth, bool edges) {
  const int N[3] = {c->size[0], c->size[1], c->size[2]};
  const int g = c->g;
  p.nmsg = 0;
  p.total = 0;
  for (int dk = -1; dk <= 1; dk++)
    for (int dj = -1; dj <= 1; dj++)
      for (int di = -1; di <= 1; di++) {
        const int d[3] = {di, dj, dk};
        const int nz = (di != 0) + (dj != 0) + (dk != 0);
        if (nz == 0 || nz == 3 || (nz == 2 && !edges)) continue;
        int rc[3];
        bool exists = true;
        for (int a = 0; a < 3; a++) {
          rc[a] = c->coord[a] + d[a];
          if (rc[a] < 0 || rc[a] >= c->div[a]) exists = false;
        }
        if (!exists) continue;
        const int dep = (nz == 1) ? depth : 1;  // faces carry `depth` layers, edges one cell in each cut direction
        BoxDesc sb, rb;
        int s0[3], r0[3], ext[3];
        for (int a = 0; a < 3; a++) {
          if (d[a] == 0) {
            s0[a] = r0[a] = 1, ext[a] = N[a];
          } else if (d[a] < 0) {
            s0[a] = 1, r0[a] = 1 - dep, ext[a] = dep;
          } else {
            s0[a] = N[a] - dep + 1, r0[a] = N[a] + 1, ext[a] = dep;
          }
        }
        // 1-based (i,j,k) -> padded 0-based
        sb.i0 = s0[0] + g - 1, sb.j0 = s0[1] + g - 1, sb.k0 = s0[2] + g - 1;
        rb.i0 = r0[0] + g - 1, rb.j0 = r0[1] + g - 1, rb.k0 = r0[2] + g - 1;
        sb.ni = rb.ni = ext[0], sb.nj = rb.nj = ext[1], sb.nk = rb.nk = ext[2];
        const int nkp = N[2] + 2 * g, nip = N[0] + 2 * g;
        const bool vec_rows = (nkp * c->eb) % 16 == 0;  // every padded k-row starts 16-byte aligned (hipMalloc'ed arrays)
        int kind = BOX_GENERIC;
        size_t cnt = (size_t)ext[0] * ext[1] * ext[2];
        if (nz == 1 && dj != 0 && !c->cfg.has(CZV_COMM_PACK_J)) {
          kind = BOX_DIRECT;  // dep whole padded planes
          cnt = (size_t)dep * nkp * nip;
        } else if (nz == 1 && di != 0 && vec_rows) {
          kind = BOX_ROWS;  // whole padded rows
          cnt = (size_t)ext[0] * ext[1] * nkp;
        } else if (nz == 1 && dk != 0) {
          kind = BOX_KPAIR;
        }
        sb.kind = rb.kind = kind;
        const int m = p.nmsg++;
        p.direct[m] = (kind == BOX_DIRECT);
        p.direct_send[m] = (size_t)sb.j0 * nkp * nip;
        p.direct_recv[m] = (size_t)rb.j0 * nkp * nip;
        if (kind != BOX_DIRECT && (p.total * c->eb) % 16) p.total += (16 - (p.total * c->eb) % 16) / c->eb;  // 16-byte aligned messages
        sb.off = rb.off = (long long)p.total;
        p.send.b[m] = sb, p.recv.b[m] = rb;
        p.peer[m] = rc[0] + c->div[0] * (rc[1] + c->div[1] * rc[2]);
        p.dir[m][0] = di, p.dir[m][1] = dj, p.dir[m][2] = dk;
        p.count[m] = cnt;
        p.off[m] = p.total;
        if (kind != BOX_DIRECT) p.total += cnt;
      }
  p.send.n = p.recv.n = p.nmsg;
  if (p.total) {
    HIP_CHECK(hipMalloc(&p.sendbuf, p.total * c->eb));
    HIP_CHECK(hipMalloc(&p.recvbuf, p.total * c->eb));
  }
}

// `skip` (a device flag: the solve has converged) turns the pack and unpack launches into no-ops; the messages themselves are ALWAYS sent --
// collectives must never depend on device state the host has not read (rank lock step).  Packed messages then carry stale
// buffer contents that nobody unpacks.  Direct messages (J faces, received into the array itself) do write the ghost planes after
// convergence: what they write are the sender's owned cells of the converged iterate, which the skipped sweeps no longer change -- the
// same values the last exchange before convergence left there for every array the solver still reads.
template <typename T>
bool exchange(CommCtx* c, const Pattern& p, T* X, const int* skip, hipStream_t st) {
  if (p.nmsg == 0) return true;
  const int nkp = c->size[2] + 2 * c->g, nip = c->size[0] + 2 * c->g;
  // grid: enough workgroups for the biggest packed box (rows for the row kind, 256 cells otherwise)
  size_t most = 0;
  bool any_packed = false;
  for (int m = 0; m < p.nmsg; m++) {
    const BoxDesc& b = p.send.b[m];
    if (b.kind == BOX_DIRECT) continue;
    any_packed = true;
    const size_t rows = (size_t)b.ni * b.nj;
    most = std::max(most, b.kind == BOX_ROWS ? rows : b.kind == BOX_KPAIR ? (rows + 255) / 256 : (p.count[m] + 255) / 256);
  }
  const dim3 grid((unsigned)std::min<size_t>(std::max<size_t>(most, 1), 2048), (unsigned)p.nmsg);
  if (any_packed) {
    hipLaunchKernelGGL((box_copy_k<T, 0>), grid, dim3(256), 0, st, (T*)p.sendbuf, X, p.send, nkp, nip, skip);
    HIP_CHECK(hipGetLastError());
  }
  if (c->tr == T_RCCL) {
    const ncclDataType_t dt = sizeof(T) == 4 ? ncclFloat : ncclDouble;
    NCCL_CHECK(ncclGroupStart());
    for (int m = 0; m < p.nmsg; m++) {
      const T* src = p.direct[m] ? X + p.direct_send[m] : (const T*)p.sendbuf + p.off[m];
      T* dst = p.direct[m] ? X + p.direct_recv[m] : (T*)p.recvbuf + p.off[m];
      NCCL_CHECK(ncclSend(src, p.count[m], dt, p.peer[m], c->nccl, st));
      NCCL_CHECK(ncclRecv(dst, p.count[m], dt, p.peer[m], c->nccl, st));
    }
    NCCL_CHECK(ncclGroupEnd());
  } else {  // LOCAL: every rank has packed; copy what each neighbour packed for me (its message in direction -d)
    c->cur = &p;
    c->cur_X = X;
    HIP_CHECK(hipStreamSynchronize(st));
    c->world->barrier(c->rank);
    for (int m = 0; m < p.nmsg; m++) {
      const CommCtx* nb = c->world->ranks[p.peer[m]];
      const Pattern* q = nb->cur;
      int mm = -1;
      for (int x = 0; x < q->nmsg; x++)
        if (q->dir[x][0] == -p.dir[m][0] && q->dir[x][1] == -p.dir[m][1] && q->dir[x][2] == -p.dir[m][2]) mm = x;
      if (mm < 0 || q->count[mm] != p.count[m] || q->peer[mm] != c->rank || q->direct[mm] != p.direct[m]) {
        cz_fatal(1, "czhip: LOCAL transport: rank %d has no matching message from rank %d\n", c->rank, p.peer[m]);
      }
      const T* src = q->direct[mm] ? (const T*)nb->cur_X + q->direct_send[mm] : (const T*)q->sendbuf + q->off[mm];
      T* dst = p.direct[m] ? X + p.direct_recv[m] : (T*)p.recvbuf + p.off[m];
      HIP_CHECK(hipMemcpyAsync(dst, src, p.count[m] * sizeof(T), hipMemcpyDeviceToDevice, st));
    }
    HIP_CHECK(hipStreamSynchronize(st));
    c->world->barrier(c->rank);  // nobody repacks before everyone has copied
  }
  if (any_packed) {
    hipLaunchKernelGGL((box_copy_k<T, 1>), grid, dim3(256), 0, st, (T*)p.recvbuf, X, p.recv, nkp, nip, skip);
    HIP_CHECK(hipGetLastError());
  }
  c->watch.note(&p == &c->deep ? "halo exchange, two layers + edges" : "halo exchange, one layer", st);
  return true;
}
}  // namespace

CommCtx* comm_create(int rank, int nproc, const int size[3], const int nID[6], int elem_bytes, const int div[3]) {
  if (boot.tr == T_NONE || boot.nproc != nproc) {
    fprintf(stderr, "czhip: %d ranks requested but no communicator was bootstrapped (cz_comm_bootstrap*)\n", nproc);
    return nullptr;
  }
  CommCtx* c = new CommCtx();
  c->cfg = CzConfig::from_env();
  c->tr = boot.tr, c->rank = rank, c->nproc = nproc, c->eb = elem_bytes;
  c->nccl = boot.nccl, c->nccl_red = boot.nccl_red ? boot.nccl_red : boot.nccl, c->world = boot.world;
  for (int a = 0; a < 3; a++) c->size[a] = size[a], c->div[a] = div[a];
  for (int f = 0; f < 6; f++) c->nID[f] = nID[f];
  c->coord[0] = rank % div[0], c->coord[1] = (rank / div[0]) % div[1], c->coord[2] = rank / (div[0] * div[1]);
  build_pattern(c, c->shallow, 1, false);
  build_pattern(c, c->deep, 2, true);
  HIP_CHECK(hipHostMalloc(&c->h_red, 16 * sizeof(double), hipHostMallocDefault));
  if (c->tr == T_LOCAL) {
    std::lock_guard<std::mutex> lk(c->world->mu);
    c->world->ranks[rank] = c;
  }
  c->watch.start(rank, c->tr == T_RCCL ? comm_timeout_s(c->cfg, false) : 0.0, c->cfg.num(CZV_COMM_DEBUG, 0));  // (the LOCAL transport bounds its own host-side waits)
  return c;
}

void comm_destroy(CommCtx* c) {
  if (!c) return;
  c->watch.stop();
  for (Pattern* p : {&c->shallow, &c->deep}) {
    if (p->sendbuf) (void)hipFree(p->sendbuf);
    if (p->recvbuf) (void)hipFree(p->recvbuf);
  }
  (void)hipHostFree(c->h_red);
  delete c;
}

bool comm_halo(CommCtx* c, void* X, const int* skip, hipStream_t st) {
  if (!c) return true;
  return c->eb == 4 ? exchange<float>(c, c->shallow, (float*)X, skip, st) : exchange<double>(c, c->shallow, (double*)X, skip, st);
}

bool comm_halo2(CommCtx* c, void* X, const int* skip, hipStream_t st) {
  if (!c) return true;
  if (c->g != 2) return false;
  return c->eb == 4 ? exchange<float>(c, c->deep, (float*)X, skip, st) : exchange<double>(c, c->deep, (double*)X, skip, st);
}

bool comm_allreduce_sum(CommCtx* c, double* d_val, int count, hipStream_t st) {
  if (!c) return true;
  if (c->tr == T_RCCL) {
    NCCL_CHECK(ncclAllReduce(d_val, d_val, count, ncclDouble, ncclSum, c->nccl_red, st));
    c->watch.note("all-reduce (sum)", st);
    return true;
  }
  LocalWorld* w = c->world;
  HIP_CHECK(hipMemcpyAsync(c->h_red, d_val, count * sizeof(double), hipMemcpyDeviceToHost, st));
  HIP_CHECK(hipStreamSynchronize(st));
  for (int i = 0; i < count; i++) w->red[(size_t)c->rank * 16 + i] = c->h_red[i];
  w->barrier(c->rank);
  for (int i = 0; i < count; i++) {
    double s = 0.0;
    for (int r = 0; r < w->n; r++) s += w->red[(size_t)r * 16 + i];
    c->h_red[i] = s;
  }
  w->barrier(c->rank);
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
    NCCL_CHECK(ncclAllReduce(d, d, 1, ncclDouble, ncclMax, c->nccl_red, czhip_internal::stream()));
    c->watch.note("all-reduce (max)", czhip_internal::stream());
    HIP_CHECK(hipStreamSynchronize(czhip_internal::stream()));
    HIP_CHECK(hipMemcpy(&v, d, sizeof(double), hipMemcpyDeviceToHost));
    (void)hipFree(d);
    return v;
  }
  LocalWorld* w = c->world;
  w->red[(size_t)c->rank * 16] = v;
  w->barrier(c->rank);
  double m = v;
  for (int r = 0; r < w->n; r++) m = std::max(m, w->red[(size_t)r * 16]);
  w->barrier(c->rank);
  return m;
}

int comm_transport_ranks(const CommCtx* c) {
  if (!c || c->tr != T_RCCL || !c->nccl) return 0;
  int n = 0;
  NCCL_CHECK(ncclCommCount(c->nccl, &n));
  return n;
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
  boot.tr = T_RCCL, boot.rank = rank, boot.nproc = nranks, boot.nccl = comm, boot.nccl_red = nullptr, boot.world = nullptr;
  // The halo exchanges run on the exchange stream, the all-reduces of the non-lagged paths and of BiCGSTAB on the compute stream.  RCCL
  // orders the operations of ONE communicator across the streams they are issued on (it makes the later stream wait for the earlier
  // operation); the driver already orders them with events the way the algorithm needs, and a second communicator over the same ranks
  // keeps RCCL from adding an order of its own between an exchange and an all-reduce that have nothing to do with each other.  Every rank
  // issues the operations of both communicators in the same program order (CZ::JACOBI / RBSOR / PBiCGSTAB), which is what concurrent
  // communicators need.  CZ_COMM_ONE_COMM=1 keeps everything on one communicator.
  if (!CzConfig::from_env().on(CZV_COMM_ONE_COMM, false)) {
    ncclComm_t red = nullptr;
    NCCL_CHECK(ncclCommSplit(comm, 0, rank, &red, nullptr));
    boot.nccl_red = red;
  }
  return 0;
}

void cz_comm_shutdown(void) {
  if (boot.tr == T_RCCL && boot.nccl_red) NCCL_CHECK(ncclCommDestroy(boot.nccl_red));
  if (boot.tr == T_RCCL && boot.nccl) NCCL_CHECK(ncclCommDestroy(boot.nccl));
  boot = Boot();
}

// LOCAL transport: a world of n ranks living in n threads of this process
void* cz_comm_local_world(int n) {
  LocalWorld* w = new LocalWorld();
  w->n = n;
  w->wait_limit = comm_timeout_s(CzConfig::from_env(), true);
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
