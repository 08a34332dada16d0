// cz_comm.h -- domain decomposition + halo exchange + all-reduce (replaces CBrick / MPI of the reference:
// cz_Evaluate.cpp:103-159, cz_comm.cpp:23-38 Comm_S, :102-120 Comm_SUM_1).
//
// Two transports behind one interface:
//   RCCL   one process per GPU; grouped ncclSend/ncclRecv of the six faces over xGMI + ncclAllReduce.
//   LOCAL  several ranks as host threads of ONE process sharing one GPU, faces moved by device-to-device copies.
//          Exists so that the complete decomposed solver path can be checked bit-for-bit against the single-domain
//          run on a one-GPU box (tests/test_gpu_decomp.py); it is not a performance path.
#ifndef CZ_COMM_H_
#define CZ_COMM_H_

#include <hip/hip_runtime.h>

struct CommCtx;

// rank / size the calling thread was bootstrapped with (0,1 when it was not)
void comm_world(int* rank, int* nproc);
// 3-D division of nproc minimising the exchanged surface; ties prefer cuts along j (contiguous faces), then i, then k
void comm_auto_division(int nproc, const int G_size[3], int G_div[3]);
// cell-ownership decomposition: local size, 1-based global head, neighbour table (I-,I+,J-,J+,K-,K+; -1 = physical)
bool comm_decompose(const int G_size[3], const int G_div[3], int nproc, int rank, int size[3], int head[3], int nID[6]);

CommCtx* comm_create(int rank, int nproc, const int size[3], const int nID[6], int elem_bytes, const int div[3]);
void comm_destroy(CommCtx*);
// one-layer exchange of the six faces of X (device pointer), stream-ordered on `st`
bool comm_halo(CommCtx*, void* X, const int* skip_flag_dev, hipStream_t st);
// two ghost layers on the faces + the 12 edges, single phase (diagonal neighbours get the edges directly); needs g == 2
bool comm_halo2(CommCtx*, void* X, const int* skip_flag_dev, hipStream_t st);
bool comm_allreduce_sum(CommCtx*, double* d_val, int count, hipStream_t st);
double comm_allreduce_max_host(CommCtx*, double v);
// ranks of the RCCL communicator behind the context (ncclCommCount); 0 for the LOCAL transport or no context
int comm_transport_ranks(const CommCtx*);

#endif
