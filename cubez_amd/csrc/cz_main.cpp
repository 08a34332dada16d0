// cz_main.cpp -- the `cz` command line, restating /root/reference/src/main.cpp:15-60 on top of the C-ABI.
//
//   ./cz gsz_x gsz_y gsz_z linear_solver IterationMax acc_coef [precond] [gdv_x gdv_y gdv_z]
//
// Multi-GPU: one process per GPU.  Instead of mpirun/MPI_Init (main.cpp:33-35) the ranks are started by any launcher
// that exports RANK, WORLD_SIZE and LOCAL_RANK (e.g. `python -m torch.distributed.run --no-python ...`); rank 0
// writes the RCCL unique id to $CZ_COMM_ID_FILE (default /tmp/cz_comm_id.$MASTER_PORT) and the others read it.
#include <unistd.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "cz_hip.h"

static std::string id_file() {
  const char* f = getenv("CZ_COMM_ID_FILE");
  if (f) return f;
  const char* port = getenv("MASTER_PORT");
  return std::string("/tmp/cz_comm_id.") + (port ? port : "0");
}

int main(int argc, char* argv[]) {
  int myRank = 0, nproc = 1;
  if (getenv("RANK")) myRank = atoi(getenv("RANK"));
  if (getenv("WORLD_SIZE")) nproc = atoi(getenv("WORLD_SIZE"));

  if (argc != 7 && argc != 8 && argc != 10 && argc != 11) {  // main.cpp:19-31
    if (myRank == 0) {
      printf("\tUsage : ./cz gsz_x, gsz_y, gsz_z, linear_solver, IterationMax, acc_coef [precond] [gdv_x, gdv_y, gdv_z]\n");
      printf("\t\tlinear_solver = {jacobi | psor | sor2sma | pbicgstab | pcr | pcr_eda | pcr_esa | pcr_rb | pcr_rb_esa | pcr_j_esa | jacobi_maf | psor_maf | sor2sma_maf | pbicgstab_maf | pcr_maf | pcr_eda_maf | pcr_esa_maf | pcr_rb_maf | pcr_rb_esa_maf}\n");
      printf("\t\tprecond = {none | jacobi | psor | sor2sma | pcr | pcr_eda | pcr_rb | pcr_rb_esa | pcr_j_esa | jacobi_maf | psor_maf | sor2sma_maf | pcr_maf | pcr_eda_maf | pcr_rb_maf | pcr_rb_esa_maf}\n\n");
      printf("\t$ ./cz 64 64 64 jacobi 4000 0.8 2 2 1\n");
      printf("\t$ ./cz 64 64 64 sor2sma 4000 1.5\n");
      printf("\t$ ./cz 64 64 64 pbicgstab 4000 1.1 sor2sma\n");
      printf("\t$ ./cz 64 64 64 pbicgstab 4000 1.1 sor2sma 2 1 3\n");
    }
    return 0;
  }

  setenv("HSA_ENABLE_IPC_MODE_LEGACY", "0", 0);  // dmabuf IPC between the ranks' processes (RCCL); must be set before HIP starts
  czhip_init(-1);  // LOCAL_RANK selects the GPU
  if (nproc > 1) {
    const int nb = cz_comm_unique_id_bytes();
    std::vector<char> id(nb);
    const std::string path = id_file();
    if (myRank == 0) {
      cz_comm_get_unique_id(id.data());
      const std::string tmp = path + ".tmp";
      FILE* f = fopen(tmp.c_str(), "wb");
      if (!f || fwrite(id.data(), 1, nb, f) != (size_t)nb) {
        printf("\tcannot write %s\n", tmp.c_str());
        return -1;
      }
      fclose(f);
      rename(tmp.c_str(), path.c_str());
    } else {
      FILE* f = nullptr;
      for (int tries = 0; tries < 600 && !(f = fopen(path.c_str(), "rb")); tries++) usleep(100000);
      if (!f || fread(id.data(), 1, nb, f) != (size_t)nb) {
        printf("\trank %d: cannot read %s\n", myRank, path.c_str());
        return -1;
      }
      fclose(f);
    }
    cz_comm_bootstrap(myRank, nproc, id.data());
    if (myRank == 0) {
      usleep(2000000);
      unlink(path.c_str());
    }
  }

  cz_handle* cz = cz_create();
  cz_set_debug(cz, 1);  // main.cpp:38-42: debug mode is hard-wired on
  {
    const char* pf = getenv("CZ_PROFILE");  // profiling.txt (cz_Evaluate.cpp:506-545) unless CZ_PROFILE=0
    cz_set_profile(cz, pf ? atoi(pf) : 1);
  }
  if (0 == cz_evaluate(cz, argc, argv)) {  // main.cpp:45-52
    if (myRank == 0) printf("\n\tSolver error.\n\n");
    return -1;
  }
  cz_destroy(cz);
  if (nproc > 1) cz_comm_shutdown();
  czhip_finalize();
  return 0;
}
