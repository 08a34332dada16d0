// cz_main.cpp -- the `cz` command line, restating /root/reference/src/main.cpp:15-60 on top of the C-ABI.
//
//   ./cz gsz_x gsz_y gsz_z linear_solver IterationMax acc_coef [precond] [gdv_x gdv_y gdv_z]
//
// Multi-GPU: one process per GPU.  Instead of mpirun/MPI_Init (main.cpp:33-35) the ranks are started by any launcher
// that exports RANK, WORLD_SIZE and LOCAL_RANK (e.g. `python -m torch.distributed.run --no-python ...`); rank 0
// writes the RCCL unique id to $CZ_COMM_ID_FILE (default /tmp/cz_comm_id.$MASTER_PORT) and the others read it.
// The file starts with a job key -- $CZ_JOB_ID, or MASTER_ADDR:MASTER_PORT -- and the time rank 0 wrote it; a reader keeps polling until
// the key is its own and the record is not older than the reader itself (less a minute): a file left behind by an earlier job on the same
// port is not joined.  Rank 0 removes any old file first and creates the new one exclusively, mode 0600 (no symlink is followed).
#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <string>
#include <vector>

#include "cz_config.h"
#include "cz_hip.h"

static std::string id_file(const CzConfig& cfg) {
  if (cfg.has(CZV_COMM_ID_FILE)) return cfg.str(CZV_COMM_ID_FILE);
  return std::string("/tmp/cz_comm_id.") + (cfg.has(CZV_MASTER_PORT) ? cfg.str(CZV_MASTER_PORT) : "0");
}

int main(int argc, char* argv[]) {
  const CzConfig cfg = CzConfig::from_env();  // the launcher's variables and the switches, read once (cz_config.h)
  const int myRank = cfg.num(CZV_RANK, 0), nproc = cfg.num(CZV_WORLD_SIZE, 1);

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
    const std::string path = id_file(cfg);
    // The record: job key, creation time, id.  Job key: $CZ_JOB_ID, else MASTER_ADDR:MASTER_PORT -- values every rank of a job shares
    // whatever started it (one launcher, a wrapper script per rank, several nodes with the file on a shared file system).  A file a dead job
    // left behind on the same address and port is told apart by its age: rank 0 stamps the record, a reader takes a record only when the
    // stamp is not older than its own start minus a minute (ranks of one job start together; set CZ_JOB_ID where that does not hold).
    char key[256];
    if (cfg.has(CZV_JOB_ID)) snprintf(key, sizeof(key), "%s", cfg.str(CZV_JOB_ID));
    else snprintf(key, sizeof(key), "%s:%s", cfg.has(CZV_MASTER_ADDR) ? cfg.str(CZV_MASTER_ADDR) : "-", cfg.has(CZV_MASTER_PORT) ? cfg.str(CZV_MASTER_PORT) : "0");
    const long long started = (long long)time(nullptr);
    std::vector<char> rec(sizeof(key) + sizeof(long long) + nb, 0);
    bool ok = true;
    if (myRank == 0) {
      cz_comm_get_unique_id(id.data());
      const long long stamp = (long long)time(nullptr);
      memcpy(rec.data(), key, sizeof(key));
      memcpy(rec.data() + sizeof(key), &stamp, sizeof(stamp));
      memcpy(rec.data() + sizeof(key) + sizeof(stamp), id.data(), nb);
      const std::string tmp = path + ".tmp";
      unlink(path.c_str());  // a file an earlier job left behind
      unlink(tmp.c_str());
      const int fd = open(tmp.c_str(), O_WRONLY | O_CREAT | O_EXCL | O_NOFOLLOW, 0600);
      if (fd < 0 || write(fd, rec.data(), rec.size()) != (ssize_t)rec.size() || close(fd) != 0 || rename(tmp.c_str(), path.c_str()) != 0) {
        printf("\tcannot write %s\n", path.c_str());
        ok = false;
      }
    } else {
      ok = false;
      for (int tries = 0; tries < 600 && !ok; tries++) {  // 60 s
        const int fd = open(path.c_str(), O_RDONLY | O_NOFOLLOW);
        if (fd >= 0) {
          long long stamp = 0;
          if (read(fd, rec.data(), rec.size()) == (ssize_t)rec.size() && !strncmp(rec.data(), key, sizeof(key))) {
            memcpy(&stamp, rec.data() + sizeof(key), sizeof(stamp));
            if (stamp >= started - 60) ok = true;
          }
          close(fd);
        }
        if (!ok) usleep(100000);
      }
      if (ok) memcpy(id.data(), rec.data() + sizeof(key) + sizeof(long long), nb);
      else printf("\trank %d: no communicator id for job '%s' in %s after 60 s (the key is $CZ_JOB_ID or MASTER_ADDR:MASTER_PORT and must be the same on every rank)\n", myRank, key, path.c_str());
    }
    if (!ok) {
      czhip_finalize();
      return -1;
    }
    cz_comm_bootstrap(myRank, nproc, id.data());  // returns when every rank has joined
    if (myRank == 0) unlink(path.c_str());
  }

  cz_handle* cz = cz_create();
  cz_set_debug(cz, 1);  // main.cpp:38-42: debug mode is hard-wired on
  cz_set_profile(cz, cfg.num(CZV_PROFILE, 1));  // profiling.txt (cz_Evaluate.cpp:506-545) unless CZ_PROFILE=0
  if (0 == cz_evaluate(cz, argc, argv)) {  // main.cpp:45-52
    if (myRank == 0) printf("\n\tSolver error.\n\n");
    cz_destroy(cz);
    if (nproc > 1) cz_comm_shutdown();
    czhip_finalize();
    return -1;
  }
  cz_destroy(cz);
  if (nproc > 1) cz_comm_shutdown();
  czhip_finalize();
  return 0;
}
