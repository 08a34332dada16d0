// cz_config.h -- every environment variable the library and the `cz` command line read, in ONE table, read by ONE function.
//
// The reference has no configuration besides its command line (main.cpp:15-60); what is here are the launcher's variables of a multi-process
// run (RANK, WORLD_SIZE, ... instead of mpirun) and the switches of this implementation's own choices (A/B partners of measurements, test
// aids).  Round 3 had 31 names read by 36 getenv calls scattered over five files.  Now: CzConfig::from_env() is the only
// place that asks the environment; a consumer parses once per object it configures -- czhip_init for the per-thread kernel context, CZ::CZ for
// a driver, cz_comm_bootstrap / the transport for a communicator, main() for the command line -- and keeps the copy (tests change the
// environment between two drivers of one process, so "once per process" would be wrong).  czhip_config_describe() (include/cz_hip.h) returns
// the table with the values in force; CZ::Setup prints it under CZ_COMM_DEBUG.
#ifndef CZ_CONFIG_H_
#define CZ_CONFIG_H_

#include <cstdio>
#include <cstdlib>
#include <string>

enum CzVar {
  // ---- launcher (one process per GPU; set by torch.distributed.run, srun, a wrapper script ...)
  CZV_RANK, CZV_WORLD_SIZE, CZV_LOCAL_RANK, CZV_MASTER_ADDR, CZV_MASTER_PORT, CZV_JOB_ID, CZV_COMM_ID_FILE,
  // ---- driver (CZ)
  CZV_COMM_DEBUG, CZV_OVERLAP, CZV_LAG_REDUCE, CZV_COMM_CUS, CZV_BICG_FUSE, CZV_BICG_DEVSC, CZV_BICG_ALIAS, CZV_SPH, CZV_PROFILE, CZV_TEST_SKEW,
  // ---- transport (cz_comm.cpp)
  CZV_COMM_TIMEOUT, CZV_COMM_PACK_J, CZV_COMM_ONE_COMM,
  // ---- kernels (czhip_init)
  CZV_TUNING, CZV_T2, CZV_T2_MAP, CZV_T2_ROWS, CZV_T2_KWIN, CZV_T2_PRE, CZV_RB4, CZV_UNIT_COEF, CZV_FUSE_FIN, CZV_PCR, CZV_PCR_PIPE, CZV_PCR_PIPE_PROF, CZV_PCR_WG_PER_CU, CZV_PCR_MAX_WG, CZV_PCR_SLOTS, CZV_PSOR,
  // ---- diagnostics
  CZV_FATAL_LOG,
  CZV_COUNT
};

struct CzVarDef {
  const char* name;
  const char* dflt;  // what an unset variable means (text, for the table)
  const char* what;
};

inline const CzVarDef* cz_var_defs() {
  static const CzVarDef defs[CZV_COUNT] = {
      {"RANK", "0", "rank of this process (launcher)"},
      {"WORLD_SIZE", "1", "number of ranks (launcher)"},
      {"LOCAL_RANK", "0", "selects the GPU: device = LOCAL_RANK mod device count (launcher)"},
      {"MASTER_ADDR", "-", "with MASTER_PORT: the job key of the communicator-id record (cz command line)"},
      {"MASTER_PORT", "0", "see MASTER_ADDR; also names the default id file /tmp/cz_comm_id.<port>"},
      {"CZ_JOB_ID", "", "job key of the communicator-id record instead of MASTER_ADDR:MASTER_PORT"},
      {"CZ_COMM_ID_FILE", "", "where rank 0 of the cz command line leaves the RCCL unique id for the other ranks"},
      {"CZ_COMM_DEBUG", "0", "one line per rank about what a decomposed run decided + the collective watchdog (300 s)"},
      {"CZ_OVERLAP", "1", "decomposed fused passes: shell slabs + exchange on a second stream beside the interior"},
      {"CZ_LAG_REDUCE", "1", "decomposed checked runs: residual all-reduce and test one pass behind, on the exchange stream"},
      {"CZ_COMM_CUS", "2", "CUs per XCD the interior launch of a decomposed pass leaves to the exchange stream (launch geometry)"},
      {"CZ_BICG_FUSE", "1", "BiCGSTAB: the vector updates that make a preconditioner solve's right-hand side are made by its first pass"},
      {"CZ_BICG_DEVSC", "1", "BiCGSTAB: alpha and omega made on the device behind their dot products (one host wait per iteration)"},
      {"CZ_BICG_ALIAS", "1", "BiCGSTAB without a preconditioner: the solves read p and s themselves instead of cleared-and-copied p_, s_"},
      {"CZ_SPH", "0", "write p_%05d.sph / e_%05d.sph like the reference's -D_aurora_=1 build (cz_utility.f90:17-47)"},
      {"CZ_PROFILE", "1", "cz command line: write profiling.txt (cz_Evaluate.cpp:506-545)"},
      {"CZ_TEST_SKEW", "", "test aid \"rank,milliseconds\": that rank sleeps before every look at the convergence flag"},
      {"CZ_COMM_TIMEOUT", "", "seconds a collective may stay incomplete (default: 300 under CZ_COMM_DEBUG, 120 LOCAL transport, else none; 0 = wait for ever)"},
      {"CZ_COMM_PACK_J", "0", "A/B: J faces through pack buffers instead of sent from / received into the array"},
      {"CZ_COMM_ONE_COMM", "0", "one RCCL communicator for halos and all-reduces instead of two"},
      {"CZHIP_TUNING", "", "stencil_k shape \"threads,m,tj,pf\""},
      {"CZHIP_T2", "", "two-stage pass \"enable[,threads,2,tj]\" (0 = chosen per launch by pair_tj_model)"},
      {"CZHIP_T2_MAP", "1", "two-stage pass: balanced (segment, chunk) table per XCD where bands would idle"},
      {"CZHIP_T2_ROWS", "1", "vector kernels take rows whose length is no multiple of the vector width"},
      {"CZHIP_T2_KWIN", "", "two-stage pass: vectors per k window (0 = whole rows where they fit, -1 = chosen per launch)"},
      {"CZHIP_T2_PRE", "1", "two-stage pass on small grids: all operands of a chunk requested before its first plane step (jacobi2p_k<PRE>)"},
      {"CZHIP_RB4", "1", "red-black SOR, single domain: two iterations per pass over memory (rb4_k) \"enable[,vectors per window[,planes per chunk]]\""},
      {"CZHIP_UNIT_COEF", "1", "jacobi2p_k / rb4_k: the form without the six multiplications where the six coefficients are exactly 1 (same bits)"},
      {"CZHIP_FUSE_FIN", "1", "residual finalised by the last workgroup of the sweep (0: separate reduce + check launches)"},
      {"CZHIP_PCR", "", "line SOR form \"fast[,variant]\": 0 literal, 1 table + d in LDS, 2 table + d in registers"},
      {"CZHIP_PCR_PIPE", "", "lexicographic line SOR \"one_launch[,timeout s[,groups[,rows per thread]]]\""},
      {"CZHIP_PCR_PIPE_PROF", "", "development aid (-DCZ_LEX_PROF builds): file for the strip timeline"},
      {"CZHIP_PCR_WG_PER_CU", "0", "pcr_lex_wg_k: workgroups per CU (0 = launcher's choice)"},
      {"CZHIP_PCR_MAX_WG", "0", "pcr_lex_wg_k: workgroups in all (0 = launcher's choice)"},
      {"CZHIP_PCR_SLOTS", "0", "pcr_lex_wg_k: lines per hand-off ring (0 = launcher's choice)"},
      {"CZHIP_PSOR", "", "point SOR \"one_launch[,workgroups per CU]\""},
      {"CZ_FATAL_LOG", "", "file every fatal exit of the library appends its message to (besides stderr)"},
  };
  return defs;
}

struct CzConfig {
  bool set[CZV_COUNT];
  std::string val[CZV_COUNT];

  static CzConfig from_env() {
    CzConfig c;
    const CzVarDef* d = cz_var_defs();
    for (int v = 0; v < CZV_COUNT; v++) {
      const char* e = getenv(d[v].name);  // <- the one place the library asks the environment (cz_fatal's CZ_FATAL_LOG aside: it must not allocate)
      c.set[v] = e != nullptr;
      c.val[v] = e ? e : "";
    }
    return c;
  }
  bool has(CzVar v) const { return set[v]; }
  const char* str(CzVar v) const { return set[v] ? val[v].c_str() : nullptr; }
  int num(CzVar v, int dflt) const { return set[v] ? atoi(val[v].c_str()) : dflt; }
  double real(CzVar v, double dflt) const { return set[v] ? atof(val[v].c_str()) : dflt; }
  bool on(CzVar v, bool dflt) const { return set[v] ? atoi(val[v].c_str()) != 0 : dflt; }
  // "NAME=value" for what is set, "NAME (default: ...)" otherwise, one per line
  std::string describe(bool only_set = false) const {
    std::string s;
    const CzVarDef* d = cz_var_defs();
    for (int v = 0; v < CZV_COUNT; v++) {
      if (set[v]) s += std::string(d[v].name) + "=" + val[v] + "\n";
      else if (!only_set) s += std::string(d[v].name) + " (unset; default " + (d[v].dflt[0] ? d[v].dflt : "none") + ")\n";
    }
    return s;
  }
};

#endif
