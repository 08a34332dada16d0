"""GPU test (-m gpu) of the `cz` command line (cubez_amd/cz_f32, cz_f64): the reference's usage, stdout lines, history file
and exit codes (src/main.cpp:15-60, cz_Evaluate.cpp:210-218, 492-496, 558; cz_Poisson.cpp:71)."""
import json
import os
import re
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
GOLDEN = os.path.join(ROOT, "tests", "golden")
CASES = {c["tag"]: c for c in json.load(open(os.path.join(GOLDEN, "solver_cases.json")))}


def _run(prec, args, cwd, env=None):
    exe = os.path.join(ROOT, "cubez_amd", f"cz_{prec}")
    assert os.path.exists(exe), "build the CLI: make -C cubez_amd/csrc"
    return subprocess.run([exe] + [str(a) for a in args], cwd=cwd, capture_output=True, text=True, timeout=300,
                          env=dict(os.environ, **(env or {})))


def test_usage_on_wrong_argc(tmp_path):
    r = _run("f32", [64, 64, 64], tmp_path)
    assert r.returncode == 0 and "Usage : ./cz" in r.stdout and "linear_solver" in r.stdout


def test_invalid_solver_exits_like_the_reference(tmp_path):
    r = _run("f32", [32, 32, 32, "lsor_simd", 10, 1.0], tmp_path)   # a name the reference CLI does not know either
    assert r.returncode == 0 and "Invalid solver" in r.stdout


@pytest.mark.parametrize("tag", ["jacobi_32x32x32_f32", "sor2sma_32x32x32_f64", "jacobi_48x40x36_f32", "pbicgstab_jacobi_64x64x64_f64",
                                 "jacobi_maf_32x32x32_f32", "pbicgstab_maf_sor2sma_maf_32x32x32_f64", "psor_32x32x32_f32",
                                 "pcr_rb_32x32x32_f32", "pcr_rb_esa_32x32x32_f32", "pcr_j_esa_32x32x32_f64", "pcr_32x32x32_f32",
                                 "pbicgstab_psor_32x32x32_f64", "pbicgstab_pcr_rb_esa_32x32x32_f64"])
def test_cli_matches_reference_run(tmp_path, tag):
    c = CASES[tag]
    args = list(c["gsz"]) + [c["solver"], c["itr_max"], c["coef"]] + ([c["precond"]] if c["precond"] else [])
    r = _run(c["prec"], args, tmp_path)
    assert r.returncode == 0, r.stdout + r.stderr
    m = re.search(r"Iter = (\d+)  Res = ([0-9.e+-]+)", r.stdout)
    assert m, r.stdout
    assert int(m.group(1)) == c["iter"]
    tol = 1e-3 if c["prec"] == "f32" else 1e-6
    assert abs(float(m.group(2)) - c["res"]) <= tol * c["res"]
    # history file: same name, header and line format as the reference
    hist = open(os.path.join(tmp_path, f"{c['solver']}.txt")).read().splitlines()
    ref = open(os.path.join(GOLDEN, f"hist_{tag}.txt")).read().splitlines()
    assert hist[0] == ref[0] == "Itration      Residual"
    assert len(hist) == len(ref)
    for a, b in zip(hist[1:], ref[1:]):
        assert re.fullmatch(r" *\d+, +[0-9.]+e[+-]\d\d", a)
        ia, ra = a.split(",")
        ib, rb = b.split(",")
        assert int(ia) == int(ib) and abs(float(ra) - float(rb)) <= max(tol, 2e-6) * float(rb)
    # debug epilogue (main.cpp hard-wires debug mode): analytic max error
    m = re.search(r"Error max = ([0-9.e+-]+) at \((\d+) (\d+) (\d+)\)", r.stdout)
    assert m, r.stdout
    assert abs(float(m.group(1)) - c["errmax"]) <= 1e-5 * c["errmax"]
    if not c["solver"].startswith("pbicgstab"):
        assert [int(m.group(i)) for i in (2, 3, 4)] == c["errloc"]


@pytest.mark.parametrize("tag", ["jacobi_32x32x32_f32", "sor2sma_32x32x32_f64"])
def test_cli_writes_the_reference_sph_files(tmp_path, tag):
    """debug epilogue (cz_Evaluate.cpp:553-561): p_00000.sph / e_00000.sph, the format of fileout_t (cz_utility.f90:33-44,
    compiled into the reference only with -D_aurora_=1; CZ_SPH=1 here).  Fields are bit-identical => so are the files."""
    import numpy as np
    from oracle import cz_oracle as O
    c = CASES[tag]
    r = _run(c["prec"], list(c["gsz"]) + [c["solver"], c["itr_max"], c["coef"]], tmp_path, env={"CZ_SPH": "1"})
    assert r.returncode == 0, r.stdout + r.stderr
    k = O.Kernels("oracle", c["prec"])
    R = k.real
    pitch = R(1.0 / float(R(c["gsz"][2] - 1)))  # cz_Evaluate.cpp:88, as oracle/cz_oracle.py
    org = np.zeros(3, dtype=R)
    P = np.load(os.path.join(GOLDEN, c["field"]))
    assert open(os.path.join(tmp_path, "p_00000.sph"), "rb").read() == O.sph_bytes(c["gsz"], P, pitch, org)
    e = k.alloc(c["gsz"])
    k.exact_t(c["gsz"], e, pitch, org)
    assert open(os.path.join(tmp_path, "e_00000.sph"), "rb").read() == O.sph_bytes(c["gsz"], e, pitch, org)


def test_cli_writes_no_sph_by_default(tmp_path):
    r = _run("f32", [16, 16, 16, "jacobi", 4, 0.8], tmp_path)
    assert r.returncode == 0 and not [f for f in os.listdir(tmp_path) if f.endswith(".sph")]


def test_cli_writes_the_section_report(tmp_path):
    """profiling.txt (cz_Evaluate.cpp:506-545): PMlib's basic-report table, sections under the reference's labels."""
    r = _run("f64", [32, 32, 32, "pbicgstab", 50, 0.8, "jacobi"], tmp_path)
    assert r.returncode == 0, r.stdout + r.stderr
    txt = open(os.path.join(tmp_path, "profiling.txt")).read()
    assert "# PMlib Basic Report" in txt and "Total execution time" in txt
    rows = {}
    for ln in txt.splitlines():
        m = re.match(r"\t(.{33}): +(\d+) +([0-9.e+-]+) +([0-9.]+) ", ln)
        if m:
            rows[m.group(1).strip()] = (int(m.group(2)), float(m.group(3)), float(m.group(4)))
    assert {"Blas_AX", "Dot1 / Dot2", "Blas_Residual"} <= set(rows), rows
    assert rows["Blas_AX"][0] == 2 * 9            # two SpMV per iteration, 9 iterations (golden: Iter = 9)
    assert rows["Blas_Residual"][0] == 1          # r = b - Ax once
    assert abs(sum(v[2] for v in rows.values()) - 100.0) < 0.5


def test_cli_profile_can_be_switched_off(tmp_path):
    r = _run("f32", [16, 16, 16, "jacobi", 4, 0.8], tmp_path, env={"CZ_PROFILE": "0"})
    assert r.returncode == 0 and not os.path.exists(os.path.join(tmp_path, "profiling.txt"))


@pytest.mark.parametrize("prec,args", [("f32", [37, 29, 61, "jacobi", 40, 0.8]), ("f32", [37, 29, 61, "sor2sma", 40, 1.5]),
                                        ("f64", [29, 33, 63, "pbicgstab", 30, 0.8, "jacobi"]), ("f64", [21, 26, 65, "jacobi_maf", 30, 0.8])],
                         ids=lambda v: v if isinstance(v, str) else "_".join(map(str, v)))
def test_scalar_kernels_of_rounds_1_and_2_give_the_history_of_the_vector_kernels(tmp_path, prec, args):
    """Rows that are no multiple of the vector width took the scalar kernels (V = 1) until round 3; CZHIP_T2_ROWS=0 still sends them there (the
    'before' of profiles/r03/unaligned_k_extent.txt).  Two independent code paths for the same arithmetic: the history files must agree --
    byte for byte for the stationary solvers (double-accumulated residuals in a fixed tree differ only in tree shape: compare to 1e-12)."""
    out = {}
    for rows in ("1", "0"):
        d = tmp_path / rows
        d.mkdir()
        r = _run(prec, args, d, env={"CZHIP_T2_ROWS": rows})
        assert r.returncode == 0, r.stdout + r.stderr
        m = re.search(r"Iter = (\d+)  Res = ([0-9.e+-]+)", r.stdout)
        e = re.search(r"Error max = ([0-9.e+-]+) at \((\d+) (\d+) (\d+)\)", r.stdout)
        assert m and e, r.stdout
        out[rows] = (int(m.group(1)), float(m.group(2)), e.groups(), open(os.path.join(d, f"{args[3]}.txt")).read().splitlines())
    assert out["1"][0] == out["0"][0]
    assert out["1"][2] == out["0"][2]  # the analytic error of the final field, as printed: the fields agree
    tol = 1e-12 if args[3] != "pbicgstab" else 1e-6
    assert abs(out["1"][1] - out["0"][1]) <= max(tol, 2e-6) * out["0"][1]  # (the printed residual carries 7 digits)
    assert len(out["1"][3]) == len(out["0"][3])
    for a, b in zip(out["1"][3][1:], out["0"][3][1:]):
        ra, rb = float(a.split(",")[1]), float(b.split(",")[1])
        assert abs(ra - rb) <= 2e-6 * rb
