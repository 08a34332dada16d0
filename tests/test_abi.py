"""CPU-side checks of the drop-in boundary: the libraries load without a GPU and export every symbol
include/cz_hip.h declares (no compute calls here)."""
import ctypes
import os
import re

import pytest

from cubez_amd import lib

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def _declared_symbols():
    src = open(os.path.join(ROOT, "include", "cz_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b([a-z_0-9]+)\s*\(", src)) - {"defined", "sizeof"})


@pytest.mark.parametrize("prec", ["f32", "f64"])
def test_library_exports_every_declared_symbol(prec):
    path = lib.lib_path(prec)
    assert os.path.exists(path), f"{path} not built: run __graft_entry__.build()"
    h = ctypes.CDLL(path)
    declared = _declared_symbols()
    assert len(declared) >= 40
    missing = [s for s in declared if not hasattr(h, s)]
    assert not missing, missing
    assert sorted(lib.ABI_SYMBOLS) == declared
    assert h.czhip_real_bytes() == (4 if prec == "f32" else 8)
    h.czhip_arch.restype = ctypes.c_char_p
    assert h.czhip_arch() == b"gfx950"


def test_missing_library_fails_loudly(monkeypatch):
    monkeypatch.setattr(lib, "_HERE", "/nonexistent")
    monkeypatch.setattr(lib, "_LOADED", {})
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        lib.load("f32")


def test_a_fatal_exit_of_the_library_leaves_a_line(tmp_path):
    """Every fatal exit inside the library goes through cz_fatal (cz_internal.h): message on stderr, appended to $CZ_FATAL_LOG, streams
    flushed, exit code 1.  Without a GPU the first such exit is czhip_init's "no HIP device" -- which is also the statement that the
    library has no CPU fallback.  (With a GPU present the call succeeds: the check is then that nothing is logged.)"""
    import subprocess
    import sys
    log = tmp_path / "fatal.log"
    env = dict(os.environ, CZ_FATAL_LOG=str(log))
    code = "import sys; sys.path.insert(0, %r); import cubez_amd; print('rc', cubez_amd.load('f32').czhip_init(0), flush=True)" % ROOT
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    if os.path.exists("/dev/kfd") and r.returncode == 0:
        assert "rc 0" in r.stdout and not log.exists()
        return
    assert r.returncode == 1, (r.returncode, r.stdout, r.stderr)
    assert "no HIP device available" in r.stderr and "no CPU fallback" in r.stderr
    text = log.read_text()
    assert "no HIP device available" in text and "exit 1" in text


def test_the_configuration_table_is_complete_and_read_in_one_place():
    """cz_config.h: every environment variable the library reads is a row of one table, read by one function; czhip_config_describe lists them
    with the values in force (no GPU needed).  And nothing else in the library asks the environment (cz_fatal's CZ_FATAL_LOG aside)."""
    import subprocess
    h = ctypes.CDLL(lib.lib_path("f32"))
    h.czhip_config_describe.restype = ctypes.c_char_p
    os.environ["CZ_COMM_CUS"] = "3"
    try:
        text = h.czhip_config_describe(0).decode()
        only = h.czhip_config_describe(1).decode()
    finally:
        os.environ.pop("CZ_COMM_CUS")
    assert "CZ_COMM_CUS=3" in text and "CZ_COMM_CUS=3" in only
    for name in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "CZ_OVERLAP", "CZ_LAG_REDUCE", "CZ_BICG_FUSE", "CZ_BICG_DEVSC", "CZ_BICG_ALIAS", "CZHIP_T2", "CZHIP_T2_KWIN",
                 "CZHIP_T2_PRE", "CZHIP_PSOR", "CZHIP_PCR_PIPE", "CZ_COMM_TIMEOUT", "CZ_FATAL_LOG"):
        assert name in text, name
    assert "CZ_COMM_CUS_MASK" not in text  # the CU-mask path was removed in round 4
    src = os.path.join(ROOT, "cubez_amd", "csrc")
    hits = subprocess.run(["grep", "-rn", "getenv(", src], capture_output=True, text=True).stdout.splitlines()
    hits = [l for l in hits if not l.split(":", 2)[2].lstrip().startswith("//")]
    assert len(hits) == 2 and any("cz_config.h" in l for l in hits) and any("cz_internal.h" in l for l in hits), hits
