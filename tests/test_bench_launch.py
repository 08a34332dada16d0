"""bench.py starts its own ranks: `python bench.py --gpus N` without a launcher spawns
`python -m torch.distributed.run ... bench.py <same args>` as a child, relays rank 0's JSON line and the child's exit code.
--dry-launch stops after the rendezvous and the broadcast of the communicator id, so the whole launch path runs here without a GPU
(the id is a placeholder then: ncclGetUniqueId needs a device)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*extra, timeout=300):
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(extra), capture_output=True, text=True, timeout=timeout, env=env)


def test_dry_launch_of_two_ranks_prints_one_json_line():
    r = _run("--gpus", "2", "--dry-launch")
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["dry_launch"] is True and out["n_gpus"] == 2
    assert out["id_agrees_on_all_ranks"] is True
    assert out["division"] == [1, 2, 1] and out["global_grid"] == [512, 1024, 512]


def test_a_failing_rank_gives_a_nonzero_exit_code_and_no_line():
    r = _run("--gpus", "2", "--dry-launch", "--div", "1,3,1")  # 3 bricks for 2 ranks: every rank exits with an error
    assert r.returncode != 0
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
