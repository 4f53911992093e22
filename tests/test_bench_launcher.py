"""bench.py's rank bookkeeping without a GPU (--stub): `--gpus N` alone starts N ranks, a launcher's WORLD_SIZE must agree with
--gpus, and the printed line's n_gpus is the number of ranks that ran."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(args, env=None, timeout=240):
    e = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    e.update(env or {})
    return subprocess.run([sys.executable, BENCH] + args, env=e, capture_output=True, text=True, timeout=timeout)


@pytest.mark.parametrize("n", [1, 2, 3])
def test_gpus_flag_starts_that_many_ranks(n):
    r = _run(["--gpus", str(n), "--steps", "3", "--warmup", "1", "--stub"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, "exactly one JSON line (rank 0): %r" % r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == n and out["steps"] == 3 and out["warmup"] == 1 and out["scaling"] == "weak"


def test_launcher_world_size_must_match_gpus():
    r = _run(["--gpus", "2", "--stub"], env={"WORLD_SIZE": "1", "RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE" in r.stderr


def test_under_torchrun_style_environment():
    """What the driver does for N > 1: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N."""
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29631",
                        BENCH, "--gpus", "2", "--steps", "2", "--warmup", "0", "--stub"], capture_output=True, text=True, timeout=300,
                       env={k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")})
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1 and json.loads(lines[0])["n_gpus"] == 2


def test_a_failing_rank_ends_the_others_quickly():
    """Fail fast and collectively: rank 1 exits with code 3 before the rendezvous; the launcher stops the ranks that wait for it and
    returns non-zero well inside the rendezvous timeout."""
    import time
    t0 = time.monotonic()
    r = _run(["--gpus", "3", "--steps", "2", "--warmup", "0", "--stub", "--stub-fail-rank", "1"], timeout=60)
    assert r.returncode == 3, (r.returncode, r.stderr[-1000:])
    assert time.monotonic() - t0 < 10.0
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")], "no result line from a failed job"


def test_launcher_time_limit():
    """... and an overall time limit: rank 1 never reaches the rendezvous, rank 0 waits for it; after --launch-timeout seconds the
    launcher stops both and reports 124."""
    import time
    t0 = time.monotonic()
    r = _run(["--gpus", "2", "--steps", "2", "--warmup", "0", "--stub", "--stub-hang-rank", "1", "--launch-timeout", "3"], timeout=60)
    assert r.returncode == 124, (r.returncode, r.stderr[-1000:])
    assert time.monotonic() - t0 < 20.0
