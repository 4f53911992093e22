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


@pytest.mark.parametrize("n", [1, 2, 3])
def test_selfplay_scaling_flags_and_leg_reduction_over_gloo(n):
    """--selfplay-games (the strong form: games IN TOTAL), --selfplay-games-per-gpu (the weak form) and --selfplay-rehearsal-ranks (one GPU: the shard a
    rank gets at that N), as n ranks see them, and the one-collective reduction that ends a leg (MAX of the times, SUM of the counts) over gloo."""
    r = _run(["--gpus", str(n), "--steps", "1", "--warmup", "0", "--stub", "--selfplay-games", "1000", "--selfplay-games-per-gpu", "300", "--selfplay-rehearsal-ranks", "8"])
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert out["reduce_leg"] == {"times_max": [float(n), 0.25], "counts_sum": [n * (n + 1) // 2, 10 * n]}
    sp = out["selfplay_pipeline"]
    assert sp["strong_total"] == 1000 and sp["weak_total"] == 300 * n
    assert sp["games_over_all_shards"] == {"strong": 1000, "weak": 300 * n}          # every game on exactly one rank, in both forms
    assert sp["rank0_shard"]["strong"] == [0, 1000 // n] and sp["rank0_shard"]["weak"] == [0, 300]
    assert sp["rehearsal_games"] == (125 if n == 1 else None)


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
