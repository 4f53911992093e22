"""Bounded slices of the random-configuration stress runs (tools/stress_k3.py, tools/stress_k6.py, tools/stress_selfplay.py) inside the suite the driver runs: K3 and K6 against
the oracle on configurations drawn from a fixed random stream -- seeds, game ids, openings and late positions, rollout counts (all three lane
forms of the rollouts), playout counts, partly filled workgroups, second searches on the same evaluators -- for a fixed number of seconds each.
The stream is the same every run, so a failure names a configuration that reproduces."""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu

TOOLS = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools")
if TOOLS not in sys.path:
    sys.path.insert(0, TOOLS)

SLICE_SECONDS = float(os.environ.get("GMK_STRESS_SLICE", "25"))


def test_k3_stress_slice(oracle):
    import stress_k3
    configurations, games, bad, forms = stress_k3.run(SLICE_SECONDS, verbose=False)
    assert not bad, bad[:5]
    assert configurations >= 20 and games >= 300, (configurations, games)
    assert set(forms) == {"quads", "pairs", "one lane"}, forms                    # every lane form of the rollouts was met


def test_k6_stress_slice(oracle):
    import stress_k6
    searches, games, bad = stress_k6.run(SLICE_SECONDS, verbose=False)
    assert not bad, bad[:5]
    assert searches >= 6 and games >= 80, (searches, games)


def test_selfplay_stress_slice(oracle):
    """Whole games: the persistent self-play loops of K3 and K6 with the reference agent's semantics (kept subtree, root noise drawn on the device)
    against the oracle's kept-tree game loops on random configurations (tools/stress_selfplay.py; 1 029 runs / 10 765 games without a mismatch in
    profiles/r04_selfplay_stress_parity.txt)."""
    import stress_selfplay
    runs, games, bad = stress_selfplay.run(SLICE_SECONDS, verbose=False)
    assert not bad, bad[:5]
    assert runs >= 20 and games >= 150, (runs, games)
