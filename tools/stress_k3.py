"""K3 against the oracle on randomly drawn configurations: seeds, game numbers, opening lengths (and late no-five positions), rollout counts
(which pick the lane form of the rollouts: quads, pairs, one lane), playout counts.  Visit counts, the bits of the root value and the tree size
of every game must be equal.  tools/stress_k3.py [seconds]; run(budget, seed) is what tests/test_stress_gpu.py calls for a bounded slice"""
import ctypes as C
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gomokuai_amd import lib as G
from oracle import oracle as O

cls = lambda c: ((c % 15) // 2 + c // 15) % 2
blacks, whites = [c for c in range(225) if cls(c) == 0], [c for c in range(225) if cls(c) == 1]


def late_positions(rng, n):
    moves = np.zeros((n, 225), dtype=np.uint8); lens = np.zeros(n, dtype=np.int32)
    for g in range(n):
        b, w = list(rng.permutation(blacks)), list(rng.permutation(whites))
        seq = []
        while b or w:
            if b: seq.append(b.pop())
            if w: seq.append(w.pop())
        moves[g] = seq; lens[g] = rng.randint(120, 224)
    return moves, lens


def run(budget=240.0, seed=20261004, verbose=True):
    """Random configurations from the stream of RandomState(seed) until `budget` seconds have passed (the stream is the same every time: a failure
    names its configuration).  Returns (configurations, games compared, mismatches as strings, lane forms met)."""
    rng = np.random.RandomState(seed)
    G.init(0)
    t0 = time.time(); searches = games = 0; forms = {}; bad = []
    while time.time() - t0 < budget:
        n = int(rng.randint(1, 40)); R = int(rng.choice([1, 2, 3, 5, 5, 5, 8, 12, 16, 17, 20, 32, 33, 40, 64])); P = int(rng.randint(20, 160))
        seed_k = int(rng.randint(1, 2**31)); first_id = int(rng.randint(0, 2**30)); c_puct = float(rng.choice([1.0, 2.5, 5.0]))
        if rng.rand() < 0.3:
            moves, lens = late_positions(rng, n)
        else:
            moves, lens, _ = G.synth_boards(n, int(rng.randint(0, 2)), first_board=int(rng.randint(0, 2**24)))
            lens = np.minimum(lens, int(rng.randint(0, 50))).astype(np.int32)
        planes = G.moves_to_planes(moves, lens)
        last = np.array([moves[g, lens[g] - 1] if lens[g] > 0 else -1 for g in range(n)], dtype=np.int16)
        t = G.BatchedMCTS(n, playouts_capacity=P, c_puct=c_puct, c_rollouts=R, seed=seed_k)
        t.set_roots(planes, last, first_game_id=first_id)
        t.run(P)
        visits, q, rv, nodes, status = t.root_stats()
        t.close()
        form = "quads" if 4 * R <= 64 else "pairs" if 2 * R <= 64 else "one lane"
        forms[form] = forms.get(form, 0) + 1
        for g in range(n):
            b = O.new_board()
            for i in range(int(lens[g])):
                O.lib().go_board_apply(C.byref(b), int(moves[g, i]), 1)
            om = O.MCTS(P, c_puct, R, seed_k, first_id + g)
            om.run_playouts(b)
            ov, _, _ = om.root_children()
            ok = (ov == visits[g]).all() and np.float32(q[g]).tobytes() == np.float32(om.root_value).tobytes() and nodes[g] == om.size
            if not ok:
                bad.append("configuration %d: n %d R %d P %d seed %d first %d game %d stones %d" % (searches, n, R, P, seed_k, first_id, g, lens[g]))
                if verbose:
                    print("MISMATCH: " + bad[-1], flush=True)
            games += 1
        searches += 1
        if verbose and searches % 50 == 0:
            print("%d configurations, %d games compared, %d mismatches, %.0f s" % (searches, games, len(bad), time.time() - t0), flush=True)
    return searches, games, bad, forms


if __name__ == "__main__":
    searches, games, bad, forms = run(float(sys.argv[1]) if len(sys.argv) > 1 else 240.0)
    print("K3 stress parity: %d configurations (%s), %d games compared with the oracle (visits, root value bits, tree size): %d mismatches" % (searches, ", ".join("%s %d" % kv for kv in sorted(forms.items())), games, len(bad)))
    sys.exit(1 if bad else 0)
