"""K6 (and through it K2's device functions) against the oracle on randomly drawn positions: clustered and random synthetic games cut at random
lengths, 100 .. 700 playouts, several c_puct, 1 .. 40 games per search (workgroups partly filled), a second search on the same handle from a
continued position (the evaluator is synchronised, not rebuilt).  Compared exactly: visit counts, the bits of values and priors, the move to
play, tree size, the number of evaluator updates, and the evaluator state left behind (scores, density, pattern and compound flag words).
tools/stress_k6.py [seconds]; run(budget, seed) is what tests/test_stress_gpu.py calls for a bounded slice"""
import ctypes as C
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gomokuai_amd import lib as G
from oracle import oracle as O

L = None


def same(stats, dev, g, orc):
    v, q, p, best = orc.root_children()
    ok = stats["status"][g] == 0 and (stats["visits"][g] == v).all() and (stats["priors"][g].view(np.uint32) == p.view(np.uint32)).all() and \
         (stats["values"][g].view(np.uint32) == q.view(np.uint32)).all() and stats["best"][g] == best and stats["root_visits"][g] == orc.root_visits and \
         np.float32(stats["root_value"][g]).view(np.uint32) == np.float32(orc.root_value).view(np.uint32) and stats["n_nodes"][g] == orc.n_nodes and \
         stats["evaluator_updates"][g] == orc.evaluator_updates
    ev = L.go_trad_evaluator(orc.h)
    scores = np.zeros((4, 225), np.int32); density = np.zeros((2, 2, 225), np.int32)
    pd = np.zeros((226, 8), np.uint32); cd = np.zeros((226, 3), np.uint32)
    L.go_eval_get_scores(C.c_void_p(ev), scores.ctypes.data); L.go_eval_get_density(C.c_void_p(ev), density.ctypes.data)
    L.go_eval_get_pattern_dist(C.c_void_p(ev), pd.ctypes.data); L.go_eval_get_compound_dist(C.c_void_p(ev), cd.ctypes.data)
    return bool(ok) and (dev["scores"][g] == scores).all() and (dev["density"][g] == density).all() and (dev["pattern_dist"][g] == pd).all() and (dev["compound_dist"][g] == cd).all()


def run(budget=240.0, seed=4101, verbose=True):
    """Random configurations from the stream of RandomState(seed) until `budget` seconds have passed (the stream is the same every time: a failure
    names its configuration).  Returns (searches, game searches compared, mismatches as strings)."""
    global L
    rng = np.random.RandomState(seed)
    G.init(0)
    L = O.lib()
    t0 = time.time(); searches = games = 0; bad = []
    while time.time() - t0 < budget:
        n = int(rng.randint(1, 41)); P = int(rng.randint(100, 700)); c_puct = float(rng.choice([2.0, 5.0])); kind = int(rng.randint(0, 2))
        moves, lens, _ = G.synth_boards(n, kind, first_board=int(rng.randint(0, 2**24)))
        cut = [int(min(lens[g], rng.randint(0, 61))) for g in range(n)]
        pos = [[int(m) for m in moves[g, :cut[g]]] for g in range(n)]
        t = G.TraditionalMCTS(n, node_capacity=1 << 17, c_puct=c_puct)
        t.set_positions(pos)
        t.run(P)
        stats, dev = t.root_stats(), t.read_evaluators()
        orcs = []
        for g in range(n):
            orc = O.TraditionalMCTS(c_puct)
            orc.search(pos[g], P)
            orcs.append(orc)
            if not same(stats, dev, g, orc):
                bad.append("search %d (first): n %d P %d kind %d game %d stones %d" % (searches, n, P, kind, g, cut[g]))
                if verbose:
                    print("MISMATCH " + bad[-1], flush=True)
            games += 1
        # the game goes on by two moves of its own record (where there are any): a second search on the same handle and evaluator
        more = [pos[g] + [int(m) for m in moves[g, cut[g]:min(int(lens[g]), cut[g] + 2)]] for g in range(n)]
        t.set_positions(more)
        t.run(P // 2)
        stats, dev = t.root_stats(), t.read_evaluators()
        for g in range(n):
            orcs[g].search(more[g], P // 2)
            if not same(stats, dev, g, orcs[g]):
                bad.append("search %d (second): n %d P %d kind %d game %d stones %d" % (searches + 1, n, P, kind, g, len(more[g])))
                if verbose:
                    print("MISMATCH " + bad[-1], flush=True)
            games += 1
        t.close()
        searches += 2
        if verbose and searches % 20 == 0:
            print("%d searches, %d game searches compared, %d mismatches, %.0f s" % (searches, games, len(bad), time.time() - t0), flush=True)
    return searches, games, bad


if __name__ == "__main__":
    searches, games, bad = run(float(sys.argv[1]) if len(sys.argv) > 1 else 240.0)
    print("K6 stress parity: %d searches, %d game searches compared with the oracle (visits, value and prior bits, best move, tree size, evaluator updates, evaluator state with its flag words): %d mismatches" % (searches, games, len(bad)))
    sys.exit(1 if bad else 0)
