"""K8 (PoolRAVE search; its rollout runs on a quad of lanes since round 3) against the oracle on randomly drawn positions, seeds and game ids:
visit counts, value / prior / AMAF value bits, AMAF visit counts, the move to play, the tree size.  tools/stress_k8.py [seconds]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gomokuai_amd import lib as G
from oracle import oracle as O

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 180.0
rng = np.random.RandomState(8801)
G.init(0)
t0 = time.time(); searches = games = mismatches = 0
while time.time() - t0 < budget:
    n = int(rng.randint(1, 33)); P = int(rng.randint(50, 500)); c_puct = float(rng.choice([1e-4, 2.0, 5.0])); seed = int(rng.randint(1, 2**62)); first_id = int(rng.randint(0, 2**30))
    moves, lens, _ = G.synth_boards(n, int(rng.randint(0, 2)), first_board=int(rng.randint(0, 2**24)))
    pos = [[int(m) for m in moves[g, :int(min(lens[g], rng.randint(0, 62)))]] for g in range(n)]
    t = G.PoolRAVEMCTS(n, node_capacity=1 << 17, c_puct=c_puct, seed=seed, first_game_id=first_id)
    t.set_positions(pos)
    t.run(P)
    stats = t.root_stats()
    t.close()
    for g in range(n):
        orc = O.PoolRAVEMCTS(c_puct, 0.0, seed=seed, game_id=first_id + g)
        orc.run(pos[g], P)
        v, q, p, av, aq, best = orc.root_children()
        ok = stats["status"][g] == 0 and (stats["visits"][g] == v).all() and (stats["priors"][g].view(np.uint32) == p.view(np.uint32)).all() and \
             (stats["values"][g].view(np.uint32) == q.view(np.uint32)).all() and (stats["amaf_visits"][g] == av).all() and \
             (stats["amaf_values"][g].view(np.uint32) == aq.view(np.uint32)).all() and stats["best"][g] == best and stats["root_visits"][g] == orc.root_visits and \
             np.float32(stats["root_value"][g]).view(np.uint32) == np.float32(orc.root_value).view(np.uint32) and stats["n_nodes"][g] == orc.size
        if not ok:
            mismatches += 1
            print("MISMATCH: n %d P %d c_puct %g seed %d first %d game %d stones %d" % (n, P, c_puct, seed, first_id, g, len(pos[g])), flush=True)
        games += 1
    searches += 1
    if searches % 25 == 0:
        print("%d searches, %d games compared, %d mismatches, %.0f s" % (searches, games, mismatches, time.time() - t0), flush=True)
print("K8 stress parity: %d searches, %d games compared with the oracle (visits, value / prior / AMAF bits, best move, tree size): %d mismatches" % (searches, games, mismatches))
sys.exit(1 if mismatches else 0)
