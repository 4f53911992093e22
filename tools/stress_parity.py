#!/usr/bin/env python3
"""One-off differential stress run (not part of the suite): larger parity sweeps of K1, K3, K6 against the oracle."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gomokuai_amd import lib as G
from oracle import oracle as O
G.init(); O.lib()
t0 = time.time()
for kind in (0, 1):
    n = 65536
    moves, lens, planes = G.synth_boards(n, kind, first_board=7000000 + kind * 100000)
    got = G.eval_batch_host(planes)
    ref = O.replay_batch(moves, lens)
    bad = sum(int((a != b).reshape(n, -1).any(1).sum()) for a, b in zip(ref, got))
    print("K1 kind %d: %d boards, mismatching arrays: %d, oracle error flags %d  (%.1f s)" % (kind, n, bad, int((ref[3] & 2).astype(bool).sum()), time.time() - t0), flush=True)
# K3: 96 games x 800 playouts
n, P = 96, 800
moves, lens, _ = G.synth_boards(n, 0, first_board=31337)
lens = np.minimum(lens, 6).astype(np.int32)
planes = G.moves_to_planes(moves, lens)
last = np.array([moves[i, lens[i] - 1] for i in range(n)], dtype=np.int16)
tree = G.BatchedMCTS(n, playouts_capacity=P)
tree.set_roots(planes, last, first_game_id=1000); tree.run(P)
visits, q, rv, nodes, status = tree.root_stats()
bad = 0
for g in range(n):
    b = O.new_board()
    for i in range(int(lens[g])): O.lib().go_board_apply(C.byref(b), int(moves[g, i]), 1)
    om = O.MCTS(P, 5.0, 5, G.DEFAULT_SEED, 1000 + g); om.run_playouts(b)
    ov, oq, _ = om.root_children()
    bad += int((ov != visits[g]).any()) + int(np.float32(q[g]).tobytes() != np.float32(om.root_value).tobytes()) + int(nodes[g] != om.size)
print("K3: %d games x %d playouts, mismatches %d  (%.1f s)" % (n, P, bad, time.time() - t0), flush=True)
# K6: 14 games x 6000 playouts
n, P = 14, 6000
moves, lens, _ = G.synth_boards(n, 1, first_board=555)
pos = [[int(m) for m in moves[g, :min(int(lens[g]), 3 + 2 * g)]] for g in range(n)]
t = G.TraditionalMCTS(n, node_capacity=1 << 20); t.set_positions(pos); t.run(P)
st = t.root_stats(); bad = 0
for g in range(n):
    o = O.TraditionalMCTS(5.0); o.search(pos[g], P)
    v, qq, p, best = o.root_children()
    bad += int((v != st["visits"][g]).any()) + int((qq.view(np.uint32) != st["values"][g].view(np.uint32)).any()) + int(best != st["best"][g]) + int(o.n_nodes != st["n_nodes"][g]) + int(o.evaluator_updates != st["evaluator_updates"][g])
print("K6: %d games x %d playouts, mismatches %d, status %s  (%.1f s)" % (n, P, bad, st["status"].tolist(), time.time() - t0), flush=True)
# K8: 64 games x 3000 playouts, then a step and 1000 more from the kept subtrees
n, P = 64, 3000
moves, lens, _ = G.synth_boards(n, 0, first_board=777)
pos = [[int(m) for m in moves[g, :min(int(lens[g]), g % 30)]] for g in range(n)]
t = G.PoolRAVEMCTS(n, node_capacity=(P + 1200) * 225, c_puct=2.0, first_game_id=300); t.set_positions(pos); t.run(P)
orcs = [O.PoolRAVEMCTS(2.0, 0.0, seed=G.DEFAULT_SEED, game_id=300 + g) for g in range(n)]
def rave_bad(st):
    bad = 0
    for g in range(n):
        v, qq, p, av, aq, best = orcs[g].root_children()
        bad += int((v != st["visits"][g]).any()) + int((qq.view(np.uint32) != st["values"][g].view(np.uint32)).any()) + int(best != st["best"][g]) \
             + int((av != st["amaf_visits"][g]).any()) + int((aq.view(np.uint32) != st["amaf_values"][g].view(np.uint32)).any())
    return bad
for g in range(n): orcs[g].run(pos[g], P)
st = t.root_stats(); bad = rave_bad(st)
print("K8: %d games x %d playouts, mismatches %d, status %s  (%.1f s)" % (n, P, bad, sorted(set(st["status"].tolist())), time.time() - t0), flush=True)
t.step()
for g in range(n):
    if st["best"][g] >= 0: pos[g] = pos[g] + [orcs[g].step_forward()]
t.run(1000)
for g in range(n): orcs[g].run(pos[g], 1000)
st = t.root_stats(); bad = rave_bad(st)
print("K8 after a step: +1000 playouts from the kept subtrees, mismatches %d, root visits max %d  (%.1f s)" % (bad, int(st["root_visits"].max()), time.time() - t0), flush=True)
