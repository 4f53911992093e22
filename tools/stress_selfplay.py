"""The persistent self-play loops with the reference agent's semantics (kept subtree + Default::AddNoise from the counter-based sampler, inside ONE
launch) against the oracle's kept-tree game loops on randomly drawn configurations: seeds, first game ids, playout counts, noise (alpha, epsilon),
opening lengths, slot counts, c_puct.  Every move, game length, winner and recorded visit count of every game must be equal.
tools/stress_selfplay.py [seconds]; run(budget, seed) for a bounded slice."""
import ctypes as C
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gomokuai_amd import lib as G
from gomokuai_amd import selfplay
from oracle import oracle as O


def oracle_k3_game(game_id, playouts, seed, c_puct, c_rollouts, noise, opening):
    L = O.lib()
    b = O.new_board()
    for mv in opening:
        L.go_board_apply(C.byref(b), int(mv), 1)
    m = O.MCTS(playouts, c_puct, c_rollouts, seed, game_id)
    if noise:
        m.set_noise(noise[0], noise[1], sampler=1)
    moves, visits = [int(x) for x in opening], []
    while b.cur_player != 0:
        m.sync_with_board(b)
        _, _, v = m.eval_state(b)
        mv = m.step_forward()
        visits.append(v)
        moves.append(mv)
        L.go_board_apply(C.byref(b), mv, 1)
    return moves, visits, b.winner


def oracle_k6_game(game_id, playouts, seed, c_puct, noise, opening):
    L = O.lib()
    b = O.new_board()
    for mv in opening:
        L.go_board_apply(C.byref(b), int(mv), 1)
    t = O.TraditionalMCTS(c_puct)
    if noise:
        t.set_noise(noise[0], noise[1], seed, game_id, sampler=1)
    moves, visits = [int(x) for x in opening], []
    while b.cur_player != 0:
        t.run(moves, playouts)
        visits.append(t.root_children()[0].copy())
        mv = t.step_forward()
        moves.append(mv)
        L.go_board_apply(C.byref(b), mv, 1)
    return moves, visits, b.winner


def same(rec, g, opening, moves, visits, winner):
    n = int(rec.lens[g])
    if [int(x) for x in rec.moves[g, :n]] != moves or int(rec.winner[g]) != winner:
        return False
    return all((rec.visits[g, len(opening) + t].numpy().astype(np.uint32) == np.minimum(v, 65535)).all() for t, v in enumerate(visits))


def run(budget=240.0, seed=20261005, verbose=True):
    rng = np.random.RandomState(seed)
    G.init(0)
    t0 = time.time(); runs = games = 0; bad = []
    while time.time() - t0 < budget:
        which = "K3" if rng.rand() < 0.6 else "K6"
        s = int(rng.randint(1, 2**31)); first = int(rng.randint(0, 2**28)); plies = int(rng.randint(0, 9))
        noise = None if rng.rand() < 0.2 else (float(rng.choice([0.05, 0.3, 1.0])), float(rng.choice([0.25, 0.5])))
        if which == "K3":
            n = int(rng.randint(2, 25)); P = int(rng.randint(15, 90)); R = int(rng.choice([1, 3, 5, 5, 8])); c_puct = float(rng.choice([1.0, 5.0]))
            slots = None if rng.rand() < 0.3 else int(rng.randint(1, n + 6))
            rec = selfplay.play_games(n, P, seed=s, first_game_id=first, c_puct=c_puct, c_rollouts=R, opening_plies=plies, reuse_subtree=True, root_noise=noise, slots=slots).cpu()
        else:
            n = int(rng.randint(2, 12)); P = int(rng.randint(30, 110)); c_puct = float(rng.choice([2.0, 5.0])); plies = min(plies, 6)
            slots = None if rng.rand() < 0.3 else int(rng.randint(1, n + 4))
            rec = selfplay.play_supervisor_games(n, P, c_puct=c_puct, seed=s, first_game_id=first, opening_plies=plies, reuse_subtree=True, root_noise=noise, slots=slots,
                                                 device_loop="persistent").cpu()
        if rec.overflow:
            bad.append("run %d (%s): arena overflow" % (runs, which))
        m, l, _ = G.synth_boards(n, 0, seed=s, first_board=first)
        for g in range(n):
            opening = [int(x) for x in m[g, :min(int(l[g]), plies)]]
            ref = oracle_k3_game(first + g, P, s, c_puct, R, noise, opening) if which == "K3" else oracle_k6_game(first + g, P, s, c_puct, noise, opening)
            if not same(rec, g, opening, *ref):
                bad.append("run %d (%s): n %d P %d seed %d first %d plies %d noise %s slots %s game %d" % (runs, which, n, P, s, first, plies, noise, slots, g))
                if verbose:
                    print("MISMATCH " + bad[-1], flush=True)
            games += 1
        runs += 1
        if verbose and runs % 10 == 0:
            print("%d runs, %d games compared, %d mismatches, %.0f s" % (runs, games, len(bad), time.time() - t0), flush=True)
    return runs, games, bad


if __name__ == "__main__":
    runs, games, bad = run(float(sys.argv[1]) if len(sys.argv) > 1 else 240.0)
    print("self-play stress parity (persistent loops, kept subtrees, device-drawn root noise): %d runs, %d whole games compared with the oracle's game loops (moves, winners, visit counts): %d mismatches" % (runs, games, len(bad)))
    sys.exit(1 if bad else 0)
