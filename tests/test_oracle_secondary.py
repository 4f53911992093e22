"""Secondary evidence for the oracle: values SURVEY.md Appendix B.2 records from the real reference
sources (survey-session probe).  They are not the reference's own fixtures, so they do not by
themselves pin parity, but every one of them is reproduced bit for bit."""
import ctypes as C
import os

import numpy as np


def test_production_table_shape(oracle):
    O = oracle
    ac = O.default_ac()
    assert ac.n_patterns == 294
    lens = [ac.patterns[i].len for i in range(294)]
    assert (lens.count(5), lens.count(6), lens.count(7)) == (12, 66, 216)
    assert ac.size == 1024
    assert O.lib().go_ac_used_slots(ac) == 850
    assert list(ac.invariants) == [0, 11, 234, 3, 800]
    # four pairs of patterns have equal sort keys (ACAutomata.cpp:78-80): the trie depends on the
    # order std::sort leaves them in; the oracle uses libstdc++'s, as a g++ build of the reference does.
    assert ac.sort_ties == 4


def test_diagonal_multiplier_is_exact_six_fifths(oracle):
    """Pattern.cpp:151-152: int(delta * 1.2 * score) == delta * (6 * score / 5) for every table score."""
    O = oracle
    ac = O.default_ac()
    for i in range(ac.n_patterns):
        s = ac.patterns[i].score
        assert int(1 * 1.2 * s) == (6 * s) // 5
        assert int(-1 * 1.2 * s) == -((6 * s) // 5)


def test_kifu_scores(oracle):
    O = oracle
    ev = O.Evaluator()
    for (x, y) in [(7, 7), (8, 7), (7, 6), (7, 8), (6, 9)]:
        _, err = ev.apply(y * 15 + x)
        assert err == 0
    s = ev.scores()
    assert s.sum(axis=1).tolist() == [9948, 3708, 7776, 16256]
    assert (s != 0).sum(axis=1).tolist() == [39, 27, 43, 53]
    assert s.max(axis=1).tolist() == [490, 330, 480, 640]
    assert s.argmax(axis=1).tolist() == [9 * 15 + 8, 9 * 15 + 8, 6 * 15 + 6, 6 * 15 + 6]
    assert s[3].reshape(15, 15)[7].tolist() == [0, 0, 0, 0, 370, 190, 520, 0, 0, 160, 160, 0, 0, 0, 0]
    assert s[0].reshape(15, 15)[8].tolist() == [0, 0, 0, 0, 310, 310, 310, 0, 460, 346, 310, 0, 0, 0, 0]
    tot = ev.pattern_dist()[225]
    assert [(int(v & 0xffff), int(v >> 16)) for v in tot] == [(3, 3), (6, 14), (1, 1)] + [(0, 0)] * 5
    assert not ev.compound_dist().any()
    d = ev.density()
    assert (d[0, 0].sum(), d[0, 1].sum(), d[1, 0].sum(), d[1, 1].sum()) == (43, 125, 71, 217)
    assert d[:, :, 7 * 15 + 7].tolist() == [[-3, -9], [-3, -8]]


class _MT(C.Structure):
    _fields_ = [("mt", C.c_uint32 * 624), ("idx", C.c_int)]


def test_replay_hash(oracle):
    O = oracle
    L = O.lib()
    L.go_mt_seed.argtypes = [C.POINTER(_MT), C.c_uint32]
    L.go_mt_next.argtypes = [C.POINTER(_MT)]
    L.go_mt_next.restype = C.c_uint32
    mt = _MT()
    L.go_mt_seed(C.byref(mt), 5489)
    assert L.go_mt_next(C.byref(mt)) == 3499211612      # MT19937 reference first output
    L.go_mt_seed(C.byref(mt), 12345)
    ev = O.Evaluator()
    h, moves, ended = 1469598103934665603, 0, 0
    for _ in range(200):
        ev.reset()
        for _ply in range(60):
            b = ev.board
            if b.cur_player == 0:
                break
            while True:
                mid = L.go_mt_next(C.byref(mt)) % 225
                if L.go_board_check_move(C.byref(b), mid):
                    break
            _, err = ev.apply(mid)
            assert err == 0
            moves += 1
        ended += ev.board.cur_player == 0
        for v in ev.scores().reshape(-1):
            h ^= int(v) & 0xffffffff
            h = (h * 1099511628211) & 0xffffffffffffffff
        n = ev.board.nrec
    assert h == 0x6a454cca8155d73b
    assert (moves, ended) == (11934, 6)
    ev.revert(n)
    assert not ev.scores().any() and not ev.pattern_dist().any() and not ev.compound_dist().any()


def test_mcts_kat(oracle):
    O = oracle
    L = O.lib()
    L.go_mcts_use_mt19937.argtypes = [C.c_void_p, C.c_uint32]
    m = O.MCTS(800)
    L.go_mcts_use_mt19937(m.h, 777)
    b = O.new_board()
    q, pi, v = m.eval_state(b)
    assert m.root_visits == 800
    assert abs(float(q) - (-0.045)) < 1e-6            # SURVEY quotes the value to 3 digits
    assert int(v.argmax()) == 5 * 15 + 7 and int(v.max()) == 52
    assert m.size == 178417
    assert b.nrec == 0 and b.cur_player == 1          # mcts_unittest.cpp:26-35: board invariant
    assert abs(float(pi.sum()) - 1.0) < 1e-4


def test_philox_kat(oracle):
    """Random123 known answers for Philox4x32-10."""
    O = oracle
    assert O.philox([0] * 4, [0] * 2).tolist() == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    assert O.philox([0xffffffff] * 4, [0xffffffff] * 2).tolist() == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    assert O.philox([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0]).tolist() == \
        [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]


def test_oracle_tables_are_built_once_under_threads():
    """The oracle builds its automaton on first use; tools/stress_parity.py and bench.py's all-cores baseline call it from many host threads, and a
    fresh process whose FIRST calls come from eight threads at once must get what a serial call gets (the lazy singleton used to publish its
    pointer before the tables behind it were finished: the first boards of some threads were evaluated on a half-built automaton)."""
    import subprocess, sys, textwrap
    code = textwrap.dedent("""
        import numpy as np, sys
        from concurrent.futures import ThreadPoolExecutor
        sys.path.insert(0, %r)
        from oracle import oracle as O
        rng = np.random.default_rng(1)
        n = 64
        moves = np.zeros((n, 225), np.uint8); lens = np.full(n, 30, np.int32)
        for g in range(n):
            moves[g, :30] = rng.permutation(225)[:30]
        with ThreadPoolExecutor(8) as pool:
            parts = list(pool.map(lambda i: O.replay_batch(moves[8 * i:8 * i + 8], lens[8 * i:8 * i + 8]), range(8)))
        threaded = [np.concatenate([p[k] for p in parts]) for k in range(4)]
        serial = O.replay_batch(moves, lens)
        print(all((a == b).all() for a, b in zip(threaded, serial)))
        """) % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for _ in range(3):
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
        assert r.stdout.strip() == "True", (r.stdout[-300:], r.stderr[-500:])
