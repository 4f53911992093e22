"""K2 parity on the GPU: the device-resident incremental Evaluator (gmk_evalstate_*) vs the oracle's restatement of
Evaluator::applyMove / revertMove, INCLUDING the history-dependent per-cell 2-bit flag words of m_patternDist and
m_compoundDist, which K1 cannot produce from the position."""
import numpy as np
import pytest

from gomokuai_amd import lib as G

pytestmark = pytest.mark.gpu


def _oracle_states(O, scripts):
    out = []
    for script in scripts:
        ev = O.Evaluator()
        for mv in script:
            if mv >= 0:
                _, err = ev.apply(int(mv))
                assert err == 0
            elif mv == -2:
                ev.revert(1)
        b = ev.board
        out.append((ev.scores(), ev.density(), ev.pattern_dist(), ev.compound_dist(), (b.nrec, b.cur_player, b.winner), list(b.record[:b.nrec])))
    return out


def _compare(states, ref):
    for g, (scores, density, pdist, cdist, meta, record) in enumerate(ref):
        assert (states["scores"][g] == scores).all(), "scores game %d" % g
        assert (states["density"][g] == density).all(), "density game %d" % g
        assert (states["pattern_dist"][g] == pdist).all(), "pattern flags / totals game %d" % g
        assert (states["compound_dist"][g] == cdist).all(), "compound flags / totals game %d" % g
        assert tuple(int(v) for v in states["meta"][g][:3]) == meta and states["meta"][g][3] == 0
        assert list(states["record"][g][:meta[0]]) == record


@pytest.mark.parametrize("kind", [0, 1])
def test_apply_sequences_match_oracle(oracle, kind):
    n = 48
    moves, lens, _ = G.synth_boards(n, kind, first_board=321)
    k = int(lens.max())
    script = np.full((n, k), -1, dtype=np.int16)
    for g in range(n):
        script[g, :lens[g]] = moves[g, :lens[g]]
    st = G.EvaluatorStates(n)
    st.update(script)
    _compare(st.read(), _oracle_states(oracle, [list(s) for s in script]))
    # the per-cell flags really are history dependent: K1-style position outputs agree, flag words need the replay
    assert st.read()["pattern_dist"][:, :225].any()
    st.close()


def test_apply_revert_mix_and_incremental_launches(oracle):
    """Moves arrive over several launches, with reverts (also of a winning move) and illegal moves in between."""
    rng = np.random.RandomState(4)
    n = 24
    moves, lens, _ = G.synth_boards(n, 1, first_board=999)
    scripts = []
    for g in range(n):
        s = []
        for i in range(int(lens[g])):
            s.append(int(moves[g, i]))
            r = rng.rand()
            if r < 0.15:
                s.append(-2)                         # take it back ...
                s.append(int(moves[g, i]))           # ... and play it again
            elif r < 0.25:
                s.append(int(moves[g, i]))           # occupied cell: ignored (Evaluator::applyMove checks checkMove)
        scripts.append(s)
    k = max(len(s) for s in scripts)
    script = np.full((n, k), -1, dtype=np.int16)
    for g, s in enumerate(scripts):
        script[g, :len(s)] = s
    st = G.EvaluatorStates(n)
    for lo in range(0, k, 7):                        # 7 entries per launch
        st.update(script[:, lo:lo + 7])
    _compare(st.read(), _oracle_states(oracle, scripts))
    # reverting everything returns every member to zero
    st.update(np.full((n, k), -2, dtype=np.int16))
    z = st.read()
    assert not z["scores"].any() and not z["density"].any() and not z["pattern_dist"].any() and not z["compound_dist"].any()
    assert (z["meta"][:, 0] == 0).all() and (z["meta"][:, 1] == 1).all()
    st.close()


def test_position_outputs_agree_with_k1(oracle):
    """scores / density / totals of the incremental state == K1's from-scratch evaluation of the same position."""
    n = 64
    moves, lens, planes = G.synth_boards(n, 1, first_board=5000)
    k = int(lens.max())
    script = np.full((n, k), -1, dtype=np.int16)
    for g in range(n):
        script[g, :lens[g]] = moves[g, :lens[g]]
    st = G.EvaluatorStates(n)
    st.update(script)
    s = st.read()
    scores, density, totals, status = G.eval_batch_host(planes)
    assert (s["scores"] == scores).all() and (s["density"] == density).all()
    assert (s["pattern_dist"][:, 225, :] == totals[:, :8]).all() and (s["compound_dist"][:, 225, :] == totals[:, 8:]).all()
    st.close()


def test_dense_games_to_the_full_board_and_back(oracle):
    """Games without a five played to 150 .. 225 stones (shuffled tie games), then taken back completely: the incremental state
    (saturating flag words included) follows the oracle at the densest positions, up to the full board, and returns to the
    empty evaluator."""
    rng = np.random.RandomState(21)
    cls = lambda c: ((c % 15) // 2 + c // 15) % 2
    blacks, whites = [c for c in range(225) if cls(c) == 0], [c for c in range(225) if cls(c) == 1]
    n = 12
    scripts = []
    for g in range(n):
        b, w = list(rng.permutation(blacks)), list(rng.permutation(whites))
        seq = []
        while b or w:
            if b:
                seq.append(int(b.pop()))
            if w:
                seq.append(int(w.pop()))
        scripts.append(seq[:225 if g < 2 else int(rng.randint(150, 226))])
    k = max(len(s) for s in scripts)
    script = np.full((n, k), -1, dtype=np.int16)
    for g, s in enumerate(scripts):
        script[g, :len(s)] = s
    st = G.EvaluatorStates(n)
    st.update(script)
    states = st.read()
    _compare(states, _oracle_states(oracle, scripts))
    assert states["meta"][0][0] == 225 and states["meta"][0][2] == 0      # a full board without a winner (the tie itself is Evaluator::checkGameEnd's finding)
    st.update(np.full((n, k), -2, dtype=np.int16))
    back = st.read()
    assert not back["scores"].any() and not back["pattern_dist"].any() and not back["compound_dist"].any() and (back["meta"][:, 0] == 0).all()
    st.close()


def test_a_full_launch_of_games_with_reverts(oracle):
    """2 400 games at once (more than a launch of nine per CU holds, the last workgroup partly filled): every game applies its clustered
    synthetic game, takes a third of it back and plays on with other cells of the game; scores, packed density words, pattern and
    compound flag words, move records: all equal the oracle's."""
    n = 2400
    rng = np.random.RandomState(11)
    moves, lens, _ = G.synth_boards(n, 1, first_board=5000)
    scripts = []
    for g in range(n):
        m = [int(c) for c in moves[g, :lens[g]]]
        back = len(m) // 3
        tail = m[len(m) - back:]
        rng.shuffle(tail)
        scripts.append(m + [-2] * back + tail[: max(1, back - 1)])
    k = max(len(sc) for sc in scripts)
    script = np.full((n, k), -1, dtype=np.int16)
    for g, sc in enumerate(scripts):
        script[g, :len(sc)] = sc
    st = G.EvaluatorStates(n)
    st.update(script)
    states = st.read()
    ref = _oracle_states(oracle, scripts)
    for g, (scores, density, pdist, cdist, meta, record) in enumerate(ref):
        same = (states["scores"][g] == scores).all() and (states["density"][g] == density).all() and (states["pattern_dist"][g] == pdist).all() and \
               (states["compound_dist"][g] == cdist).all() and tuple(int(v) for v in states["meta"][g][:3]) == meta and list(states["record"][g][:meta[0]]) == record
        assert same, "game %d" % g
    assert (states["compound_dist"][:, :225] != 0).any(1).mean() > 0.2
    st.close()
