"""Whole games on the GPU (K3 + gmk_mcts_advance) vs the oracle playing the same games move by move."""
import ctypes as C

import numpy as np
import pytest

from gomokuai_amd import lib as G
from gomokuai_amd import selfplay

pytestmark = pytest.mark.gpu


def _oracle_game(O, game_id, playouts, seed, opening):
    L = O.lib()
    b = O.new_board()
    for mv in opening:
        L.go_board_apply(C.byref(b), int(mv), 1)
    moves, visits = list(opening), []
    while b.cur_player != 0:
        m = O.MCTS(playouts, 5.0, 5, seed, game_id)        # fresh tree every move (MCTS::reset + syncWithBoard)
        _, _, v = m.eval_state(b)
        mv = m.step_forward()
        visits.append(v)
        moves.append(mv)
        L.go_board_apply(C.byref(b), mv, 1)
    return moves, visits, b.winner


@pytest.mark.parametrize("opening_plies", [0, 4])
def test_games_match_oracle(oracle, opening_plies):
    n, playouts, seed, first = 5, 40, 99, 300
    rec = selfplay.play_games(n, playouts, seed=seed, first_game_id=first, opening_plies=opening_plies).cpu()
    assert not rec.overflow
    for g in range(n):
        L = int(rec.lens[g])
        opening = [int(x) for x in rec.moves[g, :opening_plies]] if opening_plies else []
        if opening_plies:
            m, l, _ = G.synth_boards(n, 0, seed=seed, first_board=first)
            assert opening == [int(x) for x in m[g, :min(int(l[g]), opening_plies)]]
        moves, visits, winner = _oracle_game(oracle, first + g, playouts, seed, opening)
        assert [int(x) for x in rec.moves[g, :L]] == moves, "game %d" % g
        assert int(rec.winner[g]) == winner
        for t, v in enumerate(visits):
            assert (rec.visits[g, len(opening) + t].numpy().astype(np.uint32) == v).all()


def test_records_do_not_depend_on_sharding():
    a = selfplay.play_games(6, 30, seed=5, first_game_id=0, record_visits=False).cpu()
    b1 = selfplay.play_games(2, 30, seed=5, first_game_id=0, record_visits=False).cpu()
    b2 = selfplay.play_games(4, 30, seed=5, first_game_id=2, record_visits=False).cpu()
    assert (a.moves[:2] == b1.moves).all() and (a.moves[2:] == b2.moves).all()
    assert (a.winner[:2] == b1.winner).all() and (a.winner[2:] == b2.winner).all()
    samples_ok = selfplay.play_games(1, 30, seed=5).samples(0)
    assert samples_ok[0][0].shape == (6, 15, 15) and len(samples_ok) == int(a.lens[0])


@pytest.mark.parametrize("opening_plies,reuse,noise", [(0, False, None), (4, False, None), (3, True, (0.05, 0.25))])
def test_continuous_batching_gives_the_same_games(opening_plies, reuse, noise):
    """gmk_selfplay_run: 23 games through 5 slots (a slot whose game ends takes the next unstarted game inside the step kernel)
    == the same 23 games played side by side in lock step: moves, winners, lengths and every recorded visit count.  A game's random
    streams (rollouts, root noise) are keyed by the game's global id, not by the slot it runs in."""
    n, playouts = 23, 30
    a = selfplay.play_games(n, playouts, seed=77, first_game_id=900, opening_plies=opening_plies, reuse_subtree=reuse, root_noise=noise, lockstep=True).cpu()      # the host-driven loop
    b = selfplay.play_games(n, playouts, seed=77, first_game_id=900, opening_plies=opening_plies, reuse_subtree=reuse, root_noise=noise, slots=5).cpu()
    assert not a.overflow and not b.overflow
    assert (a.lens == b.lens).all() and (a.winner == b.winner).all() and (a.moves == b.moves).all()
    assert (a.visits == b.visits).all()
    assert int(a.lens.min()) >= 9                       # whole games
    c = selfplay.play_games(n, playouts, seed=77, first_game_id=900, opening_plies=opening_plies, reuse_subtree=reuse, root_noise=noise, slots=64).cpu()      # more slots than games
    assert (a.moves == c.moves).all() and (a.winner == c.winner).all()
    # three search handles side by side (own streams and host threads), 6 slots between them, and two handles with all games at once
    for kw in ({"slots": 6, "handles": 3}, {"slots": None, "handles": 2}):
        d = selfplay.play_games(n, playouts, seed=77, first_game_id=900, opening_plies=opening_plies, reuse_subtree=reuse, root_noise=noise, **kw).cpu()
        assert not d.overflow
        assert (a.lens == d.lens).all() and (a.winner == d.winner).all() and (a.moves == d.moves).all() and (a.visits == d.visits).all()


def _oracle_game_reuse(O, game_id, playouts, seed, noise=None, sampler=0, opening=()):
    """One MCTS object for the whole game, as agents/mcts.py:17-21 drives it: sync, search, step_forward()."""
    L = O.lib()
    b = O.new_board()
    for mv in opening:
        L.go_board_apply(C.byref(b), int(mv), 1)
    m = O.MCTS(playouts, 5.0, 5, seed, game_id)
    if noise:
        m.set_noise(*noise, sampler=sampler)
    moves, visits = list(opening), []
    while b.cur_player != 0:
        m.sync_with_board(b)
        _, _, v = m.eval_state(b)
        mv = m.step_forward()
        visits.append(v)
        moves.append(mv)
        L.go_board_apply(C.byref(b), mv, 1)
    return moves, visits, b.winner


def test_subtree_reuse_matches_oracle(oracle):
    """reuse_subtree=1: the chosen child's subtree is the next search's tree (MCTS::stepForward, MCTS.cpp:129-134),
    without root noise (Default::AddNoise draws from a random_device-seeded std::gamma_distribution: unpinned)."""
    n, playouts, seed, first = 4, 50, 1234, 40
    rec = selfplay.play_games(n, playouts, seed=seed, first_game_id=first, reuse_subtree=True).cpu()
    assert not rec.overflow
    for g in range(n):
        moves, visits, winner = _oracle_game_reuse(oracle, first + g, playouts, seed)
        L = int(rec.lens[g])
        assert [int(x) for x in rec.moves[g, :L]] == moves, "game %d" % g
        assert int(rec.winner[g]) == winner
        for t, v in enumerate(visits):
            assert (rec.visits[g, t].numpy().astype(np.uint32) == np.minimum(v, 65535)).all()


def test_root_noise_matches_oracle(oracle):
    """The reference's self-play configuration: subtree reuse + Default::AddNoise(alpha 0.05, epsilon 0.25) at the
    start of every search (MCTS.cpp:182).  Same std::gamma_distribution<float> on both sides, Philox-derived seeds."""
    n, playouts, seed, first = 3, 50, 77, 900
    rec = selfplay.play_games(n, playouts, seed=seed, first_game_id=first, reuse_subtree=True, root_noise=(0.05, 0.25), noise_sampler="std").cpu()
    plain = selfplay.play_games(n, playouts, seed=seed, first_game_id=first, reuse_subtree=True).cpu()
    assert not rec.overflow
    differs = False
    for g in range(n):
        moves, visits, winner = _oracle_game_reuse(oracle, first + g, playouts, seed, noise=(0.05, 0.25))
        L = int(rec.lens[g])
        assert [int(x) for x in rec.moves[g, :L]] == moves, "game %d" % g
        assert int(rec.winner[g]) == winner
        for t, v in enumerate(visits):
            assert (rec.visits[g, t].numpy().astype(np.uint32) == np.minimum(v, 65535)).all()
        differs |= [int(x) for x in plain.moves[g, :int(plain.lens[g])]] != moves
    assert differs                      # the noise does change the games


@pytest.mark.parametrize("slots", [None, 5, 23, 40])
def test_reference_semantics_in_one_launch(oracle, slots):
    """The reference agent's per-move semantics -- the chosen child's subtree is the next search's tree (MCTS.cpp:129-147) and Default::AddNoise
    (0.05, 0.25) runs before every search (MCTS.cpp:179-183) -- inside ONE persistent launch (per-game arena flip, noise drawn by the wavefront
    from the counter-based sampler of include/gomoku_noise.h) == the lock-step loop with the same sampler == the oracle's game loop with it:
    moves, lengths, winners and every recorded visit count, for 23 games through 5 / 23 / 40 slots and all at once."""
    n, playouts, seed, first, noise, plies = 23, 30, 4242, 700, (0.05, 0.25), 3
    one = selfplay.play_games(n, playouts, seed=seed, first_game_id=first, opening_plies=plies, reuse_subtree=True, root_noise=noise, slots=slots).cpu()
    step = selfplay.play_games(n, playouts, seed=seed, first_game_id=first, opening_plies=plies, reuse_subtree=True, root_noise=noise, slots=slots, lockstep=True).cpu()
    assert not one.overflow and not step.overflow
    assert (one.lens == step.lens).all() and (one.winner == step.winner).all() and (one.moves == step.moves).all() and (one.visits == step.visits).all()
    m, l, _ = G.synth_boards(n, 0, seed=seed, first_board=first)
    for g in range(n):
        opening = [int(x) for x in m[g, :min(int(l[g]), plies)]]
        moves, visits, winner = _oracle_game_reuse(oracle, first + g, playouts, seed, noise=noise, sampler=1, opening=opening)
        L = int(one.lens[g])
        assert [int(x) for x in one.moves[g, :L]] == moves, "game %d" % g
        assert int(one.winner[g]) == winner
        for t, v in enumerate(visits):
            assert (one.visits[g, len(opening) + t].numpy().astype(np.uint32) == np.minimum(v, 65535)).all(), "game %d move %d" % (g, t)


def test_kept_subtrees_in_one_launch_without_noise(oracle):
    """reuse_subtree alone inside the persistent launch == the oracle's kept-tree game loop."""
    n, playouts, seed, first = 6, 50, 1234, 40
    rec = selfplay.play_games(n, playouts, seed=seed, first_game_id=first, reuse_subtree=True, slots=4).cpu()
    assert not rec.overflow
    for g in range(n):
        moves, visits, winner = _oracle_game_reuse(oracle, first + g, playouts, seed)
        L = int(rec.lens[g])
        assert [int(x) for x in rec.moves[g, :L]] == moves, "game %d" % g
        assert int(rec.winner[g]) == winner
        for t, v in enumerate(visits):
            assert (rec.visits[g, t].numpy().astype(np.uint32) == np.minimum(v, 65535)).all()


def test_device_tuples_match_the_reference_loop():
    """K4 + K5 on the device against tuples made by the REFERENCE's own loop: tests/golden/reference_tuples.npz holds what
    dual_play(verbose=True) (agents/utils.py:29-63) returned for six games (tests/golden/make_reference_tuples.py, run in the build
    container on this CorePyExt).  The device gets only the move lists and winners; its encoded states and values must equal the
    reference loop's, tuple for tuple."""
    import os
    import torch
    ref = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_tuples.npz"))
    n = int(ref["lens"].shape[0])
    dev = torch.device("cuda", 0)
    rec = selfplay.GameRecords(torch.from_numpy(ref["moves"]).to(dev), torch.from_numpy(ref["lens"]).to(dev), torch.from_numpy(ref["winner"]).to(dev),
                               torch.ones((n, 225, 225), dtype=torch.int16, device=dev), 0)
    G.init(0)
    states, values, pi = (t.cpu().numpy() for t in rec.to_samples())
    assert states.shape[0] == int(ref["values"].shape[0])
    assert (states == ref["states"]).all()
    assert (values == ref["values"].astype(np.float32)).all()
    assert np.abs(pi.sum(1) - 1.0).max() < 1e-3


def test_device_planes_match_an_independent_restatement(oracle):
    """K4's feature planes against tests/helpers.py: encoded_states_of -- a restatement of core/py_ext/src/game_ext.hpp:87-104 that shares no
    code with the product (the fixture's states came from this repo's own Board.encoded_states) -- and that restatement against
    the oracle's board (a third statement of the same lines), for the fixture games of the reference's dual_play and for games played here."""
    import os
    import torch
    import helpers
    ref = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_tuples.npz"))
    dev = torch.device("cuda", 0)
    played = selfplay.play_games(5, 30, seed=99, first_game_id=11).cpu()
    for moves, lens, winner in ((ref["moves"], ref["lens"], ref["winner"]), (played.moves.numpy(), played.lens.numpy(), played.winner.numpy())):
        n = int(lens.shape[0])
        rec = selfplay.GameRecords(torch.from_numpy(np.ascontiguousarray(moves)).to(dev), torch.from_numpy(np.ascontiguousarray(lens)).to(dev),
                                   torch.from_numpy(np.ascontiguousarray(winner)).to(dev), torch.ones((n, 225, 225), dtype=torch.int16, device=dev), 0)
        states = rec.to_samples()[0].cpu().numpy()
        k = 0
        for g in range(n):
            ob = oracle.new_board()
            for ply in range(int(lens[g])):
                want = helpers.encoded_states_of([int(m) for m in moves[g, :ply]])
                third = np.zeros((6, 15, 15), dtype=np.uint8)
                oracle.lib().go_board_encoded_states(C.byref(ob), third.ctypes.data)
                assert (want == third).all(), (g, ply)
                assert (states[k] == want).all(), (g, ply)
                oracle.lib().go_board_apply(C.byref(ob), int(moves[g, ply]), 1)
                k += 1
        assert k == states.shape[0]


def test_device_samples_match_host_and_augmentation():
    """K4 + K5 on the device == the host construction of the tuples (GameRecords.samples, itself held to the reference loop's
    fixtures in tests/test_selfplay.py), and the eight-fold copies == the order network/data_helper.py:36-55 prescribes (restated
    in tests/helpers.py, whose permutation property the reference's own test states: tests/test_pyext.py)."""
    import helpers
    rec = selfplay.play_games(3, 40, seed=21, first_game_id=5)
    states, values, pi = (t.cpu().numpy() for t in rec.to_samples())
    k = 0
    for g in range(3):
        host = rec.samples(g)
        for st, val, p in host:
            assert (states[k] == st).all() and values[k] == np.float32(val)
            assert np.abs(pi[k] - p).max() <= 1e-6          # float32 log / double exp on device vs libm: tolerance
            k += 1
    assert k == states.shape[0]
    a_states, a_values, a_pi = (t.cpu().numpy() for t in rec.to_samples(augment=True))
    assert a_states.shape[0] == 8 * k
    ref = helpers.augment([(states[i], values[i], pi[i]) for i in range(k)])
    for i, (st, val, p) in enumerate(ref):
        assert (a_states[i] == st).all() and a_values[i] == val and (a_pi[i] == p).all()


def test_supervisor_self_play(oracle):
    """Complete games of the pattern-guided supervisor against itself (K6 per move): the records replay to the recorded
    winner on the oracle's board, are reproducible, and turn into training tuples on the device."""
    import ctypes as C
    from gomokuai_amd import selfplay
    rec = selfplay.play_supervisor_games(6, 80, opening_plies=3, first_game_id=11)
    again = selfplay.play_supervisor_games(6, 80, opening_plies=3, first_game_id=11)
    assert not rec.overflow and (rec.moves.cpu() == again.moves.cpu()).all() and (rec.winner.cpu() == again.winner.cpu()).all()
    r = rec.cpu()
    for g in range(len(rec)):
        b = oracle.new_board()
        L = int(r.lens[g])
        for i in range(L):
            assert oracle.lib().go_board_check_move(C.byref(b), int(r.moves[g, i]))
            oracle.lib().go_board_apply(C.byref(b), int(r.moves[g, i]), 1)
        assert b.cur_player == 0 and b.winner == int(r.winner[g])          # the game ended exactly with the last recorded move
        assert L >= 9
    states, values, pi = rec.to_samples(first_move=3)
    assert states.shape[0] == int((r.lens - 3).sum()) and float(pi.sum(1).sub(1).abs().max()) < 1e-3
    # with the subtree kept and root noise: still legal, finished games, reproducible from the seed
    a = selfplay.play_supervisor_games(4, 80, opening_plies=2, first_game_id=3, reuse_subtree=True, root_noise=(0.05, 0.25), seed=5)
    b2 = selfplay.play_supervisor_games(4, 80, opening_plies=2, first_game_id=3, reuse_subtree=True, root_noise=(0.05, 0.25), seed=5)
    assert (a.moves.cpu() == b2.moves.cpu()).all() and not a.overflow
    ra = a.cpu()
    for g in range(len(a)):
        b = oracle.new_board()
        for i in range(int(ra.lens[g])):
            assert oracle.lib().go_board_check_move(C.byref(b), int(ra.moves[g, i]))
            oracle.lib().go_board_apply(C.byref(b), int(ra.moves[g, i]), 1)
        assert b.cur_player == 0 and b.winner == int(ra.winner[g])


def test_poolrave_self_play(oracle):
    """The same loop with MCTS(PoolRAVEPolicy) on both sides (K8 per move, subtree kept, root noise): legal, finished,
    reproducible games."""
    import ctypes as C
    from gomokuai_amd import selfplay
    kw = dict(opening_plies=2, first_game_id=21, c_puct=2.0, reuse_subtree=True, root_noise=(0.05, 0.25), seed=9, policy="poolrave")
    a = selfplay.play_supervisor_games(5, 120, **kw)
    b2 = selfplay.play_supervisor_games(5, 120, **kw)
    assert (a.moves.cpu() == b2.moves.cpu()).all() and (a.winner.cpu() == b2.winner.cpu()).all() and not a.overflow
    ra = a.cpu()
    for g in range(len(a)):
        b = oracle.new_board()
        for i in range(int(ra.lens[g])):
            assert oracle.lib().go_board_check_move(C.byref(b), int(ra.moves[g, i]))
            oracle.lib().go_board_apply(C.byref(b), int(ra.moves[g, i]), 1)
        assert b.cur_player == 0 and b.winner == int(ra.winner[g])
    with pytest.raises(ValueError):
        selfplay.play_supervisor_games(1, 10, policy="other")


def test_supervisor_against_candidates(oracle):
    """The reference's data generation pairing (network/data_helper.py:15-63, config.py:6-20): supervisor vs candidate with random
    sides, both players' searches recorded -- batched: legal, finished, reproducible games and training tuples for every move."""
    import ctypes as C
    from gomokuai_amd import selfplay
    sup = ("traditional_mcts", {"c_puct": 5.0, "c_iterations": 150})
    for cand in (("random_mcts", {"c_puct": 5.0, "c_iterations": 120}), ("rave_mcts", {"c_puct": 5.0, "c_iterations": 120})):
        rec, sup_black = selfplay.play_match_games(10, sup, cand, seed=77, first_game_id=5, opening_plies=2)
        again, sb2 = selfplay.play_match_games(10, sup, cand, seed=77, first_game_id=5, opening_plies=2)
        assert (rec.moves.cpu() == again.moves.cpu()).all() and (sup_black == sb2).all() and not rec.overflow
        assert sup_black.any() and (~sup_black).any()
        r = rec.cpu()
        visits = r.visits.numpy().view(np.uint16)
        for g in range(len(rec)):
            b = oracle.new_board()
            L = int(r.lens[g])
            for i in range(L):
                assert oracle.lib().go_board_check_move(C.byref(b), int(r.moves[g, i]))
                oracle.lib().go_board_apply(C.byref(b), int(r.moves[g, i]), 1)
            assert b.cur_player == 0 and b.winner == int(r.winner[g]) and L >= 9
            for i in range(2, L):                                # every recorded move carries the visit counts of the player who made it
                mover_is_sup = (i % 2 == 0) == bool(sup_black[g])
                n = int(visits[g, i].sum())
                assert n == (150 if mover_is_sup else 120) - 1 or n == 0 or n < 150, (g, i, n)
                assert visits[g, i, int(r.moves[g, i])] == visits[g, i].max()
        states, values, pi = rec.to_samples(first_move=2)
        assert states.shape[0] == int((r.lens - 2).sum())
    # through 3 slots per group: every game still legal and finished
    rec, sup_black = selfplay.play_match_games(14, sup, ("rave_mcts", {"c_puct": 5.0, "c_iterations": 100}), seed=78, opening_plies=2, slots=3)
    r = rec.cpu()
    for g in range(len(rec)):
        b = oracle.new_board()
        for i in range(int(r.lens[g])):
            assert oracle.lib().go_board_check_move(C.byref(b), int(r.moves[g, i]))
            oracle.lib().go_board_apply(C.byref(b), int(r.moves[g, i]), 1)
        assert b.cur_player == 0 and b.winner == int(r.winner[g]) and int(r.lens[g]) >= 9
    with pytest.raises(ValueError):
        selfplay.play_match_games(2, sup, ("botzone", {"program": "x"}))


def test_network_self_play(oracle):
    """AlphaZero-style self-play: K7 in lock step with the fused network K9 at the leaves, subtree kept, root noise: legal,
    finished, reproducible games; the same games with the plain PyTorch module (the two networks agree to 1e-7, ties aside)."""
    import ctypes as C
    import torch
    from gomokuai_amd import selfplay
    from gomokuai_amd.network import FusedPolicyValueNetwork, PolicyValueNetwork
    net = PolicyValueNetwork(seed=4).cuda().eval()
    fused = FusedPolicyValueNetwork(net)
    a = selfplay.play_network_games(6, fused, 40, opening_plies=2, first_game_id=9, seed=3)
    b2 = selfplay.play_network_games(6, fused, 40, opening_plies=2, first_game_id=9, seed=3)
    assert (a.moves.cpu() == b2.moves.cpu()).all() and not a.overflow
    ra = a.cpu()
    for g in range(len(a)):
        b = oracle.new_board()
        for i in range(int(ra.lens[g])):
            assert oracle.lib().go_board_check_move(C.byref(b), int(ra.moves[g, i]))
            oracle.lib().go_board_apply(C.byref(b), int(ra.moves[g, i]), 1)
        assert b.cur_player == 0 and b.winner == int(ra.winner[g])
    states, values, pi = a.to_samples(first_move=2)
    assert states.shape[0] == int((ra.lens - 2).sum()) and float(pi.sum(1).sub(1).abs().max()) < 1e-3
    c = selfplay.play_network_games(6, net, 40, opening_plies=2, first_game_id=9, seed=3, reuse_subtree=False, root_noise=None)
    d = selfplay.play_network_games(6, fused, 40, opening_plies=2, first_game_id=9, seed=3, reuse_subtree=False, root_noise=None)
    same = (c.moves.cpu() == d.moves.cpu()).all(1)
    assert int(same.sum()) >= 4                              # a tie between two children decided by 1e-7 may send a game elsewhere
    # 10 games through 3 slots (fresh roots): the same games as all at once (the search of a game does not depend on its slot)
    e = selfplay.play_network_games(10, fused, 30, opening_plies=2, first_game_id=40, seed=3, reuse_subtree=False, root_noise=None, slots=3)
    f = selfplay.play_network_games(10, fused, 30, opening_plies=2, first_game_id=40, seed=3, reuse_subtree=False, root_noise=None)
    assert (e.moves.cpu() == f.moves.cpu()).all() and (e.winner.cpu() == f.winner.cpu()).all() and int(e.lens.min()) >= 9
    fused.close()


def test_supervisor_self_play_with_slots(oracle):
    """Continuous batching: 14 games through 4 slots (a finished game hands its slot to the next one): every game is legal and
    finished, and the pattern-guided games are the ones the all-at-once run plays unless the evaluator's history-dependent flag
    words (which differ between a fresh and a handed-over slot) tip a search."""
    import ctypes as C
    from gomokuai_amd import selfplay
    a = selfplay.play_supervisor_games(14, 80, opening_plies=3, first_game_id=30, slots=4)
    b2 = selfplay.play_supervisor_games(14, 80, opening_plies=3, first_game_id=30, slots=4)
    full = selfplay.play_supervisor_games(14, 80, opening_plies=3, first_game_id=30)
    assert (a.moves.cpu() == b2.moves.cpu()).all() and not a.overflow
    ra = a.cpu()
    for g in range(len(a)):
        b = oracle.new_board()
        for i in range(int(ra.lens[g])):
            assert oracle.lib().go_board_check_move(C.byref(b), int(ra.moves[g, i]))
            oracle.lib().go_board_apply(C.byref(b), int(ra.moves[g, i]), 1)
        assert b.cur_player == 0 and b.winner == int(ra.winner[g]) and int(ra.lens[g]) >= 9
    same = (a.moves.cpu() == full.moves.cpu()).all(1)
    assert int(same.sum()) >= 10
    states, _, _ = a.to_samples(first_move=3)
    assert states.shape[0] == int((ra.lens - 3).sum())
    # slots with kept subtrees and root noise (a handed-over slot starts from a new root, the others keep stepping): legal, finished, reproducible
    kw = dict(opening_plies=2, first_game_id=50, slots=3, reuse_subtree=True, root_noise=(0.05, 0.25), seed=12)
    for policy, c in (("traditional", 5.0), ("poolrave", 2.0)):
        c1 = selfplay.play_supervisor_games(9, 90, c_puct=c, policy=policy, **kw)
        c2 = selfplay.play_supervisor_games(9, 90, c_puct=c, policy=policy, **kw)
        assert (c1.moves.cpu() == c2.moves.cpu()).all() and not c1.overflow
        rc = c1.cpu()
        for g in range(len(c1)):
            b = oracle.new_board()
            for i in range(int(rc.lens[g])):
                assert oracle.lib().go_board_check_move(C.byref(b), int(rc.moves[g, i]))
                oracle.lib().go_board_apply(C.byref(b), int(rc.moves[g, i]), 1)
            assert b.cur_player == 0 and b.winner == int(rc.winner[g]) and int(rc.lens[g]) >= 9


def test_slots_do_not_change_noisy_supervisor_games():
    """Root noise (and PoolRAVE's rollouts) are keyed by the GAME's id, not by the slot it happens to run in: games played through a
    few slots equal the same games played all side by side, noise included; and the games of a slot are not copies of each other."""
    for policy, playouts in (("traditional", 60), ("poolrave", 40)):
        # (Default::AddNoise only touches roots that have children: kept subtrees, i.e. every search of a game but its first)
        kw = dict(c_puct=5.0 if policy == "traditional" else 2.0, seed=1234, first_game_id=50, opening_plies=0, root_noise=(0.3, 0.25), policy=policy, reuse_subtree=True)
        a = selfplay.play_supervisor_games(9, playouts, **kw).cpu()
        b = selfplay.play_supervisor_games(9, playouts, slots=3, **kw).cpu()
        assert (a.lens == b.lens).all() and (a.moves == b.moves).all() and (a.winner == b.winner).all(), policy
        games = {tuple(int(x) for x in b.moves[g, :int(b.lens[g])]) for g in range(9)}
        assert len(games) > 3, policy                    # nine different games, not three sets of copies


@pytest.mark.parametrize("reuse,noise", [(True, (0.05, 0.25)), (True, None), (False, None)])
def test_device_resident_network_loop_plays_the_host_loops_games(reuse, noise):
    """gmk_az_advance (MCTS::stepForward's move, the record with the root's visit counts, Board::applyMove's victory check and the
    re-rooting as ONE kernel per ply) against the host-driven loop it replaces (numpy boards, root statistics down and moves up every
    ply): the same moves, lengths, winners and visit counts, with kept subtrees and root noise (the reference agent's semantics), with
    kept subtrees alone and with new roots; a move cap cuts both the same way."""
    from gomokuai_amd.network import FusedPolicyValueNetwork, PolicyValueNetwork
    from helpers import PaddedNetwork
    net = FusedPolicyValueNetwork(PolicyValueNetwork(seed=6).cuda().eval())
    fused = PaddedNetwork(net, 9)                                # (the device loop hands the network the games still played only)
    kw = dict(opening_plies=2, first_game_id=21, seed=5, reuse_subtree=reuse, root_noise=noise)
    dev = selfplay.play_network_games(9, fused, 24, device_loop=True, **kw)
    host = selfplay.play_network_games(9, fused, 24, device_loop=False, **kw)
    rd, rh = dev.cpu(), host.cpu()
    assert not dev.overflow and not host.overflow
    assert (rd.lens == rh.lens).all() and (rd.winner == rh.winner).all() and int(rd.lens.min()) >= 9
    for g in range(9):
        n = int(rd.lens[g])
        assert (rd.moves[g, :n] == rh.moves[g, :n]).all()
        assert (rd.visits[g, :n] == rh.visits[g, :n]).all()
    cd = selfplay.play_network_games(9, fused, 24, device_loop=True, max_moves=5, **kw).cpu()
    ch = selfplay.play_network_games(9, fused, 24, device_loop=False, max_moves=5, **kw).cpu()
    assert (cd.lens == ch.lens).all() and int(cd.lens.max()) <= 7 and (cd.moves[:, :7] == ch.moves[:, :7]).all()
    net.close()


def test_network_self_play_through_slots_on_the_device():
    """gmk_az_set_slots: eleven games through four slots play the games that eleven slots play (a game's search does not depend on the
    slot that runs it, its root noise is keyed by its global id): with kept subtrees and root noise, and with new roots; the host-driven
    slot loop (new roots only) agrees as well."""
    from gomokuai_amd.network import FusedPolicyValueNetwork, PolicyValueNetwork
    from helpers import PaddedNetwork
    net = FusedPolicyValueNetwork(PolicyValueNetwork(seed=8).cuda().eval())
    fused = PaddedNetwork(net, 11)
    for reuse, noise in ((True, (0.05, 0.25)), (False, None)):
        kw = dict(opening_plies=2, first_game_id=70, seed=2, reuse_subtree=reuse, root_noise=noise)
        few = selfplay.play_network_games(11, fused, 20, slots=4, **kw).cpu()
        full = selfplay.play_network_games(11, fused, 20, **kw).cpu()
        assert (few.lens == full.lens).all() and (few.winner == full.winner).all() and int(few.lens.min()) >= 9
        for g in range(11):
            n = int(few.lens[g])
            assert (few.moves[g, :n] == full.moves[g, :n]).all() and (few.visits[g, :n] == full.visits[g, :n]).all()
    host = selfplay.play_network_games(11, fused, 20, slots=4, device_loop=False, opening_plies=2, first_game_id=70, seed=2, reuse_subtree=False, root_noise=None).cpu()
    assert (host.lens == few.lens).all() and (host.moves == few.moves).all()
    net.close()


def test_network_self_play_reports_a_full_arena():
    """K7 reports a full node arena as status bit 1 (value 2): play_network_games must pass it on as GameRecords.overflow."""
    import torch
    from gomokuai_amd.network import PolicyValueNetwork
    net = PolicyValueNetwork(seed=3).cuda().eval()
    ok = selfplay.play_network_games(4, net, 12, opening_plies=2, reuse_subtree=False, root_noise=None, max_moves=6)
    assert not ok.overflow
    tight = selfplay.play_network_games(4, net, 12, opening_plies=2, reuse_subtree=False, root_noise=None, max_moves=6, node_capacity=600)
    assert tight.overflow


def test_a_full_arena_in_an_early_game_of_a_slot_is_reported():
    """Continuous batching: the arena-full bit of a slot must survive the hand-over to the slot's next game (it was cleared with the
    rest of the status word, so only an overflow in a slot's LAST game was seen).  Twelve games through three slots with room for ~a fifth of a
    search's nodes: every search overflows, the records say so; with the default capacity they do not."""
    small = selfplay.play_games(12, 60, first_game_id=5, slots=3, handles=1, node_capacity=2048)
    assert small.overflow
    assert (small.cpu().lens > 0).all()
    fine = selfplay.play_games(12, 60, first_game_id=5, slots=3, handles=1)
    assert not fine.overflow


def test_a_move_cap_is_honoured_whatever_the_batching():
    """max_moves truncates the games on every path (the device-resident loop plays whole games only: a capped call must not take it)."""
    rec = selfplay.play_games(10, 40, first_game_id=3, slots=4, handles=1, max_moves=7).cpu()
    assert int(rec.lens.max()) <= 7


@pytest.mark.parametrize("policy,c_puct", [("traditional", 5.0), ("poolrave", 2.0)])
@pytest.mark.parametrize("reuse,noise", [(False, None), (True, None), (True, (0.05, 0.25))])
def test_device_resident_supervisor_loop_plays_the_host_loops_games(policy, c_puct, reuse, noise):
    """gmk_trad_selfplay_run (search, MCTS::stepForward's move, end-of-game check and slot hand-over as kernels, the records written by game
    id on the device) against the host-driven loops it replaces (numpy boards, root_stats down and positions up every ply): the same moves,
    lengths, winners and per-ply root visit counts, all games at once and through a few slots (continuous batching: a finished game's slot
    goes to the next unstarted game in slot order, as the host loop does it)."""
    kw = dict(c_puct=c_puct, policy=policy, opening_plies=2, first_game_id=21, seed=99, reuse_subtree=reuse, root_noise=noise)
    for slots in (None, 3):
        host = selfplay.play_supervisor_games(10, 70, slots=slots, device_loop=False, **kw).cpu()
        dev = selfplay.play_supervisor_games(10, 70, slots=slots, device_loop="lockstep", **kw)
        assert not dev.overflow and not host.overflow
        dev = dev.cpu()
        assert (dev.lens == host.lens).all() and (dev.winner == host.winner).all(), (policy, reuse, noise, slots)
        assert (dev.moves == host.moves).all() and (dev.visits == host.visits).all(), (policy, reuse, noise, slots)
        assert int(dev.lens.min()) >= 9


def test_persistent_supervisor_loop_plays_the_games_of_the_all_at_once_loop(oracle):
    """gmk_trad_selfplay_run with persistent = 1: one launch, every slot's wavefront plays game after game at its own pace and takes the
    next unstarted game from a counter; a game starts on a fresh evaluator, so -- whatever slot it lands in, whatever the order the slots
    finish in -- its record is the one the host-driven loop plays with all games side by side: moves, lengths, winners, per-ply root visit
    counts; twice the same; every game legal and finished on the oracle's board."""
    kw = dict(c_puct=5.0, policy="traditional", opening_plies=2, first_game_id=61, seed=7)
    host = selfplay.play_supervisor_games(23, 60, device_loop=False, **kw).cpu()
    for slots in (5, 23, 40):
        a = selfplay.play_supervisor_games(23, 60, slots=slots, device_loop="persistent", **kw)
        assert not a.overflow
        a = a.cpu()
        assert (a.lens == host.lens).all() and (a.winner == host.winner).all(), slots
        assert (a.moves == host.moves).all() and (a.visits == host.visits).all(), slots
    legal, end_ply, winner = oracle.replay_games(a.moves.numpy(), a.lens.numpy())
    assert legal.all() and (end_ply == a.lens.numpy()).all() and (winner == a.winner.numpy()).all()
    with pytest.raises(ValueError):          # host-drawn noise cannot come inside the one launch
        selfplay.play_supervisor_games(4, 10, device_loop="persistent", reuse_subtree=True, root_noise=(0.05, 0.25), noise_sampler="std")


def _oracle_supervisor_game(O, game_id, playouts, seed, opening, noise=None, c_puct=5.0):
    """One MCTS(TraditionalPolicy) object for the whole game, as agents/mcts.py:17-21 drives it: runPlayouts on the kept tree (syncWithBoard,
    AddNoise with the counter-based sampler, the playouts), then stepForward()'s move on the board."""
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


@pytest.mark.parametrize("noise", [None, (0.05, 0.25)])
def test_supervisor_reference_semantics_in_one_launch(oracle, noise):
    """K6 with the reference agent's per-move semantics -- the chosen child's subtree kept (MCTS.cpp:129-147), Default::AddNoise before every search
    (MCTS.cpp:179-183) -- inside ONE persistent launch (the subtree compacted into the slot's other arena by the wavefront, the noise drawn by it
    from the counter-based sampler) == the host-driven loop with the same sampler (all games side by side) == the oracle's kept-tree game loop:
    moves, lengths, winners, per-ply root visit counts; 23 games through 5, 23 and 40 slots."""
    n, playouts, seed, first = 23, 60, 7, 61
    kw = dict(c_puct=5.0, policy="traditional", opening_plies=2, first_game_id=first, seed=seed, reuse_subtree=True, root_noise=noise)
    host = selfplay.play_supervisor_games(n, playouts, device_loop=False, **kw).cpu()
    assert not host.overflow
    for slots in (5, 23, 40):
        a = selfplay.play_supervisor_games(n, playouts, slots=slots, device_loop="persistent", **kw)
        assert not a.overflow
        a = a.cpu()
        assert (a.lens == host.lens).all() and (a.winner == host.winner).all(), slots
        assert (a.moves == host.moves).all() and (a.visits == host.visits).all(), slots
    m, l, _ = G.synth_boards(n, 0, seed=seed, first_board=first)
    for g in range(0, n, 3):
        opening = [int(x) for x in m[g, :min(int(l[g]), 2)]]
        moves, visits, winner = _oracle_supervisor_game(oracle, first + g, playouts, seed, opening, noise)
        L = int(a.lens[g])
        assert [int(x) for x in a.moves[g, :L]] == moves, "game %d" % g
        assert int(a.winner[g]) == winner
        for t, v in enumerate(visits):
            assert (a.visits[g, len(opening) + t].numpy().astype(np.uint32) == np.minimum(v, 65535)).all(), "game %d move %d" % (g, t)


def test_supervisor_loop_stops_after_max_steps():
    """max_steps (the throughput measurements' switch, lock-step form of the device loop): after that many moves per slot the loop ends, games
    still running keep the moves they have (winner 0), and those moves are the first moves of the whole games."""
    whole = selfplay.play_supervisor_games(6, 50, opening_plies=2, first_game_id=9, device_loop="lockstep").cpu()
    cut = selfplay.play_supervisor_games(6, 50, opening_plies=2, first_game_id=9, device_loop="lockstep", max_steps=3).cpu()
    assert (cut.lens == 5).all() and (cut.winner == 0).all()
    assert (cut.moves[:, :5] == whole.moves[:, :5]).all() and (cut.visits[:, :5] == whole.visits[:, :5]).all()
    with pytest.raises(ValueError):
        selfplay.play_supervisor_games(2, 10, device_loop=False, max_steps=3)
