"""CorePyExt on the GPU: the reference-shaped agent loop (MCTSAgent.eval_state -> dual_play) runs its
searches through gmk_mcts_* and agrees with the CPU oracle on the same seeds."""
import ctypes as C

import numpy as np
import pytest

from gomokuai_amd import core

import helpers

pytestmark = pytest.mark.gpu


def test_eval_state_matches_oracle(oracle):
    O = oracle
    core.set_seed(12345)
    m = core.MCTS(c_iterations=300, policy=core.RandomPolicy(5.0, 5))
    b = core.Board()
    ob = O.new_board()
    for mv in (112, 98, 127):
        b.apply_move(core.Position(mv))
        O.lib().go_board_apply(C.byref(ob), mv, 1)
    q, pi = m.eval_state(b)
    om = O.MCTS(300, 5.0, 5, 12345, 0)                              # first MCTS after set_seed has game id 0
    oq, opi, ovisits = om.eval_state(ob)
    assert np.float32(q).tobytes() == np.float32(oq).tobytes()
    assert [c.node_visits for c in m.root.children] == [int(v) for i, v in enumerate(ovisits) if ob.states[1][i]]
    assert np.abs(pi - opi).max() <= 1e-6                           # float32 log/exp order is unpinned (Eigen): tolerance
    assert m.size == om.size and m.iterations == 300 and len(b.move_record) == 3
    m.step_forward()
    assert m.root.position.id == int(np.argmax(ovisits))


def test_config1_one_game_1000_playouts(oracle):
    """BASELINE.json configs[0] as it is stated: a single 15x15 game from the empty board, pure-rollout MCTS of 1 000 playouts per move
    (agents/mcts.py: RandomMCTSAgent -> MCTS(c_iterations=1000, RandomPolicy(c_puct=5, c_rollouts=5))), through CorePyExt: root visit counts,
    the bits of Q and the tree size of every search equal the oracle's MCTS.cpp restatement on the same seeds, the subtree kept between
    moves as MCTS::stepForward does and the root noise of MCTS.cpp:182 in (both sides draw it from the same counter-based stream)."""
    O = oracle
    core.set_seed(20241004)
    m = core.MCTS(c_iterations=1000, policy=core.RandomPolicy(5.0, 5))
    om = O.MCTS(1000, 5.0, 5, 20241004, 0)
    om.set_noise(0.05, 0.25)                                        # Default::AddNoise's defaults (MonteCarlo.hpp:97), mixed in before every search (MCTS.cpp:182)
    b, ob = core.Board(), O.new_board()
    for ply in range(4):
        q, pi = m.eval_state(b)
        om.sync_with_board(ob)
        oq, opi, ovisits = om.eval_state(ob)
        assert np.float32(q).tobytes() == np.float32(oq).tobytes(), ply
        assert [c.node_visits for c in m.root.children] == [int(v) for i, v in enumerate(ovisits) if ob.states[1][i]], ply
        assert m.size == om.size and m.iterations == 1000, ply
        assert np.abs(pi - opi).max() <= 1e-6
        m.step_forward()
        mv = om.step_forward()
        assert m.root.position.id == mv, ply
        b.apply_move(core.Position(mv))
        O.lib().go_board_apply(C.byref(ob), mv, 1)


def test_self_play_game_and_training_tuples():
    core.set_seed(7)
    agent = helpers.random_searcher(5.0, 5, c_iterations=60)
    data = helpers.play_game(agent, agent, record=True)
    assert len(data) >= 9
    states, score, probs = data[0]
    assert states.shape == (6, 15, 15) and states.dtype == np.uint8 and probs.shape == (225,) and probs.dtype == np.float32
    assert abs(float(probs.sum()) - 1.0) < 1e-3 and float(score) in (-1.0, 0.0, 1.0)
    assert states[2].sum() == 225 and states[5].all()               # empty board, black to move
    winner_scores = {float(s) for _, s, _ in data}
    assert winner_scores <= {-1.0, 0.0, 1.0}
    aug = helpers.augment(data)
    assert len(aug) == 8 * len(data)
    # same seed, same game
    core.set_seed(7)
    agent2 = helpers.random_searcher(5.0, 5, c_iterations=60)
    again = helpers.play_game(agent2, agent2, record=True)
    assert len(again) == len(data) and all((a[2] == b[2]).all() for a, b in zip(data, again))


def test_duration_constraint():
    import datetime
    m = core.MCTS(c_duration=datetime.timedelta(milliseconds=100))
    mv = m.get_action(core.Board())
    assert 0 <= mv.id < 225 and m.iterations >= 256 and m.size > 1


def test_traditional_policy_matches_oracle(oracle):
    """MCTS(policy=TraditionalPolicy) searches on the device (K6) and agrees with the oracle restatement move after move of
    an agent loop: the tree is kept (stepForward), Dirichlet noise goes into the root priors before every search, the
    policy's evaluator persists; including which of several equally visited children is played."""
    O = oracle
    core.set_seed(99)
    core.set_root_noise(0.05, 0.25)
    m = core.MCTS(c_iterations=400, policy=core.TraditionalPolicy(5.0))
    om = O.TraditionalMCTS(5.0)
    om.set_noise(0.05, 0.25, 99, 0)
    b = core.Board()
    played = []
    for mv in (112, 98, 127, 113):
        b.apply_move(core.Position(mv)); played.append(mv)
    kept = []
    for _ in range(5):
        q, pi = m.eval_state(b)
        om.run(played, 400)
        v, oq, p, best = om.root_children()
        assert np.float32(q).tobytes() == np.float32(om.root_value).tobytes()
        kids = {c.position.id: c for c in m.root.children}
        assert sorted(kids) == [int(i) for i in np.nonzero(p)[0]]
        assert all(kids[i].node_visits == int(v[i]) and np.float32(kids[i].action_prob).tobytes() == p[i].tobytes() for i in kids)
        assert abs(float(pi.sum()) - 1.0) < 1e-3
        kept.append(int(m.root.node_visits))
        m.step_forward()
        assert m.root.position.id == best == om.step_forward()
        b.apply_move(m.root.position); played.append(best)
    assert max(kept[1:]) > 400                           # visits carried over with the kept subtree


def test_traditional_agent_plays_a_game():
    agent = helpers.pattern_searcher(5.0, c_iterations=150)
    data = helpers.play_game(agent, agent, record=True)
    assert len(data) >= 9 and data[0][0][2].sum() == 225
    with pytest.raises(RuntimeError):
        core.MCTS(c_iterations=10, policy=core.TraditionalPolicy(5.0, 0.0, True)).get_action(core.Board())


def test_poolrave_policy_matches_oracle(oracle):
    """MCTS(policy=PoolRAVEPolicy) searches on the device (K8) and agrees with the oracle restatement move after move of an
    agent loop with the tree kept and Dirichlet noise at the root, then a whole game of that searcher against itself."""
    O = oracle
    core.set_seed(31337)
    core.set_root_noise(0.05, 0.25)
    m = core.MCTS(c_iterations=300, policy=core.PoolRAVEPolicy(2.0, 0.0))
    om = O.PoolRAVEMCTS(2.0, 0.0, seed=31337, game_id=0)
    om.set_noise(0.05, 0.25)
    b = core.Board()
    played = []
    for mv in (112, 98, 127):
        b.apply_move(core.Position(mv)); played.append(mv)
    kept = []
    for _ in range(4):
        q, pi = m.eval_state(b)
        om.run(played, 300)
        v, oq, p, av, aq, best = om.root_children()
        assert np.float32(q).tobytes() == np.float32(om.root_value).tobytes()
        kids = {c.position.id: c for c in m.root.children}
        assert sorted(kids) == [int(i) for i in np.nonzero(p)[0]]
        assert all(kids[i].node_visits == int(v[i]) and np.float32(kids[i].state_value).tobytes() == oq[i].tobytes() for i in kids)
        kept.append(int(m.root.node_visits))
        m.step_forward()
        assert m.root.position.id == best == om.step_forward()
        b.apply_move(m.root.position); played.append(best)
    assert max(kept[1:]) > 300
    agent = helpers.rave_searcher(2.0, 0.0, c_iterations=100)
    data = helpers.play_game(agent, agent, record=True)
    assert len(data) >= 9 and "PoolRAVEPolicy" in repr(agent.mcts.policy)


def test_agent_loop_keeps_the_tree_and_adds_root_noise(oracle):
    """The reference's MCTSAgent keeps its tree from move to move (syncWithBoard / stepForward, MCTS.cpp:119-147) and mixes
    Dirichlet noise into the root priors before every search (MCTS.cpp:182).  CorePyExt.MCTS does both on the device; the
    oracle's MCTS object, driven the same way with the same seeds, sees the same visit counts move after move."""
    O = oracle
    core.set_seed(2024)
    core.set_root_noise(0.05, 0.25)
    m = core.MCTS(c_iterations=150, policy=core.RandomPolicy(5.0, 5))
    om = O.MCTS(150, 5.0, 5, 2024, 0)
    om.set_noise(0.05, 0.25)
    b, ob = core.Board(), O.new_board()
    for mv in (112, 113):
        b.apply_move(core.Position(mv)); O.lib().go_board_apply(C.byref(ob), mv, 1)
    sizes = []
    for ply in range(6):
        q, pi = m.eval_state(b)
        oq, opi, ovisits = om.eval_state(ob)
        assert np.float32(q).tobytes() == np.float32(oq).tobytes(), "move %d" % ply
        assert [c.node_visits for c in m.root.children] == [int(v) for i, v in enumerate(ovisits) if ob.states[1][i]], "move %d" % ply
        assert m.size == om.size
        sizes.append(int(m.root.node_visits))
        m.step_forward()
        assert m.root.position.id == om.step_forward()
        b.apply_move(m.root.position)
        O.lib().go_board_apply(C.byref(ob), m.root.position.id, 1)
        if b.status["is_end"]:
            break
    assert max(sizes[1:]) > 150                         # the kept subtree brings visits along: more than one search's worth at the root
    # an unrelated position starts a fresh tree
    b2 = core.Board(); b2.apply_move(core.Position(0))
    m.eval_state(b2)
    assert m.root.node_visits == 150


def test_policy_with_python_evaluator_matches_oracle(oracle):
    """MCTS(policy=Policy(eval_state=f, c_puct)) (agents/alphazero.py:5-9): the search runs on the device (K7), f is called with
    the leaf Board once per playout; with the same f the oracle's MCTS sees the same tree."""
    from test_az_gpu import surrogate
    O = oracle
    calls = []

    def eval_state(board):
        calls.append(len(board.move_record))
        return surrogate(board.encoded_states())
    core.set_seed(31)
    core.set_root_noise(0.05, 0.25)
    m = core.MCTS(c_iterations=120, policy=core.Policy(eval_state=eval_state, c_puct=4.0))
    b, ob = core.Board(), O.new_board()
    for mv in (112, 98, 127, 113, 96):
        b.apply_move(core.Position(mv)); O.lib().go_board_apply(C.byref(ob), mv, 1)
    q, pi = m.eval_state(b)
    om = O.MCTS(120, 4.0, 5, 31, 0)
    om.set_evaluator(surrogate)
    oq, opi, ovisits = om.eval_state(ob)
    assert np.float32(q).tobytes() == np.float32(oq).tobytes()
    kids = {c.position.id: c.node_visits for c in m.root.children}
    assert all(kids.get(i, 0) == int(v) for i, v in enumerate(ovisits)) and sum(kids.values()) == 119
    assert len(calls) == 120 and min(calls) == 5 and max(calls) > 5 and len(b.move_record) == 5
    m.step_forward()
    assert m.root.position.id == int(np.argmax(ovisits)) == om.step_forward()
    # the next search of the agent loop starts from the kept subtree, with root noise (defaults of the reference)
    b.apply_move(m.root.position); O.lib().go_board_apply(C.byref(ob), m.root.position.id, 1)
    om.set_noise(0.05, 0.25)
    q, pi = m.eval_state(b)
    oq, opi, ovisits = om.eval_state(ob)
    assert np.float32(q).tobytes() == np.float32(oq).tobytes() and m.root.node_visits == om.root_visits > 120
    kids = {c.position.id: c.node_visits for c in m.root.children}
    assert all(kids.get(i, 0) == int(v) for i, v in enumerate(ovisits))


def test_py_conv_net_agent_plays():
    import torch
    from gomokuai_amd.network import PolicyValueNetwork
    net = PolicyValueNetwork(seed=5).cuda().eval()
    agent = helpers.network_searcher(net.eval_state, 5.0, c_iterations=24)
    b = core.Board()
    q, pi, mv = agent.evaluate(b)
    assert abs(float(pi.sum()) - 1.0) < 1e-3 and 0 <= mv.id < 225 and -1.0 <= q <= 1.0


def _children_of(m):
    return [(c.position.id, c.node_visits, np.float32(c.state_value).tobytes(), np.float32(c.action_prob).tobytes()) for c in m.root.children]


def _default_select(c_puct):
    """Default::Select (core/lib/include/algorithms/MonteCarlo.hpp:23-28, 57-68) restated in Python: double arithmetic, first maximum."""
    import math

    def select(node):
        best, best_child = -1.0, node.children[0]
        root_n = math.sqrt(float(node.node_visits))
        for child in node.children:
            score = float(child.state_value) + c_puct * float(child.action_prob) * root_n / float(child.node_visits + 1)
            if score > best:
                best, best_child = score, child
        return best_child
    return select


def _default_back_prop(node, board, value):
    """Default::BackPropogate (MonteCarlo.hpp:90-95) restated in Python, in float32 like the reference's Node fields."""
    v = np.float32(value)
    while node is not None:
        node.node_visits += 1
        q = np.float32(node.state_value)
        node.state_value = float(q + (v - q) / np.float32(node.node_visits))
        node, v = node.parent, -v


def test_python_select_plays_the_default_policys_game():
    """Policy(select=f) -- a Python callable in MCTS::playout's select stage (core/py_ext/src/mcts_ext.hpp:43-61).  With f =
    Default::Select restated in Python and the other stages left to their defaults (Default::Simulate = one random rollout, which runs
    on the device on the draws of gmk_mcts_*'s first rollout lane), the search must be MCTS(RandomPolicy(c_puct, 1))'s search, move after
    move: the root's children (visits, value bits, priors), its value, the tree's size; the subtree is kept between the moves."""
    core.set_root_noise(alpha=0.0)
    try:
        calls = [0]
        inner = _default_select(5.0)

        def counting_select(node):
            calls[0] += 1
            assert not node.is_leaf() and all(c.parent is node for c in node.children)
            return inner(node)
        results = []
        for policy_of in (lambda: core.RandomPolicy(5.0, 1), lambda: core.Policy(select=counting_select, c_puct=5.0)):
            core.set_seed(4242)                                     # the same seed and game number 0 for both searchers
            m = core.MCTS(c_iterations=150, policy=policy_of())
            b = core.Board()
            for mv in (112, 113, 97):
                b.apply_move(core.Position(mv))
            trace = []
            for _ in range(3):
                q, pi = m.eval_state(b)
                # (the RandomPolicy search reports its children's visits and priors, not their values: compared without them)
                trace.append((np.float32(q).tobytes(), [(c[0], c[1], c[3]) for c in _children_of(m)], m.size, np.float32(m.root.state_value).tobytes(), m.root.node_visits))
                m.step_forward()
                b.apply_move(m.root.position)
            results.append(trace)
        for move, (x, y) in enumerate(zip(results[0], results[1])):
            assert x[0] == y[0] and x[2:] == y[2:], "move %d: root value / size / visits %r against %r" % (move, (x[0], x[2:]), (y[0], y[2:]))
            assert x[1] == y[1], "move %d: children differ, first at %r" % (move, next((a, b_) for a, b_ in zip(x[1], y[1]) if a != b_))
        assert calls[0] > 300                                       # the callable really drove the descents
    finally:
        core.set_root_noise(alpha=0.05, epsilon=0.25)


def test_python_back_prop_and_expand_stages():
    """Policy(back_prop=f, eval_state=g): with f = Default::BackPropogate restated in Python (float32, walking Node.parent) the search equals
    Policy(eval_state=g) whose backup runs on the device; select and back_prop together too.  Policy(expand=h): h is called once per
    simulated leaf with (node, board, probabilities), its return value counts into MCTS.size, and -- as in the reference, where a Python
    callable has no way to attach children -- the node stays a leaf."""
    from test_az_gpu import surrogate
    core.set_root_noise(alpha=0.0)
    try:
        def evaluator(board):
            value, probs = surrogate(np.asarray(board.encoded_states(), dtype=np.float32))
            return float(value), probs
        traces = []
        for kw in ({}, {"back_prop": _default_back_prop}, {"select": _default_select(4.0), "back_prop": _default_back_prop}):
            core.set_seed(99)
            m = core.MCTS(c_iterations=120, policy=core.Policy(eval_state=evaluator, c_puct=4.0, **kw))
            b = core.Board()
            for mv in (112, 98):
                b.apply_move(core.Position(mv))
            trace = []
            for _ in range(2):
                q, pi = m.eval_state(b)
                trace.append((np.float32(q).tobytes(), _children_of(m), m.size, m.root.node_visits))
                m.step_forward()
                b.apply_move(m.root.position)
            traces.append(trace)
        assert traces[0] == traces[1] == traces[2]
        seen = []

        def expand(node, board, probs):
            seen.append((node.position.id, len(board.move_record), float(np.asarray(probs).sum())))
            return 7
        core.set_seed(5)
        m = core.MCTS(c_iterations=20, policy=core.Policy(expand=expand, eval_state=evaluator, c_puct=4.0))
        b = core.Board()
        b.apply_move(core.Position(112))
        m.eval_state(b)
        assert len(seen) == 20 and all(s[1] == 1 for s in seen) and m.size == 1 + 7 * 20
        assert m.root.is_leaf() and m.root.node_visits == 20
    finally:
        core.set_root_noise(alpha=0.05, epsilon=0.25)


def test_python_select_on_a_nearly_full_board():
    """The same equality from a position with 11 empty cells: the descents end at finished games (fives, full boards), whose values the
    host-driven loop backs up itself (CalcScore(node.player, winner), MCTS.cpp:174), and at leaves whose rollouts fill the board."""
    rng = np.random.RandomState(31)
    cls = lambda c: ((c % 15) // 2 + c // 15) % 2             # two colour classes that never line up five
    blacks, whites = [c for c in range(225) if cls(c) == 0], [c for c in range(225) if cls(c) == 1]
    bl, wh = list(rng.permutation(blacks)), list(rng.permutation(whites))
    seq = []
    while bl or wh:
        if bl:
            seq.append(int(bl.pop()))
        if wh:
            seq.append(int(wh.pop()))
    core.set_root_noise(alpha=0.0)
    try:
        results = []
        for policy_of in (lambda: core.RandomPolicy(5.0, 1), lambda: core.Policy(select=_default_select(5.0), c_puct=5.0)):
            core.set_seed(77)
            m = core.MCTS(c_iterations=200, policy=policy_of())
            b = core.Board()
            for mv in seq[:214]:
                b.apply_move(core.Position(mv))
            q, pi = m.eval_state(b)
            results.append((np.float32(q).tobytes(), [(c.position.id, c.node_visits) for c in m.root.children], m.size, m.root.node_visits))
        assert results[0] == results[1] and len(results[0][1]) == 11
    finally:
        core.set_root_noise(alpha=0.05, epsilon=0.25)
