"""The front ends with real searches: a Botzone-protocol driver (tests/helpers.py: BotDriver) starts
`python -m gomokuai_amd.interface botzone` as the external bot program, the way the reference's Python side talks to its C++
bot (agents/botzone.py:27-41, core/interface/src/Interface.h:9-31); the keep-alive bot and a console match run in-process."""
import io
import json
import os
import sys

import pytest

from gomokuai_amd import core, interface

import helpers

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_botzone_agent_drives_the_bot_program():
    bot = helpers.BotDriver("%s -m gomokuai_amd.interface botzone --agent traditional:5 --iterations 200 --seed 3" % sys.executable, cwd=ROOT)
    b = core.Board()
    for mv in (112, 113, 127):
        b.apply_move(core.Position(mv))
    move = bot.move(b)                                             # white to move
    assert b.check_move(move)
    # the same search in this process picks the same move (the pattern-guided search is deterministic without noise at a fresh root)
    core.set_seed(3)
    local = interface.make_agent("traditional:5", iterations=200, quiet=True)
    local.sync_with_board(b)
    assert local.get_action(b).id == move.id
    assert local.debug_message()["iterations"] == 200


def test_keep_alive_and_console_with_searches():
    core.set_seed(11)
    agent = interface.make_agent("poolrave:2", iterations=150, quiet=True)
    lines = [json.dumps({"requests": [{"x": 7, "y": 7}], "responses": []}), json.dumps({"x": 8, "y": 8})]
    out = io.StringIO()
    interface.keep_alive_botzone_interface(agent, io.StringIO("\n".join(lines) + "\n"), out)
    answers = [json.loads(a) for a in out.getvalue().split(">>>BOTZONE_REQUEST_KEEP_RUNNING<<<\n") if a.strip()]
    assert len(answers) == 2 and all(0 <= a["response"]["x"] < 15 and a["debug"]["iterations"] == 150 for a in answers)
    out = io.StringIO()
    tie = interface.console_interface(interface.make_agent("traditional:5", iterations=120, quiet=True),
                                      interface.make_agent("random-mcts:5:5", iterations=120, quiet=True), out, black_player=0)
    record = json.loads(out.getvalue().strip().splitlines()[-1])
    assert tie in (0, 1) and len(record["responses"]) >= 5 and ("Game end" in out.getvalue() or "Tie." in out.getvalue())


def test_pattern_eval_agent(oracle):
    """PatternEvalAgent (Agent.h:108-161) = the policy head of K6 without a search: its move is the first maximum of the oracle's
    EvaluationProbs + DecisiveFilter on the same position, its debug message the pattern / compound counts of K1."""
    import numpy as np
    from gomokuai_amd import lib as G
    agent = interface.make_agent("pattern")
    b = core.Board()
    agent.sync_with_board(b)
    first = agent.get_action(b)
    assert (first.x, first.y) == (7, 7)
    moves, lens, _ = G.synth_boards(12, 1, first_board=4000)
    checked = 0
    for g in range(12):
        ml = [int(m) for m in moves[g, :max(1, int(lens[g]) - 2)]]
        b = core.Board()
        for m in ml:
            b.apply_move(core.Position(m))
        if b.status["is_end"]:
            continue
        agent.sync_with_board(b)
        mv = agent.get_action(b)
        probs, _ = oracle.trad_heuristic(ml)
        assert mv.id == int(np.argmax(probs)) and b.check_move(mv), "game %d" % g
        dbg = agent.debug_message()
        mm = np.zeros((1, 64), np.uint8); mm[0, :len(ml)] = ml
        totals = oracle.replay_batch(mm, np.array([len(ml)], np.int32))[2][0]
        assert dbg["before"]["black"][0] == [int(t >> 16) for t in totals[:8]] and dbg["before"]["white"][1] == [int(t & 0xFFFF) for t in totals[8:11]]
        checked += 1
    assert checked >= 8
