"""The front ends with real searches: agents/botzone.py's BotzoneAgent (agents.BotzoneAgent) starts
`python -m gomokuai_amd.interface botzone` as the external bot program, the way the reference's Python side talks to its C++
bot (agents/botzone.py:27-41, core/interface/src/Interface.h:9-31); the keep-alive bot and a console match run in-process."""
import io
import json
import os
import sys

import pytest

from gomokuai_amd import agents, core, interface

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_botzone_agent_drives_the_bot_program():
    bot = agents.BotzoneAgent("%s -m gomokuai_amd.interface botzone --agent traditional:5 --iterations 200 --seed 3" % sys.executable, working_dir=ROOT)
    b = core.Board()
    for mv in (112, 113, 127):
        b.apply_move(core.Position(mv))
    move = bot.get_action(b)                                       # white to move
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
