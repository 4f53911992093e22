"""The Botzone / console front ends (gomokuai_amd/interface.py; core/interface/src/Interface.h:9-129) with scripted agents:
protocol handling only, no search and no GPU."""
import io
import json

from gomokuai_amd import core, interface


class Scripted(interface.Agent):
    def __init__(self, moves, label="Scripted"):
        self.moves, self.label, self.seen = list(moves), label, []

    def name(self):
        return self.label

    def sync_with_board(self, board):
        self.seen.append([p.id for p in board.move_record])

    def get_action(self, board):
        return core.Position(self.moves.pop(0))

    def debug_message(self):
        return {"left": len(self.moves)}


def _xy(cell):
    return {"x": cell % 15, "y": cell // 15}


def test_botzone_single_request():
    # we are white: the opponent (black) opened at 112, we answered 113, black played 98
    req = {"requests": [_xy(112), _xy(98)], "responses": [_xy(113)]}
    agent, out = Scripted([127]), io.StringIO()
    assert interface.botzone_interface(agent, io.StringIO(json.dumps(req)), out) == 0
    assert agent.seen == [[112, 113, 98]]
    assert json.loads(out.getvalue()) == {"response": _xy(127), "debug": {"left": 0}}
    # we are black: the first request is {-1, -1}, which the board rejects
    agent, out = Scripted([112]), io.StringIO()
    interface.botzone_interface(agent, io.StringIO(json.dumps({"requests": [{"x": -1, "y": -1}], "responses": []})), out)
    assert agent.seen == [[]] and json.loads(out.getvalue())["response"] == _xy(112)


def test_keep_alive_protocol():
    lines = [json.dumps({"requests": [_xy(112)], "responses": []}), json.dumps(_xy(98)), json.dumps(_xy(99))]
    agent, out = Scripted([113, 114, 115]), io.StringIO()
    interface.keep_alive_botzone_interface(agent, io.StringIO("\n".join(lines) + "\n"), out)
    assert agent.seen == [[112], [112, 113, 98], [112, 113, 98, 114, 99]]
    answers = out.getvalue().split(">>>BOTZONE_REQUEST_KEEP_RUNNING<<<\n")
    assert [json.loads(a)["response"] for a in answers[:3]] == [_xy(113), _xy(114), _xy(115)] and answers[3] == ""


def test_console_match_and_record():
    c = lambda y, x: y * 15 + x
    black = Scripted([c(7, 3), c(7, 4), c(7, 4), c(7, 5), c(7, 6), c(7, 7)], "B")      # the second (7,4) is an invalid move
    white = Scripted([c(0, 0), c(0, 2), c(0, 4), c(0, 6)], "W")
    out = io.StringIO()
    assert interface.console_interface(black, white, out, black_player=0) == 0
    text = out.getvalue()
    assert "Invalid move: (4, 7)" in text and "Game end. Winner: 0.B" in text
    record = json.loads(text.strip().splitlines()[-1])
    # from the winner's (black's) point of view: requests start with {-1,-1} and hold white's moves
    assert record["requests"] == [{"x": -1, "y": -1}] + [_xy(m) for m in (c(0, 0), c(0, 2), c(0, 4), c(0, 6))]
    assert record["responses"] == [_xy(m) for m in (c(7, 3), c(7, 4), c(7, 5), c(7, 6), c(7, 7))]
    # a record in that form restores to the same game through the bot front end
    restored = Scripted([0])
    interface.botzone_interface(restored, io.StringIO(json.dumps({"requests": record["requests"][:5], "responses": record["responses"][:4]})), io.StringIO())
    assert restored.seen == [[c(7, 3), c(0, 0), c(7, 4), c(0, 2), c(7, 5), c(0, 4), c(7, 6), c(0, 6)]]


def test_human_agent_and_take_back():
    c = lambda y, x: y * 15 + x
    typed = io.StringIO("7 7\n-1 -1\na 7\n")                      # black plays (7,7) ... later takes two moves back, then (10,7)
    human = interface.HumanAgent(typed, io.StringIO())
    b = core.Board()
    first = human.get_action(b)
    assert (first.x, first.y) == (7, 7)
    back = human.get_action(b)
    assert (back.x, back.y) == (-1, -1)
    third = human.get_action(b)
    assert (third.x, third.y) == (10, 7)
    assert interface.RandomAgent().name() == "RandomAgent" and 0 <= interface.RandomAgent().get_action(b).id < 225
    assert "x" not in interface.board_text(b) and interface.board_text(b).count(".") == 225
