"""The reference's front ends (core/interface/src/Interface.h:9-129, Agent.h:19-170) on top of CorePyExt: a Botzone bot
(one JSON request per process, or the keep-alive protocol), and the console match between two agents that ends with the
game as Botzone-readable JSON.  The searches behind the agents run on the GPU (K3 / K6 / K8 through core.MCTS); this module
is host-side interop only.

    python -m gomokuai_amd.interface botzone   --agent traditional:5 --ms 960        < request.json
    python -m gomokuai_amd.interface keepalive --agent random:5:5    --iterations 2000
    python -m gomokuai_amd.interface console   --agent traditional:5 --agent2 traditional:7 --ms 1000
"""
import datetime
import json
import random
import sys

import numpy as np


def _core():
    from . import core
    return core


def _pos_json(p):
    return {"x": int(p.x), "y": int(p.y)}                          # to_json(Position) (Agent.h:15)


def _pos_from(core, j):
    return core.Position(int(j["x"]), int(j["y"]))                 # from_json (Agent.h:16)


# ---------------- agents (Agent.h:19-170) ----------------
class Agent:
    def name(self):
        raise NotImplementedError

    def get_action(self, board):
        raise NotImplementedError

    def debug_message(self):
        return None

    def sync_with_board(self, board):
        pass

    def reset(self):
        pass


class HumanAgent(Agent):
    """Agent.h:34-51: two hexadecimal coordinates from the input stream; -1 -1 asks the console to take two moves back."""

    def __init__(self, instream=None, outstream=None):
        self.instream, self.outstream = instream or sys.stdin, outstream or sys.stdout

    def name(self):
        return "HumanAgent"

    def get_action(self, board):
        core = _core()
        self.outstream.write("\nInput your move({-1 -1} to revert): ")
        self.outstream.flush()
        tokens = []
        while len(tokens) < 2:
            line = self.instream.readline()
            if not line:
                raise EOFError("HumanAgent: input stream ended")
            tokens += line.split()
        x, y = (int(t, 16) if not t.startswith("-") else -int(t[1:], 16) for t in tokens[:2])
        return core.Position(x, y)


class RandomAgent(Agent):
    """Agent.h:53-62: Board::getRandomMove."""

    def name(self):
        return "RandomAgent"

    def get_action(self, board):
        return board.random_move()


class MCTSAgent(Agent):
    """Agent.h:64-106: MCTS(duration, policy); the move is the argmax of evalState's probabilities (not stepForward's choice),
    the state value goes to stdout like the reference's `cout << state_value`."""

    def __init__(self, policy, milliseconds=None, iterations=None, quiet=False):
        self.policy, self.ms, self.iterations, self.quiet, self.mcts = policy, milliseconds, iterations, quiet, None

    def name(self):
        return "MCTSAgent:%s" % ("%dms" % self.ms if self.iterations is None else "%dit" % self.iterations)

    def sync_with_board(self, board):
        core = _core()
        if self.mcts is None:
            last = board.move_record[-1] if board.move_record else core.Position(-1)
            last_player = -board.status["cur_player"] if board.status["cur_player"] != core.Player.none else core.Player.white
            if self.iterations is not None:
                self.mcts = core.MCTS(c_iterations=int(self.iterations), last_move=last, last_player=last_player, policy=self.policy)
            else:
                self.mcts = core.MCTS(c_duration=datetime.timedelta(milliseconds=self.ms), last_move=last, last_player=last_player, policy=self.policy)
        else:
            self.mcts.sync_with_board(board)

    def get_action(self, board):
        value, probs = self.mcts.eval_state(board)
        if not self.quiet:
            print(value)
        return _core().Position(int(np.argmax(probs)))             # maxCoeff: the first maximum

    def debug_message(self):
        return {"iterations": int(self.mcts.iterations), "duration": "%dms" % int(self.mcts.duration.total_seconds() * 1000)}

    def reset(self):
        if self.mcts is not None:
            self.mcts.reset()


class PatternEvalAgent(Agent):
    """Agent.h:108-161: no search, the move with the largest Heuristic::EvaluationProbs after DecisiveFilter on the agent's own
    incremental evaluator (Heuristic.hpp:16-28, 94-161); the centre on an empty board.  On the GPU that is the policy head of K6:
    the root priors after one playout of a one-game handle, whose evaluator persists and is synchronised like the reference's.
    The debug message gives the pattern and compound counts before and after the move (K1 on the two positions)."""

    def __init__(self):
        self.tree, self.moves, self.this_move = None, [], None

    def name(self):
        return "PatternEvalAgent"

    def sync_with_board(self, board):
        self.moves = [int(p.id) for p in board.move_record]

    def get_action(self, board):
        from . import lib as G
        core = _core()
        if not self.moves:
            self.this_move = core.Position(7, 7)
        else:
            if self.tree is None:
                self.tree = G.TraditionalMCTS(1, node_capacity=1024)
            self.tree.set_positions([self.moves])
            self.tree.run(1)
            self.this_move = core.Position(int(np.argmax(self.tree.root_stats()["priors"][0])))      # maxCoeff: the first maximum
        return self.this_move

    def debug_message(self):
        from . import lib as G
        after = self.moves + [int(self.this_move.id)]
        moves = np.zeros((2, 64 if len(after) <= 64 else 225), np.uint8)
        moves[0, :len(self.moves)] = self.moves
        moves[1, :len(after)] = after
        planes = G.moves_to_planes(moves, np.array([len(self.moves), len(after)], np.int32))
        totals = G.eval_batch_host(planes)[2]
        def message(t):                                            # Record: white in the low, black in the high 16 bits (Pattern.cpp:390-393)
            return {name: [[int((t[i] >> sh) & 0xFFFF) for i in range(8)], [int((t[8 + i] >> sh) & 0xFFFF) for i in range(3)]]
                    for name, sh in (("black", 16), ("white", 0))}
        return {"before": message(totals[0]), "current": message(totals[1])}

    def reset(self):
        if self.tree is not None:
            self.tree.reset_evaluators()


def make_agent(spec, milliseconds=960, iterations=None, quiet=False):
    """'random', 'human', 'pattern', 'random-mcts[:c_puct[:c_rollouts]]', 'traditional[:c_puct]', 'poolrave[:c_puct[:c_bias]]'."""
    core = _core()
    kind, *args = spec.split(":")
    num = [float(a) for a in args]
    if kind == "random":
        return RandomAgent()
    if kind == "human":
        return HumanAgent()
    if kind == "pattern":
        return PatternEvalAgent()
    if kind == "random-mcts":
        policy = core.RandomPolicy(num[0] if num else 5.0, int(num[1]) if len(num) > 1 else 5)
    elif kind == "traditional":
        policy = core.TraditionalPolicy(num[0] if num else 5.0)
    elif kind == "poolrave":
        policy = core.PoolRAVEPolicy(num[0] if num else 2.0, num[1] if len(num) > 1 else 0.0)
    else:
        raise ValueError("unknown agent '%s'" % spec)
    return MCTSAgent(policy, milliseconds=milliseconds, iterations=iterations, quiet=quiet)


# ---------------- front ends (Interface.h:9-129) ----------------
def _restore(core, board, request):
    """requests has one entry more than responses; {"x": -1, "y": -1} (we play black) is rejected by applyMove: a no-op."""
    responses = request.get("responses", [])
    for i in range(len(responses)):
        board.apply_move(_pos_from(core, request["requests"][i]), False)
        board.apply_move(_pos_from(core, responses[i]), False)
    board.apply_move(_pos_from(core, request["requests"][len(responses)]), False)


def botzone_interface(agent, instream=None, outstream=None):
    """Interface.h:9-31: one request, one response."""
    core = _core()
    instream, outstream = instream or sys.stdin, outstream or sys.stdout
    board = core.Board()
    _restore(core, board, json.loads(instream.read()))
    agent.sync_with_board(board)
    out = {"response": _pos_json(agent.get_action(board)), "debug": agent.debug_message()}
    outstream.write(json.dumps(out) + "\n")
    outstream.flush()
    return 0


def keep_alive_botzone_interface(agent, instream=None, outstream=None, max_turns=None):
    """Interface.h:33-62: the first line restores the position, every later line is the opponent's move; each answer is
    followed by the keep-running marker.  Ends when the input ends (the reference loops until it is killed)."""
    core = _core()
    instream, outstream = instream or sys.stdin, outstream or sys.stdout
    board = core.Board()
    turn = 0
    while max_turns is None or turn < max_turns:
        line = instream.readline()
        if not line:
            break
        if not line.strip():
            continue
        request = json.loads(line)
        if turn == 0:
            _restore(core, board, request if "requests" in request else {"requests": [request], "responses": []})
        else:
            board.apply_move(_pos_from(core, request), False)
        agent.sync_with_board(board)
        board.apply_move(agent.get_action(board))
        out = {"response": _pos_json(board.move_record[-1]), "debug": agent.debug_message()}
        outstream.write(json.dumps(out) + "\n>>>BOTZONE_REQUEST_KEEP_RUNNING<<<\n")
        outstream.flush()
        turn += 1
    return 0


def board_text(board):
    states = board.move_states
    core = _core()
    black, white = states[core.Player.black], states[core.Player.white]
    rows = ["  " + " ".join("%X" % x for x in range(15))]
    for y in range(15):
        rows.append("%X " % y + " ".join("x" if black[y][x] else "o" if white[y][x] else "." for x in range(15)))
    return "\n".join(rows) + "\n"


def console_interface(agent0, agent1, outstream=None, black_player=None):
    """Interface.h:64-127: a match on the console; returns 1 for a tie, else 0, and prints the record from the winner's point of view
    as Botzone JSON ({-1,-1} first in requests when the winner played black)."""
    core = _core()
    out = outstream or sys.stdout
    board = core.Board()
    if black_player is None:
        black_player = random.randrange(2)
    agents = [agent0, agent1]
    index = {core.Player.white: 1 - black_player, core.Player.black: black_player}
    out.write("black: %d.%s\nwhite: %d.%s\n\n%s\n-------------------------\n" % (
        index[core.Player.black], agents[index[core.Player.black]].name(), index[core.Player.white], agents[index[core.Player.white]].name(), board_text(board)))
    cur = core.Player.black
    while True:
        i = index[cur]
        agent = agents[i]
        agent.sync_with_board(board)
        move = agent.get_action(board)
        if move.x == -1 and move.y == -1:
            board.revert_move(2)
            out.write(board_text(board))
            continue
        result = board.apply_move(move)
        if result == cur:
            out.write("Invalid move: %s\n" % str(move))
            continue
        out.write("\n%d.%s's move: %s:\n\n%s\nDebug Messages:%s\n-------------------------\n" % (i, agent.name(), str(move), board_text(board), json.dumps(agent.debug_message())))
        if result == core.Player.none:
            break
        cur = result
    winner = board.status["winner"]
    if winner != core.Player.none:
        out.write("\nGame end. Winner: %d.%s\nRecord JSON:\n" % (index[winner], agents[index[winner]].name()))
    else:
        out.write("\nTie.\n")
    records = {"requests": [], "responses": []}
    for i, p in enumerate(board.move_record):
        if i == 0 and winner == core.Player.black:
            records["requests"].append({"x": -1, "y": -1})
        if (i % 2 == 0) == (winner == core.Player.black):
            records["responses"].append(_pos_json(p))
        else:
            records["requests"].append(_pos_json(p))
    out.write(json.dumps(records) + "\n")
    out.flush()
    return int(winner == core.Player.none)


def main(argv=None):
    import argparse
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("mode", choices=["botzone", "keepalive", "console"])
    ap.add_argument("--agent", default="traditional:5")
    ap.add_argument("--agent2", default="traditional:7", help="console mode: the second agent")
    ap.add_argument("--ms", type=int, default=960, help="search time per move (the reference's default constraint)")
    ap.add_argument("--iterations", type=int, default=None, help="playouts per move instead of a time budget")
    ap.add_argument("--seed", type=int, default=None)
    args = ap.parse_args(argv)
    if args.seed is not None:
        _core().set_seed(args.seed)
        random.seed(args.seed)
    quiet = args.mode != "console"                                  # a bot's stdout carries the protocol only
    agent = make_agent(args.agent, args.ms, args.iterations, quiet)
    if args.mode == "botzone":
        return botzone_interface(agent)
    if args.mode == "keepalive":
        return keep_alive_botzone_interface(agent)
    return console_interface(agent, make_agent(args.agent2, args.ms, args.iterations, quiet))


if __name__ == "__main__":
    sys.exit(main())
