"""Test-side helpers (not product code): a minimal agent loop over CorePyExt.MCTS objects, the eight board symmetries the
reference's augmentation produces, and a driver for a Botzone-protocol bot program.

What they check against is the reference's CONTRACT, read from its text:
  * training tuples of one game: (Board.encoded_states(), score of the player to move given the final winner, pi) per move,
    agents/utils.py:29-63;
  * augmentation order: for i in 0..3 the arrays rotated by 90 degrees i times, then that rotation mirrored left-right,
    network/data_helper.py:36-55;
  * Botzone: the whole history as {"requests": [...], "responses": [...]} on stdin, {"response": {"x": .., "y": ..}} on stdout,
    agents/botzone.py:27-41.
Committed fixtures made by the reference's own loop: tests/golden/reference_tuples.npz (tests/golden/make_reference_tuples.py)."""
import json
import shlex
import subprocess

import numpy as np

from gomokuai_amd import core


class Searcher:
    """A player that owns one CorePyExt.MCTS (tree kept from move to move, like the reference's MCTSAgent)."""

    def __init__(self, policy, c_iterations=None, c_duration=None):
        self.policy = policy
        kwargs = {"policy": policy}
        if c_iterations is not None:
            kwargs["c_iterations"] = c_iterations
        if c_duration is not None:
            kwargs["c_duration"] = c_duration
        self.kwargs = kwargs
        self.mcts = core.MCTS(**kwargs)

    def reset(self):
        self.mcts = core.MCTS(**self.kwargs)

    def move(self, board):
        self.mcts.sync_with_board(board)
        return self.mcts.get_action(board)

    def evaluate(self, board):
        """(value, pi [225], move the search would play)"""
        self.mcts.sync_with_board(board)
        value, pi = self.mcts.eval_state(board)
        self.mcts.step_forward()
        return value, pi, self.mcts.root.position


def random_searcher(c_puct=5.0, c_rollouts=5, **budget):
    return Searcher(core.RandomPolicy(c_puct, c_rollouts), **budget)


def pattern_searcher(c_puct=5.0, **budget):
    return Searcher(core.TraditionalPolicy(c_puct), **budget)


def rave_searcher(c_puct=2.0, c_bias=0.0, **budget):
    return Searcher(core.PoolRAVEPolicy(c_puct, c_bias), **budget)


def network_searcher(eval_state, c_puct=5.0, **budget):
    return Searcher(core.Policy(eval_state=eval_state, c_puct=c_puct), **budget)


def play_game(black, white, record=False):
    """One game between two Searchers.  record=False: the winner.  record=True: the training tuples of the game,
    [(uint8[6,15,15], float score for the player then to move, float32[225] pi)]."""
    board = core.Board()
    players = {core.Player.black: black, core.Player.white: white}
    seen = []
    while not board.status["is_end"]:
        who = board.status["cur_player"]
        if record:
            _, pi, mv = players[who].evaluate(board)
            seen.append((board.encoded_states(), who, np.asarray(pi, dtype=np.float32).reshape(-1)))
        else:
            mv = players[who].move(board)
        board.apply_move(mv)
    winner = board.status["winner"]
    if not record:
        return winner
    return [(states, np.array(core.Player.calc_score(who, winner)), pi) for states, who, pi in seen]


def symmetries(planes, pi):
    """The eight (planes [k,15,15], pi [225]) copies in the reference's order: rot90^i, then its left-right mirror."""
    out = []
    grid = np.asarray(pi).reshape(15, 15)
    for turns in range(4):
        p = np.rot90(planes, turns, axes=(1, 2))
        g = np.rot90(grid, turns)
        out.append((np.ascontiguousarray(p), np.ascontiguousarray(g).reshape(-1)))
        out.append((np.ascontiguousarray(p[:, :, ::-1]), np.ascontiguousarray(g[:, ::-1]).reshape(-1)))
    return out


def augment(tuples):
    return [(p, value, q) for planes, value, pi in tuples for p, q in symmetries(planes, pi)]


class BotDriver:
    """Runs a Botzone bot program once per move: the whole history goes in on stdin, one response comes back."""

    def __init__(self, command, cwd=None):
        self.command, self.cwd = shlex.split(command), cwd

    @staticmethod
    def history(board):
        """requests = the opponent's moves (first entry (-1,-1) when this side opens), responses = this side's own moves."""
        record = [(p.x, p.y) for p in board.move_record]
        mine_first = len(record) % 2 == 0                  # the side to move made the moves at even distance from the end
        requests, responses = [], []
        if mine_first:
            requests.append({"x": -1, "y": -1})
        for i, (x, y) in enumerate(record):
            own = (i % 2 == 0) == mine_first
            (responses if own else requests).append({"x": x, "y": y})
        return {"requests": requests, "responses": responses}

    def move(self, board):
        out = subprocess.run(self.command, input=json.dumps(self.history(board)), capture_output=True, text=True, cwd=self.cwd, timeout=600)
        if out.returncode != 0:
            raise RuntimeError("bot failed: " + out.stderr[-2000:])
        reply = json.loads(out.stdout.strip().splitlines()[-1])["response"]
        return core.Position(int(reply["x"]), int(reply["y"]))


def encoded_states_of(moves):
    """The six feature planes the reference hands the network for the position after `moves` (a list of cell ids, black first), restated
    here from core/py_ext/src/game_ext.hpp:87-104 with nothing of this repo's code in between: [stones of the player to move, stones of
    the opponent, empty cells, one-hot last move, one-hot move before last, all ones iff black is to move], uint8[6, 15, 15], a plane
    indexed [y][x] with cell id = 15 y + x.  (For positions of a game that is still running: the player to move alternates from black.)"""
    planes = np.zeros((6, 15, 15), dtype=np.uint8)
    black_to_move = len(moves) % 2 == 0
    stones = {True: np.zeros(225, dtype=np.uint8), False: np.zeros(225, dtype=np.uint8)}     # keyed by "is black"
    for i, mv in enumerate(moves):
        stones[i % 2 == 0][int(mv)] = 1
    planes[0] = stones[black_to_move].reshape(15, 15)
    planes[1] = stones[not black_to_move].reshape(15, 15)
    planes[2] = (1 - stones[True] - stones[False]).reshape(15, 15)
    for back in (0, 1):
        if len(moves) > back:
            planes[3 + back].reshape(-1)[int(moves[len(moves) - 1 - back])] = 1
    planes[5][:] = 1 if black_to_move else 0
    return planes


class PaddedNetwork:
    """A network behind a fixed batch size: the rows it is given, padded with empty positions to `rows`, so that its dense layers (library
    GEMMs, whose summation order may follow the batch size) see the same shapes whatever the number of live games.  For tests that want
    the games of two self-play loops to be equal bit for bit although one of them hands the network live games only."""

    def __init__(self, network, rows):
        self.network, self.rows = network, rows

    def __call__(self, states):
        import torch
        n = states.shape[0]
        if n == self.rows:
            return self.network(states)
        padded = torch.zeros((self.rows,) + tuple(states.shape[1:]), dtype=states.dtype, device=states.device)
        padded[:n] = states
        value, probs = self.network(padded)
        return value[:n], probs[:n]
