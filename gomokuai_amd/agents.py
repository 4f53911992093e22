"""Host-side mirror of the reference's agent interface for the hot path: agents/agent.py:7-33,
agents/mcts.py:5-34 and agents/utils.py:5-63 (same names, arguments and return shapes), on top of the
MI355X CorePyExt.  Written for this repo; the reference's own files run unchanged on the same module
(see tests/test_pyext.py and INTEGRATION.md)."""
import json
import os
from subprocess import PIPE, Popen

import numpy as np

from .core import Board, GameConfig as Game, MCTS, Player, Policy, PoolRAVEPolicy, Position, RandomPolicy, TraditionalPolicy


class Agent:
    """Base class, also the random agent (agents/agent.py:7-33)."""

    def get_action(self, state):
        return state.random_move()

    def eval_state(self, state):
        return 0, np.full((Game["width"], Game["height"]), 1 / Game["board_size"]), self.get_action(state)

    def reset(self):
        pass

    def __repr__(self):
        return "Base Random Agent"


RandomAgent = Agent


class MCTSAgent(Agent):
    """agents/mcts.py:5-27: c_iterations=... or c_duration=... as the constraint."""

    def __init__(self, policy=None, **constraint):
        self.mcts = MCTS(policy=policy, **constraint)

    def get_action(self, state):
        self.mcts.sync_with_board(state)
        return self.mcts.get_action(state)

    def eval_state(self, state):
        self.mcts.sync_with_board(state)
        Q, pi = self.mcts.eval_state(state)
        self.mcts.step_forward()
        return Q, pi, self.mcts.root.position

    def reset(self):
        self.mcts.reset()

    def __repr__(self):
        return "MCTS Agent with {}".format(self.mcts.policy.__class__.__name__)


def RandomMCTSAgent(c_puct, c_rollouts=5, **constraint):
    return MCTSAgent(policy=RandomPolicy(c_puct, c_rollouts), **constraint)


def RAVEAgent(c_puct, c_bias, **constraint):
    """agents/mcts.py:36-40 of the reference: MCTS with PoolRAVEPolicy (the search runs on the GPU, K8)."""
    return MCTSAgent(policy=PoolRAVEPolicy(c_puct, c_bias), **constraint)


def TraditionalAgent(c_puct, c_bias=0.0, use_rave=False, **constraint):
    """agents/mcts.py:44-48 of the reference: the pattern-guided searcher ("traditional_mcts" in config.py:9-12)."""
    return MCTSAgent(policy=TraditionalPolicy(c_puct, c_bias, use_rave), **constraint)


def PyConvNetAgent(network, c_puct, **constraint):
    """agents/alphazero.py:5-9: MCTS guided by network.eval_state(board) -> (value, probs[225]).  The tree search runs on the
    GPU (K7), the network is called once per playout like in the reference; lib.AlphaZeroMCTS is the batched form."""
    return MCTSAgent(policy=Policy(eval_state=network.eval_state, c_puct=c_puct), **constraint)


def dual_play(agents, board=None, verbose=False):
    """agents/utils.py:5-63: {Player.black: a1, Player.white: a2} -> winner, or the training tuples
    [(uint8[6,15,15] states, float score, float32[225] probs)] when verbose."""
    if board is None:
        board = Board()
    elif board.status["is_end"]:
        board.reset()
    result = [] if verbose else Player.none
    while True:
        cur_agent = agents[board.status["cur_player"]]
        if verbose:
            _, action_probs, next_move = cur_agent.eval_state(board)
            result.append([board.encoded_states(), board.status["cur_player"], action_probs])
        else:
            next_move = cur_agent.get_action(board)
        board.apply_move(next_move)
        if board.status["is_end"]:
            winner = board.status["winner"]
            if verbose:
                return [(s[0], np.array(Player.calc_score(s[1], winner)), s[2]) for s in result]
            return winner


def eval_agents(agents, num_games=9, verbose=False):
    """agents/utils.py:66-103: win rates of two agents over num_games games, colours swapped every game, a tie counts half."""
    board = Board()
    players = [Player.black, Player.white]
    win_cnts = np.zeros(2)
    for i in range(num_games):
        winner = dual_play(dict(zip(players, agents)), board)
        if winner in players:
            win_cnts[players.index(winner)] += 1
        else:
            win_cnts += 0.5
        if verbose:
            print("Round {} ends, winner is {};".format(i + 1, winner))
        players.reverse()
        board.reset()
        for agent in agents:
            agent.reset()
    return tuple(win_cnts / num_games)


class BotzoneAgent(Agent):
    """agents/botzone.py:11-45: an external Botzone-protocol program as an agent.  The position goes to the program's
    stdin as {"requests": [...], "responses": [...]} (the opponent's moves and its own, {"x": -1, "y": -1} first when it
    plays black), its stdout answers {"response": {"x": .., "y": ..}}."""

    def __init__(self, program, keep_alive=False, working_dir="."):
        self.program = program
        self.working_dir = os.path.realpath(working_dir)
        self.keep_alive = keep_alive

    def get_action(self, state):
        return self._communicate(state)

    def eval_state(self, state):
        action = self._communicate(state)
        action_probs = np.zeros(Game["board_size"])
        action_probs[int(action)] = 1.0
        return 0.0, action_probs, action

    @staticmethod
    def _parse_state(state):
        record = [{"x": p.x, "y": p.y} for p in state.move_record]
        offset = len(record) % 2
        padding = [{"x": -1, "y": -1}] * (1 - offset)
        return json.dumps({"requests": padding + record[1 - offset::2], "responses": record[offset::2]})

    def _communicate(self, state):
        bot = Popen(self.program, shell=True, stdin=PIPE, cwd=self.working_dir, stdout=PIPE, universal_newlines=True)
        output, _ = bot.communicate(self._parse_state(state))
        bot.terminate()
        response = json.loads(output)["response"]
        return Position(response["x"], response["y"])

    def __repr__(self):
        return "Botzone Agent at <{}/{}>".format(self.working_dir, self.program)


def augment_game_data(data):
    """network/data_helper.py:36-55: 4 rotations x {identity, fliplr} of every (states, value, probs) sample."""
    out = []
    for states, value, probs in data:
        for i in range(4):
            rot_states = np.array([np.rot90(s, i) for s in states])
            rot_probs = np.rot90(probs.reshape(Game["height"], Game["width"]), i)
            out.append((rot_states, value, rot_probs.flatten()))
            out.append((np.array([np.fliplr(s) for s in rot_states]), value, np.fliplr(rot_probs).flatten()))
    return out
