"""ctypes binding of the CPU oracle (oracle/libgomoku_oracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  Nothing under gomokuai_amd/ may import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libgomoku_oracle.so")

N = 225
MAX_PATTERNS = 512
MAX_DAT = 8192
WHITE, NONE, BLACK = -1, 0, 1
TYPE_NAMES = ["DeadOne", "LiveOne", "DeadTwo", "LiveTwo", "DeadThree", "LiveThree", "DeadFour", "LiveFour", "Five"]


def build(force=False):
    srcs = [os.path.join(_HERE, f) for f in ("go_board.c", "go_ac.c", "go_eval.c", "go_mcts.c", "go_scratch.c", "go_trad.c", "go_rave.c", "go_stdsort.cpp", "gomoku_oracle.h",
                                             "go_eval_internal.h", os.path.join("..", "include", "gomoku_noise.h"))]
    if force or not os.path.exists(_SO) or any(os.path.getmtime(s) > os.path.getmtime(_SO) for s in srcs if os.path.exists(s)):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _SO


class Board(C.Structure):
    _fields_ = [("cur_player", C.c_int8), ("winner", C.c_int8), ("states", (C.c_uint8 * N) * 3),
                ("counts", C.c_int32 * 3), ("record", C.c_int16 * N), ("nrec", C.c_int32)]


class Pattern(C.Structure):
    _fields_ = [("str", C.c_char * 8), ("len", C.c_int8), ("favour", C.c_int8), ("type", C.c_int8), ("score", C.c_int32)]


class AC(C.Structure):
    _fields_ = [("n_patterns", C.c_int), ("patterns", Pattern * MAX_PATTERNS), ("size", C.c_int),
                ("base", C.c_int32 * MAX_DAT), ("check", C.c_int32 * MAX_DAT), ("fail", C.c_int32 * MAX_DAT),
                ("invariants", C.c_int32 * 5), ("sort_ties", C.c_int)]


_lib = None


EVAL_FN = C.CFUNCTYPE(None, C.POINTER(C.c_uint8), C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_void_p)


def lib():
    global _lib
    if _lib is not None:
        return _lib
    build()
    L = C.CDLL(_SO)
    P = C.POINTER
    L.go_board_reset.argtypes = [P(Board)]
    L.go_board_apply.argtypes = [P(Board), C.c_int, C.c_int]
    L.go_board_revert.argtypes = [P(Board), C.c_int]
    L.go_board_check_move.argtypes = [P(Board), C.c_int]
    L.go_board_check_end.argtypes = [P(Board)]
    L.go_board_random_move.argtypes = [P(Board), C.c_uint]
    L.go_board_encoded_states.argtypes = [P(Board), C.c_void_p]
    L.go_board_replay_games.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    L.go_board_replay_games.restype = None
    L.go_encode_char.argtypes = [C.c_char]
    L.go_ac_build.argtypes = [P(AC), P(C.c_char_p), P(C.c_int), P(C.c_int), C.c_int]
    L.go_ac_build_default.argtypes = [P(AC)]
    L.go_default_ac.restype = P(AC)
    L.go_ac_augment.argtypes = [P(Pattern), C.c_int, C.c_int]
    L.go_ac_match.argtypes = [P(AC), C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int]
    L.go_ac_used_slots.argtypes = [P(AC)]
    L.go_eval_new.restype = C.c_void_p
    for f in ("go_eval_free", "go_eval_reset"):
        getattr(L, f).argtypes = [C.c_void_p]
        getattr(L, f).restype = None
    L.go_eval_apply.argtypes = [C.c_void_p, C.c_int, P(C.c_int)]
    L.go_eval_revert.argtypes = [C.c_void_p, C.c_int]
    L.go_eval_check_end.argtypes = [C.c_void_p]
    L.go_eval_board.argtypes = [C.c_void_p]
    L.go_eval_board.restype = P(Board)
    for f in ("go_eval_get_scores", "go_eval_get_density", "go_eval_get_pattern_dist", "go_eval_get_compound_dist"):
        getattr(L, f).argtypes = [C.c_void_p, C.c_void_p]
        getattr(L, f).restype = None
    L.go_eval_line_view.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    L.go_eval_replay_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.go_eval_replay_batch.restype = None
    L.go_scratch_eval_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.go_scratch_eval_batch.restype = None
    L.go_philox4x32.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    L.go_philox4x32.restype = None
    L.go_trad_new.argtypes = [C.c_double]
    L.go_trad_new.restype = C.c_void_p
    L.go_trad_free.argtypes = [C.c_void_p]
    L.go_trad_search.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_uint64]
    L.go_trad_search.restype = None
    L.go_trad_set_noise.argtypes = [C.c_void_p, C.c_float, C.c_float, C.c_uint64, C.c_uint32]
    L.go_trad_set_noise.restype = None
    L.go_trad_run.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_uint64]
    L.go_trad_run.restype = None
    L.go_trad_step_forward.argtypes = [C.c_void_p]
    L.go_trad_root_children.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.go_trad_root_visits.argtypes = [C.c_void_p]
    L.go_trad_root_visits.restype = C.c_uint64
    L.go_trad_root_value.argtypes = [C.c_void_p]
    L.go_trad_root_value.restype = C.c_float
    L.go_trad_n_nodes.argtypes = [C.c_void_p]
    L.go_trad_evaluator_updates.argtypes = [C.c_void_p]
    L.go_trad_evaluator_updates.restype = C.c_uint64
    L.go_trad_evaluator.argtypes = [C.c_void_p]
    L.go_trad_evaluator.restype = C.c_void_p
    L.go_trad_heuristic.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    L.go_trad_heuristic.restype = C.c_float
    L.go_rave_new.argtypes = [C.c_double, C.c_double, C.c_uint64, C.c_uint32]
    L.go_rave_new.restype = C.c_void_p
    L.go_rave_free.argtypes = [C.c_void_p]
    L.go_rave_set_noise.argtypes = [C.c_void_p, C.c_float, C.c_float]
    L.go_rave_set_noise.restype = None
    L.go_rave_run.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_uint64]
    L.go_rave_run.restype = None
    L.go_rave_step_forward.argtypes = [C.c_void_p]
    L.go_rave_root_children.argtypes = [C.c_void_p] + [C.c_void_p] * 5
    L.go_rave_root_visits.argtypes = [C.c_void_p]
    L.go_rave_root_visits.restype = C.c_uint64
    L.go_rave_root_value.argtypes = [C.c_void_p]
    L.go_rave_root_value.restype = C.c_float
    L.go_rave_size.argtypes = [C.c_void_p]
    L.go_rave_size.restype = C.c_uint64
    L.go_mcts_set_evaluator.argtypes = [C.c_void_p, EVAL_FN, C.c_void_p]
    L.go_mcts_set_evaluator.restype = None
    L.go_mcts_new.argtypes = [C.c_uint64, C.c_double, C.c_int, C.c_uint64, C.c_uint32]
    L.go_mcts_new.restype = C.c_void_p
    L.go_mcts_free.argtypes = [C.c_void_p]
    L.go_mcts_reset.argtypes = [C.c_void_p]
    L.go_mcts_sync_with_board.argtypes = [C.c_void_p, P(Board)]
    L.go_mcts_run_playouts.argtypes = [C.c_void_p, P(Board)]
    L.go_mcts_get_action.argtypes = [C.c_void_p, P(Board)]
    L.go_mcts_eval_state.argtypes = [C.c_void_p, P(Board), C.c_void_p, C.c_void_p]
    L.go_mcts_eval_state.restype = C.c_float
    L.go_mcts_step_forward.argtypes = [C.c_void_p]
    L.go_mcts_step_forward_move.argtypes = [C.c_void_p, C.c_int]
    L.go_mcts_size.argtypes = [C.c_void_p]
    L.go_mcts_size.restype = C.c_uint64
    L.go_mcts_root_position.argtypes = [C.c_void_p]
    L.go_mcts_root_player.argtypes = [C.c_void_p]
    L.go_mcts_root_visits.argtypes = [C.c_void_p]
    L.go_mcts_root_visits.restype = C.c_uint64
    L.go_mcts_root_value.argtypes = [C.c_void_p]
    L.go_mcts_root_value.restype = C.c_float
    L.go_mcts_root_children.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.go_mcts_root_children.restype = None
    L.go_mcts_alg_bytes.argtypes = [C.c_void_p]
    L.go_mcts_alg_bytes.restype = C.c_uint64
    L.go_mcts_set_noise.argtypes = [C.c_void_p, C.c_float, C.c_float]
    L.go_mcts_set_noise.restype = None
    L.go_mcts_set_noise_sampler.argtypes = [C.c_void_p, C.c_int]
    L.go_mcts_set_noise_sampler.restype = None
    L.go_trad_set_noise_sampler.argtypes = [C.c_void_p, C.c_int]
    L.go_trad_set_noise_sampler.restype = None
    L.go_noise_gamma.argtypes = [C.c_float, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint64]
    L.go_noise_gamma.restype = C.c_float
    L.go_noise_mix225.argtypes = [C.c_void_p, C.c_float, C.c_float, C.c_uint32, C.c_uint32, C.c_uint64]
    L.go_noise_mix225.restype = None
    L.go_noise_log.argtypes = [C.c_double]
    L.go_noise_log.restype = C.c_double
    L.go_noise_exp.argtypes = [C.c_double]
    L.go_noise_exp.restype = C.c_double
    L.go_visits_to_pi.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    L.go_visits_to_pi.restype = None
    _lib = L
    return L


def encode(s):
    """Mapping.h:40-48 applied to a python string -> uint8 codes."""
    L = lib()
    return np.array([L.go_encode_char(ch.encode()) for ch in s], dtype=np.uint8)


def default_ac():
    return lib().go_default_ac().contents


def build_ac(protos):
    """protos: list of (proto_str, type, score)."""
    ac = AC()
    n = len(protos)
    strs = (C.c_char_p * n)(*[p[0].encode() for p in protos])
    types = (C.c_int * n)(*[p[1] for p in protos])
    scores = (C.c_int * n)(*[p[2] for p in protos])
    rc = lib().go_ac_build(C.byref(ac), strs, types, scores, n)
    assert rc == 0
    return ac


def match(ac, codes):
    codes = np.ascontiguousarray(codes, dtype=np.uint8)
    pat = np.zeros(256, dtype=np.int32)
    off = np.zeros(256, dtype=np.int32)
    m = lib().go_ac_match(C.byref(ac), codes.ctypes.data, len(codes), pat.ctypes.data, off.ctypes.data, 256)
    return [(ac.patterns[int(pat[i])], int(off[i])) for i in range(m)]


class Evaluator:
    def __init__(self):
        self.L = lib()
        self.h = self.L.go_eval_new()

    def __del__(self):
        if getattr(self, "h", None):
            self.L.go_eval_free(self.h)
            self.h = None

    def reset(self):
        self.L.go_eval_reset(self.h)

    def apply(self, move):
        err = C.c_int(0)
        r = self.L.go_eval_apply(self.h, int(move), C.byref(err))
        return r, err.value

    def revert(self, count=1):
        return self.L.go_eval_revert(self.h, count)

    def check_end(self):
        return bool(self.L.go_eval_check_end(self.h))

    @property
    def board(self):
        return self.L.go_eval_board(self.h).contents

    def scores(self):
        out = np.zeros((4, N), dtype=np.int32)
        self.L.go_eval_get_scores(self.h, out.ctypes.data)
        return out

    def density(self):
        out = np.zeros((2, 2, N), dtype=np.int32)
        self.L.go_eval_get_density(self.h, out.ctypes.data)
        return out

    def pattern_dist(self):
        out = np.zeros((N + 1, 8), dtype=np.uint32)
        self.L.go_eval_get_pattern_dist(self.h, out.ctypes.data)
        return out

    def compound_dist(self):
        out = np.zeros((N + 1, 3), dtype=np.uint32)
        self.L.go_eval_get_compound_dist(self.h, out.ctypes.data)
        return out

    def line_view(self, pos, direction):
        out = np.zeros(13, dtype=np.uint8)
        self.L.go_eval_line_view(self.h, int(pos), int(direction), out.ctypes.data)
        return out


def replay_games(moves, lens):
    """moves u8[n][stride], lens i32[n] -> (legal bool[n], end_ply i32[n], winner i8[n]): every game through Board::applyMove with its
    victory check (oracle/go_board.c: go_board_replay_games)."""
    moves = np.ascontiguousarray(moves, dtype=np.uint8)
    lens = np.ascontiguousarray(lens, dtype=np.int32)
    n, stride = moves.shape
    legal = np.zeros(n, dtype=np.int8)
    end_ply = np.zeros(n, dtype=np.int32)
    winner = np.zeros(n, dtype=np.int8)
    lib().go_board_replay_games(moves.ctypes.data, lens.ctypes.data, stride, n, legal.ctypes.data, end_ply.ctypes.data, winner.ctypes.data)
    return legal.astype(bool), end_ply, winner


def replay_batch(moves, lens):
    """moves u8[n][stride], lens i32[n] -> (scores i32[n,4,225], density i32[n,2,2,225], totals u32[n,11], status i32[n])."""
    moves = np.ascontiguousarray(moves, dtype=np.uint8)
    lens = np.ascontiguousarray(lens, dtype=np.int32)
    n, stride = moves.shape
    scores = np.zeros((n, 4, N), dtype=np.int32)
    density = np.zeros((n, 2, 2, N), dtype=np.int32)
    totals = np.zeros((n, 11), dtype=np.uint32)
    status = np.zeros(n, dtype=np.int32)
    lib().go_eval_replay_batch(moves.ctypes.data, lens.ctypes.data, stride, n,
                               scores.ctypes.data, density.ctypes.data, totals.ctypes.data, status.ctypes.data)
    return scores, density, totals, status


def scratch_batch(moves, lens, lead=6, trail=6):
    """From-scratch formulation (go_scratch.c) on the final positions of the move lists."""
    moves = np.ascontiguousarray(moves, dtype=np.uint8)
    lens = np.ascontiguousarray(lens, dtype=np.int32)
    n, stride = moves.shape
    scores = np.zeros((n, 4, N), dtype=np.int32)
    density = np.zeros((n, 2, 2, N), dtype=np.int32)
    totals = np.zeros((n, 11), dtype=np.uint32)
    status = np.zeros(n, dtype=np.int32)
    lib().go_scratch_eval_batch(moves.ctypes.data, lens.ctypes.data, stride, n, lead, trail,
                                scores.ctypes.data, density.ctypes.data, totals.ctypes.data, status.ctypes.data)
    return scores, density, totals, status


def philox(ctr, key):
    c = np.array(ctr, dtype=np.uint32)
    k = np.array(key, dtype=np.uint32)
    o = np.zeros(4, dtype=np.uint32)
    lib().go_philox4x32(c.ctypes.data, k.ctypes.data, o.ctypes.data)
    return o


class MCTS:
    def __init__(self, c_iterations, c_puct=5.0, c_rollouts=5, seed=0x9E3779B97F4A7C15, game_id=0):
        self.L = lib()
        self.h = self.L.go_mcts_new(c_iterations, c_puct, c_rollouts, seed, game_id)

    def __del__(self):
        if getattr(self, "h", None):
            self.L.go_mcts_free(self.h)
            self.h = None

    def set_noise(self, alpha, epsilon, sampler=0):
        """sampler 0: std::gamma_distribution over std::mt19937 (host); 1: the counter-based sampler of include/gomoku_noise.h"""
        self.L.go_mcts_set_noise(self.h, alpha, epsilon)
        self.L.go_mcts_set_noise_sampler(self.h, int(sampler))

    def set_evaluator(self, fn):
        """fn(states uint8[6,15,15]) -> (value, probs float32[225]): Policy(eval_state=fn) of the reference (agents/alphazero.py:5-9)."""
        def thunk(states, value, probs, _user):
            v, p = fn(np.ctypeslib.as_array(states, shape=(6, 15, 15)))
            value[0] = float(np.float32(v))
            np.ctypeslib.as_array(probs, shape=(N,))[:] = np.asarray(p, dtype=np.float32)
        self._eval_cb = EVAL_FN(thunk)                 # keep the callback alive
        self.L.go_mcts_set_evaluator(self.h, self._eval_cb, None)

    def eval_state(self, board):
        probs = np.zeros(N, dtype=np.float32)
        visits = np.zeros(N, dtype=np.uint32)
        q = self.L.go_mcts_eval_state(self.h, C.byref(board), probs.ctypes.data, visits.ctypes.data)
        return q, probs, visits

    def get_action(self, board):
        return self.L.go_mcts_get_action(self.h, C.byref(board))

    def run_playouts(self, board):
        self.L.go_mcts_run_playouts(self.h, C.byref(board))

    def sync_with_board(self, board):
        self.L.go_mcts_sync_with_board(self.h, C.byref(board))

    def step_forward(self, move=None):
        if move is None:
            return self.L.go_mcts_step_forward(self.h)
        return self.L.go_mcts_step_forward_move(self.h, int(move))

    def reset(self):
        self.L.go_mcts_reset(self.h)

    @property
    def size(self):
        return self.L.go_mcts_size(self.h)

    @property
    def root_visits(self):
        return self.L.go_mcts_root_visits(self.h)

    @property
    def root_value(self):
        return self.L.go_mcts_root_value(self.h)

    @property
    def root_position(self):
        return self.L.go_mcts_root_position(self.h)

    @property
    def alg_bytes(self):
        return self.L.go_mcts_alg_bytes(self.h)

    def root_children(self):
        v = np.zeros(N, dtype=np.uint32)
        q = np.zeros(N, dtype=np.float32)
        p = np.zeros(N, dtype=np.float32)
        self.L.go_mcts_root_children(self.h, v.ctypes.data, q.ctypes.data, p.ctypes.data)
        return v, q, p


def new_board():
    b = Board()
    lib().go_board_reset(C.byref(b))
    return b


def visits_to_pi(visits, n_moves):
    v = np.ascontiguousarray(visits, dtype=np.uint32)
    pi = np.zeros(N, dtype=np.float32)
    lib().go_visits_to_pi(v.ctypes.data, int(n_moves), pi.ctypes.data)
    return pi


class TraditionalMCTS:
    """TraditionalPolicy search (go_trad.c): fresh root per search, the policy's evaluator persists across searches."""
    def __init__(self, c_puct=5.0):
        self.L = lib()
        self.h = self.L.go_trad_new(c_puct)

    def __del__(self):
        if getattr(self, "h", None):
            self.L.go_trad_free(self.h)
            self.h = None

    def search(self, moves, playouts):
        m = np.ascontiguousarray(moves, dtype=np.uint8)
        self.L.go_trad_search(self.h, m.ctypes.data, len(m), int(playouts))

    def set_noise(self, alpha, epsilon, seed, game_id=0, sampler=0):
        self.L.go_trad_set_noise(self.h, alpha, epsilon, seed, game_id)
        self.L.go_trad_set_noise_sampler(self.h, int(sampler))

    def run(self, moves, playouts):
        """runPlayouts on the kept tree (syncWithBoard, AddNoise, playouts)."""
        m = np.ascontiguousarray(moves, dtype=np.uint8)
        self.L.go_trad_run(self.h, m.ctypes.data, len(m), int(playouts))

    def step_forward(self):
        return self.L.go_trad_step_forward(self.h)

    def root_children(self):
        v = np.zeros(N, dtype=np.uint32)
        q = np.zeros(N, dtype=np.float32)
        p = np.zeros(N, dtype=np.float32)
        best = self.L.go_trad_root_children(self.h, v.ctypes.data, q.ctypes.data, p.ctypes.data)
        return v, q, p, best

    @property
    def root_visits(self):
        return self.L.go_trad_root_visits(self.h)

    @property
    def root_value(self):
        return self.L.go_trad_root_value(self.h)

    @property
    def n_nodes(self):
        return self.L.go_trad_n_nodes(self.h)

    @property
    def evaluator_updates(self):
        return self.L.go_trad_evaluator_updates(self.h)


class PoolRAVEMCTS:
    """PoolRAVEPolicy search on a kept tree (go_rave.c)."""
    def __init__(self, c_puct=2.0, c_bias=0.0, seed=0x9E3779B97F4A7C15, game_id=0):
        self.L = lib()
        self.h = self.L.go_rave_new(c_puct, c_bias, seed, game_id)

    def __del__(self):
        if getattr(self, "h", None):
            self.L.go_rave_free(self.h)
            self.h = None

    def set_noise(self, alpha, epsilon):
        self.L.go_rave_set_noise(self.h, alpha, epsilon)

    def run(self, moves, playouts):
        m = np.ascontiguousarray(moves, dtype=np.uint8)
        self.L.go_rave_run(self.h, m.ctypes.data, len(m), int(playouts))

    def step_forward(self):
        return self.L.go_rave_step_forward(self.h)

    def root_children(self):
        """(visits, values, priors, amaf_visits, amaf_values) by cell, and the move stepForward() would make."""
        v = np.zeros(N, dtype=np.uint32)
        q = np.zeros(N, dtype=np.float32)
        p = np.zeros(N, dtype=np.float32)
        av = np.zeros(N, dtype=np.uint32)
        aq = np.zeros(N, dtype=np.float32)
        best = self.L.go_rave_root_children(self.h, v.ctypes.data, q.ctypes.data, p.ctypes.data, av.ctypes.data, aq.ctypes.data)
        return v, q, p, av, aq, best

    @property
    def root_visits(self):
        return self.L.go_rave_root_visits(self.h)

    @property
    def root_value(self):
        return self.L.go_rave_root_value(self.h)

    @property
    def size(self):
        return self.L.go_rave_size(self.h)


def trad_heuristic(moves):
    """(probs after DecisiveFilter, value) of TraditionalPolicy::hybridSimulate at the position after `moves`."""
    m = np.ascontiguousarray(moves, dtype=np.uint8)
    probs = np.zeros(N, dtype=np.float32)
    value = lib().go_trad_heuristic(m.ctypes.data, len(m), probs.ctypes.data)
    return probs, value
