"""ctypes binding of libgomoku_hip.so (include/gomoku_hip.h)."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libgomoku_hip.so")
# tools/*.sh ask for the profiling flavour of the library explicitly (GMK_HIP_LIB=prof; built by `python -m gomokuai_amd.build
# --profile`): only that one reads GMK_*_PHASE_MASK / GMK_*_PROFILE.  Anything else loads the production library, which ignores them.
if os.environ.get("GMK_HIP_LIB") == "prof":
    _SO = os.path.join(_HERE, "libgomoku_hip_prof.so")
elif os.environ.get("GMK_HIP_LIB", "").endswith(".so"):          # an experiment build of tools/k1_variants.py, by path
    _SO = os.environ["GMK_HIP_LIB"]

N = 225


class GmkError(RuntimeError):
    pass


class TableInfo(C.Structure):
    _fields_ = [("n_patterns", C.c_int32), ("n_states", C.c_int32), ("dat_size", C.c_int32),
                ("max_emissions", C.c_int32), ("trans_words", C.c_int32), ("emit_words", C.c_int32),
                ("invariants", C.c_int32 * 5)]


_lib = None

# gmk_*_set_option (include/gomoku_hip.h)
OPT_NOISE_SAMPLER, OPT_LOCKSTEP = 1, 2
NOISE_SAMPLERS = {"std": 0, "counter": 1}      # std::gamma_distribution on the host / the counter-based sampler of include/gomoku_noise.h on the device

# every symbol include/gomoku_hip.h declares (checked by tests/test_cabi.py)
EXPORTS = [
    "gmk_init", "gmk_shutdown", "gmk_pool_release", "gmk_pool_poison", "gmk_last_error", "gmk_device_info",
    "gmk_tables_info", "gmk_tables_pattern", "gmk_tables_copy", "gmk_tables_copy_dat", "gmk_tables_scan",
    "gmk_synth_boards", "gmk_moves_to_planes",
    "gmk_eval_batch", "gmk_eval_batch_host", "gmk_eval_launch_info",
    "gmk_mcts_create", "gmk_mcts_destroy", "gmk_mcts_set_roots", "gmk_mcts_set_game_ids", "gmk_mcts_run", "gmk_mcts_root_stats",
    "gmk_mcts_alg_bytes", "gmk_mcts_launch_info", "gmk_visits_to_pi", "gmk_mcts_advance", "gmk_mcts_step", "gmk_mcts_step_host", "gmk_mcts_add_root_noise", "gmk_mcts_set_option", "gmk_mcts_reserve", "gmk_selfplay_run", "gmk_samples_from_records",
    "gmk_evalstate_create", "gmk_evalstate_destroy", "gmk_evalstate_reset", "gmk_evalstate_update", "gmk_evalstate_update_host", "gmk_evalstate_read",
    "gmk_az_create", "gmk_az_destroy", "gmk_az_set_roots", "gmk_az_select", "gmk_az_expand", "gmk_az_select_host", "gmk_az_expand_host", "gmk_az_read_node_host", "gmk_az_read_children_host", "gmk_az_set_leaf_host", "gmk_az_rollout_host", "gmk_az_expand_stages_host", "gmk_az_write_stats_host", "gmk_az_step", "gmk_az_advance", "gmk_az_set_slots", "gmk_az_live_games", "gmk_az_set_game_ids", "gmk_az_add_root_noise", "gmk_az_set_option", "gmk_az_root_stats",
    "gmk_trad_create", "gmk_trad_destroy", "gmk_trad_reset_evaluators", "gmk_trad_set_game_ids", "gmk_trad_set_positions", "gmk_trad_run", "gmk_trad_step", "gmk_trad_add_root_noise", "gmk_trad_set_option", "gmk_trad_reserve", "gmk_trad_root_stats", "gmk_trad_read_evaluators", "gmk_trad_run_poolrave", "gmk_trad_root_amaf", "gmk_trad_selfplay_run", "gmk_pvnet_create", "gmk_pvnet_destroy", "gmk_pvnet_forward", "gmk_pvnet_set_dense", "gmk_pvnet_evaluate",
]


def _torch_first():
    """PyTorch-ROCm bundles its own copy of the HIP runtime (torch/lib/libamdhip64.so) next to the system one this
    library links (libamdhip64.so.7).  Both can serve one process, but only when torch's copy brings the GPU up
    first; the other order leaves torch with "No HIP GPUs are available".  So: initialise torch.cuda before the
    first HIP call of this library whenever torch is importable (it provides device memory and streams here)."""
    try:
        import torch
    except ImportError:
        return
    if torch.cuda.is_available():
        torch.cuda.init()


def load():
    """Loads the HIP library; raises if it has not been built (no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    _torch_first()
    if not os.path.exists(_SO):
        raise GmkError("libgomoku_hip.so is missing: run `python -m gomokuai_amd.build` "
                       "(there is no CPU fallback for the compute path)")
    L = C.CDLL(_SO)
    vp = C.c_void_p
    L.gmk_last_error.restype = C.c_char_p
    L.gmk_init.argtypes = [C.c_int]
    L.gmk_device_info.argtypes = [C.POINTER(C.c_int), C.POINTER(C.c_size_t), C.c_char_p, C.c_int]
    L.gmk_tables_info.argtypes = [C.POINTER(TableInfo)]
    L.gmk_tables_pattern.argtypes = [C.c_int, C.c_char_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.gmk_tables_copy.argtypes = [vp, vp, vp]
    L.gmk_tables_copy_dat.argtypes = [vp, vp, vp]
    L.gmk_tables_scan.argtypes = [vp, C.c_int, vp, vp, C.c_int]
    L.gmk_synth_boards.argtypes = [C.c_uint64, C.c_uint32, C.c_int, C.c_int, vp, C.c_int, vp, vp]
    L.gmk_moves_to_planes.argtypes = [vp, C.c_int, vp, C.c_int, vp]
    L.gmk_eval_batch.argtypes = [vp, C.c_int, vp, vp, vp, vp, vp]
    L.gmk_eval_batch_host.argtypes = [vp, C.c_int, vp, vp, vp, vp]
    L.gmk_eval_launch_info.argtypes = [C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.gmk_mcts_create.argtypes = [C.c_int, C.c_int, C.c_double, C.c_int, C.c_uint64, C.POINTER(vp)]
    L.gmk_mcts_destroy.argtypes = [vp]
    L.gmk_mcts_set_roots.argtypes = [vp, vp, vp, C.c_uint32]
    L.gmk_mcts_set_game_ids.argtypes = [vp, vp]
    L.gmk_trad_set_game_ids.argtypes = [vp, vp]
    L.gmk_az_set_game_ids.argtypes = [vp, vp]
    L.gmk_mcts_run.argtypes = [vp, C.c_int, vp]
    L.gmk_mcts_root_stats.argtypes = [vp, vp, vp, vp, vp, vp]
    L.gmk_mcts_alg_bytes.argtypes = [vp, C.POINTER(C.c_uint64)]
    L.gmk_mcts_launch_info.argtypes = [vp, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.gmk_visits_to_pi.argtypes = [vp, C.c_int, vp]
    L.gmk_mcts_advance.argtypes = [vp, vp, vp, vp, vp, vp, C.c_int, vp]
    L.gmk_mcts_step.argtypes = [vp, vp, vp, vp, vp, vp, vp, C.c_int, vp]
    L.gmk_mcts_step_host.argtypes = [vp, vp, C.c_int]
    L.gmk_selfplay_run.argtypes = [vp, C.c_int, C.c_uint32, C.c_int, C.c_int, C.c_float, C.c_float, vp, C.c_int, vp, vp, vp, vp, vp, C.POINTER(C.c_int32), vp]
    L.gmk_mcts_add_root_noise.argtypes = [vp, C.c_float, C.c_float, vp]
    L.gmk_mcts_set_option.argtypes = [vp, C.c_int, C.c_int]
    L.gmk_mcts_reserve.argtypes = [vp, C.c_int]
    L.gmk_trad_reserve.argtypes = [vp, C.c_int]
    L.gmk_evalstate_create.argtypes = [C.c_int, C.POINTER(vp)]
    L.gmk_evalstate_destroy.argtypes = [vp]
    L.gmk_evalstate_reset.argtypes = [vp]
    L.gmk_evalstate_update.argtypes = [vp, vp, C.c_int, vp]
    L.gmk_evalstate_update_host.argtypes = [vp, vp, C.c_int]
    L.gmk_evalstate_read.argtypes = [vp, vp, vp, vp, vp, vp, vp]
    L.gmk_az_create.argtypes = [C.c_int, C.c_int, C.c_double, C.POINTER(vp)]
    L.gmk_az_destroy.argtypes = [vp]
    L.gmk_az_set_roots.argtypes = [vp, vp, vp]
    L.gmk_az_select.argtypes = [vp, vp, vp]
    L.gmk_az_expand.argtypes = [vp, vp, vp, vp]
    L.gmk_az_select_host.argtypes = [vp, vp, vp]
    L.gmk_az_step.argtypes = [vp, vp]
    L.gmk_az_live_games.argtypes = [vp, C.POINTER(C.c_int32)]
    L.gmk_az_set_slots.argtypes = [vp, C.c_int, vp, C.c_int, vp]
    L.gmk_az_advance.argtypes = [vp, vp, vp, vp, vp, C.c_int, C.POINTER(C.c_int32), vp]
    L.gmk_az_add_root_noise.argtypes = [vp, C.c_float, C.c_float, C.c_uint64, C.c_uint32]
    L.gmk_az_set_option.argtypes = [vp, C.c_int, C.c_int]
    L.gmk_az_expand_host.argtypes = [vp, vp, vp]
    L.gmk_az_root_stats.argtypes = [vp] * 8
    L.gmk_trad_create.argtypes = [C.c_int, C.c_int, C.POINTER(vp)]
    L.gmk_trad_destroy.argtypes = [vp]
    L.gmk_trad_reset_evaluators.argtypes = [vp]
    L.gmk_trad_set_positions.argtypes = [vp, vp, vp]
    L.gmk_trad_run.argtypes = [vp, C.c_int, C.c_double, vp]
    L.gmk_trad_step.argtypes = [vp, vp]
    L.gmk_trad_add_root_noise.argtypes = [vp, C.c_float, C.c_float, C.c_uint64, C.c_uint32]
    L.gmk_trad_set_option.argtypes = [vp, C.c_int, C.c_int]
    L.gmk_trad_root_stats.argtypes = [vp] * 10
    L.gmk_trad_read_evaluators.argtypes = [vp, vp, vp, vp, vp, vp, vp]
    L.gmk_trad_run_poolrave.argtypes = [vp, C.c_int, C.c_double, C.c_uint64, C.c_uint32, vp]
    L.gmk_trad_selfplay_run.argtypes = [vp, C.c_int, C.c_int, C.c_uint32, C.c_int, C.c_double, C.c_uint64, C.c_int, C.c_float, C.c_float,
                                        vp, C.c_int, vp, vp, vp, vp, vp, C.c_int, C.c_int, C.POINTER(C.c_int32), C.POINTER(C.c_int32), vp]
    L.gmk_trad_root_amaf.argtypes = [vp, vp, vp]
    L.gmk_pvnet_create.argtypes = [vp] * 10 + [C.POINTER(vp)]
    L.gmk_pvnet_destroy.argtypes = [vp]
    L.gmk_pvnet_forward.argtypes = [vp, vp, C.c_int, vp, vp, vp]
    L.gmk_pvnet_set_dense.argtypes = [vp] * 6 + [C.c_float]
    L.gmk_pvnet_evaluate.argtypes = [vp, vp, C.c_int, vp, vp, vp]
    L.gmk_samples_from_records.argtypes = [vp, vp, vp, vp, vp, vp, C.c_int, C.c_int, vp, vp, vp, vp]
    _lib = L
    return L


def _check(rc):
    if rc < 0:
        raise GmkError("libgomoku_hip: %s (status %d)" % (load().gmk_last_error().decode(), rc))
    return rc


def init(device=0):
    _check(load().gmk_init(device))


def pool_poison(on=True):
    """Diagnostic: reused device blocks of the library's pool are filled with 0xA5 before the next handle gets them (gmk_pool_poison)."""
    _check(load().gmk_pool_poison(1 if on else 0))


def release_pool():
    """Returns the device blocks the library keeps from destroyed handles (up to 224 GB of tree arenas) to the driver."""
    _check(load().gmk_pool_release())


def device_info():
    cu = C.c_int()
    mem = C.c_size_t()
    name = C.create_string_buffer(256)
    _check(load().gmk_device_info(C.byref(cu), C.byref(mem), name, 256))
    return {"cu_count": cu.value, "hbm_bytes": mem.value, "name": name.value.decode()}


# ---------------- pattern tables (host) ----------------
def tables_info():
    info = TableInfo()
    _check(load().gmk_tables_info(C.byref(info)))
    return info


def tables_pattern(i):
    s = C.create_string_buffer(8)
    fav, typ, score = C.c_int(), C.c_int(), C.c_int()
    _check(load().gmk_tables_pattern(i, s, C.byref(fav), C.byref(typ), C.byref(score)))
    return s.value.decode(), fav.value, typ.value, score.value


def tables_copy():
    info = tables_info()
    trans = np.zeros(info.trans_words, dtype=np.uint32)
    emit = np.zeros(info.emit_words, dtype=np.uint16)
    pinfo = np.zeros(info.n_patterns * 2, dtype=np.uint32)
    _check(load().gmk_tables_copy(trans.ctypes.data, emit.ctypes.data, pinfo.ctypes.data))
    return trans, emit, pinfo


def tables_copy_dat():
    info = tables_info()
    base = np.zeros(info.dat_size, dtype=np.int32)
    check = np.zeros(info.dat_size, dtype=np.int32)
    fail = np.zeros(info.dat_size, dtype=np.int32)
    _check(load().gmk_tables_copy_dat(base.ctypes.data, check.ctypes.data, fail.ctypes.data))
    return base, check, fail


def tables_scan(codes):
    codes = np.ascontiguousarray(codes, dtype=np.uint8)
    pats = np.zeros(512, dtype=np.int32)
    offs = np.zeros(512, dtype=np.int32)
    m = _check(load().gmk_tables_scan(codes.ctypes.data, len(codes), pats.ctypes.data, offs.ctypes.data, 512))
    return list(zip(pats[:m].tolist(), offs[:m].tolist()))


# ---------------- synthetic workloads (host) ----------------
DEFAULT_SEED = 0x9E3779B97F4A7C15


def synth_boards(n, kind=0, seed=DEFAULT_SEED, first_board=0, stride=64):
    """-> (moves u8[n,stride], lens i32[n], planes u16[n,2,16])."""
    moves = np.zeros((n, stride), dtype=np.uint8)
    lens = np.zeros(n, dtype=np.int32)
    planes = np.zeros((n, 2, 16), dtype=np.uint16)
    _check(load().gmk_synth_boards(seed, first_board, n, kind, moves.ctypes.data, stride, lens.ctypes.data, planes.ctypes.data))
    return moves, lens, planes


def moves_to_planes(moves, lens):
    moves = np.ascontiguousarray(moves, dtype=np.uint8)
    lens = np.ascontiguousarray(lens, dtype=np.int32)
    n, stride = moves.shape
    planes = np.zeros((n, 2, 16), dtype=np.uint16)
    _check(load().gmk_moves_to_planes(moves.ctypes.data, stride, lens.ctypes.data, n, planes.ctypes.data))
    return planes


# ---------------- K1: batched position evaluation ----------------
def eval_batch_host(planes):
    """planes u16[n,2,16] (host) -> scores i32[n,4,225], density i32[n,2,2,225], totals u32[n,11], status i32[n].
    Runs on the GPU through gmk_eval_batch_host; raises without one."""
    init()
    planes = np.ascontiguousarray(planes, dtype=np.uint16)
    n = planes.shape[0]
    scores = np.zeros((n, 4, N), dtype=np.int32)
    density = np.zeros((n, 2, 2, N), dtype=np.int32)
    totals = np.zeros((n, 11), dtype=np.uint32)
    status = np.zeros(n, dtype=np.int32)
    _check(load().gmk_eval_batch_host(planes.ctypes.data, n, scores.ctypes.data, density.ctypes.data,
                                      totals.ctypes.data, status.ctypes.data))
    return scores, density, totals, status


def eval_batch(d_planes, n, d_scores=None, d_density=None, d_totals=None, d_status=None, stream=None):
    """Device-pointer form (ints, e.g. torch.Tensor.data_ptr()); asynchronous on `stream`."""
    _check(load().gmk_eval_batch(d_planes, n, d_scores, d_density, d_totals, d_status, stream))


def eval_launch_info(n):
    g, b, l = C.c_int(), C.c_int(), C.c_int()
    _check(load().gmk_eval_launch_info(n, C.byref(g), C.byref(b), C.byref(l)))
    return {"grid": g.value, "block": b.value, "lds_bytes": l.value}


# ---------------- K3: batched MCTS (RandomPolicy) ----------------
class BatchedMCTS:
    """n_games independent searches on the GPU (gmk_mcts_*).  Fresh roots per set_roots()."""

    def __init__(self, n_games, playouts_capacity=800, c_puct=5.0, c_rollouts=5, seed=DEFAULT_SEED, node_capacity=None):
        init()
        self.n = n_games
        self.cap = node_capacity if node_capacity is not None else playouts_capacity * 225 + 1
        h = C.c_void_p()
        _check(load().gmk_mcts_create(n_games, self.cap, c_puct, c_rollouts, seed, C.byref(h)))
        self.h = h

    def close(self):
        if getattr(self, "h", None) and load is not None:
            load().gmk_mcts_destroy(self.h)
            self.h = None

    __del__ = close

    def set_roots(self, planes, last_moves, first_game_id=0):
        planes = np.ascontiguousarray(planes, dtype=np.uint16)
        last = np.ascontiguousarray(last_moves, dtype=np.int16)
        assert planes.shape == (self.n, 2, 16) and last.shape == (self.n,)
        _check(load().gmk_mcts_set_roots(self.h, planes.ctypes.data, last.ctypes.data, first_game_id))

    STATUS_TERMINAL, STATUS_ARENA_FULL, STATUS_ILLEGAL_STEP = 1, 2, 4      # bits of root_stats()[4]

    def set_game_ids(self, ids):
        """Global id of every game (uint32[n]) instead of first_game_id + g: the id keys the game's random streams."""
        ids = np.ascontiguousarray(ids, dtype=np.uint32)
        assert ids.shape == (self.n,)
        _check(load().gmk_mcts_set_game_ids(self.h, ids.ctypes.data))

    def run(self, playouts, stream=None):
        _check(load().gmk_mcts_run(self.h, playouts, stream))

    def root_stats(self):
        visits = np.zeros((self.n, N), dtype=np.uint32)
        q = np.zeros(self.n, dtype=np.float32)
        rv = np.zeros(self.n, dtype=np.uint32)
        nodes = np.zeros(self.n, dtype=np.uint32)
        status = np.zeros(self.n, dtype=np.int32)
        _check(load().gmk_mcts_root_stats(self.h, visits.ctypes.data, q.ctypes.data, rv.ctypes.data, nodes.ctypes.data, status.ctypes.data))
        return visits, q, rv, nodes, status

    def advance(self, d_moves, d_visits, d_lens, d_winner, d_unfinished, reuse_subtree=False, stream=None):
        """One self-play move for every unfinished game (device pointers as ints)."""
        _check(load().gmk_mcts_advance(self.h, d_moves, d_visits, d_lens, d_winner, d_unfinished, int(reuse_subtree), stream))

    def step(self, d_forced_moves, d_moves, d_visits, d_lens, d_winner, d_unfinished, reuse_subtree=False, stream=None):
        """MCTS::stepForward(move) per game: d_forced_moves int16[n] on the device, -1 = the most visited child."""
        _check(load().gmk_mcts_step(self.h, d_forced_moves, d_moves, d_visits, d_lens, d_winner, d_unfinished, int(reuse_subtree), stream))

    def add_root_noise(self, alpha=0.05, epsilon=0.25, stream=None):
        _check(load().gmk_mcts_add_root_noise(self.h, alpha, epsilon, stream))

    def set_option(self, option, value):
        """gmk_mcts_set_option: OPT_NOISE_SAMPLER -> NOISE_SAMPLERS["std" | "counter"], OPT_LOCKSTEP -> 0 / 1."""
        _check(load().gmk_mcts_set_option(self.h, int(option), int(value)))

    def reserve(self, two_arenas=False):
        """gmk_mcts_reserve: allocate the tree arenas now (two per game for the persistent loop with kept subtrees)."""
        _check(load().gmk_mcts_reserve(self.h, int(bool(two_arenas))))

    def selfplay_run(self, n_total, first_game_id, playouts, d_moves, d_visits, d_lens, d_winner, open_moves=None, open_lens=None,
                     reuse_subtree=False, root_noise=None, stream=None):
        """gmk_selfplay_run: the handle's games are slots that play n_total whole games between them (continuous batching on the
        device).  open_moves uint8[n_total, stride] / open_lens int32[n_total] (host) or None; outputs are device pointers (ints),
        indexed by game.  Returns the number of search launches it took."""
        played = C.c_int32()
        om = ol = None
        stride = 0
        if open_moves is not None:
            om = np.ascontiguousarray(open_moves, dtype=np.uint8)
            ol = np.ascontiguousarray(open_lens, dtype=np.int32)
            assert om.ndim == 2 and om.shape[0] == n_total and ol.shape == (n_total,)
            stride = om.shape[1]
        alpha, eps = root_noise if root_noise is not None else (0.0, 0.0)
        _check(load().gmk_selfplay_run(self.h, int(n_total), int(first_game_id), int(playouts), int(reuse_subtree), float(alpha), float(eps),
                                       None if om is None else om.ctypes.data, stride, None if ol is None else ol.ctypes.data,
                                       d_moves, d_visits, d_lens, d_winner, C.byref(played), stream))
        return played.value

    def alg_bytes(self):
        b = C.c_uint64()
        _check(load().gmk_mcts_alg_bytes(self.h, C.byref(b)))
        return b.value

    def launch_info(self):
        g, b, l = C.c_int(), C.c_int(), C.c_int()
        _check(load().gmk_mcts_launch_info(self.h, C.byref(g), C.byref(b), C.byref(l)))
        return {"grid": g.value, "block": b.value, "lds_bytes": l.value}


def visits_to_pi(visits, stones):
    v = np.ascontiguousarray(visits, dtype=np.uint32)
    pi = np.zeros(N, dtype=np.float32)
    _check(load().gmk_visits_to_pi(v.ctypes.data, int(stones), pi.ctypes.data))
    return pi


def samples_from_records(d_moves, d_lens, d_visits, d_winner, d_sample_game, d_sample_move, n_samples, augment,
                         d_states, d_values, d_pi, stream=None):
    """Device-pointer form of gmk_samples_from_records (K4 + K5)."""
    _check(load().gmk_samples_from_records(d_moves, d_lens, d_visits, d_winner, d_sample_game, d_sample_move, n_samples,
                                           int(augment), d_states, d_values, d_pi, stream))


# ---------------- K2: incrementally maintained evaluator states ----------------
class EvaluatorStates:
    """n_games device-resident Evaluator objects (gmk_evalstate_*): apply / revert moves, read the members back."""

    APPLY_NONE, REVERT = -1, -2

    def __init__(self, n_games):
        init()
        self.n = n_games
        h = C.c_void_p()
        _check(load().gmk_evalstate_create(n_games, C.byref(h)))
        self.h = h

    def close(self):
        if getattr(self, "h", None) and load is not None:
            load().gmk_evalstate_destroy(self.h)
            self.h = None

    __del__ = close

    def reset(self):
        _check(load().gmk_evalstate_reset(self.h))

    def update(self, moves):
        """moves int16[n, k]: per game k entries (cell >= 0 apply, -1 nothing, -2 revert the last move)."""
        moves = np.ascontiguousarray(moves, dtype=np.int16)
        assert moves.ndim == 2 and moves.shape[0] == self.n
        _check(load().gmk_evalstate_update_host(self.h, moves.ctypes.data, moves.shape[1]))

    def read(self):
        out = {"scores": np.zeros((self.n, 4, N), np.int32), "density": np.zeros((self.n, 2, 2, N), np.int32),
               "pattern_dist": np.zeros((self.n, 226, 8), np.uint32), "compound_dist": np.zeros((self.n, 226, 3), np.uint32),
               "meta": np.zeros((self.n, 4), np.int32), "record": np.zeros((self.n, 228), np.uint8)}
        _check(load().gmk_evalstate_read(self.h, out["scores"].ctypes.data, out["density"].ctypes.data, out["pattern_dist"].ctypes.data,
                                         out["compound_dist"].ctypes.data, out["meta"].ctypes.data, out["record"].ctypes.data))
        return out


# ---------------- K6: pattern-guided search (TraditionalPolicy), one tree + one evaluator per game ----------------
class TraditionalMCTS:
    """n_games searches of MCTS(policy=TraditionalPolicy(c_puct)) run side by side on the GPU (gmk_trad_*).
    set_positions(move lists) = a fresh root at that position (the games' evaluators persist and are synchronised, like
    the reference's policy object); run(playouts) iterates MCTS::playout; root_stats() reads the roots."""

    def __init__(self, n_games, node_capacity=1 << 20, c_puct=5.0):
        init()
        self.n, self.c_puct = n_games, float(c_puct)
        h = C.c_void_p()
        _check(load().gmk_trad_create(n_games, int(node_capacity), C.byref(h)))
        self.h = h

    def close(self):
        if getattr(self, "h", None) and load is not None:       # (module globals are gone at interpreter shutdown)
            load().gmk_trad_destroy(self.h)
            self.h = None

    __del__ = close

    STATUS_ARENA_FULL, STATUS_EVALUATOR_ERROR, STATUS_BOARD_ONLY_REVERT, STATUS_ILLEGAL_STEP = 1, 2, 4, 8      # bits of root_stats()["status"]

    def reset_evaluators(self):
        _check(load().gmk_trad_reset_evaluators(self.h))

    def set_game_ids(self, ids):
        """The game each slot is playing, relative to the first_game_id of add_root_noise / PoolRAVEMCTS (uint32[n]; default: the
        slot number): random streams belong to the game, not to the slot it runs in."""
        ids = np.ascontiguousarray(ids, dtype=np.uint32)
        assert ids.shape == (self.n,)
        _check(load().gmk_trad_set_game_ids(self.h, ids.ctypes.data))

    def set_positions(self, move_lists, lens=None):
        """One move list per game, or (with lens) the arrays themselves: moves uint8[n, 225], lens int32[n]; a negative length
        leaves that game's position and tree as they are (not on the first call)."""
        if lens is not None:
            moves = np.ascontiguousarray(move_lists, dtype=np.uint8)
            lens = np.ascontiguousarray(lens, dtype=np.int32)
            assert moves.shape == (self.n, N) and lens.shape == (self.n,)
        else:
            moves = np.zeros((self.n, N), np.uint8)
            lens = np.zeros(self.n, np.int32)
            assert len(move_lists) == self.n
            for g, ml in enumerate(move_lists):
                lens[g] = len(ml)
                moves[g, :len(ml)] = ml
        _check(load().gmk_trad_set_positions(self.h, moves.ctypes.data, lens.ctypes.data))

    def run(self, playouts, stream=0):
        _check(load().gmk_trad_run(self.h, int(playouts), self.c_puct, stream))

    def step(self, moves=None):
        """MCTS::stepForward: per game the cell to step to (int16[n]), -1 / None = the most visited child; the subtree is kept."""
        if moves is None:
            _check(load().gmk_trad_step(self.h, None))
        else:
            m = np.ascontiguousarray(moves, dtype=np.int16)
            assert m.shape == (self.n,)
            _check(load().gmk_trad_step(self.h, m.ctypes.data))

    def add_root_noise(self, alpha=0.05, epsilon=0.25, seed=DEFAULT_SEED, first_game_id=0):
        _check(load().gmk_trad_add_root_noise(self.h, alpha, epsilon, seed, first_game_id))

    _POOLRAVE = 0

    def set_option(self, option, value):
        """gmk_trad_set_option: OPT_NOISE_SAMPLER -> NOISE_SAMPLERS["std" | "counter"], OPT_LOCKSTEP -> 0 / 1."""
        _check(load().gmk_trad_set_option(self.h, int(option), int(value)))

    def reserve(self, two_arenas=False):
        """gmk_trad_reserve: the two arenas per slot of the persistent loop with kept subtrees, now."""
        _check(load().gmk_trad_reserve(self.h, int(bool(two_arenas))))

    def selfplay_run(self, n_total, first_game_id, playouts, d_moves, d_visits, d_lens, d_winner, open_moves=None, open_lens=None,
                     reuse_subtree=False, root_noise=None, seed=DEFAULT_SEED, stream=None, max_steps=0, persistent=False):
        """gmk_trad_selfplay_run: the handle's games are slots that play n_total whole games between them, the loop resident on the
        device (search, MCTS::stepForward's move, end-of-game check and slot hand-over are kernels).  open_moves uint8[n_total, stride] /
        open_lens int32[n_total] (host) or None; the outputs are device pointers (ints), indexed by game.
        Returns (search launches, whether a search stopped at its node capacity)."""
        steps, overflow = C.c_int32(), C.c_int32()
        om = ol = None
        stride = 0
        if open_moves is not None:
            om = np.ascontiguousarray(open_moves, dtype=np.uint8)
            ol = np.ascontiguousarray(open_lens, dtype=np.int32)
            assert om.ndim == 2 and om.shape[0] == n_total and ol.shape == (n_total,)
            stride = om.shape[1]
        alpha, eps = root_noise if root_noise is not None else (0.0, 0.0)
        _check(load().gmk_trad_selfplay_run(self.h, self._POOLRAVE, int(n_total), int(first_game_id), int(playouts), self.c_puct, int(seed),
                                            int(bool(reuse_subtree)), float(alpha), float(eps),
                                            None if om is None else om.ctypes.data, stride, None if ol is None else ol.ctypes.data,
                                            d_moves, d_visits, d_lens, d_winner, int(bool(persistent)), int(max_steps), C.byref(overflow), C.byref(steps), stream))
        return steps.value, bool(overflow.value)

    def root_stats(self):
        out = {"visits": np.zeros((self.n, N), np.uint32), "values": np.zeros((self.n, N), np.float32), "priors": np.zeros((self.n, N), np.float32),
               "best": np.zeros(self.n, np.int32), "root_visits": np.zeros(self.n, np.uint32), "root_value": np.zeros(self.n, np.float32),
               "n_nodes": np.zeros(self.n, np.int32), "status": np.zeros(self.n, np.int32), "evaluator_updates": np.zeros(self.n, np.uint64)}
        _check(load().gmk_trad_root_stats(self.h, *[out[k].ctypes.data for k in
               ("visits", "values", "priors", "best", "root_visits", "root_value", "n_nodes", "status", "evaluator_updates")]))
        return out

    def read_evaluators(self):
        out = {"scores": np.zeros((self.n, 4, N), np.int32), "density": np.zeros((self.n, 2, 2, N), np.int32),
               "pattern_dist": np.zeros((self.n, 226, 8), np.uint32), "compound_dist": np.zeros((self.n, 226, 3), np.uint32),
               "meta": np.zeros((self.n, 4), np.int32), "record": np.zeros((self.n, 228), np.uint8)}
        _check(load().gmk_trad_read_evaluators(self.h, out["scores"].ctypes.data, out["density"].ctypes.data, out["pattern_dist"].ctypes.data,
                                               out["compound_dist"].ctypes.data, out["meta"].ctypes.data, out["record"].ctypes.data))
        return out


class PoolRAVEMCTS(TraditionalMCTS):
    """n_games searches of MCTS(policy=PoolRAVEPolicy(c_puct)) side by side on the GPU (K8): the tree, step, noise and root
    statistics of TraditionalMCTS, playouts with one random rollout each and RAVE::BackPropogate<true>."""

    _POOLRAVE = 1

    def __init__(self, n_games, node_capacity=1 << 20, c_puct=2.0, seed=DEFAULT_SEED, first_game_id=0):
        super().__init__(n_games, node_capacity, c_puct)
        self.seed, self.first_game_id = int(seed), int(first_game_id)

    def run(self, playouts, stream=0):
        _check(load().gmk_trad_run_poolrave(self.h, int(playouts), self.c_puct, self.seed, self.first_game_id, stream))

    def add_root_noise(self, alpha=0.05, epsilon=0.25, seed=None, first_game_id=None):
        super().add_root_noise(alpha, epsilon, self.seed if seed is None else seed, self.first_game_id if first_game_id is None else first_game_id)

    def root_stats(self):
        out = super().root_stats()
        out["amaf_visits"] = np.zeros((self.n, N), np.uint32)
        out["amaf_values"] = np.zeros((self.n, N), np.float32)
        _check(load().gmk_trad_root_amaf(self.h, out["amaf_visits"].ctypes.data, out["amaf_values"].ctypes.data))
        return out


# ---------------- K7: network-guided search, many games in lock step ----------------
class AlphaZeroMCTS:
    """n_games searches of MCTS(policy=Policy(eval_state=network.eval_state, c_puct)) (agents/alphazero.py:5-9) advancing one
    playout per step: select() writes the leaves' feature planes into `states` (torch float32 [n, 6, 15, 15] on the GPU), the
    caller's network maps them to (value [n], probs [n, 225]), expand() grows the trees and backs the values up."""

    def __init__(self, n_games, node_capacity=1 << 16, c_puct=5.0):
        import torch
        init()
        self.n = n_games
        h = C.c_void_p()
        _check(load().gmk_az_create(n_games, int(node_capacity), float(c_puct), C.byref(h)))
        self.h = h
        self.states = torch.zeros((n_games, 6, 15, 15), dtype=torch.float32, device="cuda")
        self.live = n_games                                        # rows of the leaf batch (see select)

    def close(self):
        if getattr(self, "h", None) and load is not None:
            load().gmk_az_destroy(self.h)
            self.h = None

    __del__ = close

    STATUS_OVER, STATUS_ARENA_FULL, STATUS_ILLEGAL_STEP = 1, 2, 4      # bits of root_stats()["status"]

    def set_game_ids(self, ids):
        """The game each slot is playing, relative to add_root_noise's first_game_id (uint32[n]; default: the slot number)."""
        ids = np.ascontiguousarray(ids, dtype=np.uint32)
        assert ids.shape == (self.n,)
        _check(load().gmk_az_set_game_ids(self.h, ids.ctypes.data))

    def set_roots(self, planes, last_moves):
        """planes uint16[n,2,16]; last_moves int16[n,2] = (last move, the one before), -1 where there is none."""
        planes = np.ascontiguousarray(planes, dtype=np.uint16)
        last_moves = np.ascontiguousarray(last_moves, dtype=np.int16)
        assert planes.shape == (self.n, 2, 16) and last_moves.shape == (self.n, 2)
        _check(load().gmk_az_set_roots(self.h, planes.ctypes.data, last_moves.ctypes.data))
        self.n_total = None
        self._refresh_live()

    def _refresh_live(self):
        """self.live = rows of the leaf batch: the games still played, in slot order (gmk_az_live_games)"""
        n = C.c_int32(0)
        _check(load().gmk_az_live_games(self.h, C.byref(n)))
        self.live = n.value

    def select(self, stream=None):
        import torch
        stream = torch.cuda.current_stream().cuda_stream if stream is None else stream
        _check(load().gmk_az_select(self.h, self.states.data_ptr(), stream))
        return self.states[:self.live]                             # the games still played (all of them until advance() ends one)

    def expand(self, values, probs, stream=None):
        import torch
        assert values.dtype == torch.float32 and probs.dtype == torch.float32 and values.is_contiguous() and probs.is_contiguous()
        assert values.numel() == self.live and probs.numel() == self.live * N
        stream = torch.cuda.current_stream().cuda_stream if stream is None else stream
        _check(load().gmk_az_expand(self.h, values.data_ptr(), probs.data_ptr(), stream))

    def step(self, moves=None):
        """MCTS::stepForward for every game: int16[n] cells, -1 / None = the most visited child; the subtree is kept."""
        if moves is None:
            _check(load().gmk_az_step(self.h, None))
        else:
            m = np.ascontiguousarray(moves, dtype=np.int16)
            assert m.shape == (self.n,)
            _check(load().gmk_az_step(self.h, m.ctypes.data))

    def set_slots(self, n_total, open_moves=None, open_lens=None):
        """Continuous batching (gmk_az_set_slots): the n slots of this handle play n_total games between them, from their openings
        (uint8[n_total, stride], int32[n_total]); takes set_roots's place.  advance() then wants records of n_total rows."""
        self.n_total = int(n_total)
        if open_moves is None:
            _check(load().gmk_az_set_slots(self.h, int(n_total), None, 0, None))
        else:
            m = np.ascontiguousarray(open_moves, dtype=np.uint8)
            l = np.ascontiguousarray(open_lens, dtype=np.int32)
            assert m.ndim == 2 and m.shape[0] == n_total and l.shape == (n_total,)
            _check(load().gmk_az_set_slots(self.h, int(n_total), m.ctypes.data, m.shape[1], l.ctypes.data))
        self._refresh_live()

    def advance(self, moves, visits, lens, winner, reuse_subtree=True, stream=None):
        """One self-play move for every game still played, on the device (gmk_az_advance): moves uint8[n,225], visits int16/uint16
        [n,225,225] or None, lens int32[n], winner int8[n] are torch tensors on the GPU (the games' records, openings included);
        returns the number of games that go on."""
        import torch
        stream = torch.cuda.current_stream().cuda_stream if stream is None else stream
        rows = getattr(self, "n_total", None) or self.n
        assert moves.dtype == torch.uint8 and moves.shape == (rows, N) and lens.dtype == torch.int32 and lens.shape == (rows,) and winner.dtype == torch.int8 and winner.shape == (rows,)
        assert visits is None or (visits.element_size() == 2 and visits.shape == (rows, N, N))
        unfinished = C.c_int32(0)
        _check(load().gmk_az_advance(self.h, moves.data_ptr(), visits.data_ptr() if visits is not None else None, lens.data_ptr(), winner.data_ptr(),
                                     int(bool(reuse_subtree)), C.byref(unfinished), stream))
        self._refresh_live()
        return unfinished.value

    def add_root_noise(self, alpha=0.05, epsilon=0.25, seed=DEFAULT_SEED, first_game_id=0):
        _check(load().gmk_az_add_root_noise(self.h, alpha, epsilon, seed, first_game_id))

    def set_option(self, option, value):
        """gmk_az_set_option: OPT_NOISE_SAMPLER -> NOISE_SAMPLERS["std" | "counter"]."""
        _check(load().gmk_az_set_option(self.h, int(option), int(value)))

    def search(self, network, playouts, graph=False):
        """`playouts` lock-step playouts; network(states) -> (value [n], probs [n, 225]) on the GPU.  graph=True: the first playout
        runs eagerly, then ONE playout step (select kernel, the network's kernels, expand kernel) is captured into a hipGraph and
        replayed for the rest, which removes the launch gaps between the ~10 kernels of a step; the network must be capturable
        (no host synchronisation; PolicyValueNetwork and FusedPolicyValueNetwork are)."""
        import torch

        def step():
            values, probs = network(self.select())
            self.expand(values.contiguous(), probs.contiguous())

        if not graph or playouts < 3:
            for _ in range(playouts):
                step()
            return
        step()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            step()
        for _ in range(playouts - 1):
            g.replay()

    def root_stats(self):
        out = {"visits": np.zeros((self.n, N), np.uint32), "values": np.zeros((self.n, N), np.float32), "priors": np.zeros((self.n, N), np.float32),
               "root_visits": np.zeros(self.n, np.uint32), "root_value": np.zeros(self.n, np.float32),
               "n_nodes": np.zeros(self.n, np.int32), "status": np.zeros(self.n, np.int32)}
        _check(load().gmk_az_root_stats(self.h, *[out[k].ctypes.data for k in ("visits", "values", "priors", "root_visits", "root_value", "n_nodes", "status")]))
        return out
