"""Batched self-play on the GPU and the one exchange step of the path: gathering finished game records.

Replaces the reference's data generation loop -- `network/data_helper.py:58-83` (`simulate_game_data` ->
`dual_play(..., verbose=True)`) run by `DATA_CONFIG["process_num"]` = 2 Python processes that append to a
`multiprocessing.Manager().list()` (`data_helper.py:97-113, 152-169`) -- by thousands of concurrent games per
GPU (K3 + `gmk_mcts_advance`) sharded over ranks by global game id, and ONE gather of the compact records
(moves, per-move root visit counts, winner) to rank 0 over `torch.distributed` (RCCL on MI355X, gloo on CPU).
"""
import os

import numpy as np
import torch

from . import lib as G

N = 225


class GameRecords:
    """Fixed-stride game records: moves u8[n,225], lens i32[n], winner i8[n], visits u16[n,225,225] (optional)."""

    def __init__(self, moves, lens, winner, visits=None, first_game_id=0, overflow=False):
        self.moves, self.lens, self.winner, self.visits, self.first_game_id = moves, lens, winner, visits, first_game_id
        self.overflow = overflow            # a tree arena filled up during some search (results then deviate)

    def __len__(self):
        return int(self.lens.shape[0])

    def cpu(self):
        return GameRecords(self.moves.cpu(), self.lens.cpu(), self.winner.cpu(),
                           None if self.visits is None else self.visits.cpu(), self.first_game_id, self.overflow)

    def to_samples(self, augment=False, first_move=0):
        """All training tuples of all games, built ON THE GPU (K4 + K5, optionally with the eight-fold augmentation of
        network/data_helper.py:36-55): -> (states uint8[S,6,15,15], values float32[S], pi float32[S,225]) as torch
        tensors on the records' device, samples ordered by (game, move[, symmetry]).  first_move skips opening plies
        that were not searched."""
        assert self.visits is not None and self.moves.is_cuda
        dev = self.moves.device
        lens = self.lens.cpu().numpy()
        game = np.concatenate([np.full(max(int(l) - first_move, 0), g, dtype=np.int32) for g, l in enumerate(lens)] or [np.zeros(0, np.int32)])
        move = np.concatenate([np.arange(first_move, int(l), dtype=np.int32) for l in lens] or [np.zeros(0, np.int32)])
        n = int(game.shape[0])
        copies = 8 if augment else 1
        d_game, d_move = torch.from_numpy(game).to(dev), torch.from_numpy(move).to(dev)
        states = torch.empty((n * copies, 6, 15, 15), dtype=torch.uint8, device=dev)
        values = torch.empty(n * copies, dtype=torch.float32, device=dev)
        pi = torch.empty((n * copies, N), dtype=torch.float32, device=dev)
        G.samples_from_records(self.moves.data_ptr(), self.lens.data_ptr(), self.visits.data_ptr(), self.winner.data_ptr(),
                               d_game.data_ptr(), d_move.data_ptr(), n, augment, states.data_ptr(), values.data_ptr(), pi.data_ptr(),
                               torch.cuda.current_stream(dev).cuda_stream)
        return states, values, pi

    def samples(self, game):
        """The training tuples of one game as `dual_play(verbose=True)` returns them (agents/utils.py:36-40, 55-59):
        [(uint8[6,15,15] encoded states, float score for the player to move, float32[225] pi)]."""
        assert self.visits is not None
        rec = self.cpu()
        L = int(rec.lens[game])
        winner = int(rec.winner[game])
        cell = np.zeros(N, dtype=np.int8)
        out = []
        for t in range(L):
            cur = 1 if t % 2 == 0 else -1
            states = np.zeros((6, N), dtype=np.uint8)
            states[0] = cell == cur
            states[1] = cell == -cur
            states[2] = cell == 0
            if t >= 1:
                states[3, int(rec.moves[game, t - 1])] = 1
            if t >= 2:
                states[4, int(rec.moves[game, t - 2])] = 1
            states[5] = cur == 1
            pi = G.visits_to_pi(rec.visits[game, t].numpy().astype(np.uint32), t)
            out.append((states.reshape(6, 15, 15), np.array(float(cur * winner)), pi))
            cell[int(rec.moves[game, t])] = cur
        return out


# Games in flight per MI355X for whole-game RandomPolicy self-play, and the number of search handles they are split over.  The loop is ONE
# persistent launch per handle: 8 192 slots are the 2 048 wavefronts (four games each, two per SIMD) the chip holds, so one handle of 8 192 slots fills
# it and keeps it filled until the games run out.  Measured in one process, alternating (tools/k3_handles_ab.py, 32 768 games x 800 playouts, M playouts/s):
# one handle x 8 192 slots 134.9 / 135.5; one x 16 384 (the second half of the wavefronts queues behind the first) 127.3 / 126.9; two handles x 8 192 on
# two streams and host threads -- round 3's default, the best plan of the LOCK-STEP loop, where two handles filled each other's ends of launches --
# 127.9 / 112.4 (and the run-to-run spread VERDICT r3 noted); two x 4 096: 91.
# With kept subtrees a slot owns TWO arenas of three times the nodes (the kept subtree + the new search; 17 MB per slot at 800 playouts per move):
# 8 192 slots are 142 GB of the 288.
SLOTS_PER_GPU = 8192
HANDLES_FROM_GAMES = 1 << 30        # (no longer a default: handles="auto" is one handle; handles=k is still honoured)
SLOTS_PER_GPU_KEPT = 8192


def plan_games(n_games, slots="auto", handles="auto", opening_plies=0, reuse_subtree=False):
    """How play_games spreads n_games over search handles and slots: a list of (first game, one past the last game, slots) per handle,
    or None for the plain lock-step loop (one handle, every game in flight from the start).  Pure bookkeeping, no GPU."""
    if n_games <= 0:
        raise ValueError("play_games: n_games must be positive")
    if handles == "auto":
        handles = 1
    if slots == "auto":
        per_gpu = SLOTS_PER_GPU_KEPT if reuse_subtree else SLOTS_PER_GPU
        slots = per_gpu if n_games > per_gpu and opening_plies <= 8 else None
    handles = max(1, min(int(handles), n_games))
    if not ((slots is not None and slots < n_games) or handles > 1):
        return None
    if opening_plies > 8:
        raise ValueError("play_games: slots and handles take openings of at most 8 plies")
    total_slots = n_games if slots is None else max(handles, min(int(slots), n_games))
    blocks = [((n_games * i) // handles, (n_games * (i + 1)) // handles) for i in range(handles)]
    return [(lo, hi, min(hi - lo, -(-total_slots // handles))) for lo, hi in blocks]


def play_games(n_games, playouts, seed=G.DEFAULT_SEED, first_game_id=0, c_puct=5.0, c_rollouts=5,
               opening_plies=0, record_visits=True, reuse_subtree=False, root_noise=None, max_moves=N, device=None,
               node_capacity=None, slots="auto", handles="auto", noise_sampler="counter", lockstep=False, prepare_only=False):
    """Plays n_games complete games on the current GPU: every move = one K3 search of `playouts` playouts for all
    unfinished games, then `gmk_mcts_advance`.  Game g uses the global id first_game_id + g for its RNG streams, so
    the records do not depend on how games are spread over GPUs, slots or handles.
    slots: at most that many games in flight, with CONTINUOUS BATCHING on the device (gmk_selfplay_run): a slot whose game ends takes
    the next unstarted game inside the step kernel, so the searches stay full instead of waiting for the longest game of the batch.
    "auto" = SLOTS_PER_GPU when there are more games than that, else all games at once; None = all games at once.
    handles: the games are split into that many contiguous blocks, each with its own search handle, HIP stream and host thread
    (the slots are shared out between them); "auto" = 1 (one persistent launch fills the chip, see SLOTS_PER_GPU).
    reuse_subtree + root_noise=(alpha, epsilon) are the reference agent's per-move semantics (agents/mcts.py:17-21: the chosen child's subtree is
    the next search's tree, MCTS.cpp:129-147, and Default::AddNoise runs before every search, MCTS.cpp:182).  noise_sampler: "counter" = the
    counter-based Dirichlet sampler of include/gomoku_noise.h, drawn inside the searching kernel, so that the whole run is ONE persistent launch
    (slots / handles != None); "std" = std::gamma_distribution on the host, which needs a launch boundary per move (lock step).  lockstep=True
    forces the search-by-search loop (the form the tests hold the persistent one to).
    prepare_only: create the search handles this call would create, allocate their tree arenas and give them back -- to the library's block pool,
    where the same call without prepare_only finds them (the arenas are tens of GB, and the driver clears memory it has handed out before at seconds
    per 24 GB: a self-play job pays that once, not per batch; a measurement keeps it out of its timed region this way).  Returns None."""
    G.init(torch.cuda.current_device() if device is None else device.index)
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else device
    stream = torch.cuda.current_stream(dev).cuda_stream
    # (the device-resident loop plays whole games: a cap on the moves per game takes the lock-step path below, which honours it)
    plan = plan_games(n_games, slots, handles, opening_plies, reuse_subtree) if max_moves >= N else None
    if plan is None and max_moves >= N and opening_plies <= 8 and not lockstep:
        plan = [(0, n_games, n_games)]                  # all games at once, the loop on the device all the same (gmk_selfplay_run: ONE launch where it can)
    if plan is not None:
        handles = len(plan)
        open_moves = open_lens = None
        if opening_plies > 0:
            m, l, _ = G.synth_boards(n_games, 0, seed=seed, first_board=first_game_id)
            open_moves, open_lens = m, np.minimum(l, opening_plies).astype(np.int32)
        cap = node_capacity if node_capacity is not None else playouts * N * (3 if reuse_subtree else 1) + 1
        d_moves = torch.zeros((n_games, N), dtype=torch.uint8, device=dev)
        d_lens = torch.zeros(n_games, dtype=torch.int32, device=dev)
        d_winner = torch.zeros(n_games, dtype=torch.int8, device=dev)
        d_visits = torch.zeros((n_games, N, N), dtype=torch.int16, device=dev) if record_visits else None
        noise = root_noise if reuse_subtree else None
        blocks = [(lo, hi) for lo, hi, _ in plan]
        trees = [G.BatchedMCTS(n_slots, c_puct=c_puct, c_rollouts=c_rollouts, seed=seed, node_capacity=cap) for _, _, n_slots in plan]
        for tree in trees:
            tree.set_option(G.OPT_NOISE_SAMPLER, G.NOISE_SAMPLERS[noise_sampler])
            tree.set_option(G.OPT_LOCKSTEP, int(bool(lockstep)))
        if prepare_only:
            persistent = not lockstep and not (reuse_subtree and root_noise is not None and noise_sampler != "counter")
            for tree in trees:
                tree.reserve(two_arenas=reuse_subtree and persistent)
            for tree in trees:
                tree.close()
            return None

        def run_block(i, hip_stream):
            lo, hi = blocks[i]
            trees[i].selfplay_run(hi - lo, first_game_id + lo, playouts, d_moves[lo:].data_ptr(), d_visits[lo:].data_ptr() if record_visits else None,
                                  d_lens[lo:].data_ptr(), d_winner[lo:].data_ptr(), None if open_moves is None else open_moves[lo:hi],
                                  None if open_lens is None else open_lens[lo:hi], reuse_subtree, noise, hip_stream)

        if handles == 1:
            run_block(0, stream)
        else:
            import threading
            torch.cuda.current_stream(dev).synchronize()          # the record buffers above are zeroed before another stream writes them
            side = [torch.cuda.Stream(dev) for _ in range(handles)]
            failed = [None] * handles

            def worker(i):
                try:
                    torch.cuda.set_device(dev)                     # the HIP device is a property of the thread
                    run_block(i, side[i].cuda_stream)              # returns with its stream drained
                except BaseException as exc:
                    failed[i] = exc

            threads = [threading.Thread(target=worker, args=(i,)) for i in range(handles)]
            for t in threads:
                t.start()
            for t in threads:
                t.join()
            for exc in failed:
                if exc is not None:
                    raise exc
        overflow = False
        for tree in trees:
            overflow = overflow or bool((tree.root_stats()[4] & G.BatchedMCTS.STATUS_ARENA_FULL).any())
            tree.close()
        return GameRecords(d_moves, d_lens, d_winner, d_visits, first_game_id, overflow)
    if prepare_only:
        return None
    planes = np.zeros((n_games, 2, 16), dtype=np.uint16)
    last = np.full(n_games, -1, dtype=np.int16)
    moves0 = np.zeros((n_games, N), dtype=np.uint8)
    lens0 = np.zeros(n_games, dtype=np.int32)
    if opening_plies > 0:
        m, l, _ = G.synth_boards(n_games, 0, seed=seed, first_board=first_game_id)
        lens0 = np.minimum(l, opening_plies).astype(np.int32)
        planes = G.moves_to_planes(m, lens0)
        for g in range(n_games):
            moves0[g, :lens0[g]] = m[g, :lens0[g]]
            last[g] = m[g, lens0[g] - 1] if lens0[g] > 0 else -1
    cap = node_capacity if node_capacity is not None else playouts * N * (3 if reuse_subtree else 1) + 1
    tree = G.BatchedMCTS(n_games, c_puct=c_puct, c_rollouts=c_rollouts, seed=seed, node_capacity=cap)
    tree.set_option(G.OPT_NOISE_SAMPLER, G.NOISE_SAMPLERS[noise_sampler])
    tree.set_roots(planes, last, first_game_id)
    d_moves = torch.from_numpy(moves0).to(dev)
    d_lens = torch.from_numpy(lens0).to(dev)
    d_winner = torch.zeros(n_games, dtype=torch.int8, device=dev)
    d_visits = torch.zeros((n_games, N, N), dtype=torch.int16, device=dev) if record_visits else None
    d_unfinished = torch.ones(1, dtype=torch.int32, device=dev)
    for _ in range(max_moves):
        if root_noise is not None and reuse_subtree:      # Default::AddNoise at the start of every search (MCTS.cpp:182)
            tree.add_root_noise(root_noise[0], root_noise[1], stream)
        tree.run(playouts, stream)
        tree.advance(d_moves.data_ptr(), d_visits.data_ptr() if record_visits else None, d_lens.data_ptr(),
                     d_winner.data_ptr(), d_unfinished.data_ptr(), reuse_subtree, stream)
        if int(d_unfinished.item()) == 0:             # 4-byte readback per move: the only host sync in the loop
            break
    status = tree.root_stats()[4]
    tree.close()
    return GameRecords(d_moves, d_lens, d_winner, d_visits, first_game_id, bool((status & G.BatchedMCTS.STATUS_ARENA_FULL).any()))


class _HostGames:
    """The boards of n games on the host, all games handled at once with numpy (no Python loop per game): Board::applyMove with
    its victory check (Game.cpp:37-49, 88-136: five or more through the new stone wins, a full board is a tie) for the moves a
    batched searcher just chose.  The searchers only propose legal moves; `over` games take none."""

    def __init__(self, n):
        self.n = n
        self.moves = np.zeros((n, N), dtype=np.uint8)
        self.lens = np.zeros(n, dtype=np.int32)
        self.stones = np.zeros((n, 15, 15), dtype=np.int8)      # +1 black, -1 white
        self.over = np.zeros(n, dtype=bool)
        self.winner = np.zeros(n, dtype=np.int8)

    def apply(self, played):
        """played int[n]: the cell each game plays, -1 for none; returns the indices of the games that moved."""
        played = np.asarray(played)
        idx = np.nonzero((played >= 0) & ~self.over)[0]
        if len(idx) == 0:
            return idx
        cells = played[idx].astype(np.int64)
        y, x = cells // 15, cells % 15
        colour = np.where(self.lens[idx] % 2 == 0, 1, -1).astype(np.int8)      # black moves on even stone counts
        assert (self.stones[idx, y, x] == 0).all(), "a searcher proposed an occupied cell"
        self.stones[idx, y, x] = colour
        self.moves[idx, self.lens[idx]] = cells
        self.lens[idx] += 1
        five = np.zeros(len(idx), dtype=bool)
        for dy, dx in ((0, 1), (1, 0), (1, 1), (1, -1)):
            run = np.ones(len(idx), dtype=np.int32)
            for sign in (1, -1):
                alive = np.ones(len(idx), dtype=bool)
                for k in range(1, 5):
                    yy, xx = y + sign * k * dy, x + sign * k * dx
                    inside = (yy >= 0) & (yy < 15) & (xx >= 0) & (xx < 15)
                    alive &= inside
                    alive[alive] = self.stones[idx[alive], yy[alive], xx[alive]] == colour[alive]
                    run += alive
            five |= run >= 5
        self.winner[idx[five]] = colour[five]
        self.over[idx] = five | (self.lens[idx] == N)
        return idx

    def open_with(self, moves, lens, plies):
        for i in range(plies):
            played = np.where(lens > i, moves[:, i].astype(np.int64), -1)
            self.apply(played)

    def last_two(self):
        last = np.full((self.n, 2), -1, dtype=np.int16)
        g = np.arange(self.n)
        has1, has2 = self.lens > 0, self.lens > 1
        last[has1, 0] = self.moves[g[has1], self.lens[has1] - 1]
        last[has2, 1] = self.moves[g[has2], self.lens[has2] - 2]
        return last


def _play_supervisor_on_device(n_games, n_slots, playouts, c_puct, seed, first_game_id, opening_plies, device, node_capacity, policy, reuse_subtree, root_noise, max_steps=0,
                               persistent=False, noise_sampler="std", prepare_only=False):
    """play_supervisor_games with the loop resident on the device (gmk_trad_selfplay_run): the searches, MCTS::stepForward's move, the
    end-of-game check and the hand-over of a finished game's slot are kernels; the host reads four bytes per move.  Same games, same
    records as the host-driven loops below (tests/test_selfplay_gpu.py holds them to each other)."""
    G.init(torch.cuda.current_device() if device is None else device.index)
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else device
    stream = torch.cuda.current_stream(dev).cuda_stream
    open_moves = open_lens = None
    if opening_plies > 0:
        m, l, _ = G.synth_boards(n_games, 0, seed=seed, first_board=first_game_id)
        open_moves, open_lens = m, np.minimum(l, opening_plies).astype(np.int32)
    cap = node_capacity if node_capacity is not None else min((3 if reuse_subtree else 1) * playouts * 226 + 1, (1 << 24) - 1)
    if policy == "poolrave":
        tree = G.PoolRAVEMCTS(n_slots, node_capacity=cap, c_puct=c_puct, seed=seed, first_game_id=first_game_id)
    elif policy == "traditional":
        tree = G.TraditionalMCTS(n_slots, node_capacity=cap, c_puct=c_puct)
    else:
        raise ValueError("play_supervisor_games: policy must be 'traditional' or 'poolrave'")
    tree.set_option(G.OPT_NOISE_SAMPLER, G.NOISE_SAMPLERS[noise_sampler])
    if prepare_only:                                             # the arenas this run would allocate go to the library's pool (see play_games)
        tree.reserve(two_arenas=bool(reuse_subtree and persistent))
        tree.close()
        return None
    d_moves = torch.zeros((n_games, N), dtype=torch.uint8, device=dev)
    d_lens = torch.zeros(n_games, dtype=torch.int32, device=dev)
    d_winner = torch.zeros(n_games, dtype=torch.int8, device=dev)
    d_visits = torch.zeros((n_games, N, N), dtype=torch.int16, device=dev)
    torch.cuda.current_stream(dev).synchronize()
    try:
        _, overflow = tree.selfplay_run(n_games, first_game_id, playouts, d_moves.data_ptr(), d_visits.data_ptr(), d_lens.data_ptr(), d_winner.data_ptr(),
                                        open_moves, open_lens, reuse_subtree, root_noise, seed, stream, max_steps, persistent)
    finally:
        tree.close()
    return GameRecords(d_moves, d_lens, d_winner, d_visits, first_game_id, overflow)


def play_supervisor_games(n_games, playouts, c_puct=5.0, seed=G.DEFAULT_SEED, first_game_id=0, opening_plies=0, max_moves=N,
                          device=None, node_capacity=None, reuse_subtree=False, root_noise=None, policy="traditional", slots=None, device_loop=True, max_steps=0,
                          noise_sampler="counter", prepare_only=False):
    """n_games complete games of the reference's self-play SUPERVISOR against itself (config.py:9-12: "traditional_mcts",
    MCTS(TraditionalPolicy) on both sides), all games side by side on the current GPU: every move = one K6 search of
    `playouts` playouts per unfinished game (the games' evaluators are kept and synchronised like the policy objects of
    the reference; with reuse_subtree the chosen child's subtree is kept too, MCTS::stepForward, and root_noise =
    (alpha, epsilon) mixes Default::AddNoise into the root priors before every search), then MCTS::stepForward's choice
    is played.  Without noise the search is deterministic; variety then comes from the openings (synthetic generator,
    `opening_plies` plies of game first_game_id + g).  policy="poolrave" plays the same loop with MCTS(PoolRAVEPolicy)
    (agents/mcts.py:36-40) on both sides: K8 searches, random rollouts seeded by (seed, first_game_id + g).
    slots: at most that many games are in flight; a game that ends hands its slot -- tree arena, evaluator,
    wavefront -- to the next unstarted game, so the GPU stays full instead of waiting for the longest game of the batch (a search
    costs the same time however many of its games are still alive: one wavefront per game, latency bound).
    device_loop (default): the loop runs on the device (gmk_trad_selfplay_run; whole games only, so a max_moves cap takes the host loop);
    device_loop=False: the host drives it ply by ply with numpy boards (root_stats down, positions up every ply) -- the same games.
    max_steps > 0 (device loop only): stop after that many moves per slot, whatever is unfinished (throughput measurements with every slot busy).
    noise_sampler: "counter" (default) = the counter-based Dirichlet sampler of include/gomoku_noise.h, drawn on the device -- inside the ONE launch of the
    persistent loop, so that the reference agent's semantics (reuse_subtree + root_noise) run at the pace of the plain loop; "std" = std::gamma_distribution
    over std::mt19937 on the host (lock step).  PoolRAVE games draw "std" whatever is asked (their loop is lock step anyway).
    Returns the same GameRecords as play_games (moves, per-move root visit counts, winner), so to_samples() / gather_records() apply."""
    if policy not in ("traditional", "poolrave"):
        raise ValueError("play_supervisor_games: policy must be 'traditional' or 'poolrave'")
    if device_loop and max_moves >= N:
        n_slots = n_games if slots is None else max(1, min(int(slots), n_games))
        # "persistent": one launch, every slot plays game after game at its own pace (TraditionalPolicy, whole games; kept subtrees are compacted
        # inside the launch, root noise comes from the counter-based sampler); each game on a fresh evaluator, so the records equal the all-at-once
        # loop's whatever the slots.  Chosen by itself when games outnumber slots or the reference agent's semantics are asked for and the
        # configuration allows it; device_loop="lockstep" keeps the search-by-search loop (a slot's evaluator carries over).
        if policy == "poolrave":
            noise_sampler = "std"
        noisy = root_noise is not None and reuse_subtree
        can_persist = policy == "traditional" and not max_steps and not (noisy and noise_sampler != "counter")
        if device_loop == "persistent" and not can_persist:
            raise ValueError("play_supervisor_games: the persistent loop plays TraditionalPolicy games to their end, with root noise from the counter-based sampler only")
        persistent = can_persist and (device_loop == "persistent" or (device_loop is True and (n_slots < n_games or reuse_subtree)))
        return _play_supervisor_on_device(n_games, n_slots, playouts, c_puct, seed, first_game_id,
                                          opening_plies, device, node_capacity, policy, reuse_subtree, root_noise, max_steps, persistent, noise_sampler, prepare_only)
    if prepare_only:
        return None
    if max_steps:
        raise ValueError("play_supervisor_games: max_steps is a switch of the device-resident loop")
    if slots is not None and slots < n_games:
        return _play_supervisor_slots(n_games, int(slots), playouts, c_puct, seed, first_game_id, opening_plies, device, node_capacity, policy,
                                      reuse_subtree, root_noise, noise_sampler)
    G.init(torch.cuda.current_device() if device is None else device.index)
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else device
    stream = torch.cuda.current_stream(dev).cuda_stream
    games = _HostGames(n_games)                                  # host boards: Board::applyMove / checkGameEnd for all games at once
    if opening_plies > 0:
        m, l, _ = G.synth_boards(n_games, 0, seed=seed, first_board=first_game_id)
        games.open_with(m, l, opening_plies)
    visits = np.zeros((n_games, N, N), dtype=np.uint16)
    cap = node_capacity if node_capacity is not None else min((3 if reuse_subtree else 1) * playouts * 226 + 1, (1 << 24) - 1)
    if policy == "poolrave":
        tree = G.PoolRAVEMCTS(n_games, node_capacity=cap, c_puct=c_puct, seed=seed, first_game_id=first_game_id)
    elif policy == "traditional":
        tree = G.TraditionalMCTS(n_games, node_capacity=cap, c_puct=c_puct)
    else:
        raise ValueError("play_supervisor_games: policy must be 'traditional' or 'poolrave'")
    tree.set_option(G.OPT_NOISE_SAMPLER, G.NOISE_SAMPLERS["std" if policy == "poolrave" else noise_sampler])
    overflow = False
    for ply in range(max_moves):
        if games.over.all():
            break
        if ply == 0 or not reuse_subtree:
            tree.set_positions(games.moves, games.lens)
        if root_noise is not None:
            tree.add_root_noise(root_noise[0], root_noise[1], seed=seed, first_game_id=first_game_id)
        tree.run(playouts, stream)
        st = tree.root_stats()
        overflow |= bool((st["status"] & G.TraditionalMCTS.STATUS_ARENA_FULL).any())
        played = np.where(games.over, -1, st["best"]).astype(np.int16)
        games.over |= played < 0                                 # no child: nothing the policy wants to play (cannot happen on a live board)
        at = games.lens.copy()
        moved = games.apply(played)
        visits[moved, at[moved]] = np.minimum(st["visits"][moved], 65535)
        if reuse_subtree:
            tree.step(played)                               # finished games ask for -1 on a childless root: nothing moves
    tree.close()
    return GameRecords(torch.from_numpy(games.moves).to(dev), torch.from_numpy(games.lens).to(dev), torch.from_numpy(games.winner).to(dev),
                       torch.from_numpy(visits.view(np.int16)).to(dev), first_game_id, overflow)


def _play_supervisor_slots(n_games, slots, playouts, c_puct, seed, first_game_id, opening_plies, device, node_capacity, policy,
                           reuse_subtree=False, root_noise=None, noise_sampler="std"):
    G.init(torch.cuda.current_device() if device is None else device.index)
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else device
    stream = torch.cuda.current_stream(dev).cuda_stream
    games = _HostGames(n_games)
    if opening_plies > 0:
        m, l, _ = G.synth_boards(n_games, 0, seed=seed, first_board=first_game_id)
        games.open_with(m, l, opening_plies)
    visits = np.zeros((n_games, N, N), dtype=np.uint16)
    cap = node_capacity if node_capacity is not None else min((3 if reuse_subtree else 1) * playouts * 226 + 1, (1 << 24) - 1)
    if policy == "poolrave":
        tree = G.PoolRAVEMCTS(slots, node_capacity=cap, c_puct=c_puct, seed=seed, first_game_id=first_game_id)
    elif policy == "traditional":
        tree = G.TraditionalMCTS(slots, node_capacity=cap, c_puct=c_puct)
    else:
        raise ValueError("play_supervisor_games: policy must be 'traditional' or 'poolrave'")
    tree.set_option(G.OPT_NOISE_SAMPLER, G.NOISE_SAMPLERS["std" if policy == "poolrave" else noise_sampler])
    active = np.arange(slots)                                    # the game in each slot
    next_game = slots
    overflow = False
    fresh = np.ones(slots, dtype=bool)                           # slots whose game starts from a new root at the next search
    for _ in range((n_games // slots + 2) * N):
        # a finished game hands its slot to the next unstarted one
        done = np.nonzero(games.over[active])[0]
        take = min(len(done), n_games - next_game)
        active[done[:take]] = np.arange(next_game, next_game + take)
        fresh[done[:take]] = True
        next_game += take
        if games.over[active].all():
            break
        tree.set_game_ids(active)                                # root noise and rollout streams are the GAME's (first_game_id + its number), whatever slot it runs in
        if not reuse_subtree:
            tree.set_positions(games.moves[active], games.lens[active])
        elif fresh.any():                                        # the others keep the subtree gmk_trad_step left them
            tree.set_positions(games.moves[active], np.where(fresh, games.lens[active], -1))
        fresh[:] = False
        if root_noise is not None:
            tree.add_root_noise(root_noise[0], root_noise[1], seed=seed, first_game_id=first_game_id)
        tree.run(playouts, stream)
        st = tree.root_stats()
        overflow |= bool((st["status"] & G.TraditionalMCTS.STATUS_ARENA_FULL).any())
        best = np.where(games.over[active], -1, st["best"])
        games.over[active[(best < 0) & ~games.over[active]]] = True
        played = np.full(n_games, -1, dtype=np.int64)
        played[active] = best
        at = games.lens.copy()
        moved = games.apply(played)
        slot_of = np.full(n_games, -1, dtype=np.int64)
        slot_of[active] = np.arange(slots)
        visits[moved, at[moved]] = np.minimum(st["visits"][slot_of[moved]], 65535)
        if reuse_subtree:
            tree.step(best.astype(np.int16))                     # finished games ask for -1 on a childless root: nothing moves
    tree.close()
    return GameRecords(torch.from_numpy(games.moves).to(dev), torch.from_numpy(games.lens).to(dev), torch.from_numpy(games.winner).to(dev),
                       torch.from_numpy(visits.view(np.int16)).to(dev), first_game_id, overflow)


def play_network_games(n_games, network, playouts, c_puct=5.0, seed=G.DEFAULT_SEED, first_game_id=0, opening_plies=0, max_moves=N,
                       device=None, node_capacity=None, reuse_subtree=True, root_noise=(0.05, 0.25), slots=None, device_loop=True, noise_sampler="counter"):
    """n_games complete games of the network-guided searcher against itself (agents/alphazero.py:5-9 on both sides: the
    reference's AlphaZero self-play), all games in lock step on the current GPU: every move = `playouts` lock-step playouts of
    K7 with `network(states [n,6,15,15]) -> (value [n], probs [n,225])` at the leaves (network.FusedPolicyValueNetwork = K9),
    then every game plays its most visited child (MCTSAgent.eval_state -> MCTS::stepForward); with reuse_subtree the child's
    subtree is kept and root_noise = (alpha, epsilon) is mixed into the root priors before every search (MCTS.cpp:182).
    slots: at most that many games in flight, a finished game hands its slot to the next one (see play_supervisor_games; on the
    host-driven loop with fresh roots only).  device_loop (all games at once): the move, the record, the end-of-game check and the re-rooting are one
    kernel per ply (gmk_az_advance) and the host sees four bytes per move; noise_sampler "counter" (default) draws the root noise on the device
    (az_root_noise_kernel, include/gomoku_noise.h), "std" on the host with std::gamma_distribution (priors down and up every ply); False = the host-driven loop it replaced (numpy boards, root statistics down and moves up every ply), kept for the
    tests that compare the two.  Returns GameRecords like play_games."""
    if slots is not None and slots < n_games and (max_moves < N or (device_loop and opening_plies > 8)):
        # (the slot loops play whole games, the device-resident one from openings of at most 8 plies -- gmk_az_set_slots' limit: say so instead of
        # ignoring the cap or failing inside the library)
        raise ValueError("play_network_games: with slots, games are played to their end (max_moves >= 225)" +
                         (" from openings of at most 8 plies on the device loop" if device_loop else "") + "; play all games at once (slots=None) for a move cap")
    G.init(torch.cuda.current_device() if device is None else device.index)
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else device
    if slots is not None and slots < n_games and device_loop:
        # continuous batching on the device (gmk_az_set_slots + gmk_az_advance): the batch the network sees stays full of live games
        # until fewer games than slots remain, with kept subtrees and root noise as well
        games = _HostGames(n_games)
        if opening_plies > 0:
            m, l, _ = G.synth_boards(n_games, 0, seed=seed, first_board=first_game_id)
            games.open_with(m, l, opening_plies)
        slots = int(slots)
        cap = node_capacity if node_capacity is not None else min((3 if reuse_subtree else 1) * playouts * N + 1, (1 << 24) - 1)
        tree = G.AlphaZeroMCTS(slots, node_capacity=cap, c_puct=c_puct)
        tree.set_option(G.OPT_NOISE_SAMPLER, G.NOISE_SAMPLERS[noise_sampler])
        tree.set_slots(n_games, games.moves[:, :max(int(games.lens.max()), 1)], games.lens)
        d_moves, d_lens = torch.from_numpy(games.moves).to(dev), torch.from_numpy(games.lens).to(dev)
        d_winner = torch.zeros(n_games, dtype=torch.int8, device=dev)
        d_visits = torch.zeros((n_games, N, N), dtype=torch.int16, device=dev)
        with torch.no_grad():
            for _ in range((n_games // slots + 2) * N):
                if root_noise is not None:
                    tree.add_root_noise(root_noise[0], root_noise[1], seed=seed, first_game_id=first_game_id)
                tree.search(network, playouts)
                if tree.advance(d_moves, d_visits, d_lens, d_winner, reuse_subtree) == 0:
                    break
        overflow = bool((tree.root_stats()["status"] & G.AlphaZeroMCTS.STATUS_ARENA_FULL).any())
        tree.close()
        return GameRecords(d_moves, d_lens, d_winner, d_visits, first_game_id, overflow)
    if slots is not None and slots < n_games:
        if reuse_subtree:
            raise ValueError("play_network_games: slots on the host-driven loop need fresh roots (reuse_subtree=False)")
        games = _HostGames(n_games)
        if opening_plies > 0:
            m, l, _ = G.synth_boards(n_games, 0, seed=seed, first_board=first_game_id)
            games.open_with(m, l, opening_plies)
        visits = np.zeros((n_games, N, N), dtype=np.uint16)
        slots = int(slots)
        tree = G.AlphaZeroMCTS(slots, node_capacity=node_capacity if node_capacity is not None else min(playouts * N + 1, (1 << 24) - 1), c_puct=c_puct)
        active, next_game, overflow = np.arange(slots), slots, False
        with torch.no_grad():
            for _ in range((n_games // slots + 2) * N):
                done = np.nonzero(games.over[active])[0]
                take = min(len(done), n_games - next_game)
                active[done[:take]] = np.arange(next_game, next_game + take)
                next_game += take
                if games.over[active].all():
                    break
                tree.set_game_ids(active)
                sub_moves, sub_lens = games.moves[active], games.lens[active]
                last = np.full((slots, 2), -1, dtype=np.int16)
                rows = np.arange(slots)
                last[sub_lens > 0, 0] = sub_moves[rows[sub_lens > 0], sub_lens[sub_lens > 0] - 1]
                last[sub_lens > 1, 1] = sub_moves[rows[sub_lens > 1], sub_lens[sub_lens > 1] - 2]
                tree.set_roots(G.moves_to_planes(sub_moves, sub_lens), last)
                tree.search(network, playouts)
                st = tree.root_stats()
                overflow |= bool((st["status"] & G.AlphaZeroMCTS.STATUS_ARENA_FULL).any())
                visited = st["visits"].max(1) > 0
                best = np.where(games.over[active] | ~visited, -1, st["visits"].argmax(1))
                games.over[active[(best < 0) & ~games.over[active]]] = True
                played = np.full(n_games, -1, dtype=np.int64)
                played[active] = best
                at = games.lens.copy()
                moved = games.apply(played)
                slot_of = np.full(n_games, -1, dtype=np.int64)
                slot_of[active] = np.arange(slots)
                visits[moved, at[moved]] = np.minimum(st["visits"][slot_of[moved]], 65535)
        tree.close()
        return GameRecords(torch.from_numpy(games.moves).to(dev), torch.from_numpy(games.lens).to(dev), torch.from_numpy(games.winner).to(dev),
                           torch.from_numpy(visits.view(np.int16)).to(dev), first_game_id, overflow)
    games = _HostGames(n_games)
    if opening_plies > 0:
        m, l, _ = G.synth_boards(n_games, 0, seed=seed, first_board=first_game_id)
        games.open_with(m, l, opening_plies)
    cap = node_capacity if node_capacity is not None else min((3 if reuse_subtree else 1) * playouts * N + 1, (1 << 24) - 1)
    tree = G.AlphaZeroMCTS(n_games, node_capacity=cap, c_puct=c_puct)
    tree.set_option(G.OPT_NOISE_SAMPLER, G.NOISE_SAMPLERS[noise_sampler])
    tree.set_roots(G.moves_to_planes(games.moves, games.lens), games.last_two())
    if device_loop:
        d_moves, d_lens = torch.from_numpy(games.moves).to(dev), torch.from_numpy(games.lens).to(dev)
        d_winner = torch.zeros(n_games, dtype=torch.int8, device=dev)
        d_visits = torch.zeros((n_games, N, N), dtype=torch.int16, device=dev)
        with torch.no_grad():
            for ply in range(max_moves):
                if root_noise is not None:
                    tree.add_root_noise(root_noise[0], root_noise[1], seed=seed, first_game_id=first_game_id)
                tree.search(network, playouts)
                if tree.advance(d_moves, d_visits, d_lens, d_winner, reuse_subtree) == 0:
                    break
        overflow = bool((tree.root_stats()["status"] & G.AlphaZeroMCTS.STATUS_ARENA_FULL).any())
        tree.close()
        return GameRecords(d_moves, d_lens, d_winner, d_visits, first_game_id, overflow)
    visits = np.zeros((n_games, N, N), dtype=np.uint16)
    overflow = False
    with torch.no_grad():
        for ply in range(max_moves):
            if games.over.all():
                break
            if ply > 0 and not reuse_subtree:
                tree.set_roots(G.moves_to_planes(games.moves, games.lens), games.last_two())
            if root_noise is not None:
                tree.add_root_noise(root_noise[0], root_noise[1], seed=seed, first_game_id=first_game_id)
            tree.search(network, playouts)
            st = tree.root_stats()
            overflow |= bool((st["status"] & G.AlphaZeroMCTS.STATUS_ARENA_FULL).any())
            visited = st["visits"].max(1) > 0                    # no child was visited: nothing to play (cannot happen on a live board)
            played = np.where(games.over | ~visited, -1, st["visits"].argmax(1)).astype(np.int16)      # max_element: the first maximum in child (= cell) order
            games.over |= played < 0
            at = games.lens.copy()
            moved = games.apply(played)
            visits[moved, at[moved]] = np.minimum(st["visits"][moved], 65535)
            if reuse_subtree:
                tree.step(played)
    tree.close()
    return GameRecords(torch.from_numpy(games.moves).to(dev), torch.from_numpy(games.lens).to(dev), torch.from_numpy(games.winner).to(dev),
                       torch.from_numpy(visits.view(np.int16)).to(dev), first_game_id, overflow)


class _Searcher:
    """One side of play_match_games: a batched searcher for a fixed set of games, fresh roots every move."""

    def __init__(self, spec, n, playouts, seed, first_game_id):
        kind, kw = spec
        self.kind, self.n, self.playouts = kind, n, int(kw.get("c_iterations", playouts))
        c_puct = float(kw.get("c_puct", 5.0))
        self.first_game_id = first_game_id
        if kind == "traditional_mcts":                       # AGENT_MAP names of the reference (agents/__init__.py, config.py:9-19)
            self.tree = G.TraditionalMCTS(n, node_capacity=min(self.playouts * 226 + 256, (1 << 24) - 1), c_puct=c_puct)
        elif kind == "rave_mcts":
            self.tree = G.PoolRAVEMCTS(n, node_capacity=min(self.playouts * 226 + 256, (1 << 24) - 1), c_puct=c_puct, seed=seed, first_game_id=first_game_id)
        elif kind == "random_mcts":
            self.tree = G.BatchedMCTS(n, playouts_capacity=self.playouts, c_puct=c_puct, c_rollouts=int(kw.get("c_rollouts", 5)), seed=seed)
        else:
            raise ValueError("play_match_games: unknown agent '%s' (traditional_mcts, rave_mcts, random_mcts)" % kind)

    def search(self, moves, lens, games, stream):
        """Root visit counts [n, 225] and the move MCTS::stepForward() would make per game (-1: none), from fresh roots.
        games int[n]: the number of the game in each slot; its random streams are keyed by first_game_id + that number."""
        if self.kind == "random_mcts":
            planes = G.moves_to_planes(moves, lens)
            last = np.where(lens > 0, moves[np.arange(self.n), np.maximum(lens, 1) - 1].astype(np.int16), np.int16(-1))
            self.tree.set_roots(planes, last, self.first_game_id)
            self.tree.set_game_ids((self.first_game_id + np.asarray(games)).astype(np.uint32))
            self.tree.run(self.playouts, stream)
            stats = self.tree.root_stats()
            visits = stats[0]
            best = np.where(visits.max(1) > 0, visits.argmax(1), -1).astype(np.int32)        # max_element: the first maximum in child (= cell) order
            return visits, best, bool((stats[4] & G.BatchedMCTS.STATUS_ARENA_FULL).any())
        self.tree.set_game_ids(np.asarray(games, dtype=np.uint32))
        self.tree.set_positions(moves, lens)
        self.tree.run(self.playouts, stream)
        st = self.tree.root_stats()
        return st["visits"], st["best"], bool((st["status"] & G.TraditionalMCTS.STATUS_ARENA_FULL).any())

    def close(self):
        self.tree.close()


def play_match_games(n_games, supervisor, candidate, playouts=400, seed=G.DEFAULT_SEED, first_game_id=0, opening_plies=0,
                     max_moves=N, device=None, slots=None):
    """The reference's data generation pairing (network/data_helper.py:15-22, 56-63; config.py:6-20): every game is played by
    the SUPERVISOR against a CANDIDATE, sides drawn at random per game, and both players' searches are recorded.  supervisor /
    candidate = (name, kwargs) as in DATA_CONFIG["schedule"]: ("traditional_mcts" | "rave_mcts" | "random_mcts", {"c_puct": ..,
    "c_iterations": .., "c_rollouts": ..}).  All games run side by side on the current GPU: the games in which the supervisor
    has black and the games in which it has white form two groups, each with one batched searcher per agent (K6 / K8 / K3), so
    that every ply is two searches (one per group) covering all unfinished games; roots are fresh at every move and every
    game's random streams are keyed by its own global id first_game_id + g, whatever group and slot it runs in.  slots: games in flight per group (default: all of them); a
    finished game hands its slot to the next unstarted game of its group, which keeps the searches full (see play_supervisor_games).
    Returns (GameRecords, supervisor_is_black bool[n]); the records' visit counts at move i are those of the player who made it."""
    G.init(torch.cuda.current_device() if device is None else device.index)
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else device
    stream = torch.cuda.current_stream(dev).cuda_stream
    rng = np.random.RandomState(int((seed ^ (seed >> 32) ^ first_game_id) & 0xFFFFFFFF))
    sup_black = rng.randint(0, 2, n_games).astype(bool)           # random.shuffle(players) per game (data_helper.py:57-58)
    games = _HostGames(n_games)
    if opening_plies > 0:
        m, l, _ = G.synth_boards(n_games, 0, seed=seed, first_board=first_game_id)
        games.open_with(m, l, opening_plies)
    visits = np.zeros((n_games, N, N), dtype=np.uint16)
    groups = [np.nonzero(sup_black)[0], np.nonzero(~sup_black)[0]]
    active, waiting, searchers = [], [], []                      # per group: the game in each slot, the games not started yet, [supervisor, candidate]
    for idx in groups:
        k = len(idx) if slots is None else min(int(slots), len(idx))
        active.append(idx[:k].copy())
        waiting.append(list(idx[k:]))
        searchers.append([_Searcher(spec, k, playouts, seed, first_game_id) if k else None for spec in (supervisor, candidate)])
    overflow = False
    for _ in range((2 + (0 if slots is None else n_games // max(1, int(slots)))) * max_moves):
        if games.over.all():
            break
        for gi in (0, 1):
            act = active[gi]
            if len(act) == 0:
                continue
            done = np.nonzero(games.over[act])[0]                 # a finished game hands its slot to the next one of the group
            for slot in done[:len(waiting[gi])]:
                act[slot] = waiting[gi].pop(0)
            if games.over[act].all():
                continue
            for who in (0, 1):                                    # normally one of the two: the games of a group move in lock step
                # group 0: the supervisor has black, i.e. it moves on even stone counts
                turn = ((games.lens[act] % 2 == 0) == (gi == 0)) == (who == 0)
                sel = turn & ~games.over[act]
                if not sel.any():
                    continue
                v, best, full = searchers[gi][who].search(games.moves[act], games.lens[act], act, stream)
                overflow |= full
                played = np.full(n_games, -1, dtype=np.int64)
                played[act[sel]] = best[sel]
                games.over[act[sel & (best < 0)]] = True
                at = games.lens.copy()
                moved = games.apply(played)
                slot_of = np.full(n_games, -1, dtype=np.int64)
                slot_of[act] = np.arange(len(act))
                visits[moved, at[moved]] = np.minimum(v[slot_of[moved]], 65535)
    for pair in searchers:
        for srch in pair:
            if srch is not None:
                srch.close()
    rec = GameRecords(torch.from_numpy(games.moves).to(dev), torch.from_numpy(games.lens).to(dev), torch.from_numpy(games.winner).to(dev),
                      torch.from_numpy(visits.view(np.int16)).to(dev), first_game_id, overflow)
    return rec, sup_black


def dump_batches(samples, path, batch_size=512):
    """The reference appends mini-batches to `latest.train.hdf5` as three growing datasets `state_batch`, `value_batch`,
    `probs_batch` of shape (batches, batch_size, ...) (network/data_helper.py:176-194).  h5py is written when it can be
    imported; otherwise the same three arrays go to `<path>.npz` (numpy's archive), with the same names and shapes.
    `samples` = (states, values, pi) as GameRecords.to_samples() returns them; a trailing partial batch is dropped like
    the reference's generator does.  Returns the number of batches written."""
    states, values, pi = (t.cpu().numpy() if hasattr(t, "cpu") else np.asarray(t) for t in samples)
    n_batches = states.shape[0] // batch_size
    if n_batches == 0:
        return 0
    cut = n_batches * batch_size
    arrays = {"state_batch": states[:cut].reshape(n_batches, batch_size, *states.shape[1:]).astype(np.float32),
              "value_batch": values[:cut].reshape(n_batches, batch_size).astype(np.float32),
              "probs_batch": pi[:cut].reshape(n_batches, batch_size, pi.shape[1]).astype(np.float32)}
    try:
        import h5py
    except ImportError:
        h5py = None
    if h5py is not None and not path.endswith(".npz"):
        with h5py.File(path, "a") as hf:
            for name, arr in arrays.items():
                if name not in hf:
                    hf.create_dataset(name, (0, *arr.shape[1:]), maxshape=(None, *arr.shape[1:]))
                hf[name].resize(hf[name].shape[0] + arr.shape[0], axis=0)
                hf[name][-arr.shape[0]:] = arr
    else:
        path = path if path.endswith(".npz") else path + ".npz"
        if os.path.exists(path):
            with np.load(path) as old:
                arrays = {name: np.concatenate([old[name], arr]) for name, arr in arrays.items()}
        np.savez(path, **arrays)
    return n_batches


def pack_records(rec):
    """The wire form of a rank's records, ONE uint8 tensor on the records' device: lens int32[n], winner int8[n],
    then only what was played: moves uint8[sum(lens)] and, if visits were recorded, uint16[sum(lens)][225] (the root visit counts
    of the searched plies; plies that were not searched -- openings -- carry their zero rows).  ~27 KB per game instead of the
    fixed-stride 101 KB."""
    lens = rec.lens.to(torch.int32).contiguous()
    n = int(lens.shape[0])
    if n == 0:
        return torch.zeros(0, dtype=torch.uint8, device=lens.device)
    played = torch.arange(N, device=lens.device)[None, :] < lens[:, None]            # [n, 225]
    parts = [lens.view(torch.uint8).reshape(-1), rec.winner.to(torch.int8).contiguous().view(torch.uint8).reshape(-1),
             rec.moves.contiguous().view(torch.uint8).reshape(n, N)[played]]
    if rec.visits is not None:
        parts.append(rec.visits.contiguous().view(torch.int16).reshape(n, N, N)[played].contiguous().view(torch.uint8).reshape(-1))
    return torch.cat(parts)


def unpack_records_into(buf, n, has_visits, moves, lens, winner, visits, at):
    """pack_records undone into rows [at, at + n) of preallocated fixed-stride arrays (rows beyond a game's length must be zero already)."""
    dev = buf.device
    lens[at:at + n] = buf[:4 * n].view(torch.int32)
    winner[at:at + n] = buf[4 * n:5 * n].view(torch.int8)
    my_lens = lens[at:at + n]
    total = int(my_lens.sum()) if n else 0
    played = torch.arange(N, device=dev)[None, :] < my_lens[:, None]
    moves[at:at + n][played] = buf[5 * n:5 * n + total]
    if has_visits and visits is not None:
        flat = buf[5 * n + total:5 * n + total + total * N * 2]
        if (5 * n + total) % 2:                                  # int16 views need an even byte offset
            flat = flat.clone()
        visits[at:at + n][played] = flat.view(torch.int16).reshape(total, N)


def unpack_records(buf, n, has_visits, first_game_id=0, overflow=False):
    """pack_records undone: fixed-stride GameRecords on buf's device (what to_samples() / K4 + K5 read)."""
    dev = buf.device
    moves = torch.zeros((n, N), dtype=torch.uint8, device=dev)
    lens = torch.zeros((n,), dtype=torch.int32, device=dev)
    winner = torch.zeros((n,), dtype=torch.int8, device=dev)
    visits = torch.zeros((n, N, N), dtype=torch.int16, device=dev) if has_visits else None
    unpack_records_into(buf, n, has_visits, moves, lens, winner, visits, 0)
    return GameRecords(moves, lens, winner, visits, first_game_id, overflow)


class GatherError(RuntimeError):
    """Raised on EVERY rank when any rank could not contribute to gather_records."""


def gather_records(rec, dst=0, group=None):
    """The one exchange step of the path (network/data_helper.py:97-113: the worker processes' `Manager().list()`): every rank
    contributes its records in the compact wire form of pack_records, rank `dst` receives them with one grouped batch of
    point-to-point transfers -- every other rank posts ONE send, dst posts world - 1 receives (RCCL has no native gather; over xGMI
    the 7 peers send concurrently on their own links) -- and unpacks the concatenation in rank order (= global game id order).
    The same code runs on every torch.distributed backend (RCCL on the GPU box, gloo in the CPU tests); with world size 1 it is the
    identity.  Returns the gathered GameRecords on dst, None elsewhere."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return rec
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    dev = rec.lens.device
    # Packing is local work that can fail on one rank alone (memory): the ranks agree on it BEFORE the first collective, so that a failure
    # is raised on every rank (GatherError) instead of leaving the others waiting in all_gather for a rank that has left.
    mine, failure = None, None
    try:
        mine = pack_records(rec)
    except Exception as exc:                                    # noqa: BLE001
        failure = exc
    ok = torch.tensor([0 if failure is not None else 1], dtype=torch.int32, device=dev)
    dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=group)
    if int(ok[0]) == 0:
        raise GatherError("gather_records: packing failed on %s" % ("this rank: %s: %s" % (type(failure).__name__, failure) if failure is not None else "another rank"))
    # sizes first: [games, bytes, first game id, arena overflow seen, visits recorded]
    meta = torch.tensor([len(rec), int(mine.numel()), int(rec.first_game_id), int(bool(rec.overflow)), int(rec.visits is not None)], dtype=torch.int64, device=dev)
    metas = [torch.zeros_like(meta) for _ in range(world)]
    dist.all_gather(metas, meta, group=group)
    metas = [[int(v) for v in m] for m in metas]
    if rank == dst:
        bufs = [mine if r == dst else torch.empty(metas[r][1], dtype=torch.uint8, device=dev) for r in range(world)]
        ops = [dist.P2POp(dist.irecv, bufs[r], r, group) for r in range(world) if r != dst and metas[r][1] > 0]
    else:
        ops = [dist.P2POp(dist.isend, mine, dst, group)] if mine.numel() > 0 else []
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()
    if rank != dst:
        return None
    has_visits = all(m[4] for m in metas if m[0] > 0) and any(m[0] > 0 for m in metas)
    overflow = any(m[3] for m in metas)                          # a rank whose arenas overflowed must not be reported as clean
    first = min((m[2] for m in metas if m[0] > 0), default=rec.first_game_id)
    n_all = sum(m[0] for m in metas)
    if n_all == 0:
        return GameRecords(rec.moves[:0], rec.lens[:0], rec.winner[:0], None if rec.visits is None else rec.visits[:0], first, overflow)
    # The fixed-stride form is allocated ONCE for all ranks and every rank's bytes are unpacked into their rows, the wire buffer of a rank
    # freed as soon as it has been read: 32 768 games are 3.3 GB of visit rows, and per-rank parts plus a concatenation would hold them twice.
    moves = torch.zeros((n_all, N), dtype=torch.uint8, device=dev)
    lens = torch.zeros((n_all,), dtype=torch.int32, device=dev)
    winner = torch.zeros((n_all,), dtype=torch.int8, device=dev)
    visits = torch.zeros((n_all, N, N), dtype=torch.int16, device=dev) if has_visits else None
    at = 0
    for r in range(world):                                       # rank order = global game id order
        if metas[r][0] > 0:
            unpack_records_into(bufs[r], metas[r][0], bool(metas[r][4]), moves, lens, winner, visits, at)
            at += metas[r][0]
        bufs[r] = None
    del mine
    return GameRecords(moves, lens, winner, visits, first, overflow)


def shard(n_total, rank, world):
    """Contiguous id blocks: game g lives on rank g * world // n_total."""
    lo = (n_total * rank) // world
    hi = (n_total * (rank + 1)) // world
    return lo, hi - lo
