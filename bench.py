#!/usr/bin/env python3
"""Benchmark of the MI355X hot path.  Contract: python bench.py --gpus N --steps K --warmup W prints ONE
JSON line on rank 0.  A step = one pass of K1 (gmk_eval_batch) over one resident batch of synthetic
boards (BASELINE.json configs[1]: 65 536 random 15x15 boards per GPU)."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ALG_BYTES_PER_EVAL = 64 + 3600 + 3600 + 44 + 4      # planes in; scores, density, totals, status out
HBM_PEAK_GBS = 8000.0                               # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def measured_traffic(kernel, units_key, units):
    """HBM bytes per launch from the committed rocprofv3 PMC passes (profiles/traffic.json), scaled to this launch."""
    try:
        t = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))[kernel]
        return t["hbm_bytes_per_launch"] * units / t[units_key]
    except Exception:
        return None


REDUCE_DEVICE = None          # torch device the cross-rank reductions run on: the GPU with RCCL, the CPU in the gloo rehearsal mode


class LegFailed(RuntimeError):
    """A leg of the bench failed on some rank; raised on every rank (see all_ranks_ok)."""


def all_ranks_ok(ok, torch, dev, distributed):
    """Every rank learns whether local work succeeded everywhere (all-reduce MIN of a flag): ranks that sit in one another's barriers
    must take the same branch, or the ones that carried on wait for the one that left until the process group's timeout."""
    if not distributed:
        return bool(ok)
    flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=REDUCE_DEVICE or dev)
    torch.distributed.all_reduce(flag, op=torch.distributed.ReduceOp.MIN)
    return bool(int(flag[0]))


def local_stage(what, fn, torch, dev, distributed):
    """Runs local (collective-free) work of a leg, then lets the ranks agree on its outcome; raises LegFailed on all of them if one failed."""
    result, failure = None, None
    try:
        result = fn()
    except Exception as exc:                                    # noqa: BLE001
        failure = "%s: %s" % (type(exc).__name__, exc)
    if not all_ranks_ok(failure is None, torch, dev, distributed):
        raise LegFailed("%s failed on %s" % (what, "this rank (%s)" % failure if failure else "another rank"))
    return result


def measured_counter(kernel, key):
    """A derived figure of the committed rocprofv3 PMC passes (profiles/traffic.json), e.g. the share of VALU lanes that were active."""
    try:
        return json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))[kernel].get(key)
    except Exception:
        return None


def cpu_baseline(n_sample, kind):
    """The CPU oracle (restatement of the reference's incremental evaluator: replay of each move list)
    timed single-threaded on this host, on a bounded sample of the same workload."""
    from gomokuai_amd import lib as G
    from oracle import oracle as O
    moves, lens, _ = G.synth_boards(n_sample, kind)
    O.lib()
    t0 = time.perf_counter()
    O.replay_batch(moves, lens)
    dt = time.perf_counter() - t0
    out = {"value": n_sample / dt, "unit": "board-evals/s", "cores": 1, "kind": "port",
           "sample": "%d boards of the same synthetic set, in-order replay through the oracle evaluator, %.1f s" % (n_sample, dt)}
    # the same port with one thread per host core available to this process (ctypes drops the GIL)
    from concurrent.futures import ThreadPoolExecutor
    cores = min(len(os.sched_getaffinity(0)), 16)          # the GPU box gives one GPU a 16-core share
    rounds = 4
    cuts = [n_sample * i // cores for i in range(cores + 1)]
    t0 = time.perf_counter()
    with ThreadPoolExecutor(cores) as pool:
        for _ in range(rounds):
            list(pool.map(lambda i: O.replay_batch(moves[cuts[i]:cuts[i + 1]], lens[cuts[i]:cuts[i + 1]]), range(cores)))
    dt = time.perf_counter() - t0
    out["all_cores"] = {"value": rounds * n_sample / dt, "unit": "board-evals/s", "cores": cores,
                        "sample": "%d x the same %d boards split over %d threads, %.1f s" % (rounds, n_sample, cores, dt)}
    return out


def mcts_openings(G, np, n, first):
    """BASELINE configs[2]: every game starts from a 4-ply random opening of the synthetic generator."""
    moves, lens, _ = G.synth_boards(n, 0, first_board=first)
    lens = np.minimum(lens, 4).astype(np.int32)
    planes = G.moves_to_planes(moves, lens)
    last = np.array([moves[i, lens[i] - 1] for i in range(n)], dtype=np.int16)
    return moves, lens, planes, last


def bench_mcts(args, G, torch, dev, rank, world, distributed):
    """Secondary metric of BASELINE.json: MCTS playouts/s, configs[2] (4 096 games x 800 playouts per GPU,
    RandomPolicy c_puct=5 c_rollouts=5, fresh roots).  One step = one whole search of all games (one launch)."""
    import numpy as np
    n, P = args.mcts_games, args.mcts_playouts
    _, _, planes, last = mcts_openings(G, np, n, rank * n)
    tree = G.BatchedMCTS(n, playouts_capacity=P)
    stream = torch.cuda.current_stream().cuda_stream
    tree.set_roots(planes, last, first_game_id=rank * n)
    tree.run(P, stream)                                   # warm-up search
    torch.cuda.synchronize()
    times = []
    for _ in range(args.mcts_reps):
        tree.set_roots(planes, last, first_game_id=rank * n)
        torch.cuda.synchronize()
        if distributed:
            torch.distributed.barrier()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        tree.run(P, stream)
        e1.record()
        torch.cuda.synchronize()
        times.append(e0.elapsed_time(e1))
    ms = sorted(times)[len(times) // 2]
    alg = tree.alg_bytes()
    if distributed:
        t = torch.tensor([ms], dtype=torch.float64, device=REDUCE_DEVICE or dev)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        ms = float(t[0])
    tree.close()
    saturated = None
    if args.mcts_saturated_games > n:
        # configs[2]'s 4 096 games are one wavefront per SIMD (20 480 rollout lanes of 65 536): the same search with the chip full
        ns = args.mcts_saturated_games
        _, _, planes_s, last_s = mcts_openings(G, np, ns, rank * ns)
        big = G.BatchedMCTS(ns, playouts_capacity=P)
        big.set_roots(planes_s, last_s, first_game_id=rank * ns)
        torch.cuda.synchronize()
        if distributed:
            torch.distributed.barrier()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        big.run(P, stream)
        e1.record()
        torch.cuda.synchronize()
        ms_s = e0.elapsed_time(e1)
        if distributed:
            t = torch.tensor([ms_s], dtype=torch.float64, device=REDUCE_DEVICE or dev)
            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
            ms_s = float(t[0])
        big.close()
        saturated = {"games_per_gpu": ns, "value": ns * world * P / (ms_s * 1e-3), "unit": "playouts/s", "ms_per_search": ms_s,
                     "note": "the same search with %d games per GPU (4 wavefronts per SIMD): what the chip sustains when the batch fills it" % ns}
    achieved = alg / (ms * 1e-3) / 1e9
    return {"metric": "mcts-playouts/s", "value": n * world * P / (ms * 1e-3), "unit": "playouts/s", "ms_per_search": ms, "saturated": saturated,
            "lane_utilisation": measured_counter("mcts_playouts_kernel", "valu_lane_utilisation"),
            "lane_utilisation_source": "profiles/traffic.json (committed rocprofv3 PMC passes; not re-measured in this run)",
            "config": {"workload": "batched MCTS (K3), %d games x %d playouts per GPU, RandomPolicy c_puct=5 c_rollouts=5, 4-ply openings, fresh roots" % (n, P)},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": measured_traffic("mcts_playouts_kernel", "playouts_per_launch", n * P),
                         "traffic_source": "profiles/traffic.json (committed rocprofv3 PMC passes; not re-measured in this run)",
                         "kernel": "mcts_playouts_kernel", "kernel_ms": ms, "alg_bytes_per_launch": alg,
                         "note": "tree bytes only (select 8 B/child, expand 16 B/node, backup 16 B/level); rollouts run in LDS/registers, the kernel is latency/issue bound"}}


def bench_evalstate(args, G, torch, dev, rank, world, distributed):
    """K2: the incrementally maintained evaluator (Evaluator::applyMove / revertMove, Pattern.cpp:306-342), which bounds K6: n games
    side by side, each applies the moves of its clustered synthetic game (<= 60) and takes them back again, one launch each way."""
    import numpy as np
    n, k = args.evalstate_games, 60
    moves, lens, _ = G.synth_boards(n, 1, first_board=rank * n)
    script = np.full((n, k), -1, np.int16)
    for g in range(n):
        m = min(int(lens[g]), k)
        script[g, :m] = moves[g, :m]
    d_apply = torch.from_numpy(script).to(dev)
    d_revert = torch.full((n, k), -2, dtype=torch.int16, device=dev)
    d_revert[torch.from_numpy(script < 0).to(dev)] = -1
    states = G.EvaluatorStates(n)
    L = G.load()
    stream = torch.cuda.current_stream().cuda_stream
    def both():
        L.gmk_evalstate_update(states.h, d_apply.data_ptr(), k, stream)
        L.gmk_evalstate_update(states.h, d_revert.data_ptr(), k, stream)
    both()
    torch.cuda.synchronize()
    if distributed:
        torch.distributed.barrier()
    reps = 20
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        both()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    if distributed:
        t = torch.tensor([ms], dtype=torch.float64, device=REDUCE_DEVICE or dev)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        ms = float(t[0])
    meta = states.read()["meta"]
    states.close()
    updates = 2 * int((script >= 0).sum())
    return {"metric": "evaluator-updates/s", "value": updates * world / (ms * 1e-3), "unit": "updates/s", "ms_per_pair_of_launches": ms,
            "us_per_update_per_game": ms * 1e3 / (updates / n),
            "config": {"workload": "incremental evaluator states (K2), %d games per GPU, each applies its clustered synthetic game (<= 60 moves, %.1f on average) and reverts it" % (n, updates / 2 / n),
                       "states_back_at_the_empty_board": bool((meta[:, 0] == 0).all())},
            "note": "one wavefront per game, the 17.8 KB state in LDS: a chain of dependent LDS round trips per update, bound by latency, not by bandwidth; no roofline fraction is claimed"}


def cpu_baseline_evalstate(G, n_sample):
    """The oracle's evaluator applying the same games move by move (in-order replay), single thread."""
    from oracle import oracle as O
    moves, lens, _ = G.synth_boards(n_sample, 1)
    O.lib()
    t0 = time.perf_counter()
    O.replay_batch(moves, lens)
    dt = time.perf_counter() - t0
    return {"value": float(lens.sum()) / dt, "unit": "updates/s", "cores": 1, "kind": "port",
            "sample": "%d clustered synthetic games (%d moves) applied move by move through the oracle evaluator, %.1f s" % (n_sample, int(lens.sum()), dt)}


def trad_positions(G, n, first):
    """K6 workload: every game starts from a 12-ply clustered opening of the synthetic generator (threats and compounds present)."""
    moves, lens, _ = G.synth_boards(n, 1, first_board=first)
    return [[int(m) for m in moves[g, :min(int(lens[g]), 12)]] for g in range(n)]


def bench_trad(args, G, torch, dev, rank, world, distributed):
    """the self-play supervisor, MCTS(TraditionalPolicy) (config.py:9-12), n games side by side (K6).
    One step = one search of every game (one launch)."""
    import numpy as np
    n, P = args.trad_games, args.trad_playouts
    pos = trad_positions(G, n, rank * n)
    tree = G.TraditionalMCTS(n, node_capacity=args.trad_nodes, c_puct=5.0)
    stream = torch.cuda.current_stream().cuda_stream
    tree.set_positions(pos)
    tree.run(20, stream)                                   # warm-up
    torch.cuda.synchronize()
    times = []
    for _ in range(2):
        updates_before = tree.root_stats()["evaluator_updates"].astype(np.float64)
        tree.reset_evaluators()
        tree.set_positions(pos)
        torch.cuda.synchronize()
        if distributed:
            torch.distributed.barrier()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        tree.run(P, stream)
        e1.record()
        torch.cuda.synchronize()
        times.append(e0.elapsed_time(e1))
    ms = min(times)
    st = tree.root_stats()
    if distributed:
        t = torch.tensor([ms], dtype=torch.float64, device=REDUCE_DEVICE or dev)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        ms = float(t[0])
    tree.close()
    saturated = None
    if args.trad_saturated_games > n:
        # 2 048 games are exactly the wavefronts the chip holds (8 per CU: the evaluator states fill the LDS), so that launch lasts as long as
        # its slowest game; with four times the games the workgroups that finish early are replaced: the search rate without that wait
        ns = args.trad_saturated_games
        big = G.TraditionalMCTS(ns, node_capacity=args.trad_nodes, c_puct=5.0)
        big.set_positions(trad_positions(G, ns, rank * ns))
        torch.cuda.synchronize()
        if distributed:
            torch.distributed.barrier()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        big.run(P, stream)
        e1.record()
        torch.cuda.synchronize()
        ms_s = e0.elapsed_time(e1)
        if distributed:
            t = torch.tensor([ms_s], dtype=torch.float64, device=REDUCE_DEVICE or dev)
            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
            ms_s = float(t[0])
        big.close()
        saturated = {"value": ns * world * P / (ms_s * 1e-3), "unit": "playouts/s", "ms_per_search": ms_s, "games_per_gpu": ns,
                     "note": "four workgroups queued per CU: a finished game's wavefront slot is refilled by the dispatcher"}
    return {"metric": "supervisor-playouts/s", "value": n * world * P / (ms * 1e-3), "unit": "playouts/s", "ms_per_search": ms, "saturated": saturated,
            "config": {"workload": "pattern-guided MCTS (K6, TraditionalPolicy c_puct=5), %d games x %d playouts per GPU, 12-ply clustered openings, fresh roots" % (n, P),
                       "nodes_per_game_mean": float(st["n_nodes"].mean()), "evaluator_updates_per_playout": float((st["evaluator_updates"].astype(np.float64) - updates_before).mean()) / P,
                       "games_stopped_at_node_capacity": int((st["status"] & 1).sum())},
            "note": "a serial chain per game (one wavefront each): bound by LDS / HBM latency, not by bandwidth; no roofline fraction is claimed"}


def bench_supervisor_pipeline(args, G, torch, dev, rank, world, distributed, bare_playouts_per_s):
    """The supervisor's self-play loop (network/data_helper.py:56-83 with config.py:9-12's "traditional_mcts" on both sides) resident on
    the device: --sup-games games per GPU through --trad-games slots (gmk_trad_selfplay_run, persistent: ONE launch in which every slot's
    wavefront plays game after game at its own pace), every move one K6 search of --trad-playouts playouts -- from a new root every move, and
    (reference_semantics) as the reference's agent plays them: the chosen child's subtree kept (MCTS.cpp:129-147) and Default::AddNoise(0.05, 0.25)
    before every search (MCTS.cpp:179-183), both inside the same one launch.  Weak scaling (games per GPU fixed); the time is the slowest
    rank's.  Beside it the lock-step form of the loop (search by search for all slots) with every slot busy."""
    import time
    from gomokuai_amd import selfplay
    n, slots, P = args.sup_games, args.trad_games, args.trad_playouts
    def run(what, **kw):
        # (the arenas first, untimed: see selfplay_leg)
        local_stage(what + " (arenas)", lambda: selfplay.play_supervisor_games(first_game_id=rank * n, opening_plies=2, slots=slots, prepare_only=True, **kw), torch, dev, distributed)
        def play():
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            r = selfplay.play_supervisor_games(first_game_id=rank * n, opening_plies=2, slots=slots, **kw)
            torch.cuda.synchronize()
            return r, time.perf_counter() - t0
        if distributed:
            torch.distributed.barrier()
        rec, seconds = local_stage(what, play, torch, dev, distributed)
        searched = int(rec.lens.sum()) - int(torch.clamp(rec.lens, max=2).sum())      # searched plies (the two opening plies are given)
        (seconds,), (searched, overflow) = reduce_leg(torch, dev, distributed, [seconds], [searched, int(bool(rec.overflow))])
        return seconds, searched, bool(overflow)
    seconds, moves, overflow = run("supervisor self-play", n_games=n, playouts=P, node_capacity=args.trad_nodes)
    rate = moves * P / seconds
    # the same games as the reference's agent plays them (a kept subtree is searched on top of: three times the nodes, in each of a slot's two arenas)
    ref_seconds, ref_moves, ref_overflow = run("supervisor self-play (reference semantics)", n_games=n, playouts=P, node_capacity=min(3 * args.trad_nodes, (1 << 24) - 1),
                                               reuse_subtree=True, root_noise=REFERENCE_NOISE)
    ref_rate = ref_moves * P / ref_seconds
    # ... and the lock-step form of the loop (one search launch per move for all slots, then the step kernels; what host-drawn root noise
    # and PoolRAVE use) while every slot is busy (eight games queued per slot, stopped after 40 moves per slot): what that loop's step
    # kernels, slot hand-over and four bytes to the host per move take from the bare search rate
    steady_steps = 40
    seconds2, moves2, _ = run("supervisor self-play (busy slots)", n_games=8 * slots, playouts=P, node_capacity=args.trad_nodes, max_steps=steady_steps)
    busy_rate = moves2 * P / seconds2
    return {"metric": "supervisor self-play playouts/s", "value": rate, "unit": "playouts/s", "seconds": seconds, "games": n * world, "searched_moves": moves,
            "games_per_s": n * world / seconds, "arena_overflow": overflow,
            "reference_semantics": {"value": ref_rate, "unit": "playouts/s", "seconds": ref_seconds, "searched_moves": ref_moves, "games_per_s": n * world / ref_seconds,
                                    "arena_overflow": ref_overflow, "share_of_new_root_playouts_per_s": ref_rate / rate,
                                    "config": {"workload": "the same %d games per GPU through %d slots in ONE persistent launch, as the reference's agent plays them: kept subtree (MCTS.cpp:129-147) + "
                                                           "Default::AddNoise(0.05, 0.25) before every search (MCTS.cpp:179-183), drawn by the searching wavefront from the counter-based "
                                                           "sampler (include/gomoku_noise.h)" % (n, slots)}},
            "busy_slots": {"value": busy_rate, "unit": "playouts/s", "steps": steady_steps, "searched_moves": moves2, "seconds": seconds2,
                           "share_of_bare_search_rate": busy_rate / bare_playouts_per_s if bare_playouts_per_s else None,
                           "note": "includes the setup of %d queued games (records, openings) and the first search's evaluator syncs" % (8 * slots)},
            "share_of_bare_search_rate": rate / bare_playouts_per_s if bare_playouts_per_s else None,
            "config": {"workload": "supervisor self-play (K6, persistent device-resident game loop: gmk_trad_selfplay_run in ONE launch), %d games per GPU through %d slots, "
                                   "%d playouts per move, 2-ply openings, fresh root every move" % (n, slots, P),
                       "host_traffic": "none while the games run (records by game id in HBM); busy_slots: the lock-step form of the loop, 4 bytes per move"}}


def rave_positions(G, np, n, first):
    """K8 workload: the 4-ply random openings of the K3 measurement."""
    moves, lens, _, _ = mcts_openings(G, np, n, first)
    return [[int(m) for m in moves[g, :int(lens[g])]] for g in range(n)]


def bench_rave(args, G, torch, dev, rank, world, distributed):
    """MCTS(PoolRAVEPolicy) (agents/mcts.py:36-40, PoolRAVE.h:7-52), n games side by side (K8).  One step = one search of
    every game (one launch)."""
    import numpy as np
    n, P = args.rave_games, args.rave_playouts
    pos = rave_positions(G, np, n, rank * n)
    tree = G.PoolRAVEMCTS(n, node_capacity=P * 222 + 512, c_puct=2.0, first_game_id=rank * n)
    stream = torch.cuda.current_stream().cuda_stream
    tree.set_positions(pos)
    tree.run(10, stream)                                   # warm-up
    torch.cuda.synchronize()
    times = []
    for _ in range(3):
        tree.set_positions(pos)
        torch.cuda.synchronize()
        if distributed:
            torch.distributed.barrier()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        tree.run(P, stream)
        e1.record()
        torch.cuda.synchronize()
        times.append(e0.elapsed_time(e1))
    ms = min(times)
    st = tree.root_stats()
    if distributed:
        t = torch.tensor([ms], dtype=torch.float64, device=REDUCE_DEVICE or dev)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        ms = float(t[0])
    tree.close()
    return {"metric": "poolrave-playouts/s", "value": n * world * P / (ms * 1e-3), "unit": "playouts/s", "ms_per_search": ms,
            "config": {"workload": "PoolRAVE MCTS (K8, PoolRAVEPolicy c_puct=2), %d games x %d playouts per GPU, 4-ply openings, fresh roots" % (n, P),
                       "nodes_per_game_mean": float(st["n_nodes"].mean()), "games_stopped_at_node_capacity": int((st["status"] & 1).sum())},
            "note": "a serial chain per game (tree walk, one random rollout, per-level reductions): bound by latency, not by bandwidth; no roofline fraction is claimed"}


def cpu_baseline_rave(G, playouts):
    """The oracle's restatement of the same search (oracle/go_rave.c), single thread, ~5 s."""
    import numpy as np
    from oracle import oracle as O
    pos = rave_positions(G, np, 2048, 0)
    t0 = time.perf_counter()
    k = 0
    while time.perf_counter() - t0 < 5 and k < len(pos):
        O.PoolRAVEMCTS(2.0, 0.0, game_id=k).run(pos[k], playouts)
        k += 1
    dt = time.perf_counter() - t0
    return {"value": k * playouts / dt, "unit": "playouts/s", "cores": 1, "kind": "port",
            "sample": "%d searches of %d playouts from the same openings, oracle PoolRAVEPolicy restatement, %.1f s" % (k, playouts, dt)}


def bench_az(args, G, torch, dev, rank, world, distributed):
    """BASELINE configs[4]: network-guided MCTS (K7) in lock step, PolicyValueNetwork (PyTorch-ROCm, float32, random weights)
    at the leaves.  One step = one playout of every game = select kernel + network forward + expand kernel."""
    import numpy as np
    from gomokuai_amd.network import FusedPolicyValueNetwork, PolicyValueNetwork
    n, P = args.az_games, args.az_playouts
    moves, lens, planes, _ = mcts_openings(G, np, n, rank * n)
    last = np.stack([moves[np.arange(n), lens - 1], moves[np.arange(n), lens - 2]], 1).astype(np.int16)
    net = PolicyValueNetwork(seed=1).to(dev).eval()
    fused = FusedPolicyValueNetwork(net)                   # K9: the convolutions as one fused f32-MFMA kernel, the dense layers + softmax / tanh as a second
    tree = G.AlphaZeroMCTS(n, node_capacity=(P + 4) * 225 + 1)
    tree.set_roots(planes, last)
    with torch.no_grad():
        tree.search(fused, 3)                              # warm-up
        torch.cuda.synchronize()
        if distributed:
            torch.distributed.barrier()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        tree.search(fused, P)                              # (a hipGraph of the step, search(..., graph=True), gains nothing: the queue never runs dry)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1)
        s = tree.states.clone()
        def timed(fn, reps=10):
            fn()
            e0.record()
            for _ in range(reps):
                fn()
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) / reps
        net_ms, trunk_ms, torch_ms = timed(lambda: fused(s)), timed(lambda: fused.trunk(s)), timed(lambda: net(s))
        dv, dp = fused(s), net(s)
        err = max(float((dv[0] - dp[0]).abs().max()), float((dv[1] - dp[1]).abs().max()))
    tree.close()
    # whole games of this searcher against itself (AlphaZero-style self-play, the reference agent's semantics: kept subtrees, root noise before
    # every search), the loop resident on the device: gmk_az_advance per ply, continuous batching (gmk_az_set_slots), the leaf batch = the
    # games still played
    pipe_s, pipe_moves = 0.0, 0
    if args.az_selfplay_games > 0:
        import time
        from gomokuai_amd import selfplay
        torch.cuda.synchronize()
        if distributed:
            torch.distributed.barrier()
        def play():                                              # (gigabytes of records, activations and gather buffers: local work that can fail on one rank alone)
            t0 = time.perf_counter()
            rec = selfplay.play_network_games(args.az_selfplay_games, fused, args.az_selfplay_playouts, first_game_id=rank * args.az_selfplay_games, opening_plies=2,
                                              slots=max(1, args.az_selfplay_games // 2), reuse_subtree=True, root_noise=(0.05, 0.25))
            torch.cuda.synchronize()
            return time.perf_counter() - t0, int(rec.lens.sum()) - 2 * args.az_selfplay_games
        pipe_s, pipe_moves = local_stage("az self-play", play, torch, dev, distributed)
    (ms, pipe_s), (pipe_moves,) = reduce_leg(torch, dev, distributed, [ms, pipe_s], [pipe_moves])
    fused.close()
    pipeline = None
    if args.az_selfplay_games > 0:
        pipeline = {"metric": "self-play games/s", "value": args.az_selfplay_games * world / pipe_s, "unit": "games/s", "playouts_per_s": pipe_moves * args.az_selfplay_playouts / pipe_s,
                    "seconds": pipe_s, "moves": pipe_moves,
                    "config": {"workload": "network-guided self-play (K7 + K9), %d games per GPU through %d slots, %d playouts per move, kept subtrees and root noise (0.05, 0.25), 2-ply openings; "
                                           "loop resident on the device (gmk_az_set_slots / gmk_az_advance), leaf batch = the games still played" % (args.az_selfplay_games, max(1, args.az_selfplay_games // 2), args.az_selfplay_playouts)}}
    conv_flop = 2.0 * 225 * (54 * 32 + 288 * 64 + 576 * 128 + 128 * 6) * n
    return {"metric": "network-guided-playouts/s", "value": n * world * P / (ms * 1e-3), "unit": "playouts/s", "ms_per_step": ms / P,
            "network_ms_per_step": net_ms,
            "config": {"workload": "network-guided MCTS (K7), %d games x %d lock-step playouts per GPU, PolicyValueNetwork float32 with random weights, 4-ply openings" % (n, P)},
            "roofline": {"bound": "mfma", "achieved": conv_flop / (trunk_ms * 1e-3) / 1e12, "peak": 157.3, "unit": "TFLOP/s",
                         "frac": conv_flop / (trunk_ms * 1e-3) / 1e12 / 157.3, "traffic": measured_traffic("pvnet_trunk_kernel", "positions_per_launch", n), "traffic_source": "profiles/traffic.json (committed rocprofv3 PMC passes; not re-measured in this run)", "kernel": "pvnet_trunk_kernel", "kernel_ms": trunk_ms,
                         "alg_flop_per_launch": conv_flop, "note": "dense f32-input MFMA peak (MI355X_MICROARCH.md); the convolution FLOPs of the 225 real pixels"},
            "pytorch_module_ms_per_step": torch_ms, "max_abs_diff_vs_pytorch_module": err, "selfplay_pipeline": pipeline,
            "note": "the step is the network's forward pass, two HIP kernels (gmk_pvnet_evaluate): K9's fused convolution trunk and the dense layers with softmax / tanh, "
                    "both float32 MFMA; the select and expand kernels take the remainder"}


REFERENCE_NOISE = (0.05, 0.25)      # Default::AddNoise's defaults (MonteCarlo.hpp:97), what MCTS::runPlayouts calls it with (MCTS.cpp:182)


def selfplay_leg(args, torch, dev, rank, world, distributed, total_games, reference, scaling):
    """One run of the self-play data pipeline (network/data_helper.py:58-113): total_games IN TOTAL, sharded by global game id; every rank plays
    its shard to the end (K3, the loop resident on the device: gmk_selfplay_run), builds the training tuples on the device (K4 + K5), and the
    compact records travel to rank 0 (selfplay.gather_records: one batch of point-to-point transfers, RCCL over xGMI).  reference: the
    reference agent's per-move semantics (agents/mcts.py:17-21) -- the chosen child's subtree is the next search's tree (MCTS.cpp:129-147) and
    Default::AddNoise(0.05, 0.25) runs before every search (MCTS.cpp:179-183) -- instead of a new root every move without noise.  The time
    is the slowest rank's, barrier to barrier.  Returns the result object on rank 0, None elsewhere."""
    import time
    from gomokuai_amd import selfplay
    first, n = selfplay.shard(total_games, rank, world)
    P = args.mcts_playouts
    kw = dict(first_game_id=first, reuse_subtree=reference, root_noise=REFERENCE_NOISE if reference else None)
    # The tree arenas of this leg (47 GB, 142 GB with kept subtrees) are allocated -- and given back to the library's block pool -- BEFORE the timed
    # region: the driver clears memory it has handed out before at seconds per 24 GB, which a self-play job pays once, not per batch.
    local_stage("self-play (arenas)", lambda: selfplay.play_games(n, P, prepare_only=True, **kw), torch, dev, distributed)
    torch.cuda.synchronize()
    if distributed:
        torch.distributed.barrier()
    t0 = time.perf_counter()
    # (the local stages -- they allocate gigabytes -- end in an agreement of the ranks: see local_stage)
    def play():
        r = selfplay.play_games(n, P, **kw)
        torch.cuda.synchronize()
        return r
    rec = local_stage("self-play", play, torch, dev, distributed)
    t1 = time.perf_counter()
    def tuples():
        r = rec.to_samples(augment=False)
        torch.cuda.synchronize()
        return r
    states, values, pi = local_stage("training tuples", tuples, torch, dev, distributed)
    t2 = time.perf_counter()
    if distributed:
        # the gather's destination allocates the fixed-stride records of ALL ranks through torch (26 GB of visit rows for 8 x 32 768 games), and the
        # library's pool holds this leg's arenas (up to 142 GB) where torch's allocator cannot see them: give them back first (untimed: the next leg
        # allocates its own before its clock starts)
        from gomokuai_amd import lib as G
        G.release_pool()
        torch.cuda.synchronize()
    t2b = time.perf_counter()
    # the rehearsal mode's gloo has no device transfers: there the same exchange runs on host copies of the records
    gathered = selfplay.gather_records(rec if REDUCE_DEVICE is None else rec.cpu())
    torch.cuda.synchronize()
    if distributed:
        torch.distributed.barrier()
    t3 = time.perf_counter()
    # everything that can fail is above (inside local_stage / the gather's own agreement); from here on: packing numbers, then the reductions
    moves = int(rec.lens.sum())
    n_tuples = int(states.shape[0])
    overflow = bool(rec.overflow)
    times = [t1 - t0, t2 - t1, t3 - t2b, (t2 - t0) + (t3 - t2b), float(rec.lens.max())]
    times, (moves, n_tuples, n_overflow) = reduce_leg(torch, dev, distributed, times, [moves, n_tuples, int(overflow)])
    if rank != 0:
        return None
    plan = selfplay.plan_games(n, reuse_subtree=reference) or [(0, n, n)]
    return {"metric": "self-play games/s", "value": total_games / times[3], "unit": "games/s", "scaling": scaling,
            "playouts_per_s": moves * P / times[0], "seconds": times[3], "play_s": times[0], "tuples_s": times[1], "gather_s": times[2],
            "moves": moves, "training_tuples": n_tuples, "gathered_games": len(gathered), "gathered_first_game_id": int(gathered.first_game_id),
            "arena_overflow": bool(n_overflow), "mean_game_moves": moves / total_games, "longest_game_moves": int(times[4]),
            "config": {"workload": "self-play data pipeline (K3 -> K4 + K5 -> gather), %d games in total x %d playouts per move, RandomPolicy c_puct=5 "
                                   "c_rollouts=5, empty openings, visit counts recorded; %s" % (total_games, P,
                                   "the reference agent's per-move semantics: kept subtree (MCTS.cpp:129-147) + Default::AddNoise(0.05, 0.25) before every search (MCTS.cpp:179-183), "
                                   "drawn inside the ONE persistent launch from the counter-based sampler (include/gomoku_noise.h)" if reference else "a new root every move, no root noise (ONE persistent launch per handle)"),
                       "games_per_gpu": n, "slots_per_gpu": sum(sl for _, _, sl in plan), "search_handles_per_gpu": len(plan),
                       "parallelism": "games sharded by global id, one gather of the compact records to rank 0"}}


def reduce_leg(torch, dev, distributed, times, counts):
    """The reductions at the end of a leg: MAX of the times, SUM of the counts -- ONE collective (a float64 vector, the counts are far below 2^53), so
    that ranks cannot meet in different collectives; nothing between the leg's last agreement and this call may raise."""
    if not distributed:
        return list(times), list(counts)
    k = len(times)
    t = torch.tensor(list(times) + [float(c) for c in counts], dtype=torch.float64, device=REDUCE_DEVICE or dev)
    gathered = [torch.zeros_like(t) for _ in range(torch.distributed.get_world_size())]
    torch.distributed.all_gather(gathered, t)
    stacked = torch.stack(gathered)
    return [float(v) for v in stacked[:, :k].max(0).values], [int(round(float(v))) for v in stacked[:, k:].sum(0)]


def bench_selfplay(args, torch, dev, rank, world, distributed):
    """BASELINE configs[3].  (a) The STRONG form north_star states: --selfplay-games games IN TOTAL over the ranks -- with its reference_semantics
    twin (the same games as the reference's agent plays them: kept subtree + root noise before every search).  (b) The WEAK form:
    --selfplay-games-per-gpu games per rank (on one GPU with the defaults it is the same run as (a), which is then not repeated).
    (c) On one GPU, a REHEARSAL of the shard a rank gets at N = 8 in the strong form (total / 8 games): what "near-linear scaling to 8 GPUs"
    will read on that leg, before a node shows up."""
    strong = selfplay_leg(args, torch, dev, rank, world, distributed, args.selfplay_games, False, "strong")
    strong_ref = selfplay_leg(args, torch, dev, rank, world, distributed, args.selfplay_games, True, "strong")
    weak_total = args.selfplay_games_per_gpu * world
    weak = weak_ref = None
    if args.selfplay_games_per_gpu > 0 and weak_total != args.selfplay_games:
        weak = selfplay_leg(args, torch, dev, rank, world, distributed, weak_total, False, "weak")
        weak_ref = selfplay_leg(args, torch, dev, rank, world, distributed, weak_total, True, "weak")
    rehearsal = None
    if world == 1 and args.selfplay_rehearsal_ranks > 1 and args.selfplay_games >= args.selfplay_rehearsal_ranks:
        shard_games = args.selfplay_games // args.selfplay_rehearsal_ranks
        small = selfplay_leg(args, torch, dev, rank, world, distributed, shard_games, False, "strong")
        small_ref = selfplay_leg(args, torch, dev, rank, world, distributed, shard_games, True, "strong")
        rehearsal = {"ranks_rehearsed": args.selfplay_rehearsal_ranks, "games": shard_games,
                     "new_roots": {k: small[k] for k in ("value", "playouts_per_s", "seconds", "play_s", "moves", "mean_game_moves", "longest_game_moves")},
                     "reference_semantics": {k: small_ref[k] for k in ("value", "playouts_per_s", "seconds", "play_s", "moves", "mean_game_moves", "longest_game_moves")},
                     "tail": {"slots_busy_share": {"new_roots": small["mean_game_moves"] / small["longest_game_moves"], "reference_semantics": small_ref["mean_game_moves"] / small_ref["longest_game_moves"]},
                              "note": "every game of the shard has a slot of its own and starts at once, a search is a serial chain of %d playouts, and the launch ends with the longest game: "
                                      "mean game length / longest game length is the share of slot-time that plays.  Game lengths are not known in advance (no longest-first)" % args.mcts_playouts},
                     "predicted_strong_efficiency_at_%d" % args.selfplay_rehearsal_ranks: {"new_roots": small["value"] / strong["value"], "reference_semantics": small_ref["value"] / strong_ref["value"]},
                     "note": "one GPU playing the shard a rank gets when the strong form's %d games are split over %d ranks: its games/s over the full batch's games/s on the same GPU "
                             "= the strong-scaling efficiency that leg can reach at that N before any gather cost (a shard is ONE persistent launch: it ends with its longest game, "
                             "and %d games leave the chip's %d wavefront slots half empty from the start)" % (args.selfplay_games, args.selfplay_rehearsal_ranks, shard_games, 2048)}
    if rank != 0:
        return None
    out = strong
    ref_keys = ("value", "unit", "playouts_per_s", "seconds", "play_s", "tuples_s", "gather_s", "moves", "training_tuples", "arena_overflow", "mean_game_moves", "longest_game_moves", "config")
    out["reference_semantics"] = {k: strong_ref[k] for k in ref_keys}
    out["reference_semantics"]["share_of_new_root_playouts_per_s"] = strong_ref["playouts_per_s"] / strong["playouts_per_s"]
    if weak is not None:
        out["weak_scaling"] = {k: weak[k] for k in ref_keys + ("scaling",)}
        out["weak_scaling"]["reference_semantics"] = {k: weak_ref[k] for k in ref_keys}
    else:
        out["weak_scaling"] = {"note": "--selfplay-games-per-gpu x ranks = --selfplay-games: on this run the weak form IS the strong form above (games per GPU %d)" % args.selfplay_games_per_gpu}
    if rehearsal is not None:
        out["shard_rehearsal"] = rehearsal
    return out


def cpu_baseline_trad(G, playouts):
    """The oracle's restatement of the same search (oracle/go_trad.c), single thread, ~5 s."""
    from oracle import oracle as O
    pos = trad_positions(G, 512, 0)
    t0 = time.perf_counter()
    k = 0
    while time.perf_counter() - t0 < 5 and k < len(pos):
        O.TraditionalMCTS(5.0).search(pos[k], playouts)
        k += 1
    dt = time.perf_counter() - t0
    return {"value": k * playouts / dt, "unit": "playouts/s", "cores": 1, "kind": "port",
            "sample": "%d searches of %d playouts from the same openings, oracle TraditionalPolicy restatement, %.1f s" % (k, playouts, dt)}


def cpu_baseline_mcts(playouts):
    """BASELINE configs[0]-style CPU point: the oracle's MCTS restatement, single thread, a few whole searches."""
    import ctypes as C
    import numpy as np
    from gomokuai_amd import lib as G
    from oracle import oracle as O
    n_max = 2048
    moves, lens, _, _ = mcts_openings(G, np, n_max, 0)
    L = O.lib()
    t0 = time.perf_counter()
    done = 0
    for g in range(n_max):
        b = O.new_board()
        for i in range(int(lens[g])):
            L.go_board_apply(C.byref(b), int(moves[g, i]), 1)
        O.MCTS(playouts, 5.0, 5, G.DEFAULT_SEED, g).run_playouts(b)
        done += playouts
        if time.perf_counter() - t0 > 10:
            break
    dt = time.perf_counter() - t0
    return {"value": done / dt, "unit": "playouts/s", "cores": 1, "kind": "port",
            "sample": "%d searches of %d playouts from the same 4-ply openings, oracle MCTS restatement, %.1f s" % (done // playouts, playouts, dt)}


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--boards", type=int, default=65536, help="boards per GPU (BASELINE configs[1])")
    ap.add_argument("--kind", type=int, default=0, help="0 random-opening, 1 clustered")
    ap.add_argument("--cpu-sample", type=int, default=65536)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true", help="launch the timed K1 steps one by one instead of replaying a hipGraph")
    ap.add_argument("--mcts-games", type=int, default=4096, help="games per GPU for the secondary MCTS measurement (BASELINE configs[2]); 0 = skip")
    ap.add_argument("--mcts-playouts", type=int, default=800)
    ap.add_argument("--mcts-reps", type=int, default=3)
    ap.add_argument("--mcts-saturated-games", type=int, default=16384, help="games per GPU for the saturated-batch K3 figure beside configs[2]; 0 = skip")
    ap.add_argument("--selfplay-games", type=int, default=32768,
                    help="games IN TOTAL (over all GPUs) for the self-play pipeline measurement (BASELINE configs[3]); 0 = skip")
    ap.add_argument("--selfplay-games-per-gpu", type=int, default=32768,
                    help="games PER GPU for the weak-scaling form of the self-play pipeline (skipped where it coincides with --selfplay-games); 0 = skip")
    ap.add_argument("--selfplay-rehearsal-ranks", type=int, default=8,
                    help="one-GPU runs only: also play --selfplay-games / this many games, the shard a rank gets at that N (shard_rehearsal); 0 = skip")
    ap.add_argument("--evalstate-games", type=int, default=2304, help="games per GPU for the incremental-evaluator measurement (K2); 0 = skip")
    ap.add_argument("--az-games", type=int, default=4096, help="games per GPU for the network-guided search measurement (K7); 0 = skip")
    ap.add_argument("--az-playouts", type=int, default=60)
    ap.add_argument("--az-selfplay-games", type=int, default=8192, help="whole games per GPU of the network-guided searcher against itself (through half as many slots: tools/az_selfplay_slots.py); 0 = skip")
    ap.add_argument("--az-selfplay-playouts", type=int, default=32)
    ap.add_argument("--trad-games", type=int, default=2048, help="games per GPU for the pattern-guided search measurement (K6); 0 = skip")
    ap.add_argument("--trad-playouts", type=int, default=1000)
    ap.add_argument("--trad-nodes", type=int, default=1 << 18, help="node capacity per game")
    ap.add_argument("--trad-saturated-games", type=int, default=8192, help="games per GPU for the saturated-batch K6 figure beside the 2 048-game one; 0 = skip")
    ap.add_argument("--sup-games", type=int, default=16384, help="games per GPU for the supervisor self-play measurement (played through --trad-games slots; 8 192 until round 4: one game in a hundred is a "
                         "225-move draw, i.e. ~9 s of searches in a row, and with four games per slot the whole run lasted no longer than that -- the rate measured the tail); 0 = skip")
    ap.add_argument("--rave-games", type=int, default=4096, help="games per GPU for the PoolRAVE search measurement (K8); 0 = skip")
    ap.add_argument("--rave-playouts", type=int, default=400)
    ap.add_argument("--launch-timeout", type=float, default=1500.0, help="seconds after which `--gpus N` without a launcher stops its rank processes")
    ap.add_argument("--settle-ms", type=float, default=30.0,
                    help="milliseconds of untimed K1 launches ahead of the W warm-up steps: the chip's clock governor needs ~20 ms of continuous "
                         "work after an idle spell before the time per launch is steady (profiles/r03_ramp.txt); 0 = none")
    ap.add_argument("--stub-fail-rank", type=int, default=-1, help="--stub only: this rank exits with code 3 before the rendezvous (launcher test)")
    ap.add_argument("--stub-hang-rank", type=int, default=-1, help="--stub only: this rank never reaches the rendezvous (launcher time-limit test)")
    ap.add_argument("--stub", action="store_true",
                    help="launcher self-test without a GPU: ranks rendezvous over gloo, the step is a no-op and the line says so (tests/test_bench_launcher.py)")
    return ap.parse_args(argv)


def launch_ranks(n, argv, timeout_s=1500.0):
    """`python bench.py --gpus N` without a launcher around it: start N fresh rank processes of this script, one per GPU of the
    node (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in their environment, as torch.distributed.run would set them), and wait.
    This process has not touched the GPU (no torch import yet): the ranks are children, nothing is exec'ed over a GPU process."""
    import socket
    import subprocess
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    procs = []
    for rank in range(n):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env))
    # Fail fast and collectively: the ranks sit in barriers and reductions of one another, so the first rank that exits non-zero (or
    # the overall time limit) ends the others -- terminate, then kill -- and its code is the launcher's.  Only these children are
    # signalled, by pid.
    deadline = time.monotonic() + timeout_s
    code = 0
    while True:
        states = [p.poll() for p in procs]
        failed = [c for c in states if c not in (None, 0)]
        if failed:
            code = abs(failed[0])
            break
        if all(c == 0 for c in states):
            return 0
        if time.monotonic() > deadline:
            print("bench.py: ranks still running after %.0f s: stopping them" % timeout_s, file=sys.stderr)
            code = 124
            break
        time.sleep(0.05)
    for p in procs:
        if p.poll() is None:
            p.terminate()
    t_kill = time.monotonic() + 5.0
    for p in procs:
        try:
            p.wait(timeout=max(0.0, t_kill - time.monotonic()))
        except subprocess.TimeoutExpired:
            p.kill()
            p.wait()
    return code


def stub_rank(args, world, rank):
    """--stub: what the launcher and the rank bookkeeping do, without a GPU: rendezvous (gloo), barrier, max over ranks, one line."""
    if rank == args.stub_fail_rank:
        sys.exit(3)
    if rank == args.stub_hang_rank:
        time.sleep(3600)
    import datetime
    import torch
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=120))
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        pass
    elapsed = time.perf_counter() - t0 + 1e-9
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t[0])
        dist.barrier()
    # the legs' end-of-leg reduction (MAX of times, SUM of counts in ONE collective) and the self-play shards of both scaling forms, as the ranks see them
    from gomokuai_amd.selfplay import shard
    times, counts = reduce_leg(torch, torch.device("cpu"), world > 1, [1.0 + rank, 0.25], [rank + 1, 10])
    weak_total = args.selfplay_games_per_gpu * world
    mine = {"strong": shard(args.selfplay_games, rank, world), "weak": shard(weak_total, rank, world)}
    _, shard_games = reduce_leg(torch, torch.device("cpu"), world > 1, [], [mine["strong"][1], mine["weak"][1]])
    if rank == 0:
        print(json.dumps({"metric": "board-evals/s", "value": args.boards * world * args.steps / elapsed, "unit": "board-evals/s", "n_gpus": world,
                          "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
                          "scaling": "weak", "vs_baseline": None, "dtype": "int32", "data": "stub",
                          "reduce_leg": {"times_max": times, "counts_sum": counts},
                          "selfplay_pipeline": {"strong_total": args.selfplay_games, "weak_total": weak_total, "games_over_all_shards": {"strong": shard_games[0], "weak": shard_games[1]},
                                                "rank0_shard": {"strong": list(mine["strong"]), "weak": list(mine["weak"])},
                                                "rehearsal_games": args.selfplay_games // args.selfplay_rehearsal_ranks if world == 1 and args.selfplay_rehearsal_ranks > 1 else None},
                          "config": {"workload": "launcher self-test (--stub): no kernel ran"}}))
    if world > 1:
        dist.destroy_process_group()


def main():
    args = parse_args()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args.gpus, sys.argv[1:], args.launch_timeout))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if world != args.gpus:
        sys.exit("bench.py: --gpus %d but WORLD_SIZE=%d: the line's n_gpus must be the number of ranks that ran" % (args.gpus, world))
    if args.stub:
        return stub_rank(args, world, rank)

    import datetime
    import numpy as np
    import torch
    from gomokuai_amd import lib as G

    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    distributed = world > 1
    if distributed:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("GMK_BENCH_BACKEND", "nccl")
        if backend == "nccl":                                    # RCCL: one rank per GPU
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank), timeout=datetime.timedelta(seconds=600))
        else:
            # rehearsal of the N > 1 path on a box with fewer GPUs than ranks (GMK_BENCH_BACKEND=gloo): the ranks share the GPUs there are,
            # barriers and reductions go over gloo on the CPU; everything else is the code the RCCL run executes
            global REDUCE_DEVICE
            REDUCE_DEVICE = torch.device("cpu")
            torch.cuda.set_device(local_rank % torch.cuda.device_count())
            dist.init_process_group(backend, timeout=datetime.timedelta(seconds=600))
    else:
        torch.cuda.set_device(0)
    dev = torch.device("cuda", torch.cuda.current_device())
    G.init(torch.cuda.current_device())

    # independent boards shard across ranks by global board id; no collective on the data path
    n = args.boards
    _, _, planes = G.synth_boards(n, args.kind, first_board=rank * n)
    d_planes = torch.from_numpy(planes.view(np.int16).reshape(n, 32)).to(dev)
    d_scores = torch.empty((n, 4, 225), dtype=torch.int32, device=dev)
    d_density = torch.empty((n, 2, 2, 225), dtype=torch.int32, device=dev)
    d_totals = torch.empty((n, 11), dtype=torch.int32, device=dev)
    d_status = torch.empty((n,), dtype=torch.int32, device=dev)
    def step(stream=None):
        G.eval_batch(d_planes.data_ptr(), n, d_scores.data_ptr(), d_density.data_ptr(), d_totals.data_ptr(),
                     d_status.data_ptr(), torch.cuda.current_stream().cuda_stream if stream is None else stream)

    step()                                                   # the library's first call uploads its constant tables: it cannot be part of a capture
    torch.cuda.synchronize()
    # the K timed steps are captured once into a hipGraph (K kernel nodes) and replayed: no per-launch host work in the
    # timed region ("graph-launched"); --no-graph, or a failed capture, launches them one by one instead
    graph = None
    if not args.no_graph:
        try:
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                for _ in range(args.steps):
                    step()
        except Exception as exc:                             # noqa: BLE001
            print("bench: hipGraph capture failed (%s); launching step by step" % exc, file=sys.stderr)
            graph = None
            torch.cuda.synchronize()
    # Settling: after an idle spell (everything above is host work) the chip's clock governor takes ~20 ms of continuous work before
    # the time per launch is steady -- launches 10 to 40 after idle run 15-20 % SLOWER than launches 100 on (tools/ramp_probe.py,
    # profiles/r03_ramp.txt) -- and the timed region of K = 20 steps is 3 ms long.  So the same step runs untimed for --settle-ms
    # of device time first (reported in the line), then the W warm-up steps, then the timed K; no gap between them is long enough
    # (> ~2 ms) to send the governor back.
    settle = {"ms": 0.0, "launches": 0}
    cold_ms = None
    if args.settle_ms > 0:
        # (for the record, the same measurement WITHOUT settling first: W warm-up steps from idle, then K timed steps)
        for _ in range(args.warmup):
            step()
        if graph is not None:
            graph.replay()
        torch.cuda.synchronize()
        c0, c1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        c0.record()
        if graph is not None:
            graph.replay()
        else:
            for _ in range(args.steps):
                step()
        c1.record()
        torch.cuda.synchronize()
        cold_ms = c0.elapsed_time(c1) / args.steps
        s0, s1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s0.record()
        while True:
            for _ in range(16):
                step()
            settle["launches"] += 16
            s1.record()
            s1.synchronize()
            settle["ms"] = s0.elapsed_time(s1)
            if settle["ms"] >= args.settle_ms or settle["launches"] >= 4096:
                break
    for _ in range(args.warmup):
        step()
    if graph is not None:
        graph.replay()                                       # one untimed replay: the graph is uploaded here
    torch.cuda.synchronize()
    if distributed:
        dist.barrier()
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    if graph is not None:
        graph.replay()
    else:
        for _ in range(args.steps):
            step()
    ev1.record()
    torch.cuda.synchronize()
    if distributed:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    kernel_ms = ev0.elapsed_time(ev1) / args.steps           # HIP events on the launch stream
    if distributed:
        t = torch.tensor([elapsed, kernel_ms], dtype=torch.float64, device=REDUCE_DEVICE or dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed, kernel_ms = float(t[0]), float(t[1])

    mcts = None
    if args.mcts_games > 0:
        mcts = bench_mcts(args, G, torch, dev, rank, world, distributed)

    incremental = None
    if args.evalstate_games > 0:
        incremental = bench_evalstate(args, G, torch, dev, rank, world, distributed)

    trad = None
    if args.trad_games > 0:
        trad = bench_trad(args, G, torch, dev, rank, world, distributed)

    sup_pipeline = sup_error = None
    if args.sup_games > 0 and args.trad_games > 0:
        try:
            sup_pipeline = bench_supervisor_pipeline(args, G, torch, dev, rank, world, distributed, trad["value"] if trad else None)
        except Exception as exc:                                  # noqa: BLE001
            sup_error = "%s: %s" % (type(exc).__name__, exc)
        if not all_ranks_ok(sup_error is None, torch, dev, distributed) and sup_error is None:
            sup_pipeline, sup_error = None, "LegFailed: the leg failed on another rank"

    rave = None
    if args.rave_games > 0:
        rave = bench_rave(args, G, torch, dev, rank, world, distributed)

    az = az_error = None
    if args.az_games > 0:
        try:
            az = bench_az(args, G, torch, dev, rank, world, distributed)
        except Exception as exc:                                  # noqa: BLE001
            az_error = "%s: %s" % (type(exc).__name__, exc)
        if not all_ranks_ok(az_error is None, torch, dev, distributed) and az_error is None:
            az, az_error = None, "LegFailed: the leg failed on another rank"

    pipeline = pipeline_error = None
    if args.selfplay_games >= world:                            # every rank needs a game (the same decision on all ranks: the leg has barriers)
        # the headline line must not be lost to this leg: a failure is reported in its place -- by every rank alike (the leg's local stages
        # end in an agreement of the ranks, so no rank is left in a barrier of the leg; whatever still escapes that is agreed on here)
        try:
            pipeline = bench_selfplay(args, torch, dev, rank, world, distributed)
        except Exception as exc:                                  # noqa: BLE001
            pipeline_error = "%s: %s" % (type(exc).__name__, exc)
        if not all_ranks_ok(pipeline_error is None, torch, dev, distributed) and pipeline_error is None:
            pipeline, pipeline_error = None, "LegFailed: the leg failed on another rank"

    if rank == 0:
        achieved = ALG_BYTES_PER_EVAL * n / (kernel_ms * 1e-3) / 1e9
        out = {
            "metric": "board-evals/s",
            "value": n * world * args.steps / elapsed,
            "unit": "board-evals/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            # every untimed launch of the step ahead of the timed region: the table upload (1), with --settle-ms > 0 the cold measurement (W, the graph's
            # upload replay K, its K timed steps -- reported as roofline.without_settling) and the settle launches, then the W warm-up steps and the graph's upload replay
            "warmup_effective": 1 + (args.warmup + (args.steps if graph is not None else 0) + args.steps + settle["launches"] if args.settle_ms > 0 else 0)
                                + args.warmup + (args.steps if graph is not None else 0),
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "int32",
            "data": "synthetic",
            "config": {"workload": "batched AC-automaton position eval (K1), %d %s boards per GPU, 15x15, inputs resident in HBM"
                                   % (n, "random-opening" if args.kind == 0 else "clustered"),
                       "boards_per_gpu": n, "parallelism": "boards sharded by rank, no collective" + ("" if os.environ.get("GMK_BENCH_BACKEND", "nccl") == "nccl" or world == 1
                                                                                                       else " (REHEARSAL: %d ranks over gloo sharing %d GPU(s))" % (world, torch.cuda.device_count())),
                       "launch": "hipGraph replay of the K steps" if graph is not None else "K stream launches",
                       "settle": {"untimed_launches_before_warmup": settle["launches"], "ms": settle["ms"],
                                  "why": "clock governor: time per launch is steady only after ~20 ms of continuous work (profiles/r03_ramp.txt); --settle-ms 0 disables"}},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS,
                         "traffic": measured_traffic("eval_positions_kernel", "boards_per_launch", n),
                         "traffic_source": "profiles/traffic.json (committed rocprofv3 PMC passes of this kernel; not re-measured in this run)",
                         "kernel": "eval_positions_kernel", "kernel_ms": kernel_ms,
                         "alg_bytes_per_launch": ALG_BYTES_PER_EVAL * n,
                         "without_settling": None if cold_ms is None else {
                             "kernel_ms": cold_ms, "frac": ALG_BYTES_PER_EVAL * n / (cold_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                             "note": "this rank, the same W warm-up + K timed steps straight after the idle spell of the set-up, before config.settle: "
                                     "the K steps then fall into the clock governor's dip (profiles/r03_ramp.txt)"}},
        }
        if mcts is not None:
            out["secondary"] = mcts
        if pipeline is not None or pipeline_error is not None:
            out["selfplay_pipeline"] = pipeline if pipeline is not None else {"error": pipeline_error}
        if incremental is not None:
            out["incremental"] = incremental
        if trad is not None:
            out["supervisor"] = trad
        if sup_pipeline is not None or sup_error is not None:
            out["supervisor_pipeline"] = sup_pipeline if sup_pipeline is not None else {"error": sup_error}
        if rave is not None:
            out["poolrave"] = rave
        if az is not None or az_error is not None:
            out["network_guided"] = az if az is not None else {"error": az_error}
        if not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.cpu_sample, args.kind)
            if mcts is not None:
                out["secondary"]["cpu_baseline"] = cpu_baseline_mcts(args.mcts_playouts)
            if incremental is not None:
                out["incremental"]["cpu_baseline"] = cpu_baseline_evalstate(G, 16384)
            if trad is not None:
                out["supervisor"]["cpu_baseline"] = cpu_baseline_trad(G, args.trad_playouts)
            if rave is not None:
                out["poolrave"]["cpu_baseline"] = cpu_baseline_rave(G, args.rave_playouts)
        print(json.dumps(out))
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
