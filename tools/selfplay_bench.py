#!/usr/bin/env python3
"""BASELINE configs[3]: the self-play data pipeline.  Every rank plays its shard of games to the end on its GPU
(K3 + gmk_mcts_advance), turns the records into training tuples on the device (K4 + K5), and the compact records are
gathered to rank 0 (RCCL over xGMI when launched with torch.distributed.run, identity on one GPU).
    python tools/selfplay_bench.py --games 4096 --playouts 800
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 tools/selfplay_bench.py --games 32768"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--games", type=int, default=4096, help="total games over all ranks")
    ap.add_argument("--playouts", type=int, default=800)
    ap.add_argument("--reuse", action="store_true", help="keep the subtree of the played move (MCTS::stepForward)")
    ap.add_argument("--noise", action="store_true", help="Default::AddNoise(0.05, 0.25) before every search (needs --reuse)")
    ap.add_argument("--augment", action="store_true")
    ap.add_argument("--lockstep", action="store_true", help="random: search-by-search launches instead of ONE persistent launch")
    ap.add_argument("--sampler", default="counter", choices=["counter", "std"], help="random: where the root noise is drawn (std: on the host, which forces lock step)")
    ap.add_argument("--handles", default="auto")
    ap.add_argument("--host-loop", action="store_true", help="traditional / poolrave / network: drive the games from the host ply by ply (the loop the device-resident one replaced)")
    ap.add_argument("--slots", default=None, help="games in flight: a finished game hands its slot to the next one (random: on the device, gmk_selfplay_run; default there: "
                    "selfplay.SLOTS_PER_GPU when a rank has more games than that); 0 = all games at once")
    ap.add_argument("--policy", default="random", choices=["random", "traditional", "poolrave", "network"],
                    help="who plays: RandomPolicy (K3), TraditionalPolicy (K6), PoolRAVEPolicy (K8), the fused policy-value network (K7 + K9)")
    args = ap.parse_args()
    import torch
    import torch.distributed as dist
    from gomokuai_amd import selfplay
    world, rank, local = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    first, n = selfplay.shard(args.games, rank, world)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    noise = (0.05, 0.25) if args.noise else None
    slots = None if args.slots in ("0", 0) else int(args.slots) if args.slots is not None else ("auto" if args.policy == "random" else None)
    if args.policy == "random":
        rec = selfplay.play_games(n, args.playouts, first_game_id=first, reuse_subtree=args.reuse, root_noise=noise, slots=slots, lockstep=args.lockstep,
                                  noise_sampler=args.sampler, handles=args.handles if args.handles == "auto" else int(args.handles))
    elif args.policy == "network":
        from gomokuai_amd.network import FusedPolicyValueNetwork, PolicyValueNetwork
        net = FusedPolicyValueNetwork(PolicyValueNetwork(seed=1).cuda().eval())
        rec = selfplay.play_network_games(n, net, args.playouts, first_game_id=first, opening_plies=2, reuse_subtree=args.reuse, root_noise=noise, device_loop=not args.host_loop)
    else:
        rec = selfplay.play_supervisor_games(n, args.playouts, c_puct=5.0 if args.policy == "traditional" else 2.0, first_game_id=first, opening_plies=2,
                                             reuse_subtree=args.reuse, root_noise=noise, policy=args.policy, slots=slots, device_loop=not args.host_loop)
    torch.cuda.synchronize()
    t_play = time.perf_counter() - t0
    t1 = time.perf_counter()
    states, values, pi = rec.to_samples(augment=args.augment)
    torch.cuda.synchronize()
    t_samples = time.perf_counter() - t1
    t2 = time.perf_counter()
    allrec = selfplay.gather_records(rec)
    torch.cuda.synchronize()
    t_gather = time.perf_counter() - t2
    moves = int(rec.lens.sum())
    if world > 1:
        tt = torch.tensor([t_play, t_samples, t_gather, float(moves)], dtype=torch.float64, device="cuda")
        mx = tt.clone()
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        dist.all_reduce(tt, op=dist.ReduceOp.SUM)
        t_play, t_samples, t_gather, moves = float(mx[0]), float(mx[1]), float(mx[2]), int(tt[3])
    if rank == 0:
        print(json.dumps({"workload": "self-play pipeline", "policy": args.policy, "games": args.games, "n_gpus": world, "playouts_per_move": args.playouts,
                          "moves": moves, "samples": moves * (8 if args.augment else 1), "arena_overflow": rec.overflow,
                          "play_s": t_play, "samples_s": t_samples, "gather_s": t_gather,
                          "games_per_s": args.games / (t_play + t_samples + t_gather),
                          "playouts_per_s": moves * args.playouts / t_play,
                          "gathered_games": len(allrec) if allrec is not None else None}))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
