#!/usr/bin/env python3
"""K7 + K9 whole-game self-play (selfplay.play_network_games: kept subtrees, device-drawn root noise, 32 playouts per move) at several slot counts:
games/s and playouts/s of 4 096 games."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gomokuai_amd import lib as G, selfplay
from gomokuai_amd.network import FusedPolicyValueNetwork, PolicyValueNetwork
torch.cuda.set_device(0); G.init(0)
games, playouts = int(os.environ.get("GAMES", "4096")), 32
net = FusedPolicyValueNetwork(PolicyValueNetwork(seed=1).cuda().eval())
for slots in [int(a) for a in sys.argv[1:]] or [1024, 2048, 4096]:
    for rep in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        rec = selfplay.play_network_games(games, net, playouts, first_game_id=0, opening_plies=2, slots=slots, reuse_subtree=True, root_noise=(0.05, 0.25))
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        moves = int(rec.lens.sum()) - 2 * games
    print("slots %5d: %.2f s, %.0f games/s, %.2f M playouts/s, %d moves" % (slots, dt, games / dt, moves * playouts / dt / 1e6, moves), flush=True)
