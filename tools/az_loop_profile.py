"""Where a ply of the network-guided self-play loop (selfplay.play_network_games, slots on the device) spends its time:
root noise (host), the search (select kernel, network, expand kernel per playout) and gmk_az_advance."""
import sys, time
import numpy as np
sys.path.insert(0, '.')
import torch
from gomokuai_amd import lib as G, selfplay
from gomokuai_amd.network import FusedPolicyValueNetwork, PolicyValueNetwork

n_games, slots, playouts = int(sys.argv[1]) if len(sys.argv) > 1 else 8192, int(sys.argv[2]) if len(sys.argv) > 2 else 2048, 32
reuse, noise = True, (0.05, 0.25)
G.init(0)
net = FusedPolicyValueNetwork(PolicyValueNetwork(seed=1).cuda().eval())
games = selfplay._HostGames(n_games)
m, l, _ = G.synth_boards(n_games, 0)
games.open_with(m, l, 2)
tree = G.AlphaZeroMCTS(slots, node_capacity=3 * playouts * 225 + 1)
tree.set_slots(n_games, games.moves[:, :2], games.lens)
dev = torch.device("cuda")
d_moves, d_lens = torch.from_numpy(games.moves).to(dev), torch.from_numpy(games.lens).to(dev)
d_winner = torch.zeros(n_games, dtype=torch.int8, device=dev)
d_visits = torch.zeros((n_games, 225, 225), dtype=torch.int16, device=dev)
t_noise = t_search = t_adv = t_sel = t_net = t_exp = 0.0
plies = 0
with torch.no_grad():
    while True:
        torch.cuda.synchronize(); t0 = time.perf_counter()
        tree.add_root_noise(noise[0], noise[1])
        torch.cuda.synchronize(); t1 = time.perf_counter()
        if plies % 10 == 5:                                        # every tenth ply step by step
            for _ in range(playouts):
                a = time.perf_counter(); st = tree.select(); torch.cuda.synchronize(); b = time.perf_counter()
                v, p = net(st); torch.cuda.synchronize(); c = time.perf_counter()
                tree.expand(v.contiguous(), p.contiguous()); torch.cuda.synchronize(); d = time.perf_counter()
                t_sel += b - a; t_net += c - b; t_exp += d - c
        else:
            tree.search(net, playouts)
        torch.cuda.synchronize(); t2 = time.perf_counter()
        left = tree.advance(d_moves, d_visits, d_lens, d_winner, reuse)
        torch.cuda.synchronize(); t3 = time.perf_counter()
        t_noise += t1 - t0; t_search += t2 - t1; t_adv += t3 - t2
        plies += 1
        if left == 0:
            break
tot = t_noise + t_search + t_adv
print("%d games through %d slots, %d playouts per move: %d plies in %.2f s: noise %.1f %%, search %.1f %%, advance %.1f %% (%.2f / %.2f / %.2f ms per ply)"
      % (n_games, slots, playouts, plies, tot, 100 * t_noise / tot, 100 * t_search / tot, 100 * t_adv / tot, 1e3 * t_noise / plies, 1e3 * t_search / plies, 1e3 * t_adv / plies))
k = t_sel + t_net + t_exp
print("a playout step, synchronised after every launch: select %.3f ms, network %.3f ms, expand %.3f ms" % tuple(1e3 * x / (playouts * ((plies + 4) // 10)) for x in (t_sel, t_net, t_exp)))
