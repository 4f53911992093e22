"""The reference's data-generation pairing (selfplay.play_match_games: the supervisor against a candidate, sides drawn per game, both
searchers from fresh roots every move) is driven from the host ply by ply.  What that costs: the share of the wall time that is NOT a
search kernel, at the reference's candidate budget (400 iterations) and a supervisor budget scaled down from its 20 000."""
import sys, time
sys.path.insert(0, '.')
import torch
from gomokuai_amd import lib as G, selfplay

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
sup_playouts = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
cand = sys.argv[3] if len(sys.argv) > 3 else "rave_mcts"
G.init(0)
gpu = {"traditional_mcts": 0.0, cand: 0.0}
calls = {"traditional_mcts": 0, cand: 0}
run_of = {}
for cls in (G.TraditionalMCTS, G.PoolRAVEMCTS, G.BatchedMCTS):
    run_of[cls] = cls.run
def timed_run(cls, kind):
    def run(self, *a, **k):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        r = run_of[cls](self, *a, **k)
        torch.cuda.synchronize(); gpu[kind] += time.perf_counter() - t0; calls[kind] += 1
        return r
    return run
G.TraditionalMCTS.run = timed_run(G.TraditionalMCTS, "traditional_mcts")
if cand == "rave_mcts":
    G.PoolRAVEMCTS.run = timed_run(G.PoolRAVEMCTS, cand)
else:
    G.BatchedMCTS.run = timed_run(G.BatchedMCTS, cand)
torch.cuda.synchronize(); t0 = time.perf_counter()
rec, sup_black = selfplay.play_match_games(n, ("traditional_mcts", {"c_puct": 5.0, "c_iterations": sup_playouts}), (cand, {"c_puct": 5.0, "c_iterations": 400}), opening_plies=2)
torch.cuda.synchronize(); total = time.perf_counter() - t0
moves = int(rec.lens.sum()) - 2 * n
search = sum(gpu.values())
print("%d games, supervisor %d iterations per move against %s at 400: %.2f s, %d moves; search kernels %.2f s (supervisor %.2f s in %d searches, candidate %.2f s in %d), "
      "everything else (positions up, root statistics down, numpy boards, handle set-up) %.2f s = %.1f %%" %
      (n, sup_playouts, cand, total, moves, search, gpu["traditional_mcts"], calls["traditional_mcts"], gpu[cand], calls[cand], total - search, 100 * (total - search) / total))
per_ply_host = (total - search) / max(1, calls["traditional_mcts"] + calls[cand])
sup_per_search = gpu["traditional_mcts"] / max(1, calls["traditional_mcts"])
print("per search: host side %.1f ms; a supervisor search of 20 000 iterations would take %.2f s: host share at the reference's budget ~%.1f %%" %
      (1e3 * per_ply_host, sup_per_search * 20000 / sup_playouts, 100 * per_ply_host / (per_ply_host + 0.5 * (sup_per_search * 20000 / sup_playouts + gpu[cand] / max(1, calls[cand])))))
