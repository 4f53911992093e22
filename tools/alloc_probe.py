"""How long the tree arenas of a self-play batch take to allocate and free (hipMalloc / hipFree of tens of GB), process after process."""
import sys, time
sys.path.insert(0, '.')
import torch
from gomokuai_amd import lib as G
G.init(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    t = G.BatchedMCTS(n, playouts_capacity=800)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    t.close()
    torch.cuda.synchronize(); t2 = time.perf_counter()
    print("rep %d: create %.3f s, close %.3f s (%d games x 800 playouts: %.1f GB)" % (rep, t1 - t0, t2 - t1, n, n * (800 * 225 + 1) * 16 / 1e9))
