#!/usr/bin/env python3
"""K1 timing aid: kernel time per launch with optional outputs switched off (NULL output pointers are allowed by the C-ABI).
usage: eval_time.py [variant ...] [check]   variants: all, no-density, no-scores, neither (default: every one of them);
check: also print a position-weighted checksum of the four outputs (tools/k1_variants.py compares it across builds)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gomokuai_amd import lib as G
torch.cuda.set_device(0); G.init(0)
n = int(os.environ.get("GMK_EVAL_BOARDS", "65536"))
_, _, planes = G.synth_boards(n, int(os.environ.get("GMK_EVAL_KIND", "0")))
if os.environ.get("GMK_EVAL_STRATIFY"):
    # diagnostic: the boards dealt to the 256 workgroup ranges like cards, in order of their stone count: every workgroup gets the same mix
    # (what is left of the kernel's time against the natural order is what uneven workgroup sums cost)
    stones = np.unpackbits(planes.view(np.uint8).reshape(n, -1), axis=1).sum(1)
    order = np.argsort(stones, kind="stable")
    w = 256
    dealt = np.empty(n, dtype=np.int64)
    dealt[(np.arange(n) % w) * (n // w) + np.arange(n) // w] = order
    planes = planes[dealt]
dev = torch.device("cuda", 0)
d_planes = torch.from_numpy(planes.view(np.int16).reshape(n, 32)).to(dev)
d_scores = torch.empty((n, 900), dtype=torch.int32, device=dev)
d_density = torch.empty((n, 1024), dtype=torch.int32, device=dev)      # room for the aligned-layout experiment (GMK_EVAL_PHASE_MASK bit 8)
d_totals = torch.empty((n, 11), dtype=torch.int32, device=dev)
d_status = torch.empty((n,), dtype=torch.int32, device=dev)
stream = torch.cuda.current_stream().cuda_stream
variants = {"all": (1, 1), "no-density": (1, 0), "no-scores": (0, 1), "neither": (0, 0)}
reps = int(os.environ.get("GMK_EVAL_REPS", "100"))
check = "check" in sys.argv
for name in ([a for a in sys.argv[1:] if a != "check"] or list(variants)):
    sc, de = variants[name]
    f = lambda: G.eval_batch(d_planes.data_ptr(), n, d_scores.data_ptr() if sc else 0, d_density.data_ptr() if de else 0, d_totals.data_ptr(), d_status.data_ptr(), stream)
    for _ in range(10): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    print("%-12s %.4f ms" % (name, e0.elapsed_time(e1) / reps))
    if check and name == "all":
        cs = 0
        for t in (d_scores, d_density.view(-1)[:n * 900], d_totals, d_status):
            v = t.contiguous().view(-1).long()
            cs = (cs * 1000003 + int((v * (torch.arange(v.numel(), device=dev) % 65521 + 1)).sum().item())) % (1 << 61)
        print("checksum %x" % cs)
