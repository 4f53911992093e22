#!/usr/bin/env python3
"""What the box's HBM takes for plain coalesced writes (torch fill / copy), for comparison with K1's 472 MB of outputs per launch."""
import torch
torch.cuda.set_device(0)
for mb in (236, 472, 944):
    x = torch.empty(mb * 1024 * 1024 // 4, dtype=torch.int32, device="cuda")
    y = torch.empty_like(x)
    for name, f, nbytes in (("fill", lambda: x.fill_(7), x.numel() * 4), ("copy", lambda: y.copy_(x), x.numel() * 8)):
        for _ in range(5): f()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50): f()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 50
        print("%4d MB %s: %.4f ms  %.2f TB/s" % (mb, name, ms, nbytes / ms / 1e9))
