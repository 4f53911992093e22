#!/usr/bin/env python3
"""K9 timing aid: the fused trunk kernel against the PyTorch (MIOpen) module, float32, n positions."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gomokuai_amd import lib as G
from gomokuai_amd.network import FusedPolicyValueNetwork, PolicyValueNetwork
torch.cuda.set_device(0); G.init(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
net = PolicyValueNetwork(seed=1).cuda().eval()
fused = FusedPolicyValueNetwork(net)
states = (torch.rand((n, 6, 15, 15), device="cuda") > 0.7).float()
def timed(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
with torch.no_grad():
    t_ref = timed(lambda: net(states))
    t_trunk = timed(lambda: fused.trunk(states))
    t_fused = timed(lambda: fused(states))
    t_torch_dense = timed(lambda: fused.dense_reference(states))
    rv, rp = net(states); v, p = fused(states)
flop = n * 2 * 225 * (54 * 32 + 288 * 64 + 576 * 128 + 128 * 6)
print("n=%d: torch module %.3f ms; fused trunk %.3f ms (%.1f TFLOP/s f32), with the dense kernel %.3f ms (with PyTorch's dense layers %.3f ms); max |dvalue| %.2e, max |dprobs| %.2e" %
      (n, t_ref, t_trunk, flop / t_trunk / 1e9, t_fused, t_torch_dense, float((v - rv).abs().max()), float((p - rp).abs().max())))
