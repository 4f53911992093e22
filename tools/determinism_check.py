#!/usr/bin/env python3
"""Runs the searches at bench scale twice and compares everything they report: a data race inside a kernel would show up as a
difference between two runs of the same inputs (the parity tests cover small batches; this covers full occupancy)."""
import hashlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gomokuai_amd import lib as G
torch.cuda.set_device(0); G.init(0)


def digest(arrays):
    h = hashlib.sha256()
    for a in arrays:
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()[:16]


def k1():
    _, _, planes = G.synth_boards(65536, 1, first_board=31)
    return digest(G.eval_batch_host(planes))


def k3():
    n = 4096
    moves, lens, _ = G.synth_boards(n, 0)
    lens = np.minimum(lens, 4).astype(np.int32)
    t = G.BatchedMCTS(n, playouts_capacity=800)
    t.set_roots(G.moves_to_planes(moves, lens), np.array([moves[g, lens[g] - 1] for g in range(n)], np.int16), first_game_id=0)
    t.run(800); out = t.root_stats(); t.close()
    return digest(out)


def k6():
    n = 2048
    moves, lens, _ = G.synth_boards(n, 1)
    t = G.TraditionalMCTS(n, node_capacity=1 << 18)
    t.set_positions([[int(m) for m in moves[g, :min(int(lens[g]), 12)]] for g in range(n)])
    t.run(1000); st = t.root_stats(); ev = t.read_evaluators(); t.close()
    return digest([st[k] for k in sorted(st)] + [ev[k] for k in sorted(ev)])


def k8():
    n = 4096
    moves, lens, _ = G.synth_boards(n, 0)
    t = G.PoolRAVEMCTS(n, node_capacity=400 * 222 + 512)
    t.set_positions([[int(m) for m in moves[g, :4]] for g in range(n)])
    t.run(400); st = t.root_stats(); t.close()
    return digest([st[k] for k in sorted(st)])


def k9():
    from gomokuai_amd.network import FusedPolicyValueNetwork, PolicyValueNetwork
    net = PolicyValueNetwork(seed=1).cuda().eval(); fused = FusedPolicyValueNetwork(net)
    states = (torch.rand((4096, 6, 15, 15), device="cuda", generator=torch.Generator("cuda").manual_seed(1)) > 0.7).float()
    p, v = fused.trunk(states); torch.cuda.synchronize()
    out = digest([p.cpu().numpy(), v.cpu().numpy()]); fused.close()
    return out


bad = 0
for name, fn in (("K1", k1), ("K3", k3), ("K6", k6), ("K8", k8), ("K9", k9)):
    t0 = time.time(); a = fn(); b = fn()
    print("%s: %s %s %s  (%.1f s)" % (name, a, b, "same" if a == b else "DIFFERENT", time.time() - t0), flush=True)
    bad += a != b
sys.exit(1 if bad else 0)
