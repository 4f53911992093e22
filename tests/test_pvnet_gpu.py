"""K9: the policy-value network in HIP -- the fused convolution trunk (gmk_pvnet_forward) and, behind it, the dense layers with softmax / tanh
(gmk_pvnet_evaluate), both on the f32 matrix cores -- against the plain PyTorch float32 module of the same architecture (gomokuai_amd/network.py, network/model_tf.py:28-66).  Both compute in float32; the sums run in
different orders, so the bar is a tolerance: 2e-5 absolute on activations of order 1 and on the final value / probabilities."""
import numpy as np
import pytest
import torch

from gomokuai_amd import lib as G
from gomokuai_amd.network import FusedPolicyValueNetwork, PolicyValueNetwork

pytestmark = pytest.mark.gpu
TOL = 2e-5


def _reference_trunk(net, states):
    x = states
    for conv in net.conv:
        x = torch.relu(conv(x))
    p = torch.relu(net.policy_conv(x)).permute(0, 2, 3, 1).reshape(x.shape[0], -1)
    v = torch.relu(net.value_conv(x)).permute(0, 2, 3, 1).reshape(x.shape[0], -1)
    return p, v


def _encoded_states(n, first=0):
    """real inputs: the six feature planes of positions from the synthetic generator"""
    from oracle import oracle as O
    import ctypes as C
    moves, lens, _ = G.synth_boards(n, 1, first_board=first)
    out = np.zeros((n, 6, 15, 15), np.float32)
    for g in range(n):
        b = O.new_board()
        for i in range(int(lens[g]) - 1):
            O.lib().go_board_apply(C.byref(b), int(moves[g, i]), 1)
        planes = np.zeros(6 * 225, np.uint8)
        O.lib().go_board_encoded_states(C.byref(b), planes.ctypes.data)
        out[g] = planes.reshape(6, 15, 15)
    return out


@pytest.mark.parametrize("n", [1, 5, 300, 1029])
def test_trunk_matches_torch(n):
    G.init()
    torch.manual_seed(n)
    net = PolicyValueNetwork(seed=3).cuda().eval()
    with torch.no_grad():
        for m in net.modules():                                  # non-zero biases: the initialiser leaves them at zero
            if hasattr(m, "bias") and m.bias is not None:
                m.bias.uniform_(-0.2, 0.2)
    fused = FusedPolicyValueNetwork(net)
    states = torch.rand((n, 6, 15, 15), device="cuda") * 2 - 0.5
    with torch.no_grad():
        rp, rv = _reference_trunk(net, states)
        rvalue, rprobs = net(states)
    p, v = fused.trunk(states)
    assert float((p - rp).abs().max()) < TOL * max(1.0, float(rp.abs().max())), float((p - rp).abs().max())
    assert float((v - rv).abs().max()) < TOL * max(1.0, float(rv.abs().max()))
    value, probs = fused(states)
    assert float((value - rvalue).abs().max()) < TOL and float((probs - rprobs).abs().max()) < TOL
    assert float(rp.abs().max()) > 0.1 and (rp > 0).any() and (rp == 0).any()          # the comparison is not vacuous
    fused.close()


def test_real_positions_and_search(oracle):
    """Feature planes of real positions (zeros, ones, the border effects of 'same' padding at every edge), and a lock-step
    search (K7) that calls the fused network: same visit counts as with the PyTorch module unless two children tie to rounding."""
    G.init()
    net = PolicyValueNetwork(seed=1).cuda().eval()
    fused = FusedPolicyValueNetwork(net)
    states = torch.from_numpy(_encoded_states(64)).cuda()
    with torch.no_grad():
        rvalue, rprobs = net(states)
    value, probs = fused(states)
    assert float((value - rvalue).abs().max()) < TOL and float((probs - rprobs).abs().max()) < TOL
    n, playouts = 32, 40
    moves, lens, _ = G.synth_boards(n, 0)
    lens = np.minimum(lens, 4).astype(np.int32)
    planes = G.moves_to_planes(moves, lens)
    last = np.stack([moves[np.arange(n), lens - 1], moves[np.arange(n), lens - 2]], 1).astype(np.int16)
    stats = []
    for network in (net, fused):
        tree = G.AlphaZeroMCTS(n, node_capacity=playouts * 225 + 1)
        tree.set_roots(planes, last)
        with torch.no_grad():
            tree.search(network, playouts)
        stats.append(tree.root_stats())
        tree.close()
    assert (stats[0]["root_visits"] == stats[1]["root_visits"]).all()
    assert (stats[0]["visits"] == stats[1]["visits"]).mean() > 0.99
    # the same search with the playout step replayed from a hipGraph: identical trees
    tree = G.AlphaZeroMCTS(n, node_capacity=playouts * 225 + 1)
    tree.set_roots(planes, last)
    with torch.no_grad():
        tree.search(fused, playouts, graph=True)
    replayed = tree.root_stats()
    tree.close()
    for k in ("visits", "root_visits", "n_nodes"):
        assert (replayed[k] == stats[1][k]).all()
    assert (replayed["values"].view(np.uint32) == stats[1]["values"].view(np.uint32)).all()
    assert np.abs(stats[0]["root_value"] - stats[1]["root_value"]).max() < 1e-4
    fused.close()


def test_eval_state_for_agents():
    """FusedPolicyValueNetwork.eval_state(board) is what PyConvNetAgent calls per playout (agents/alphazero.py:5-9)."""
    from gomokuai_amd import core
    import helpers
    G.init()
    net = PolicyValueNetwork(seed=2).cuda().eval()
    fused = FusedPolicyValueNetwork(net)
    b = core.Board()
    for mv in (112, 113, 97):
        b.apply_move(core.Position(mv))
    v0, p0 = net.eval_state(b)
    v1, p1 = fused.eval_state(b)
    assert abs(v0 - v1) < TOL and np.abs(p0 - p1).max() < TOL and abs(float(p1.sum()) - 1) < 1e-4
    agent = helpers.network_searcher(fused.eval_state, 5.0, c_iterations=30)
    move = agent.move(b)
    assert b.check_move(move)
    fused.close()


@pytest.mark.parametrize("n", [1, 15, 16, 17, 100, 1029, 4096])
def test_dense_layers_in_hip(n):
    """gmk_pvnet_evaluate = trunk kernel + dense kernel: value / probabilities against the module and against PyTorch's dense layers on the
    kernel's own trunk outputs (the tighter check: only the second kernel differs), at batch sizes around the kernel's 16-position workgroups;
    every row of probabilities sums to one, the value lies in [-1, 1]."""
    G.init()
    net = PolicyValueNetwork(seed=11).cuda().eval()
    with torch.no_grad():
        for m in net.modules():
            if hasattr(m, "bias") and m.bias is not None:
                m.bias.uniform_(-0.3, 0.3)
        net.policy_dense.weight.mul_(60.0)                        # logits of a few units: a softmax that is far from uniform
    fused = FusedPolicyValueNetwork(net)
    torch.manual_seed(100 + n)
    states = (torch.rand((n, 6, 15, 15), device="cuda") > 0.6).float()
    with torch.no_grad():
        rvalue, rprobs = net(states)
    dvalue, dprobs = fused.dense_reference(states)
    value, probs = fused(states)
    assert value.shape == (n,) and probs.shape == (n, 225)
    assert float((value - dvalue).abs().max()) < 2e-6 and float((probs - dprobs).abs().max()) < 5e-6
    assert float((value - rvalue).abs().max()) < TOL and float((probs - rprobs).abs().max()) < TOL
    assert float((probs.sum(1) - 1).abs().max()) < 1e-5 and float(value.abs().max()) <= 1.0
    assert float(probs.max()) > 2.0 / 225, float(probs.max())       # not vacuous: the distribution has structure
    # a second, smaller batch through the same handle (the scratch between the kernels is reused), then a larger one (it grows)
    for m in (max(1, n // 3), 2 * n + 5):
        s2 = (torch.rand((m, 6, 15, 15), device="cuda") > 0.5).float()
        v2, p2 = fused(s2)
        dv2, dp2 = fused.dense_reference(s2)
        assert float((v2 - dv2).abs().max()) < 2e-6 and float((p2 - dp2).abs().max()) < 5e-6
    fused.close()


def test_evaluate_needs_the_dense_layers():
    """gmk_pvnet_evaluate before gmk_pvnet_set_dense is an error, not a guess."""
    import ctypes as C
    G.init()
    net = PolicyValueNetwork(seed=1).cuda().eval()
    host = lambda t: np.ascontiguousarray(t.detach().float().cpu().numpy())
    arrays = [host(net.conv[0].weight), host(net.conv[0].bias), host(net.conv[1].weight), host(net.conv[1].bias), host(net.conv[2].weight), host(net.conv[2].bias),
              host(net.policy_conv.weight).reshape(4, 128), host(net.policy_conv.bias), host(net.value_conv.weight).reshape(2, 128), host(net.value_conv.bias)]
    h = C.c_void_p()
    G._check(G.load().gmk_pvnet_create(*[a.ctypes.data for a in arrays], C.byref(h)))
    states = torch.zeros((2, 6, 15, 15), device="cuda")
    value, probs = torch.empty(2, device="cuda"), torch.empty((2, 225), device="cuda")
    rc = G.load().gmk_pvnet_evaluate(h, states.data_ptr(), 2, value.data_ptr(), probs.data_ptr(), None)
    assert rc != 0 and b"gmk_pvnet_set_dense" in G.load().gmk_last_error()
    G.load().gmk_pvnet_destroy(h)


def test_errors():
    G.init()
    net = PolicyValueNetwork(seed=1).cuda().eval()
    fused = FusedPolicyValueNetwork(net)
    with pytest.raises(AssertionError):
        fused.trunk(torch.zeros((2, 6, 15, 15), device="cuda", dtype=torch.float64))
    p, v = fused.trunk(torch.zeros((0, 6, 15, 15), device="cuda"))
    assert p.shape == (0, 900) and v.shape == (0, 450)
    fused.close()
