"""PolicyValueNetwork in PyTorch-ROCm: the architecture of the reference's TF1 graph (network/model_tf.py:28-66), for
inference at the leaves of the network-guided search (K7, lib.AlphaZeroMCTS).  Training is out of scope;
weights are randomly initialised like `tf.global_variables_initializer()` does when no checkpoint exists
(model_tf.py:165-172: glorot-uniform kernels, zero biases), or loaded from a state dict.

    inputs   float32 [B, 6, 15, 15]   Board.encoded_states() (core/py_ext/src/game_ext.hpp:87-104)
    shared   conv3x3 'same' + ReLU: 6 -> 32 -> 64 -> 128
    policy   conv1x1 -> 4 + ReLU, flatten, dense 225, softmax
    value    conv1x1 -> 2 + ReLU, flatten, dense 64 + ReLU, dense 1, tanh
The TF graph runs channels-last and flattens (h, w, c); this module keeps NCHW tensors and permutes before the dense
layers, so a TF checkpoint's dense kernels can be loaded without reordering.
"""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F


class PolicyValueNetwork(nn.Module):
    def __init__(self, seed=0):
        super().__init__()
        self.conv = nn.ModuleList([nn.Conv2d(6, 32, 3, padding=1), nn.Conv2d(32, 64, 3, padding=1), nn.Conv2d(64, 128, 3, padding=1)])
        self.policy_conv = nn.Conv2d(128, 4, 1)
        self.policy_dense = nn.Linear(4 * 225, 225)
        self.value_conv = nn.Conv2d(128, 2, 1)
        self.value_hidden = nn.Linear(2 * 225, 64)
        self.value_out = nn.Linear(64, 1)
        gen = torch.Generator().manual_seed(seed)
        for m in self.modules():
            if isinstance(m, (nn.Conv2d, nn.Linear)):
                fan_out = m.weight.shape[0] * (m.weight[0][0].numel() if m.weight.dim() == 4 else 1)
                fan_in = m.weight.shape[1] * (m.weight[0][0].numel() if m.weight.dim() == 4 else 1)
                limit = float(np.sqrt(6.0 / (fan_in + fan_out)))                        # glorot_uniform, TF's default
                with torch.no_grad():
                    m.weight.copy_((torch.rand(m.weight.shape, generator=gen) * 2 - 1) * limit)
                    m.bias.zero_()

    def forward(self, states):
        """states float32 [B, 6, 15, 15] -> (value [B], probs [B, 225])."""
        x = states
        for conv in self.conv:
            x = F.relu(conv(x))
        p = F.relu(self.policy_conv(x)).permute(0, 2, 3, 1).reshape(x.shape[0], -1)     # tf.layers.flatten of an NHWC tensor
        probs = F.softmax(self.policy_dense(p), dim=1)
        v = F.relu(self.value_conv(x)).permute(0, 2, 3, 1).reshape(x.shape[0], -1)
        value = torch.tanh(self.value_out(F.relu(self.value_hidden(v)))).reshape(-1)
        return value, probs

    @torch.no_grad()
    def eval_state(self, board):
        """PolicyValueNetwork.eval_state (model_tf.py:136-145): one position -> (value, probs[225]) on the host."""
        dev = next(self.parameters()).device
        states = torch.from_numpy(np.asarray(board.encoded_states(), dtype=np.float32)[None]).to(dev)
        value, probs = self(states)
        return float(value[0]), probs[0].cpu().numpy()


class FusedPolicyValueNetwork:
    """The same function as PolicyValueNetwork.forward in two HIP kernels on the f32 matrix cores (gmk_pvnet_evaluate): the convolutions (99 % of
    the arithmetic) as ONE fused kernel whose activations never leave LDS (K9), and the three dense layers with softmax / tanh as a second one.
    float32 throughout; sums run in a different order than MIOpen's / rocBLAS's, so outputs agree with the module's to rounding
    (tests/test_pvnet_gpu.py: 2e-5), not bit for bit.  Takes the weights of `net` at construction."""

    def __init__(self, net):
        import ctypes as C
        from . import lib as G
        G.init()
        self.net, self.G, self.h = net, G, C.c_void_p()
        host = lambda t: np.ascontiguousarray(t.detach().float().cpu().numpy())
        arrays = [host(net.conv[0].weight), host(net.conv[0].bias), host(net.conv[1].weight), host(net.conv[1].bias),
                  host(net.conv[2].weight), host(net.conv[2].bias), host(net.policy_conv.weight).reshape(4, 128), host(net.policy_conv.bias),
                  host(net.value_conv.weight).reshape(2, 128), host(net.value_conv.bias)]
        arrays = [np.ascontiguousarray(a) for a in arrays]
        G._check(G.load().gmk_pvnet_create(*[a.ctypes.data for a in arrays], C.byref(self.h)))
        dense = [host(net.policy_dense.weight), host(net.policy_dense.bias), host(net.value_hidden.weight), host(net.value_hidden.bias),
                 host(net.value_out.weight).reshape(64)]
        assert dense[0].shape == (225, 900) and dense[2].shape == (64, 450)
        G._check(G.load().gmk_pvnet_set_dense(self.h, *[a.ctypes.data for a in dense], float(net.value_out.bias.detach().cpu().reshape(-1)[0])))

    def close(self):
        if getattr(self, "h", None) and getattr(self, "G", None) is not None and self.G.load is not None:
            self.G.load().gmk_pvnet_destroy(self.h)
            self.h = None

    __del__ = close

    @torch.no_grad()
    def trunk(self, states):
        """states float32 [B, 6, 15, 15] on the GPU -> (relu(policy conv) [B, 900], relu(value conv) [B, 450]), flattened (pixel, channel)."""
        assert states.is_cuda and states.dtype == torch.float32 and states.is_contiguous() and tuple(states.shape[1:]) == (6, 15, 15)
        n = states.shape[0]
        pflat = torch.empty((n, 900), dtype=torch.float32, device=states.device)
        vflat = torch.empty((n, 450), dtype=torch.float32, device=states.device)
        self.G._check(self.G.load().gmk_pvnet_forward(self.h, states.data_ptr(), n, pflat.data_ptr(), vflat.data_ptr(),
                                                      torch.cuda.current_stream(states.device).cuda_stream))
        return pflat, vflat

    @torch.no_grad()
    def __call__(self, states):
        """states float32 [B, 6, 15, 15] on the GPU -> (value [B], probs [B, 225]); both kernels go to torch's current stream."""
        assert states.is_cuda and states.dtype == torch.float32 and states.is_contiguous() and tuple(states.shape[1:]) == (6, 15, 15)
        n = states.shape[0]
        value = torch.empty((n,), dtype=torch.float32, device=states.device)
        probs = torch.empty((n, 225), dtype=torch.float32, device=states.device)
        self.G._check(self.G.load().gmk_pvnet_evaluate(self.h, states.data_ptr(), n, value.data_ptr(), probs.data_ptr(),
                                                       torch.cuda.current_stream(states.device).cuda_stream))
        return value, probs

    @torch.no_grad()
    def dense_reference(self, states):
        """The dense layers through PyTorch on the kernel's trunk outputs (what __call__ did before the second kernel existed): a check, not a path."""
        pflat, vflat = self.trunk(states)
        probs = F.softmax(self.net.policy_dense(pflat), dim=1)
        value = torch.tanh(self.net.value_out(F.relu(self.net.value_hidden(vflat)))).reshape(-1)
        return value, probs

    @torch.no_grad()
    def eval_state(self, board):
        """PolicyValueNetwork.eval_state (model_tf.py:136-145) through the fused kernel: one position -> (value, probs[225]) on the
        host; what MCTS(policy=Policy(eval_state=network.eval_state, c_puct)) -- the reference's PyConvNetAgent, agents/alphazero.py:5-9 -- calls once per playout."""
        dev = next(self.net.parameters()).device
        states = torch.from_numpy(np.asarray(board.encoded_states(), dtype=np.float32)[None]).to(dev)
        value, probs = self(states)
        return float(value[0]), probs[0].cpu().numpy()
