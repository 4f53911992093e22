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
