"""Deterministic (numpy RandomState) weights for the style-net fixtures, shared by make_golden.py and the tests
so that the 3.5 M-parameter encoder/decoder weights need not be stored."""
import numpy as np
import torch


def fill_style_weights(seq, seed):
    rs = np.random.RandomState(seed)
    for m in seq.modules():
        if isinstance(m, torch.nn.Conv2d):
            fan_in = m.weight.shape[1] * m.weight.shape[2] * m.weight.shape[3]
            w = rs.standard_normal(m.weight.shape).astype(np.float32) * np.float32(np.sqrt(2.0 / fan_in))
            b = rs.standard_normal(m.bias.shape).astype(np.float32) * np.float32(0.05)
            with torch.no_grad():
                m.weight.copy_(torch.from_numpy(w))
                m.bias.copy_(torch.from_numpy(b))
