"""Deterministic weights for the width-128 / 256 fixtures (tests/golden/make_encoder_wide_golden.py): every tensor of a
state_dict is a pure function of its NAME and shape, so the generator (which fills the REFERENCE's modules) and the tests (which
fill gcc_amd's) agree without the 3 MB of initial weights in the fixture.  Own code, not the reference's initialisation: the
reference's nn.Linear / nn.Embedding defaults depend on construction order; the values here have the same scales."""
import zlib

import torch


def tensor_for(name, like, salt=0):
    g = torch.Generator().manual_seed((zlib.crc32(name.encode()) + 7919 * salt) & 0x7FFFFFFF)
    if not like.dtype.is_floating_point:                     # num_batches_tracked
        return torch.zeros_like(like)
    if name.endswith(".eps"):                                # GINConv's eps buffer (learn_eps=False): 0
        return torch.zeros_like(like)
    if name.endswith("running_mean"):
        return torch.zeros(like.shape) + 0.05 * torch.randn(like.shape, generator=g)
    if name.endswith("running_var"):
        return 1.0 + 0.2 * torch.rand(like.shape, generator=g)
    if like.dim() >= 2:                                      # Linear / Embedding weights: uniform(+-1 / sqrt(fan_in))
        bound = 1.0 / float(like.shape[-1]) ** 0.5
        return (torch.rand(like.shape, generator=g) * 2 - 1) * bound
    if "batch_norm" in name or ".bn." in name or name.endswith("bn.weight") or name.endswith("bn.bias"):
        if name.endswith("weight"):
            return 0.5 + torch.rand(like.shape, generator=g)             # gamma in (0.5, 1.5): its gradient is exercised
        return (torch.rand(like.shape, generator=g) * 2 - 1) * 0.3
    return (torch.rand(like.shape, generator=g) * 2 - 1) * 0.1           # Linear biases


def fill_(module, salt=0):
    sd = module.state_dict()
    module.load_state_dict({k: tensor_for(k, v, salt).to(v.dtype) for k, v in sd.items()})
    return module
