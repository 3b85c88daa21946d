"""Deterministic, name-keyed weights so the reference network (fixture generation) and the
product network (tests) hold identical parameters without committing megabytes of floats."""
import hashlib

import torch


def fill_deterministic(module, seed=0):
    """Overwrite every parameter/buffer of `module` in place; value depends only on (seed, name, shape)."""
    sd = module.state_dict()
    for name, t in sd.items():
        h = int.from_bytes(hashlib.sha256(("%d:%s" % (seed, name)).encode()).digest()[:7], "little")
        g = torch.Generator().manual_seed(h)
        if not t.is_floating_point():
            continue
        if name.endswith("group_norm.weight") or (name.endswith(".weight") and t.dim() == 1):
            v = 1.0 + 0.2 * torch.randn(t.shape, generator=g)
        elif t.dim() == 1:
            v = 0.1 * torch.randn(t.shape, generator=g)
        else:
            fan_in = t[0].numel()
            v = torch.randn(t.shape, generator=g) / (fan_in ** 0.5)
        sd[name] = v.to(t.dtype)
    module.load_state_dict(sd)
    return module
