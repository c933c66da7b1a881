"""Seeded inputs shared by tests/golden/make_render_golden.py (which runs the reference on them) and
tests/test_reference_python_cpu.py (which runs the oracle on them): nothing here touches the reference."""
import numpy as np
import torch

from oracle import tcnn_oracle as T

CONFIGS = {"syn": dict(scale=0.5, esf=0.0, fill=0.10, n=160), "real": dict(scale=2.0, esf=1 / 256, fill=0.20, n=128)}


def make_field(scale, seed=5):
    """An oracle Field with weights large enough for densities between ~0 and ~1e3: rays saturate at different
    depths, so the test-time loop drops rays and regroups its samples over several iterations."""
    f = T.Field(scale=scale, seed=seed)
    f.table = f.table * 3e4
    f.density_w = f.density_w * 2.0
    return f


def make_rays(n, scale, seed):
    g = torch.Generator().manual_seed(seed)
    o = (torch.rand(n, 3, generator=g) - 0.5) * (3.0 * scale)
    target = (torch.rand(n, 3, generator=g) - 0.5) * scale
    d = target - o
    d = d / d.norm(dim=1, keepdim=True)
    d[: n // 8] = -d[: n // 8]                                             # some rays look away from the box
    return o.contiguous(), d.contiguous()


def field_checksum(f):
    return np.array([float(f.table.double().sum()), float(f.density_w.double().sum()), float(f.rgb_w.double().sum())])
