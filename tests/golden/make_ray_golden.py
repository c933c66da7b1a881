"""Regenerates tests/golden/ray_golden.npz from the reference's OWN ray helpers (datasets/ray_utils.py), imported from
/root/reference.  Runs only where the reference is mounted; the fixture travels with the repo.  kornia is not installed
here: `create_meshgrid(H, W, False)` is provided by a stand-in that returns what kornia documents for
normalized_coordinates=False -- a (1, H, W, 2) grid of (x, y) pixel indices -- nothing else of kornia is used by the
functions exercised here."""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference/datasets/ray_utils.py"


def _kornia_stub():
    m = types.ModuleType("kornia")

    def create_meshgrid(height, width, normalized_coordinates=True, device=None, dtype=torch.float32):
        assert not normalized_coordinates
        xs = torch.arange(width, dtype=dtype, device=device); ys = torch.arange(height, dtype=dtype, device=device)
        gy, gx = torch.meshgrid(ys, xs, indexing="ij")
        return torch.stack([gx, gy], -1).unsqueeze(0)
    m.create_meshgrid = create_meshgrid
    return m


def main():
    sys.modules.setdefault("kornia", _kornia_stub())
    spec = importlib.util.spec_from_file_location("ref_ray_utils", REF)
    ref = importlib.util.module_from_spec(spec); spec.loader.exec_module(ref)
    g = torch.Generator().manual_seed(11)
    H, W = 6, 9
    K = torch.tensor([[11.5, 0, 4.25], [0, 12.25, 3.5], [0, 0, 1]])
    dirs = ref.get_ray_directions(H, W, K)
    c2w = torch.randn(3, 4, generator=g)
    c2w_b = torch.randn(H * W, 3, 4, generator=g)
    o1, d1 = ref.get_rays(dirs, c2w)
    o2, d2 = ref.get_rays(dirs, c2w_b)
    v = torch.randn(7, 3, generator=g) * torch.tensor([1e-4, 1e-2, 0.3, 1.0, 2.0, 3.1, 0.0]).view(7, 1)
    R = ref.axisangle_to_R(v)
    R1 = ref.axisangle_to_R(v[3])
    # train.py:86-89 applied to a gathered copy of the poses
    poses = c2w_b[:7].clone(); dT = torch.randn(7, 3, generator=g) * 0.1
    poses[..., :3] = R @ poses[..., :3]; poses[..., 3] += dT
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ray_golden.npz")
    np.savez(out, H=H, W=W, K=K.numpy(), dirs=dirs.numpy(), c2w=c2w.numpy(), c2w_b=c2w_b.numpy(), o1=o1.numpy(), d1=d1.numpy(),
             o2=o2.numpy(), d2=d2.numpy(), v=v.numpy(), R=R.numpy(), R1=R1.numpy(), dT=dT.numpy(), poses_opt=poses.numpy())
    print("wrote", out)


if __name__ == "__main__":
    main()
