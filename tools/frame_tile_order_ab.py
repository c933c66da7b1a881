"""Round 5: does the ORDER of the rays of a frame matter to the frame loop?  A wave of the thread-per-ray marcher takes 64 consecutive rays
-- 64 pixels of one image row in the reference's protocol -- and lasts as long as its longest lane.  Here the pixel directions are
permuted once into 8x8 (and 4x16, 16x4) tiles per wave and the same 40 held-out poses are rendered; nothing in the library changes.
Usage: frame_tile_order_ab.py [workload] [steps]"""
import os
import sys
import time
import types

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from ngp_pl_amd import synthetic as syn  # noqa: E402
from ngp_pl_amd.rendering import render  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "lego"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
args = types.SimpleNamespace(rays=0, res=800, images=100)
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
loop = bench.Loop(wl, args, dev, 0, 1, None)
loop.trainer.steps_per_epoch = max(steps // 30, 1)
loop.steps(steps)
poses = syn.hemisphere_poses(40, seed=999).to(dev)
W = H = loop.data.W


def tiled(th, tw):
    idx = torch.arange(H * W, device=dev).view(H // th, th, W // tw, tw).permute(0, 2, 1, 3).reshape(-1)
    return idx


orders = {"row-major (the protocol)": None, "8x8 tiles": tiled(8, 8), "4x16 tiles": tiled(4, 16), "16x4 tiles": tiled(16, 4), "2x32 tiles": tiled(2, 32)}
with torch.no_grad():
    ref = None
    for rnd in range(2):
        for name, perm in orders.items():
            dirs = loop.data.directions if perm is None else loop.data.directions[perm].contiguous()
            ro, rd = syn.get_rays(dirs, poses[0]); out = render(loop.model, ro, rd, test_time=True)
            if perm is None:
                ref = out
            else:
                same = torch.equal(out["rgb"], ref["rgb"][perm]) and int(out["total_samples"]) == int(ref["total_samples"])
            times = []
            for i in range(poses.shape[0]):
                torch.cuda.synchronize(); t = time.perf_counter()
                ro, rd = syn.get_rays(dirs, poses[i]); out = render(loop.model, ro, rd, test_time=True)
                torch.cuda.synchronize(); times.append(time.perf_counter() - t)
            mean = sum(times) / len(times)
            print("round %d  %-26s %.1f fps (%.3f ms)%s" % (rnd, name, 1 / mean, mean * 1e3, "" if perm is None else "  same pixels: %s" % same), flush=True)
