"""TEST INFRASTRUCTURE -- CPU restatement of the reference's `render()` (both branches) and of
`NGP.update_density_grid` given the sampled cells, built from the other oracle pieces:

  * vren kernels : `vren_oracle.Oracle` / `Reference` (plain-C restatement or the reference's own
                   .cu compiled for the host),
  * the field    : `tcnn_oracle.Field` (fp32 torch-CPU restatement of the tiny-cuda-nn modules;
                   PARITY UNPINNED, see its header).

Each function follows the reference file:line it cites.  PINNED: tests/test_reference_python_cpu.py compares these
functions with what the reference's OWN rendering.py / networks.py / custom_functions.py / losses.py produce when they are
run on the CPU over the reference's kernels compiled for the host (tests/golden/ref_harness.py, fixture
tests/golden/render_golden.npz): sample counts, packing, t and dt bit for bit, composited outputs to 2e-6, the occupancy
merge and packed bits bit for bit.  (The tiny-cuda-nn modules were stood in by tcnn_oracle on both sides: that half stays
unpinned.)  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; sizes are meant
to stay small (pure Python loop over iterations, C kernels and torch-CPU inside).
"""
import numpy as np
import torch

MAX_SAMPLES = 1024          # rendering.py:7
NEAR_DISTANCE = 0.01        # rendering.py:8


def _prologue(vr, rays_o, rays_d, scale):
    """rendering.py:27-29: AABB intersection, hits closer than the near plane pushed to it."""
    centre = np.zeros((1, 3), np.float32)
    half = np.full((1, 3), scale, np.float32)
    _, hits_t, _ = vr.ray_aabb_intersect(rays_o, rays_d, centre, half, 1)
    hits_t = hits_t.copy()
    near = (hits_t[:, 0, 0] >= 0) & (hits_t[:, 0, 0] < NEAR_DISTANCE)
    hits_t[near, 0, 0] = NEAR_DISTANCE
    return hits_t


def _field(field, xyzs, dirs, quantize):
    with torch.no_grad():
        s, c, _ = field.forward(torch.from_numpy(xyzs), torch.from_numpy(dirs), quantize=quantize)
    return s.numpy().astype(np.float32), c.numpy().astype(np.float32)


def render_rays_test(vr, field, rays_o, rays_d, density_bitfield, cascades=1, scale=0.5, grid_size=128,
                     exp_step_factor=0.0, T_threshold=1e-4, max_samples=MAX_SAMPLES, quantize=True):
    """`__render_rays_test` (rendering.py:46-118): iterative march / infer / composite over the
    alive rays with N_samples = max(min(N_rays // N_alive, 64), min_samples) (:69-70), rays dropped
    when they return no sample or saturate (:91,105), white background for exp_step_factor == 0
    (:111-116).  Returns opacity, depth, rgb, total_samples, n_iterations."""
    rays_o = np.ascontiguousarray(rays_o, np.float32); rays_d = np.ascontiguousarray(rays_d, np.float32)
    n_rays = rays_o.shape[0]
    hits_t = np.ascontiguousarray(_prologue(vr, rays_o, rays_d, scale)[:, 0])      # (R,2), advanced in place (:84)
    opacity = np.zeros(n_rays, np.float32); depth = np.zeros(n_rays, np.float32); rgb = np.zeros((n_rays, 3), np.float32)
    alive = np.arange(n_rays, dtype=np.int64)
    min_samples = 1 if exp_step_factor == 0 else 4                                  # :60
    samples = total = iterations = 0
    while samples < max_samples:                                                    # :65
        n_alive = len(alive)
        if n_alive == 0:
            break
        n_step = max(min(n_rays // n_alive, 64), min_samples)
        samples += n_step
        iterations += 1
        xyzs, dirs, deltas, ts, n_eff = vr.raymarching_test(rays_o, rays_d, hits_t, alive, density_bitfield, cascades, scale,
                                                            exp_step_factor, grid_size, MAX_SAMPLES, n_step)
        total += int(n_eff.sum())
        flat_x = xyzs.reshape(-1, 3); flat_d = dirs.reshape(-1, 3)
        valid = ~np.all(flat_d == 0, axis=1)                                        # :92
        if valid.sum() == 0:
            break
        sigmas = np.zeros(len(flat_x), np.float32); rgbs = np.zeros((len(flat_x), 3), np.float32)
        s, c = _field(field, flat_x[valid], flat_d[valid], quantize)
        sigmas[valid] = s; rgbs[valid] = c
        alive = np.ascontiguousarray(alive)
        vr.composite_test_fw(sigmas.reshape(n_alive, n_step), rgbs.reshape(n_alive, n_step, 3), deltas, ts, hits_t, alive,
                             T_threshold, n_eff, opacity, depth, rgb)
        alive = alive[alive >= 0]                                                   # :105
    bg = 1.0 if exp_step_factor == 0 else 0.0
    out_rgb = rgb + bg * (1.0 - opacity)[:, None]
    return opacity, depth, out_rgb.astype(np.float32), total, iterations


def render_rays_train(vr, field, rays_o, rays_d, density_bitfield, noise, cascades=1, scale=0.5, grid_size=128,
                      exp_step_factor=0.0, T_threshold=1e-4, quantize=True):
    """`__render_rays_train` (rendering.py:121-163) with the marcher's jitter given explicitly
    (custom_functions.py:83 draws it).  Returns a dict like the reference's."""
    rays_o = np.ascontiguousarray(rays_o, np.float32); rays_d = np.ascontiguousarray(rays_d, np.float32)
    hits_t = np.ascontiguousarray(_prologue(vr, rays_o, rays_d, scale)[:, 0])
    rays_a, xyzs, dirs, deltas, ts, counter = vr.raymarching_train(rays_o, rays_d, hits_t, density_bitfield, cascades, scale,
                                                                   exp_step_factor, noise, grid_size, MAX_SAMPLES)
    if len(xyzs):
        sigmas, rgbs = _field(field, xyzs, dirs, quantize)
    else:
        sigmas = np.zeros(0, np.float32); rgbs = np.zeros((0, 3), np.float32)
    total, opacity, depth, rgb, ws = vr.composite_train_fw(sigmas, rgbs, deltas, ts, rays_a, T_threshold)
    bg = 1.0 if exp_step_factor == 0 else 0.0                                       # :153-161 (no random_bg)
    return {"rgb": rgb + bg * (1.0 - opacity)[:, None], "opacity": opacity, "depth": depth, "ws": ws, "rays_a": rays_a,
            "deltas": deltas, "ts": ts, "rm_samples": int(counter[0]), "vr_samples": int(total.sum()),
            "sigmas": sigmas, "rgbs": rgbs, "xyzs": xyzs, "dirs": dirs}


def update_density_grid(vr, density_grid, cells, sigmas, density_threshold, decay=0.95):
    """The merge half of `NGP.update_density_grid` (networks.py:256-268) for ONE cascade given the
    sampled cell indices and the densities evaluated there: tmp[cells] = sigma; grid = where(grid < 0,
    grid, max(grid * decay, tmp)); threshold = min(mean(grid > 0), density_threshold); pack bits.
    Returns (new grid, bitfield, threshold).  Duplicate cells: the last write wins here; torch's indexed assignment
    leaves the winner unspecified (tests/test_reference_python_cpu.py checks those cells for membership)."""
    grid = np.asarray(density_grid, np.float32).copy()
    tmp = np.zeros_like(grid)
    tmp[np.asarray(cells, np.int64)] = np.asarray(sigmas, np.float32)
    grid = np.where(grid < 0, grid, np.maximum(grid * np.float32(decay), tmp)).astype(np.float32)
    pos = grid[grid > 0]
    mean = float(pos.mean()) if len(pos) else 0.0
    thr = min(mean, density_threshold)
    bitfield = np.zeros(len(grid) // 8, np.uint8)
    vr.packbits(grid, thr, bitfield)
    return grid, bitfield, thr


def mark_invisible_cells(vr, K, poses, img_wh, cascades, grid_size, scale, chunk=64 ** 3):
    """`NGP.mark_invisible_cells` (networks.py:197-238) restated in torch on the CPU: for every cell of every cascade (Morton
    order, `get_all_cells` :155-167) the centre `(coord / (G - 1) * 2 - 1) * (s - s / G)`, s = min(2^(c-1), scale) (:216-221), is
    taken to every camera -- p_cam = R^T (x - t), (u d, v d, d) = K p_cam (:222-226) --; count_grid = fraction of the cameras that
    have the cell inside their image at depth >= NEAR_DISTANCE (:227-231), density_grid = -1 where no camera does or where one
    has it inside its image closer than the near plane, else 0 (:232-238).  Returns (density_grid, count_grid), each (cascades, G^3)
    float32 torch tensors.  `chunk` bounds the temporaries.  Pinned to the reference's own output by tests/test_reference_python_cpu.py."""
    K = torch.as_tensor(K, dtype=torch.float32); poses = torch.as_tensor(poses, dtype=torch.float32)
    n_cams = poses.shape[0]
    n_cells = grid_size ** 3
    coords = torch.from_numpy(vr.morton3D_invert(np.arange(n_cells, dtype=np.int32)).astype(np.float32))      # cell m of the grid sits at Morton index m
    density = torch.zeros(cascades, n_cells); count = torch.zeros(cascades, n_cells)
    world_to_cam = poses[:, :3, :3].transpose(1, 2)
    cam_origin = -world_to_cam @ poses[:, :3, 3:]
    W, H = img_wh
    for c in range(cascades):
        s = min(2 ** (c - 1), scale)
        span = s - s / grid_size
        for lo in range(0, n_cells, chunk):
            centres = ((coords[lo:lo + chunk] / (grid_size - 1) * 2 - 1) * span).T
            proj = K @ (world_to_cam @ centres + cam_origin)                          # (cams, 3, cells): u*d, v*d, d
            depth = proj[:, 2]
            u, v = proj[:, 0] / depth, proj[:, 1] / depth
            inside = (depth >= 0) & (u >= 0) & (u < W) & (v >= 0) & (v < H)
            seen_by = (inside & (depth >= NEAR_DISTANCE)).sum(0) / n_cams
            clipped = (inside & (depth < NEAR_DISTANCE)).any(0)
            count[c, lo:lo + chunk] = seen_by
            density[c, lo:lo + chunk] = torch.where((seen_by > 0) & ~clipped, 0., -1.)
    return density, count
