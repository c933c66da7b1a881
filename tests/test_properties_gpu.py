"""GPU, BASELINE.json full sizes (one 800x800 frame = 640 000 rays; 8192-ray batches): size-independent
properties of the HIP path -- packing/sortedness, determinism, recomputation identities, linearity,
gradient checksums -- where the CPU oracle would take minutes."""
import ctypes as C
import math

import numpy as np
import pytest
import torch

from ngp_pl_amd import synthetic as syn

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def frame():
    K = syn.intrinsics(800)
    dirs = syn.get_ray_directions(800, 800, K, device="cuda")
    pose = syn.hemisphere_poses(3, seed=5)[1].cuda()
    ro, rd = syn.get_rays(dirs, pose)
    grid = syn.analytic_density_grid(1, 0.5, 128)
    bf = torch.from_numpy(syn.pack_bitfield_np(grid, 10.0)).cuda()
    return ro, rd, bf


def occupied(bf, xyz, scale=0.5, G=128):
    """bit lookup of raymarching.cu:214-220 for cascade 0 in torch."""
    n = (0.5 * (xyz / scale + 1) * G).clamp(0, G - 1).to(torch.int64)
    def ex(v):
        v = (v * 0x00010001) & 0xFF0000FF; v = (v * 0x00000101) & 0x0F00F00F
        v = (v * 0x00000011) & 0xC30C30C3; v = (v * 0x00000005) & 0x49249249
        return v
    idx = ex(n[:, 0]) | (ex(n[:, 1]) << 1) | (ex(n[:, 2]) << 2)
    return ((bf[idx >> 3].to(torch.int64) >> (idx & 7)) & 1).bool()


def test_full_frame_march_properties(frame):
    import ngp_pl_amd.vren as vren
    ro, rd, bf = frame
    n = ro.shape[0]
    assert n == 640000
    _, hits_t, _ = vren.ray_aabb_intersect(ro, rd, torch.zeros(1, 3).cuda(), torch.full((1, 3), 0.5).cuda(), 1)
    ht = hits_t[:, 0].contiguous()
    ht[(ht[:, 0] >= 0) & (ht[:, 0] < 0.01), 0] = 0.01
    noise = torch.rand(n, device="cuda")
    out1 = vren.raymarching_train(ro, rd, ht, bf, 1, 0.5, 0.0, noise, 128, 1024)
    out2 = vren.raymarching_train(ro, rd, ht, bf, 1, 0.5, 0.0, noise, 128, 1024)
    for a, b in zip(out1, out2):                      # deterministic, ray-ordered packing (the reference's is atomics-ordered)
        assert torch.equal(a, b)
    rays_a, xyzs, dirs, deltas, ts, counter = out1
    S = int(counter[0])
    assert int(counter[1]) == n and xyzs.shape[0] == S and S > 1_000_000
    cnt = rays_a[:, 2]
    assert torch.equal(rays_a[:, 0], torch.arange(n, device="cuda"))
    assert torch.equal(rays_a[:, 1], torch.cumsum(cnt, 0) - cnt) and int(cnt.sum()) == S     # exclusive scan = packing offsets
    assert int(cnt.max()) <= 1024 and int(cnt[ht[:, 0] < 0].sum()) == 0                       # misses emit nothing
    ray = torch.repeat_interleave(torch.arange(n, device="cuda"), cnt)
    # recomputation identities, bit for bit: xyz = fma(t, d, o), dir = d, delta = sqrt3/1024
    exp_xyz = torch.addcmul(ro[ray].double(), ts[:, None].double(), rd[ray].double()).float()   # fma == correctly rounded o + t*d
    assert torch.equal(xyzs, exp_xyz)
    assert torch.equal(dirs, rd[ray])
    assert torch.equal(deltas, torch.full_like(deltas, np.float32(1.73205080757) / np.float32(1024)))
    # samples lie inside their ray's [t1, t2) and strictly increase along the ray
    assert bool((ts >= ht[ray, 0]).all()) and bool((ts < ht[ray, 1]).all())
    same = ray[1:] == ray[:-1]
    assert bool((ts[1:][same] > ts[:-1][same]).all())
    assert bool(occupied(bf, xyzs).all())              # every emitted sample sits in an occupied cell
    # test-time marching visits the same samples when it starts unjittered: first sample of each ray
    h2 = ht.clone()
    alive = torch.arange(n, device="cuda")
    x2, d2, de2, t2, ne = vren.raymarching_test(ro, rd, h2, alive, bf, 1, 0.5, 0.0, 128, 1024, 2)
    assert bool(occupied(bf, x2.view(-1, 3)[(d2.view(-1, 3) != 0).any(1)]).all())
    assert bool((h2[:, 0] >= ht[:, 0]).all()) and int(ne.max()) <= 2


def test_composite_linearity_and_closed_form(frame):
    import ngp_pl_amd.vren as vren
    ro, rd, bf = frame
    sel = torch.randperm(ro.shape[0], device="cuda")[:8192]
    ro, rd = ro[sel].contiguous(), rd[sel].contiguous()
    _, hits_t, _ = vren.ray_aabb_intersect(ro, rd, torch.zeros(1, 3).cuda(), torch.full((1, 3), 0.5).cuda(), 1)
    rays_a, xyzs, dirs, deltas, ts, _ = vren.raymarching_train(ro, rd, hits_t[:, 0].contiguous(), bf, 1, 0.5, 0.0,
                                                               torch.rand(8192, device="cuda"), 128, 1024)
    S = ts.shape[0]
    sig = torch.rand(S, device="cuda") ** 2 * 60
    c1, c2 = torch.rand(S, 3, device="cuda"), torch.rand(S, 3, device="cuda")
    o1 = vren.composite_train_fw(sig, c1, deltas, ts, rays_a, 1e-4)
    o2 = vren.composite_train_fw(sig, c2, deltas, ts, rays_a, 1e-4)
    o3 = vren.composite_train_fw(sig, 0.3 * c1 + 0.7 * c2, deltas, ts, rays_a, 1e-4)
    assert torch.allclose(o3[3], 0.3 * o1[3] + 0.7 * o2[3], atol=2e-6)            # rgb is linear in the sample colours
    assert torch.equal(o1[1], o2[1]) and torch.equal(o1[4], o2[4])                 # opacity / weights do not depend on them
    assert torch.allclose(o1[1], o1[4].new_zeros(8192).index_add_(0, torch.repeat_interleave(torch.arange(8192, device="cuda"), rays_a[:, 2]), o1[4]), atol=1e-5)   # opacity = sum of weights
    assert bool((o1[1] <= 1 + 1e-5).all()) and bool((o1[4] >= 0).all())
    # constant sigma: opacity = 1 - exp(-sigma * sum(delta)) for rays that do not hit the early stop
    o4 = vren.composite_train_fw(torch.full_like(sig, 2.0), c1, deltas, ts, rays_a, 0.0)
    want = 1 - torch.exp(-2.0 * float(deltas[0]) * rays_a[:, 2].float())
    assert torch.allclose(o4[1], want, atol=2e-5)
    assert torch.equal(o4[0], rays_a[:, 2])                                        # threshold 0 never stops: all samples counted


def test_hashgrid_linearity_and_gradient_checksum():
    from ngp_pl_amd import _lib
    from ngp_pl_amd._lib import call, ptr, stream
    meta = _lib.GridMeta()
    call("ngp_grid_meta_init", C.byref(meta), 16, 2, 19, 16, float(math.exp(math.log(2048 * 0.5 / 16) / 15)))
    total = meta.offset[16]
    S = 300000
    x = (torch.rand(S, 3, device="cuda") - 0.5).contiguous()
    mn = torch.full((3,), -0.5, device="cuda"); mx = torch.full((3,), 0.5, device="cuda")

    def enc(table):
        f = torch.empty(16, S, 2, dtype=torch.float16, device="cuda")
        call("ngp_hashgrid_fwd", ptr(x), ptr(mn), ptr(mx), ptr(table), C.byref(meta), S, ptr(f), stream())
        return f.float()
    t1 = (torch.rand(total, 2, device="cuda") - 0.5).half(); t2 = (torch.rand(total, 2, device="cuda") - 0.5).half()
    assert torch.allclose(enc((t1.float() + t2.float()).half()), enc(t1) + enc(t2), atol=3e-3)     # linear in the table (f16 rounding)
    ones = enc(torch.full((total, 2), 0.25, device="cuda").half())
    assert torch.allclose(ones, torch.full_like(ones, 0.25), atol=1e-3)                            # trilinear weights sum to 1
    # checksum of the scatter: sum over a level's gradient entries == sum over samples of that level's feature gradient
    g = (torch.randn(16, S, 2, device="cuda") * 0.01).half()
    grad = torch.empty(total, 2, dtype=torch.float16, device="cuda")
    call("ngp_hashgrid_bwd_sliced", ptr(x), ptr(mn), ptr(mx), ptr(g), C.byref(meta), S, None, None, ptr(grad), stream())
    for l in range(16):
        got = grad[meta.offset[l]:meta.offset[l + 1]].double().sum(0)
        want = g[l].double().sum(0)
        scale = g[l].double().abs().sum(0)
        assert bool(((got - want).abs() <= 2e-3 * scale).all()), (l, got, want)


def test_adam_matches_torch_adam():
    """ngp_adam_step == torch.optim.Adam(eps=1e-15) (= apex FusedAdam's update, train.py:131), incl.
    gradient unscale, f16 copy and gradient zeroing; skipped when found_inf is set."""
    from ngp_pl_amd._lib import call, ptr, stream
    n = 1_000_003
    g = torch.Generator(device="cuda").manual_seed(0)
    p0 = torch.randn(n, device="cuda", generator=g)
    ref = p0.clone().requires_grad_(True)
    opt = torch.optim.Adam([ref], lr=1e-2, eps=1e-15)
    p = p0.clone(); ph = torch.empty(n, dtype=torch.float16, device="cuda")
    m = torch.zeros(n, device="cuda"); v = torch.zeros(n, device="cuda")
    for step in range(1, 4):
        grad = torch.randn(n, device="cuda", generator=g) * 10 ** float(step - 3)
        grad16 = (grad * 128).half()
        ref.grad = grad16.float() / 128
        opt.step()
        call("ngp_adam_step", ptr(p), ptr(ph), ptr(grad16), 0, ptr(m), ptr(v), n, 1e-2, 0.9, 0.999, 1e-15, 0.0, step, 128.0, None, stream())
        assert torch.allclose(p, ref.detach(), rtol=1e-5, atol=1e-6)
        assert torch.equal(ph, p.half()) and int((grad16 != 0).sum()) == 0            # f16 working copy refreshed, gradient consumed
    flag = torch.ones(1, dtype=torch.int32, device="cuda")
    before = p.clone()
    g2 = torch.ones(n, device="cuda")
    call("ngp_adam_step", ptr(p), ptr(ph), ptr(g2), 1, ptr(m), ptr(v), n, 1e-2, 0.9, 0.999, 1e-15, 0.0, 4, 1.0, ptr(flag), stream())
    assert torch.equal(p, before) and int((g2 != 0).sum()) == 0                        # GradScaler semantics: skip but clear


def test_adam_field_launch_is_bit_identical_to_the_three_separate_launches():
    """ngp_adam_step_field (grid table + both MLP blocks in one launch) == ngp_adam_step on the grid and
    ngp_adam_step_partials on each MLP block: every parameter, moment, working copy and the zeroed gradient."""
    from ngp_pl_amd._lib import call, ptr, stream
    DEV = "cuda"
    torch.manual_seed(5)
    n_grid, n_d, n_r, rows = 1_000_003, 3072, 7168, 37        # grid size not a multiple of 4: the scalar tail is covered too

    def state(n):
        return [torch.randn(n, device=DEV) * 0.1, torch.zeros(n, dtype=torch.float16, device=DEV),
                torch.rand(n, device=DEV) * 1e-3, torch.rand(n, device=DEV) * 1e-6]          # param, param_h, m, v
    grid, dens, rgb = state(n_grid), state(n_d), state(n_r)
    grad = (torch.randn(n_grid, device=DEV) * 0.3).half()
    grad[::3] = 0
    pd = torch.randn(rows, n_d, device=DEV); pr = torch.randn(rows, n_r, device=DEV)
    ref = [[t.clone() for t in s] for s in (grid, dens, rgb)]
    ref_grad = grad.clone()
    hyper = (1e-2, 0.9, 0.999, 1e-15, 0.0, 7, 128.0, None, stream())
    g, d, r = ref
    call("ngp_adam_step_partials", ptr(d[0]), ptr(d[1]), ptr(pd), rows, ptr(d[2]), ptr(d[3]), n_d, *hyper)
    call("ngp_adam_step", ptr(g[0]), ptr(g[1]), ptr(ref_grad), 0, ptr(g[2]), ptr(g[3]), n_grid, *hyper)
    call("ngp_adam_step_partials", ptr(r[0]), ptr(r[1]), ptr(pr), rows, ptr(r[2]), ptr(r[3]), n_r, *hyper)
    keep = [[t.clone() for t in s] for s in (grid, dens, rgb)]
    grad_keep = grad.clone()
    call("ngp_adam_step_field", ptr(grid[0]), ptr(grid[1]), ptr(grad), ptr(grid[2]), ptr(grid[3]), n_grid,
         ptr(dens[0]), ptr(dens[1]), ptr(pd), ptr(dens[2]), ptr(dens[3]), n_d,
         ptr(rgb[0]), ptr(rgb[1]), ptr(pr), ptr(rgb[2]), ptr(rgb[3]), n_r, rows, *hyper[:7], 1, hyper[7], None, hyper[8])
    torch.cuda.synchronize()
    for got, want in zip((grid, dens, rgb), ref):
        for a, b in zip(got, want):
            assert torch.equal(a, b)
    assert torch.equal(grad, ref_grad) and not grad.any()
    # zero_grid_grad = 0 (what the trainer passes: its table backwards overwrite the gradient): same update, gradient left alone
    g2, d2, r2 = keep
    before = grad_keep.clone()
    call("ngp_adam_step_field", ptr(g2[0]), ptr(g2[1]), ptr(grad_keep), ptr(g2[2]), ptr(g2[3]), n_grid,
         ptr(d2[0]), ptr(d2[1]), ptr(pd), ptr(d2[2]), ptr(d2[3]), n_d,
         ptr(r2[0]), ptr(r2[1]), ptr(pr), ptr(r2[2]), ptr(r2[3]), n_r, rows, *hyper[:7], 0, hyper[7], None, hyper[8])
    torch.cuda.synchronize()
    for got, want in zip(keep, ref):
        for a, b in zip(got, want):
            assert torch.equal(a, b)
    assert torch.equal(grad_keep, before) and bool(grad_keep.any())
    assert torch.equal(grid[1], grid[0].half()) and bool(grid[1].any())    # the update happened and refreshed the working copy


def test_adam_step_count_lives_next_to_the_skip_flag():
    """apex / GradScaler leave the optimizer's step count unchanged on a skipped step.  The skip flag is a device word, so the count of
    APPLIED steps is one too (`step_state`): three calls with flags (0, 1, 0) must equal TWO plain calls (host steps 1, 2) on the
    first and third gradient -- the bias correction of the third call is that of step 2, not 3 -- and the counts read back 2."""
    from ngp_pl_amd._lib import call, ptr, stream
    DEV = "cuda"
    torch.manual_seed(6)
    n_grid, n_d, n_r, rows = 40_000, 3072, 7168, 5

    def state(n):
        return [torch.randn(n, device=DEV) * 0.1, torch.zeros(n, dtype=torch.float16, device=DEV), torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)]
    grads = [((torch.randn(n_grid, device=DEV) * 0.3).half(), torch.randn(rows, n_d, device=DEV), torch.randn(rows, n_r, device=DEV)) for _ in range(3)]

    def run(calls, with_state):
        torch.manual_seed(7)
        grid, dens, rgb = state(n_grid), state(n_d), state(n_r)
        st = torch.zeros(4, dtype=torch.int32, device=DEV) if with_state else None
        flags = [torch.tensor([f], dtype=torch.int32, device=DEV) for f in (0, 1)]
        for k, (gi, flag) in enumerate(calls):
            g, pd, pr = grads[gi]
            call("ngp_adam_step_field", ptr(grid[0]), ptr(grid[1]), ptr(g.clone()), ptr(grid[2]), ptr(grid[3]), n_grid,
                 ptr(dens[0]), ptr(dens[1]), ptr(pd), ptr(dens[2]), ptr(dens[3]), n_d,
                 ptr(rgb[0]), ptr(rgb[1]), ptr(pr), ptr(rgb[2]), ptr(rgb[3]), n_r, rows, 1e-2, 0.9, 0.999, 1e-15, 0.0, k + 1, 128.0, 0,
                 ptr(flags[flag]) if with_state else None, ptr(st), stream())
        torch.cuda.synchronize()
        return grid, dens, rgb, st
    a = run([(0, 0), (1, 1), (2, 0)], True)
    b = run([(0, 0), (2, 0)], False)
    for got, want in zip(a[:3], b[:3]):
        for x, y in zip(got, want):
            torch.testing.assert_close(x, y, rtol=2e-6, atol=1e-9)
    st = a[3].tolist()
    assert st[3 & 1] == 2 and st[2 + (3 & 1)] == 2, st          # slot (calls & 1) of each pair holds the applied count
    # ... and WITHOUT the device count the third call would have corrected for step 3: a visibly different update
    c = run([(0, 0), (2, 0)], False)
    call_step3 = run([(0, 0), (1, 1), (2, 0)], True)
    assert torch.equal(c[0][0], b[0][0]) and torch.equal(call_step3[0][0], a[0][0])      # deterministic


@pytest.mark.parametrize("n_rays", [1, 777, 8192, 20000])
def test_fused_composite_loss_launch_equals_the_three_separate_launches(n_rays):
    """ngp_composite_train_fw_loss == ngp_composite_train_fw + ngp_active_scan + ngp_nerf_loss: composited outputs,
    offsets, live count and backward seeds bit for bit; the two scalar sums to float rounding (different, but
    fixed, summation order)."""
    from ngp_pl_amd._lib import call, lib, ptr, stream
    dev = "cuda"
    g = torch.Generator(device=dev).manual_seed(n_rays)
    counts = torch.randint(0, 90, (n_rays,), device=dev, generator=g)
    counts[::7] = 0                                                       # rays that miss everything
    S = int(counts.sum())
    rays_a = torch.stack([torch.arange(n_rays, device=dev), torch.cumsum(counts, 0) - counts, counts], 1).contiguous()
    sigmas = torch.rand(S, device=dev, generator=g) * 60; rgbs = torch.rand(S, 3, device=dev, generator=g)
    deltas = torch.full((S,), 1.7e-3, device=dev); ts = torch.rand(S, device=dev, generator=g) + 0.5
    gt = torch.rand(n_rays, 3, device=dev, generator=g); bg = torch.ones(3, device=dev)
    f32 = dict(dtype=torch.float32, device=dev)

    def outputs():
        return dict(total=torch.empty(n_rays, dtype=torch.int64, device=dev), opacity=torch.empty(n_rays, **f32),
                    depth=torch.empty(n_rays, **f32), rgb=torch.empty(n_rays, 3, **f32), ws=torch.empty(S, **f32),
                    offs=torch.empty(n_rays, dtype=torch.int32, device=dev), n_active=torch.empty(1, dtype=torch.int32, device=dev),
                    stats=torch.empty(2, **f32), d_rgb=torch.empty(n_rays, 3, **f32), d_o=torch.empty(n_rays, **f32))
    a = outputs()
    call("ngp_composite_train_fw", ptr(sigmas), ptr(rgbs), ptr(deltas), ptr(ts), ptr(rays_a), 1e-4, n_rays, S, ptr(a["total"]),
         ptr(a["opacity"]), ptr(a["depth"]), ptr(a["rgb"]), ptr(a["ws"]), ptr(a["offs"]), stream())
    call("ngp_active_scan", ptr(a["offs"]), n_rays, ptr(a["n_active"]), stream())
    call("ngp_nerf_loss", ptr(a["rgb"]), ptr(a["opacity"]), ptr(gt), ptr(bg), 1e-3, 128.0, n_rays, ptr(a["stats"]), ptr(a["stats"][1:]),
         ptr(a["d_rgb"]), ptr(a["d_o"]), stream())
    nbytes = lib().ngp_composite_train_fw_loss_workspace_bytes(n_rays)
    assert 8 * n_rays <= nbytes <= 8 * n_rays + 24
    wsp = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    for _ in range(2):
        b = outputs()
        call("ngp_composite_train_fw_loss", ptr(sigmas), ptr(rgbs), ptr(deltas), ptr(ts), ptr(rays_a), 1e-4, n_rays, S, ptr(b["total"]),
             ptr(b["opacity"]), ptr(b["depth"]), ptr(b["rgb"]), ptr(b["ws"]), ptr(b["offs"]), ptr(b["n_active"]), ptr(gt), ptr(bg),
             1e-3, 128.0, ptr(b["stats"]), ptr(b["stats"][1:]), ptr(b["d_rgb"]), ptr(b["d_o"]), ptr(wsp), nbytes, stream())
        torch.cuda.synchronize()
        for k in a:
            if k == "stats":
                assert torch.allclose(a[k], b[k], rtol=1e-5, atol=0), (a[k], b[k])
            else:
                assert torch.equal(a[k], b[k]), k


@pytest.mark.parametrize("n_rays", [1, 3, 777, 8192, 20000])
def test_composite_pair_without_the_scan_kernel_equals_the_pair_with_it(n_rays):
    """ngp_composite_train_fw_loss_counts + ngp_composite_train_bw_tail (the backward's workgroups prefix the rows' live counts,
    the last one writes n_active, the first adds the loss terms) == ngp_composite_train_fw_loss + ngp_composite_train_bw: per-ray
    outputs, seeds, sample gradients, the ACTIVE LIST and the positions copied in list order, n_active (device + pinned host word)
    bit for bit; loss and squared-error sums to float rounding (another fixed summation order)."""
    from ngp_pl_amd._lib import call, lib, ptr, stream
    dev = "cuda"
    g = torch.Generator(device=dev).manual_seed(100 + n_rays)
    counts = torch.randint(0, 90, (n_rays,), device=dev, generator=g)
    counts[::7] = 0
    if n_rays == 3:
        counts[:] = torch.tensor([5, 0, 70], device=dev)
    S = max(int(counts.sum()), 1)
    if int(counts.sum()) == 0:
        counts[0] = 1
    rays_a = torch.stack([torch.arange(n_rays, device=dev), torch.cumsum(counts, 0) - counts, counts], 1).contiguous()
    sigmas = torch.rand(S, device=dev, generator=g) * 60; rgbs = torch.rand(S, 3, device=dev, generator=g)
    deltas = torch.full((S,), 1.7e-3, device=dev); ts = torch.rand(S, device=dev, generator=g) + 0.5
    xyzs = torch.rand(S, 3, device=dev, generator=g)
    gt = torch.rand(n_rays, 3, device=dev, generator=g); bg = torch.ones(3, device=dev)
    zeros = torch.zeros(n_rays, device=dev)
    f32 = dict(dtype=torch.float32, device=dev); i32 = dict(dtype=torch.int32, device=dev)
    nbytes = lib().ngp_composite_train_fw_loss_workspace_bytes(n_rays)

    def outputs():
        return dict(total=torch.empty(n_rays, dtype=torch.int64, device=dev), opacity=torch.empty(n_rays, **f32),
                    depth=torch.empty(n_rays, **f32), rgb=torch.empty(n_rays, 3, **f32), ws=torch.empty(S, **f32),
                    offs=torch.empty((n_rays + 3) // 4 * 4, **i32), n_active=torch.full((1,), -7, **i32),
                    stats=torch.empty(2, **f32), d_rgb=torch.empty(n_rays, 3, **f32), d_o=torch.empty(n_rays, **f32),
                    ds=torch.empty(S, **f32), dc=torch.empty(S, 3, **f32), active=torch.full((S,), -1, **i32), x_act=torch.full((S, 3), -1.0, **f32),
                    wsp=torch.empty(nbytes, dtype=torch.uint8, device=dev))
    a, b = outputs(), outputs()
    host = torch.full((4,), -1, dtype=torch.int32).pin_memory()
    call("ngp_composite_train_fw_loss", ptr(sigmas), ptr(rgbs), ptr(deltas), ptr(ts), ptr(rays_a), 1e-4, n_rays, S, ptr(a["total"]),
         ptr(a["opacity"]), ptr(a["depth"]), ptr(a["rgb"]), ptr(a["ws"]), ptr(a["offs"]), ptr(a["n_active"]), ptr(gt), ptr(bg),
         1e-3, 128.0, ptr(a["stats"]), ptr(a["stats"][1:]), ptr(a["d_rgb"]), ptr(a["d_o"]), ptr(a["wsp"]), nbytes, stream())
    call("ngp_composite_train_bw", ptr(a["d_o"]), ptr(zeros), ptr(a["d_rgb"]), None, ptr(sigmas), ptr(rgbs), ptr(a["ws"]), ptr(deltas), ptr(ts),
         ptr(rays_a), ptr(a["opacity"]), ptr(a["depth"]), ptr(a["rgb"]), 1e-4, n_rays, S, ptr(a["ds"]), ptr(a["dc"]), ptr(a["offs"]),
         ptr(a["active"]), ptr(xyzs), ptr(a["x_act"]), stream())
    call("ngp_composite_train_fw_loss_counts", ptr(sigmas), ptr(rgbs), ptr(deltas), ptr(ts), ptr(rays_a), 1e-4, n_rays, S, ptr(b["total"]),
         ptr(b["opacity"]), ptr(b["depth"]), ptr(b["rgb"]), ptr(b["ws"]), ptr(b["offs"]), ptr(gt), ptr(bg), 1e-3, 128.0,
         ptr(b["d_rgb"]), ptr(b["d_o"]), ptr(b["wsp"]), nbytes, stream())
    call("ngp_composite_train_bw_tail", ptr(b["d_o"]), ptr(zeros), ptr(b["d_rgb"]), None, ptr(sigmas), ptr(rgbs), ptr(b["ws"]), ptr(deltas), ptr(ts),
         ptr(rays_a), ptr(b["opacity"]), ptr(b["depth"]), ptr(b["rgb"]), 1e-4, n_rays, S, ptr(b["ds"]), ptr(b["dc"]), ptr(b["offs"]),
         ptr(b["active"]), ptr(xyzs), ptr(b["x_act"]), ptr(b["n_active"]), host.data_ptr(), ptr(b["stats"]), ptr(b["stats"][1:]), ptr(b["wsp"]), nbytes, stream())
    torch.cuda.synchronize()
    n_act = int(a["n_active"])
    assert n_act == int(b["n_active"]) == int(host[0]) and 0 < n_act <= S
    # the counts the forward left are the differences of the offsets the scan kernel wrote
    offs = a["offs"][:n_rays].long(); cnt = b["offs"][:n_rays].long()
    assert torch.equal(torch.cumsum(cnt, 0) - cnt, offs)
    for k in ("total", "opacity", "depth", "rgb", "ws", "d_rgb", "d_o", "ds", "dc"):
        assert torch.equal(a[k], b[k]), k
    assert torch.equal(a["active"][:n_act], b["active"][:n_act]) and torch.equal(a["x_act"][:n_act], b["x_act"][:n_act])
    assert torch.allclose(a["stats"], b["stats"], rtol=1e-5, atol=0), (a["stats"], b["stats"])


def test_native_get_rays_matches_the_torch_statement():
    """ngp_get_rays (one launch) vs datasets/ray_utils.py:50-74 written with torch ops: origins exact, directions to float rounding
    of a 3-term dot product; tensors that want a gradient keep the differentiable torch path."""
    from ngp_pl_amd import synthetic as syn
    dirs = syn.get_ray_directions(123, 77, syn.intrinsics(123, 77), device="cuda")
    pose = syn.hemisphere_poses(3, seed=4)[1].cuda()
    ro, rd = syn.get_rays(dirs, pose)
    want_d = dirs @ pose[:, :3].T
    assert ro.shape == rd.shape == dirs.shape and ro.is_contiguous() and rd.is_contiguous()
    assert torch.equal(ro, pose[:, 3].expand_as(dirs)) and float((rd - want_d).abs().max()) <= 2e-6 * float(want_d.abs().max())
    p2 = pose.clone().requires_grad_(True)
    ro2, rd2 = syn.get_rays(dirs, p2)
    (rd2.sum() + ro2.sum()).backward()
    assert p2.grad is not None and float(p2.grad.abs().max()) > 0


def test_fast_stream_query_follows_torchs_current_stream():
    """_lib.stream() asks torch's C entry point for the raw handle of the current stream (0.3 us instead of the 12 us of
    torch.cuda.current_stream().cuda_stream): it must follow `torch.cuda.stream(...)` contexts like the public call."""
    from ngp_pl_amd import _lib
    assert _lib.stream() == torch.cuda.current_stream().cuda_stream
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        assert _lib.stream() == s.cuda_stream == torch.cuda.current_stream().cuda_stream
    assert _lib.stream() == torch.cuda.current_stream().cuda_stream != s.cuda_stream


@pytest.mark.parametrize("world,n_chunks", [(1, 1), (2, 2), (8, 1), (8, 4), (5, 3)])
def test_piece_adam_of_every_rank_equals_the_whole_table_adam(world, n_chunks):
    """`ngp_adam_step_field_pieces` for EVERY rank of a world that does not exist on the test box: the gradient is cut into the pieces
    a reduce-scatter of n_chunks chunks would deliver to each rank, every rank's launch updates its pieces of one shared copy of
    (master, moments, f16 table), and the result must equal ONE whole-table `ngp_adam_step_field` bit for bit -- piece boundaries,
    the short last pieces behind the end of the table and the untouched padding included."""
    from ngp_pl_amd._lib import call, ptr, stream
    DEV = "cuda"
    torch.manual_seed(60 + world + n_chunks)
    n_grid, n_d, n_r, rows = 16 * 6151, 3072, 7168, 3                     # n_grid a multiple of 16, not of world x n_chunks x 8
    piece = -(-n_grid // (n_chunks * world * 8)) * 8
    chunk, padded = world * piece, n_chunks * world * piece

    def state(n, pad=0):
        return [torch.randn(n + pad, device=DEV) * 0.1, torch.zeros(n + pad, dtype=torch.float16, device=DEV),
                torch.rand(n + pad, device=DEV) * 1e-3, torch.rand(n + pad, device=DEV) * 1e-6]
    grid = state(n_grid, padded - n_grid)
    for t in grid:
        t[n_grid:] = 0                                                   # the padding behind the table
    dens, rgb = state(n_d), state(n_r)
    grad = torch.zeros(padded, dtype=torch.float16, device=DEV)
    grad[:n_grid] = (torch.randn(n_grid, device=DEV) * 0.3).half()
    pd = torch.randn(rows, n_d, device=DEV); pr = torch.randn(rows, n_r, device=DEV)
    ref = [[t.clone() for t in s] for s in (grid, dens, rgb)]
    hyper = (1e-2, 0.9, 0.999, 1e-15, 0.0, 5, 128.0)
    g, d, r = ref
    call("ngp_adam_step_field", ptr(g[0]), ptr(g[1]), ptr(grad.clone()), ptr(g[2]), ptr(g[3]), n_grid, ptr(d[0]), ptr(d[1]), ptr(pd), ptr(d[2]), ptr(d[3]), n_d,
         ptr(r[0]), ptr(r[1]), ptr(pr), ptr(r[2]), ptr(r[3]), n_r, rows, *hyper, 0, None, None, stream())
    for rank in range(world):
        shard = torch.cat([grad[c * chunk + rank * piece:c * chunk + (rank + 1) * piece] for c in range(n_chunks)]).contiguous()
        dd = [[t.clone() for t in s] for s in (dens, rgb)] if rank else [dens, rgb]        # (every rank updates the MLP blocks: keep rank 0's)
        call("ngp_adam_step_field_pieces", ptr(grid[0]), ptr(grid[1]), ptr(shard), ptr(grid[2]), ptr(grid[3]), n_grid, piece, n_chunks, world, rank,
             ptr(dd[0][0]), ptr(dd[0][1]), ptr(pd), ptr(dd[0][2]), ptr(dd[0][3]), n_d, ptr(dd[1][0]), ptr(dd[1][1]), ptr(pr), ptr(dd[1][2]), ptr(dd[1][3]), n_r,
             rows, *hyper, None, None, None, stream())
    torch.cuda.synchronize()
    for got, want in zip((grid, dens, rgb), ref):
        for a, b in zip(got, want):
            assert torch.equal(a, b)
    assert all(not bool(t[n_grid:].any()) for t in grid)


def test_overflow_guard_skips_a_step_with_a_non_finite_weight_gradient():
    """The native step's GradScaler (train.py:274 runs the reference under Lightning's precision=16: a step whose gradients hold an
    inf / NaN is skipped): with the loss seeds scaled past f16's range the field backward's weight-gradient sums are not finite, the
    device flag goes up and the optimizer launch changes NOTHING -- parameters, f16 working copies, both moments -- and does not
    advance the bias-correction count; the next ordinary step applies as step 1.  (Found in round 6: 30 000 steps on the lego_hard
    scene ended in NaN weights around step 25 600 -- one overflowing sample poisons every weight of both networks for good.)"""
    from ngp_pl_amd import synthetic as syn
    from ngp_pl_amd.networks import NGP
    from ngp_pl_amd.trainer import Trainer
    torch.manual_seed(3)
    dev = torch.device("cuda")
    K = syn.intrinsics(64)
    dirs = syn.get_ray_directions(64, 64, K)
    poses = syn.hemisphere_poses(4, seed=1)
    g = torch.Generator().manual_seed(5)
    ro, rd = syn.get_rays(dirs[torch.randint(4096, (2048,), generator=g)], poses[torch.randint(4, (2048,), generator=g)])
    ro, rd = ro.to(dev).contiguous(), rd.to(dev).contiguous()
    gt, _ = syn.render_ground_truth(ro, rd, n_steps=64)
    gt = gt.contiguous()

    def state(tr, m):
        enc, net = m.xyz_encoder, m.rgb_net
        parts = [enc.params.detach(), net.params.detach(), enc._half.get(enc.params), net._half.get(net.params)]
        parts += list(tr.opt.moments("enc")) + list(tr.opt.moments("rgb"))
        return [p.clone() for p in parts]

    m = NGP(scale=0.5).to(dev)
    tr = Trainer(m, loss_scaler=False)                    # (the fixed scale: the dynamic one is the next test's)
    tr.step(ro, rd, gt)                                   # an ordinary step: applied
    assert tr.skipped_steps() == (0, 0)
    before = state(tr, m)
    tr.grad_scale = 1e30                                  # GradScaler-style factor on the loss seeds: every f16 gradient overflows
    tr.step(ro, rd, gt)
    tr.grad_scale = 1.0
    after = state(tr, m)
    assert tr.opt.t == 2 and tr.skipped_steps() == (1, 1)
    for a, b in zip(before, after):
        assert torch.equal(a, b)                          # nothing moved, nothing became NaN
    tr.step(ro, rd, gt)                                   # back to normal: applies, as the SECOND applied step
    assert tr.skipped_steps() == (1, 1) and tr.opt.applied_steps() == (2, 2)
    moved = state(tr, m)
    assert not torch.equal(moved[0], after[0]) and all(bool(torch.isfinite(t.float()).all()) for t in moved)
    # the same two applied steps without the skipped one in between: identical parameters (the skipped step left no trace)
    torch.manual_seed(3)
    m2 = NGP(scale=0.5).to(dev)
    tr2 = Trainer(m2, loss_scaler=False)
    tr2.step(ro, rd, gt); tr2.step(ro, rd, gt)
    # (the march's jitter is drawn per march: the third march of `tr` differs from the second of `tr2`, so compare moments' finiteness
    # and the bias-correction count only)
    assert tr2.opt.applied_steps() == (2, 2)


def _scaler_batch(dev, n=2048):
    from ngp_pl_amd import synthetic as syn
    K = syn.intrinsics(64)
    dirs = syn.get_ray_directions(64, 64, K)
    poses = syn.hemisphere_poses(4, seed=1)
    g = torch.Generator().manual_seed(5)
    ro, rd = syn.get_rays(dirs[torch.randint(4096, (n,), generator=g)], poses[torch.randint(4, (n,), generator=g)])
    ro, rd = ro.to(dev).contiguous(), rd.to(dev).contiguous()
    gt, _ = syn.render_ground_truth(ro, rd, n_steps=64)
    return ro, rd, gt.contiguous()


def test_dynamic_loss_scale_equals_the_same_static_scale_bit_for_bit():
    """The device-side loss scaler (GradScaler's rule on top of tiny-cuda-nn's 128, train.py:274 under precision=16) at a scale that
    neither grows nor overflows is the SAME arithmetic as that factor applied statically through `grad_scale`: the field backward
    multiplies its f32 seeds by a power of two, the optimizer divides by it -- parameters, f16 copies and moments after 12 steps are
    bit-identical.  And both differ from the fixed 128 alone (gradients that flushed to zero there survive)."""
    from ngp_pl_amd.networks import NGP
    from ngp_pl_amd.trainer import Trainer
    dev = torch.device("cuda")
    ro, rd, gt = _scaler_batch(dev)

    def run(**kw):
        torch.manual_seed(3)
        m = NGP(scale=0.5).to(dev)
        tr = Trainer(m, **kw)
        for _ in range(12):
            tr.step(ro, rd, gt)
        enc, net = m.xyz_encoder, m.rgb_net
        return tr, [p.clone() for p in (enc.params.detach(), net.params.detach(), enc._half.get(enc.params), net._half.get(net.params),
                                         *tr.opt.moments("enc"), *tr.opt.moments("rgb"))]
    tr_d, dyn = run(loss_scaler=dict(init_scale=256.0, growth_interval=10 ** 9))
    tr_s, sta = run(loss_scaler=False, grad_scale=256.0)
    tr_f, fix = run(loss_scaler=False)
    assert tr_d.skipped_steps() == (0, 0) and tr_s.skipped_steps() == (0, 0)
    assert tr_d.loss_scale_state() == (256.0, 12) and tr_s.loss_scale_state() == (1.0, 0)
    for a, b in zip(dyn, sta):
        assert torch.equal(a, b)
    assert not torch.equal(dyn[0], fix[0])


def test_loss_scaler_backs_off_on_overflow_and_grows_after_clean_steps():
    """GradScaler's rule, decided on the device: from a scale that must overflow (2^40 on top of 128) every step is skipped and halves
    the scale until the f16 chain fits; then `growth_interval` clean steps double it, the doubled scale overflows again or not, and so
    on.  Checked against the rule replayed on the host from the skip counts; no parameter ever goes non-finite."""
    from ngp_pl_amd.networks import NGP
    from ngp_pl_amd.trainer import Trainer
    dev = torch.device("cuda")
    ro, rd, gt = _scaler_batch(dev)
    torch.manual_seed(3)
    m = NGP(scale=0.5).to(dev)
    tr = Trainer(m, loss_scaler=dict(init_scale=2.0 ** 40, growth_interval=3))
    scale, tracker, skipped = 2.0 ** 40, 0, 0
    seen = []
    # ... and against torch's own GradScaler fed the same skip decisions (a one-element parameter whose gradient is inf on the steps
    # the device skipped): the scale the reference's Lightning run would be at
    theirs = torch.amp.GradScaler("cuda", init_scale=2.0 ** 40, growth_interval=3)
    dummy = torch.nn.Parameter(torch.zeros(1, device=dev))
    dummy_opt = torch.optim.SGD([dummy], lr=0.0)
    for i in range(60):
        tr.step(ro, rd, gt)
        now = tr.skipped_steps()[0]
        was_skipped = now > skipped
        if was_skipped:                         # this step was skipped: backoff
            scale, tracker = scale * 0.5, 0
        else:
            tracker += 1
            if tracker >= 3:
                scale, tracker = scale * 2.0, 0
        skipped = now
        assert tr.loss_scale_state() == (scale, tracker), (i, tr.loss_scale_state(), scale, tracker)
        dummy.grad = torch.full((1,), float("inf") if was_skipped else 1.0, device=dev)
        theirs.scale(torch.zeros((), device=dev))           # (initialises the scaler's device state on first use)
        theirs.step(dummy_opt)
        theirs.update()
        assert theirs.get_scale() == scale, (i, theirs.get_scale(), scale)
        seen.append(scale)
    assert skipped >= 10 and tr.opt.applied_steps()[0] == 60 - skipped          # it took many halvings to come down from 2^40 ...
    assert min(seen) < 2.0 ** 30 and seen[-1] >= min(seen)                       # ... and it came back up afterwards
    assert any(b > a for a, b in zip(seen, seen[1:]))                            # (growth happened)
    enc, net = m.xyz_encoder, m.rgb_net
    assert bool(torch.isfinite(enc.params).all()) and bool(torch.isfinite(net.params).all())
