"""CPU: bench.py's plumbing around the measured legs -- the one JSON line is printed whatever a leg next to `value` does
(exception, a CPU baseline that does not come back), and the CPU-baseline worker runs from a file without a GPU."""
import importlib.util
import json
import os
import sys
import types

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _load_bench():
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


class _FakeLoop:
    description, rays, exchange = "fake workload", 8192, None

    def __init__(self, workload, args, dev, rank, world, dist):
        self.model, self.data = object(), object()
        self.trainer = types.SimpleNamespace(global_step=545)

    def run(self, setup_steps, warmup, steps, min_timed=200):
        return dict(rays_per_s=1.0e7, ms_per_step=0.8192, timed_windows=10, timed_steps_total=10 * steps, window_ms_per_step_min_max=[0.8, 0.83],
                    ms_per_step_hip_events=0.81, cold_start={"ms_per_step": 2.0, "window": "steps [5, 25)"},
                    metrics=dict(rm_s=40.0, vr_s=20.0, psnr=25.0, loss=0.01))


def test_the_line_is_printed_whatever_the_side_legs_do(monkeypatch, capfd, tmp_path):
    monkeypatch.setenv("NGP_BENCH_DETAIL", str(tmp_path / "bench_detail.json"))
    b = _load_bench()
    import ngp_pl_amd.bench_support as support
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "1", "--steps", "20", "--warmup", "5", "--secondary"])
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        monkeypatch.delenv(k, raising=False)
    monkeypatch.setattr(torch.cuda, "is_available", lambda: True)
    monkeypatch.setattr(torch.cuda, "set_device", lambda d: None)
    monkeypatch.setattr(torch.cuda, "empty_cache", lambda: None)
    monkeypatch.setattr(b, "Loop", _FakeLoop)
    monkeypatch.setattr(b, "kernel_roofline", lambda loop, ms=None: {"bound": "hbm", "achieved": 1.0, "peak": 8000.0, "unit": "GB/s", "frac": 1.25e-4, "traffic": None})
    calls = []

    def fps(model, data, n_frames, chunk_scale=1, probe_cap=0):
        calls.append(chunk_scale)
        if chunk_scale == 1:
            raise RuntimeError("frame loop failed")
        return {"fps": 200.0}
    monkeypatch.setattr(support, "render_fps", fps)

    def api(loop):
        raise ValueError("autograd leg failed")
    monkeypatch.setattr(b, "api_path_rate", api)
    monkeypatch.setattr(b, "cpu_baseline", lambda model, data: {"value": None, "unit": "rays/s", "cores": 8, "kind": "port", "sample": "not measured in this run: timeout"})

    def secondary(name, args, dev, late_steps=0, roofline=True):
        if name == "unbounded":
            raise RuntimeError("out of memory")
        return {"workload": name, "rays_per_s": 2.0e7, "ms_per_step": 0.4, "global_step_at_end": 430, "samples_per_ray_composited": 30.0,
                "samples_per_ray_marched": 50.0, "train_psnr": 20.0}
    monkeypatch.setattr(b, "secondary_line", secondary)
    saved = os.dup(1)
    try:
        b.main()
    finally:
        os.dup2(saved, 1); os.close(saved)          # main() points fd 1 at stderr for everything but the line
    out, _ = capfd.readouterr()
    lines = [ln for ln in out.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out
    assert len(lines[0]) < 4096                     # the driver keeps a tail of stdout: the line must fit (BENCH_r05's 22 KB did not)
    line = json.loads(lines[0])
    assert line["value"] == 1.0e7 and line["steps"] == 20 and line["warmup"] == 5 and line["n_gpus"] == 1 and line["roofline"]["bound"] == "hbm"
    assert line["cpu_baseline"]["kind"] == "port" and line["render_fps_800x800_regrouped"] == 200.0
    assert {"render_fps_800x800", "api_path", "api_path_reference_files"} <= set(line["legs_failed"])       # failed legs are named in the line
    with open(tmp_path / "bench_detail.json") as f:
        d = json.load(f)                            # every leg's full record: next to the script
    assert d["value"] == 1.0e7 and d["steps"] == 20 and d["warmup"] == 5 and d["n_gpus"] == 1 and d["roofline"]["bound"] == "hbm"
    # `render_fps_800x800` is the reference's protocol and chunking (it fails here); the regrouped loop is the extra
    assert d["render_fps_800x800_regrouped"]["fps"] == 200.0 and "field_state" in d["render_fps_800x800_regrouped"] and calls == [1, 4]
    assert "frame loop failed" in d["render_fps_800x800"]["error"]
    assert "random-init" not in d["data"] and "TRAINED inside this run" in d["data"]
    assert "error" in d["api_path_reference_files"]                     # (a fake loop cannot drive it: recorded, not raised)
    pts = d["sensitivity"]["points"]
    assert [p_["live_samples_per_ray"] for p_ in pts] == sorted(p_["live_samples_per_ray"] for p_ in pts) and len(pts) == 4
    assert d["api_path"]["error"].startswith("ValueError")
    assert d["cpu_baseline"]["value"] is None and d["cpu_baseline"]["kind"] == "port"
    assert "out of memory" in d["secondary"][0]["error"] and d["secondary"][1]["rays_per_s"] == 2.0e7
    assert "roofline" in d and "builder_profile" not in d          # (builder-recorded profile data only ever sits under builder_* keys)


def _run_bench_with_fakes(tmp_path, body, deadline):
    """bench.py's main() in a child process with the measured pieces replaced by `body` (python source defining the fakes)."""
    import subprocess
    script = tmp_path / "drive.py"
    script.write_text("""
import importlib.util, os, sys, time, types, torch
ROOT = %r
sys.path.insert(0, ROOT)
spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)
import ngp_pl_amd.bench_support as support
torch.cuda.is_available = lambda: True
torch.cuda.set_device = lambda d: None
torch.cuda.empty_cache = lambda: None
class FakeLoop:
    description, rays, exchange = "fake workload", 8192, None
    def __init__(self, workload, args, dev, rank, world, dist):
        self.model, self.data = object(), object()
        self.trainer = types.SimpleNamespace(global_step=545)
    def run(self, setup_steps, warmup, steps, min_timed=200):
        return dict(rays_per_s=1.0e7, ms_per_step=0.8192, timed_windows=10, timed_steps_total=10 * steps, window_ms_per_step_min_max=[0.8, 0.83],
                    ms_per_step_hip_events=0.81, cold_start=None, metrics=dict(rm_s=40.0, vr_s=20.0, psnr=25.0, loss=0.01))
b.Loop = FakeLoop
b.kernel_roofline = lambda loop, ms=None: {"bound": "hbm", "frac": 0.1}
b.cpu_baseline = lambda model, data: {"value": 100.0, "unit": "rays/s", "cores": 8, "kind": "port", "sample": "fake"}
support.render_fps = lambda model, data, n_frames, **kw: {"fps": 200.0}
b.api_path_rate = lambda loop: {"rays_per_s": 1.0}
%s
sys.argv = ["bench.py", "--gpus", "1", "--steps", "20", "--warmup", "5", "--no-secondary", "--deadline", "%g"]
b.main()
""" % (ROOT, body, deadline))
    return subprocess.run([sys.executable, str(script)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=120,
                          env=dict(os.environ, NGP_BENCH_DETAIL=str(tmp_path / "bench_detail.json")))


def _detail(tmp_path):
    with open(tmp_path / "bench_detail.json") as f:
        return json.load(f)


def test_a_leg_that_never_returns_costs_its_budget_not_the_line(tmp_path):
    """A leg that spins for ever (a hung kernel under a host poll): the watchdog prints the line with the headline, the legs that
    were complete and the timeout recorded for the hung one, dumps the stacks to stderr, and the process ends with status 0."""
    body = """
def spin(model, data, n_frames, **kw):
    if kw.get("chunk_scale", 1) == 1:
        while True:
            pass
    return {"fps": 200.0}
support.render_fps = spin
"""
    import time
    t = time.perf_counter()
    r = _run_bench_with_fakes(tmp_path, body, deadline=25.0)
    assert time.perf_counter() - t < 60
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert r.returncode == 0 and len(lines) == 1, (r.stdout, r.stderr)
    line = json.loads(lines[0])
    assert len(lines[0]) < 4096
    assert line["value"] == 1.0e7 and line["roofline"]["frac"] == 0.1 and line["cpu_baseline"]["value"] == 100.0 and "timeout" in line["error"]
    assert {"render_fps_800x800", "render_fps_800x800_regrouped", "api_path"} <= set(line["legs_failed"])
    d = _detail(tmp_path)
    assert d["value"] == 1.0e7 and d["roofline"]["frac"] == 0.1 and d["cpu_baseline"]["value"] == 100.0
    assert "timeout in render_fps_800x800" in d["render_fps_800x800"]["error"]
    assert d["render_fps_800x800_regrouped"]["error"].startswith("not run: timeout")
    assert d["api_path"]["error"].startswith("not run: timeout") and "timeout" in d["error"]
    assert "stacks of all threads" in r.stderr and "in spin" in r.stderr           # faulthandler names the spinning frame


def test_a_hang_before_the_headline_still_prints_one_line_and_fails(tmp_path):
    body = """
def never(self, setup_steps, warmup, steps, min_timed=200):
    while True:
        pass
FakeLoop.run = never
"""
    r = _run_bench_with_fakes(tmp_path, body, deadline=12.0)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert r.returncode == 1 and len(lines) == 1, (r.stdout, r.stderr)
    d = json.loads(lines[0])
    assert d["value"] is None and "timeout in setup" in d["error"] and d["steps"] == 20


def test_cpu_baseline_runs_in_a_child_process_and_is_bounded():
    """cpu_baseline(): inputs to a file, oracle in a child process (no GPU in there), JSON back; a child that does not finish in
    time leaves value null and the reason in `sample`."""
    b = _load_bench()
    from ngp_pl_amd.bench_support import GpuDataset
    from ngp_pl_amd.networks import NGP
    torch.manual_seed(0)
    m = NGP(scale=0.5)
    data = GpuDataset(64, 4, "cpu", seed=0)
    assert 1 <= b.usable_cpus() <= (os.cpu_count() or 1)
    r = b.cpu_baseline(m, data, budget_s=1.0, timeout_s=240)          # empty occupancy grid: the steps find no samples
    assert r["kind"] == "port" and r["unit"] == "rays/s" and r["value"] > 0 and r["cores"] == min(b.usable_cpus(), 32)
    m.density_bitfield.fill_(255)
    r = b.cpu_baseline(m, data, budget_s=1.0, timeout_s=240)          # full grid: real steps through the oracle
    assert r["value"] > 0 and "full steps of 256 rays" in r["sample"]
    r = b.cpu_baseline(m, data, budget_s=30.0, timeout_s=2.0)
    assert r["value"] is None and "did not finish" in r["sample"] and r["kind"] == "port"


def test_lego_hard_ground_truth_is_the_volume_rendering_of_its_scene():
    """bench_support.volumetric_ground_truth (sorted entry / exit events, inside-count sweep, per-segment transmittance-weighted colour)
    against brute-force quadrature (24 000 steps per ray) of the same piecewise-constant density on random pixels: the scene of the
    `sensitivity` leg is a consistent volumetric scene (mean colour error 3e-4 at sigma 60, 1.3e-3 at sigma 15), it stays inside the [-0.5, 0.5] box at
    both sizes the bench uses, and a lower density lets rays through (more live samples per ray is what the leg is for)."""
    from ngp_pl_amd import bench_support as B, synthetic as syn
    for size in (1.0, 1.2):
        boxes, spheres = B.lego_hard_scene(size)
        assert max(abs(c[i]) + h[i] for c, h in boxes for i in range(3)) < 0.5 and max(abs(c[i]) + r for c, r in spheres for i in range(3)) < 0.5
    boxes, spheres = B.lego_hard_scene()
    assert len(boxes) > 100                                   # studs, tread bars, cabin walls ...
    K = syn.intrinsics(96)
    dirs = syn.get_ray_directions(96, 96, K)
    pose = syn.hemisphere_poses(2, seed=4)[1]
    ro, rd = syn.get_rays(dirs, pose)
    pick = torch.randperm(ro.shape[0], generator=torch.Generator().manual_seed(0))[:160]
    o, d = ro[pick].contiguous(), rd[pick].contiguous()
    opacities = {}
    for sigma in (60.0, 15.0):
        got = B.volumetric_ground_truth(o, d, boxes, spheres, sigma)
        n, tn, tf = 24000, 0.6, 2.6                              # cameras sit at radius 1.5: the box lies in t = [0.6, 2.6]; a 4 mm bar = 48 steps
        t = tn + (tf - tn) * (torch.arange(n) + 0.5) / n
        x = o[:, None] + t[None, :, None] * d[:, None]
        inside = torch.zeros(x.shape[:-1], dtype=torch.bool)
        for c, h in boxes:
            inside |= ((x - x.new_tensor(c)).abs() <= x.new_tensor(h)).all(-1)
        for c, r in spheres:
            inside |= (x - x.new_tensor(c)).norm(dim=-1) <= r
        a = 1 - torch.exp(-sigma * inside.float() * ((tf - tn) / n))
        T = torch.cumprod(torch.cat([torch.ones_like(a[:, :1]), 1 - a[:, :-1]], 1), 1)
        w = a * T
        dn = d / d.norm(dim=-1, keepdim=True)
        want = (w[..., None] * syn.colour(x, dn[:, None].expand_as(x))).sum(1) + (1 - w.sum(1))[:, None]
        err = (got - want).abs().max(1).values
        assert float(err.mean()) < 2e-3 and float(err.quantile(0.95)) < 1.5e-2, (sigma, float(err.mean()), float(err.quantile(0.95)))
        opacities[sigma] = float(w.sum(1).mean())
    assert opacities[15.0] < opacities[60.0]


def test_reference_files_loader_leaves_the_interpreter_as_it_found_it():
    """oracle/ref_on_binding.load(): the reference's modules come up over the two aliases and the aliases are gone again afterwards
    (`import vren` of another test must not find this package's binding by accident); its NGP has the product NGP's state dict."""
    import sys
    from oracle import ref_on_binding as R
    if not R.available():
        pytest.skip("the reference's models/*.py are neither mounted nor staged")
    before = {k: sys.modules.get(k) for k in ("vren", "tinycudann", "torch_scatter", "kornia", "models", "losses")}
    mods = R.load()
    assert {k: sys.modules.get(k) for k in before} == before
    assert mods.rendering.MAX_SAMPLES == 1024 and mods.networks.NGP.__module__ == "models.networks"
    from ngp_pl_amd.networks import NGP
    theirs, ours = R.make_model(0.5, "cpu"), NGP(scale=0.5)
    ours.register_training_buffers()
    assert list(theirs.state_dict()) == list(ours.state_dict())
    assert all(a.shape == b.shape and a.dtype == b.dtype for a, b in zip(theirs.state_dict().values(), ours.state_dict().values()))
