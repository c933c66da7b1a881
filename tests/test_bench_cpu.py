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


def test_the_line_is_printed_whatever_the_side_legs_do(monkeypatch, capfd):
    b = _load_bench()
    import ngp_pl_amd.bench_support as support
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "1", "--steps", "20", "--warmup", "5"])
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        monkeypatch.delenv(k, raising=False)
    monkeypatch.setattr(torch.cuda, "is_available", lambda: True)
    monkeypatch.setattr(torch.cuda, "set_device", lambda d: None)
    monkeypatch.setattr(torch.cuda, "empty_cache", lambda: None)
    monkeypatch.setattr(b, "Loop", _FakeLoop)
    monkeypatch.setattr(b, "kernel_roofline", lambda loop: {"bound": "hbm", "achieved": 1.0, "peak": 8000.0, "unit": "GB/s", "frac": 1.25e-4, "traffic": None})
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

    def secondary(name, args, dev):
        if name == "unbounded":
            raise RuntimeError("out of memory")
        return {"workload": name, "rays_per_s": 2.0e7}
    monkeypatch.setattr(b, "secondary_line", secondary)
    saved = os.dup(1)
    try:
        b.main()
    finally:
        os.dup2(saved, 1); os.close(saved)          # main() points fd 1 at stderr for everything but the line
    out, _ = capfd.readouterr()
    lines = [ln for ln in out.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out
    d = json.loads(lines[0])
    assert d["value"] == 1.0e7 and d["steps"] == 20 and d["warmup"] == 5 and d["n_gpus"] == 1 and d["roofline"]["bound"] == "hbm"
    assert d["render_fps_800x800"]["fps"] == 200.0 and "field_state" in d["render_fps_800x800"] and calls == [4, 1]
    assert "frame loop failed" in d["render_fps_800x800_reference_chunking"]["error"]
    assert d["api_path"]["error"].startswith("ValueError")
    assert d["cpu_baseline"]["value"] is None and d["cpu_baseline"]["kind"] == "port"
    assert "out of memory" in d["secondary"][0]["error"] and d["secondary"][1]["rays_per_s"] == 2.0e7


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
    assert r["value"] > 0 and "full training steps of 256 rays" in r["sample"]
    r = b.cpu_baseline(m, data, budget_s=30.0, timeout_s=2.0)
    assert r["value"] is None and "did not finish" in r["sample"] and r["kind"] == "port"
