"""GPU: bench.py's contract on a shortened run -- one JSON line with the fields the driver and the judge read."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LINE_LIMIT = 4096          # the driver keeps a tail of stdout: BENCH_r05's 22 KB line came back unparsed


def _run_bench(argv, tmp_path, timeout, env=None):
    """bench.py as the driver runs it; returns (completed process, the ONE line parsed, the detail record it left beside it)."""
    detail = str(tmp_path / "bench_detail.json")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + argv, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                       timeout=timeout, env=dict(env or os.environ, NGP_BENCH_DETAIL=detail))
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1 and lines[0].startswith("{"), r.stdout[-2000:]
    assert len(lines[0]) < LINE_LIMIT, len(lines[0])
    with open(detail) as f:
        return r, json.loads(lines[0]), json.load(f)


def test_bench_prints_one_json_line_with_roofline_and_cold_start(tmp_path):
    _, line, d = _run_bench(["--gpus", "1", "--steps", "10", "--warmup", "3", "--setup-steps", "48", "--images", "8", "--res", "200",
                             "--no-cpu-baseline", "--no-render"], tmp_path, 600)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "cold_start", "timed_windows", "timed_steps_total", "api_path"):
        assert k in d, k
    # the line carries the contract fields and the roofline as an object; everything else is one scalar per leg
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "api_path_rays_per_s", "cold_start_rays_per_s"):
        assert k in line, k
    assert abs(line["value"] - d["value"]) < 1e-4 * d["value"] and abs(line["roofline"]["frac"] - d["roofline"]["frac"]) < 1e-4
    assert set(line["roofline"]) >= {"bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "avg_ms"}
    assert d["n_gpus"] == 1 and d["steps"] == 10 and d["warmup"] == 3 and d["unit"] == "rays/s" and d["scaling"] == "weak"
    assert d["timed_windows"] * d["steps"] == d["timed_steps_total"] >= 200
    assert d["value"] > 1e6 and abs(d["value"] - 8192 / (d["ms_per_step"] * 1e-3)) < 1e-3 * d["value"]
    assert d["config"]["setup_steps_untimed"] == 48 and "workload" in d["config"]
    roof = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "samples_active_per_launch", "stages"):
        assert k in roof, k
    assert roof["peak"] == 8000.0 and 0 < roof["frac"] < 1 and abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-9
    assert roof["samples_active_per_launch"] <= roof["samples_marched_per_launch"]
    cold = d["cold_start"]
    assert cold["ms_per_step"] > 0 and cold["window"].startswith("steps [3, 13)")


def test_driver_command_verbatim(tmp_path):
    """The driver's literal command, no skip flags: ONE line on stdout within the time the driver allows, carrying `roofline`
    and `cpu_baseline`, exit status 0.  (Round 2's driver run of this command was killed at 1800 s with an empty stdout.)"""
    import time
    t = time.perf_counter()
    r, line, d = _run_bench(["--gpus", "1", "--steps", "20", "--warmup", "5"], tmp_path, 300)
    wall = time.perf_counter() - t
    assert "error" not in d, (d["error"], r.stderr[-3000:])
    # what the driver parses: value + roofline + cpu_baseline as objects, the three train numbers and both FPS figures as scalars
    assert "error" not in line and "legs_failed" not in line, line
    assert line["value"] > 3e6 and line["steps"] == 20 and line["warmup"] == 5 and line["n_gpus"] == 1 and line["dtype"] == "f16/f32"
    assert 0 < line["roofline"]["frac"] < 1 and line["roofline"]["bound"] == "hbm" and line["roofline"]["avg_ms"] > 0
    assert line["cpu_baseline"]["kind"] in ("port", "reference") and line["cpu_baseline"]["cores"] >= 1
    for k in ("render_fps_800x800", "render_fps_800x800_hard", "psnr", "api_path_rays_per_s", "api_path_plain_rays_per_s",
              "api_path_reference_files_rays_per_s", "configs3_unbounded_rays_per_s", "configs2_16k_rays_per_s", "sensitivity_rays_per_s"):
        assert k in line, (k, line)
    assert "workload" in line["config"] and "model" not in line["config"]
    assert d["value"] > 3e6 and d["steps"] == 20 and d["warmup"] == 5 and d["n_gpus"] == 1
    roof, cpu = d["roofline"], d["cpu_baseline"]
    assert 0 < roof["frac"] < 1 and roof["bound"] == "hbm" and roof["whole_step"]["frac"] > 0
    assert cpu["kind"] in ("port", "reference") and cpu["cores"] >= 1 and (cpu["value"] is None or cpu["value"] > 0)
    for leg in ("render_fps_800x800", "render_fps_800x800_regrouped", "api_path", "api_path_plain", "api_path_reference_files", "full_run",
                "sensitivity"):
        assert "error" not in d[leg], (leg, d[leg])
    assert "reference's chunking" in d["render_fps_800x800"]["loop"]             # the headline FPS is the reference-protocol figure
    ref_files = d["api_path_reference_files"]
    assert ref_files["rays_per_s"] > 1e5 and ref_files["state_dict_missing"] == [] and ref_files["state_dict_unexpected"] == []
    rf = d["render_fps_800x800_reference_files"]                                 # test.ipynb cell 2 around the reference's own rendering.py
    assert "error" not in rf and rf["fps"] > 1 and rf["total_samples"] == rf["total_samples_ngp_render_test_frame"], rf
    assert rf["max_abs_rgb_difference_to_ngp_render_test_frame"] < 2e-3, rf       # (their module path rounds h / SH once more than the fused field)
    sec = d["secondary"]                                                         # configs[3] / configs[2] recipes: on by default
    assert len(sec) == 2 and all("error" not in x and x["rays_per_s"] > 1e6 and 0 < x["roofline"]["frac"] < 1 for x in sec), sec
    assert sec[0]["cascades"] == 6 and sec[1]["rays_per_batch"] == 16384
    pts = d["sensitivity"]["points"]
    assert len(pts) >= 3 and all(p_["rays_per_s"] > 1e6 for p_ in pts)
    assert max(p_["live_samples_per_ray"] for p_ in pts) > 1.5 * d["config"]["samples_per_ray_composited"]      # the hard scene IS harder
    assert "TRAINED inside this run" in d["data"]
    assert wall < 240, wall
    assert "[bench" in r.stderr and "headline complete" in r.stderr              # leg-by-leg progress is on by default


@pytest.mark.parametrize("exchange", ["sharded", "allreduce", "direct"])
def test_bench_under_a_one_rank_rccl_group(exchange, tmp_path):
    """The multi-GPU path of bench.py on a real RCCL process group of ONE rank (what `torchrun --nproc-per-node 1 bench.py` runs):
    process group, parameter broadcast, the gradient exchange installed on the trainer (sharded: reduce-scatter -> shard Adam ->
    all-gather; or the all-reduce), the timed windows, the exchange stage timed on its own, teardown."""
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
               NGP_DDP_EXCHANGE=exchange, HSA_ENABLE_IPC_MODE_LEGACY="0", NGP_BENCH_BOTH_MODES="1",      # (the other mode's leg, which only runs at world > 1 otherwise)
               NGP_DP_EVAL_STEPS="100")
    _, line, d = _run_bench(["--gpus", "1", "--steps", "10", "--warmup", "3", "--setup-steps", "48", "--images", "8", "--res", "200",
                             "--no-cpu-baseline", "--no-render", "--no-api"], tmp_path, 400, env=env)
    assert "error" not in d, d.get("error")
    # the N-GPU line is the same compact line + what identifies the exchange
    assert line["exchange"] == exchange and line["exchange_impl"] == "native" and line["exchange_ms"] > 0 and line["exposed_exchange_ms"] is not None
    assert line["rccl_ranks"] == 1 and line["exchange_mode"] == exchange and set(line["exchange_modes_ms_per_step"]) == {"sharded", "allreduce", "direct"}
    assert d["n_gpus"] == 1 and d["exchange"] == exchange and d["exchange_ms"] > 0 and d["value"] > 1e6 and d["exchange_impl"] == "native"
    assert d["config"]["train_psnr"] > 10 and d["march_guards"] == [0, 0, 0, 0]
    modes = d["exchange_modes"]
    assert set(modes) == {"sharded", "allreduce", "direct"}, modes
    for other in modes:
        if other != exchange:
            assert "error" not in modes[other], modes
            assert modes[other]["exchange_ms"] > 0 and modes[other]["ms_per_step"] > 0
    assert modes[exchange]["exposed_exchange_ms"] is not None
    # the evaluation sharded over the ranks ran with the exchange installed, and the line says what RCCL itself saw
    ev = d["dp_eval"]
    assert "error" not in ev, ev
    assert ev["rccl_ranks"] == 1 and ev["rccl_rank"] == 0 and ev["rccl_version"] > 20000 and ev["exchange_mode"] == exchange
    assert ev["n_poses"] == 40 and ev["poses_per_rank"] == [40] and len(ev["render_fps_per_gpu"]) == 1 and ev["render_fps_per_gpu"][0] > 1
    assert ev["psnr"] > 10 and ev["render_fps_aggregate"] == ev["render_fps_per_gpu"][0]
