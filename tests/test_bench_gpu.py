"""bench.py contract: one JSON line on stdout with the fields the driver reads (small batch, 2 steps)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

REQUIRED = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
            "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"}


def test_bench_prints_one_json_line_with_roofline_and_cpu_baseline():
    env = dict(os.environ, DCLIP_BENCH_CPU_SECONDS="1")
    p = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--steps", "2", "--warmup", "1", "--batch", "16"],
                       capture_output=True, text=True, timeout=600, env=env, cwd=REPO)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    assert REQUIRED <= set(d), REQUIRED - set(d)
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["unit"] == "images/s"
    assert d["dtype"] == "f32" and d["scaling"] == "weak" and d["higher_is_better"] is True and d["vs_baseline"] is None
    assert abs(d["value"] - 16 * 1000.0 / d["ms_per_step"]) < 0.01 * d["value"]
    r = d["roofline"]
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and 0 < r["frac"] <= 1 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    c = d["cpu_baseline"]
    assert c["kind"] in ("port", "reference") and c["value"] > 0 and c["cores"] >= 1 and "sample" in c
    assert "workload" in d["config"] and "model" not in d["config"]
    if "clock" in d:                 # informative (hwmon of the device, when readable): the clock the roofline's 2.4 GHz got
        assert 90 <= d["clock"]["sclk_mhz"] <= 2600 and abs(r["frac_at_measured_clock"] * d["clock"]["sclk_mhz"] - r["frac"] * 2400.0) < 5.0


def test_bare_gpus_2_self_launches_its_ranks():
    """`python bench.py --gpus 2` with no launcher: bench.py starts the two ranks itself as a child torch.distributed.run
    and relays rank 0's line.  On the one-GPU test box the ranks rendezvous over gloo and share the card
    (DCLIP_DIST_BACKEND=gloo); on a multi-GPU node the same call runs over RCCL."""
    import torch
    env = dict(os.environ, DCLIP_BENCH_CPU_SECONDS="1")
    env.pop("WORLD_SIZE", None), env.pop("RANK", None), env.pop("LOCAL_RANK", None)
    if torch.cuda.device_count() < 2:
        env["DCLIP_DIST_BACKEND"] = "gloo"
    p = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--batch", "16"], capture_output=True, text=True, timeout=900, env=env, cwd=REPO)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["global_batch"] == 32 and d["config"]["parallelism"] == "dp2"
    c = d["comm"]
    assert c["ranks"] == 2 and c["grad_buckets_per_step"] >= 1 and c["grad_allreduce_bytes_per_step"] > 3e8
    assert "fwd_bwd_ms_per_step" in d and "cpu_baseline" not in d


def test_bench_extra_legs_and_same_regime_cpu_baseline():
    env = dict(os.environ, DCLIP_BENCH_CPU_SECONDS="1")
    p = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--steps", "3", "--warmup", "1", "--batch", "16"],
                       capture_output=True, text=True, timeout=600, env=env, cwd=REPO)
    assert p.returncode == 0, p.stderr[-2000:]
    d = json.loads([l for l in p.stdout.splitlines() if l.strip()][0])
    assert 0 < d["fwd_bwd_ms_per_step"] <= d["ms_per_step"] * 1.05        # optimizer excluded
    # default execution: every step launched eagerly (the frozen text tower on its second stream), sampled steps carry the
    # per-launch GEMM events; the HIP-graph replay figure is reported beside it
    assert d["config"]["execution"].startswith("eager launches"), d["config"]["execution"]
    assert d["graph_ms_per_step"] > 0 and d["roofline"]["event_sampled_steps"] >= 1, d.get("graph_error")
    # --hybrid-graph (the default of rounds 1-2): forward + backward replayed from a captured graph, sampled steps eager
    p2 = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--steps", "3", "--warmup", "1", "--batch", "16",
                         "--hybrid-graph", "--no-cpu-baseline"], capture_output=True, text=True, timeout=600, env=env, cwd=REPO)
    assert p2.returncode == 0, p2.stderr[-2000:]
    d2 = json.loads([l for l in p2.stdout.splitlines() if l.strip()][0])
    assert "HIP-graph replay" in d2["config"]["execution"] and d2["eager_ms_per_step"] > 0, d2["config"]["execution"]
    assert abs(d2["config"]["loss"] - d["config"]["loss"]) < 1e-6 * abs(d["config"]["loss"])     # same arithmetic either way
    assert "north_star" in d["cpu_baseline"]["sample"] and "c1" in d["cpu_baseline_c1"]["sample"]
    assert d["roofline"]["traffic_source"].startswith("profiles/")
