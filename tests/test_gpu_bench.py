"""bench.py's contract on a small instance of its workload (run with -m gpu): one JSON line with the fields the
driver reads, `value` = the job's average rate whatever --steps / --warmup are."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(*extra):
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--points", "2048", "--no-cpu-baseline",
                        "--min-timed", "0.05", *extra], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_line_and_job_average():
    a = run_bench("--steps", "20", "--warmup", "5")
    b = run_bench("--steps", "150", "--warmup", "0")
    for d, k, w in ((a, 20, 5), (b, 150, 0)):
        assert d["metric"].startswith("relaxation iterations/sec") and d["unit"] == "iterations/s"
        assert d["n_gpus"] == 1 and d["steps"] == k and d["warmup"] == w and d["higher_is_better"] is True
        assert d["dtype"] == "f32" and d["data"] == "synthetic" and d["vs_baseline"] is None
        assert d["value"] > 0 and abs(d["ms_per_step"] * d["value"] - 1e3) < 1e-6 * 1e3
        t = d["timing"]
        assert t["slices_per_rotation"] == -(-t["job_iterations"] // k) and t["rotations"] >= 3
        assert len(t["iterations_per_s_by_slice"]) == t["slices_per_rotation"]
        r = d["roofline"]
        assert r["bound"] == "hbm" and r["peak"] == 8000.0 and 0 < r["frac"] < 1 and r["launches"] > 0
        assert abs(r["achieved"] - r["algorithmic_bytes_per_launch"] / (r["avg_launch_us"] * 1e-6) / 1e9) < 1e-6 * r["achieved"]
        w_run = d["whole_run"]
        assert w_run["converged"] and w_run["iterations_run"] == t["job_iterations"]
    # the same job: the two slicings agree with each other and with the job timed in one piece
    assert a["timing"]["job_iterations"] == b["timing"]["job_iterations"]
    # (at this size an iteration is ~15 us, so the two synchronisations around a 20-iteration slice weigh 10-20 %; at
    # config 3's size the same comparison is within 2 %: DESIGN.md section 6)
    assert abs(a["value"] / b["value"] - 1.0) < 0.35, (a["value"], b["value"])
    assert abs(a["value"] / a["whole_run"]["iterations_per_s"] - 1.0) < 0.35


def run_bench_env(env_extra, *args):
    env = dict(os.environ, **env_extra)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], capture_output=True, text=True,
                       timeout=900, cwd=ROOT, env=env)
    assert p.returncode == 0, (p.stdout[-1500:], p.stderr[-3000:])
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_gpus_n_launched_plainly_runs_the_row_sharded_path():
    """`python bench.py --gpus 2` with no launcher around it: the script starts the two ranks itself and the line
    measures ONE config-4-style embedding row-sharded over them (strong scaling), the independent replicas following
    as a secondary field.  Rehearsal on the one GPU of this box: both ranks on device 0, gloo instead of RCCL (RCCL
    refuses two ranks on one device), 4 096 points."""
    d = run_bench_env({"TOPOLOW_BENCH_OVERSUBSCRIBE": "1", "TOPOLOW_DIST_BACKEND": "gloo",
                       "TOPOLOW_SYMMETRIC_MIN_N": "1024"},       # config 4 is above the size gate, 4 096 points are not
                      "--gpus", "2", "--points", "4096", "--steps", "20", "--warmup", "5", "--min-timed", "0.05",
                      "--no-cpu-baseline")
    # one-stage iterations ran as the symmetric sweep sharded over the ranks (tile segments, one all-reduce of the moves)
    assert d["config"]["symmetric_segments"] is True and "symm_sweep_kernel" in d["roofline"]["kernel"]
    assert d["scaling"] == "strong" and d["ranks"] == 2 and d["n_gpus"] == 1      # devices really used: one
    assert d["collective_backend"] == "gloo" and d["rccl_ranks"] == 0
    assert "config 4" in d["config"]["workload"] and d["config"]["parallelism"] == "rows/2"
    assert d["steps"] == 20 and d["warmup"] == 5 and d["value"] > 0
    assert abs(d["ms_per_step"] * d["value"] - 1e3) < 1e-6 * 1e3
    t = d["timing"]
    assert t["slices_per_rotation"] == -(-t["job_iterations"] // 20) and t["rotations"] >= 3
    assert d["whole_run"]["iterations_run"] == t["job_iterations"] and d["whole_run"]["converged"]
    r = d["roofline"]
    assert len(r["per_gpu_frac"]) == 2 and all(0 < f < 1 for f in r["per_gpu_frac"])
    rows = [x["rows"] for x in d["breakdown_ms_per_iteration"]["per_rank"]]
    assert rows == [[0, 2048], [2048, 4096]]
    rep = d["replicas"]
    assert rep["scaling"] == "weak" and rep["value"] > 0 and "replicated" in rep["config"]["workload"]
    # the base of the strong-scaling curve: the same code at world size 1
    one = run_bench_env({}, "--gpus", "1", "--mode", "sharded", "--points", "4096", "--steps", "20", "--warmup", "5",
                        "--min-timed", "0.05", "--no-cpu-baseline")
    assert one["scaling"] == "strong" and one["ranks"] == 1 and one["n_gpus"] == 1 and "replicas" not in one
    assert one["timing"]["job_iterations"] > 0 and one["config"]["parallelism"] == "rows/1"
    # same embedding, same schedule: the two runs stop within a few checks of each other at the same error
    assert abs(one["timing"]["job_iterations"] - t["job_iterations"]) <= 30
    assert abs(one["final_mae"] / d["final_mae"] - 1) < 0.02


def test_bench_refuses_more_gpus_than_the_box_has():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "64"], capture_output=True, text=True,
                       timeout=300, cwd=ROOT)
    assert p.returncode != 0 and "GPU(s) visible" in p.stderr


def test_bench_reports_the_f64_job_and_the_host():
    d = run_bench("--steps", "50", "--warmup", "5")
    f = d["precision_f64"]
    assert f["converged"] and f["iterations_per_s"] > 0 and 0 < f["frac"] < 1
    assert abs(f["final_mae"] / d["whole_run"]["final_mae"] - 1) < 0.03       # same job, same schedule, wider arithmetic
    assert "f64" in d["dtype_note"]
