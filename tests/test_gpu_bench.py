"""bench.py contract (driver-facing): one JSON line on stdout with the required keys, at N=1 and through the multi-rank
code path (two ranks sharing the single GPU over gloo via the LR_BENCH_SHARE_GPU test hook)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REQUIRED = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
            "vs_baseline", "dtype", "data", "config"}


def _json_lines(out):
    return [json.loads(l) for l in out.splitlines() if l.startswith("{")]


def test_bench_single_gpu_json_contract():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "1", "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = _json_lines(p.stdout)
    assert len(lines) == 1, p.stdout[-2000:]
    r = lines[0]
    assert REQUIRED <= set(r) and r["n_gpus"] == 1 and r["steps"] == 1 and r["unit"] == "images/s"
    assert r["config"]["workload"].startswith("configs[1]") and r["dtype"] == "f16" and r["vs_baseline"] is None
    rf = r["roofline"]
    assert rf["bound"] == "mfma" and 0 < rf["frac"] < 1 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-9
    assert r["value"] > 1.0 and abs(r["ms_per_step"] / 50 - r["per_unet_step_ms"]) < 1e-6
    assert {"unet_step_events", "vae_512x1024", "training_256x512_b16", "kernels", "batch_sensitivity"} <= set(r)
    bs = r["batch_sensitivity"]
    assert {"B1", "B2", "B8"} <= set(bs) and all(0 < bs[k]["frac_of_mfma_peak"] < 1 for k in ("B1", "B2", "B8"))


def test_bench_two_ranks_share_gpu():
    """The driver's own command line: plain `python bench.py --gpus 2` spawns its two ranks itself."""
    env = dict(os.environ, LR_BENCH_SHARE_GPU="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--no-roofline",
                        "--no-cpu-baseline"], capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = _json_lines(p.stdout)
    assert len(lines) == 1, p.stdout[-2000:]      # rank 0 only
    r = lines[0]
    assert r["n_gpus"] == 2 and r["config"]["global_batch"] == 8 and r["scaling"] == "weak" and r["config"]["ranks"] == 2
    # first-contact checks run before timing on every multi-rank launch
    st = r["selftest"]
    assert st["ok"] and {"world_size", "all_gather_rank_stamp", "all_gather_into_tensor_f16", "all_reduce_sum", "device_uniqueness"} <= set(st["stages"])
    assert st["stages"]["all_gather_rank_stamp"]["ranks_seen"] == [0, 1]


def test_bench_selftest_failure_is_reported_in_the_json_line():
    """A broken collective on first contact must be diagnosable from stdout alone: a JSON line naming the stage, exit code 1."""
    env = dict(os.environ, LR_BENCH_SHARE_GPU="1", LR_BENCH_SELFTEST_FAIL="all_gather_into_tensor_f16")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--no-roofline",
                        "--no-cpu-baseline"], capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert p.returncode != 0
    lines = _json_lines(p.stdout)
    assert lines and all(l["value"] is None for l in lines)
    assert all(l["error"]["stage"] == "selftest:all_gather_into_tensor_f16" for l in lines)
    assert lines[0]["error"]["selftest"]["stages"]["all_gather_rank_stamp"]["ok"]


def test_bench_world_size_must_match_gpus():
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1"], capture_output=True, text=True,
                       timeout=300, cwd=ROOT, env=env)
    assert p.returncode != 0 and "WORLD_SIZE" in (p.stderr + p.stdout)


def test_bench_mv_shard_four_ranks_share_gpu():
    """configs[3] in its sharded form (one canvas per rank, per-block exchange) through the driver-style command; the four ranks
    share the GPU over gloo here, so the step runs eagerly (the hipGraph with its RCCL collectives needs `nccl`)."""
    env = dict(os.environ, LR_BENCH_SHARE_GPU="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--workload", "mv5", "--mv-shard", "--selftest", "--steps", "1",
                        "--warmup", "0", "--ddim-steps", "2", "--no-roofline", "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=1500, cwd=ROOT, env=env)
    assert p.returncode == 0, p.stderr[-2000:]
    r = _json_lines(p.stdout)[0]
    assert r["n_gpus"] == 4 and r["unit"] == "samples/s" and r["scaling"] == "strong" and r["config"]["global_batch"] == 1
    assert "mv-shard x4" in r["config"]["parallelism"] and r["value"] > 0
    mv = r["selftest"]["stages"]["mv_graph_vs_eager"]      # (eager on both sides under gloo: exercises the comparison)
    assert r["selftest"]["ok"] and mv["ok"] and mv["ranks_ok"] == [True] * 4 and mv["max_abs_graph_vs_eager"] == 0.0


def test_bench_split_cfg_two_ranks_share_gpu():
    env = dict(os.environ, LR_BENCH_SHARE_GPU="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--split-cfg", "--steps", "1", "--warmup", "0",
                        "--ddim-steps", "5", "--no-roofline", "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert p.returncode == 0, p.stderr[-2000:]
    r = _json_lines(p.stdout)[0]
    assert r["n_gpus"] == 2 and r["config"]["global_batch"] == 4 and "split-cfg x2" in r["config"]["parallelism"] and r["value"] > 0


def test_train_workload_two_ranks_share_gpu():
    """Data-parallel training step (token-gradient all-reduce) through the multi-rank path, two ranks on the one GPU."""
    env = dict(os.environ, LR_BENCH_SHARE_GPU="1")
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                        "127.0.0.1", "--master-port", "29519", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload",
                        "train", "--steps", "2", "--warmup", "1"],
                       capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = _json_lines(p.stdout)
    assert len(lines) == 1, p.stdout[-2000:]
    r = lines[0]
    assert r["n_gpus"] == 2 and r["config"]["global_batch"] == 32 and r["unit"] == "samples/s"
    assert 0.5 < r["final_loss"] < 3.0 and r["value"] > 50
