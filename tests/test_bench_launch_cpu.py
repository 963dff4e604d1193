"""`python bench.py --gpus N` must start N ranks by itself (the driver may call it without a launcher) and must never
print a line whose `n_gpus` differs from `--gpus`.  Runs here without a GPU: ROMTIME_BENCH_DRYRUN=1 exercises the
launch, the gloo rendezvous, the shard arithmetic and the shape of the line with NO computation (`value` is null)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def run(args, **env_over):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(env_over)
    return subprocess.run([sys.executable, BENCH] + args, env=env, capture_output=True, text=True, timeout=300)


def json_lines(text):
    return [json.loads(l) for l in text.splitlines() if l.startswith("{")]


def test_self_launch_two_ranks_prints_one_line_with_n_gpus_2():
    res = run(["--gpus", "2", "--steps", "3", "--warmup", "1"], ROMTIME_BENCH_DRYRUN="1")
    assert res.returncode == 0, res.stderr[-2000:]
    lines = json_lines(res.stdout)
    assert len(lines) == 1, res.stdout
    line = lines[0]
    assert line["n_gpus"] == 2 and line["steps"] == 3 and line["warmup"] == 1
    assert line["config"]["rows_per_gpu"] == 500_000
    assert line["config"]["workload"] == "pod_1000000x512_r40_normalize"
    assert line["config"]["parallelism"] == "row-sharded x2"
    assert line["scaling"] == "strong" and line["dtype"] == "f64"
    assert line["dry_run"] is True and line["value"] is None          # never mistaken for a measurement
    assert set(line["fallback_counters"]) == {"eig_timeouts", "eig_general_form", "eig_one_xcd", "sets_recomputed"}


def test_launcher_world_size_must_match_gpus():
    res = run(["--gpus", "2"], ROMTIME_BENCH_DRYRUN="1", WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    assert res.returncode != 0 and "refusing" in res.stderr
    assert not json_lines(res.stdout)


def test_single_rank_dry_run_and_uneven_shards():
    res = run(["--gpus", "1"], ROMTIME_BENCH_DRYRUN="1")
    assert res.returncode == 0, res.stderr[-2000:]
    (line,) = json_lines(res.stdout)
    assert line["n_gpus"] == 1 and line["config"]["rows_per_gpu"] == 1_000_000
    res = run(["--gpus", "4"], ROMTIME_BENCH_DRYRUN="1")
    assert res.returncode == 0, res.stderr[-2000:]
    (line,) = json_lines(res.stdout)
    assert line["n_gpus"] == 4 and line["config"]["rows_per_gpu"] == 250_000


def test_no_gpu_for_the_ranks_is_an_error_not_a_one_rank_line():
    import torch

    if torch.cuda.device_count() >= 2:
        import pytest

        pytest.skip("this host has the GPUs")
    res = run(["--gpus", "2"])
    assert res.returncode != 0
    assert not json_lines(res.stdout)
