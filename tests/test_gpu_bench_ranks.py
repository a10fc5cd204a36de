"""GPU: `bench.py --gpus 2` starts two ranks by itself.  Rehearsed on one GPU: both ranks share device 0 and talk over
gloo (LVGPU_BENCH_BACKEND=gloo LVGPU_BENCH_ONE_DEVICE=1), so the numbers mean nothing; the control flow -- spawn,
frame sharding, barrier, max over ranks, one JSON line from rank 0, exit code -- is what is exercised."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(*argv, **extra_env):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update({"LVGPU_BENCH_BACKEND": "gloo", "LVGPU_BENCH_ONE_DEVICE": "1"})
    env.update(extra_env)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(argv), env=env, capture_output=True,
                          timeout=420)


def test_gpus_2_runs_two_ranks(gpu):
    p = _bench("--gpus", "2", "--frames-per-gpu", "64", "--steps", "3", "--warmup", "1")
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    lines = [l for l in p.stdout.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1                                 # rank 0 alone prints
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["scaling"] == "weak"
    assert r["config"]["frames_per_gpu"] == 64 and r["config"]["total_frames"] == 128
    assert len(r["per_rank_Mpx_s"]) == 2 and all(v > 0 for v in r["per_rank_Mpx_s"])
    assert r["roofline"]["kernel"] == "k_tile(fused)"
    assert "cpu_baseline" not in r and "eighth_gpu" not in r   # rank 0 at N = 1 only
    assert r["value"] > 0 and abs(r["value"] - 2 * 64 * 480 * 320 * 3 / (r["ms_per_step"] * 3e-3) / 1e6) < 0.01 * r["value"]
    # the value comes from device time; host time across the closing barrier is reported beside it and is no shorter
    assert r["host_ms_per_step"] >= r["ms_per_step"] * 0.999 and r["rank_sync_backend"] == "gloo"
    assert "efficiency_vs_1gpu" not in r                   # no one-GPU figure known


def test_efficiency_against_a_given_one_gpu_rate(gpu):
    p = _bench("--gpus", "2", "--frames-per-gpu", "64", "--steps", "3", "--warmup", "1", "--ref-1gpu-mpx", "100000")
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    r = json.loads([l for l in p.stdout.decode().splitlines() if l.startswith("{")][0])
    assert r["ref_1gpu_Mpx_s"] == 100000.0 and r["ref_1gpu_source"] == "--ref-1gpu-mpx"
    assert abs(r["efficiency_vs_1gpu"] - r["value"] / 2e5) < 1e-3


def test_a_failing_rank_fails_the_command(gpu):
    p = _bench("--gpus", "2", "--frames-per-gpu", "64", "--steps", "1", "--warmup", "1", "--workload", "scene", "--width", "64")
    assert p.returncode != 0                               # the scene workload rejects other sizes: every rank asserts
    assert not [l for l in p.stdout.decode().splitlines() if l.startswith("{")]


def test_placement_switch(gpu):
    """One rank, both placements: the line says where the buffers were put; the results do not depend on it (same checksum
    check inside bench.py either way)."""
    for placement, words in (("spread", "-GB arena, regions"), ("compact", "wherever the allocator puts them")):
        p = _bench("--frames-per-gpu", "64", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-eighth",
                   "--placement", placement)
        assert p.returncode == 0, p.stderr.decode()[-2000:]
        r = json.loads([l for l in p.stdout.decode().splitlines() if l.startswith("{")][0])
        assert words in r["config"]["placement"], r["config"]["placement"]
        assert r["roofline"]["kernel"] == "k_tile(fused)" and r["value"] > 0


def test_one_rank_over_rccl(gpu):
    """The collective calls of a multi-GPU run (barrier, max over ranks, gather of the per-rank times) over RCCL itself,
    with a world of one rank on this one GPU."""
    sys.path.insert(0, ROOT)
    from bench import free_port
    env = {k: v for k, v in os.environ.items() if k not in ("LVGPU_BENCH_BACKEND", "LVGPU_BENCH_ONE_DEVICE")}
    env.update({"RANK": "0", "WORLD_SIZE": "1", "LOCAL_RANK": "0", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(free_port()),
                "LVGPU_BENCH_FORCE_DIST": "1", "LVGPU_BENCH_BACKEND": "nccl"})
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--frames-per-gpu", "256", "--steps", "3", "--warmup", "1",
                        "--no-cpu-baseline", "--no-eighth"], env=env, capture_output=True, timeout=420)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    r = json.loads([l for l in p.stdout.decode().splitlines() if l.startswith("{")][-1])
    assert r["n_gpus"] == 1 and r["value"] > 0 and r["roofline"]["kernel"] == "k_tile(fused)"
