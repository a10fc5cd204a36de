"""bench.py --gpus N: the parent starts N ranks as a child process before anything touches a GPU (CPU tests; the
GPU rehearsal of the child side is tests/test_gpu_bench_ranks.py)."""
import os
import subprocess
import sys
import types

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    import importlib
    sys.modules.pop("bench", None)
    return importlib.import_module("bench")


_PARENT_PROBE = r"""
import json, os, sys, types
sys.path.insert(0, %r)
import bench
seen = {}
def fake_run(cmd, env=None):
    import torch
    seen["cmd"] = cmd
    seen["hsa_ipc"] = env.get("HSA_ENABLE_IPC_MODE_LEGACY")
    seen["cuda_initialised"] = torch.cuda.is_initialized()
    seen["maps_liblvgpu"] = "liblvgpu" in open("/proc/self/maps").read()
    seen["vvc_amd_imported"] = any(m.startswith("vvc_amd") for m in sys.modules)
    return types.SimpleNamespace(returncode=7)
bench.subprocess.run = fake_run
try:
    bench.main(["--gpus", "2", "--steps", "3", "--workload", "blend"])
except SystemExit as e:
    seen["exit"] = e.code
print(json.dumps(seen))
"""


def test_parent_spawns_before_touching_the_gpu():
    import json
    env = {k: v for k, v in os.environ.items() if k != "WORLD_SIZE"}
    env["LVGPU_BENCH_ONE_DEVICE"] = "1"                   # no device-count check on this GPU-less machine
    out = subprocess.check_output([sys.executable, "-c", _PARENT_PROBE % ROOT], env=env, timeout=300)
    seen = json.loads(out.decode().strip().splitlines()[-1])
    assert seen["exit"] == 7                              # the child's exit code is relayed
    cmd = seen["cmd"]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert cmd[cmd.index("--nproc-per-node") + 1] == "2"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    i = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[i + 1:] == ["--gpus", "2", "--steps", "3", "--workload", "blend"]
    assert not seen["cuda_initialised"] and not seen["maps_liblvgpu"] and not seen["vvc_amd_imported"]
    assert seen["hsa_ipc"] == "0"


def test_too_few_devices_fails_loudly():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "LVGPU_BENCH_ONE_DEVICE")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "64"], env=env, capture_output=True,
                       timeout=300)
    assert p.returncode != 0
    assert b"--gpus 64 asked for" in p.stderr
    assert p.stdout.strip() == b""


def test_default_batch_sizes():
    bench = _bench()
    assert bench.frames_for(0, 1, 4096) == (4096, "weak")
    for world in (2, 4, 8):                               # 65 536 frames sharded per frame over 2/4/8 GPUs
        n, scaling = bench.frames_for(0, world, 4096)
        assert n * world == 65536 and scaling == "strong"
    assert bench.frames_for(8192, 8, 4096) == (8192, "weak")
    assert bench.frames_for(0, 8, 64) == (128, "strong")  # 4K canvases: 1024 frames in total


def test_in_a_torchrun_environment_nothing_is_spawned(monkeypatch):
    bench = _bench()
    monkeypatch.setenv("WORLD_SIZE", "2")

    def no_spawn(*a, **k):
        raise AssertionError("spawned from inside a rank")

    monkeypatch.setattr(bench, "spawn_ranks", no_spawn)
    monkeypatch.setattr(bench.subprocess, "run", no_spawn)
    # the rank path needs a GPU; it must get past the spawn decision and fail later, on this machine at device set-up
    with pytest.raises(BaseException) as e:
        bench.main(["--gpus", "2"])
    assert "spawned" not in str(e.value)
