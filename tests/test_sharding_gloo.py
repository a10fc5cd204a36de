"""CPU, world_size 2 over gloo: the frame-sharded multi-process path gives the same aggregate result as one process,
and the timing reduction takes the maximum over ranks."""
import json
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run(world):
    env = dict(os.environ)
    env.update({"MASTER_ADDR": "127.0.0.1", "OMP_NUM_THREADS": "1"})
    worker = os.path.join(HERE, "_shard_worker.py")
    if world == 1:
        env.update({"RANK": "0", "WORLD_SIZE": "1", "LOCAL_RANK": "0"})
        out = subprocess.check_output([sys.executable, worker], env=env, timeout=300)
    else:
        out = subprocess.check_output(
            [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
             "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), worker], env=env, timeout=300,
            stderr=subprocess.DEVNULL)
    return json.loads(out.decode().strip().splitlines()[-1])


def test_frame_range_partitions():
    from vvc_amd.sharding import frame_range
    for total in (0, 1, 7, 4096, 65536):
        for world in (1, 2, 3, 4, 8):
            r = [frame_range(g, world, total) for g in range(world)]
            assert r[0][0] == 0 and r[-1][1] == total
            assert all(r[i][1] == r[i + 1][0] for i in range(world - 1))
            assert max(h - l for l, h in r) - min(h - l for l, h in r) <= 1
    with pytest.raises(ValueError):
        frame_range(2, 2, 10)


def test_two_ranks_equal_one_rank():
    one = _run(1)
    two = _run(2)
    assert two["world"] == 2 and one["world"] == 1
    assert one["checksum"] == two["checksum"]
    assert two["tmax"] == [2.0, 2.0] and one["tmax"] == [1.0, 2.0]
