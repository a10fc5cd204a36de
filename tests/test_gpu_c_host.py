"""GPU: a plain C program on the C ABI (examples/batch_blend_flush.c) against the CPU oracle."""
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "examples", "batch_blend_flush")


@pytest.mark.parametrize("rot", [1, 3, 0, 2])
def test_c_host_program(gpu, oracle, rot):
    if not os.path.exists(EXE):
        pytest.fail("examples/batch_blend_flush is not built (make -C examples; __graft_entry__.build() does it)")
    n, W, H = 24, 480, 320
    out = subprocess.check_output([EXE, str(n), str(rot)], timeout=120).decode()
    vals = dict(line.split(" ", 1) for line in out.strip().splitlines())
    px = W * H
    dst = oracle.synth(oracle.SYNTH_U16, 0x5EED0001, 0, n * px)
    src = oracle.synth(oracle.SYNTH_ARGB_UI, 0x5EED0002, 0, n * px)
    panel = np.empty(n * px, np.uint16)
    oracle.argb_blend_flush_batch(dst, src, panel, W, H, n, rot, 1, 4)
    assert int(vals["draw_buffers"]) == oracle.checksum(dst, 0)
    assert int(vals["panel_images"]) == oracle.checksum(panel, 0)
    assert int(vals["bad_clip_rc"]) == -1
    assert "fused" in vals["frames"]
    # the display-list phase of the example: background + 40 one-row masked fills + a translucent panel, one launch
    mask = oracle.synth(oracle.SYNTH_MASK_UI, 0x5EED0003, 0, 200 * 40).reshape(40, 200)
    frames = np.empty((n, H, W), np.uint16)
    area = (0, 0, W - 1, H - 1)
    one = np.empty((H, W), np.uint16)
    oracle.blend(one, area, area, area, color=0x1084, opa=255, swap=1)
    for r in range(40):
        row = (37, 50 + r, 236, 50 + r)
        oracle.blend(one, area, area, row, color=0xE0FF, mask=mask[r:r + 1], opa=230, swap=1)
    oracle.blend(one, area, area, (100, 20, 400, 300), color=0x00F8, opa=96, swap=1)
    frames[:] = one  # the opaque background makes every frame the same
    assert int(vals["list_draw_buffers"]) == oracle.checksum(frames.reshape(-1), 0)
    pan = np.concatenate([oracle.flush_frame(one, W, H, rot, 0)] * n)
    assert int(vals["list_panel_images"]) == oracle.checksum(pan, 0)
    assert vals["list_kernel"] == "k_blend_list(fused)" and int(vals["list_launches"]) == 1
    assert int(vals["list_steps"]) < 42 * 40  # far fewer steps than items x tiles: the rows are batched


MULTI = os.path.join(ROOT, "examples", "multi_gpu_blend_flush")


@pytest.mark.parametrize("rot", [1, 0])
def test_multi_device_c_host_sharding(gpu, oracle, rot):
    """examples/multi_gpu_blend_flush.c: one host thread + stream per shard, lvgpu_init per thread, contiguous frame
    ranges, per-shard checksums added on the host.  Rehearsed with two and three threads on device 0: the totals equal
    the one-shard run's and the oracle's, whatever the partition (25 frames do not divide by 2 or 3)."""
    if not os.path.exists(MULTI):
        pytest.fail("examples/multi_gpu_blend_flush is not built (make -C examples; __graft_entry__.build() does it)")
    n, W, H = 25, 480, 320
    px = W * H
    dst = oracle.synth(oracle.SYNTH_U16, 0x5EED0001, 0, n * px)
    src = oracle.synth(oracle.SYNTH_ARGB_UI, 0x5EED0002, 0, n * px)
    panel = np.empty(n * px, np.uint16)
    oracle.argb_blend_flush_batch(dst, src, panel, W, H, n, rot, 1, 4)
    want = (oracle.checksum(dst, 0), oracle.checksum(panel, 0))
    for shards in (1, 2, 3):
        out = subprocess.check_output([MULTI, str(n), str(shards), "1", str(rot)], timeout=120).decode()
        lines = out.strip().splitlines()
        vals = dict(line.split(" ", 1) for line in lines if not line.startswith("shard "))
        per = [line for line in lines if line.startswith("shard ")]
        assert len(per) == shards and all("fused" in line for line in per), out
        ranges = [tuple(int(v) for v in line.split()[5].split("..")) for line in per]
        assert ranges[0][0] == 0 and ranges[-1][1] == n and all(a[1] == b[0] for a, b in zip(ranges, ranges[1:]))
        assert (int(vals["draw_buffers"]), int(vals["panel_images"])) == want, (shards, out)
    # more devices than the machine has is an argument error, not a crash
    p = subprocess.run([MULTI, "8", "2", "99"], capture_output=True, timeout=60)
    assert p.returncode == 1 and b"bad arguments" in p.stderr


ARENA = os.path.join(ROOT, "examples", "region_arena")


def test_c_host_places_its_buffers_by_region(gpu, oracle):
    """examples/region_arena.c: one arena, the memory regions inside it found by timing, the batch run with one buffer
    per region and with all three in one: the same checksums either way, and the oracle's.  The rates it prints are
    not asserted (256 frames are too few to show them; `examples/region_arena` without arguments runs 4096)."""
    import torch
    if not os.path.exists(ARENA):
        pytest.fail("examples/region_arena is not built (make -C examples; __graft_entry__.build() does it)")
    torch.cuda.empty_cache()
    if torch.cuda.mem_get_info()[0] < (100 << 30):
        pytest.skip("needs 100 GB of free device memory")
    n, W, H = 256, 480, 320
    p = subprocess.run([ARENA, str(n)], capture_output=True, timeout=170)
    assert p.returncode == 0, p.stderr.decode()[-1000:]
    vals = dict(line.split(" ", 1) for line in p.stdout.decode().strip().splitlines())
    assert len(vals["map"]) == int(vals["arena_gb"]) // 8 and set(vals["map"]) <= set("012") and vals["map"][0] == "0"
    px = W * H
    dst = oracle.synth(oracle.SYNTH_U16, 0x5EED0001, 0, n * px)
    src = oracle.synth(oracle.SYNTH_ARGB_UI, 0x5EED0002, 0, n * px)
    panel = np.empty(n * px, np.uint16)
    oracle.argb_blend_flush_batch(dst, src, panel, W, H, n, 1, 1, 4)
    for placement in ("by_region", "one_region"):
        assert int(vals["draw_buffers_" + placement]) == oracle.checksum(dst, 0)
        assert int(vals["panel_images_" + placement]) == oracle.checksum(panel, 0)
        assert float(vals[placement + "_ms"]) > 0
    assert vals["kernel"] == "k_tile(fused)"
