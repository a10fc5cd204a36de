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
