"""GPU: the wire format end to end.  A fused blend + flush launch leaves the panel image in device memory; the image
is cut into the reference's transport chunks (lvgpu_flush_chunks, src/lv_port.c:466-566), every chunk is framed as the
reference's AXS15231B driver frames it (lvgpu_axs15231b_transactions, src/esp_lcd_axs15231b.c:292-337), and a small
model of the panel's frame memory, written here, consumes the transactions.  What the model's frame memory holds at the
end must be, byte for byte, the oracle's panel image.

Parity unpinned for the framing: the driver needs ESP-IDF headers that the reference tree does not hold, so the
expected transactions are this repository's reading of the driver source, and the panel model below is this test's
reading of the MIPI-DCS commands the driver uses (CASET / RASET set the column / row window, RAMWR starts writing at the
window's first row, RAMWRC goes on where the previous transfer stopped)."""
import numpy as np
import pytest

from util import to_dev, to_np
from util_panel import RAMWR, RAMWRC, PanelModel, send as _send

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("qspi", [True, False])
@pytest.mark.parametrize("rot,W,H", [(1, 480, 320), (0, 320, 480), (3, 480, 320), (2, 320, 480)])
def test_fused_launch_to_panel_memory(gpu, oracle, rot, W, H, qspi):
    """The firmware's geometry (panel 320x480, trans_size = 320*480/10): ROT_90 of the 480x320 logical frame is the
    north-star configuration, ROT_NONE of a 320x480 frame is what the shipped firmware runs.  Both byte orders must put
    the same bytes on the wire: LV_COLOR_16_SWAP 1 buffers as they are, LV_COLOR_16_SWAP 0 buffers through swap_bytes."""
    N = 2
    rng = np.random.default_rng(rot + 10 * qspi)
    area = (0, 0, W - 1, H - 1)
    hor, ver = (H, W) if rot in (1, 3) else (W, H)          # the panel as the driver sees it
    assert (hor, ver) == (320, 480)
    native = rng.integers(0, 65536, (N, H, W), dtype=np.uint16)
    src = rng.integers(0, 1 << 32, (N, H, W), dtype=np.uint32)
    grams = {}
    for swap in (1, 0):
        dst = native.byteswap() if swap else native.copy()
        exp = dst.copy()
        for f in range(N):
            oracle.img_blend(exp[f], area, area, area, src[f], 2, 255, swap, W)
        d = to_dev(dst)
        out = gpu.blend_flush(gpu.DrawCtx(d, color_16_swap=swap), gpu.BlendDsc(area, src_buf=to_dev(src), src_format=2),
                              rot, swap_bytes=1 - swap)
        assert "fused" in gpu.last_kernel()
        images = to_np(out, np.uint16).reshape(N, -1)
        for f in range(N):
            model = PanelModel(hor, ver, qspi)
            chunks = _send(gpu, images[f].tobytes(), area, W, H, rot, 320 * 480 // 10, model)
            assert len(chunks) == 10
            per = 2 if qspi else 3
            assert model.log[per - 1] == RAMWR and all(c == RAMWRC for c in model.log[2 * per - 1::per])   # :330-334
            assert (model.written == 1).all(), "every panel pixel is written exactly once"
            want = oracle.flush_frame(exp[f], W, H, rot, 1 - swap)
            assert model.gram.tobytes() == want.tobytes(), (swap, f)
            grams[(swap, f)] = model.gram.tobytes()
    for f in range(N):
        assert grams[(1, f)] == grams[(0, f)], "both byte orders put the same bytes on the wire"
        # and those bytes are big-endian RGB565 (src/display.h:26-30) of the blended native pixels
        e1 = native[f].byteswap().copy()
        oracle.img_blend(e1, area, area, area, src[f], 2, 255, 1, W)
        be = np.frombuffer(grams[(1, f)], ">u2").reshape(ver, hor)
        nat = oracle.flush_frame(e1.byteswap().copy(), W, H, rot, 0).reshape(ver, hor)
        assert np.array_equal(be, nat)


@pytest.mark.parametrize("rot", [0, 1, 2, 3])
def test_partial_panels_and_gaps(gpu, oracle, rot):
    """A panel that is not the firmware's: 200x96 logical pixels, transport buffers of 1000 pixels (chunks that do not
    divide the frame evenly, src/lv_port.c:466-478), a driver configured with gaps (src/esp_lcd_axs15231b.c:301-304)."""
    W, H, N = 200, 96, 1
    rng = np.random.default_rng(40 + rot)
    frame = rng.integers(0, 65536, (N, H, W), dtype=np.uint16)
    out = gpu.flush_rotate(to_dev(frame), rot)
    image = to_np(out, np.uint16).reshape(-1)
    hor, ver = (H, W) if rot in (1, 3) else (W, H)
    xg, yg = 4, 0   # a row gap would make no chunk start at row 0, and the driver would never send RAMWR (:330)
    for qspi in (True, False):
        model = PanelModel(hor + xg, ver + yg, qspi)
        chunks = _send(gpu, image.tobytes(), (0, 0, W - 1, H - 1), W, H, rot, 1000, model, gaps=(xg, yg))
        assert len(chunks) > 1 and sum(c[5] for c in chunks) == W * H
        want = oracle.flush_frame(frame[0], W, H, rot, 0).reshape(ver, hor)
        got = np.frombuffer(model.gram.tobytes(), np.uint16).reshape(ver + yg, hor + xg)
        assert np.array_equal(got[yg:, xg:], want)
        assert (model.written[yg:, xg:] == 1).all() and model.written[:, :xg].sum() == 0
