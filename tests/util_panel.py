"""A small model of an AXS15231B-style panel's frame memory behind the commands the reference's driver sends
(src/esp_lcd_axs15231b.c:292-337), and the loop that feeds it the transactions of a flushed frame.  This is the tests'
reading of the MIPI-DCS commands involved (CASET / RASET set the column / row window, RAMWR starts writing at the
window's first row, RAMWRC goes on where the previous transfer stopped): parity unpinned, like the framing itself."""
import numpy as np

CASET, RASET, RAMWR, RAMWRC = 0x2A, 0x2B, 0x2C, 0x3C


class PanelModel:
    """Frame memory of a hor x ver panel, 2 bytes per pixel, behind the command set panel_axs15231b_draw_bitmap uses."""

    def __init__(self, hor, ver, qspi):
        self.hor, self.ver, self.qspi = hor, ver, qspi
        self.gram = np.zeros((ver, hor, 2), np.uint8)
        self.written = np.zeros((ver, hor), np.int32)
        self.cols, self.rows = (0, hor - 1), (0, ver - 1)
        self.x, self.y = 0, 0
        self.log = []

    def _decode(self, word, want_color):
        if not self.qspi:
            return word
        opcode, cmd = word >> 24, (word >> 8) & 0xFF  # tx_param / tx_color, src/esp_lcd_axs15231b.c:173-193
        assert word & 0x00FF00FF == 0 and opcode == (0x32 if want_color else 0x02), hex(word)
        return cmd

    def tx_param(self, word, param):
        cmd = self._decode(word, False)
        lo, hi = (param[0] << 8) | param[1], (param[2] << 8) | param[3]
        assert len(param) == 4 and lo <= hi
        if cmd == CASET:
            assert hi < self.hor
            self.cols, self.x = (lo, hi), lo
        elif cmd == RASET:
            assert not self.qspi, "the QSPI interface sends no RASET (:316)"
            assert hi < self.ver
            self.rows, self.y = (lo, hi), lo
        else:
            raise AssertionError("unexpected command 0x%02x" % cmd)
        self.log.append(cmd)

    def tx_color(self, word, data):
        cmd = self._decode(word, True)
        assert cmd in (RAMWR, RAMWRC) and len(data) % 2 == 0
        if cmd == RAMWR:
            self.x, self.y = self.cols[0], self.rows[0]
        px = np.frombuffer(data, np.uint8).reshape(-1, 2)
        for i in range(0, len(px), self.cols[1] - self.x + 1):  # row by row inside the column window
            n = min(self.cols[1] - self.x + 1, len(px) - i)
            assert self.y <= self.rows[1], "pixels beyond the row window"
            self.gram[self.y, self.x:self.x + n] = px[i:i + n]
            self.written[self.y, self.x:self.x + n] += 1
            self.x += n
            if self.x > self.cols[1]:
                self.x, self.y = self.cols[0], self.y + 1
        self.log.append(cmd)


def send(gpu, panel_image_bytes, area, hor_res, ver_res, rot, trans_size, model, gaps=(0, 0)):
    chunks = gpu.flush_chunks(area, hor_res, ver_res, rot, trans_size)
    for ch in chunks:
        for is_color, cmd, param, off_px, n_bytes in gpu.axs15231b_transactions(ch, gaps[0], gaps[1], model.qspi, 16):
            if is_color:
                model.tx_color(cmd, panel_image_bytes[off_px * 2: off_px * 2 + n_bytes])
            else:
                model.tx_param(cmd, param)
    return chunks
