"""Host-side memory layout for the MI355X's 288 GB: which third of the device memory a buffer lies in matters.

The device memory behaves as three regions of 96 GB, and a kernel that reads one buffer while it writes another runs
8 % faster when the two lie in different regions: the fused blend + quarter turn over 4096 frames of 480x320 at 75.6 %
of the HBM peak instead of 70.1 % with the draw buffers and the panel images in different regions, 76.7 % with the
sources in the third (measured with all three carved out of one 264-GB allocation at swept offsets:
tools/exp/r3_placement3.py).  Inside a region the distance between the buffers changes nothing.

A process cannot ask the driver for physical addresses, and where separate allocations land depends on what earlier
processes left behind.  RegionArena therefore takes nearly all of the free device memory in ONE allocation, finds the
region boundaries inside it by timing (calibrate(): the fused launch with its panel images moved through the arena in
8-GB steps; a third of a second), and hands out buffers by region:

    arena = RegionArena()
    arena.calibrate(lvgpu)
    dst = arena.empty((n, H, W), torch.int16, region=0)      # draw buffers
    src = arena.empty((n, H, W), torch.int32, region=1)      # sources and masks
    out = arena.empty((n, W, H), torch.int16, region=2)      # panel images

A device with less than `min_bytes` free raises RuntimeError from the constructor: the caller then lives with the
allocator's placement.
"""
import torch

GB = 1 << 30
ALIGN = 2 << 20


class RegionArena:
    SLOT = 8 * GB            # granularity of the region map
    PROBE_FRAMES = 2048      # 480x320 frames of the probe launch: 0.63 GB draw buffers, 1.26 GB sources, 0.63 GB images

    def __init__(self, device=None, leave_free=16 * GB, min_bytes=200 * GB):
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        free, _total = torch.cuda.mem_get_info(self.device)
        nbytes = (free - leave_free) // self.SLOT * self.SLOT
        if nbytes < min_bytes:
            raise RuntimeError("RegionArena: %.0f GB free on %s, needs %.0f" % (free / GB, self.device, (min_bytes + leave_free) / GB))
        self.buf = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
        pad = -self.buf.data_ptr() % ALIGN
        self.base = pad
        self.n_slots = (nbytes - pad) // self.SLOT
        third = (self.n_slots + 2) // 3
        self.labels = [min(i // third, 2) for i in range(self.n_slots)]   # until calibrate(): thirds of the arena
        self.how = "assumed (thirds of the arena)"
        self.times = None
        self._set_runs()

    # -- carving ---------------------------------------------------------------------------------------------------
    def view(self, offset, shape, dtype):
        n = 1
        for s in shape:
            n *= s
        nbytes = n * torch.empty(0, dtype=dtype).element_size()
        assert offset % ALIGN == 0 and offset + nbytes <= self.n_slots * self.SLOT
        return self.buf[self.base + offset: self.base + offset + nbytes].view(dtype).view(shape)

    def _set_runs(self):
        """Per region its runs of slots, longest first, each less one slot at either end where it has four or more (a slot
        may straddle a boundary); spans[r] is the longest, the one buffers of region r come from first."""
        runs = {0: [], 1: [], 2: []}
        i = 0
        while i < self.n_slots:
            j = i
            while j + 1 < self.n_slots and self.labels[j + 1] == self.labels[i]:
                j += 1
            a, b = (i + 1, j - 1) if j - i + 1 >= 4 else (i, j)
            runs[self.labels[i]].append([a * self.SLOT, (b + 1) * self.SLOT, a * self.SLOT])   # lo, hi, next free byte
            i = j + 1
        for r in runs:
            runs[r].sort(key=lambda run: run[0] - run[1])
        self.runs = runs
        self.spans = {r: runs[r][0][:2] for r in runs if runs[r]}
        self.spilled = 0

    def empty(self, shape, dtype, region):
        """An uninitialised tensor in the given region (0, 1, 2), 2-MB aligned.  When no run of that region has room left
        it comes from whichever run of the arena has most (counted in `spilled`); RuntimeError when none has."""
        n = 1
        for s in shape:
            n *= s
        nbytes = n * torch.empty(0, dtype=dtype).element_size()
        others = sorted((run for r in self.runs if r != region for run in self.runs[r]), key=lambda run: run[2] - run[1])
        for k, run in enumerate(self.runs[region] + others):
            if run[2] + nbytes <= run[1]:
                off = run[2]
                run[2] = (off + nbytes + ALIGN - 1) // ALIGN * ALIGN
                self.spilled += k >= len(self.runs[region])
                return self.view(off, shape, dtype)
        raise RuntimeError("RegionArena: no run of the arena has %.1f GB left" % (nbytes / GB))

    # -- the region map, by timing ------------------------------------------------------------------------------------
    def _probe(self, lvgpu, anchor, slots):
        """Time of three fused 32-bit-image blend + quarter-turn launches with draw buffers and sources in slot `anchor` and
        the panel images in each of `slots` (in its upper half, so that the anchor's own slot can be probed too)."""
        n, W, H = self.PROBE_FRAMES, 480, 320
        a0 = anchor * self.SLOT
        dst = self.view(a0, (n, H, W), torch.int16)
        src = self.view(a0 + 1 * GB, (n, H, W), torch.int32)
        ctx = lvgpu.DrawCtx(dst, color_16_swap=1)
        # RGBA8888 rather than ARGB8888: the same traffic from another instantiation of the kernel, so that a kernel trace
        # of a program that calibrates and then draws ARGB8888 images keeps the two apart
        dsc = lvgpu.BlendDsc((0, 0, W - 1, H - 1), src_buf=src, opa=255, src_format=lvgpu.SRC_RGBA8888)
        times = {}
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for s in slots:
            out = self.view(s * self.SLOT + 4 * GB, (n, W, H), torch.int16)
            p = lvgpu.PreparedBlend(ctx, dsc, 1, 0, out)
            p.run()
            e0.record()
            for _ in range(3):
                p.run()
            e1.record()
            e1.synchronize()
            times[s] = e0.elapsed_time(e1) / 3
        return times

    @staticmethod
    def _split(times):
        """Slots whose launch was slow (same region as the anchor) and fast; None when the times show no two classes."""
        lo, hi = min(times.values()), max(times.values())
        if hi < 1.04 * lo:
            return None
        cut = (lo * hi) ** 0.5
        return [s for s in times if times[s] >= cut], [s for s in times if times[s] < cut]

    def calibrate(self, lvgpu):
        """Find the regions inside the arena.  Two sweeps: with the anchor in slot 0, the slots where the launch is slow
        are slot 0's region; with the anchor in the first of the others, the slow ones among them are the second region
        and the rest the third.  Leaves the thirds-of-the-arena map in place when the first sweep shows no structure."""
        all_slots = list(range(self.n_slots))
        t1 = self._probe(lvgpu, 0, all_slots)
        self.times = [t1]
        sp = self._split(t1)
        if sp is None:
            self.how = "assumed (thirds of the arena): timing showed no regions, launch times within 4 %"
            return self
        same0, others = sp
        labels = [0] * self.n_slots
        t2 = self._probe(lvgpu, others[0], others)
        self.times.append(t2)
        sp2 = self._split(t2)
        if sp2 is None:
            for s in others:
                labels[s] = 1
        else:
            for s in sp2[0]:
                labels[s] = 1
            for s in sp2[1]:
                labels[s] = 2
        self.labels = labels
        self.how = "found by timing the fused launch (%.3f / %.3f ms across / inside a region, %d frames)" % (
            min(t1.values()), max(t1.values()), self.PROBE_FRAMES)
        self._set_runs()
        return self

    def map_string(self):
        return "".join(str(l) for l in self.labels)

    def set_map(self, s):
        """Take a region map found earlier on this device (map_string() of an arena of the same size) instead of timing
        again; False, and nothing changed, when it does not fit this arena."""
        if len(s) != self.n_slots or set(s) - set("012") or s[0] != "0":
            return False
        self.labels = [int(c) for c in s]
        self.how = "given (found by timing in an earlier process)"
        self._set_runs()
        return True

    def describe(self):
        spans = ", ".join("%d: %d-%d GB" % (r, self.spans[r][0] // GB, self.spans[r][1] // GB) for r in sorted(self.spans))
        return "one %d-GB arena, regions %s; map %s (8-GB slots), %s%s" % (
            self.n_slots * self.SLOT // GB, spans, self.map_string(), self.how,
            "; %d buffers did not fit their region" % self.spilled if self.spilled else "")


if __name__ == "__main__":     # python3 -m vvc_amd.placement: calibrate on device 0 and print the region map
    from vvc_amd import lvgpu
    lvgpu.init(0)
    print(RegionArena().calibrate(lvgpu).map_string())
