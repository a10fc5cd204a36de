#!/bin/bash
# Samples the GPU's clocks and power (rocm-smi reads them from sysfs) once a second while the fused quarter-turn launch
# runs back to back for about twenty seconds in another process: do the memory / fabric clocks move under sustained load,
# and does the kernel's rate move with them?  Output: gpurun_out/clocks/{rate.txt,smi.txt}
O=gpurun_out/clocks
mkdir -p $O
rocm-smi --showclocks --showpower --showtemp > $O/smi_idle.txt 2>&1
( for i in $(seq 1 24); do echo "== t=$i"; rocm-smi --showclocks --showpower 2>/dev/null | grep -E "clk|Power|power" ; sleep 1; done ) > $O/smi.txt 2>&1 &
SMI=$!
timeout -k 10 120 python -u - > $O/rate.txt 2>&1 <<'PY'
import sys, time, torch
sys.path.insert(0, ".")
from vvc_amd import lvgpu
lvgpu.init(0)
W, H, n = 480, 320, 4096
dst = lvgpu.synth(lvgpu.SYNTH_U16, 1, 0, (n, H, W)); src = lvgpu.synth(lvgpu.SYNTH_ARGB_UI, 2, 0, (n, H, W))
out = torch.empty((n, W, H), dtype=torch.int16, device="cuda")
pb = lvgpu.PreparedBlend(lvgpu.DrawCtx(dst), lvgpu.BlendDsc((0, 0, W - 1, H - 1), src_buf=src, src_format=lvgpu.SRC_ARGB8888), 1, 0, out)
t0 = time.time()
while time.time() - t0 < 20:
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(400): pb.run()
    e1.record(); torch.cuda.synchronize()
    print("t=%5.1f  %5.1f %% of 8 TB/s" % (time.time() - t0, 100 * 10 * W * H * n / (e0.elapsed_time(e1) / 400 * 1e-3) / 8e12), flush=True)
PY
rc=$?
wait $SMI
cat $O/rate.txt; grep -c "==" $O/smi.txt; sed -n 1,30p $O/smi.txt; tail -12 $O/smi.txt
exit $rc
