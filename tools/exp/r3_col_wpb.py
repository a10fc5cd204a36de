"""Sweep of the column kernel's wavefronts per workgroup and rows per phase (GPU box): every off-grid frame height of
the shape table, the blend fused in / without the write-back / the rotation alone."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
SHAPES = [(480, 272, 4096), (480, 328, 4096), (360, 360, 4096), (464, 464, 2048), (1024, 600, 1024), (720, 720, 1024),
          (1280, 720, 512), (1920, 1080, 256), (2560, 1440, 128), (3840, 2160, 64)]
only = sys.argv[1:] or ["blend_flush", "blend_flush_nowb", "rotate"]
for mode in only:
    for (W, H, n) in SHAPES:
        ty = (H + 63) // 64
        wpbs = sorted({8, min(ty, 16), (ty + 1) // 2 if ty > 8 else ty, (ty + 2) // 3 if ty > 16 else ty, 12 if ty > 16 else 8})
        for wpb in wpbs:
            for nr in (8, 4, 2):
                if nr == 8 and wpb > 8:
                    continue
                env = dict(os.environ, LVGPU_QT_KERNEL="col", LVGPU_COL_NR=str(nr), LVGPU_COL_WPB=str(wpb))
                out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", mode, "--width", str(W), "--height", str(H),
                                      "--frames-per-gpu", str(n), "--no-cpu-baseline", "--no-eighth", "--steps", "30"], env=env,
                                     capture_output=True, text=True, timeout=200)
                if out.returncode != 0:
                    print(mode, W, H, wpb, nr, "FAILED", out.stderr[-300:], flush=True)
                    sys.exit(1)
                j = json.loads(out.stdout.strip().split("\n")[-1])
                print("%-16s %4dx%-4d tiles_y %2d  WPB=%2d NR=%d  %5.1f %%  %s" % (mode, W, H, ty, wpb, nr, j["roofline"]["frac"] * 100,
                                                                              j["roofline"]["kernel"]), flush=True)
