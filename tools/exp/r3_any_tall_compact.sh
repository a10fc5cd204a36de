for wh in "1023 600 1024" "1278 718 512"; do set -- $wh
for e in WPB=auto LVGPU_ANY_WPB=4 LVGPU_ANY_NR=8; do
echo -n "compact ${1}x${2} $e: "; env $e timeout -k 10 120 python3 bench.py --no-cpu-baseline --no-eighth --steps 30 --warmup 3 --placement compact --workload blend_flush --width $1 --height $2 --frames-per-gpu $3 2>/dev/null | tail -n 1 | python3 tools/bench_summary.py | awk '{print $(NF-3), $(NF-2)}' || exit 1
done; done
