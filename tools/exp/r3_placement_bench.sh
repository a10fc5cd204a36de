#!/bin/bash
# bench.py workloads with the buffers spread over the memory regions against the allocator's placement (GPU box).
set -o pipefail
for w in blend_flush blend_flush_nowb blend rotate blend_flush_rgb565a8 blend_rgb565a8 blend_flush_4k blend_flush_466 flush_180_466 clipped_large clipped_small fill_masked map_opaque fill_masked_argb clipped_large_argb scene fill_translucent; do
for p in spread compact; do
echo -n "$w $p "; timeout -k 10 200 python3 bench.py --workload $w --placement $p --no-cpu-baseline --no-eighth --steps 40 2>/dev/null | python3 tools/bench_summary.py | awk '{print $(NF-4), $(NF-3), $(NF-2)}' || exit 1
done; done
