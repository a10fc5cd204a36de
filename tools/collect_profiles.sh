#!/bin/bash
# Run on the GPU box (through gpurun) from the repository root: for each bench workload the rocprofv3 kernel-trace
# statistics and the two PMC passes (FETCH_SIZE, WRITE_SIZE -- separate runs, they do not fit one pass).
# Output goes to gpurun_out/prof/<workload>/{kt,fetch,write}; tools/summarize_profiles.py turns it into profiles/.
#   tools/collect_profiles.sh [workload ...]       (default: every workload below; "<name>_swap0" = --swap 0)
export TMPDIR=/tmp
OUT=gpurun_out/prof
ALL="blend_flush blend_flush_swap0 blend_flush_nowb blend rotate blend_flush_rgb565a8 blend_rgb565a8 blend_flush_4k clipped_small clipped_large fill_translucent fill_opaque map_opaque fill_masked fill_masked_argb clipped_large_argb clipped_small_argb scene blend_flush_466 flush_180_466"
# the memory regions are found once (bench.py's default finds them in every process, with probe launches of the tile
# kernel that a kernel trace would average into the workload's own)
MAP=$(timeout -k 10 120 python3 -m vvc_amd.placement 2>/dev/null | tail -n 1) || { echo "region map failed"; exit 1; }
echo "region map $MAP"
for w in ${@:-$ALL}; do
  name=${w%_swap0}
  extra=""
  [ "$name" != "$w" ] && extra="--swap 0"
  rm -rf $OUT/$w
  mkdir -p $OUT/$w
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$w/kt -- python3 bench.py --workload $name $extra --region-map $MAP --no-cpu-baseline --no-eighth > $OUT/$w/kt.log 2>&1 || { echo "$w: kernel trace failed"; exit 1; }
  timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/$w/fetch -- python3 bench.py --workload $name $extra --region-map $MAP --steps 3 --warmup 1 --no-cpu-baseline --no-eighth > $OUT/$w/fetch.log 2>&1 || { echo "$w: FETCH_SIZE pass failed"; exit 1; }
  timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/$w/write -- python3 bench.py --workload $name $extra --region-map $MAP --steps 3 --warmup 1 --no-cpu-baseline --no-eighth > $OUT/$w/write.log 2>&1 || { echo "$w: WRITE_SIZE pass failed"; exit 1; }
  echo "$w done"
done
