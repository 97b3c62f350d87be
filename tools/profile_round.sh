#!/bin/bash
# One command for the round's profile evidence (run on the GPU box via gpurun):
#   1. the rocprofv3 passes of tools/prof.sh over the default bench.py run, condensed into gpurun_out/<tag>_rocprof_summary.txt,
#      and the PMC-measured HBM traffic of the headline kernel into gpurun_out/traffic.json (bench.py's roofline.traffic fallback);
#   2. (round 4) the kernel families beside the scan, each through tools/prof_cmd.sh = kernel-trace + FETCH_SIZE + WRITE_SIZE + one
#      SQ pass, one table per command: time windows / sort / literal reducers (f1), bloom build (f3), dictionary / run-end
#      scans (f2), out-of-order windows behind a selection without and with the sort.
# usage: tools/profile_round.sh r04   -> copy gpurun_out/<tag>_*.txt and traffic.json into profiles/ (gpurun only merges gpurun_out/ back)
set -u
TAG=${1:-r04}
bash tools/prof.sh $TAG > gpurun_out/prof_$TAG.files 2>&1
python3 tools/prof_summary.py gpurun_out/prof_$TAG gpurun_out/traffic.json > gpurun_out/${TAG}_rocprof_summary.txt
tail -n 3 gpurun_out/${TAG}_rocprof_summary.txt
if [ "${KNOX_PROFILE_FAMILIES:-1}" = "1" ]; then
  bash tools/prof_cmd.sh $TAG windows tools/bench_windows.py --profile > gpurun_out/prof_${TAG}_windows.log 2>&1
  bash tools/prof_cmd.sh $TAG bloom_build tools/bench_bloom_build.py > gpurun_out/prof_${TAG}_bloom.log 2>&1
  bash tools/prof_cmd.sh $TAG encoded tools/bench_encoded.py 28 > gpurun_out/prof_${TAG}_encoded.log 2>&1
  export KNOX_SHUFFLED_ONLY=masked
  bash tools/prof_cmd.sh $TAG shuffled tools/bench_shuffled.py > gpurun_out/prof_${TAG}_shuffled.log 2>&1
  export KNOX_SHUFFLED_DTYPE=f64  # Sum over float64: the exact fixed-point adders
  bash tools/prof_cmd.sh $TAG shuffled_f64 tools/bench_shuffled.py > gpurun_out/prof_${TAG}_shuffled_f64.log 2>&1
  unset KNOX_SHUFFLED_DTYPE
  export KNOX_WINDOW_DIRECT=0     # the same shape through the sort (what float sums, var / std and larger window counts still take)
  bash tools/prof_cmd.sh $TAG shuffled_sorted tools/bench_shuffled.py > gpurun_out/prof_${TAG}_shuffled_sorted.log 2>&1
  unset KNOX_WINDOW_DIRECT
  unset KNOX_SHUFFLED_ONLY
  ls -la gpurun_out/${TAG}_*.txt
fi
