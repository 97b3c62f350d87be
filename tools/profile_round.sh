#!/bin/bash
# One command for the round's profile evidence (run on the GPU box via gpurun): the rocprofv3 passes of tools/prof.sh over
# the default bench.py run, condensed into profiles/<tag>_rocprof_summary.txt, and the PMC-measured HBM traffic of the
# headline kernel into profiles/traffic.json (what bench.py reports as roofline.traffic, with its source).
# usage: tools/profile_round.sh r02   (writes gpurun_out/prof_r02/, gpurun_out/r02_rocprof_summary.txt, gpurun_out/traffic.json:
# copy the last two into profiles/ -- gpurun only merges gpurun_out/ back)
set -u
TAG=${1:-r03}
ROOT=$PWD
bash tools/prof.sh $TAG > gpurun_out/prof_$TAG.files 2>&1
python3 tools/prof_summary.py gpurun_out/prof_$TAG gpurun_out/traffic.json > gpurun_out/${TAG}_rocprof_summary.txt
tail -n 3 gpurun_out/${TAG}_rocprof_summary.txt
