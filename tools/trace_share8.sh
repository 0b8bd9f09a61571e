#!/bin/bash
# timeline of a step: rocprofv3 kernel trace of the last two steps + the host-side stamps of the sdust call
#   bash tools/trace_share8.sh <tag> [extra bench args, default "--rank-share 8,1"]
# (round 6: the CORNETTO_SDUST_* / CORNETTO_SIFT_* switches this script sets exist in the development build of the library only)
export CORNETTO_LIB=${CORNETTO_LIB:-$PWD/cornetto_amd/libcornetto_hip_dev.so}
R=${GRAFT_REPO_ROOT:-/root/repo}; TAG=${1:-r05}; shift
X=${*:---rank-share 8,1}
Q="--steps 8 --warmup 2 --no-cpu --no-e2e --no-reads --no-profiles --check-steps 0 --emulate-ranks= $X"
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_share8t
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof_share8t -- python3 $R/bench.py $Q > $R/gpurun_out/${TAG}_trace.log 2>&1
cd $R
f=$(find gpurun_out/prof_share8t -name '*kernel_trace.csv' | head -1)
python3 tools/trace_gaps.py $f 2 > gpurun_out/${TAG}_timeline.txt
rm -rf gpurun_out/prof_share8t
CORNETTO_SDUST_TRACE=1 timeout 600 python3 bench.py $Q 2> gpurun_out/${TAG}_stamps.txt | cut -c1-400
tail -12 gpurun_out/${TAG}_stamps.txt
