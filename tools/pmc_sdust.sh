#!/bin/bash
# SQ counters of the sdust main kernel (separate rocprofv3 --pmc passes, no other tracing), its FETCH_SIZE / WRITE_SIZE, and the
# FETCH_SIZE calibration on known byte counts (tools/ubench/fetch_calib).  Writes tracked-size summaries:
#   gpurun_out/<tag>_sq_sdust.json   gpurun_out/<tag>_pmc_traffic.json      (copy them to profiles/)
#   bash tools/pmc_sdust.sh <tag> [mbases] [profile]
# (round 6: the CORNETTO_SDUST_* / CORNETTO_SIFT_* switches this script sets exist in the development build of the library only)
export CORNETTO_LIB=${CORNETTO_LIB:-$PWD/cornetto_amd/libcornetto_hip_dev.so}
TAG=${1:-r02}
MB=${2:-3160}
PROFILE=${3:-uniform}
R=$PWD
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
PROBE="python3 $R/tools/perf_probe.py sdust --mbases $MB --features 1 --reps 1 --profile $PROFILE"
CORNETTO_SDUST_STATS=1 $PROBE 2> $R/gpurun_out/${TAG}_stats.txt > /dev/null
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" "SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM" "SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH SQ_WAVES" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rm -rf $R/gpurun_out/pmc_${TAG}_s$i
  timeout 300 rocprofv3 --pmc $set -d $R/gpurun_out/pmc_${TAG}_s$i --output-format csv -- $PROBE > $R/gpurun_out/pmc_${TAG}_s$i.log 2>&1
done
rm -rf $R/gpurun_out/pmc_${TAG}_fetch $R/gpurun_out/pmc_${TAG}_write $R/gpurun_out/pmc_${TAG}_calib
timeout 300 rocprofv3 --pmc FETCH_SIZE -d $R/gpurun_out/pmc_${TAG}_fetch --output-format csv -- $PROBE > $R/gpurun_out/pmc_${TAG}_fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE -d $R/gpurun_out/pmc_${TAG}_write --output-format csv -- $PROBE > $R/gpurun_out/pmc_${TAG}_write.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE -d $R/gpurun_out/pmc_${TAG}_calib --output-format csv -- $R/tools/ubench/fetch_calib > $R/gpurun_out/pmc_${TAG}_calib.log 2>&1
cd $R
python3 tools/pmc_summarize.py "$TAG" "$MB" "$PROFILE"
