#!/bin/bash
# Round profile on the GPU box: rocprofv3 kernel-trace stats + separate PMC passes of ONE bench.py edit (2 denoise steps for
# the PMC passes, 50 for the stats).  Outputs under gpurun_out/prof/; tools/summarize_profile.py turns them into profiles/*.
# Usage (via gpurun): bash tools/profile_round.sh [TAG [extra bench.py arguments ...]]
#   e.g. bash tools/profile_round.sh c3 --batch 8 ; bash tools/profile_round.sh c5 --res 768 --batch 4 ; bash tools/profile_round.sh b2 --batch 2
#   (outputs then go to gpurun_out/prof_TAG/; publish with  python3 tools/summarize_profile.py --publish gpurun_out/prof_TAG rN_TAG)
TAG="$1"; [ $# -gt 0 ] && shift
EXTRA="$*"
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/prof${TAG:+_$TAG}
rm -rf $OUT && mkdir -p $OUT
B="python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-e2e --no-configs --no-calibration --table $EXTRA"
rocprofv3 --kernel-trace --stats -d $OUT/stats -o stats --output-format csv -- $B > $OUT/stats_bench.json 2> $OUT/stats.log
P="python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-e2e --no-configs --no-calibration --denoise-steps 2 $EXTRA"
rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc_fetch -o f --output-format csv -- $P > /dev/null 2> $OUT/pmc_fetch.log
rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc_write -o w --output-format csv -- $P > /dev/null 2> $OUT/pmc_write.log
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_LDS_IDX_ACTIVE -d $OUT/pmc_mfma -o m --output-format csv -- $P > /dev/null 2> $OUT/pmc_mfma.log
ls -R $OUT | head -40
# keep the merged-back payload small: aggregate the counter CSVs here
python3 tools/summarize_profile.py $OUT
rm -rf $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_mfma
find $OUT/stats -name "*kernel_trace.csv" -delete
