#!/bin/bash
# Kernel timeline of one denoise step: bash tools/timeline.sh  (via gpurun)
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/timeline
rm -rf $OUT && mkdir -p $OUT
rocprofv3 --kernel-trace -d $OUT/trace -o t --output-format csv -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-e2e --no-configs --denoise-steps 6 > $OUT/bench.json 2> $OUT/bench.err
python3 tools/timeline.py $OUT/trace $OUT/timeline.json | head -150
find $OUT/trace -name "*.csv" -size +1M -delete
