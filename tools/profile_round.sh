#!/bin/bash
# Evidence of a round on the MI355X box: kernel trace + stats, FETCH / WRITE
# passes, two SQ passes of the headline bench; summaries for profiles/.
#   usage (on the GPU box, from the repo root): bash tools/profile_round.sh r03z
# rocprofv3 gets the program itself after "--" (python3 bench.py ...), counters
# in runs of their own (no tracing flags next to --pmc).
set -x
TAG=${1:-r03}
export TMPDIR=/tmp
OUT=gpurun_out/$TAG
mkdir -p $OUT
B="python3 bench.py --no-cpu-baseline --no-secondary"
python3 bench.py --steps 10 --warmup 3 > $OUT/bench.json 2> $OUT/bench.err || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- $B --steps 2 --warmup 1 > $OUT/bench_under_rocprof.json 2> $OUT/trace.err || exit 1
rocprofv3 --output-format csv --pmc FETCH_SIZE -d $OUT/pmc_fetch -o f -- $B --steps 1 --warmup 0 --no-profile > /dev/null 2> $OUT/pmc_f.err || exit 1
rocprofv3 --output-format csv --pmc WRITE_SIZE -d $OUT/pmc_write -o w -- $B --steps 1 --warmup 0 --no-profile > /dev/null 2> $OUT/pmc_w.err || exit 1
rocprofv3 --output-format csv --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_ANY -d $OUT/sq_a -o a -- $B --steps 1 --warmup 0 --no-profile > /dev/null 2> $OUT/sq_a.err || exit 1
rocprofv3 --output-format csv --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $OUT/sq_b -o b -- $B --steps 1 --warmup 0 --no-profile > /dev/null 2> $OUT/sq_b.err || exit 1
python3 tools/pmc_summary.py $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_traffic.json
python3 tools/sq_summary.py $OUT/sq_counters.json $OUT/sq_a $OUT/sq_b > $OUT/sq_counters_summary.txt
STATS=$(find $OUT/trace -name "*kernel_stats.csv" | head -1)
cp "$STATS" $OUT/kernel_stats.csv
TRACE=$(find $OUT/trace -name "*kernel_trace.csv" | head -1)
python3 tools/timeline.py "$TRACE" > $OUT/timeline_one_step.txt 2>/dev/null
# the raw counter dumps are large: keep the summaries
rm -rf $OUT/pmc_fetch $OUT/pmc_write $OUT/sq_a $OUT/sq_b
find $OUT/trace -name "*.csv" ! -name "*kernel_stats.csv" -size +8M -delete
ls -la $OUT
