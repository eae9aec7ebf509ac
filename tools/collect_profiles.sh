#!/bin/bash
# Round profile collection on the GPU box (writes under gpurun_out/; copy what is to be judged into profiles/).
#   bash tools/collect_profiles.sh r02
# 1. rocprofv3 --kernel-trace --stats of the default bench command
# 2. HBM traffic per query: separate --pmc FETCH_SIZE and --pmc WRITE_SIZE passes (MI355X_MICROARCH.md), per query
set -u
R=${1:-r03}
OUT=gpurun_out/$R
mkdir -p $OUT
export TMPDIR=/tmp
python3 bench.py --no-cpu-baseline > $OUT/bench_for_rows.json 2> $OUT/bench_for_rows.err
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --no-cpu-baseline --extra-queries "" --no-scan-form --no-hash-path > $OUT/bench_under_rocprof.json 2> $OUT/trace.log
python3 tools/pmc_summary.py $OUT/trace > $OUT/kernel_trace_summary.txt 2>&1
find $OUT/trace -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats.csv
# the same with ONE lane (queries of a step one behind the other on one stream): there every launch of a kernel has the chip to itself,
# and the average duration in the stats is the kernel's own (under the default the overlapped steps' launches share the chip)
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_lanes1 -- python3 bench.py --no-cpu-baseline --extra-queries "" --no-scan-form --no-hash-path --lanes 1 > $OUT/bench_under_rocprof_lanes1.json 2> $OUT/trace_lanes1.log
python3 tools/pmc_summary.py $OUT/trace_lanes1 > $OUT/kernel_trace_summary_lanes1.txt 2>&1
find $OUT/trace_lanes1 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats_lanes1.csv
ITERS=5
for q in q1 q3 q5 q6 q9; do
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 600 rocprofv3 --pmc $c --output-format csv -d $OUT/pmc/${q}_$c -- python3 tools/run_queries.py --sf 10 --queries $q --iters $ITERS > $OUT/pmc_${q}_$c.log 2>&1
  done
done
# the reference-width leg (bench.py `reference_width`: 8-byte columns, fixed-shape kernels): what its kernels really move
for q in q1 q3 q6 q5 q9; do
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 600 rocprofv3 --pmc $c --output-format csv -d $OUT/pmc_refwidth/${q}_$c -- python3 tools/run_queries.py --sf 10 --queries $q --iters $ITERS --reference-width > $OUT/pmc_refwidth_${q}_$c.log 2>&1
  done
done
ROWS=$(python3 -c "import json;print(json.dumps(json.load(open('$OUT/bench_for_rows.json'))['config']['rows_per_gpu']))")
python3 tools/pmc_per_query.py $OUT/pmc $ITERS "$ROWS" $OUT/pmc_traffic.json > $OUT/pmc_traffic_summary.txt 2>&1
python3 tools/pmc_per_query.py $OUT/pmc_refwidth $ITERS "$ROWS" $OUT/pmc_traffic_reference_width.json > $OUT/pmc_traffic_reference_width_summary.txt 2>&1
bash tools/collect_query_traces.sh $R "$ROWS"
# drop the bulky raw traces, keep the summaries
find $OUT -name "*.csv" -size +2M -delete
du -sh $OUT
