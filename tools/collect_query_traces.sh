#!/bin/bash
# rocprofv3 kernel trace of each hot-path query on its own -> sum of kernel durations per run (tools/trace_per_query.py)
#   bash tools/collect_query_traces.sh r02 '<rows json>'
set -u
R=${1:-r02}
ROWS=${2:-$(python3 -c "import json;print(json.dumps(json.load(open('profiles/r02_bench.json'))['config']['rows_per_gpu']))")}
OUT=gpurun_out/$R
mkdir -p $OUT/qtrace
export TMPDIR=/tmp
for q in q1 q3 q5 q6 q9; do
  timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT/qtrace/$q -- python3 tools/run_queries.py --sf 10 --queries $q --iters 8 > $OUT/qtrace_$q.log 2>&1
done
python3 tools/trace_per_query.py $OUT/qtrace q1,q3,q5,q6,q9 "$ROWS" > $OUT/per_query_kernel_sums.txt 2>&1
find $OUT/qtrace -name "*.csv" -size +1M -delete
find $OUT/qtrace -name "*.db" -delete
