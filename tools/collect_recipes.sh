#!/bin/bash
# Keep the sources of the kernels the configured workloads specialise at run time (GPU box): bench.py's queries at SF=10 and smoke()'s at
# SF=0.01, from an empty cache so that every kernel is compiled — and so kept.  Copy gpurun_out/<R>/recipes/*.hip into
# sdqlpy_amd/jit_recipes/ afterwards; build() compiles them ahead of time (__graft_entry__._compile_recipes).
#   bash tools/collect_recipes.sh r04
R=${1:-r05}
OUT=gpurun_out/$R/recipes
rm -rf $OUT /tmp/jit_empty; mkdir -p $OUT
rm -f sdqlpy_amd/jit_recipes/*.hip          # (this box's copy of the tree: smoke()'s build() would compile the kept ones again, and so keep the stale among them)
export SDQLPY_AMD_JIT_RECIPES=$OUT SDQLPY_AMD_JIT_CACHE=/tmp/jit_empty
python3 bench.py --no-cpu-baseline --steps 3 --warmup 1 --steady-steps 0 > /dev/null 2> $OUT/../recipes_bench.err
python3 bench.py --force-dist --no-cpu-baseline --steps 3 --warmup 1 --steady-steps 0 > /dev/null 2> $OUT/../recipes_bench_dist.err      # the partitioned join's programs (rebuild with a gate, the probe side's stage)
python3 __graft_entry__.py --smoke > $OUT/../recipes_smoke.log 2>&1
ls $OUT | wc -l
