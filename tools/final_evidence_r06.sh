set -u
mkdir -p gpurun_out/r06
export SDQLPY_COMMIT=${SDQLPY_COMMIT:-unknown}
bash tools/collect_profiles.sh r06 > gpurun_out/collect_r06.log 2>&1
G=gpurun_out/r06
python bench.py --steps 20 --warmup 5 > $G/bench.json 2> $G/bench.err
python bench.py --force-dist --trivial-collectives --no-cpu-baseline > $G/bench_dist_issued.json 2> $G/bench_dist_issued.err
python bench.py --force-dist --no-cpu-baseline > $G/bench_dist_skipped.json 2> $G/bench_dist_skipped.err
SDQLPY_AMD_DIST_GRAPHS=0 SDQLPY_AMD_DIST_LANES=0 python bench.py --force-dist --trivial-collectives --no-cpu-baseline > $G/bench_dist_issued_calls.json 2> /dev/null
python bench.py --sf 100 --queries q5,q9 --no-cpu-baseline --no-reference-width --no-scan-form --no-hash-path --extra-queries "" --steps 10 > $G/bench_sf100_q5_q9.json 2> $G/bench_sf100_q5_q9.err
python bench.py --sf 100 --no-cpu-baseline --no-reference-width --no-scan-form --no-hash-path --extra-queries "" --steps 10 > $G/bench_sf100_step.json 2> $G/bench_sf100_step.err
for n in 2 8; do timeout 900 python bench.py --gpus $n --share-gpu --no-cpu-baseline --steps 10 > $G/bench_share_$n.json 2> $G/bench_share_$n.err; done
timeout 1200 python bench.py --gpus 8 --share-gpu --global-sf 100 --no-cpu-baseline --steps 5 > $G/bench_share_8_sf100.json 2> $G/bench_share_8_sf100.err
(timeout 900 python tools/fuzz_more.py 400 80; timeout 900 python tools/fuzz_walks.py 2000 16; timeout 900 python tools/sweep_queries.py 0.3,2; timeout 900 python tools/sweep_queries.py 0.3,2 "" direct_index=0,row_index=0,grouped_index=0,feature_min_rows=0,coarse_kb=1) > $G/fuzz_and_sweep.txt 2>&1
timeout 2400 python -m pytest tests -m gpu -q > $G/gpu_suite.log 2>&1
python -c "import __graft_entry__ as g; g.smoke()" > $G/smoke.log 2>&1
tail -3 $G/gpu_suite.log
