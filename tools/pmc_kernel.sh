#!/bin/bash
# Counter passes (separate runs, MI355X_MICROARCH.md) over one query:  bash tools/pmc_kernel.sh q9 gpurun_out/r02i/q9pmc
Q=${1:-q9}; OUT=${2:-gpurun_out/pmc_$Q}; ITERS=${3:-3}
mkdir -p $OUT
export TMPDIR=/tmp
i=0
for grp in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum" \
           "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_LDS_ATOMIC" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" \
           "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum" \
           "SQ_WAVES SQ_LEVEL_WAVES SQ_INSTS_VALU_INT64 SQ_INSTS_VMEM_WR SQ_INSTS_SMEM"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --output-format csv -d $OUT/p$i -- python3 tools/run_queries.py --sf 10 --queries $Q --iters $ITERS > $OUT/p$i.log 2>&1
done
python3 tools/pmc_summary.py $OUT > $OUT/summary.txt 2>&1
find $OUT -name "*.csv" -delete
find $OUT -name "*.log" -size +64k -delete
grep -E "k_lookup_agg|k_probe_agg|k_stage|k_build_lookup|k_key_set|k_dense|xk_" $OUT/summary.txt | cut -c1-400
