#!/bin/bash
# A/B of the run-time skeletons on one query: every variant in a process of its own (macros are part of the specialised source).
#   bash tools/x_exp.sh q3 gpurun_out/r04/exp_q3.txt "X8_PIPE=1 XV_PIPE=1" "x_waves=12" ...
# each further argument is one variant: NAME=value in UPPER case are macros (SDQLPY_AMD_X_DEFINES), a bare UPPER-case word W sets
# SDQLPY_AMD_X_W=1 in the environment, lower case name=value are options (sdqh_set_option)
Q=$1; OUT=$2; shift 2
mkdir -p $(dirname $OUT)
: > $OUT
for variant in "base" "$@"; do
  defs=""; opts=""; envs=""
  if [ "$variant" != "base" ]; then
    for w in $variant; do
      case "$w" in
        [A-Z]*=*) defs="$defs $w";;
        [A-Z]*) envs="$envs SDQLPY_AMD_X_$w=1";;
        *) opts="$opts,$w";;
      esac
    done
  fi
  echo "######## variant: $variant" >> $OUT
  env $envs SDQLPY_AMD_X_DEFINES="$defs" SDQLPY_AMD_JIT_CACHE=/tmp/jit_exp python3 tools/probe.py --sf ${SF:-10} --queries $Q --iters ${ITERS:-9} --configs "${opts#,}" 2>&1 | grep -E "xk_|k_stage|k_probe|k_lookup_agg|k_build_lookup|sum of kernels|wall|Error|error" >> $OUT
done
cat $OUT
