#!/bin/bash
mkdir -p gpurun_out/r02h
rocprofv3 --list-avail > gpurun_out/r02h/list_avail.txt 2>&1
grep -oE "(SQ_[A-Z_0-9]+|TCC_[A-Za-z_0-9]+|TCP_[A-Za-z_0-9]+|TA_[A-Za-z_0-9]+|FETCH_SIZE|WRITE_SIZE|MemUnitStalled|L2CacheHit|LDSBankConflict|MeanOccupancy[A-Za-z]*|OccupancyPercent|MemUnitBusy|VALUBusy|SALUBusy|LdsUtil[A-Za-z]*)" gpurun_out/r02h/list_avail.txt | sort -u > gpurun_out/r02h/counters.txt
wc -l gpurun_out/r02h/counters.txt
rm -f gpurun_out/r02h/list_avail.txt
