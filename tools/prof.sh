#!/bin/bash
# tools/prof.sh <tag> <round-dir> [passes...]   -- rocprofv3 kernel trace + PMC passes of the default bench (run on the GPU box)
# Extra environment (DEME_ARITH, DEME_HIP_LIB, BENCH_ARGS) is passed through.  Summaries land in gpurun_out/<round-dir>/.
TAG=$1; RD=$2; shift 2
PASSES=${@:-"trace sqA sqB fetch write tcp ea"}
OUT=$PWD/gpurun_out/$RD; mkdir -p $OUT
ROOT=$PWD
export TMPDIR=/tmp
ARGS=${BENCH_ARGS:-"--steps 80 --warmup 10 --no-cpu-baseline --state-cache /tmp/deme_bed_${TAG}.npz"}
python $ROOT/bench.py $ARGS > /dev/null 2>&1   # builds the cached bed (untimed, unprofiled)
declare -A PMC
PMC[sqA]="SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES"
PMC[sqB]="SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_INST_LEVEL_VMEM"
PMC[fetch]="FETCH_SIZE GRBM_GUI_ACTIVE"
PMC[write]="WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"
PMC[tcp]="TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum"
PMC[ta]="TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum"
PMC[lds]="SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INST_LEVEL_LDS"
PMC[ea]="TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum"
cd /tmp
for P in $PASSES; do
  rm -rf /tmp/prof_$P
  if [ "$P" = trace ]; then
    rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$P -o p -- python $ROOT/bench.py $ARGS > $OUT/${TAG}_bench.json 2> /tmp/prof_$P.err
    f=$(find /tmp/prof_$P -name 'p_kernel_trace.csv' | head -1)
    python $ROOT/profiles/summarize.py $f $OUT/${TAG}_kernels.txt 60 > /dev/null
    cat $OUT/${TAG}_kernels.txt | cut -c1-150
  else
    if [ "$P" = fetch ] || [ "$P" = write ]; then export DEME_PMC_CALIB=1; else unset DEME_PMC_CALIB; fi
    rocprofv3 --kernel-trace --pmc ${PMC[$P]} --output-format csv -d /tmp/prof_$P -o p -- python $ROOT/bench.py $ARGS > /tmp/prof_$P.json 2> /tmp/prof_$P.err
    f=$(find /tmp/prof_$P -name 'p_counter_collection.csv' | head -1)
    if [ -z "$f" ]; then echo "pass $P produced no counters"; tail -5 /tmp/prof_$P.err; continue; fi
    python $ROOT/profiles/summarize_pmc.py $f $OUT/${TAG}_${P}_pmc.txt 60 | grep -E "^kernel|forces|tile|integrate|copyBuffer|index|elementwise" | cut -c1-260
  fi
done
