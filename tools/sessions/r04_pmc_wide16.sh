#!/bin/bash
# counter passes over the 16-wide sweeps: LDS conflicts / waits, wave cycles, HBM bytes
# usage: bash tools/sessions/r04_pmc_wide16.sh  (on the GPU box; results under gpurun_out/r04pmc16)
export TMPDIR=/tmp
O=gpurun_out/r04pmc16
mkdir -p $O
SQ="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
SQ2="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES"
for cfg in "sym16:--symmetric one-triangle --rhs 16" "wide16:--rhs 16" "trans16:--trans T --rhs 16"; do
  name=${cfg%%:*}; args=${cfg#*:}
  for pass in "sq:$SQ" "sq2:$SQ2" "fetch:FETCH_SIZE" "write:WRITE_SIZE"; do
    pn=${pass%%:*}; ctr=${pass#*:}
    timeout -k 10 300 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $O/${name}_$pn -o p -- python3 bench.py $args --steps 3 --warmup 1 --no-cpu-baseline --no-warm-build > $O/${name}_$pn.json 2> $O/${name}_$pn.err || echo "FAILED $name $pn"
    echo "$name $pn done"
  done
done
