#!/bin/bash
# Run on the GPU box (via gpurun): per-wave instruction / cycle counters of the encode and decode kernels, fused and
# separate launches -> gpurun_out/pmc_mix/instruction_mix.json   (bash tools/pmc_instruction_mix.sh)
set -o pipefail
export TMPDIR=/tmp
O=$PWD/gpurun_out/pmc_mix
rm -rf $O; mkdir -p $O
SETS=("SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES" "SQ_INSTS_BRANCH SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES")
for mode in fused unfused; do
  if [ $mode = unfused ]; then export ALAC_HIP_FUSED=0 ALAC_HIP_DEC_FUSED=0; fi
  i=0
  for s in "${SETS[@]}"; do
    timeout -k 10 300 rocprofv3 --pmc $s --output-format csv -d $O/${mode}_$i -o pmc -- python3 bench.py --steps 2 --warmup 1 --cpu-packets 0 > $O/${mode}_$i.log 2>&1 || exit 1
    find $O/${mode}_$i -name "*counter_collection.csv" -exec cp {} $O/${mode}_$i.csv \;
    rm -rf $O/${mode}_$i
    i=$((i+1))
  done
done
python3 tools/pmc_instruction_mix.py $O > $O/instruction_mix.json && ls -la $O
