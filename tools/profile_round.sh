#!/bin/bash
# Run on the GPU box (via gpurun): bench line, kernel-trace stats, the two HBM counter passes and the two instruction
# counter passes for ONE shape (one command = one shape: --no-legs, one encode pass per step).
#   bash tools/profile_round.sh TAG [bench args...]      e.g.  r03_10k      /  r03_125k --packets 125000  /  r03_24 --bit-depth 24
# then here:  python tools/pmc_tables.py gpurun_out/prof_TAG KEY --passes 4 --decode-passes 4   (KEY = 16bit_stereo_10000 ...)
set -o pipefail
TAG=$1; shift
export TMPDIR=/tmp
O=$PWD/gpurun_out/prof_$TAG
mkdir -p $O
python3 -c "import alac_amd; print(alac_amd.source_fingerprint())" > $O/fingerprint.txt || exit 9
COMMON="--no-legs --repeats 1 --cpu-packets 0"
timeout -k 10 500 python3 bench.py --steps 20 --warmup 3 --no-legs "$@" > $O/bench.json 2> $O/bench.err || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o kt -- python3 bench.py --steps 10 --warmup 2 $COMMON "$@" > $O/kt.log 2>&1 || exit 2
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -o fetch -- python3 bench.py --steps 3 --warmup 1 $COMMON "$@" > $O/fetch.log 2>&1 || exit 3
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -o write -- python3 bench.py --steps 3 --warmup 1 $COMMON "$@" > $O/write.log 2>&1 || exit 4
timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d $O/i0 -o i0 -- python3 bench.py --steps 3 --warmup 1 $COMMON "$@" > $O/i0.log 2>&1 || exit 5
timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_WAVE_CYCLES --output-format csv -d $O/i1 -o i1 -- python3 bench.py --steps 3 --warmup 1 $COMMON "$@" > $O/i1.log 2>&1 || exit 6
find $O/kt -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
find $O/fetch -name "*counter_collection.csv" -exec cp {} $O/pmc_fetch_size.csv \;
find $O/write -name "*counter_collection.csv" -exec cp {} $O/pmc_write_size.csv \;
find $O/i0 -name "*counter_collection.csv" -exec cp {} $O/pmc_insts_0.csv \;
find $O/i1 -name "*counter_collection.csv" -exec cp {} $O/pmc_insts_1.csv \;
rm -rf $O/kt $O/fetch $O/write $O/i0 $O/i1
ls -la $O
