#!/bin/bash
# Run on the GPU box (via gpurun): bench line, kernel-trace stats and the two HBM counter passes for one tag.
#   bash tools/profile_round.sh TAG [bench args...]
set -o pipefail
TAG=$1; shift
export TMPDIR=/tmp
O=$PWD/gpurun_out/prof_$TAG
mkdir -p $O
timeout -k 10 400 python3 bench.py --steps 20 --warmup 3 "$@" > $O/bench.json 2> $O/bench.err || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o kt -- python3 bench.py --steps 10 --warmup 2 --cpu-packets 0 "$@" > $O/kt.log 2>&1 || exit 2
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -o fetch -- python3 bench.py --steps 3 --warmup 1 --cpu-packets 0 "$@" > $O/fetch.log 2>&1 || exit 3
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -o write -- python3 bench.py --steps 3 --warmup 1 --cpu-packets 0 "$@" > $O/write.log 2>&1 || exit 4
find $O -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
find $O/fetch -name "*counter_collection.csv" -exec cp {} $O/pmc_fetch_size.csv \;
find $O/write -name "*counter_collection.csv" -exec cp {} $O/pmc_write_size.csv \;
rm -rf $O/kt $O/fetch $O/write
ls -la $O
