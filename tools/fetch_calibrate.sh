#!/bin/bash
# On the GPU box: build tools/fetch_calibrate.hip, run it under the two HBM counter passes, summarise (tools/fetch_calibrate.py).
set -o pipefail
export TMPDIR=/tmp
O=$PWD/gpurun_out/fetch_cal
mkdir -p $O
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/fetch_calibrate.hip -o $O/fetch_cal || exit 1
$O/fetch_cal > $O/expect.txt || exit 2
timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/f -o f -- $O/fetch_cal > $O/f.log 2>&1 || exit 3
timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/w -o w -- $O/fetch_cal > $O/w.log 2>&1 || exit 4
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o kt -- $O/fetch_cal > $O/kt.log 2>&1 || exit 5
find $O/f -name "*counter_collection.csv" -exec cp {} $O/pmc_fetch.csv \;
find $O/w -name "*counter_collection.csv" -exec cp {} $O/pmc_write.csv \;
find $O/kt -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
rm -rf $O/f $O/w $O/kt $O/fetch_cal
python3 tools/fetch_calibrate.py $O | tee $O/summary.txt
