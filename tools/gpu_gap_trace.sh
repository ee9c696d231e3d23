#!/bin/bash
# Kernel-trace timeline of a few steady-state steps: per-kernel duration and the idle gap before it (single lane), ws on/off.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
E=$R/gpurun_out/gaps
mkdir -p $E
for v in ws nows; do
  if [ $v = nows ]; then export TRTX_CONV_NOWS=1; else unset TRTX_CONV_NOWS; fi
  (cd /tmp && TRTX_LANES=1 timeout 300 rocprofv3 --kernel-trace --output-format csv -d $E/t_$v -o t -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $E/$v.log 2>&1)
  f=$(find $E/t_$v -name '*kernel_trace.csv' | head -1)
  python $R/tools/gap_table.py $f > $E/gaps_$v.txt 2>&1
  tail -4 $E/gaps_$v.txt
  rm -rf $E/t_$v
done
