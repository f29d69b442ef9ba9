#!/bin/bash
# rocprofv3 kernel trace of one arrangement of tools/partition_probe.py (GPU box, repository root):
#   bash tools/ubench/partition_prof.sh chain 8
# (r04: this wrapper existed only on the GPU box and its run ended in a SIGSEGV at interpreter exit - cause and fix: partition_probe.py,
# destroy_masked_streams.  The program goes directly after `--`.)
R=$(pwd); mode=${1:-chain}; k=${2:-8}; O=$R/gpurun_out/partition_prof; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$mode -o out -- python3 $R/tools/partition_probe.py --only $mode --k $k --iters 3 > $O/$mode.log 2>&1
echo "exit code $?" >> $O/$mode.log
tail -5 $O/$mode.log
