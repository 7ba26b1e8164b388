#!/bin/bash
# kernel trace of one bench step: every kernel by total time (tools/trace_hist.py); usage (through gpurun): tools/gpu_trace.sh [workload=C3]
WL=${1:-C3}
cd /tmp && export TMPDIR=/tmp HL_BENCH_DIR=/tmp/hlb && mkdir -p /tmp/hlb && cd $GRAFT_REPO_ROOT && out=$PWD/gpurun_out && mkdir -p $out
rm -rf $out/trace_$WL
timeout -k 10 500 rocprofv3 --kernel-trace -d $out/trace_$WL -o p --output-format csv -- python bench.py --workload $WL --steps 1 --warmup 1 --no-cpu-baseline --no-graph > $out/trace_$WL.log 2>&1
f=$(find $out/trace_$WL -name "p_kernel_trace.csv")
python tools/trace_hist.py $f > $out/trace_${WL}_kernels.txt
find $out -name "*.csv" -size +1M -delete
cut -c1-150,170-230 $out/trace_${WL}_kernels.txt | head -60
