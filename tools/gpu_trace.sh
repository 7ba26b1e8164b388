cd /tmp && export TMPDIR=/tmp HL_BENCH_DIR=/tmp/hlb && mkdir -p /tmp/hlb && cd $GRAFT_REPO_ROOT && out=$PWD/gpurun_out && mkdir -p $out
rm -rf $out/r04t_trace
timeout -k 10 500 rocprofv3 --kernel-trace -d $out/r04t_trace -o p --output-format csv -- python bench.py --workload C3 --steps 1 --warmup 1 --no-cpu-baseline --no-graph > $out/r04t_trace.log 2>&1
f=$(find $out/r04t_trace -name "p_kernel_trace.csv")
python tools/trace_hist.py $f > $out/r04t_all.txt
python tools/trace_neigh.py $f scan_impl 1000 > $out/r04t_scan.txt
find $out -name "*.csv" -size +1M -delete
cut -c1-200 $out/r04t_scan.txt | head -120
