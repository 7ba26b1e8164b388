# Do two passes in flight use the GPU better than one?  Two processes run a C3 slice each at the same time; compare their step
# times with a process alone (tools/gpu_two_procs.sh, through gpurun).
cd /tmp && export TMPDIR=/tmp HL_BENCH_DIR=/tmp/hlb && mkdir -p /tmp/hlb && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/two
timeout -k 5 240 python -u tools/slice_probe.py C3 0 3 2>&1 | grep --line-buffered "^rep" | sed -u "s/^/solo /" | tee gpurun_out/two/two.txt || exit 1
(timeout -k 5 400 python -u tools/slice_probe.py C3 0 6 2>&1 | grep --line-buffered "^rep" | sed -u "s/^/P0 /" | tee -a gpurun_out/two/two.txt) &
P0=$!
(timeout -k 5 400 python -u tools/slice_probe.py C3 1 6 2>&1 | grep --line-buffered "^rep" | sed -u "s/^/P1 /" | tee -a gpurun_out/two/two1.txt) &
P1=$!
wait $P0; wait $P1
cat gpurun_out/two/two1.txt >> gpurun_out/two/two.txt
