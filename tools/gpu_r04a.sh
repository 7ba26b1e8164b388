set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04a
nproc > gpurun_out/r04a/nproc.txt; python -c "import os; print(len(os.sched_getaffinity(0)))" >> gpurun_out/r04a/nproc.txt; free -g >> gpurun_out/r04a/nproc.txt
timeout -k 10 900 python -m pytest tests/test_gpu_workloads_oracle.py -x -q -s > gpurun_out/r04a/test_workloads.log 2>&1 || echo "TEST FAILED" >> gpurun_out/r04a/test_workloads.log
tail -5 gpurun_out/r04a/test_workloads.log
timeout -k 10 600 python bench.py > gpurun_out/r04a/bench_c3.json 2> gpurun_out/r04a/bench_c3.err || echo "BENCH FAILED"
tail -c 1500 gpurun_out/r04a/bench_c3.json
./tools/bin/valu_calib > gpurun_out/r04a/valu_calib.txt
timeout -k 10 120 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d gpurun_out/r04a/calib_pmc -o p --output-format csv -- ./tools/bin/valu_calib > gpurun_out/r04a/valu_calib_pmc.txt 2>&1 || true
cat gpurun_out/r04a/valu_calib.txt
