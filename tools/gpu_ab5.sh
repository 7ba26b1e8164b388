cd /tmp && export TMPDIR=/tmp HL_BENCH_DIR=/tmp/hlb && mkdir -p /tmp/hlb && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r04k
( timeout -k 5 200 python -u tools/slice_probe.py C3 0 3 2>&1 | grep --line-buffered -E "^rep|rror" | cut -c1-300
  timeout -k 10 700 python -u tools/c5_ab.py C5 17 60 - HLMI_LONG_BLOCKS=1024 HLMI_LONG_BLOCKS=512 HLMI_LONG_MAIN_STREAM=1 2>&1 | grep --line-buffered -vE "^\s*$" | cut -c1-700 ) | tee gpurun_out/r04k/ab5.txt
