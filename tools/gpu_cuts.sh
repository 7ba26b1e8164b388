# where the lanes align the set-aside pieces: tools/gpu_cuts.sh (through gpurun)
cd /tmp && export TMPDIR=/tmp HL_BENCH_DIR=/tmp/hlb && mkdir -p /tmp/hlb && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/cuts
for c in "40,70,90" "60" "50,85" "30,55,75,90" "40,70,90"; do
  echo "cuts $c" | tee -a gpurun_out/cuts/cuts.txt
  timeout -k 5 240 env HLMI_SET_ASIDE_CUTS=$c python -u tools/slice_probe.py C3 0 4 2>&1 | grep --line-buffered -E "^rep|rror|wall_s" | cut -c1-1500 | sed -E 's/"(align_tasks|align_tasks_dp|anchors|chain_groups[a-z_]*|pieces)": [0-9]+,? ?//g' | sed -u "s/^/[$c] /" | tee -a gpurun_out/cuts/cuts.txt || exit 1
done
