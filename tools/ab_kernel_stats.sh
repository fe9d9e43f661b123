# ab_kernel_stats.sh CONFIG PATTERN: one-context kernel durations (rocprofv3) of the working library and of trgt_amd/libtrgt_hip_prev.so, alternating, same box
CFG=${1:-4}; PAT=${2:-ppl}
for rep in 1 2; do
  unset TRGT_HIP_LIB; echo "== new"; bash tools/one_context_stats.sh $CFG 2>&1 | grep -i "$PAT" | cut -c1-50,60-110
  export TRGT_HIP_LIB=$PWD/trgt_amd/libtrgt_hip_prev.so; echo "== prev"; bash tools/one_context_stats.sh $CFG 2>&1 | grep -i "$PAT" | cut -c1-50,60-110
done
