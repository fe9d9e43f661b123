#!/bin/bash
# what the e2e leg measures inside bench runs of different shapes (which earlier part of a run slows it down).   gpurun -- bash tools/e2e_in_bench.sh
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/e2e_in_bench; mkdir -p $O; cd $R
i=0
for flags in "--no-cpu-baseline --no-streaming --no-legs" "--no-streaming --no-legs" "--no-cpu-baseline --no-legs" "--no-cpu-baseline --no-streaming" ""; do
  i=$((i+1))
  python bench.py --steps 20 --warmup 3 $flags --detail $O/d$i.json > $O/l$i.json 2> $O/e$i.log
  python - $O/d$i.json "$flags" <<'PY'
import json, sys
e = json.load(open(sys.argv[1]))["e2e"]
print("[%s] ingest %.0f gpu %.0f write_dd %.0f pipeline %.0f  stages %s" % (sys.argv[2], e["ingest_loci_per_s"], e["gpu_loci_per_s"], e["write_loci_per_s_device_deflate"], e["pipeline_loci_per_s"], e["pipeline_stage_ms_per_chunk"]["device_ingest_device_deflate"]))
PY
done 2>&1 | tee $O/summary.txt
