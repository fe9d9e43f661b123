#!/bin/bash
# the e2e leg of bench.py against the writer's formatting threads (0 = the library's default), with the writer's phase times.   gpurun -- bash tools/e2e_writer_sweep.sh [tag] "0 8 16"
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-e2e_writer}; mkdir -p $O; cd $R
for t in ${2:-0 8 16}; do
  TRGT_WRITER_TRACE=1 BENCH_WRITER_THREADS=$t python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-streaming --no-legs --detail $O/detail_$t.json > $O/line_$t.json 2> $O/err_$t.log
  python - $O/detail_$t.json $t <<'PY'
import json, sys
e = json.load(open(sys.argv[1]))["e2e"]
print("writer threads %s:" % sys.argv[2], {k: e.get(k) for k in ("ingest_loci_per_s", "write_loci_per_s", "write_loci_per_s_device_deflate", "pipeline_loci_per_s", "pipeline_loci_per_s_device_ingest_host_deflate")}, e["pipeline_stage_ms_per_chunk"]["device_ingest_device_deflate"])
PY
  grep "\[writer\]" $O/err_$t.log | tail -4
done 2>&1 | tee $O/summary.txt
