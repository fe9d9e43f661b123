#!/bin/bash
# the e2e leg of bench.py (synthetic BAM -> ingestion -> GPU -> VCF + spanning BAM) under several ingestion settings on ONE box:
# "callers:waves" pairs (waves 0 = the library's default).   gpurun -- bash tools/e2e_sweep.sh [tag] "3:0 3:16 5:16"
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-e2e_sweep}; mkdir -p $O; cd $R
for cw in ${2:-3:0 3:16 5:12 5:16}; do
  c=${cw%%:*}; w=${cw##*:}
  BENCH_INGEST_CALLERS=$c BENCH_INFLATE_WAVES=$w python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-streaming --no-legs --detail $O/detail_${c}_${w}.json > $O/line_${c}_$w.json 2> $O/err_${c}_$w.log
  python - $O/detail_${c}_${w}.json $c $w <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
e = d.get("e2e", d.get("config", {}).get("e2e", {}))
keys = ["ingest_loci_per_s", "ingest_loci_per_s_one_caller", "ingest_loci_per_s_host", "gpu_loci_per_s", "write_loci_per_s_device_deflate", "pipeline_loci_per_s", "pipeline_loci_per_s_device_deflate", "pipeline_loci_per_s_host_ingest"]
print("callers %s waves %s:" % (sys.argv[2], sys.argv[3]), {k: e.get(k) for k in keys if k in e} or sorted(e.keys())[:40])
PY
done 2>&1 | tee $O/summary.txt
