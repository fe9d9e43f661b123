cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/kt; mkdir -p $O; cd $R
rocprofv3 --kernel-trace --stats -d $O/kt -o bench -- python bench.py --steps 3 --warmup 1 > $O/bench.json 2> $O/bench.err
python tools/rocprof_summary.py $(find $O/kt -name "*.db" | head -1) | cut -c1-60,90-160 | head -14
