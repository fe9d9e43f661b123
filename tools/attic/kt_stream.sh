cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/ks; rm -rf $O; mkdir -p $O; cd $R
rocprofv3 --kernel-trace --stats -d $O/kt -o b -- python bench.py --config 2 --steps 20 --warmup 2 --no-cpu-baseline > $O/bench.json 2> $O/bench.err
python tools/rocprof_summary.py $(find $O/kt -name "*.db" | head -1) | head -24 | cut -c1-60,92-150
python tools/kq.py < $O/bench.json
