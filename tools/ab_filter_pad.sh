#!/bin/bash
# A/B of the pre-filter's text-window layout on ONE box: the default against a build with the windows padded by one dword per 32
# (make -C trgt_amd/csrc EXTRA=-DTRGT_FLT_PAD OBJDIR=pad OUT=../libtrgt_hip_pad.so): parity tests of the padded build, one-context kernel
# durations alternating, LDS counters of both.   gpurun -- bash tools/ab_filter_pad.sh
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/ab_filter_pad; mkdir -p $O; cd $R
PAD=$R/trgt_amd/libtrgt_hip_pad.so
{
echo "== parity of the padded build"; TRGT_HIP_LIB=$PAD python -m pytest tests/test_filter_gpu.py tests/test_windows_gpu.py tests/test_long_reads_gpu.py -q -m gpu 2>&1 | tail -2
for rep in 1 2; do for lib in default pad; do
  if [ $lib = pad ]; then export TRGT_HIP_LIB=$PAD; else unset TRGT_HIP_LIB; fi
  rm -rf $O/kt; rocprofv3 --kernel-trace --stats -d $O/kt -o b -- python bench.py --config 2 --steps 20 --warmup 2 --contexts 1 --no-streaming --no-cpu-baseline > $O/bench_$lib.json 2> $O/bench.err
  echo "== $lib (run $rep): $(tail -1 $O/bench_$lib.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('one-context value', d['value'], 'ms/step', d['ms_per_step'], 'parity', d.get('parity'))")"
  python tools/rocprof_summary.py $(find $O/kt -name "*.db" | head -1) | grep "wfa_filter" | cut -c1-60,92-150
done; done
for lib in default pad; do
  if [ $lib = pad ]; then export TRGT_HIP_LIB=$PAD; else unset TRGT_HIP_LIB; fi
  for set in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS"; do
    rocprofv3 --kernel-trace --pmc $set --kernel-include-regex "wfa_filter" -d $O/p_$lib -o p --output-format csv -- python bench.py --config 2 --steps 1 --warmup 0 --no-cpu-baseline --no-legs --no-e2e --no-streaming > $O/p_$lib.log 2>&1
  done
  echo "== $lib: counters per dispatch"; python tools/pmc_summary.py $(find $O/p_$lib -name "*counter_collection.csv") | grep -A5 "wfa_filter_kernel<[45], 2, 2, 6>"
done
} 2>&1 | tee $O/summary.txt
