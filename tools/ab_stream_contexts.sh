# streaming rates (reads in pinned host memory) against the resident rate, by contexts per GPU -- same box
for n in 4 6 8; do
  python bench.py --steps 40 --warmup 3 --no-cpu-baseline --no-legs --no-e2e --contexts $n 2>/dev/null | tail -1 | python -c "import sys,json
d=json.loads(sys.stdin.read()); c=d['config']; print('contexts=$n value', d['value'], 'streaming', c['value_streaming'], 'bam4', c['value_streaming_bam4'])"
done
