for n in 6 8 10; do
  python bench.py --config 5 --steps 40 --warmup 3 --no-streaming --no-cpu-baseline --contexts $n --ws-limit-gb 12 2>&1 | tail -1 | python -c "import sys,json
try:
  d=json.loads(sys.stdin.read()); print('contexts=$n cfg5 value', d['value'])
except Exception as e: print('contexts=$n failed', e)"
done
for n in 6 8; do
  python bench.py --config 3 --steps 40 --warmup 3 --no-streaming --no-cpu-baseline --contexts $n 2>&1 | tail -1 | python -c "import sys,json
try:
  d=json.loads(sys.stdin.read()); print('contexts=$n cfg3 value', d['value'])
except Exception as e: print('contexts=$n failed', e)"
done
