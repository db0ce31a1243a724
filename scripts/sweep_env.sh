#!/bin/bash
# bench.py at size $1 for every value of env var $2 in the remaining arguments: prints ms_per_step and TFLOP/s
SIZE=$1; VAR=$2; shift 2
for v in "$@"; do
  export $VAR=$v
  python bench.py --size $SIZE --steps 4 --warmup 2 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$VAR=$v', d['config']['n'], d['ms_per_step'], d['value'], d.get('check'))"
done
