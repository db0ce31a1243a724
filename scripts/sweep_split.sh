run() { python bench.py --size $1 --steps 3 --warmup 1 --no-cpu-baseline --no-check $3 $4 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1 $2', d['ms_per_step'], d['value'])"; }
run 32768 bs512 --blocksize 512
run 32768 bs1024 --blocksize 1024
run 32768 bs2048 --blocksize 2048
RFLU_SPLIT_SCALE=0.75 run 32768 scale.75
RFLU_SPLIT_SCALE=1.3 run 32768 scale1.3
RFLU_MAX_RESERVE=96 run 65536 maxres96
run 65536 bs1024 --blocksize 1024
RFLU_MAX_RESERVE=96 run 65536 maxres96-bs1024 --blocksize 1024
