# careful A/B of environment settings: 100 timed steps each, alternating, three rounds: bash tools/dev/ab_env_long.sh "A=1" "B=1"
for r in 1 2 3; do for v in "$@"; do
  env $v timeout -k 10 200 python bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-parity 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', round(d['ms_per_step'],3))"
done; done
