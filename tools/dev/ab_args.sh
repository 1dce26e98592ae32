# A/B of bench.py argument sets on one box: bash tools/dev/ab_args.sh "--event-every 1" "--event-every 4" ...
for a in "$@" "$@"; do
  timeout -k 10 120 python bench.py --no-cpu-baseline --no-parity $a 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); p=d['phase_ms_per_step']; print('$a', round(d['ms_per_step'],3), {k: round(v,2) for k,v in p.items() if k!='sweeps_in_launch_order'})"
done
