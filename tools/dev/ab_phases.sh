# per-phase / per-sweep A/B of train-step variants selected by environment variables: bash tools/dev/ab_phases.sh "VAR=val ..." ...
run() { env $1 timeout -k 10 120 python bench.py --no-cpu-baseline --no-parity 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); p=d['phase_ms_per_step']; print('$1', round(d['ms_per_step'],3), [round(v,3) for v in p['sweeps_in_launch_order']], {k: round(v,2) for k,v in p.items() if k!='sweeps_in_launch_order'})"; }
run "PGASR_X=0"
for v in "$@"; do run "$v"; done
run "PGASR_X=0"
