# same-box A/B of one environment knob: tools/ab_env.sh NAME VALUE   (baseline = NAME unset)
for rep in 1 2 3; do
env -u $1 timeout -k 10 100 python bench.py --no-cpu-baseline | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1 unset', d['value'], d['ms_per_step'], d['stage_ms'])"
env $1=$2 timeout -k 10 100 python bench.py --no-cpu-baseline | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1=$2', d['value'], d['ms_per_step'], d['stage_ms'])"
done
