# same-box A/B of two builds of the library, interleaved rounds: tools/ab_lib.sh <other libsglk.so> [bench args]
OTHER=$1; shift
for rep in 1 2 3; do
for lib in "" "$OTHER"; do
SGLK_LIB_PATH=$lib timeout -k 10 100 python bench.py --no-cpu-baseline --no-verify --no-a8 "$@" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('lib=${lib:-default}', d['value'], d['ms_per_step'], d['stage_ms'])"
done; done
