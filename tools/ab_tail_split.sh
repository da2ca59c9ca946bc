for rep in 1 2; do
for mode in 0 2; do
echo "--- SGLK_TAIL_SPLIT=$mode rep $rep"
SGLK_TAIL_SPLIT=$mode timeout -k 10 100 python tools/bench_ops.py moe 2>/dev/null | grep -E "\"op\": \"fused_experts_fp8\"" | python -c "
import sys,json
print(' '.join(f\"{json.loads(l)['M']}:{json.loads(l)['ms']}\" for l in sys.stdin if json.loads(l)['M']>=3000))"
SGLK_TAIL_SPLIT=$mode timeout -k 10 100 python bench.py --no-cpu-baseline | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bench', d['value'], d['ms_per_step'])"
done; done
