# Round artifact set: full GPU test suite, default bench, rocprofv3 kernel stats + PMC passes of the same bench command,
# secondary benches.  usage (on the GPU box): bash tools/round_artifacts.sh <tag>   -> gpurun_out/<tag>/
set -e
TAG=${1:-r02_v1}
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$TAG
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/$TAG/pytest_gpu.log 2>&1; tail -3 gpurun_out/$TAG/pytest_gpu.log
timeout -k 10 400 python bench.py > gpurun_out/$TAG/bench.json 2> gpurun_out/$TAG/bench.err; tail -c 900 gpurun_out/$TAG/bench.json
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$TAG/prof -- python3 bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-a8 --no-int8 --no-verify > gpurun_out/$TAG/bench_under_rocprof.json 2> gpurun_out/$TAG/prof.err
python tools/trace_step.py gpurun_out/$TAG/prof > gpurun_out/$TAG/step_timeline.txt 2>&1 || true
find gpurun_out/$TAG/prof -name "*kernel_trace.csv" -delete
echo prof done
bash tools/pmc.sh gpurun_out/$TAG/pmc > gpurun_out/$TAG/pmc.log 2>&1; tail -30 gpurun_out/$TAG/pmc.log
find gpurun_out/$TAG/pmc -name "*.csv" -size +1M -delete
timeout -k 10 700 python tools/bench_ops.py all > gpurun_out/$TAG/bench_ops.json 2> gpurun_out/$TAG/bench_ops.err; wc -l gpurun_out/$TAG/bench_ops.json
timeout -k 10 200 python tools/moe_stage_sweep.py 64 128 256 512 768 1024 1536 2048 3929 4096 8192 16384 2> /dev/null > gpurun_out/$TAG/stage_sweep.json; wc -l gpurun_out/$TAG/stage_sweep.json
