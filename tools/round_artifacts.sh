set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/v16
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/v16/pytest_gpu.log 2>&1; tail -3 gpurun_out/v16/pytest_gpu.log
timeout -k 10 300 python bench.py > gpurun_out/v16/bench.json 2> gpurun_out/v16/bench.err; tail -c 600 gpurun_out/v16/bench.json
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/v16/prof -- python3 bench.py --steps 40 --warmup 5 --no-cpu-baseline > gpurun_out/v16/bench_under_rocprof.json 2> gpurun_out/v16/prof.err
echo prof done
bash tools/pmc.sh gpurun_out/v16/pmc > gpurun_out/v16/pmc.log 2>&1; tail -30 gpurun_out/v16/pmc.log
timeout -k 10 600 python tools/bench_ops.py all > gpurun_out/v16/bench_ops.json 2> gpurun_out/v16/bench_ops.err; wc -l gpurun_out/v16/bench_ops.json
timeout -k 10 200 python tools/moe_stage_sweep.py 64 128 256 512 768 1024 1536 2048 3929 4096 8192 16384 2> /dev/null > gpurun_out/v16/stage_sweep.json; wc -l gpurun_out/v16/stage_sweep.json
