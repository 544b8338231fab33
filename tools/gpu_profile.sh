cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
if [ -z "$NOTEST" ]; then timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -3; fi
rm -rf gpurun_out/prof2; mkdir -p gpurun_out/prof2 && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof2 -- python3 bench.py --steps 50 --warmup 10 --no-cpu-baseline $BENCHARGS > gpurun_out/prof2/bench.log 2>&1; grep metric gpurun_out/prof2/bench.log | cut -c1-200
