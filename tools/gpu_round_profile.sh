# Produces the rocprofv3 evidence committed under profiles/ (run on the GPU box through gpurun).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
R=${ROUND:-r02}
rm -rf gpurun_out/$R; mkdir -p gpurun_out/$R
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$R/stats -- python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-extra > gpurun_out/$R/bench_under_rocprof.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d gpurun_out/$R/pmc_$c -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra > gpurun_out/$R/pmc_$c.log 2>&1
done
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT --output-format csv -d gpurun_out/$R/pmc_SQ -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra > gpurun_out/$R/pmc_SQ.log 2>&1
# the default bench three times (G2P varies by a few us from process to process on one box): all lines are kept, the
# median by value becomes profiles/${R}_bench.json
: > gpurun_out/$R/bench.json
for i in 1 2 3; do timeout 600 python3 bench.py >> gpurun_out/$R/bench.json 2>> gpurun_out/$R/bench.err; done
tail -1 gpurun_out/$R/bench.json | cut -c1-300
