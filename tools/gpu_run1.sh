cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02a
export WGS_MARGINS_FILE=$GRAFT_REPO_ROOT/gpurun_out/r02a/margins.jsonl
rm -f $WGS_MARGINS_FILE
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r02a/pytest.log
cat gpurun_out/r02a/pytest.log
timeout 900 python bench.py > gpurun_out/r02a/bench.json 2> gpurun_out/r02a/bench.err; echo "bench rc=$?"; tail -c 600 gpurun_out/r02a/bench.err; head -c 3000 gpurun_out/r02a/bench.json
