cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
D=gpurun_out/r04_f; mkdir -p $D
timeout 2400 python -X faulthandler -m pytest tests -m gpu -q -x --deselect tests/test_multi_gpu.py -v > $D/pytest_full.log 2>&1
grep -n "FAILED\|Fatal\|Error\|passed\|failed\|File \"/tmp" $D/pytest_full.log | tail -30
