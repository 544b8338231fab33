# full GPU suite without -x, margins collected (developer utility)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
D=gpurun_out/${TAG:-tests}
mkdir -p $D
export WGS_MARGINS_FILE=$GRAFT_REPO_ROOT/$D/margins.jsonl
rm -f $WGS_MARGINS_FILE
timeout ${TMO:-2400} python -m pytest tests -m gpu -q ${PYARGS:-} 2>&1 | tail -${TAIL:-40} > $D/pytest.log
cat $D/pytest.log
