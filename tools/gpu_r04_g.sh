cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for i in 1 2; do
for bits in 0 1048576; do
WGS_DEBUG=$bits timeout 200 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-extra --allow-debug-switches --no-floor 2>/dev/null | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); print('nofloor dbg=$bits', round(d['ms_per_step']*1e3,1), {a:round(b*1e3,1) for a,b in d['pass_ms_per_step'].items() if b>0.006})"
done
done
