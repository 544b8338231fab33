# developer utility: round 3's tree (tools/tmp_r3, exported by `git archive 57d895f`, its own library built in place) against the
# current tree on ONE box, alternating, C3 pass times (VERDICT r4 task 1a: where did 313 -> 346 us come from)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
D=$GRAFT_REPO_ROOT/gpurun_out/ab_r3; mkdir -p $D
show() { python3 - "$1" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().split("\n")[-1])
    print(sys.argv[1].split("/")[-1], round(d["ms_per_step"]*1e3,1), {a: round(b*1e3,1) for a,b in d["pass_ms_per_step"].items() if b > 0.006})
except Exception as e: print(sys.argv[1], "no json", e)
PY
}
for i in 1 2; do
  for cfg in ${CFGS:-c3}; do
    (cd tools/tmp_r3 && timeout 300 python3 bench.py --config $cfg --steps 50 --warmup 10 --no-cpu-baseline --no-extra > $D/r3_${cfg}_$i.json 2> $D/r3_${cfg}_$i.err); show $D/r3_${cfg}_$i.json
    timeout 300 python3 bench.py --config $cfg --steps 50 --warmup 10 --no-cpu-baseline --no-extra --no-live-pmc > $D/head_${cfg}_$i.json 2> $D/head_${cfg}_$i.err; show $D/head_${cfg}_$i.json
  done
done
