# Developer utility: A/B of environment switches on one box.  usage: ENVS="A=1 B=2|A=0" bash tools/gpu_env_ab.sh
cd $GRAFT_REPO_ROOT
IFS='|' read -ra SETS <<< "${ENVS:-|}"
for rep in 1 2; do
for s in "${SETS[@]}" ""; do
  echo "== [$s]"; env $s timeout 200 python3 bench.py --steps 50 --warmup 10 --no-cpu-baseline $BENCHARGS 2>&1 | grep -o '"value": [0-9.]*\|"g2p": [0-9.]*\|"p2g": [0-9.]*\|"grid sort": [0-9.]*\|"ms_per_step": [0-9.]*' | tr '\n' ' '; echo
done; done
