"""Instruction mix per kernel from a hipcc -S dump (developer utility, not a test)."""
import re, sys
path = sys.argv[1] if len(sys.argv) > 1 else "/tmp/isa/capi3.s"
lines = open(path).read().split("\n")
cur, stats = None, {}
for ln in lines:
    m = re.match(r"^(_ZN3wgs\S+):", ln)
    if m:
        cur = m.group(1); stats[cur] = dict(n=0, valu=0, pk=0, ds=0, gl=0, salu=0, wait=0, vgpr=None, scratch=None)
        continue
    if cur is None: continue
    t = ln.strip()
    if t.startswith(".vgpr_count:") or t.startswith("; NumVgprs:"):
        stats[cur]["vgpr"] = t.split()[-1]
    if t.startswith("; ScratchSize:"): stats[cur]["scratch"] = t.split()[-1]
    if t.startswith("v_"): stats[cur]["valu"] += 1; stats[cur]["n"] += 1
    if t.startswith("v_pk_"): stats[cur]["pk"] += 1
    if t.startswith("ds_"): stats[cur]["ds"] += 1; stats[cur]["n"] += 1
    if t.startswith("global_") or t.startswith("buffer_") or t.startswith("flat_"): stats[cur]["gl"] += 1; stats[cur]["n"] += 1
    if t.startswith("s_waitcnt"): stats[cur]["wait"] += 1
    if t.startswith("s_"): stats[cur]["salu"] += 1; stats[cur]["n"] += 1
    if t.startswith(".end_amdhsa_kernel") or t.startswith(".Lfunc_end"): pass
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for k, v in stats.items():
    if flt in k:
        print(f"{k[:64]:64s} n={v['n']:5d} valu={v['valu']:5d} pk={v['pk']:4d} ds={v['ds']:4d} mem={v['gl']:4d} salu={v['salu']:5d} wait={v['wait']:4d} vgpr={v['vgpr']} scratch={v['scratch']}")
