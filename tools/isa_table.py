"""Developer utility (CPU): VGPRs / scratch / LDS / occupancy per kernel from a `hipcc -S --cuda-device-only` dump, demangled.
usage: python tools/isa_table.py /tmp/capi3.s [substring ...]"""
import re, subprocess, sys
path = sys.argv[1]
flt = sys.argv[2:]
cur, st = None, {}
for ln in open(path):
    m = re.match(r"^(_ZN3wgs\S+):", ln)
    if m:
        cur = m.group(1); st[cur] = dict(n=0, vgpr=None, scratch=None, lds=None, occ=None); continue
    if cur is None: continue
    t = ln.strip()
    if t.startswith("; NumVgprs:"): st[cur]["vgpr"] = t.split()[-1]
    elif t.startswith("; ScratchSize:"): st[cur]["scratch"] = t.split()[-1]
    elif t.startswith("; LDSByteSize:"): st[cur]["lds"] = t.split()[2]
    elif t.startswith("; Occupancy:"): st[cur]["occ"] = t.split()[-1]
    elif re.match(r"^(v_|s_|ds_|global_|buffer_|flat_|scratch_)", t): st[cur]["n"] += 1
names = list(st)
dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
for k, dn in zip(names, dem):
    dn = dn.replace("wgs::", "").split("(")[0]
    if not flt or any(f in dn for f in flt):
        v = st[k]
        print(f"{dn[:64]:64s} n={v['n']:5d} vgpr={v['vgpr']:>3s} scratch={v['scratch']:>4s} lds={v['lds']:>6s} occ={v['occ']}")
