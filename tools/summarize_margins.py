"""gpurun_out/<tag>/margins.jsonl (written by the -m gpu suite when WGS_MARGINS_FILE is set, tools/gpu_tests_all.sh)
-> profiles/<round>_parity_margins.json: per check name, how many comparisons, the worst measured value, its bound and
the test it came from."""
import collections, json, sys
tag, rnd = (sys.argv[1] if len(sys.argv) > 1 else "tests"), (sys.argv[2] if len(sys.argv) > 2 else "r03")
by = collections.OrderedDict()
total = escapes = 0
for line in open(f"gpurun_out/{tag}/margins.jsonl"):
    r = json.loads(line)
    total += 1
    escapes += 1 if r.get("escape_used") else 0
    e = by.setdefault(r["name"], dict(name=r["name"], count=0, worst_value=-1.0, bound=r["bound"], worst_test=""))
    e["count"] += 1
    if r["value"] / max(r["bound"], 1e-300) > e["worst_value"] / max(e["bound"], 1e-300) or e["count"] == 1:
        e.update(worst_value=r["value"], bound=r["bound"], worst_test=r["test"])
import os, re, subprocess
suite = ""
try:
    suite = [l.strip() for l in open(f"gpurun_out/{tag}/pytest.log") if re.search(r"\d+ passed", l)][-1]
except Exception:
    pass
head = subprocess.run(["git", "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
out = dict(round=rnd, suite=suite, git_head_when_summarised=head, note="measured parity margins of the -m gpu suite on MI355X (tests/helpers.py report_margin): per check name, the worst value "
                "over all tests against its bound; values are relative RMS vs the fp64 oracle unless the name says otherwise",
           comparisons=total, comparisons_that_used_the_affine_escape=escapes, checks=list(by.values()))
json.dump(out, open(f"profiles/{rnd}_parity_margins.json", "w"), indent=1)
print(f"{total} comparisons, {escapes} used the affine escape")
for c in out["checks"]:
    print(f"{c['name']:52s} n={c['count']:4d} worst={c['worst_value']:.3g} bound={c['bound']:.3g} {c['worst_test'].split('::')[-1]}")
