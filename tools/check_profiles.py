"""Fails (exit 1) when an evidence file under profiles/ is byte-equal to the same-named file of ANOTHER round: a file called
rNN_x must come from round NN's own runs (round 4 shipped r04_parity_margins.json as a copy of r03's). Run here, on the CPU:
`python tools/check_profiles.py`; tools/gpu_round_evidence.sh's summaries are checked by it before they are committed, and
tests/test_profiles_hygiene.py runs it in the CPU suite."""
import collections, hashlib, os, re, sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "profiles")


def duplicates(root=ROOT):
    by_stem = collections.defaultdict(list)
    for f in sorted(os.listdir(root)):
        m = re.match(r"r(\d+)_(.+)$", f)
        if m:
            by_stem[m.group(2)].append((int(m.group(1)), f))
    bad = []
    for stem, files in by_stem.items():
        seen = {}
        for rnd, f in files:
            h = hashlib.sha256(open(os.path.join(root, f), "rb").read()).hexdigest()
            if h in seen:
                bad.append((seen[h], f))
            seen.setdefault(h, f)
    return bad


if __name__ == "__main__":
    bad = duplicates()
    for a, b in bad:
        print(f"profiles/{b} is a byte copy of profiles/{a}")
    sys.exit(1 if bad else 0)
