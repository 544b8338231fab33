"""-m "not gpu": the one place where lanes of a wave communicate through LDS with plain (non-atomic) accesses — P2G's
per-wave tile read-add-write, nine phases in which every lane owns a distinct node (csrc/p2g_body.inc) — relies on the
compiler keeping each phase's store ahead of the next phase's load. The source pins that with compiler barriers; this
test looks at what hipcc actually emitted for gfx950: between two phase markers there must be the
phase's LDS loads FIRST and its stores AFTER them, nothing interleaved, in every P2G kernel of both dimensions."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "wgsparkl_amd", "csrc")
HIPCC = "/opt/rocm/bin/hipcc"


@pytest.mark.parametrize("dim", [3, 2])
def test_p2g_tile_phases_keep_their_order_in_the_isa(dim, tmp_path):
    if not os.path.exists(HIPCC):
        pytest.skip("no hipcc")
    out = tmp_path / f"capi{dim}.s"
    subprocess.run([HIPCC, "-O3", "-std=c++17", "--offload-arch=gfx950", "-fno-fast-math", "-ffp-contract=on", f"-DWGS_DIM={dim}",
                    "-S", "--cuda-device-only", os.path.join(CSRC, "capi.hip"), "-o", str(out)], check=True, capture_output=True)
    text = out.read_text()
    kernels = re.split(r"^(_ZN3wgs\S+):.*$", text, flags=re.M)
    checked = 0
    for name, body in zip(kernels[1::2], kernels[2::2]):
        if "k_p2g" not in name or "k_p2g_cdf" in name:
            continue
        body = body.split(".end_amdhsa_kernel")[0] if ".end_amdhsa_kernel" in body else body
        parts = body.split("; WGS_TILE_PHASE_END")
        # one region of 9 phases per included body (the paired launch includes the body twice): 10 markers each
        assert (len(parts) - 1) % 10 == 0 and len(parts) > 1, (name, len(parts))
        for r in range((len(parts) - 1) // 10):
            for k in range(9):
                ops = [ln.split()[0] for ln in parts[1 + r * 10 + k].splitlines() if ln.strip().startswith("ds_")]
                reads = [i for i, o in enumerate(ops) if o.startswith("ds_read") or o.startswith("ds_load")]
                writes = [i for i, o in enumerate(ops) if o.startswith("ds_write") or o.startswith("ds_store")]
                assert reads and writes, (name, r, k, ops)
                assert max(reads) < min(writes), f"{name}: phase {k} of region {r}: an LDS load follows a store of the same phase: {ops}"
        checked += 1
    assert checked >= 3, checked
