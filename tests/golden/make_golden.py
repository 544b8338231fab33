"""Generates tests/golden/oracle_regression.npz: small seeded scenes advanced by the C oracle (fp64).

These are REGRESSION fixtures of this repo's own oracle (the reference cannot be executed here, see
oracle/mpm_oracle.h), committed so that an accidental change of the oracle's arithmetic is caught
on CPU, and so that the GPU box can check the HIP path against stored vectors too.
    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))

from oracle.orc import Oracle  # noqa: E402
from golden_cases import CASES  # noqa: E402


def main():
    out = {}
    for name, (make, k) in CASES.items():
        sc = make()
        ps = sc["particles"]
        st = Oracle(ps.dim, np.float64).new_state(ps, sc["params"], sc["colliders"], sc["cell_width"],
                                                  sc["grid_capacity"], sc.get("model", 0))
        st.step(k)
        cells, mv, dist, aff, closest = st.grid_records()
        out[f"{name}/pos"] = st.arr["pos"].astype(np.float64)
        out[f"{name}/vel"] = st.arr["vel"].astype(np.float64)
        out[f"{name}/def_grad"] = st.arr["def_grad"].astype(np.float64)
        out[f"{name}/affine"] = st.arr["affine"].astype(np.float64)
        out[f"{name}/cdf_affinity"] = st.arr["cdf_affinity"]
        out[f"{name}/grid_cells"] = cells
        out[f"{name}/grid_vm"] = np.asarray(mv, np.float64)
        out[f"{name}/grid_aff"] = aff
        if sc["colliders"]:
            bodies = st.collider_states()
            for key in ("rotation", "translation", "linvel", "angvel"):
                out[f"{name}/body_{key}"] = np.stack([b[key] for b in bodies])
    np.savez_compressed(os.path.join(HERE, "oracle_regression.npz"), **out)
    print("wrote", len(out), "arrays")


if __name__ == "__main__":
    main()
