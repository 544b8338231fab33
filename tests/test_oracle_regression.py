"""-m "not gpu": the C oracle reproduces its committed regression vectors (tests/golden/oracle_regression.npz,
made by tests/golden/make_golden.py), fp64 bit-for-bit up to libm differences, fp32 within round-off."""
import os

import numpy as np
import pytest

from golden_cases import CASES

GOLD = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "oracle_regression.npz"))


@pytest.mark.parametrize("name", sorted(CASES))
def test_oracle_matches_committed_vectors(oracle_libs, name):
    make, k = CASES[name]
    sc = make()
    ps = sc["particles"]
    st = oracle_libs.Oracle(ps.dim, np.float64).new_state(ps, sc["params"], sc["colliders"], sc["cell_width"],
                                                          sc["grid_capacity"], sc.get("model", 0))
    st.step(k)
    cells, mv, _, aff, _ = st.grid_records()
    assert np.array_equal(cells, GOLD[f"{name}/grid_cells"])
    assert np.array_equal(aff, GOLD[f"{name}/grid_aff"])
    assert np.array_equal(st.arr["cdf_affinity"], GOLD[f"{name}/cdf_affinity"])
    for f in ("pos", "vel", "def_grad", "affine"):
        ref = GOLD[f"{name}/{f}"]
        assert np.allclose(st.arr[f], ref, rtol=1e-9, atol=1e-9 * max(1.0, np.abs(ref).max())), f
    assert np.allclose(mv, GOLD[f"{name}/grid_vm"], rtol=1e-9, atol=1e-9 * np.abs(GOLD[f"{name}/grid_vm"]).max())
    if sc["colliders"]:
        bodies = st.collider_states()
        for key in ("rotation", "translation", "linvel", "angvel"):
            assert np.allclose(np.stack([b[key] for b in bodies]), GOLD[f"{name}/body_{key}"], rtol=0.0, atol=1e-9), key
