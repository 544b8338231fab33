"""Pins the CPU oracle against every golden vector / known answer the reference holds for the
hot path: the exclusive prefix sum of the reference's only numeric test
(src/grid/prefix_sum.rs:170-231), the neighbourhood tables of src/grid/kernel.wgsl, and the
hash / pack_key known answers. Everything else of the oracle is "parity unpinned" (see
oracle/mpm_oracle.h) and is covered by test_oracle_twin.py and test_oracle_invariants.py."""
import json
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
TABLES = json.load(open(os.path.join(HERE, "golden", "reference_tables.json")))


@pytest.fixture(scope="module")
def orc3(oracle_libs):
    return oracle_libs.Oracle(3, np.float32)


@pytest.fixture(scope="module")
def orc2(oracle_libs):
    return oracle_libs.Oracle(2, np.float32)


def test_prefix_sum_reference_vectors(orc3):
    """gpu_prefix_sum (prefix_sum.rs:183-229): all-ones, iota, random % 10_000 at LEN = 15071."""
    n = TABLES["prefix_sum_len"]
    ones = np.ones(n, np.uint32)
    iota = np.arange(n, dtype=np.uint32)
    rnd = (np.random.default_rng(0).integers(0, 2**32, n, dtype=np.uint64) % 10_000).astype(np.uint32)
    out = orc3.prefix_sum_eval_cpu(ones)
    assert np.array_equal(out, np.arange(n, dtype=np.uint32))            # out[i] = i
    out = orc3.prefix_sum_eval_cpu(iota)
    i = np.arange(n, dtype=np.uint64)
    assert np.array_equal(out, ((i * (i - 1)) // 2).astype(np.uint32))   # out[i] = i (i - 1) / 2
    for v in (ones, iota, rnd):
        cpu = orc3.prefix_sum_eval_cpu(v)
        # the restated GPU algorithm (prefix_sum.wgsl:11-93) must equal eval_cpu, as the reference asserts
        assert np.array_equal(orc3.prefix_sum_gpu_algorithm(v), cpu)
        assert np.array_equal(cpu, np.concatenate([[0], np.cumsum(v[:-1], dtype=np.uint64)]).astype(np.uint32))


@pytest.mark.parametrize("n", [0, 1, 2, 255, 256, 257, 65536, 65537])
def test_prefix_sum_edges(orc3, n):
    v = (np.arange(n, dtype=np.uint32) * 7 + 3) % 11
    cpu = orc3.prefix_sum_eval_cpu(v)
    assert np.array_equal(orc3.prefix_sum_gpu_algorithm(v), cpu)


def test_hash_and_pack_known_answers(orc3):
    for ka in TABLES["hash_known_answers_3d"]:
        assert orc3.hash(orc3.pack_key(ka["block"])) == int(ka["hash"], 16)
    assert orc3.pack_key(TABLES["pack_none_corner_3d"]) == 0xFFFFFFFF   # quirk B5
    # pack_key layout (grid.wgsl:88-95): x 11 bits, y 10 bits << 11, z 11 bits << 21
    assert orc3.pack_key([0, 0, 0]) == 0x3FF | (0x1FF << 11) | (0x3FF << 21)
    assert orc3.pack_key([-1023, -511, -1023]) == 0


def test_neighbourhood_tables(orc3, orc2):
    for orc, dim in ((orc3, 3), (orc2, 2)):
        shifts = TABLES[f"nbh_shifts_{dim}d"]
        shared = TABLES[f"nbh_shifts_shared_{dim}d"]
        for i, (s, sh) in enumerate(zip(shifts, shared)):
            assert [orc.lib.orc_nbh_shift(i, a) for a in range(dim)] == s
            assert orc.lib.orc_nbh_shift_shared(i) == sh
        # each shift of {0,1,2}^D exactly once
        assert sorted(map(tuple, shifts)) == sorted(np.ndindex(*([3] * dim)))


def test_associated_cell_rule(orc3):
    """round(x/h) - 1 with ties to even and a true division (particle3d.wgsl:41-49)."""
    h = 1.0
    xs = np.array([0.0, 0.49999997, 0.5, 0.50000006, 1.5, 2.5, -0.5, -1.5, 3.4999998, 1e-30], np.float32)
    want = np.array([-1, -1, -1, 0, 1, 1, -1, -3, 2, -1])    # rint: 0.5->0, 1.5->2, 2.5->2, -0.5->-0, -1.5->-2
    got = orc3.assoc_cells(np.stack([xs, xs, xs], 1), h)[:, 0]
    assert np.array_equal(got, want)
    b, l = orc3.block_and_local([-0.6, 4.6, 17.2], 1.0)       # cells -2, 4, 16
    assert b.tolist() == [-1, 1, 4] and l.tolist() == [2, 0, 0]
    # true division, not multiplication by the reciprocal
    hh = np.float32(0.1)
    x = np.float32(0.25)                                       # 0.25/0.1 = 2.5 (ties) in fp32 division
    assert orc3.assoc_cells(np.array([[x, x, x]]), float(hh))[0, 0] == int(np.rint(x / hh)) - 1


def test_kernel_weights(orc3):
    """eval_all (kernel.wgsl:60-66): partition of unity and zero first moment on [0.5, 1.5]."""
    o64 = type(orc3)(3, np.float64)
    for x in np.linspace(0.5, 1.5, 41):
        w = o64.eval_all(x)
        assert abs(w.sum() - 1.0) < 1e-14
        assert abs(np.dot(w, np.arange(3) - x)) < 1e-14   # sum w_i (i - x) = 0
        assert (w >= 0).all()
