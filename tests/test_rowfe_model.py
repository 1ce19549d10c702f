"""The one-limb-per-lane field and point arithmetic of the Horner tail (zkp_amd/csrc/rowfe.h) as a lane-level model over Python integers
(tools/model/rowfe_model.py: the DPP moves, the column sums, the two carry passes, instruction for instruction): values against big-integer
arithmetic, every intermediate against its register width, outputs inside the "tight" limb class -- the CPU-side half of that file's evidence
(the GPU half: tests/test_gpu_parity.py::test_row_cooperative_point_ops and every MSM parity test, whose last kernel is this chain)."""
import os
import random
import sys

import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "model"))
import rowfe_model as R  # noqa: E402


def test_model_matches_integers_and_register_widths():
    assert R.self_check(rounds=120, seed=5)


def test_dpp_moves_are_the_documented_ones():
    v = list(range(100, 164))
    assert R.shr(v, 2)[16 + 5] == v[16 + 3] and R.shr(v, 2)[16 + 1] == 0               # row_shr: zero fill below the row's lane 0
    assert R.shl(v, 7)[32 + 1] == v[32 + 8] and R.shl(v, 7)[32 + 9] == 0               # row_shl: zero fill past the row's lane 15
    assert R.bcast(v, 4)[48 + 11] == v[48 + 4]
    assert R.pull(v, [2, 3, 3, 1])[16 + 7] == v[48 + 7]


def test_multiplication_overflow_is_detected_by_the_model():
    """the width checks are live: operands one class too large must trip them (so that passing means something)"""
    big = [2**32 - 1] * 9
    a = R.rows_of(None, [big] * 4)
    with pytest.raises(R.Overflow):
        R.row_mul(a, a)


def test_identity_and_doubling_through_the_unified_addition():
    rng = random.Random(8)
    ident = (0, 1, 1, 0)
    for _ in range(3):
        p = R.random_point(rng)
        cached = lambda q: R.rows_of([(q[1] - q[0]) % R.P, (q[1] + q[0]) % R.P, 2 * q[2] % R.P, R.D2 * q[3] % R.P])
        for lhs, rhs in ((p, ident), (ident, p), (p, p), (p, (-p[0] % R.P, p[1], p[2], -p[3] % R.P))):
            got = R.row_add_cached(R.rows_of(lhs), cached(rhs))
            vals = tuple(R.value_of(got, r) for r in range(4))
            assert R.same_point(vals, R.ext_add(lhs, rhs))


def test_lane_swaps_of_the_model_are_what_the_hardware_probe_printed():
    """profiles/r05_permlane_probe.txt is the output of tools/microbench/permlane_probe.hip on an MI355X: the model's v_permlane32_swap / v_permlane16_swap
    must give the same source row for every output row (rowfe.h's movement between coordinates rests on exactly these six lines, plus the raw swap16)."""
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    want = {}
    for line in open(os.path.join(root, "profiles", "r05_permlane_probe.txt")):
        m = re.match(r"(\S+)\s+rows from: (\d) (\d) (\d) (\d)\s+\(lane order inside rows kept\)", line)
        assert m, line
        want[m.group(1)] = [int(g) for g in m.groups()[1:]]
    x = list(range(64))
    rows = lambda v: [v[16 * r] >> 4 for r in range(4)]
    h0, h1 = R.swap32(x, x)
    got = {"swap32(x,x)[0]": rows(h0), "swap32(x,x)[1]": rows(h1)}
    a, b = R.swap16(h0, h0)
    got["swap16(h0,h0)[0]"], got["swap16(h0,h0)[1]"] = rows(a), rows(b)
    a, b = R.swap16(h1, h1)
    got["swap16(h1,h1)[0]"], got["swap16(h1,h1)[1]"] = rows(a), rows(b)
    a, b = R.swap16(x, x)
    got["swap16(x,x)[0]"], got["swap16(x,x)[1]"] = rows(a), rows(b)
    assert got == want
    for v in (h0, h1, a, b):
        assert all(v[16 * r + k] == v[16 * r] + k for r in range(4) for k in range(16))
