"""zkp_amd/csrc/stmt_pairs.h on the CPU: which terms of the verifier's constraints  commitment = sum s_i P_i - c LHS  (verifier.rs:95-106) share a chain of
doublings (ZKP_OPT_JOINT_LADDER).  The rule is host code of the plan; the GPU tests check the bytes it leads to (tests/test_gpu_fused.py)."""
import ctypes

import numpy as np
import pytest

from tests.test_host_transcript_prog import _lib

UNPAIRED, ABSORBED = 0xFFFFFFFF, 0x80000000


def _pairs(cons, ns, np_):
    """cons: per constraint the point ids of its terms, in the plan's order (right-hand side terms, then the left-hand side)"""
    lib = _lib()
    toff = np.cumsum([0] + [len(c) for c in cons]).astype(np.uint32)
    tpt = np.array([p for c in cons for p in c], np.uint32)
    out = np.full(len(tpt), 0xDEADBEEF, np.uint32)
    lib.t_pair_terms.restype = ctypes.c_int
    n = lib.t_pair_terms(toff.ctypes.data_as(ctypes.c_void_p), tpt.ctypes.data_as(ctypes.c_void_p), ctypes.c_uint32(len(tpt)), ctypes.c_uint32(len(cons)),
                         ctypes.c_uint32(ns), ctypes.c_uint32(np_), out.ctypes.data_as(ctypes.c_void_p))
    return n, toff, tpt, (out if n else None)


def test_cmz_verifier_terms_pair_as_documented():
    # common ids 0..11 = X_1..X_10, A, B; per-proof ids 12..21 = C_1..C_10, 22 = P, 23 = Q, 24 = V (cred_show_10, benches/zkp.rs:27-46)
    cons = [[22, 10, 12 + i] for i in range(10)] + [list(range(10)) + [23, 24]]
    n, toff, tpt, pair = _pairs(cons, 12, 25)
    assert n == 11                                   # P's ten terms and one of {Q, V} ride; 11 chains per proof instead of 12 ladders + a table + ten walks
    for i in range(10):
        p_term, lhs = 3 * i, 3 * i + 2
        assert pair[lhs] == p_term and pair[p_term] == (ABSORBED | lhs) and pair[3 * i + 1] == UNPAIRED      # (A is common: fixed-base, never paired)
    q_term, v_term = 40, 41
    assert pair[q_term] == v_term and pair[v_term] == (ABSORBED | q_term)
    assert (pair[30:40] == UNPAIRED).all()


def test_statements_without_per_proof_partners_do_not_pair():
    # DLEQ with common G, H (benches/dleq.rs): A = x G, B = x H -> terms (G, A), (H, B): the left-hand sides have nobody to take along
    n, *_ = _pairs([[0, 2], [1, 3]], 2, 4)
    assert n == 0
    # ... with a per-proof H (benches/zkp.rs:49): B's chain takes H along (or H's takes B), A stays alone
    n, toff, tpt, pair = _pairs([[0, 1], [3, 2]], 1, 4)
    assert n == 1 and pair[0] == UNPAIRED and pair[1] == UNPAIRED and pair[2] == 3 and pair[3] == (ABSORBED | 2)


@pytest.mark.parametrize("seed", range(40))
def test_pairing_invariants_on_random_statements(seed):
    rng = np.random.default_rng(seed)
    ns, ni = int(rng.integers(0, 5)), int(rng.integers(1, 9))
    np_ = ns + ni
    cons = [[int(rng.integers(0, np_)) for _ in range(int(rng.integers(1, 7)))] for _ in range(int(rng.integers(1, 7)))]
    n, toff, tpt, pair = _pairs(cons, ns, np_)
    uses = np.bincount(tpt, minlength=np_)
    cons_of = np.repeat(np.arange(len(cons)), [len(c) for c in cons])
    if not n:
        # nothing pairs only if no constraint holds a single-use per-proof term next to another per-proof term
        for k, c in enumerate(cons):
            singles = [q for q in range(toff[k], toff[k + 1]) if tpt[q] >= ns and uses[tpt[q]] == 1]
            others = [q for q in range(toff[k], toff[k + 1]) if tpt[q] >= ns]
            assert not singles or len(others) < 2
        return
    hosts = [k for k in range(len(tpt)) if pair[k] != UNPAIRED and not pair[k] & ABSORBED]
    riders = [k for k in range(len(tpt)) if pair[k] != UNPAIRED and pair[k] & ABSORBED]
    assert len(hosts) == len(riders) == n
    for h in hosts:
        r = int(pair[h])
        assert pair[r] == (ABSORBED | h) and r != h
        assert cons_of[h] == cons_of[r]                                  # same constraint: the sum of the MSM is unchanged
        assert tpt[h] >= ns and uses[tpt[h]] == 1                        # the host is a ladder of its own anyway
        assert tpt[r] >= ns                                              # common points keep their (fixed-base / shared) tables
    assert len(set(int(pair[h]) for h in hosts)) == n                    # nobody rides twice
    # maximal per constraint: no unpaired single-use per-proof term is left next to an unpaired per-proof term
    for k in range(len(cons)):
        free = [q for q in range(toff[k], toff[k + 1]) if pair[q] == UNPAIRED and tpt[q] >= ns]
        assert not (any(uses[tpt[q]] == 1 for q in free) and len(free) >= 2)
