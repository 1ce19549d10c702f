"""Transcript programs (zkp_amd/csrc/merlin_prog.h): the compiler that turns a statement's Merlin/STROBE operation
sequence into per-block word operations, and the interpreter the GPU runs per lane, are compiled for the host and
checked byte for byte against the host Merlin (itself pinned to the merlin crate's KAT in test_host_toolbox.py):
final STROBE states, trailing position bytes, every PRF output (rng fills, challenge), identity-rejection flags."""
import ctypes
import os
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "..", "zkp_amd", "csrc")


_LIB = []


def _lib():
    """the host build of the headers under test (tests/host/tr_host_lib.cpp), rebuilt when a source is newer"""
    if not _LIB:
        out = os.path.join(HERE, "host", "tr_host_lib.so")
        srcs = [os.path.join(HERE, "host", "tr_host_lib.cpp"), os.path.join(CSRC, "host", "merlin.cpp")]
        deps = srcs + [os.path.join(CSRC, "merlin_prog.h"), os.path.join(CSRC, "host", "merlin.hpp"), os.path.join(CSRC, "stmt_pairs.h")]
        if not os.path.exists(out) or any(os.path.getmtime(d) > os.path.getmtime(out) for d in deps):
            subprocess.check_call(["g++", "-O1", "-std=c++17", "-shared", "-fPIC"] + srcs + ["-o", out])
        _LIB.append(ctypes.CDLL(out))
    return _LIB[0]


@pytest.fixture(scope="module")
def lib():
    return _lib()


@pytest.mark.parametrize("args", [
    (1, 3, 2, 4, 2, 99),        # small statement
    (2, 5, 21, 25, 11, 2),      # CMZ shape: 21 secrets, 25 points, 11 constraints; proof 2 carries an identity encoding
    (3, 2, 0, 1, 1, 0),         # no secrets
    (4, 4, 1, 0, 0, 9),         # no points, no constraints
    (5, 3, 64, 65, 1, 1),       # W64 shape
    (6, 7, 3, 3, 2, 6),         # DLEQ-like
] + [(100 + s, 2, 1 + s % 5, 1 + s % 7, 1 + s % 3, s % 3) for s in range(24)]     # positions sweep the 166-byte rate
  + [(1000 + s, 3, 2 + s, 3 + s, 2, 1) for s in range(4)])                        # labels of several STROBE blocks
def test_compiled_program_equals_host_merlin(lib, args):
    rc = lib.t_tr_selftest(*args)
    assert rc > 0, f"mismatch bits {-rc}"


def test_step_form_merges_only_what_commutes(lib):
    """merlin_prog.h: tr_steps_build folds a step without a permutation into its successor unless the successor clears bytes the first one wrote --
    a hand-built pair of each kind through the operation list and through the step form."""
    assert lib.t_tr_merge_selftest() == 1
