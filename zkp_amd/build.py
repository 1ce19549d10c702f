"""Builds the in-tree native libraries (no JIT cache: the .so files travel with the repo snapshot).

    python -m zkp_amd.build            # HIP library for gfx950 (+ host toolbox once present)
"""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
HIP_LIB = os.path.join(HERE, "libzkp_mi355x.so")
HIP_TESTHOOKS_LIB = os.path.join(HERE, "libzkp_mi355x_testhooks.so")   # the same sources with -DZKP_BUILD_TEST_HOOKS (tests / A-B tools only)
HOST_LIB = os.path.join(HERE, "libzkp_toolbox.so")


def _stale(out: str, deps) -> bool:
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    return any(os.path.getmtime(d) > t for d in deps)


def _deps(*dirs, exts=(".h", ".hip", ".cpp", ".c")):
    out = []
    for d in dirs:
        for root, _, files in os.walk(d):
            out += [os.path.join(root, f) for f in files if f.endswith(exts)]
    return out


def _hip_cmd(out: str, test_hooks: bool):
    return ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-Wno-unused-value"] + (["-DZKP_BUILD_TEST_HOOKS"] if test_hooks else []) + [
        os.path.join(CSRC, "zkp_kernels.hip"), "-o", out]


def build_hip(force: bool = False, verbose: bool = False) -> str:
    """hipcc --offload-arch=gfx950: kernels + C ABI -> zkp_amd/libzkp_mi355x.so (cross-compiles without a GPU), and the
    test-hook build of the same sources -> libzkp_mi355x_testhooks.so (the shipped library carries no measurement hooks).
    The two compilations run side by side."""
    force = force or bool(os.environ.get("ZKP_FORCE_BUILD"))
    deps = _deps(CSRC, os.path.join(HERE, "..", "include"))
    procs = []
    for out, hooks in ((HIP_LIB, False), (HIP_TESTHOOKS_LIB, True)):
        if force or _stale(out, deps):
            cmd = _hip_cmd(out, hooks)
            if verbose:
                print(" ".join(cmd), file=sys.stderr)
            procs.append((cmd, subprocess.Popen(cmd)))
        elif verbose:
            print("up to date (mtime): %s  [ZKP_FORCE_BUILD=1 rebuilds]" % out, file=sys.stderr)
    for cmd, pr in procs:
        if pr.wait() != 0:
            raise subprocess.CalledProcessError(pr.returncode, cmd)
    return HIP_LIB


def build_host(force: bool = False, verbose: bool = False):
    """g++: host-side toolbox (Merlin, Prover / Verifier / BatchVerifier over the C ABI) -> libzkp_toolbox.so"""
    host_dir = os.path.join(CSRC, "host")
    if not os.path.isdir(host_dir):
        return None
    srcs = [os.path.join(host_dir, f) for f in sorted(os.listdir(host_dir)) if f.endswith(".cpp")]
    if not srcs:
        return None
    # (host_backend.cpp compiles the kernels' point formulas for the host: ge25519.h and what it includes)
    deps = _deps(host_dir, os.path.join(HERE, "..", "include"), exts=(".h", ".hpp", ".cpp", ".map")) + [os.path.join(CSRC, f) for f in ("ge25519.h", "fe25519.h", "fe_constants.h")]
    force = force or bool(os.environ.get("ZKP_FORCE_BUILD"))
    if force or _stale(HOST_LIB, deps):
        cmd = ["g++", "-O3", "-std=c++17", "-shared", "-fPIC", "-pthread", "-Wall", "-Wno-unknown-pragmas", "-I", os.path.join(HERE, "..", "include")] + srcs + [
            "-o", HOST_LIB, "-L", HERE, "-lzkp_mi355x", "-Wl,-rpath,$ORIGIN", "-Wl,--version-script=" + os.path.join(host_dir, "exports.map")]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.check_call(cmd)
    return HOST_LIB


def build_all(force: bool = False, verbose: bool = False):
    """Rebuilds what is stale by mtime; force = True or ZKP_FORCE_BUILD=1 in the environment rebuilds everything from source."""
    mode = "forced rebuild from source" if (force or os.environ.get("ZKP_FORCE_BUILD")) else "incremental (mtime)"
    if verbose:
        print("zkp_amd.build: %s" % mode, file=sys.stderr)
    return build_hip(force, verbose), build_host(force, verbose)


if __name__ == "__main__":
    print(build_all(force="--force" in sys.argv, verbose=True))
