"""CPU tests of the product's HOST side (libzkp_toolbox.so: Merlin, scalars mod l, prover phases,
batch-verification coefficient build) against the oracle, and of the C-ABI surface: both libraries must
load and export every symbol include/*.h declares.  No GPU compute is invoked here; where the flow needs
multiscalar multiplications, the TEST substitutes the oracle's (the product never does)."""
import os
import random
import re
import sys

import ctypes
import numpy as np
import pytest

from oracle import cbind as C
from oracle import model as M
from zkp_amd import engine, toolbox as T

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module", autouse=True)
def _built():
    import __graft_entry__ as g
    g.build()


def sc(x):
    return (x % (1 << 256)).to_bytes(32, "little")


def arr(rows, width=32):
    return np.frombuffer(b"".join(rows), np.uint8).reshape(-1, width) if rows else np.zeros((0, width), np.uint8)


def _declared(header, test_hooks=False):
    """function names a header declares; the `#ifdef ZKP_BUILD_TEST_HOOKS` section only when asked for"""
    src = open(os.path.join(ROOT, "include", header)).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    hooks = re.findall(r"#ifdef ZKP_BUILD_TEST_HOOKS(.*?)#endif", src, flags=re.S)
    src = re.sub(r"#ifdef ZKP_BUILD_TEST_HOOKS.*?#endif", "", src, flags=re.S)
    if test_hooks:
        src = "".join(hooks)
    return sorted(set(re.findall(r"\b(zkp_[a-z0-9_]+)\s*\(", src)))


def test_c_abi_exports_every_declared_symbol():
    hip = ctypes.CDLL(engine.LIB_PATH)
    for name in _declared("zkp_mi355x.h"):
        assert hasattr(hip, name), name
    assert set(engine.EXPORTS) == set(_declared("zkp_mi355x.h"))
    # measurement / self-test hooks exist in the test-hook build only: the shipped library does not export them
    assert set(engine.TEST_HOOK_EXPORTS) == set(_declared("zkp_mi355x.h", test_hooks=True)) and engine.TEST_HOOK_EXPORTS
    hooks = ctypes.CDLL(engine.TESTHOOKS_LIB_PATH)
    for name in engine.TEST_HOOK_EXPORTS:
        assert hasattr(hooks, name) and not hasattr(hip, name), name
    for name in engine.EXPORTS:
        assert hasattr(hooks, name), name
    tb = ctypes.CDLL(T.LIB_PATH)
    declared = [n for n in _declared("zkp_toolbox.h") if n not in _declared("zkp_mi355x.h")]
    for name in declared:
        assert hasattr(tb, name), name
    assert set(T.EXPORTS) == set(declared)


def test_shipped_code_object_has_no_scratch_no_spills_and_no_hook_kernels(tmp_path):
    """The gfx950 code object inside libzkp_mi355x.so: no kernel uses scratch (private segment) or spills, and the self-test /
    measurement kernels (k_debug_quad: 396 B of scratch, k_noop) are not in it -- they live in the test-hook build."""
    import shutil
    import subprocess
    llvm = "/opt/rocm/lib/llvm/bin"
    if not all(os.path.exists(os.path.join(llvm, t)) for t in ("llvm-objcopy", "clang-offload-bundler", "llvm-readelf")):
        pytest.skip("ROCm llvm tools not installed")
    notes = {}
    for tag, path in (("shipped", engine.LIB_PATH), ("hooks", engine.TESTHOOKS_LIB_PATH)):
        fat, co = str(tmp_path / (tag + ".fat")), str(tmp_path / (tag + ".co"))
        subprocess.check_call([os.path.join(llvm, "llvm-objcopy"), "-O", "binary", "--only-section=.hip_fatbin", path, fat])
        subprocess.check_call([os.path.join(llvm, "clang-offload-bundler"), "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--input=" + fat,
                               "--output=" + co, "--unbundle"])
        txt = subprocess.check_output([os.path.join(llvm, "llvm-readelf"), "--notes", co], text=True)
        kernels, cur = {}, None
        for line in txt.splitlines():
            m = re.match(r"\s+\.(name|private_segment_fixed_size|sgpr_spill_count|vgpr_spill_count):\s+(\S+)", line)
            if not m:
                continue
            if m.group(1) == "name":
                cur = kernels.setdefault(m.group(2), {})
            elif cur is not None:
                cur[m.group(1)] = int(m.group(2))
        notes[tag] = kernels
    shipped, hooks = notes["shipped"], notes["hooks"]
    assert len(shipped) > 40
    bad = {k: v for k, v in shipped.items() if v.get("private_segment_fixed_size") or v.get("vgpr_spill_count") or v.get("sgpr_spill_count")}
    assert not bad, bad
    assert not any("k_debug_quad" in k or "k_debug_row" in k or "k_noop" in k for k in shipped)
    assert any("k_debug_quad" in k for k in hooks) and any("k_debug_row" in k for k in hooks) and any("k_noop" in k for k in hooks)
    shutil.rmtree(str(tmp_path), ignore_errors=True)


def test_engine_fails_loudly_without_gpu():
    """No CPU fallback: on a box without a GPU the context cannot be created."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(engine.ZkpError):
        engine.Engine(0)


def test_pipe_fails_loudly_without_gpu_and_refuses_bad_arguments():
    """zkp_pipe is a GPU object: without a device it cannot be created (the host backend is chosen with ctx == NULL on the plain calls,
    never behind a pipe); an empty device list or zero contexts is an argument error everywhere."""
    import ctypes
    import torch
    lib = T.lib()
    h = ctypes.c_void_p()
    assert lib.zkp_pipe_create(ctypes.byref(h), None, 0, 1) != 0 and not h.value
    ids = (ctypes.c_int * 1)(0)
    assert lib.zkp_pipe_create(ctypes.byref(h), ids, 1, 0) != 0 and not h.value
    assert lib.zkp_pipe_create(ctypes.byref(h), ids, 1, 2000) != 0 and not h.value        # more than 1024 contexts
    assert lib.zkp_pipe_num_contexts(None) == 0 and lib.zkp_pipe_jobs_in_flight(None) == 0
    lib.zkp_pipe_destroy(None)                                                             # no-op
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(engine.ZkpError):
        T.Pipe((0,), 1)


def test_transcript_kat_and_random_traffic():
    t = T.Transcript(b"test protocol")
    t.append_message(b"some label", b"some data")
    assert t.challenge_bytes(b"challenge", 32).hex() == "d5a21972d0d5fe320c0d263fac7fffb8145aa640af6e9bca177c03c7efcf0615"
    rng = random.Random(5)
    a, b = T.Transcript(b"x"), M.Transcript(b"x")
    for i in range(40):
        n = rng.choice([0, 1, 31, 32, 165, 166, 167, 333])
        msg = bytes(rng.randrange(256) for _ in range(n))
        a.append_message(b"lab%d" % i, msg)
        b.append_message(b"lab%d" % i, msg)
        if i % 5 == 0:
            k = rng.choice([1, 32, 64, 200])
            assert a.challenge_bytes(b"c", k) == b.challenge_bytes(b"c", k)
    c = a.clone()
    assert c.challenge_bytes(b"z", 16) == a.challenge_bytes(b"z", 16)


def test_scalars():
    rng = random.Random(6)
    lib = T.lib()
    out = ctypes.create_string_buffer(32)
    for v in (0, 1, M.L - 1, M.L, M.L + 1, (1 << 256) - 1, 1 << 256, (1 << 256) + M.L, M.L << 256, (M.L << 256) - 1, (1 << 512) - 1,
              ((1 << 512) - 1) // M.L * M.L, ((1 << 512) - 1) // M.L * M.L - 1):                   # from_bytes_mod_order_wide edges
        lib.zkp_scalar_from_wide(out, v.to_bytes(64, "little"))
        assert out.raw == sc(v % M.L), hex(v)
    for _ in range(500):
        w = bytes(rng.randrange(256) for _ in range(64))
        lib.zkp_scalar_from_wide(out, w)
        assert out.raw == sc(int.from_bytes(w, "little") % M.L)
        a, b, c = (rng.choice([0, 1, M.L - 1, M.L, (1 << 256) - 1, rng.randrange(1 << 256)]) for _ in range(3))
        lib.zkp_scalar_muladd(out, sc(a), sc(b), sc(c))
        assert out.raw == sc((a * b + c) % M.L)
        lib.zkp_scalar_neg(out, sc(a))
        assert out.raw == sc((-a) % M.L)


def _instances(mst, rng, n, which):
    from tests.test_oracle_c import _cmz_instance, _dleq_instance
    secs, encs = [], []
    common = None
    for _ in range(n):
        sec, pts = _dleq_instance(rng) if which == "dleq" else _cmz_instance(rng)
        if common is None:
            common = {k: pts[k] for k in mst.common}
        elif which == "cmz":       # common points shared by the batch; recompute the dependent instance points
            pts.update(common)
            for i in range(1, 11):
                pts[f"C_{i}"] = M.pt_add(M.pt_mul(sec[f"m_{i}"], pts["P"]), M.pt_mul(sec[f"z_{i}"], pts["A"]))
            pts["V"] = M.msm_points([sec[f"m_{i}"] for i in range(1, 11)] + [sec["minus_z_Q"]], [pts[f"X_{i}"] for i in range(1, 11)] + [pts["Q"]])
        secs.append(sec)
        encs.append({k: M.ristretto_encode(v) for k, v in pts.items()})
    return secs, encs


@pytest.mark.parametrize("which,n", [("dleq", 4), ("cmz", 2)])
def test_prover_phases_match_oracle(which, n):
    """phase A (host) -> [MSMs, here by the oracle] -> phase B (host) gives byte-identical proofs."""
    rng = random.Random(7)
    mst = M.dleq_statement() if which == "dleq" else M.cmz_statement(10)
    mod = T.dleq_module() if which == "dleq" else T.cmz_module(10)
    cst = C.Statement.from_model(mst)
    secs, encs = _instances(mst, rng, n, which)
    entropy = np.frombuffer(bytes(rng.randrange(256) for _ in range(32 * n)), np.uint8).reshape(n, 32)
    label = b"Benchmark"
    sec_arr, inst, common = mod.pack(secs, encs)
    ts = np.stack([T.Transcript(label).state for _ in range(n)])
    blind, off, scal, pidx = T.prove_phase_a(mod.statement, ts, sec_arr, inst, common, entropy)
    table = np.concatenate([common, inst.reshape(-1, 32)])
    coms, status = C.msm_many(off, scal, pidx, table, 1)          # stands in for zkp_msm_many in this CPU test only
    assert not status.any()
    chal, resp = T.prove_phase_b(mod.statement, ts, sec_arr, blind, coms.reshape(n, -1, 32))
    for j in range(n):
        ec, er, ek, eb = C.prove(cst, label, arr([sc(secs[j][k]) for k in cst.secrets]), arr([encs[j][k] for k in cst.points]), entropy[j].tobytes())
        assert (blind[j] == eb).all()
        assert (coms.reshape(n, -1, 32)[j] == ek).all()
        assert chal[j].tobytes() == ec.tobytes()
        assert (resp[j] == er).all()
    # transcripts were advanced exactly as the reference advances them: the next challenge agrees with the model
    pr, _ = mst.build_prover(M.Transcript(label), secs[0], {k: M.ristretto_decode(v) for k, v in encs[0].items()})
    pr._prove_impl(entropy[0].tobytes())
    t0 = T.Transcript(_state=ts[0])
    assert t0.challenge_bytes(b"after", 32) == pr.transcript.challenge_bytes(b"after", 32)


def test_batch_verify_build_matches_oracle_macro_order():
    rng = random.Random(8)
    mst, mod = M.dleq_statement(), T.dleq_module()
    cst = C.Statement.from_model(mst)
    n, label = 6, b"DLEQBatchTest"
    secs, encs = _instances(mst, rng, n, "dleq")
    for e in encs:
        e["G"] = encs[0]["G"]
    coms, resps = [], []
    for j in range(n):
        _, r, k, _ = C.prove(cst, label, arr([sc(secs[j]["x"])]), arr([encs[j][p] for p in cst.points]), bytes([j]) * 32)
        coms.append(k)
        resps.append(r)
    coms, resps = np.stack(coms), np.stack(resps)
    _, inst, common = mod.pack([], encs)
    w16 = np.frombuffer(bytes(rng.randrange(256) for _ in range(16 * 2 * n)), np.uint8).reshape(2, n, 16)
    ts = np.stack([T.Transcript(label).state for _ in range(n)])
    ms, mp = T.batch_verify_build(mod.statement, ts, inst, common, coms, resps, w16)
    rc, es, ep = C.batch_verify(cst, label, n, inst, common, coms, resps, w16, want_msm_inputs=True)
    assert rc == 0 and (ms == es).all() and (mp == ep).all()
    assert C.msm_optional(ms, mp) == bytes(32)                          # and the batch is in fact valid
    # identity commitment -> VerificationFailure before any arithmetic; wrong transcript count -> BatchSizeMismatch
    bad = coms.copy()
    bad[2, 1] = 0
    ts = np.stack([T.Transcript(label).state for _ in range(n)])
    with pytest.raises(T.VerificationFailure):
        T.batch_verify_build(mod.statement, ts, inst, common, bad, resps, w16)
    with pytest.raises(T.BatchSizeMismatch):
        T.batch_verify_build(mod.statement, ts[:-1], inst, common, coms, resps, w16)
    # a response that is not a canonical scalar (s + l: same residue, bytes serde would refuse, proofs.rs:27-32): the batch fails on
    # the host before any arithmetic, and in the oracle; l - 1 and 0 are canonical
    big = resps.copy()
    big[4, 0] = np.frombuffer((int.from_bytes(resps[4, 0].tobytes(), "little") + M.L).to_bytes(32, "little"), np.uint8)
    ts = np.stack([T.Transcript(label).state for _ in range(n)])
    with pytest.raises(T.VerificationFailure):
        T.batch_verify_build(mod.statement, ts, inst, common, coms, big, w16)
    assert C.batch_verify(cst, label, n, inst, common, coms, big, w16) == 1
    edge = resps.copy()
    edge[0, 0] = np.frombuffer((M.L - 1).to_bytes(32, "little"), np.uint8)
    edge[1, 0] = 0
    ts = np.stack([T.Transcript(label).state for _ in range(n)])
    T.batch_verify_build(mod.statement, ts, inst, common, coms, edge, w16)       # builds (the MSM would then fail: wrong responses)
    exactly_l = resps.copy()
    exactly_l[2, 0] = np.frombuffer(M.L.to_bytes(32, "little"), np.uint8)
    ts = np.stack([T.Transcript(label).state for _ in range(n)])
    with pytest.raises(T.VerificationFailure):
        T.batch_verify_build(mod.statement, ts, inst, common, coms, exactly_l, w16)


def test_transcript_lengths_beyond_u32_are_an_error_not_a_truncated_prefix():
    """merlin frames lengths as u32 and asserts that they fit (tests/sig_and_vrf_example.rs:224-241 is the ignored > 4 GiB case): the C
    ABI returns ZKP_TB_TOO_LONG before it reads a byte and leaves the transcript untouched."""
    lib = T.lib()
    lib.zkp_transcript_append_message.restype = ctypes.c_int
    lib.zkp_transcript_challenge_bytes.restype = ctypes.c_int
    t = T.Transcript(b"len")
    before = t.state.copy()
    small = ctypes.create_string_buffer(16)
    for n in (1 << 32, (1 << 32) + 5, 1 << 40):
        assert lib.zkp_transcript_append_message(t.state.ctypes.data_as(ctypes.c_void_p), b"msg", small, ctypes.c_size_t(n)) == -14
        assert lib.zkp_transcript_challenge_bytes(t.state.ctypes.data_as(ctypes.c_void_p), b"out", small, ctypes.c_size_t(n)) == -14
        assert (t.state == before).all()
    assert lib.zkp_transcript_append_message(t.state.ctypes.data_as(ctypes.c_void_p), b"msg", small, ctypes.c_size_t(16)) == 0
    assert not (t.state == before).all()


def test_oracle_verifiers_apply_the_canonical_scalar_rule():
    """what serde does in front of the reference's verifiers (proofs.rs:14-32): a challenge / response >= l never verifies"""
    rng = random.Random(10)
    mst = M.dleq_statement()
    cst = C.Statement.from_model(mst)
    secs, encs = _instances(mst, rng, 1, "dleq")
    pts = arr([encs[0][p] for p in cst.points])
    chal, resp, coms, _ = C.prove(cst, b"canon", arr([sc(secs[0]["x"])]), pts, bytes(32))
    w = np.arange(32, dtype=np.uint8).reshape(2, 16)
    assert C.verify_compact(cst, b"canon", pts, chal, resp) == 0 and C.verify_batchable(cst, b"canon", pts, coms, resp, w) == 0
    plus_l = lambda a: np.frombuffer((int.from_bytes(a.tobytes(), "little") + M.L).to_bytes(32, "little"), np.uint8)
    assert C.verify_compact(cst, b"canon", pts, plus_l(chal), resp) == 1
    assert C.verify_compact(cst, b"canon", pts, chal, plus_l(resp[0]).reshape(1, 32)) == 1
    assert C.verify_batchable(cst, b"canon", pts, coms, plus_l(resp[0]).reshape(1, 32), w) == 1


def test_batch_verify_build_constraint_api_order():
    """benches/dleq.rs:188-241: static G, H are allocated BEFORE the instance points A, B."""
    rng = random.Random(9)
    n, label = 4, b"DLEQBatchTest"
    G = M.BASEPOINT
    H = M.ristretto_hash_from_bytes_sha512(M.ristretto_encode(G))
    cst = C.Statement(b"DLEQProof", ["x"], [("G", True), ("H", True), ("A", False), ("B", False)],
                      [("A", [("x", "G")]), ("B", [("x", "H")])])
    st = T.Statement(b"DLEQProof")
    x = st.add_secret(b"x")
    g, h = st.add_point(b"G", True), st.add_point(b"H", True)
    a, b = st.add_point(b"A", False), st.add_point(b"B", False)
    st.constrain(a, [(x, g)])
    st.constrain(b, [(x, h)])
    Ge, He = M.ristretto_encode(G), M.ristretto_encode(H)
    As, Bs, coms, resps = [], [], [], []
    for j in range(n):
        xj = 89327492234 + j                                            # benches/dleq.rs:198
        Ae, Be = M.ristretto_encode(M.pt_mul(xj, G)), M.ristretto_encode(M.pt_mul(xj, H))
        _, r, k, _ = C.prove(cst, label, arr([sc(xj)]), arr([Ge, He, Ae, Be]), bytes([7 + j]) * 32)
        As.append(Ae); Bs.append(Be); coms.append(k); resps.append(r)
    inst = arr(As + Bs).reshape(2, n, 32)
    common = arr([Ge, He])
    coms, resps = np.stack(coms), np.stack(resps)
    w16 = np.frombuffer(bytes(rng.randrange(256) for _ in range(16 * 2 * n)), np.uint8).reshape(2, n, 16)
    ts = np.stack([T.Transcript(label).state for _ in range(n)])
    ms, mp = T.batch_verify_build(st, ts, inst, common, coms, resps, w16)
    rc, es, ep = C.batch_verify(cst, label, n, inst, common, coms, resps, w16, want_msm_inputs=True)
    assert rc == 0 and (ms == es).all() and (mp == ep).all()
    assert C.msm_optional(ms, mp) == bytes(32)


def test_proof_wire_format_roundtrip_and_rejects():
    rng = random.Random(12)
    r = [sc(rng.randrange(M.L)) for _ in range(3)]
    cp = T.CompactProof(sc(rng.randrange(M.L)), r)
    assert T.CompactProof.from_bytes(cp.to_bytes()) == cp and len(cp.to_bytes()) == 32 + 8 + 96
    bp = T.BatchableProof([bytes([i]) * 32 for i in range(1, 3)], r)
    assert T.BatchableProof.from_bytes(bp.to_bytes()) == bp and len(bp.to_bytes()) == 8 + 64 + 8 + 96
    assert T.CompactProof.from_bytes(T.CompactProof(sc(0), []).to_bytes()).responses == []
    for bad in (cp.to_bytes()[:-1], cp.to_bytes() + b"\0", b"", cp.to_bytes()[:32] + (1 << 60).to_bytes(8, "little")):
        with pytest.raises(ValueError):
            T.CompactProof.from_bytes(bad)
    noncanon = T.CompactProof(M.L.to_bytes(32, "little"), r).to_bytes()         # challenge == l: dalek's from_canonical_bytes refuses
    with pytest.raises(ValueError):
        T.CompactProof.from_bytes(noncanon)
    with pytest.raises(ValueError):
        T.BatchableProof.from_bytes(T.BatchableProof([], [((1 << 256) - 1).to_bytes(32, "little")]).to_bytes())


def test_chacha20_block_rfc8439_vector():
    """RFC 8439 section 2.3.2: the generator behind the toolbox's default entropy / batch weights."""
    import ctypes
    key = bytes(range(32))
    out = ctypes.create_string_buffer(64)
    # IETF layout: counter = 1, nonce = 00:00:00:09:00:00:00:4a:00:00:00:00 -> words 13, 14, 15 = 0x09000000, 0x4a000000, 0
    T.lib().zkp_chacha20_block(key, ctypes.c_uint64(1 | (0x09000000 << 32)), ctypes.c_uint64(0x4a000000), out)
    want = bytes.fromhex("10f1e7e4d13b5915500fdd1fa32071c4c7d1f4c733c068030422aa9ac3d46c4e"
                         "d2826446079faa0914c2d705d98b02a2b5129cd1de164eb9cbd083e8a2503c4e")
    assert out.raw == want


def test_headers_are_plain_c(tmp_path):
    """The boundary is a C ABI: both headers must compile as C99 (what bindgen / cgo / a C caller would parse) and as
    C++11, warning-free with -pedantic."""
    import subprocess
    inc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include")
    src = tmp_path / "hdr.c"
    src.write_text('#include "zkp_mi355x.h"\n#include "zkp_toolbox.h"\nint main(void) { return (int)sizeof(zkp_fused_statement) * 0; }\n')
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-pedantic", "-I", inc, "-c", str(src), "-o", str(tmp_path / "a.o")])
    subprocess.check_call(["g++", "-std=c++11", "-Wall", "-Wextra", "-Werror", "-pedantic", "-I", inc, "-x", "c++", "-c", str(src), "-o", str(tmp_path / "b.o")])


def test_batch_verify_rejects_misshapen_proofs_before_any_c_call():
    """batch_verifier.rs:142-149: a proof with the wrong number of commitments / responses is a VerificationFailure, and
    the C side never sees a buffer shorter than [N][nc][32] / [N][m][32] (no engine needed: the check comes first)."""
    mod = T.dleq_module()
    good = T.BatchableProof([bytes([1]) * 32, bytes([2]) * 32], [sc(5)])
    pts = {"A": [bytes([3]) * 32] * 2, "B": [bytes([4]) * 32] * 2, "H": [bytes([5]) * 32] * 2}
    for bad in (T.BatchableProof([bytes([1]) * 32], [sc(5)]), T.BatchableProof([bytes([1]) * 32] * 2, []),
                T.BatchableProof([bytes([1]) * 32] * 3, [sc(5)]), T.BatchableProof([bytes([1]) * 32] * 2, [sc(5), sc(6)])):
        with pytest.raises(T.VerificationFailure):
            mod.batch_verify(None, [good, bad], [T.Transcript(b"x"), T.Transcript(b"x")], pts, {"G": bytes([6]) * 32})
    with pytest.raises(T.BatchSizeMismatch):
        mod.batch_verify(None, [good, good], [T.Transcript(b"x")], pts, {"G": bytes([6]) * 32})
    # the array-level entry points refuse buffers that do not have the statement's shape
    ts = np.stack([T.Transcript(b"x").state] * 2)
    with pytest.raises(ValueError):
        T.batch_verify(None, mod.statement, ts, np.zeros((3, 2, 32), np.uint8), np.zeros((1, 32), np.uint8),
                       np.zeros((2, 1, 32), np.uint8), np.zeros((2, 1, 32), np.uint8))


def test_transcript_blob_padding_is_zero_and_ignored():
    """203 live bytes + 5 bytes of padding: written as zeros, never read."""
    t = T.Transcript(b"pad")
    t.append_message(b"l", b"m" * 300)
    assert not t.state[203:].any()
    u = T.Transcript(_state=t.state.copy())
    u.state[203:] = 0xff
    assert t.challenge_bytes(b"c", 32) == u.challenge_bytes(b"c", 32)
    assert not u.state[203:].any()


def test_interleaved_allocations_host_phase_a_matches_model():
    """Allocation order is part of the statement (prover.rs:52-73): scalars allocated after points are hashed where the
    caller put them.  Host half only (blindings come out of the transcript, so they pin the whole op sequence)."""
    rng = random.Random(31)
    label = b"InterleaveTest"
    Bp, Hp = M.BASEPOINT, M.ristretto_hash_from_bytes_sha512(b"interleaved")
    x, y = rng.randrange(1, M.L), rng.randrange(1, M.L)
    pts = {"B": Bp, "H": Hp, "A": M.pt_add(M.pt_mul(x, Bp), M.pt_mul(y, Hp)), "G": M.pt_mul(y, Bp)}
    ent = bytes(rng.randrange(256) for _ in range(32))
    mp = M.Prover(b"Interleaved", M.Transcript(label))
    vB, _ = mp.allocate_point(b"B", pts["B"]); vx = mp.allocate_scalar(b"x", x); vH, _ = mp.allocate_point(b"H", pts["H"])
    vy = mp.allocate_scalar(b"y", y); vA, _ = mp.allocate_point(b"A", pts["A"]); vG, _ = mp.allocate_point(b"G", pts["G"])
    mp.constrain(vA, [(vx, vB), (vy, vH)])
    mp.constrain(vG, [(vy, vB)])
    want = mp.prove_batchable(ent)
    st = T.Statement(b"Interleaved")
    b_ = st.add_point(b"B", False); x_ = st.add_secret(b"x"); h_ = st.add_point(b"H", False); y_ = st.add_secret(b"y")
    a_ = st.add_point(b"A", False); g_ = st.add_point(b"G", False)
    st.constrain(a_, [(x_, b_), (y_, h_)])
    st.constrain(g_, [(y_, b_)])
    ts = T.Transcript(label).state.reshape(1, -1).copy()
    secrets = arr([sc(x), sc(y)]).reshape(1, 2, 32)
    inst = arr([M.ristretto_encode(pts[k]) for k in ("B", "H", "A", "G")]).reshape(4, 1, 32)
    blind, off, scal, pidx = T.prove_phase_a(st, ts, secrets, inst, np.zeros((0, 32), np.uint8), np.frombuffer(ent, np.uint8).reshape(1, 32))
    coms = arr(list(want.commitments)).reshape(1, 2, 32)      # the model's commitments stand in for the MSM (CPU test)
    chal, resp = T.prove_phase_b(st, ts, secrets, blind, coms)
    assert [r.tobytes() for r in resp[0]] == [M.sc_to_bytes(r) for r in want.responses]


def test_c_wire_codec_edges():
    """zkp_proof_*_{encode,decode} (include/zkp_toolbox.h) directly: sizes, the consumed count, bincode's tolerance of trailing
    bytes, oversized length prefixes, output capacity, and dalek's canonical-scalar rule on every scalar field -- while a
    commitment may be ANY 32 bytes (CompressedRistretto deserialises blindly; decompress() decides later)."""
    lib = T.lib()
    rng = random.Random(13)
    r = [sc(rng.randrange(M.L)) for _ in range(3)]
    assert lib.zkp_proof_compact_size(3) == 32 + 8 + 96 and lib.zkp_proof_batchable_size(2, 3) == 8 + 64 + 8 + 96
    cp = T.CompactProof(sc(rng.randrange(M.L)), r)
    raw = cp.to_bytes()
    assert raw[:32] == cp.challenge and raw[32:40] == (3).to_bytes(8, "little") and raw[40:] == b"".join(r)
    assert T.CompactProof.from_bytes(raw + b"junk", allow_trailing=True) == cp           # bincode::deserialize ignores what follows
    with pytest.raises(ValueError):
        T.CompactProof.from_bytes(raw + b"junk")
    bp = T.BatchableProof([b"\xff" * 32, bytes(32)], r)                                    # non-canonical field element / identity: still parses
    braw = bp.to_bytes()
    assert braw[:8] == (2).to_bytes(8, "little") and T.BatchableProof.from_bytes(braw) == bp
    # capacity: the decoder never writes more than max_* elements
    chal = ctypes.create_string_buffer(32)
    out = np.zeros((2, 32), np.uint8)
    m, used = ctypes.c_uint32(0), ctypes.c_size_t(0)
    assert lib.zkp_proof_compact_decode(raw, len(raw), chal, out.ctypes.data_as(ctypes.c_void_p), 2, ctypes.byref(m), ctypes.byref(used)) == -10
    out = np.zeros((3, 32), np.uint8)
    assert lib.zkp_proof_compact_decode(raw, len(raw), chal, out.ctypes.data_as(ctypes.c_void_p), 3, ctypes.byref(m), ctypes.byref(used)) == 0
    assert m.value == 3 and used.value == len(raw) and chal.raw == cp.challenge and out.tobytes() == b"".join(r)
    # malformed inputs: truncated anywhere, absurd length prefixes, non-canonical scalars in every scalar position
    for cut in (0, 31, 39, 40, len(raw) - 1):
        with pytest.raises(ValueError):
            T.CompactProof.from_bytes(raw[:cut])
    for n_claimed in (4, 1 << 32, (1 << 64) - 1):
        with pytest.raises(ValueError):
            T.CompactProof.from_bytes(raw[:32] + n_claimed.to_bytes(8, "little") + raw[40:])
    for bad in (M.L, M.L + 1, (1 << 256) - 1):
        b32 = bad.to_bytes(32, "little")
        with pytest.raises(ValueError):
            T.CompactProof.from_bytes(b32 + raw[32:])
        with pytest.raises(ValueError):
            T.CompactProof.from_bytes(raw[:40 + 32] + b32 + raw[40 + 64:])
        with pytest.raises(ValueError):
            T.BatchableProof.from_bytes(braw[:-32] + b32)
    assert T.CompactProof.from_bytes((M.L - 1).to_bytes(32, "little") + raw[32:]).challenge == (M.L - 1).to_bytes(32, "little")
    for cut in (0, 7, 8 + 63, 8 + 64 + 7, len(braw) - 1):
        with pytest.raises(ValueError):
            T.BatchableProof.from_bytes(braw[:cut])
    empty = T.BatchableProof([], [])
    assert empty.to_bytes() == bytes(16) and T.BatchableProof.from_bytes(bytes(16)) == empty


def test_rust_sys_crate_declares_every_exported_symbol():
    """rust/zkp-mi355x-sys/src/lib.rs is unbuilt source (no Rust toolchain in the image); at least keep it in step with the
    C ABI: one `pub fn` per symbol the two headers declare, nothing else."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = open(os.path.join(root, "rust", "zkp-mi355x-sys", "src", "lib.rs")).read()
    declared = set(re.findall(r"pub fn (zkp_\w+)\s*\(", src))
    assert declared == set(engine.EXPORTS) | set(T.EXPORTS), (declared ^ (set(engine.EXPORTS) | set(T.EXPORTS)))
    assert "UNBUILT SOURCE" in src and "UNBUILT" in open(os.path.join(root, "rust", "README.md")).read()


def test_debug_transcript_env_dumps_the_op_log():
    """ZKP_DEBUG_TRANSCRIPT=1 (the reference's `debug-transcript` feature, Cargo.toml:35): every Merlin operation of the host
    transcripts goes to stderr -- label, length, leading bytes; witness bytes never."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("from zkp_amd import toolbox as T\n"
            "t = T.Transcript(b'dbg')\n"
            "t.append_message(b'msg', b'hello world')\n"
            "print(t.challenge_bytes(b'c', 8).hex())\n")
    r = subprocess.run([sys.executable, "-c", code], cwd=root, env=dict(os.environ, ZKP_DEBUG_TRANSCRIPT="1"), capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    assert '[merlin] append    label="dom-sep" len=3 data=646267' in r.stderr
    assert '[merlin] append    label="msg" len=11 data=68656c6c6f20776f726c64' in r.stderr
    assert '[merlin] challenge label="c" len=8 data=' + r.stdout.strip() in r.stderr
    quiet = subprocess.run([sys.executable, "-c", code], cwd=root, env={k: v for k, v in os.environ.items() if k != "ZKP_DEBUG_TRANSCRIPT"},
                           capture_output=True, text=True, timeout=120)
    assert quiet.returncode == 0 and "[merlin]" not in quiet.stderr and quiet.stdout == r.stdout


def test_option_numbers_agree_between_header_python_and_rust():
    """zkp_ctx_set_option's option ids are part of the ABI: include/zkp_mi355x.h's enum, zkp_amd/engine.py and rust/zkp-mi355x-sys must say the same."""
    import re
    from zkp_amd import engine as E
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    header = open(os.path.join(root, "include", "zkp_mi355x.h")).read()
    enum = {k: int(v) for k, v in re.findall(r"(ZKP_OPT_[A-Z0-9_]+) = (\d+)", header)}
    assert len(enum) >= 14 and sorted(set(enum.values())) == list(range(1, len(set(enum.values())) + 1))      # (ids are dense; ZKP_OPT_CT_MASKED_SCANS is an alias of ZKP_OPT_CT_LOOKUP)
    for k, v in enum.items():
        if hasattr(E, k):
            assert getattr(E, k) == v, k
        assert re.search(r"ZKP_OPT_[A-Z_]+", k) and ("%s:" % k) in header or k in header
    rust = open(os.path.join(root, "rust", "zkp-mi355x-sys", "src", "lib.rs")).read()
    rs = {k: int(v) for k, v in re.findall(r"pub const (ZKP_OPT_[A-Z0-9_]+): c_int = (\d+);", rust)}
    assert rs == enum
    for k in enum:                                   # every option is documented in the header's comment block
        assert (" *   %s:" % k) in header, k


def test_crossbar_layout_is_bank_conflict_free_for_every_run_structure():
    """The constant-time argument of the grouped comb walk (comb_tables.h: comb_group_xbar), checked as arithmetic: ds_bpermute_b32 is served in two groups of 32
    lanes and two lanes of a group conflict iff their source lanes differ by 32 (profiles/r05_bpermute_microbench.txt).  A half of a wavefront takes XBAR_HALF_TERMS
    consecutive entries of a list in which every table's terms form a run of >= GROUP_MIN_USES entries; a lane of run r (counted from the wavefront's first entry)
    reads lanes 8 r .. 8 r + 7.  For every run structure the constants admit: at most 8 runs per wavefront, and within a half no two POSSIBLE sources 32 apart --
    whatever the secret digits are.  The constants are read from the headers, so the test fails if somebody relaxes one of them."""
    import re
    from hypothesis import given, settings, strategies as st_
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hot = open(os.path.join(root, "zkp_amd", "csrc", "hot_tables.h")).read()
    comb = open(os.path.join(root, "zkp_amd", "csrc", "comb_tables.h")).read()
    gmin = int(re.search(r"constexpr uint32_t GROUP_MIN_USES = (\d+);", hot).group(1))
    half = int(re.search(r"XBAR_HALF_TERMS = (\d+)", comb).group(1))
    runs_max = int(re.search(r"constexpr uint32_t XBAR_RUNS = (\d+);", comb).group(1))
    assert half <= 32 and runs_max == 8

    def check(run_lengths, start):
        # entry e of the list belongs to table owner[e]; the wavefront takes entries [start, start + 2 * half)
        owner = [t for t, n in enumerate(run_lengths) for _ in range(n)]
        for w0 in range(start, len(owner) - 2 * half + 1, 2 * half):
            entries = owner[w0:w0 + 2 * half]
            tables = sorted(set(entries), key=entries.index)
            assert len(tables) <= runs_max, (run_lengths, w0)
            for h in range(2):
                runs = sorted({tables.index(t) for t in entries[h * half:(h + 1) * half]})
                sources = {8 * r + k for r in runs for k in range(8)}                      # every lane a lane of this half MAY read, over all digits
                assert len({s % 32 for s in sources}) == len(sources), (run_lengths, w0, h, runs)      # distinct banks: no two sources 32 apart

    @settings(max_examples=400, deadline=None)
    @given(st_.lists(st_.integers(min_value=gmin, max_value=3 * gmin + 5), min_size=8, max_size=40), st_.integers(min_value=0, max_value=61))
    def prop(run_lengths, start):
        check(run_lengths, start)
    prop()
    check([gmin] * 64, 0)                                   # the CMZ shape: ten terms of P per proof, proof after proof
    check([gmin] * 64, 7)
    # and the bound is tight: one use fewer per table admits a half with five tables
    with pytest.raises(AssertionError):
        for s in range(2 * half):
            check([gmin - 1] * 64, s)


def _scatter_block_to_tile_window(lin, tiles, WK):
    """k_pip_tile_scatter's block order (zkp_amd/csrc/zkp_kernels.hip): linear workgroup index -> (tile, window)"""
    full = (WK & ~7) * tiles
    if lin < full:
        grp, r = divmod(lin, 8 * tiles)
        return r >> 3, grp * 8 + (r & 7)
    return lin % tiles, lin // tiles


@pytest.mark.parametrize("tiles,WK", [(1, 1), (1, 8), (5, 120), (5, 24), (21, 24), (7, 17), (3, 7), (40, 129), (512, 17), (2, 16)])
def test_scatter_pass_block_order_is_a_bijection_that_keeps_a_window_on_one_xcd(tiles, WK):
    """Round 5: the counting sort's scatter pass deals whole windows to XCDs (workgroups go to the 8 XCDs round robin by linear index).  The kernel's index
    arithmetic, restated: every (tile, window) pair is visited exactly once, and all tiles of a window inside the full groups of 8 land on XCD window % 8."""
    seen = set()
    xcd_of = {}
    for lin in range(tiles * WK):
        t, w = _scatter_block_to_tile_window(lin, tiles, WK)
        assert 0 <= t < tiles and 0 <= w < WK
        seen.add((t, w))
        xcd_of.setdefault(w, set()).add(lin % 8)
    assert len(seen) == tiles * WK
    for w in range(WK & ~7):
        assert xcd_of[w] == {w % 8}
    # the source carries this formula
    src = open(os.path.join(ROOT, "zkp_amd", "csrc", "zkp_kernels.hip")).read()
    assert "w = grp * 8u + (r & 7u);" in src and "full = (WK & ~7u) * tiles" in src
