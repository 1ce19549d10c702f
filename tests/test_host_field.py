"""CPU tests of the DEVICE field / group headers (zkp_amd/csrc/fe25519.h, ge25519.h) compiled for the
host, against the big-integer oracle (oracle/model.py).  The same headers are what the HIP kernels
compile, so limb arithmetic, lazy-reduction bounds and the ristretto codec are proven here without
a GPU.  The *_track build carries interval bounds through every operation and aborts on any
possible 32/64-bit overflow."""
import ctypes
import os
import random
import subprocess

import pytest

from oracle import model as M

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "host", "fe_host_lib.cpp")


def _build(variant: str):
    out = os.path.join(HERE, "host", {"plain": "fe_host_lib.so", "bound-tracked": "fe_host_lib_track.so", "host-fe51": "fe_host_lib_fe51.so"}[variant])
    deps = [SRC] + [os.path.join(HERE, "..", "zkp_amd", "csrc", f) for f in ("fe25519.h", "ge25519.h", "fe_constants.h", "sc25519.h", os.path.join("host", "fe51.h"))]
    if not os.path.exists(out) or any(os.path.getmtime(d) > os.path.getmtime(out) for d in deps):
        cmd = ["g++", "-O1", "-std=c++17", "-shared", "-fPIC", SRC, "-o", out]
        if variant == "bound-tracked":
            cmd.insert(1, "-DZKP_FE_TRACK")
        if variant == "host-fe51":                     # the field of the host backend (zkp_amd/csrc/host/fe51.h) under the same point formulas
            cmd.insert(1, "-DZKP_HOST_FE51")
        subprocess.check_call(cmd)
    return ctypes.CDLL(out)


@pytest.fixture(scope="module", params=["plain", "bound-tracked", "host-fe51"])
def lib(request):
    return _build(request.param)


def fe_bytes(x):
    return (x % M.P).to_bytes(32, "little")


EDGE = [0, 1, 2, 19, M.P - 1, M.P - 2, M.P - 19, (1 << 255) - 20, (1 << 254), (1 << 29) - 1, 1 << 29,
        (1 << 232) - 1, 1 << 232, M.SQRT_M1, M.D, (M.P - 1) // 2]


def test_fe_ops(lib):
    rng = random.Random(7)
    vals = EDGE + [rng.randrange(M.P) for _ in range(200)]
    out = ctypes.create_string_buffer(32)
    for _ in range(400):
        a, b = rng.choice(vals), rng.choice(vals)
        exp = {0: a * b, 1: a * a, 2: a + b, 3: a - b, 4: -a, 5: (a - b) * (a + b), 7: 2 * a + b}
        for op, e in exp.items():
            lib.t_fe_binop(op, fe_bytes(a), fe_bytes(b), out)
            assert out.raw == fe_bytes(e), (op, a, b)
    for a in vals[:40]:
        lib.t_fe_binop(6, fe_bytes(a), fe_bytes(0), out)
        assert out.raw == fe_bytes(pow(a, (M.P - 5) // 8, M.P))


def test_fe_canonical_and_towords(lib):
    rng = random.Random(8)
    for x in [M.P - 1, M.P, M.P + 1, (1 << 255) - 1, 1 << 255, (1 << 256) - 1, 0, M.P - 19, (1 << 255) - 19 + 18]:
        assert lib.t_fe_canonical(x.to_bytes(32, "little")) == int(x < M.P)
    out = ctypes.create_string_buffer(32)
    if lib.t_is_fe51():
        Limbs5 = ctypes.c_uint64 * 5
        m51 = (1 << 51) - 1
        cases5 = [[(1 << 63) - 1] * 5, [(1 << 52) - 38] + [(1 << 52) - 2] * 4, [m51] * 5, [m51 - 18] + [m51] * 4, [m51 - 19] + [m51] * 4, [m51 + 19] + [m51] * 4, [0] * 5]
        cases5 += [[rng.randrange(1 << 63) for _ in range(5)] for _ in range(300)] + [[rng.randrange(1 << 52) for _ in range(5)] for _ in range(300)]
        for limbs in cases5:
            lib.t_fe51_towords_raw(Limbs5(*limbs), out)
            assert out.raw == fe_bytes(sum(l << (51 * i) for i, l in enumerate(limbs))), limbs
        for which, val in enumerate([M.D, 2 * M.D % M.P, M.SQRT_M1, M.INVSQRT_A_MINUS_D]):
            lib.t_fe51_const(which, out)
            assert out.raw == fe_bytes(val), which
        return
    Limbs = ctypes.c_uint32 * 9
    cases = [[0xfffffff0] * 9, [0x3fffffda] + [0x3ffffffe] * 7 + [0x00fffffe], [(1 << 29) - 1] * 8 + [(1 << 23) - 1],
             [(1 << 29) - 19] + [(1 << 29) - 1] * 7 + [(1 << 23) - 1], [(1 << 29) - 20] + [(1 << 29) - 1] * 7 + [(1 << 23) - 1]]
    cases += [[rng.randrange(0xfffffff0) for _ in range(9)] for _ in range(300)]
    for limbs in cases:
        val = sum(l << (29 * i) for i, l in enumerate(limbs))
        lib.t_fe_towords_raw(Limbs(*limbs), out)
        assert out.raw == fe_bytes(val), limbs


def _rand_point(rng):
    return M.pt_mul(rng.randrange(1, M.L), M.BASEPOINT)


def test_ristretto_codec(lib):
    rng = random.Random(9)
    out = ctypes.create_string_buffer(32)
    xyzt = ctypes.create_string_buffer(128)
    # valid points: decode gives the same affine point as the oracle, re-encode is the identity map
    for i in range(60):
        p = M.IDENTITY if i == 0 else _rand_point(rng)
        enc = M.ristretto_encode(p)
        assert lib.t_decode(enc, xyzt) == 1
        q = M.ristretto_decode(enc)
        got = [int.from_bytes(xyzt.raw[32 * k:32 * k + 32], "little") for k in range(4)]
        assert got == [q[0], q[1], 1, q[3]]
        assert lib.t_recode(enc, out) == 1 and out.raw == enc
    # random strings: same accept / reject decision as the oracle (RFC 9496 section 4.3.1)
    n_valid = 0
    for _ in range(600):
        rb = bytes(rng.randrange(256) for _ in range(31)) + bytes([rng.randrange(256) & (0xff if rng.random() < 0.1 else 0x7f)])
        ok = M.ristretto_decode(rb) is not None
        n_valid += ok
        assert lib.t_decode(rb, xyzt) == int(ok), rb.hex()
    assert n_valid > 20


# RFC 9496 appendix A.3: encodings that must be rejected
BAD_ENCODINGS = """
00ffffffffffffffffffffffffffffffffffffffffffffffffffffffffffffff
ffffffffffffffffffffffffffffffffffffffffffffffffffffffffffffff7f
f3ffffffffffffffffffffffffffffffffffffffffffffffffffffffffffff7f
edffffffffffffffffffffffffffffffffffffffffffffffffffffffffffff7f
0100000000000000000000000000000000000000000000000000000000000000
01ffffffffffffffffffffffffffffffffffffffffffffffffffffffffffff7f
ed57ffd8c914fb201471d1c3d245ce3c746fcbe63a3679d51b6a516ebebe0e20
c34c4e1826e5d403b78e246e88aa051c36ccf0aafebffe137d148a2bf9104562
c940e5a4404157cfb1628b108db051a8d439e1a421394ec4ebccb9ec92a8ac78
47cfc5497c53dc8e61c91d17fd626ffb1c49e2bca94eed052281b510b1117a24
f1c6165d33367351b0da8f6e4511010c68174a03b6581212c71c0e1d026c3c72
87260f7a2f12495118360f02c26a470f450dadf34a413d21042b43b9d93e1309
26948d35ca62e643e26a83177332e6b6afeb9d08e4268b650f1f5bbd8d81d371
4eac077a713c57b4f4397629a4145982c661f48044dd3f96427d40b147d9742f
de6a7b00deadc788eb6b6c8d20c0ae96c2f2019078fa604fee5b87d6e989ad7b
bcab477be20861e01e4a0e295284146a510150d9817763caf1a6f4b422d67042
2a292df7e32cababbd9de088d1d1abec9fc0440f637ed2fba145094dc14bea08
f4a9e534fc0d216c44b218fa0c42d99635a0127ee2e53c712f70609649fdff22
8268436f8c4126196cf64b3c7ddbda90746a378625f9813dd9b8457077256731
2810e5cbc2cc4d4eece54f61c6f69758e289aa7ab440b3cbeaa21995c2f4232b
3eb858e78f5a7254d8c9731174a94f76755fd3941c0ac93735c07ba14579630e
a45fdc55c76448c049a1ab33f17023edfb2be3581e9c7aade8a6125215e04220
d483fe813c6ba647ebbfd3ec41adca1c6130c2beeee9d9bf065c8d151c5f396e
8a2e1d30050198c65a54483123960ccc38aef6848e1ec8f5f780e8523769ba32
32888462f8b486c68ad7dd9610be5192bbeaf3b443951ac1a8118419d9fa097b
227142501b9d4355ccba290404bde41575b037693cef1f438c47f8fbf35d1165
5c37cc491da847cfeb9281d407efc41e15144c876e0170b499a96a22ed31e01e
445425117cb8c90edcbc7c1cc0e74f747f2c1efa5630a967c64f287792a48a4b
""".split()

# RFC 9496 appendix A.1: multiples 0..15 of the generator
GENERATOR_MULTIPLES = """
0000000000000000000000000000000000000000000000000000000000000000
e2f2ae0a6abc4e71a884a961c500515f58e30b6aa582dd8db6a65945e08d2d76
6a493210f7499cd17fecb510ae0cea23a110e8d5b901f8acadd3095c73a3b919
94741f5d5d52755ece4f23f044ee27d5d1ea1e2bd196b462166b16152a9d0259
da80862773358b466ffadfe0b3293ab3d9fd53c5ea6c955358f568322daf6a57
e882b131016b52c1d3337080187cf768423efccbb517bb495ab812c4160ff44e
f64746d3c92b13050ed8d80236a7f0007c3b3f962f5ba793d19a601ebb1df403
44f53520926ec81fbd5a387845beb7df85a96a24ece18738bdcfa6a7822a176d
903293d8f2287ebe10e2374dc1a53e0bc887e592699f02d077d5263cdd55601c
02622ace8f7303a31cafc63f8fc48fdc16e1c8c8d234b2f0d6685282a9076031
20706fd788b2720a1ed2a5dad4952b01f413bcf0e7564de8cdc816689e2db95f
bce83f8ba5dd2fa572864c24ba1810f9522bc6004afe95877ac73241cafdab42
e4549ee16b9aa03099ca208c67adafcafa4c3f3e4e5303de6026e3ca8ff84460
aa52e000df2e16f55fb1032fc33bc42742dad6bd5a8fc0be0167436c5948501f
46376b80f409b29dc2b5f6f0c52591990896e5716f41477cd30085ab7f10301e
e0c418f7c8d9c4cdd7395b93ea124f3ad99021bb681dfc3302a9d99a2e53e64e
""".split()


def test_rfc9496_vectors(lib):
    out = ctypes.create_string_buffer(32)
    xyzt = ctypes.create_string_buffer(128)
    for h in BAD_ENCODINGS:
        b = bytes.fromhex(h)
        assert M.ristretto_decode(b) is None
        assert lib.t_decode(b, xyzt) == 0, h
    for k, h in enumerate(GENERATOR_MULTIPLES):
        assert M.ristretto_encode(M.pt_mul(k, M.BASEPOINT)).hex() == h
        lib.t_scalarmult(k.to_bytes(32, "little"), bytes.fromhex(GENERATOR_MULTIPLES[1]), out)
        assert out.raw.hex() == h


def test_point_ops(lib):
    rng = random.Random(10)
    out = ctypes.create_string_buffer(32)
    pts = [M.IDENTITY] + [_rand_point(rng) for _ in range(12)]
    for _ in range(60):
        p, q = rng.choice(pts), rng.choice(pts)
        pe, qe = M.ristretto_encode(p), M.ristretto_encode(q)
        pd, qd = M.ristretto_decode(pe), M.ristretto_decode(qe)
        exp = {0: M.pt_add(pd, qd), 1: M.pt_add(pd, M.pt_neg(qd)), 2: M.pt_add(pd, qd), 3: M.pt_add(pd, M.pt_neg(qd)),
               4: M.pt_double(pd), 5: M.pt_add(M.pt_mul(8, pd), qd), 6: M.pt_add(pd, M.pt_neg(qd)), 7: M.pt_add(pd, M.pt_neg(qd))}
        for op, e in exp.items():
            assert lib.t_point_op(op, pe, qe, out) == 1
            assert out.raw == M.ristretto_encode(e), op


def test_scalarmult_long_chain(lib):
    rng = random.Random(11)
    out = ctypes.create_string_buffer(32)
    for s in [0, 1, M.L - 1, M.L, (1 << 256) - 1] + [rng.randrange(1 << 256) for _ in range(6)]:
        p = _rand_point(rng)
        lib.t_scalarmult(s.to_bytes(32, "little"), M.ristretto_encode(p), out)
        assert out.raw == M.ristretto_encode(M.pt_mul(s % M.L, p)), s


def test_scalar_arithmetic_mod_l(lib):
    """sc25519.h (device header, used by the GPU coefficient build) against Python integers."""
    rng = random.Random(12)
    out = ctypes.create_string_buffer(32)
    b32 = lambda x: x.to_bytes(32, "little")
    edge = [0, 1, 2, M.L - 1, M.L - 2, (1 << 252), (1 << 128) - 1, (1 << 252) - 1]
    vals = edge + [rng.randrange(M.L) for _ in range(200)]
    for _ in range(600):
        a, b = rng.choice(vals), rng.choice(vals)
        for op, e in {0: a * b % M.L, 1: (a + b) % M.L, 2: (-a) % M.L, 4: a * b % M.L}.items():
            lib.t_sc_op(op, b32(a), b32(b), out)
            assert out.raw == b32(e), (op, a, b)
    for a in [M.L, M.L + 1, (1 << 256) - 1, 1 << 255, 15 * M.L + 3] + [rng.randrange(1 << 256) for _ in range(100)]:
        lib.t_sc_op(3, b32(a), b32(0), out)                 # reduce any 256-bit value
        assert out.raw == b32(a % M.L)
        lib.t_sc_op(0, b32(a), b32(7), out)                 # first operand of a product may be non-canonical
        assert out.raw == b32(a * 7 % M.L)
    half = (M.L - 1) // 2                                    # sign folding: s -> l - s iff half < s <= l
    for a in [0, 1, half - 1, half, half + 1, M.L - 1, M.L, M.L + 1, (1 << 256) - 1, M.L - (1 << 128) + 5, 1 << 252] + \
            [rng.randrange(M.L) for _ in range(200)] + [M.L - rng.randrange(1 << 128) for _ in range(50)]:
        lib.t_sc_op(6, b32(a), b32(0), out)
        v = int.from_bytes(out.raw, "little")
        fold = half < a <= M.L                               # l itself folds to 0 (same group element)
        if a >> 255:
            assert v == a                                   # never folded (>= l), value untouched
        else:
            assert (v >> 255) == int(fold) and (v & ((1 << 255) - 1)) == (M.L - a if fold else a), a
    inv2 = (M.L + 1) // 2
    for a in [0, 1, 2, 3, M.L - 1, M.L, M.L + 1, (1 << 256) - 1] + [rng.randrange(1 << 256) for _ in range(100)]:
        lib.t_sc_op(7, b32(a), b32(0), out)                 # halving (the MSM kernels emit encode(2 * sum (s/2) P))
        assert out.raw == b32(a * inv2 % M.L)
    wide = [0, 1, M.L - 1, M.L, M.L + 1, (1 << 256) - 1, (1 << 512) - 1, (1 << 256), (1 << 256) + M.L, M.L << 256, (M.L << 256) - 1,
            ((1 << 512) - 1) // M.L * M.L, ((1 << 512) - 1) // M.L * M.L - 1] + [rng.randrange(1 << 512) for _ in range(300)]
    for x in wide:                                          # Scalar::from_bytes_mod_order_wide
        lib.t_sc_op(5, b32(x & ((1 << 256) - 1)), b32(x >> 256), out)
        assert out.raw == b32(x % M.L)


def test_double_and_compress_without_square_root(lib):
    """ristretto_dc_prepare / _finish (ge25519.h): encode(2 P) from one inversion -- the identity behind
    curve25519-dalek's double_and_compress_batch -- against the model's encode(double(P)), for random points in random
    projective scalings (and, in the bound-tracked build, with every lazy add/sub chain checked)."""
    rng = random.Random(77)
    out = ctypes.create_string_buffer(32)
    B = M.ristretto_decode(bytes.fromhex(GENERATOR_MULTIPLES[1]))
    for i in range(120):
        k = rng.randrange(1, M.L)
        Pt = M.pt_mul(k, B)
        z = rng.randrange(1, M.P) if i % 3 else 1
        rc = lib.t_double_compress(M.ristretto_encode(Pt), z.to_bytes(32, "little"), out)
        assert rc == 1 and out.raw == M.ristretto_encode(M.pt_double(Pt)), i
    assert lib.t_double_compress(bytes(32), (1).to_bytes(32, "little"), out) == 2       # identity: x = 0, caller falls back
