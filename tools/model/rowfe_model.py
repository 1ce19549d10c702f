"""Lane-level model of zkp_amd/csrc/rowfe.h: GF(2^255-19) with ONE LIMB PER LANE (9 limbs in the low lanes of a 16-lane DPP row, one row per
coordinate of an extended point), the arithmetic of the 253-doubling Horner tail of k_pip_combine.  Every function below is the HIP routine of
the same name, instruction for instruction, over lists of 64 Python ints with the DPP moves modelled exactly (row_shr / row_shl with
bound_ctrl = zero fill, row_newbcast) and the gfx950 lane swaps between rows as probed on the hardware (profiles/r05_permlane_probe.txt).  It checks (1) values against big-integer arithmetic
and (2) that no intermediate exceeds its register width for operands at the top of the limb classes the callers use.
Run:  python tools/model/rowfe_model.py     (also imported by tests/test_rowfe_model.py)"""
import random

P = 2**255 - 19
M29, M23 = 2**29 - 1, 2**23 - 1
D = (-121665 * pow(121666, P - 2, P)) % P
D2 = 2 * D % P
K = [l & 15 for l in range(64)]
R = [l >> 4 for l in range(64)]
MASK = [M29 if k < 8 else (M23 if k == 8 else 0) for k in K]
SH = [29 if k < 8 else 23 for k in K]
BIAS2P = [(0x3fffffda if k == 0 else 0x00fffffe if k == 8 else 0x3ffffffe) if k < 9 else 0 for k in K]
BIAS4P = [2 * b for b in BIAS2P]
F_X = [1216 if k == 0 else (19 if k == 1 else 0) for k in K]
G_Y = [19 if k == 0 else 0 for k in K]


class Overflow(Exception):
    pass


def u32(v, what):
    if not 0 <= v < 2**32:
        raise Overflow("%s does not fit 32 bits: %d" % (what, v))
    return v


def u64(v, what):
    if not 0 <= v < 2**64:
        raise Overflow("%s does not fit 64 bits: %d" % (what, v))
    return v


def shr(v, n):
    return [v[l - n] if K[l] - n >= 0 else 0 for l in range(64)]


def shl(v, n):
    return [v[l + n] if K[l] + n <= 15 else 0 for l in range(64)]


def bcast(v, j):
    return [v[16 * R[l] + j] for l in range(64)]


def pull(v, rowmap):
    return [v[16 * rowmap[R[l]] + K[l]] for l in range(64)]


def swap32(v, s):
    """v_permlane32_swap_b32 v, s: lanes 32 .. 63 of v <-> lanes 0 .. 31 of s (profiles/r05_permlane_probe.txt); returns (v', s')"""
    return v[:32] + s[:32], v[32:] + s[32:]


def swap16(v, s):
    """v_permlane16_swap_b32 v, s: the odd 16-lane rows of v <-> the even rows of s; returns (v', s')"""
    vo, so = list(v), list(s)
    for half in (0, 32):
        vo[half + 16:half + 32], so[half:half + 16] = s[half:half + 16], v[half + 16:half + 32]
    return vo, so


def row_bcast01(x):
    """rows 0 and 1 of x, each in front of every row: three lane swaps (rowfe.h)"""
    lo, _ = swap32(x, x)                   # rows 0 1 0 1
    return swap16(lo, lo)                  # (0 0 0 0), (1 1 1 1)


def row_bcast23(x):
    _, hi = swap32(x, x)                   # rows 2 3 2 3
    return swap16(hi, hi)


def row_bcast_all(x):
    lo, hi = swap32(x, x)
    return swap16(lo, lo) + swap16(hi, hi)


def row_mul(a, b):
    """a: zero in lanes 9..15 of every row.  b: anything there."""
    B = [bcast(b, j) for j in range(9)]
    lo = [u64(a[l] * B[0][l], "lo") for l in range(64)]
    hi = [0] * 64
    for j in range(1, 9):
        s, t = shr(a, j), shl(a, 9 - j)
        lo = [u64(lo[l] + s[l] * B[j][l], "lo") for l in range(64)]
        hi = [u64(hi[l] + t[l] * B[j][l], "hi") for l in range(64)]
    hi32 = shr([h >> 32 for h in hi], 1)
    col = [u64(lo[l] + 1216 * (hi[l] & 0xffffffff), "col") for l in range(64)]
    col = [u64(col[l] + 9728 * hi32[l], "col") for l in range(64)]
    return row_reduce(col)


def row_reduce(col):
    low = [c & MASK[l] for l, c in enumerate(col)]
    t = [c >> SH[l] for l, c in enumerate(col)]
    m = [u32(x & M29, "m") for x in t]
    h = [u32(x >> 29, "h") for x in t]
    m1, h2, X, Y = shr(m, 1), shr(h, 2), shl(h, 7), shl(m, 8)
    new = [u32(low[l] + m1[l] + h2[l], "l+m+h") for l in range(64)]
    new = [u64(new[l] + F_X[l] * X[l], "new") for l in range(64)]
    new = [u64(new[l] + G_Y[l] * Y[l], "new") for l in range(64)]
    c = [u32(n >> SH[l], "c") for l, n in enumerate(new)]
    r = [(n & 0xffffffff) & MASK[l] for l, n in enumerate(new)]
    c1, c8 = shr(c, 1), shl(c, 8)
    out = [u32(r[l] + c1[l] + G_Y[l] * c8[l], "out") for l in range(64)]
    return [o if K[l] < 9 else 0 for l, o in enumerate(out)]


def row_carry(v):
    c = [x >> SH[l] for l, x in enumerate(v)]
    r = [x & MASK[l] for l, x in enumerate(v)]
    c1, c8 = shr(c, 1), shl(c, 8)
    out = [u32(r[l] + c1[l] + G_Y[l] * c8[l], "carry out") for l in range(64)]
    return [o if K[l] < 9 else 0 for l, o in enumerate(out)]


def add(a, b):
    return [u32(x + y, "add") for x, y in zip(a, b)]


def sub2p(a, b):
    for l in range(64):
        if b[l] > BIAS2P[l]:
            raise Overflow("sub2p: subtrahend above the bias")
    return [u32(a[l] + (BIAS2P[l] - b[l]), "sub") for l in range(64)]


def sub4p(a, b):
    for l in range(64):
        if b[l] > BIAS4P[l]:
            raise Overflow("sub4p: subtrahend above the bias")
    return [u32(a[l] + (BIAS4P[l] - b[l]), "sub4") for l in range(64)]


def pick(rows, *vals):
    """lane of row r takes vals[r]"""
    return [vals[R[l]][l] for l in range(64)]


def row_double(p):
    x, y = row_bcast01(p)
    w = add(x, y)
    t = pick(None, p, p, p, w)
    s = row_mul(t, t)
    a, b = row_bcast01(s)
    h = add(b, a)
    g = sub2p(b, a)
    e = sub4p(s, h)
    f = sub4p(add(s, s), g)
    v = row_carry(pick(None, f, f, f, e))
    ff, ee = row_bcast23(v)
    m1 = pick(None, ee, g, ff, ee)
    m2 = pick(None, ff, h, g, h)
    return row_mul(m1, m2)


def row_add_cached(p, c):
    """c rows: Y2-X2, Y2+X2, 2 Z2, 2d T2 (tight)"""
    x, y = row_bcast01(p)
    t = pick(None, sub2p(y, p), add(p, x), p, p)
    u = row_mul(t, c)                      # A, B, D, C
    u0, u1, u2, u3 = row_bcast_all(u)
    o = pick(None, u1, u0, u3, u2)
    d0 = sub2p(o, u)                       # row 0: E = B - A
    d2 = sub2p(u, o)                       # row 2: F = D - C
    sm = add(u, o)                         # row 1: H, row 3: G
    v = row_carry(pick(None, d0, sm, d2, sm))   # E, H, F, G
    v0, v1, v2, v3 = row_bcast_all(v)
    m2 = pick(None, v2, v3, v3, v1)        # F, G, G, H
    m1 = pick(None, v, v, v, v0)           # E, H, F, E
    return row_mul(m1, m2)                 # X3 = E F, Y3 = H G, Z3 = F G, T3 = E H


def row_sqn(a, n):
    for _ in range(n):
        a = row_mul(a, a)
    return a


def row_invert(z):
    t0 = row_mul(z, z)
    t1 = row_sqn(t0, 2)
    t1 = row_mul(z, t1)
    t0 = row_mul(t0, t1)
    z11 = t0
    t0 = row_mul(t0, t0)
    t0 = row_mul(t1, t0)
    t1 = row_sqn(t0, 5)
    t0 = row_mul(t1, t0)
    t1 = row_sqn(t0, 10)
    t1 = row_mul(t1, t0)
    t2 = row_sqn(t1, 20)
    t1 = row_mul(t2, t1)
    t1 = row_sqn(t1, 10)
    t0 = row_mul(t1, t0)
    t1 = row_sqn(t0, 50)
    t1 = row_mul(t1, t0)
    t2 = row_sqn(t1, 100)
    t1 = row_mul(t2, t1)
    t1 = row_sqn(t1, 50)
    t0 = row_mul(t1, t0)
    t0 = row_sqn(t0, 5)
    return row_mul(t0, z11)


# ---------------------------------------------------------------------------------------------- checks against integers
def limbs_of(x, top=False):
    return [(x >> (29 * k)) & (M29 if k < 8 else M23) for k in range(9)]


def value_of(v, r):
    return sum(v[16 * r + k] << (29 * k) for k in range(9)) % P


def rows_of(vals, limb_sets=None):
    out = [0] * 64
    for r in range(4):
        ls = limb_sets[r] if limb_sets else limbs_of(vals[r])
        for k in range(9):
            out[16 * r + k] = ls[k]
    return out


def ext_double(X, Y, Z, T):
    A, B, C = X * X % P, Y * Y % P, 2 * Z * Z % P
    H, E, G = (A + B) % P, ((X + Y) ** 2 - A - B) % P, (B - A) % P
    F = (C - G) % P     # dalek's doubling: F = G - C with E = H - (X+Y)^2; both signs flipped here gives the same point
    return E * F % P, G * H % P, F * G % P, E * H % P


def ext_add(p, q):
    X1, Y1, Z1, T1 = p
    X2, Y2, Z2, T2 = q
    A, B = (Y1 - X1) * (Y2 - X2) % P, (Y1 + X1) * (Y2 + X2) % P
    C, Dd = T1 * D2 % P * T2 % P, 2 * Z1 * Z2 % P
    E, F, G, H = (B - A) % P, (Dd - C) % P, (Dd + C) % P, (B + A) % P
    return E * F % P, G * H % P, F * G % P, E * H % P


def same_point(a, b):
    """projective equality of (X : Y : Z : T)"""
    return all((a[i] * b[2] - b[i] * a[2]) % P == 0 for i in (0, 1)) and (a[3] * b[2] - b[3] * a[2]) % P == 0


def random_point(rng):
    while True:
        y = rng.randrange(P)
        u, v = (y * y - 1) % P, (D * y * y + 1) % P
        x2 = u * pow(v, P - 2, P) % P
        x = pow(x2, (P + 3) // 8, P)
        if (x * x - x2) % P:
            x = x * pow(2, (P - 1) // 4, P) % P
        if (x * x - x2) % P:
            continue
        z = rng.randrange(1, P)
        return (x * z % P, y * z % P, z, x * y % P * z % P)


TIGHT = [2**29 + 2**18] * 8 + [2**23 + 16]


def self_check(rounds=200, seed=1):
    rng = random.Random(seed)
    # multiplication at the top of the classes the callers use: (X + Y)^2, G x H, diff x tight
    top_sum = [2 * t for t in TIGHT]
    top_diff = [t + b for t, b in zip(TIGHT, BIAS2P[:9])]
    for la, lb in ((top_sum, top_sum), (top_diff, top_sum), (top_diff, TIGHT), (TIGHT, top_diff)):
        a, b = rows_of(None, [la] * 4), rows_of(None, [lb] * 4)
        out = row_mul(a, b)
        va = sum(x << (29 * k) for k, x in enumerate(la)) % P
        vb = sum(x << (29 * k) for k, x in enumerate(lb)) % P
        for r in range(4):
            assert value_of(out, r) == va * vb % P
            assert all(out[16 * r + k] <= TIGHT[k] for k in range(9)), "not tight"
    for _ in range(rounds):
        vals = [rng.randrange(P) for _ in range(8)]
        a, b = rows_of(vals[:4]), rows_of(vals[4:])
        b = [x if K[l] < 9 else rng.randrange(2**32) for l, x in enumerate(b)]      # garbage in the idle lanes of the broadcast operand
        out = row_mul(a, b)
        for r in range(4):
            assert value_of(out, r) == vals[r] * vals[4 + r] % P
            assert all(out[16 * r + k] <= TIGHT[k] for k in range(9))
            assert all(out[16 * r + k] == 0 for k in range(9, 16))
    # chains of point operations: doublings and additions keep the class and the value
    for _ in range(max(1, rounds // 20)):
        p, q = random_point(rng), random_point(rng)
        acc = rows_of(p)
        ref = p
        cq = rows_of([(q[1] - q[0]) % P, (q[1] + q[0]) % P, 2 * q[2] % P, D2 * q[3] % P])
        for step in range(12):
            if step % 4 == 3:
                acc, ref = row_add_cached(acc, cq), ext_add(ref, q)
            else:
                acc, ref = row_double(acc), ext_double(*ref)
            got = tuple(value_of(acc, r) for r in range(4))
            assert same_point(got, ref), step
            assert got[0] * got[1] % P == got[2] * got[3] % P
            for r in range(4):
                assert all(acc[16 * r + k] <= TIGHT[k] for k in range(9))
    # the inversion chain (k_encode_invert): z^(p - 2), 0 -> 0, from the top of the tight class too
    vals = [rng.randrange(1, P), 0, 1, P - 1]
    out = row_invert(rows_of(vals))
    for r in range(4):
        assert value_of(out, r) == pow(vals[r], P - 2, P)
    top = sum(x << (29 * k) for k, x in enumerate(TIGHT)) % P
    assert value_of(row_invert(rows_of(None, [TIGHT] * 4)), 2) == pow(top, P - 2, P)
    return True


if __name__ == "__main__":
    self_check()
    print("row-limb field model: ok")
