"""The error budget of the matrix-core filter (csrc/icp_fused_kernels.hip, terms (a)-(h)), checked as arithmetic on the CPU.

The GPU tests check the kernel's lower bound L against exact run minima (tests/test_gpu_filter_bounds.py).  This file checks the
ANALYSIS the band E rests on, without a GPU: it rebuilds the surrogate s of a (source, target) pair exactly as the kernel and the
operand image form it - float32 centring, two bfloat16 pieces per coordinate, the nine products the contraction keeps (the three
low x low products are left out since round 4), tt' in three pieces, the rounded-up bias, the two float32 additions - with the
contraction itself summed EXACTLY (float64 holds every term), and asserts, pair by pair,

    s + A  <=  (1 + 2^-10) d^2 + E        A = 65 u N: the worst-case model of the matrix unit's own accumulation (term (g)),
                                          which tools/probe_mfma_bf16.hip measures on the hardware (tests/test_gpu_mfma_probe.py)

i.e. L = max((s - E) * 0.99902, 0) <= d^2 for the contract's float32 d^2, on random geometry and on the lattices where each term
is largest.  It also reports how much of the budget (353 u N) the worst pair used."""
import numpy as np

U = 2.0 ** -24


def bf16(x):
    """float32 -> bfloat16 (round to nearest even), returned as float32 values."""
    x = np.asarray(x, dtype=np.float32)
    u = x.view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16).astype(np.uint32) << 16
    return r.astype(np.uint32).view(np.float32)


def bf16_up(x):
    """float32 (positive) -> the bfloat16 at or above it, as the kernel rounds the bias: (bits + 0xFFFF) >> 16."""
    u = np.asarray(x, dtype=np.float32).view(np.uint32).astype(np.uint64)
    return (((u + 0xFFFF) >> 16).astype(np.uint32) << 16).view(np.float32)


def split(v):
    h = bf16(v)
    l = bf16((v - h).astype(np.float32))
    return h, l


def surrogate(q, t, c, r2):
    """s, E, N and the contract's d^2 for pairs (q[i], t[i]) around the group centre c with the group's r^2."""
    f = np.float32
    qc = (q - c).astype(f)                                     # fl(q - c), fl(t - c): the float32 subtractions of the bound
    tc = (t - c).astype(f)
    qp = (f(-2.0) * qc).astype(f)                              # q' = -2 qc, exact
    qh, ql = split(qp)
    th, tl = split(tc)
    v = (qh + ql).astype(f)                                    # the kernel's `value = hf + lf` (float32)
    vv = v.astype(np.float64)
    # qq = 0.25 * fma(vz, vz, fma(vy, vy, vx * vx)) in float32: each step rounded once
    a = (vv[:, 0] * vv[:, 0]).astype(f).astype(np.float64)
    a = (vv[:, 1] * vv[:, 1] + a).astype(f).astype(np.float64)
    qq = (f(0.25) * (vv[:, 2] * vv[:, 2] + a).astype(f)).astype(f)
    ee = (np.float64(2.6702880859375e-05) * (qq.astype(np.float64) + np.float64(r2)) + 7.5231638e-37).astype(f)   # one fma
    bias = bf16_up((2.0 * ee.astype(np.float64) + qq.astype(np.float64)).astype(f))
    adj = (qq - bias).astype(f)
    tq = th.astype(np.float64) + tl.astype(np.float64)         # the quantised target, |tc'|^2 from double as xtarget_kernels does
    ttq = (tq * tq).sum(axis=1).astype(f)
    t1 = bf16(ttq)
    rem = (ttq - t1).astype(f)
    t2 = bf16(rem)
    t3 = bf16((rem - t2).astype(f))
    D = np.float64
    kept = (th.astype(D) * qh.astype(D) + th.astype(D) * ql.astype(D) + tl.astype(D) * qh.astype(D)).sum(axis=1)   # slots 0-2, 4-6, 8-10
    m = kept + t1.astype(D) + t2.astype(D) + t3.astype(D) + bias.astype(D)                                          # exact contraction
    s = (m.astype(f) + adj).astype(f)                          # fl(MFMA + fl(qq - b)); the MFMA result itself is a float32
    n_ = (qc.astype(D) ** 2).sum(axis=1) + D(r2)
    # the contract's squared distance: every product and sum rounded to float32 (oracle dist2)
    dx, dy, dz = [(q[:, k] - t[:, k]).astype(f) for k in range(3)]
    d2 = ((dx * dx).astype(f) + (dy * dy).astype(f)).astype(f)
    d2 = (d2 + (dz * dz).astype(f)).astype(f)
    return s.astype(D), ee.astype(D), n_, d2.astype(D), m


def check(q, t, c, r2=None, label=""):
    q = np.ascontiguousarray(q, dtype=np.float32)
    t = np.ascontiguousarray(t, dtype=np.float32)
    c = np.asarray(c, dtype=np.float32)
    tc = (t - c).astype(np.float32).astype(np.float64)
    th, tl = split((t - c).astype(np.float32))
    tq = th.astype(np.float64) + tl.astype(np.float64)
    need = max((tc ** 2).sum(axis=1).max(), (tq ** 2).sum(axis=1).max())
    if r2 is None:
        r2 = float(np.nextafter(np.float32(need), np.float32(np.inf)))     # the image's r^2: the next float up
    assert r2 >= need
    s, ee, n_, d2, m = surrogate(q, t, c, r2)
    assert (m > 0).all(), "the bias must keep every contraction result positive (unsigned minima)"
    acc = 65.0 * U * n_                                         # term (g), worst-case model
    excess = (s + acc) - ((1.0 + 2.0 ** -10) * d2 + ee)
    assert (excess <= 0).all(), f"{label}: {int((excess > 0).sum())} pairs break s <= (1 + 2^-10) d^2 + E; worst {excess.max():.3e}"
    lower = np.maximum((s + acc - ee) * 0.99902, 0.0)
    assert (lower <= d2).all(), label
    used = ((s - (1.0 + 2.0 ** -10) * d2) / (U * n_)).max()    # what the pair consumed of the budget, in u N (without (g))
    return used


def test_budget_on_random_geometry():
    rng = np.random.default_rng(2026)
    worst = 0.0
    for scale in (1e-3, 1.0, 40.0):
        c = (rng.standard_normal(3) * scale).astype(np.float32)
        t = (c + rng.standard_normal((200_000, 3)) * 0.05 * scale).astype(np.float32)      # a compact group of tiles
        q = (c + rng.standard_normal((200_000, 3)) * rng.choice([0.02, 0.05, 1.0, 30.0], (200_000, 1)) * scale).astype(np.float32)
        worst = max(worst, check(q, t, c, label=f"random, scale {scale}"))
        near = (t + rng.standard_normal(t.shape) * 1e-4 * scale).astype(np.float32)        # the pairs a filter exists for
        worst = max(worst, check(near, t, c, label=f"near pairs, scale {scale}"))
    assert worst < 353.1, worst                                  # the derived worst case without (g)'s 65: (a)-(f) + (h) < 288.1
    print(f"largest use of the budget on random geometry: {worst:.1f} u N of 448")


def test_budget_where_the_dropped_products_are_largest():
    """Both points on the lattice 2^e (1 + a/256), a = 1 (mod 4): every coordinate half-way between two bfloat16 values (low piece a
    full half ulp), all low pieces of the targets positive and all of q' = -2 q negative, so the three dropped products add up; four
    values per axis just above the power of two and r^2 = the points' own radius, so that N is as small as the term allows."""
    a = 1.0 + (4.0 * np.arange(4) + 1.0) / 256.0               # just above the power of two: there the half ulp is largest relative to the value
    g = np.stack(np.meshgrid(a, a, a, indexing="ij"), axis=-1).reshape(-1, 3)
    rng = np.random.default_rng(7)
    worst = 0.0
    for e in (-12, 0, 9):
        pts = (g * 2.0 ** e).astype(np.float32)
        t = pts[rng.integers(0, len(pts), 300_000)]
        q = pts[rng.integers(0, len(pts), 300_000)]
        worst = max(worst, check(q, t, np.zeros(3), label=f"lattice 2^{e}"))
        worst = max(worst, check(t, t, np.zeros(3), label=f"lattice 2^{e}, the point itself"))
    assert 200.0 < worst < 353.1, worst                          # the lattice really reaches the (h) term's size (256 u N), not beyond the budget
    print(f"largest use of the budget on the low x low lattice: {worst:.1f} u N of 448")


def test_budget_at_the_edges_of_float32():
    rng = np.random.default_rng(3)
    for scale in (1e-18, 1e15):
        c = np.zeros(3, dtype=np.float32)
        t = (rng.standard_normal((100_000, 3)) * scale).astype(np.float32)
        q = (t[rng.permutation(len(t))] * np.float32(1.001)).astype(np.float32)
        check(q, t, c, label=f"scale {scale}")
