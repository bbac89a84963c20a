"""k_pitch walks only the coarse lags whose ratio num / den lies within 2^-12 of the second-largest ratio instead of all 147
(nnn_kernels.hip, "find_best_pitch over the coarse lags").  The claim that this leaves the reference's result untouched
(src/pitch.rs:372-405: order-dependent f32 comparisons num_a den_b > num_b den_a) is restated here in numpy and tried on
inputs built to sit on the edge: ratios that differ in the last bits, exact ties, ratios spread evenly through the band, few
or no positive correlations, tiny and huge magnitudes."""
import numpy as np

f32 = np.float32
NLAG = 147


def walk(num, den, order):
    """The reference's running (best, second) over the lags in `order` (ascending indices)."""
    bn, sn, bd, sd, b, s = f32(-1), f32(-1), f32(0), f32(0), 0, 1
    for i in order:
        n, e = num[i], den[i]
        if not n == n:            # NaN: correlation not positive
            continue
        if f32(n * sd) > f32(sn * e):
            if f32(n * bd) > f32(bn * e):
                sn, sd, s = bn, bd, b
                bn, bd, b = n, e, i
            else:
                sn, sd, s = n, e, i
    return b, s


def banded(num, den):
    with np.errstate(all="ignore"):
        r = (num * (f32(1) / den)).astype(f32)
    ok = r >= 0
    rr = np.where(ok, r, f32(-1))
    top = np.sort(rr)[::-1]
    m2 = top[1]
    nmax = num[ok].max() if ok.any() else f32(0)
    dmax = den[ok].max() if ok.any() else f32(1)
    with np.errstate(all="ignore"):
        tame = m2 > f32(1e-30) and f32(nmax * dmax) < f32(1e37)
    thr = f32(m2 * f32(1 - 1 / 4096)) if tame else f32(0)
    marked = np.nonzero(rr >= thr)[0]
    return walk(num, den, marked), len(marked)


def cases(rng):
    for _ in range(300):                                   # plain: correlations of either sign, energies >= 1
        c = rng.standard_normal(NLAG).astype(f32) * f32(10 ** rng.uniform(-3, 10))
        den = np.maximum(f32(1), (rng.random(NLAG) * 10 ** rng.uniform(0, 11)).astype(f32))
        yield np.where(c > 0, c * c, f32(np.nan)).astype(f32), den
    for _ in range(300):                                   # ratios a few ulp apart, many exact ties
        den = np.maximum(f32(1), (rng.random(NLAG) * 1e6).astype(f32))
        base = f32(rng.uniform(1e-3, 1e6))
        k = rng.integers(-3, 4, NLAG)
        num = (den * base * (1 + k * 6e-8)).astype(f32)
        num[rng.random(NLAG) < 0.3] = np.nan
        yield num, den
    for _ in range(300):                                   # ratios spread through and around the band
        den = np.maximum(f32(1), (rng.random(NLAG) * 1e4).astype(f32))
        base = f32(rng.uniform(1, 1e5))
        num = (den * base * (1 - rng.random(NLAG) * 10 ** rng.uniform(-7, -2))).astype(f32)
        yield num, den
    for npos in (0, 1, 2, 3):                              # hardly any positive correlation
        for _ in range(20):
            num = np.full(NLAG, np.nan, f32)
            idx = rng.choice(NLAG, npos, replace=False)
            num[idx] = (rng.random(npos) * 1e5).astype(f32)
            yield num, np.maximum(f32(1), (rng.random(NLAG) * 1e5).astype(f32))
    for scale in (1e-38, 1e-30, 1e-20, 1e20, 1e30, 3e38):   # products that underflow or overflow: the full walk
        for _ in range(20):
            num = (rng.random(NLAG) * scale).astype(f32)
            yield num, np.maximum(f32(1), (rng.random(NLAG) * 1e8).astype(f32))


def test_banded_walk_equals_the_full_walk():
    rng = np.random.default_rng(5)
    n, steps = 0, 0
    with np.errstate(all="ignore"):
        for num, den in cases(rng):
            want = walk(num, den, range(NLAG))
            got, m = banded(num, den)
            assert got == want, (n, want, got)
            n += 1
            steps += m
    assert n > 1000
    print(f"{n} cases, {steps / n:.1f} of {NLAG} lags walked on average")
