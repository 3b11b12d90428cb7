"""common.h:div_by_rcp — x / b through the correctly rounded reciprocal y = RN(1 / b) and two Markstein steps
(q = RN(x y); r = fma(-q, b, x); q = fma(r, y, q); twice) — must be the IEEE float32 quotient bit for bit: the cull and the row-sum
encoder rely on it for their cell selection (trunc(x / cell) decides the grid cell, part_base_embedder.py:115-117).  Emulated here
in exact rational arithmetic with a correct round-to-nearest-even to float32, over the cell sizes of every level of the inb_377 grids
and random operands (no GPU needed)."""
from fractions import Fraction

import numpy as np

from invr import params
from invr.config import make_cfg, PART_NAMES


def rn32(fr):
    """Nearest float32 (ties to even) of an exact Fraction, as a numpy float32."""
    if fr == 0:
        return np.float32(0.0)
    c = np.float32(float(fr))                      # within an ulp or two of the answer (double rounding): fix up among the neighbours
    cands = [c, np.nextafter(c, np.float32(np.inf)), np.nextafter(c, np.float32(-np.inf))]
    best, bd = None, None
    for v in cands:
        d = abs(Fraction(float(v)) - fr)
        even = (np.frombuffer(np.float32(v).tobytes(), np.uint32)[0] & 1) == 0
        if bd is None or d < bd or (d == bd and even):
            best, bd = v, d
    return np.float32(best)


def fma32(a, b, c):
    return rn32(Fraction(float(a)) * Fraction(float(b)) + Fraction(float(c)))


def div_by_rcp(x, b, y):
    q = rn32(Fraction(float(x)) * Fraction(float(y)))
    r = fma32(-q, b, x)
    q = fma32(r, y, q)
    r = fma32(-q, b, x)
    return fma32(r, y, q)


def test_reciprocal_form_is_the_ieee_quotient():
    cfg = make_cfg()
    cells = set()
    for name in PART_NAMES:
        sp = params.part_grid_spec(cfg, name)
        cells.update(float(np.float32(c)) for c in sp['size'][:sp['L']])            # entries_size = 1 / (res - 1) as float32 (:54)
    sp = params.deformer_grid_spec(cfg)
    cells.update(float(np.float32(c)) for c in sp['size'][:sp['L']])
    rng = np.random.default_rng(5)
    divisors = [np.float32(c) for c in sorted(cells)] + [np.float32(v) for v in np.exp(rng.uniform(np.log(1e-5), np.log(1e5), 40))]
    divisors += [np.float32(1.7934), np.float32(2.05), np.float32(0.62)]        # extents of the volumes' bounds, roughly
    n = 0
    for b in divisors:
        if (np.frombuffer(np.float32(b).tobytes(), np.uint32)[0] & 0x7fffff) == 0x7fffff:
            continue                                # Markstein's exception: the kernels fall back to the hardware division
        y = rn32(Fraction(1) / Fraction(float(b)))
        xs = np.concatenate([rng.uniform(-3, 3, 120), np.exp(rng.uniform(np.log(1e-17), np.log(9e5), 60)) * rng.choice([-1, 1], 60),
                             # quotients within an ulp of an integer: where a wrong last bit would flip trunc()
                             [float(np.nextafter(np.float32(k * float(b)), np.float32(s))) for k in (1, 2, 7, 100, 1023) for s in (-np.inf, np.inf)]])
        for x in xs.astype(np.float32):
            if not (8.7e-19 < abs(float(x)) < 1.0e6):
                continue                            # outside div_exact's range: hardware division
            want = rn32(Fraction(float(x)) / Fraction(float(b)))
            got = div_by_rcp(x, b, y)
            assert got == want and np.signbit(got) == np.signbit(want), (float(x), float(b), float(got), float(want))
            n += 1
    assert n > 12000
