"""The numeric fact the guarded slab test rests on (csrc/mr_traverse.h, node_slabs_guarded): for regular
operands the product RN(a * RN(1/d)) and the reference's quotient RN(a / d) are fewer than 4 bit patterns apart, have the same
sign and are zero together -- so two slab distances more than 16 patterns apart compare the same way in both forms.  numpy's
float32 multiply and divide are the IEEE operations the device executes (v_mul_f32, and the division the oracle restates)."""
import numpy as np


def _regular(rng, n, lo_exp, hi_exp):
    mant = rng.integers(0, 1 << 23, n, dtype=np.uint32)
    exp = rng.integers(lo_exp + 127, hi_exp + 127, n, dtype=np.uint32)
    sign = rng.integers(0, 2, n, dtype=np.uint32)
    return ((sign << 31) | (exp << 23) | mant).view(np.float32)


def _patterns_apart(x, y):
    xi, yi = x.view(np.uint32).astype(np.int64), y.view(np.uint32).astype(np.int64)
    return np.abs(xi - yi)


def test_product_and_quotient_are_fewer_than_four_patterns_apart():
    rng = np.random.default_rng(4)
    worst = 0
    for _ in range(8):
        n = 1 << 20
        a = _regular(rng, n, -59, 61)           # corner - o of a regular ray in a regular node (lane_is_regular, mr_api.cpp)
        d = _regular(rng, n, -40, 40)           # direction component
        # mantissas at the ends of the binade, where an ulp changes size
        a[: n // 16] = (a[: n // 16].view(np.uint32) | np.uint32(0x7FFFF0)).view(np.float32)
        d[n // 16: n // 8] = (d[n // 16: n // 8].view(np.uint32) & np.uint32(0xFF80000F)).view(np.float32)
        inv = np.float32(1.0) / d
        prod = a * inv
        quot = a / d
        assert np.array_equal(np.signbit(prod), np.signbit(quot))
        assert not (prod == 0).any() and not (quot == 0).any()
        worst = max(worst, int(_patterns_apart(prod, quot).max()))
    assert worst <= 3, worst


def test_a_zero_numerator_gives_the_same_signed_zero():
    d = np.array([0.5, -0.5, 3.0, -7.25], np.float32)
    for a in (np.float32(0.0), np.float32(-0.0)):
        prod, quot = a * (np.float32(1.0) / d), a / d
        assert np.array_equal(prod.view(np.uint32), quot.view(np.uint32))


def test_values_more_than_sixteen_patterns_apart_keep_their_order():
    """the guard's threshold with its margin: move both values by up to 3 patterns either way, the comparison cannot flip"""
    rng = np.random.default_rng(5)
    x = _regular(rng, 1 << 18, -30, 30)
    xi = x.view(np.uint32)
    for gap in (17, 18, 40):
        y = (xi + np.uint32(gap)).view(np.float32)                    # same sign, `gap` patterns further from zero
        for dx in (-3, 0, 3):
            for dy in (-3, 0, 3):
                x2 = (xi.astype(np.int64) + dx).astype(np.uint32).view(np.float32)
                y2 = (y.view(np.uint32).astype(np.int64) + dy).astype(np.uint32).view(np.float32)
                assert np.array_equal(x < y, x2 < y2) and np.array_equal(x > y, x2 > y2)
