"""The wave reductions of csrc/dsq_wave.h run the xor butterfly (partners 32, 16, 8, 4, 2, 1) with
v_permlane32_swap / v_permlane16_swap and DPP row rotations instead of shuffles.  This lane-level model checks
the claim the kernels rely on: for a commutative operation the result is bit-identical to
`v = op(v, shfl_xor(v, m))`, whichever way `row_ror:4` rotates (i + 4 or i - 4 inside a 16-lane row), because
after the xor-8 step the values have period 8 inside a row."""
import numpy as np
import pytest

W = 64


def xor_butterfly(v, op):
    v = list(v)
    for m in (32, 16, 8, 4, 2, 1):
        v = [op(v[i], v[i ^ m]) for i in range(W)]
    return v


def swap_rows(v, width):
    """v_permlane{32,16}_swap with both operands = v: (a, c) = (value of the lower block, of the upper block)
    for every pair of neighbouring blocks of `width` lanes."""
    a, c = [None] * W, [None] * W
    for i in range(W):
        lower = (i // (2 * width)) * 2 * width + i % width
        a[i], c[i] = v[lower], v[lower + width]
    return a, c


def row_ror(v, k, direction):
    return [v[(i // 16) * 16 + (i % 16 + direction * k) % 16] for i in range(W)]


def quad_perm(v, perm):
    return [v[(i // 4) * 4 + perm[i % 4]] for i in range(W)]


def valu_butterfly(v, op, direction):
    v = list(v)
    for width in (32, 16):
        a, c = swap_rows(v, width)
        v = [op(a[i], c[i]) for i in range(W)]
    for k in (8, 4):
        o = row_ror(v, k, direction)
        v = [op(v[i], o[i]) for i in range(W)]
    for perm in ((2, 3, 0, 1), (1, 0, 3, 2)):
        o = quad_perm(v, perm)
        v = [op(v[i], o[i]) for i in range(W)]
    return v


@pytest.mark.parametrize("direction", [+1, -1])
@pytest.mark.parametrize("seed", range(5))
def test_sum_butterfly_bits(direction, seed):
    rng = np.random.default_rng(seed)
    # wide dynamic range: any change of partner or order would show up in the rounding
    x = (rng.normal(size=W) * 10.0 ** rng.integers(-8, 8, W)).tolist()
    ref = xor_butterfly(x, lambda p, q: p + q)
    got = valu_butterfly(x, lambda p, q: p + q, direction)
    assert all(np.float64(r).tobytes() == np.float64(g).tobytes() for r, g in zip(ref, got))
    assert len({np.float64(g).tobytes() for g in got}) == 1  # every lane holds the same total


@pytest.mark.parametrize("direction", [+1, -1])
def test_int_max_butterfly(direction):
    rng = np.random.default_rng(7)
    x = rng.integers(-1000, 1000, W).tolist()
    assert valu_butterfly(x, max, direction) == xor_butterfly(x, max) == [max(x)] * W
