"""CPU checks of oracle/lowprec.py: the e4m3 / f16 operand formats of the "high" precision forward and the corrected product they
build (the GPU tests compare the kernels' images with torch.float8_e4m3fn; here the numpy restatement is pinned on the same type and
the arithmetic claim - corrections of both operands bring the f16 product to ~2^-16 - is checked without a GPU)."""
import numpy as np
import torch

from oracle import lowprec as lp


def test_e4m3_round_matches_torch_float8_e4m3fn_on_every_code_and_on_ties():
    codes = torch.arange(256, dtype=torch.uint8)
    vals = codes.view(torch.float8_e4m3fn).float().numpy().astype(np.float64)
    finite = np.isfinite(vals)
    assert finite.sum() == 254                                    # 0x7f / 0xff are NaN, no infinities
    assert np.array_equal(lp.e4m3_round(vals[finite]), vals[finite])        # every representable value is a fixed point
    assert np.max(np.abs(vals[finite])) == 448.0
    rng = np.random.default_rng(0)
    a = np.concatenate([rng.standard_normal(20000) * 10.0 ** rng.uniform(-4, 2.6, 20000),
                        (vals[finite][:-1] + vals[finite][1:]) / 2.0,           # exact ties between neighbours (and across zero)
                        [0.0, 1e-9, -1e-9, 2.0 ** -10, 2.0 ** -9, 3 * 2.0 ** -10, 447.9, 448.0]])
    a = a[np.abs(a) <= 448.0].astype(np.float32).astype(np.float64)
    want = torch.from_numpy(a.astype(np.float32)).to(torch.float8_e4m3fn).float().numpy().astype(np.float64)
    assert np.array_equal(lp.e4m3_round(a), want)
    # beyond 448 the restatement (like the kernels) saturates; the bare format conversion does not promise that
    assert np.array_equal(lp.e4m3_round(np.array([449.0, 480.0, 1e9, -1e9])), np.array([448.0, 448.0, 448.0, -448.0]))


def test_f16_round_matches_torch_half():
    rng = np.random.default_rng(1)
    a = (rng.standard_normal(50000) * 10.0 ** rng.uniform(-9, 4, 50000)).astype(np.float32)
    assert np.array_equal(lp.f16_round(a), torch.from_numpy(a).half().double().numpy())


def test_corrected_product_closes_the_f16_operand_rounding():
    """MoE-head-like contraction (|x| ~ 1.3, |w| ~ 0.04, K = 1024) with the product's scale set (6, 17, 18, 7): the plain f16 product sits
    ~2e-3 from the float64 one, with the weights' correction alone the x rounding is left (~1.5e-3), with both ~1e-4 or better - the
    numbers the GPU kernels are then held to (tests/test_gpu_kernels.py)."""
    rng = np.random.default_rng(2)
    M, N, K = 48, 64, 1024
    x = (rng.standard_normal((M, K)) * 1.3).astype(np.float32)
    w = (rng.standard_normal((N, K)) * 0.04).astype(np.float32)
    ref = x.astype(np.float64) @ w.astype(np.float64).T
    plain = lp.f16_round(x) @ lp.f16_round(w).T
    both = lp.corrected_product(x, w, 6, 17, 18, 7)
    w_only = lp.corrected_product(x, w, 6, 17, 18, 7, correct_x=False)
    e_plain, e_w, e_both = (np.max(np.abs(v - ref)) for v in (plain, w_only, both))
    assert e_both < 1e-4 and e_both * 10 < e_plain and e_w < e_plain and e_both < e_w, (e_plain, e_w, e_both)
    # weights-only form on an LSTM-like operand (|h| <= 1, scales 7 / 17): the weight term drops by >= 10x
    h = np.tanh(rng.standard_normal((M, K))).astype(np.float32)
    wl = (rng.standard_normal((N, K)) * 0.05).astype(np.float32)
    exact_h = lp.f16_round(h) @ wl.astype(np.float64).T            # activations f16-rounded on both sides: isolate the weight term
    e16 = np.max(np.abs(lp.f16_round(h) @ lp.f16_round(wl).T - exact_h))
    e8 = np.max(np.abs(lp.corrected_product(h, wl, 7, 17, 17, 7, correct_x=False) - exact_h))
    assert e8 * 10 < e16, (e16, e8)


# OCP Microscaling Formats (MX) v1.0, table "FP6 E2M3": bias 1, no Inf / NaN, max normal 7.5, min normal 1.0, max subnormal 0.875, min subnormal 0.125
E2M3_TABLE = [0.0, 0.125, 0.25, 0.375, 0.5, 0.625, 0.75, 0.875,           # exponent field 0: subnormals m / 8
              1.0, 1.125, 1.25, 1.375, 1.5, 1.625, 1.75, 1.875,           # 1: 2^0 (1 + m / 8)
              2.0, 2.25, 2.5, 2.75, 3.0, 3.25, 3.5, 3.75,                 # 2: 2^1
              4.0, 4.5, 5.0, 5.5, 6.0, 6.5, 7.0, 7.5]                     # 3: 2^2


def test_e2m3_codes_match_the_format_table_bit_for_bit():
    """All 64 codes of OCP FP6 e2m3 against the written-out value table (sign bit 5, exponent bits 4-3, mantissa bits 2-0), every value a
    fixed point of the rounding, ties to even between every pair of neighbours, saturation at 7.5."""
    codes = np.arange(64)
    want = np.array(E2M3_TABLE + [-v for v in E2M3_TABLE])
    assert np.array_equal(lp.e2m3_decode(codes), want)
    nz = codes != 32                                                          # -0 is written as +0
    assert np.array_equal(lp.e2m3_encode(want[nz]), codes[nz]) and lp.e2m3_encode(-0.0) == 0
    t = np.array(E2M3_TABLE)
    mid = (t[:-1] + t[1:]) / 2.0
    even = np.where(np.arange(31) % 2 == 0, t[:-1], t[1:])                    # the neighbour with an even code
    assert np.array_equal(lp.e2m3_round(mid), even) and np.array_equal(lp.e2m3_round(-mid), -even)
    assert np.array_equal(lp.e2m3_round(np.nextafter(mid, 0)), t[:-1]) and np.array_equal(lp.e2m3_round(np.nextafter(mid, 9)), t[1:])
    assert np.array_equal(lp.e2m3_round([7.74, 7.76, 8.0, 1e9, -1e9]), [7.5, 7.5, 7.5, 7.5, -7.5])
    rng = np.random.default_rng(3)
    a = rng.standard_normal(20000) * 4.0
    r = lp.e2m3_round(a)
    near = t[np.argmin(np.abs(np.minimum(np.abs(a), 7.5)[:, None] - t[None, :]), axis=1)] * np.sign(a)
    assert np.array_equal(np.abs(r - a) <= np.abs(near - a) + 1e-15, np.ones_like(a, bool))


def test_fp6_packing_is_32_codes_in_24_bytes_element_i_at_bit_6i():
    c = np.zeros(32, np.int64)
    c[0], c[1], c[5], c[31] = 0b000001, 0b100000, 0b111111, 0b101010
    b = lp.pack_fp6(c)
    assert b.shape == (24,)
    bits = int.from_bytes(bytes(b.tolist()), "little")
    assert bits == (1 << 0) | (0b100000 << 6) | (0b111111 << 30) | (0b101010 << 186)
    rng = np.random.default_rng(4)
    codes = rng.integers(0, 64, (5, 7, 128))
    assert np.array_equal(lp.unpack_fp6(lp.pack_fp6(codes)), codes) and lp.pack_fp6(codes).shape == (5, 7, 96)


def test_row_scales_and_the_e2m3_corrected_product():
    """Per-row power-of-two scales put every row maximum into [4, 8); on the operands of the L1 level (|h| <= 1 against weights of
    sigma 0.05, K = 1024) the e2m3 low-order image removes >= 10x of the f16 weight-rounding term - the 2^-12-sized correction needs ~4
    bits, which is why FP6 is enough where e4m3 was used in round 3 - and with both corrections the product is within 2e-4 of float64."""
    m = np.array([5.0, 0.03, 0.0, 8.0, 7.99, 4.0, 3.99, 1e-6])
    s = lp.row_scale_exp(m)
    scaled = m * 2.0 ** s
    assert np.all((scaled[m > 0] >= 4.0) & (scaled[m > 0] < 8.0)) and s[2] == 0
    rng = np.random.default_rng(5)
    M, N, K = 48, 64, 1024
    h = np.tanh(rng.standard_normal((M, K))).astype(np.float32)
    w = (rng.standard_normal((N, K)) * 0.05).astype(np.float32)
    exact_h = lp.f16_round(h) @ w.astype(np.float64).T
    e16 = np.max(np.abs(lp.f16_round(h) @ lp.f16_round(w).T - exact_h))
    e6 = np.max(np.abs(lp.corrected_product6(h, w, correct_x=False) - exact_h))
    e8 = np.max(np.abs(lp.corrected_product(h, w, 7, 17, 17, 7, correct_x=False) - exact_h))
    assert e6 * 10 < e16 and e8 <= e6, (e16, e6, e8)
    x = (rng.standard_normal((M, K)) * 1.3).astype(np.float32)
    ref = x.astype(np.float64) @ w.astype(np.float64).T
    e_plain = np.max(np.abs(lp.f16_round(x) @ lp.f16_round(w).T - ref))
    e_both = np.max(np.abs(lp.corrected_product6(x, w) - ref))
    assert e_both < 2e-4 and e_both * 8 < e_plain, (e_plain, e_both)


def test_time_dithered_f16_images_round_to_the_two_neighbours_with_bounded_discrepancy():
    """oracle/lowprec.py::f16_dither_images (the restatement of evc_cast_f32_to_f16_dither): every image element is one of the value's two f16
    neighbours; f16 values (zero, ones, subnormal steps) are reproduced in every image; over ANY run of consecutive images the number of
    round-ups is within 3 of (run length) x frac (2.03 measured at 20 steps) - the property that makes a recurrence's weight-rounding errors cancel over its steps - so
    the mean of T images is far closer to w than the round-to-nearest image; images are a pure function of (w, seed) and differ between seeds."""
    rng = np.random.default_rng(5)
    w = (rng.standard_normal((96, 160)) * np.logspace(-7, 1, 160)[None, :]).astype(np.float32)
    w[0, :8] = [0.0, 1.0, -1.0, 2.0 ** -24, -2.0 ** -24, 3e-8, -3e-8, 65000.0]
    T = 20
    im = lp.f16_dither_images(w, T, seed=7)
    assert im.dtype == np.float16 and im.shape == (T,) + w.shape
    dn, up = lp.f16_neighbours(w)
    assert np.all(dn.astype(np.float64) <= w) and np.all(w <= up.astype(np.float64))
    adjacent = (up == dn) | (np.nextafter(dn, np.float16(np.inf)) == up)
    assert adjacent.all()
    assert np.all((im == dn[None]) | (im == up[None]))
    exact = w.astype(np.float16).astype(np.float32) == w
    assert exact[0, :5].all() and np.all(im[:, exact] == w[exact].astype(np.float16)[None])
    gap = up.astype(np.float64) - dn.astype(np.float64)
    frac = np.where(gap > 0, (w.astype(np.float64) - dn.astype(np.float64)) / np.where(gap > 0, gap, 1.0), 0.0)
    ups = (im == up[None]) & (gap > 0)[None]
    worst = 0.0
    for a in range(T):
        c = np.cumsum(ups[a:], axis=0)
        n = np.arange(1, T - a + 1).reshape((-1,) + (1,) * w.ndim)
        worst = max(worst, float(np.abs(c - n * frac[None]).max()))
    assert worst < 3.0, worst
    err_mean = np.abs(im.astype(np.float64).mean(0) - w)[~exact]
    err_rtn = np.abs(lp.f16_round(w) - w)[~exact]
    assert err_mean.mean() < 0.2 * err_rtn.mean()
    assert np.array_equal(im, lp.f16_dither_images(w, T, seed=7)) and not np.array_equal(im, lp.f16_dither_images(w, T, seed=8))
    assert np.array_equal(im[:5], lp.f16_dither_images(w, 5, seed=7))               # image t does not depend on T
