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
