"""DbofModel and FrameLevelLogisticModel towers (through the C ABI) against the
float64 oracle: forward, gradients, BN moving averages.  pytest -m gpu."""
import numpy as np
import pytest
import torch

from oracle import model_math as mm

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _np(t):
    return t.detach().cpu().double().numpy()


def _params(tower):
    pre = tower.scope + "/"
    return {k[len(pre):]: _np(v) for k, v in tower.state_dict().items()}


def _rel(a, b):
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))


@pytest.mark.parametrize("B,F,C,Hd,V,S,tol", [(6, 64, 128, 64, 40, 8, 8e-3), (9, 128, 256, 64, 33, 30, 8e-3),
                                                 (64, 1152, 8192, 1024, 4716, 30, 5e-3)])
def test_dbof_forward_backward(B, F, C, Hd, V, S, tol):
    """DBoF feeds O(1) batch-normalised activations and O(1/sqrt(K)) weights to the bf16 GEMMs,
    so the 2^-9 operand rounding shows up at ~2e-3 in the probabilities (measured 2.3e-3 at the
    BASELINE cfg-4 dims).  This is ABOVE north_star's 1e-3: DESIGN.md lists it as an open gap
    (split-bf16 parity mode).  The tolerance here pins the current behaviour."""
    from efficientvideoclassification_youtube8m_amd.towers import DbofTower
    rng = np.random.default_rng(B + C)
    q, x, n, labels = mm.synthetic_batch(B, seed=B, feature_size=F, vocab_size=V, dtype=np.float32)
    tw = DbofTower(B, 300, F, V, iterations=S, cluster_size=C, hidden_size=Hd, device=DEV, seed=3)
    # non-trivial BN affine parameters
    for k in tw.names:
        if k.endswith("/gamma") or k.endswith("/beta"):
            tw.store.p(k).add_(torch.from_numpy(rng.standard_normal(tw.store.p(k).shape).astype(np.float32) * 0.2).to(DEV))
    P = _params(tw)
    u = rng.random((B, S)).astype(np.float32)
    pred = tw.forward(torch.from_numpy(x).to(DEV), torch.from_numpy(n).to(DEV), torch.from_numpy(u).to(DEV))
    xn = mm.l2_normalize(x.astype(np.float64), 2)
    ref_pred, cache = mm.dbof_fwd(xn, n, u, P)
    assert np.array_equal(tw.idx.cpu().numpy(), mm.sample_random_frames_index(u, n))     # int32 truncation, bit-exact
    err = np.abs(_np(pred) - ref_pred).max()
    print('dbof pred err %.2e' % err)
    assert err < tol
    dp = mm.cross_entropy_grad(ref_pred, labels)
    tw.backward(torch.from_numpy(dp.astype(np.float32)).to(DEV))
    gref = mm.dbof_bwd(dp, cache)
    for k, g in gref.items():
        got = tw.store.g(k)
        got = _np(got.t() if got.dim() == 2 else got)
        # relu6 / max-pool masks can flip on values within bf16 rounding of 0, 6 or a tie, which moves
        # single entries by a whole dy; judge each tensor by its relative L2 error (some are
        # analytically zero: cluster_bn/beta is cancelled by hidden1_bn's mean subtraction).
        l2 = float(np.linalg.norm(got - g) / (np.linalg.norm(g) + 1e-30))
        # (1% flipped relu6 masks = 10% relative L2; the kernels themselves are checked to 1e-4 in
        #  test_gpu_kernels.py::test_batchnorm_relu6_pool_kernels)
        assert l2 < 0.3 or np.abs(got - g).max() < 2e-3, (k, l2, np.abs(got - g).max())
    # moving averages: moving -= (1-0.999)*(moving - batch)
    mu = cache[3][3]
    assert np.allclose(_np(tw.buffers["input_bn/moving_mean"]), 0.001 * mu, rtol=1e-3, atol=1e-7)
    var = cache[5][4]
    assert np.allclose(_np(tw.buffers["cluster_bn/moving_variance"]), 1 - 0.001 * (1 - var), rtol=1e-3, atol=1e-6)
    # "high" precision forward (split-bf16 operands): the north-star 1e-3 holds with O(1) activations
    tw.set_precision("high")
    pred_h = tw.forward(torch.from_numpy(x).to(DEV), torch.from_numpy(n).to(DEV), torch.from_numpy(u).to(DEV))
    err_h = np.abs(_np(pred_h) - ref_pred).max()
    print('dbof pred err (high precision) %.2e' % err_h)
    assert err_h < (1e-3 if B >= 64 else 3e-3)
    tw.set_precision("bf16")
    # eval mode uses the moving statistics
    p_eval = tw.forward(torch.from_numpy(x).to(DEV), torch.from_numpy(n).to(DEV), torch.from_numpy(u).to(DEV),
                        is_training=False)
    assert torch.isfinite(p_eval).all()


def test_logistic_forward_backward_and_step():
    from efficientvideoclassification_youtube8m_amd.towers import LogisticTower
    from efficientvideoclassification_youtube8m_amd.distill import SingleTowerGraph
    B, F, V = 32, 1152, 4716                                          # BASELINE config 1 shape
    q, x, n, labels = mm.synthetic_batch(B, seed=2, dtype=np.float32)
    tw = LogisticTower(B, 300, F, V, device=DEV, seed=1)
    tw.store.p(tw.Bn).normal_(0, 0.1)
    P = _params(tw)
    g = SingleTowerGraph(tw)
    out = g.step(torch.from_numpy(x).to(DEV), torch.from_numpy(labels.astype(np.uint8)).to(DEV),
                 torch.from_numpy(n).to(DEV), apply=False)
    xn = mm.l2_normalize(x.astype(np.float64), 2)
    p_ref, avg = mm.logistic_fwd(xn, n, P["fully_connected/weights"], P["fully_connected/biases"])
    assert np.abs(_np(out["predictions"]) - p_ref).max() < 1e-3
    assert abs(out["loss"].item() - mm.cross_entropy_loss(p_ref, labels)) / mm.cross_entropy_loss(p_ref, labels) < 1e-4
    dW, db = mm.logistic_bwd(mm.cross_entropy_grad(p_ref, labels), p_ref, avg)
    assert _rel(_np(tw.store.g(tw.W).t()), dW) < 2e-2
    assert _rel(_np(tw.store.g(tw.Bn)), db) < 2e-2
