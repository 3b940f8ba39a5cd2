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
    """DBoF feeds O(1) batch-normalised activations and O(1/sqrt(K)) weights to the bf16 GEMMs, so the 2^-9 operand rounding
    shows up at ~2e-3 in the probabilities (measured 2.3e-3 at the BASELINE cfg-4 dims) - above north_star's 1e-3; the
    "high" forward checked at the end of this test closes that gap (split-bf16: 5.6e-6 at cfg-4 dims; f16 + e4m3 corrections of both
    operands, the default at cfg-4 dims: ~1e-5) and is what bench.py times as other_configs.cfg4.high.  `tol` pins the bf16 behaviour.

    Gradients: relu6 kinks and max-pool ties are DECISIONS taken on values that carry the forward's bf16 rounding; one flipped
    decision moves a whole dy.  So (1) the tower's decisions (argmax frame per (video, cluster), relu6 masks of both layers) are
    compared with the oracle's: they may differ only where the oracle's margin is within the rounding of the compared values
    (0.06 on O(1..6) activations) and on few entries; (2) the gradients are compared with the oracle's reverse mode run on the
    TOWER'S decisions - the same smooth function on both sides - at relative L2 < 3e-2 per tensor."""
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
    # ---- (1) the discrete decisions -------------------------------------------------------------------------------
    act_bn, am_ref, pooled_ref, hid_bn = cache[4], cache[6], cache[7], cache[8]
    a3 = mm.relu6(act_bn).reshape(B, S, C)
    am_gpu = tw.arg.cpu().numpy().astype(np.int64)[:B, :C]
    pooled_gpu, h6_gpu = _np(tw.pooled), _np(tw.h6)
    bi, ci = np.meshgrid(np.arange(B), np.arange(C), indexing="ij")
    differ = am_gpu != am_ref
    margin = pooled_ref - a3[bi, am_gpu, ci]                  # how far below the oracle's maximum the tower's choice lies
    assert margin.min() >= 0.0 and float(differ.mean()) < 0.05, float(differ.mean())
    assert margin[differ].max(initial=0.0) < 0.06, margin[differ].max(initial=0.0)       # only near-ties (or saturated: both 0 / 6)
    mask_sel_ref = (pooled_ref > 0) & (pooled_ref < 6)
    mask_sel_gpu = (pooled_gpu > 0) & (pooled_gpu < 6)
    dk = np.minimum(np.abs(pooled_ref), np.abs(pooled_ref - 6.0))                         # distance of the oracle's value to a kink
    flips = mask_sel_gpu != mask_sel_ref
    assert float(flips.mean()) < 0.02 and dk[flips].max(initial=0.0) < 0.06, (float(flips.mean()), dk[flips].max(initial=0.0))
    mask_h_ref = (hid_bn > 0) & (hid_bn < 6)
    mask_h_gpu = (h6_gpu > 0) & (h6_gpu < 6)
    dkh = np.minimum(np.abs(hid_bn), np.abs(hid_bn - 6.0))
    flips_h = mask_h_gpu != mask_h_ref
    assert float(flips_h.mean()) < 0.02 and dkh[flips_h].max(initial=0.0) < 0.06, (float(flips_h.mean()), dkh[flips_h].max(initial=0.0))
    # ---- (2) the gradients on the tower's decisions ------------------------------------------------------------------
    dp = mm.cross_entropy_grad(ref_pred, labels)
    tw.backward(torch.from_numpy(dp.astype(np.float32)).to(DEV))
    gref = mm.dbof_bwd(dp, cache, routing=(am_gpu, mask_sel_gpu, mask_h_gpu))
    gplain = mm.dbof_bwd(dp, cache)
    worst = {}
    for k, g in gref.items():
        got = tw.store.g(k)
        got = _np(got.t() if got.dim() == 2 else got)
        scale = np.linalg.norm(gplain[k]) + np.linalg.norm(g)
        if k.endswith("/beta"):
            # A beta of a batch-norm that feeds matmul + batch-norm is cancelled by the next layer's mean subtraction wherever
            # the relu6 in between does not saturate: its gradient is a sum of terms that cancel - analytically zero for
            # input_bn (the tower writes exact zeros), zero or small for cluster_bn depending on the masks.
            # Judge it on the scale of the same layer's gamma gradient (the same terms without the cancellation).
            ref_scale = max(np.abs(g).max(), np.abs(gref[k[:-5] + "/gamma"]).max())
            assert np.abs(got - g).max() < 3e-2 * ref_scale, (k, np.abs(got - g).max(), ref_scale)
            continue
        l2 = float(np.linalg.norm(got - g) / (np.linalg.norm(g) + 1e-30))
        worst[k] = round(l2, 4)
        assert l2 < 3e-2, (k, l2, np.abs(got - g).max())
    print("dbof gradient relative L2 on the tower's routing:", worst)
    # moving averages: moving -= (1-0.999)*(moving - batch)
    mu = cache[3][3]
    assert np.allclose(_np(tw.buffers["input_bn/moving_mean"]), 0.001 * mu, rtol=1e-3, atol=1e-7)
    var = cache[5][4]
    assert np.allclose(_np(tw.buffers["cluster_bn/moving_variance"]), 1 - 0.001 * (1 - var), rtol=1e-3, atol=1e-6)
    # "high" precision forward: the north-star 1e-3 holds with O(1) activations - split-bf16 operands (three products per contraction) at
    # the small sizes, f16 + e4m3 corrections of both operands in one launch per contraction where every K is a multiple of 128 (cfg 4)
    tw.set_precision("high")
    assert (tw.CW in getattr(tw, "shadow_w8", {})) == (F % 128 == 0 and F >= 256 and C >= 512 and Hd >= 512)
    pred_h = tw.forward(torch.from_numpy(x).to(DEV), torch.from_numpy(n).to(DEV), torch.from_numpy(u).to(DEV))
    err_h = np.abs(_np(pred_h) - ref_pred).max()
    print('dbof pred err (high precision) %.2e' % err_h)
    assert err_h < (1e-3 if B >= 64 else 3e-3)
    tw.set_precision("bf16")
    # eval mode uses the moving statistics
    p_eval = tw.forward(torch.from_numpy(x).to(DEV), torch.from_numpy(n).to(DEV), torch.from_numpy(u).to(DEV),
                        is_training=False)
    assert torch.isfinite(p_eval).all()


@pytest.mark.parametrize("precision", ["bf16", "high"])
def test_dbof_cfg4_batch_512_properties(precision):
    """BASELINE cfg 4 at its full batch (512 videos x 30 sampled frames = 15 360 rows through the 8192-cluster kernel - the
    configuration bench.py times; the oracle tests above stop at B = 64): sampled frame indices bit-exact, finite predictions in
    [0, 1] in training mode, two training iterations lower the loss and move the batch-norm averages; with the moving
    statistics (is_training=False: no cross-video coupling through the batch moments) the videos are independent - the first 64
    rows are the B = 64 tower's predictions although every tile count differs."""
    from efficientvideoclassification_youtube8m_amd.towers import DbofTower
    from efficientvideoclassification_youtube8m_amd.distill import SingleTowerGraph
    B, S = 512, 30
    rng = np.random.default_rng(4)
    q = rng.integers(0, 256, (B, 300, 1152), dtype=np.uint8)
    n = rng.integers(1, 301, B).astype(np.int32)
    labels = np.zeros((B, 4716), np.uint8)
    labels[np.arange(B)[:, None], rng.integers(0, 4716, (B, 3))] = 1
    u = rng.random((B, S)).astype(np.float32)
    qd, yd, nd, ud = torch.from_numpy(q).to(DEV), torch.from_numpy(labels).to(DEV), torch.from_numpy(n).to(DEV), torch.from_numpy(u).to(DEV)
    tw = DbofTower(B, 300, 1152, 4716, iterations=S, cluster_size=8192, hidden_size=1024, device=DEV, seed=3)
    if precision != "bf16":
        tw.set_precision(precision)
    g = SingleTowerGraph(tw)
    losses = []
    for _ in range(3):
        out = g.step(qd, yd, nd, uniform=ud)
        losses.append(float(out["loss"]))
        assert torch.isfinite(out["predictions"]).all()
        assert float(out["predictions"].min()) >= 0.0 and float(out["predictions"].max()) <= 1.0
    assert np.array_equal(tw.idx.cpu().numpy(), mm.sample_random_frames_index(u, n))
    assert g.global_step == 3 and losses[2] < losses[0], losses
    assert float(tw.buffers["cluster_bn/moving_mean"].abs().max()) > 0
    p512 = tw.forward(qd, nd, ud, is_training=False).clone()
    tw64 = DbofTower(64, 300, 1152, 4716, iterations=S, cluster_size=8192, hidden_size=1024, device=DEV, seed=3)
    if precision != "bf16":
        tw64.set_precision(precision)
    tw64.load_state_dict(tw.state_dict())
    p64 = tw64.forward(qd[:64], nd[:64], ud[:64], is_training=False)
    d = (p64 - p512[:64]).abs().max().item()
    print("dbof B=512 %s: losses %s, first-64 rows vs the B=64 tower %.2e" % (precision, [round(v, 2) for v in losses], d))
    assert d < (2e-5 if precision == "high" else 2e-4), d


@pytest.mark.parametrize("precision", ["bf16", "high"])
@pytest.mark.parametrize("pooling,bn,random_frames", [("average", True, True), ("max", False, True), ("max", True, False),
                                                       ("average", False, False), ("max", True, True)])
def test_dbof_non_default_branches(pooling, bn, random_frames, precision):
    """The DbofModel flag values no launcher of the reference selects - dbof_pooling_method average, dbof_add_batch_norm False
    (cluster_biases / hidden1_biases), sample_random_frames False (SampleRandomSequence) - on towers.DbofGenericTower against the
    float64 oracle (oracle/model_math.py::dbof_general_fwd / _bwd, itself checked against finite differences on the CPU): sampled
    indices bit-exact, predictions, gradients by relative L2, one SingleTowerGraph training step; the default combination on the
    same tower must agree with the fused DbofTower.  precision "high" (round 5: split-bf16 forward products on this tower) holds
    north_star's 1e-3 on the predictions of every branch; bf16 is bounded at 8e-3."""
    from efficientvideoclassification_youtube8m_amd.towers import DbofGenericTower, DbofTower
    from efficientvideoclassification_youtube8m_amd.distill import SingleTowerGraph
    B, F, C, Hd, V, S = 16, 128, 256, 64, 40, 8
    rng = np.random.default_rng(11)
    q, x, n, labels = mm.synthetic_batch(B, seed=3, feature_size=F, vocab_size=V, dtype=np.float32)
    n[:3] = (1, 5, 300)                                              # shorter than the sequence / full length
    x[np.arange(300)[None, :] >= n[:, None]] = 0.0
    tw = DbofGenericTower(B, 300, F, V, iterations=S, cluster_size=C, hidden_size=Hd, device=DEV, seed=3, pooling=pooling,
                          add_batch_norm=bn, random_frames=random_frames)
    for k in tw.names:                                               # non-trivial affine / bias parameters
        if k.endswith("/gamma") or k.endswith("/beta") or k.endswith("_biases"):
            tw.store.p(k).add_(torch.from_numpy(rng.standard_normal(tw.store.p(k).shape).astype(np.float32) * 0.2).to(DEV))
    if precision != "bf16":
        tw.set_precision(precision)
    P = _params(tw)
    P["_iterations"] = S
    u = rng.random((B, S)).astype(np.float32) if random_frames else rng.random((B, 1)).astype(np.float32)
    xd, nd, ud = torch.from_numpy(x).to(DEV), torch.from_numpy(n).to(DEV), torch.from_numpy(u).to(DEV)
    pred = tw.forward(xd, nd, ud)
    xn = mm.l2_normalize(x.astype(np.float64), 2)
    ref_pred, cache = mm.dbof_general_fwd(xn, n, u if random_frames else u[:, 0], P, pooling=pooling, add_batch_norm=bn, random_frames=random_frames)
    assert np.array_equal(tw.idx.cpu().numpy(), cache[0])            # bit-exact indices (int32 truncation / clipping)
    err = np.abs(_np(pred) - ref_pred).max()
    assert err < (1e-3 if precision == "high" else 8e-3), err       # (high: measured ~1e-5; 1e-3 is north_star's bound)
    dp = mm.cross_entropy_grad(ref_pred, labels)
    tw.backward(torch.from_numpy(dp.astype(np.float32)).to(DEV))
    # the tower's discrete decisions (relu6 kinks, max-pool choice) are taken on bf16-rounded values: few may differ from the
    # oracle's, and only where the oracle's value is close to the kink / the tie; the gradients are then compared on the TOWER'S
    # decisions (the same smooth function on both sides), as in test_dbof_forward_backward
    pre, hpre = cache[3], cache[7]
    a6, h6 = _np(tw.a6), _np(tw.h6)
    mask_c, mask_h = (a6 > 0) & (a6 < 6), (h6 > 0) & (h6 < 6)
    for got_m, ref_v in ((mask_c, pre), (mask_h, hpre)):
        flips = got_m != ((ref_v > 0) & (ref_v < 6))
        dk = np.minimum(np.abs(ref_v), np.abs(ref_v - 6.0))
        assert float(flips.mean()) < 0.02 and dk[flips].max(initial=0.0) < 0.08, (float(flips.mean()), dk[flips].max(initial=0.0))
    am = None
    if pooling == "max":
        am = tw.arg.cpu().numpy().astype(np.int64)
        a3 = mm.relu6(pre).reshape(B, S, C)
        bi_, ci_ = np.meshgrid(np.arange(B), np.arange(C), indexing="ij")
        margin = a3.max(axis=1) - a3[bi_, am, ci_]
        assert margin.min() >= 0.0 and margin.max() < 0.08, margin.max()       # another frame only on a near-tie
    gref = mm.dbof_general_bwd(dp, cache, routing=(am, mask_c, mask_h))
    worst = {}
    for k, g in gref.items():
        got = tw.store.g(k)
        got = _np(got.t() if got.dim() == 2 else got)
        if k.endswith("/beta") and np.abs(g).max() < 1e-3 * np.abs(gref[k[:-5] + "/gamma"]).max():
            continue                                                 # (a beta cancelled by the next layer's mean subtraction: see test_dbof_forward_backward)
        l2 = float(np.linalg.norm(got - g) / (np.linalg.norm(g) + 1e-30))
        worst[k] = round(l2, 4)
        assert l2 < 3e-2, (k, l2)                                    # bf16 operands of the backward products
    print("dbof generic (%s, bn %s, random_frames %s, %s): pred err %.2e, gradient relative L2 %s" % (pooling, bn, random_frames, precision, err, worst))
    if pooling == "max" and bn and random_frames:                    # the default combination: the fused tower computes the same function
        tf_ = DbofTower(B, 300, F, V, iterations=S, cluster_size=C, hidden_size=Hd, device=DEV, seed=5)
        tf_.load_state_dict(tw.state_dict())
        if precision != "bf16":
            tf_.set_precision(precision)
        assert (tf_.forward(xd, nd, ud) - pred).abs().max().item() < (1e-3 if precision == "high" else 8e-3)
    g = SingleTowerGraph(tw)
    l0 = float(g.step(xd, torch.from_numpy(labels.astype(np.uint8)).to(DEV), nd, uniform=ud)["loss"])
    for _ in range(3):
        out = g.step(xd, torch.from_numpy(labels.astype(np.uint8)).to(DEV), nd, uniform=ud)
    assert g.global_step == 4 and torch.isfinite(out["predictions"]).all() and float(out["loss"]) < l0


def test_dbof_model_flag_surface_for_non_default_branches():
    """DbofModel.create_model picks the generic tower for a non-default flag combination (the reference's `x or FLAG` defaults,
    cs/frame_level_models.py:118-122), refuses 'none' pooling with the reason, and rejects an unknown method like cs/model_utils.py:83."""
    from efficientvideoclassification_youtube8m_amd import frame_level_models as flm
    from efficientvideoclassification_youtube8m_amd.flags import FLAGS
    from efficientvideoclassification_youtube8m_amd.towers import DbofGenericTower
    from efficientvideoclassification_youtube8m_amd.model_utils import SampleRandomSequence
    q, x, n, labels = mm.synthetic_batch(8, seed=4, feature_size=64, vocab_size=24, dtype=np.float32)
    xd, nd = torch.from_numpy(x).to(DEV), torch.from_numpy(n).to(DEV)
    old = (FLAGS.dbof_pooling_method, FLAGS.dbof_add_batch_norm, FLAGS.sample_random_frames)
    try:
        FLAGS.dbof_pooling_method, FLAGS.dbof_add_batch_norm, FLAGS.sample_random_frames = "average", False, False
        m = flm.DbofModel()
        out = m.create_model(xd, 24, nd, iterations=8, cluster_size=128, hidden_size=64)
        assert isinstance(m.towers["model"], DbofGenericTower) and out["predictions"].shape == (8, 24)
        assert m.towers["model"].pooling == "average" and not m.towers["model"].bn and not m.towers["model"].random_frames
        FLAGS.dbof_pooling_method = "none"
        with pytest.raises(NotImplementedError):
            flm.DbofModel().create_model(xd, 24, nd, iterations=8, cluster_size=128, hidden_size=64)
        FLAGS.dbof_pooling_method = "attention"
        with pytest.raises(ValueError):
            flm.DbofModel().create_model(xd, 24, nd, iterations=8, cluster_size=128, hidden_size=64)
    finally:
        FLAGS.dbof_pooling_method, FLAGS.dbof_add_batch_norm, FLAGS.sample_random_frames = old
    u = torch.tensor([0.0, 0.5, 0.999, 0.3, 0.1, 0.2, 0.7, 0.9], device=DEV)
    seq = SampleRandomSequence(xd, nd, 5, uniform=u)
    idx = mm.sample_random_sequence_index(u.cpu().numpy(), n, 5)
    assert torch.equal(seq.cpu(), torch.from_numpy(x[np.arange(8)[:, None], idx, :]))


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
