"""NetVLAD tower (extension: the reference's NetVLADModel is an empty stub - no reference math) against its own float64
oracle (oracle/model_math.py::netvlad_fwd/bwd, itself checked against finite differences on the CPU): kernels, forward,
gradients, a training step through SingleTowerGraph and the create_model interface.  pytest -m gpu."""
import numpy as np
import pytest
import torch

from oracle import model_math as mm

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _np(t):
    return t.detach().cpu().double().numpy()


def _params(tw):
    pre = tw.scope + "/"
    return {k[len(pre):]: _np(v) for k, v in tw.state_dict().items()}


@pytest.mark.parametrize("B,F,K,H,V,S,u8", [(16, 64, 64, 64, 40, 10, False), (32, 128, 64, 128, 33, 30, True)])
def test_netvlad_forward_backward_against_oracle(B, F, K, H, V, S, u8):
    from efficientvideoclassification_youtube8m_amd.towers import NetVladTower
    rng = np.random.default_rng(B + K)
    q, x, n, labels = mm.synthetic_batch(B, seed=B, feature_size=F, vocab_size=V, dtype=np.float32)
    tw = NetVladTower(B, 300, F, V, iterations=S, cluster_size=K, hidden_size=H, device=DEV, seed=3)
    for k in tw.names:                                   # non-trivial batch-norm scale / offset
        if k.endswith("/gamma") or k.endswith("/beta"):
            tw.store.p(k).add_(torch.from_numpy(rng.standard_normal(tw.store.p(k).shape).astype(np.float32) * 0.2).to(DEV))
    P = _params(tw)
    assert P["hidden1_weights"].shape == (F * K, H) and P["cluster_weights2"].shape == (F, K)
    u = rng.random((B, S)).astype(np.float32)
    xin = torch.from_numpy(q).to(DEV) if u8 else torch.from_numpy(x).to(DEV)
    pred = tw.forward(xin, torch.from_numpy(n).to(DEV), torch.from_numpy(u).to(DEV))
    xn = mm.l2_normalize(x.astype(np.float64), 2)
    ref_pred, cache = mm.netvlad_fwd(xn, n, u, P)
    assert np.array_equal(tw.idx.cpu().numpy(), mm.sample_random_frames_index(u, n))
    # the f32 pieces against the oracle's intermediates (bf16 only enters through the two GEMMs)
    a3, V_ref, Y_ref = cache[3], cache[7], cache[11]
    assert np.abs(_np(tw.a).reshape(B, S, K) - a3).max() < 2e-2                      # assignment: bf16 logits through a softmax
    Yk = Y_ref.reshape(B, F, K).transpose(0, 2, 1).reshape(B, K * F)               # oracle order f*K+k -> cluster-major
    assert np.abs(tw.Y_bf[:B].float().cpu().numpy() - Yk).max() < 2e-2
    err = np.abs(_np(pred) - ref_pred).max()
    print("netvlad pred err %.2e" % err)
    assert err < 5e-3
    dp = mm.cross_entropy_grad(ref_pred, labels)
    tw.backward(torch.from_numpy(dp.astype(np.float32)).to(DEV))
    gref = mm.netvlad_bwd(dp, cache)
    sd_g = {}
    for k in tw.names:
        g = tw.store.g(k)
        g = _np(g.t() if g.dim() == 2 else g)
        if k == tw.HW:
            g = g.reshape(K, F, H).transpose(1, 0, 2).reshape(F * K, H)
        sd_g[k] = g
    for k, g in gref.items():
        l2 = float(np.linalg.norm(sd_g[k] - g) / (np.linalg.norm(g) + 1e-30))
        # (relu6 masks can flip on values within bf16 rounding of 0 / 6: judge each tensor by its relative L2 error)
        assert l2 < 0.12 or np.abs(sd_g[k] - g).max() < 1e-3, (k, l2, np.abs(sd_g[k] - g).max())
    # state_dict round trip (hidden1_weights row order f*K+k <-> the internal cluster-major columns)
    tw2 = NetVladTower(B, 300, F, V, iterations=S, cluster_size=K, hidden_size=H, device=DEV, seed=9)
    tw2.load_state_dict(tw.state_dict())
    assert torch.equal(tw2.store.master, tw.store.master)


def test_netvlad_kernels_in_f32_against_numpy():
    """csrc/evc_netvlad.hip alone (no bf16 anywhere): softmax fwd/bwd, aggregation fwd/bwd, centre gradient, the two
    normalisations fwd/bwd - against numpy float64 to f32 accuracy."""
    from efficientvideoclassification_youtube8m_amd import ops
    rng = np.random.default_rng(2)
    B, S, K, F = 5, 7, 24, 40
    R = B * S
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a, np.float32)).to(DEV)
    act = rng.standard_normal((R, K)); mean = rng.standard_normal(K) * 0.1; var = rng.random(K) + 0.5
    ga = 1 + 0.2 * rng.standard_normal(K); be = 0.2 * rng.standard_normal(K)
    a = torch.empty((R, K), device=DEV)
    ops.netvlad_softmax_fwd(d(act), R, K, d(mean), d(var), d(ga), d(be), a)
    y = (act - mean) / np.sqrt(var + 1e-3) * ga + be
    e = np.exp(y - y.max(1, keepdims=True)); a_ref = e / e.sum(1, keepdims=True)
    assert np.abs(_np(a) - a_ref).max() < 1e-6
    da = rng.standard_normal((R, K)); dz = torch.empty((R, K), device=DEV)
    ops.netvlad_softmax_bwd(d(a_ref), d(da), R, K, dz)
    assert np.abs(_np(dz) - a_ref * (da - (a_ref * da).sum(1, keepdims=True))).max() < 1e-5
    r = rng.standard_normal((R, F)); mu = rng.standard_normal(F) * 0.1; vf = rng.random(F) + 0.5
    gf = 1 + 0.2 * rng.standard_normal(F); bf = 0.2 * rng.standard_normal(F); c2 = rng.standard_normal((K, F))
    xbn = ((r - mu) / np.sqrt(vf + 1e-3) * gf + bf).reshape(B, S, F)
    a3 = a_ref.reshape(B, S, K)
    V = torch.empty((B, K, F), device=DEV); asum = torch.empty((B, K), device=DEV)
    ops.netvlad_aggregate_fwd(d(a_ref), d(r), B, S, K, F, d(mu), d(vf), d(gf), d(bf), d(c2), V, asum)
    V_ref = np.einsum("bsk,bsf->bkf", a3, xbn) - a3.sum(1)[:, :, None] * c2[None]
    assert np.abs(_np(V) - V_ref).max() < 1e-4 and np.abs(_np(asum) - a3.sum(1)).max() < 1e-5
    n1 = torch.empty((B, K), device=DEV); n2 = torch.empty(B, device=DEV)
    Yb = torch.empty((B, K * F), dtype=torch.bfloat16, device=DEV); Yf = torch.empty((B, K * F), device=DEV)
    ops.netvlad_normalize_fwd(d(V_ref), B, K, F, n1, n2, Yb, Yf)
    n1r = np.sqrt((V_ref ** 2).sum(2)); U = V_ref / n1r[:, :, None]; n2r = np.sqrt((U ** 2).sum((1, 2))); Y = U / n2r[:, None, None]
    assert np.abs(_np(Yf).reshape(B, K, F) - Y).max() < 1e-6 and np.abs(_np(n1) - n1r).max() < 1e-4 and torch.equal(Yb, Yf.bfloat16())
    dY = rng.standard_normal((B, K, F)); dV = torch.empty((B, K, F), device=DEV)
    ops.netvlad_normalize_bwd(d(V_ref), d(n1r), d(n2r), d(dY.reshape(B, K * F)), B, K, F, dV)
    dU = (dY - Y * (Y * dY).sum((1, 2), keepdims=True)) / n2r[:, None, None]
    dV_ref = (dU - U * (U * dU).sum(2, keepdims=True)) / n1r[:, :, None]
    assert np.abs(_np(dV) - dV_ref).max() < 1e-4 * max(1.0, np.abs(dV_ref).max())
    da_o = torch.empty((R, K), device=DEV); dx_o = torch.empty((R, F), device=DEV); dc2 = torch.empty((K, F), device=DEV)
    ops.netvlad_aggregate_bwd(d(a_ref), d(r), B, S, K, F, d(mu), d(vf), d(gf), d(bf), d(c2), d(dV_ref), da_o, dx_o)
    da_ref = np.einsum("bkf,bsf->bsk", dV_ref, xbn) - np.einsum("bkf,kf->bk", dV_ref, c2)[:, None, :]
    dx_ref = np.einsum("bsk,bkf->bsf", a3, dV_ref)
    assert np.abs(_np(da_o).reshape(B, S, K) - da_ref).max() < 1e-3 * max(1.0, np.abs(da_ref).max())
    assert np.abs(_np(dx_o).reshape(B, S, F) - dx_ref).max() < 1e-4 * max(1.0, np.abs(dx_ref).max())
    ops.netvlad_dcenters(d(a3.sum(1)), d(dV_ref), B, K, F, dc2)
    assert np.abs(_np(dc2) + np.einsum("bk,bkf->kf", a3.sum(1), dV_ref)).max() < 1e-4 * max(1.0, np.abs(dV_ref).max() * B)


def test_netvlad_through_create_model_and_train_main(tmp_path):
    from efficientvideoclassification_youtube8m_amd import frame_level_models, train
    from efficientvideoclassification_youtube8m_amd.flags import FLAGS
    FLAGS.reset()
    FLAGS.parse(["--netvlad_cluster_size", "64", "--netvlad_hidden_size", "64", "--iterations", "8"])
    B, F, V = 8, 64, 20
    q, x, n, labels = mm.synthetic_batch(B, seed=4, feature_size=F, vocab_size=V, dtype=np.float32)
    m = frame_level_models.NetVLADModel()
    out = m.create_model(torch.from_numpy(x).to(DEV), V, torch.from_numpy(n).to(DEV), normalize_input=True)
    assert set(out) == {"predictions"} and tuple(out["predictions"].shape) == (B, V)
    assert 0.0 <= float(out["predictions"].min()) and float(out["predictions"].max()) <= 1.0
    assert m.create_model_inference(None, V, 10, None) is None                       # the reference's stub
    FLAGS.reset()
    res = train.main(["--train_data_pattern", "synthetic", "--train_dir", str(tmp_path) + "/", "--frame_features", "True",
                      "--feature_names", "rgb, audio", "--feature_sizes", "64, 64", "--model", "NetVLADModel", "--batch_size", "16",
                      "--iterations", "10", "--netvlad_cluster_size", "64", "--netvlad_hidden_size", "64", "--num_epochs", "1",
                      "--synthetic_videos", "64", "--start_new_model", "True", "--base_learning_rate", "0.01"])
    FLAGS.reset()
    assert res["iterations"] == 4 and res["graph"].global_step == 4
    losses = [h[1]["loss"] for h in res["history"]]
    assert all(np.isfinite(l) for l in losses) and losses[-1] < losses[0]            # it trains
    sd = torch.load(train.latest_checkpoint(str(tmp_path) + "/"))
    assert sd["model/hidden1_weights"].shape == (128 * 64, 64) and sd["model/cluster_weights2"].shape == (128, 64)
