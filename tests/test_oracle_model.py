"""Self-checks of oracle/model_math.py (CPU).

The TF arithmetic is not runnable here ("parity unpinned"), so the oracle is
anchored by: (i) the known answers the reference publishes (README.md:93-121),
(ii) an independent second restatement in torch float64 autograd written in
this file, (iii) finite differences, (iv) the exhaustive integer facts of
SURVEY.md Appendix D-5/D-6.
"""
import numpy as np
import pytest
import torch

from oracle import model_math as mm


def tiny_params(rng, F=6, H=4, L=2, V=5, M=2):
    p = mm.init_hlstm_params(rng, F, H, L, V, M)
    for k in p:  # non-zero biases so bias paths are exercised
        if k.endswith("bias") or k.endswith("biases"):
            p[k] = rng.standard_normal(p[k].shape) * 0.1
    return p


# ---- second, independent restatement (torch autograd, float64) ------------

def t_lstm_seq(x, lengths, layers):
    B, T, _ = x.shape
    H = layers[0][1].shape[0] // 4
    c = [x.new_zeros(B, H) for _ in layers]
    h = [x.new_zeros(B, H) for _ in layers]
    for t in range(T):
        act = (t < lengths).unsqueeze(1)
        inp = x[:, t]
        for l, (k, b) in enumerate(layers):
            z = torch.cat([inp, h[l]], 1) @ k + b
            i, j, f, o = z.chunk(4, 1)
            cn = c[l] * torch.sigmoid(f + 1.0) + torch.sigmoid(i) * torch.tanh(j)
            hn = torch.tanh(cn) * torch.sigmoid(o)
            c[l] = torch.where(act, cn, c[l])
            h[l] = torch.where(act, hn, h[l])
            inp = hn
    return torch.cat([s for l in range(len(layers)) for s in (c[l], h[l])], 1)


def t_hlstm(x, n, P, C, L=2, M=2):
    B, T, F = x.shape
    Lc = T // C
    outs = []
    l1 = [(P["RNN_L1/rnn/multi_rnn_cell/cell_%d/basic_lstm_cell/kernel" % l],
           P["RNN_L1/rnn/multi_rnn_cell/cell_%d/basic_lstm_cell/bias" % l]) for l in range(L)]
    l2 = [(P["RNN_L2/rnn/multi_rnn_cell/cell_%d/basic_lstm_cell/kernel" % l],
           P["RNN_L2/rnn/multi_rnn_cell/cell_%d/basic_lstm_cell/bias" % l]) for l in range(L)]
    for i in range(C):
        ln = torch.clamp(n - Lc * i, 0, Lc)
        outs.append(t_lstm_seq(x[:, i * Lc:(i + 1) * Lc], ln, l1))
    l2in = torch.stack(outs, 1)
    state = t_lstm_seq(l2in, torch.ceil(n.double() / Lc).long(), l2)
    V = P["classifier/experts/biases"].shape[0] // M
    g = torch.softmax((state @ P["classifier/gates/weights"]).reshape(-1, M + 1), 1)
    e = torch.sigmoid((state @ P["classifier/experts/weights"] + P["classifier/experts/biases"]).reshape(-1, M))
    p = (g[:, :M] * e).sum(1).reshape(B, V)
    return state, p


def t_ce(p, y):
    return (-(y * torch.log(p + 1e-5) + (1 - y) * torch.log(1 - p + 1e-5))).sum(1).mean()


def test_teacher_student_step_matches_torch_autograd():
    rng = np.random.default_rng(0)
    B, F, H, V, every_n = 3, 6, 4, 5, 30       # student sees 10 frames -> 5 chunks of 2
    T = mm.init_hlstm_params  # noqa
    teacher, student = tiny_params(rng, F, H, 2, V), tiny_params(rng, F, H, 2, V)
    x_raw = rng.standard_normal((B, 300, F))
    n = np.array([300, 137, 31], np.int32)
    x_raw[np.arange(300)[None] >= n[:, None]] = 0
    labels = rng.random((B, V)) > 0.6
    out = mm.teacher_student_step(x_raw, n, labels, teacher, student, every_n, num_inputs_to_lstm=20)

    tt = {k: torch.tensor(v, requires_grad=True) for k, v in teacher.items()}
    ts = {k: torch.tensor(v, requires_grad=True) for k, v in student.items()}
    xr = torch.tensor(x_raw)
    xn = xr * torch.rsqrt(torch.clamp((xr * xr).sum(2, keepdim=True), min=1e-12))
    y = torch.tensor(labels.astype(np.float64))
    nt = torch.tensor(n.astype(np.int64))
    st, pt = t_hlstm(xn, nt, tt, 20)
    loss_t = 2.0 * 1e-8 * 0.5 * ((tt["classifier/gates/weights"] ** 2).sum()
                                + (tt["classifier/experts/weights"] ** 2).sum()) + t_ce(pt, y)
    loss_t.backward()
    idx = list(range(0, 300, every_n))
    ns = torch.tensor(mm.student_num_frames(n, every_n))
    ss, ps = t_hlstm(xn[:, idx], ns, ts, 5)
    std, ptd = st.detach(), pt.detach()
    lrep = ((std - ss) ** 2).sum(1).mean()
    P = ptd / ptd.sum(1, keepdim=True)
    Q = ps / ps.sum(1, keepdim=True)
    lpred = (P * (P.log() - Q.log())).sum()
    lce = t_ce(ps, y)
    reg = 1e-8 * 0.5 * ((ts["classifier/gates/weights"] ** 2).sum() + (ts["classifier/experts/weights"] ** 2).sum())
    total = lrep + lpred + lce + lrep + 2.0 * reg
    total.backward()

    assert np.allclose(out["teacher_state"], st.detach().numpy(), atol=1e-12)
    assert np.allclose(out["teacher_predictions"], pt.detach().numpy(), atol=1e-12)
    assert np.allclose(out["student_state"], ss.detach().numpy(), atol=1e-12)
    assert out["final_loss"] == pytest.approx(loss_t.item(), rel=1e-12)
    assert out["student_loss_state"] == pytest.approx(lrep.item(), rel=1e-12)
    assert out["pred_loss"] == pytest.approx(lpred.item(), rel=1e-10, abs=1e-14)
    assert out["total_student_loss"] == pytest.approx(total.item(), rel=1e-12)
    for k in mm.HLSTM_PARAM_ORDER:
        assert np.allclose(out["teacher_grads"][k], tt[k].grad.numpy(), rtol=1e-9, atol=1e-13), k
        assert np.allclose(out["student_grads"][k], ts[k].grad.numpy(), rtol=1e-9, atol=1e-13), k


def test_folded_equals_unfolded_chunk_loops():
    rng = np.random.default_rng(1)
    P = tiny_params(rng)
    x = rng.standard_normal((4, 20, 6))
    n = np.array([20, 13, 4, 0])
    s1, p1, _ = mm.hlstm_fwd(x, n, P, 5)
    s2, p2 = mm.hlstm_fwd_unfolded(x, n, P, 5)
    assert np.array_equal(s1, s2) and np.array_equal(p1, p2)
    assert np.all(s1[3] == 0)   # zero-length video: all-zero state (dynamic_rnn semantics)


def test_lstm_finite_differences():
    rng = np.random.default_rng(2)
    layers = [(rng.standard_normal((3 + 4, 16)) * 0.4, rng.standard_normal(16) * 0.1),
              (rng.standard_normal((4 + 4, 16)) * 0.4, rng.standard_normal(16) * 0.1)]
    x = rng.standard_normal((3, 5, 3))
    lens = np.array([5, 2, 0])
    w = rng.standard_normal((3, 16))
    f = lambda: float((mm.multi_rnn_seq_fwd(x, lens, layers, False)[0] * w).sum())
    s, cache = mm.multi_rnn_seq_fwd(x, lens, layers)
    dx, grads = mm.multi_rnn_seq_bwd(w.copy(), cache, layers)
    eps = 1e-6
    for arr, g in ((layers[0][0], grads[0][0]), (layers[1][1], grads[1][1]), (x, dx)):
        it = np.nditer(arr, flags=["multi_index"])
        cnt = 0
        while not it.finished and cnt < 25:
            i = it.multi_index
            old = arr[i]
            arr[i] = old + eps; fp = f()
            arr[i] = old - eps; fm = f()
            arr[i] = old
            assert (fp - fm) / (2 * eps) == pytest.approx(g[i], rel=1e-5, abs=1e-8)
            it.iternext(); cnt += 1


def test_known_answer_initial_losses():
    """README.md:116: 'Teacher_Loss: 1914.09 | L_REP: 1.16 | L_PRED: 0.01 | L_CE: 1914.1'
    at step 2 (first iteration, random init, batch 256).  Real dimensions
    (F=1152, H=1024, 2 layers, V=4716), small batch, synthetic inputs: at init the
    MoE gives p ~= 1/3 per class so CE ~= 4716*(-log(2/3)) + ~3.4*log 2 ~= 1914."""
    rng = np.random.default_rng(3)
    B = 2
    _, x, n, y = mm.synthetic_batch(B, seed=11, dtype=np.float32)
    teacher = mm.init_hlstm_params(rng, dtype=np.float32)
    student = mm.init_hlstm_params(rng, dtype=np.float32)
    out = mm.teacher_student_step(x, n, y, teacher, student, 10, with_grads=False)
    assert out["x_student"].shape == (B, 30, 1152)                    # README.md:100-102
    assert out["teacher_predictions"].shape == (B, 4716)
    assert abs(out["label_loss"] - 1914.1) / 1914.1 < 0.005, out["label_loss"]
    assert abs(out["student_label_loss"] - 1914.1) / 1914.1 < 0.005
    assert out["pred_loss"] / B < 0.01                                 # L_PRED 0.01 at B=256 => tiny per video
    assert 0.0 < out["student_loss_state"] < 10.0   # L_REP 1.16 in the log on real data; data-dependent

def test_known_answer_second_logged_iteration():
    """README.md:116,121 - the only reference-held evidence about the UPDATE (cs/train.py:329-334,413-418,516-533): after one
    iteration (both train ops, per-tensor clip_by_norm(1.0) + Adam(1e-3)) on a fresh batch of 256 the log reads

        training step 2| ... Teacher_Loss: 1914.09| L_REP: 1.16| L_PRED: 0.01| L_CE: 1914.1
        training step 4| ... Teacher_Loss: 1908.12| L_REP: 1.52| L_PRED: 0.01| L_CE: 1913.41

    i.e. the teacher's CE falls by ~6, the student's several times less (~0.7: its clipped update also serves 2 L_REP + L_PRED),
    L_REP rises (the towers move apart once both are updated), L_PRED stays ~0, and `global_step` advances by 2 per iteration
    (both create_train_op's increment the shared counter).  Two iterations of the CPU restatement (oracle/torch_cpu.py) on fresh
    synthetic batches reproduce that pattern; the magnitudes are data-dependent (the log is YouTube-8M at batch 256, this is
    uint8-uniform noise at batch 16), so the assertions are the signs and the ordering, with loose bands around the logged sizes."""
    from oracle import torch_cpu as tc
    torch.set_num_threads(min(8, torch.get_num_threads()))
    rng = np.random.default_rng(5)
    teacher = tc.to_torch(mm.init_hlstm_params(rng, dtype=np.float32))
    student = tc.to_torch(mm.init_hlstm_params(rng, dtype=np.float32))
    opt_t, opt_s = tc.Adam(teacher, lr=1e-3), tc.Adam(student, lr=1e-3)
    B, logged, global_step = 16, [], 0
    for it in range(2):
        _, x, n, y = mm.synthetic_batch(B, seed=100 + it, dtype=np.float32)      # a FRESH batch per iteration, like the input queue
        out = tc.teacher_student_iteration(torch.from_numpy(x), n, torch.from_numpy(y.astype(np.float32)), teacher, student, 10,
                                           opt_t, opt_s)
        global_step += 2                                                         # two train ops per sess.run (Appendix D-2)
        logged.append((global_step, out["label_loss"], out["student_loss_state"], out["pred_loss"] / B, out["student_label_loss"]))
    (s0, t0, rep0, kl0, ce0), (s1, t1, rep1, kl1, ce1) = logged
    print("iteration log (global_step, Teacher_Loss, L_REP, L_PRED per video, L_CE):", logged)
    assert (s0, s1) == (2, 4) and opt_t.t == 2 and opt_s.t == 2
    assert abs(t0 - 1914.1) / 1914.1 < 0.005 and abs(ce0 - 1914.1) / 1914.1 < 0.005      # README.md:116
    d_teacher, d_student = t0 - t1, ce0 - ce1
    assert 1.0 < d_teacher < 40.0, d_teacher                  # logged: 5.97
    assert 0.0 < d_student < d_teacher / 2.5, (d_student, d_teacher)     # logged: 0.69 = 8.7x less
    assert rep1 > rep0 > 0.0                                  # logged: 1.16 -> 1.52
    assert 0.0 <= kl0 < 0.01 and 0.0 <= kl1 < 0.05            # logged 0.01 for the batch SUM of 256 videos


def test_every_n_index_lists_and_counts():
    assert mm.every_n_indices(10) == list(range(0, 300, 10)) and len(mm.every_n_indices(10)) == 30  # README.md:100-102
    assert len(mm.every_n_indices(30)) == 10
    assert len(mm.every_n_indices(7)) == 43 and mm.student_max_frames(7) == 42          # Appendix D-6
    n = np.arange(0, 301, dtype=np.int32)
    for every_n in (5, 10, 15, 20, 30):
        s = 300 // every_n
        assert np.array_equal(mm.student_num_frames(n, every_n), (n.astype(np.int64) * s) // 300)
    # float64 true-division quirk (Appendix D-5): differs from floor(n*S/300)
    diff = {e: [int(v) for v in n if mm.student_num_frames([v], e)[0] != (int(v) * (300 // e)) // 300]
            for e in (2, 3, 4, 6)}
    assert diff == {2: [110, 158, 194, 220, 246], 3: [87, 171, 174], 4: [220], 6: [174]}
    ok = [e for e in range(1, 301) if _admissible(e)]
    assert ok == [1, 2, 3, 4, 5, 6, 10, 12, 15, 20, 30, 60]


def _admissible(e):
    try:
        mm.validate_every_n(e)
        return True
    except ValueError:
        return False


def test_chunk_lengths():
    l1, l2 = mm.hlstm_chunk_lengths(np.array([300, 137, 15, 16, 0]), 20, 15)
    assert l1[0].tolist() == [15] * 20 and l2[0] == 20
    assert l1[1].tolist() == [15] * 9 + [2] + [0] * 10 and l2[1] == 10
    assert l2.tolist() == [20, 10, 1, 2, 0]
    n = np.arange(0, 301)
    for L in (15, 6, 2, 1, 3, 30):
        assert np.array_equal(mm.hlstm_chunk_lengths(n, 300 // L if L * (300 // L) == 300 else 1, L)[1],
                              -(-n // L))       # float32 ceil == integer ceil-div (Appendix A-13)


def test_adam_clip_decay():
    g = np.array([3.0, 4.0])
    assert np.allclose(mm.clip_by_norm(g, 1.0), g / 5.0)
    assert np.allclose(mm.clip_by_norm(g * 0.01, 1.0), g * 0.01)
    p, m, v = mm.adam_step(np.zeros(2), g, np.zeros(2), np.zeros(2), 1, 1e-3)
    lr_t = 1e-3 * np.sqrt(1 - 0.999) / (1 - 0.9)
    assert np.allclose(p, -lr_t * (0.1 * g) / (np.sqrt(0.001 * g * g) + 1e-8))
    assert mm.exponential_decay(1e-3, 20000, 256, 4000000, 0.5) == pytest.approx(0.5e-3)


def test_dbof_and_logistic_grads_finite_difference():
    rng = np.random.default_rng(5)
    B, T, F, C, Hd, V = 3, 12, 5, 7, 4, 6
    P = mm.init_dbof_params(rng, F, C, Hd, V)
    for k in P:
        if "bn/" in k:
            P[k] = P[k] + rng.standard_normal(P[k].shape) * 0.2
    x = rng.standard_normal((B, T, F))
    n = np.array([12, 7, 3])
    u = rng.random((B, 4)).astype(np.float32)
    w = rng.standard_normal((B, V))
    pred, cache = mm.dbof_fwd(x, n, u, P)
    g = mm.dbof_bwd(w, cache)
    f = lambda: float((mm.dbof_fwd(x, n, u, P)[0] * w).sum())
    eps = 1e-6
    for k in ("cluster_weights", "hidden1_weights", "cluster_bn/gamma", "input_bn/beta", "hidden1_bn/beta"):
        arr = P[k]
        for i in list(np.ndindex(arr.shape))[:12]:
            old = arr[i]
            arr[i] = old + eps; fp = f()
            arr[i] = old - eps; fm = f()
            arr[i] = old
            assert (fp - fm) / (2 * eps) == pytest.approx(g[k][i], rel=2e-4, abs=1e-7), (k, i)
    idx = mm.sample_random_frames_index(u, n)
    assert idx.dtype == np.int32 and np.all(idx < n[:, None]) and np.all(idx >= 0)
    W = rng.standard_normal((F, V)); b = rng.standard_normal(V)
    p, avg = mm.logistic_fwd(x, n, W, b)
    assert np.allclose(avg, x.sum(1) / n[:, None])


def test_torch_cpu_baseline_matches_the_float64_oracle():
    """oracle/torch_cpu.py (the float32 PyTorch-CPU restatement bench.py times as `cpu_baseline`, structured like the
    reference graph: one dynamic_rnn per chunk at batch B, autograd) against model_math (float64, chunks folded into
    the batch, hand-written reverse mode): losses, gradients and one clip+Adam step."""
    import torch
    from oracle import torch_cpu as tc
    rng = np.random.default_rng(3)
    B, F, H, V, every_n = 5, 12, 8, 7, 10
    teacher = mm.init_hlstm_params(rng, F, H, 2, V)
    student = mm.init_hlstm_params(rng, F, H, 2, V)
    for p in (teacher, student):                     # larger weights: states well away from zero
        for k in p:
            p[k] = p[k] * 3.0
    x = rng.standard_normal((B, 300, F))
    n = np.array([300, 181, 150, 31, 9])
    x[np.arange(300)[None] >= n[:, None]] = 0
    y = rng.random((B, V)) > 0.6
    ref = mm.teacher_student_step(x, n, y, teacher, student, every_n)
    tt, ts = tc.to_torch(teacher), tc.to_torch(student)
    opt_t, opt_s = tc.Adam(tt), tc.Adam(ts)
    out = tc.teacher_student_iteration(torch.tensor(x, dtype=torch.float32), n, torch.tensor(y, dtype=torch.float32), tt, ts,
                                       every_n, opt_t, opt_s)
    for k in ("label_loss", "student_label_loss", "student_loss_state", "pred_loss"):
        assert abs(out[k] - ref[k]) <= 1e-5 * abs(ref[k]) + 1e-7, (k, out[k], float(ref[k]))
    for got, want in ((out["teacher_grads"], ref["teacher_grads"]), (out["student_grads"], ref["student_grads"])):
        for k in mm.HLSTM_PARAM_ORDER:
            g = got[k].numpy()
            assert np.abs(g - want[k]).max() <= 1e-4 * np.abs(want[k]).max() + 1e-7, k
    new_t = mm.apply_train_op(teacher, ref["teacher_grads"], {}, 1, 1e-3, 1.0)
    for k in mm.HLSTM_PARAM_ORDER:
        assert np.abs(tt[k].detach().numpy() - new_t[k]).max() < 2e-5, k        # one Adam step moves each weight by ~1e-3
    # the single-tower modes
    o_t = tc.teacher_student_iteration(torch.tensor(x, dtype=torch.float32), n, torch.tensor(y, dtype=torch.float32), tt, ts,
                                       every_n, mode="teacher")
    o_s = tc.teacher_student_iteration(torch.tensor(x, dtype=torch.float32), n, torch.tensor(y, dtype=torch.float32), tt, ts,
                                       30, mode="student")
    assert "student_grads" not in o_t and "teacher_grads" not in o_s and np.isfinite(o_s["student_label_loss"])


def test_netvlad_oracle_backward_matches_finite_differences():
    """The NetVLAD extension has no reference math (empty stub): its oracle is checked against itself - hand-written
    reverse mode vs central finite differences of the cross-entropy loss, every parameter tensor."""
    rng = np.random.default_rng(11)
    B, T, F, K, H, V, S = 6, 12, 8, 5, 7, 9, 4
    P = mm.init_netvlad_params(rng, F, K, H, V)
    for k in P:
        if k.endswith("gamma") or k.endswith("beta"):
            P[k] = P[k] + 0.2 * rng.standard_normal(P[k].shape)
    x = rng.standard_normal((B, T, F))
    n = rng.integers(1, T + 1, B)
    u = rng.random((B, S)).astype(np.float32)
    y = rng.random((B, V)) > 0.6

    def loss(Q):
        p, _ = mm.netvlad_fwd(x, n, u, Q)
        return mm.cross_entropy_loss(p, y)

    pred, cache = mm.netvlad_fwd(x, n, u, P)
    grads = mm.netvlad_bwd(mm.cross_entropy_grad(pred, y), cache)
    assert set(grads) == set(P)
    eps = 1e-6
    for k in P:
        flat = P[k].reshape(-1)
        for j in rng.choice(flat.size, size=min(6, flat.size), replace=False):
            old = flat[j]
            flat[j] = old + eps
            lp = loss(P)
            flat[j] = old - eps
            lm = loss(P)
            flat[j] = old
            fd = (lp - lm) / (2 * eps)
            got = grads[k].reshape(-1)[j]
            assert abs(fd - got) <= 1e-5 * max(1.0, abs(fd)) + 1e-7, (k, j, fd, got)


def test_dbof_general_oracle_matches_the_default_form_and_finite_differences():
    """oracle/model_math.py::dbof_general_fwd / _bwd (every trainable flag combination of cs/frame_level_models.py:108-195): the
    default combination reproduces dbof_fwd / dbof_bwd exactly; the non-default one (average pooling, biases instead of batch-norm,
    SampleRandomSequence) has its hand-written reverse mode checked against central finite differences; the sequence indices
    follow cs/model_utils.py:22-36 (start from the float32 product, clipping at n - 1)."""
    rng = np.random.default_rng(0)
    B, F, C, H, V, S = 4, 8, 16, 8, 6, 5
    q, x, n, labels = mm.synthetic_batch(B, seed=1, feature_size=F, vocab_size=V, dtype=np.float64)
    P = mm.init_dbof_params(rng, F, C, H, V)
    u = rng.random((B, S)).astype(np.float32)
    xn = mm.l2_normalize(x, 2)
    p0, c0 = mm.dbof_fwd(xn, n, u, P)
    p1, c1 = mm.dbof_general_fwd(xn, n, u, P)
    assert np.array_equal(p0, p1)
    dp = mm.cross_entropy_grad(p0, labels)
    g0, g1 = mm.dbof_bwd(dp, c0), mm.dbof_general_bwd(dp, c1)
    for k in g0:
        assert np.allclose(g0[k], g1[k], rtol=1e-12, atol=1e-15), k
    P2 = dict(P)
    P2["cluster_biases"], P2["hidden1_biases"], P2["_iterations"] = rng.standard_normal(C) / np.sqrt(F), rng.standard_normal(H) * 0.01, S
    for k in ("cluster_weights", "hidden1_weights"):
        P2[k] = P2[k] * 3
    u1 = rng.random(B).astype(np.float32)
    kw = dict(pooling="average", add_batch_norm=False, random_frames=False)

    def loss(P_):
        return mm.cross_entropy_loss(mm.dbof_general_fwd(xn, n, u1, P_, **kw)[0], labels)
    p, c = mm.dbof_general_fwd(xn, n, u1, P2, **kw)
    g = mm.dbof_general_bwd(mm.cross_entropy_grad(p, labels), c)
    for k in ("cluster_weights", "cluster_biases", "hidden1_weights", "hidden1_biases"):
        a = P2[k]
        for cnt, i in enumerate(np.ndindex(a.shape)):
            if cnt % 7:
                continue
            old = a[i]
            a[i] = old + 1e-6
            lp = loss(P2)
            a[i] = old - 1e-6
            lm = loss(P2)
            a[i] = old
            fd = (lp - lm) / 2e-6
            if abs(fd) > 1e-7:
                assert abs(fd - g[k][i]) < 1e-5 * abs(fd) + 1e-9, (k, i, fd, g[k][i])
    idx = mm.sample_random_sequence_index(np.array([0.0, 0.5, 0.999, 0.3], np.float32), np.array([300, 3, 40, 1]), 5)
    assert idx.tolist() == [[0, 1, 2, 3, 4], [0, 1, 2, 2, 2], [35, 36, 37, 38, 39], [0, 0, 0, 0, 0]]
