"""numpy restatement of the reference's teacher/student hot path (TEST ORACLE).

Parity unpinned at the TensorFlow boundary (see ``oracle/__init__.py``).
Every function cites the reference call site it follows (paths relative to
``/root/reference/code_student_uniform/`` = ``cs/``) and, where the arithmetic
is TensorFlow's, the TF-1.x op whose published semantics it restates.

Default dtype is float64 (oracle of record); ``dtype=np.float32`` is used only
by ``bench.py``'s ``cpu_baseline`` leg to time the same algorithm at the
reference's precision.

Layout conventions are the reference's (TF): LSTM ``kernel`` is
``[in + H, 4H]`` with gate column blocks in the order i, j, f, o; MoE
``gates/weights`` is ``[K, V*(M+1)]`` and ``experts/weights`` ``[K, V*M]``
with column = class*(M+1)+m / class*M+m.
"""
from __future__ import annotations

import math
from typing import Dict, List, Sequence, Tuple

import numpy as np

# --------------------------------------------------------------------------
# small helpers
# --------------------------------------------------------------------------


def sigmoid(x):
    """tf.nn.sigmoid (numerically stable form)."""
    out = np.empty_like(x)
    pos = x >= 0
    out[pos] = 1.0 / (1.0 + np.exp(-x[pos]))
    ex = np.exp(x[~pos])
    out[~pos] = ex / (1.0 + ex)
    return out


def l2_normalize(x, axis=-1, epsilon=1e-12):
    """tf.nn.l2_normalize(x, dim) as called at cs/train.py:256.

    x * rsqrt(max(sum(x^2, dim), epsilon)).
    """
    ss = np.sum(np.square(x), axis=axis, keepdims=True)
    return x / np.sqrt(np.maximum(ss, epsilon))


def dequantize(q, max_q=2.0, min_q=-2.0):
    """cs/utils.py:10-25 Dequantize: q*(range/255) + (range/512 + min)."""
    rng = max_q - min_q
    return q * (rng / 255.0) + (rng / 512.0 + min_q)


# --------------------------------------------------------------------------
# every_n frame sub-sampling (integer-exact part of the path)
# --------------------------------------------------------------------------


def every_n_indices(every_n: int, max_frames_before_sampling: int = 300) -> List[int]:
    """cs/train.py:265-269: ``while every_n*k <= 299: append(every_n*k)``."""
    idx, k = [], 0
    while every_n * k <= max_frames_before_sampling - 1:
        idx.append(every_n * k)
        k += 1
    return idx


def student_max_frames(every_n: int, max_frames_before_sampling: int = 300) -> int:
    """cs/train.py:262-263: ``int(300/every_n)`` (py2 int division, then int())."""
    return int(max_frames_before_sampling // every_n)


def student_num_frames(num_frames, every_n: int) -> np.ndarray:
    """cs/train.py:264.

    ``tf.cast(tf.multiply(tf.divide(num_frames, 300), S), tf.int64)``:
    tf.divide on int32 is true division in float64; the product is float64;
    the cast truncates toward zero.  Implemented literally (it differs from
    floor(n*S/300) for some every_n).
    """
    s = student_max_frames(every_n)
    n = np.asarray(num_frames).astype(np.float64)
    return np.trunc((n / np.float64(300)) * np.float64(s)).astype(np.int64)


def subsample_frames(x, every_n: int):
    """cs/train.py:270-272: transpose -> gather(list_index_to_retain) -> transpose."""
    return x[:, every_n_indices(every_n), :]


def validate_every_n(every_n: int, num_inputs_l1: int = 5, max_frames: int = 300):
    """Admissible every_n.

    The index list has ceil(300/n) entries but the student graph is built for
    S=int(300/n) frames and tf.split(.., 5) needs S % 5 == 0
    (cs/frame_level_models.py:286,307); anything else fails at graph build.
    """
    if every_n <= 0:
        raise ValueError("every_n must be positive")
    n_idx = len(every_n_indices(every_n, max_frames))
    s = student_max_frames(every_n, max_frames)
    if n_idx != s or s % num_inputs_l1 != 0:
        raise ValueError(
            "every_n=%d is not admissible: %d gathered frames vs S=%d, S %% %d = %d"
            % (every_n, n_idx, s, num_inputs_l1, s % num_inputs_l1 if s else -1))


# --------------------------------------------------------------------------
# LSTM (tf.contrib.rnn.BasicLSTMCell / MultiRNNCell / tf.nn.dynamic_rnn)
# --------------------------------------------------------------------------


def lstm_cell_fwd(x, h, c, kernel, bias, forget_bias=1.0):
    """BasicLSTMCell(num_units, forget_bias=1.0, state_is_tuple=False) step.

    cs/frame_level_models.py:223-224.  z = [x,h] @ kernel + bias;
    i,j,f,o = split(z,4); c' = c*sigmoid(f+fb) + sigmoid(i)*tanh(j);
    h' = tanh(c')*sigmoid(o).
    """
    z = np.concatenate([x, h], axis=1) @ kernel + bias
    H = h.shape[1]
    i = sigmoid(z[:, 0 * H:1 * H])
    j = np.tanh(z[:, 1 * H:2 * H])
    f = sigmoid(z[:, 2 * H:3 * H] + forget_bias)
    o = sigmoid(z[:, 3 * H:4 * H])
    c_new = c * f + i * j
    tc = np.tanh(c_new)
    h_new = tc * o
    return h_new, c_new, (i, j, f, o, tc)


def multi_rnn_seq_fwd(x, lengths, layers, keep_cache=True):
    """dynamic_rnn(MultiRNNCell([BasicLSTMCell]*L, state_is_tuple=False), x,
    sequence_length=lengths, dtype=f32) -> final state.

    cs/frame_level_models.py:221-227,247-249.  Zero initial state; for
    t >= lengths[b] the state row is copied through; the returned state is
    concat([c0,h0,c1,h1,...], 1).

    x: [B,T,F]; lengths: [B] ints; layers: list of (kernel [in+H,4H], bias [4H]).
    Returns (state [B, 2*L*H], cache).
    """
    B, T, _ = x.shape
    L = len(layers)
    H = layers[0][1].shape[0] // 4
    dt = x.dtype
    c = [np.zeros((B, H), dt) for _ in range(L)]
    h = [np.zeros((B, H), dt) for _ in range(L)]
    lengths = np.asarray(lengths)
    cache = []
    for t in range(T):
        active = (t < lengths)[:, None]
        inp = x[:, t, :]
        step = []
        for l, (kernel, bias) in enumerate(layers):
            h_new, c_new, gates = lstm_cell_fwd(inp, h[l], c[l], kernel, bias)
            if keep_cache:
                step.append((inp, h[l], c[l], gates))
            c[l] = np.where(active, c_new, c[l])
            h[l] = np.where(active, h_new, h[l])
            inp = h_new
        cache.append(step)
    state = np.concatenate([s for l in range(L) for s in (c[l], h[l])], axis=1)
    return state, (cache, lengths, x.shape, H)


def multi_rnn_seq_bwd(dstate, cache, layers, need_dx=True):
    """Reverse-mode of multi_rnn_seq_fwd (what tf.gradients builds for the
    while-loop of dynamic_rnn).  Returns (dx [B,T,F] or None, [(dkernel, dbias)])."""
    steps, lengths, xshape, H = cache
    B, T, F = xshape
    L = len(layers)
    dt = dstate.dtype
    dc = [dstate[:, (2 * l) * H:(2 * l + 1) * H].copy() for l in range(L)]
    dh = [dstate[:, (2 * l + 1) * H:(2 * l + 2) * H].copy() for l in range(L)]
    grads = [(np.zeros_like(k), np.zeros_like(b)) for k, b in layers]
    dx = np.zeros((B, T, F), dt) if need_dx else None
    for t in range(T - 1, -1, -1):
        active = (t < lengths)[:, None].astype(dt)
        d_above = None
        for l in range(L - 1, -1, -1):
            inp, h_prev, c_prev, (i, j, f, o, tc) = steps[t][l]
            kernel = layers[l][0]
            dh_new = active * dh[l]
            if d_above is not None:
                dh_new = dh_new + active * d_above
            dc_new = active * dc[l] + dh_new * o * (1.0 - tc * tc)
            do = dh_new * tc
            di = dc_new * j
            dj = dc_new * i
            df = dc_new * c_prev
            dz = np.concatenate(
                [di * i * (1 - i), dj * (1 - j * j), df * f * (1 - f), do * o * (1 - o)], axis=1)
            xin = np.concatenate([inp, h_prev], axis=1)
            grads[l][0][...] += xin.T @ dz
            grads[l][1][...] += dz.sum(axis=0)
            dxin = dz @ kernel.T
            nin = inp.shape[1]
            d_above = dxin[:, :nin]
            dh[l] = (1.0 - active) * dh[l] + dxin[:, nin:]
            dc[l] = (1.0 - active) * dc[l] + dc_new * f
        if need_dx:
            dx[:, t, :] = d_above
    return dx, grads


# --------------------------------------------------------------------------
# HierarchicalLstmModel (cs/frame_level_models.py:198-338)
# --------------------------------------------------------------------------


def hlstm_chunk_lengths(num_frames, num_chunks: int, chunk_len: int):
    """Per-chunk L1 lengths and the L2 length.

    Teacher: cs/frame_level_models.py:238-240  min(L, max(0, n - int(L)*i)).
    Student: cs/frame_level_models.py:308-310  same in int64.
    L2:      :256 / :327   int32(ceil(float32(n) / L)).
    Returns (len_l1 [B, C] int64, len_l2 [B] int32).
    """
    n = np.asarray(num_frames).astype(np.int64)
    i = np.arange(num_chunks, dtype=np.int64)[None, :]
    len_l1 = np.minimum(chunk_len, np.maximum(0, n[:, None] - chunk_len * i))
    len_l2 = np.ceil(n.astype(np.float32) / np.float32(chunk_len)).astype(np.int32)
    return len_l1, len_l2


def hlstm_fwd(x, num_frames, params: Dict[str, np.ndarray], num_chunks: int,
              num_layers: int = 2, num_mixtures: int = 2, keep_cache=True):
    """HierarchicalLstmModel.create_model / create_model_inference.

    cs/frame_level_models.py:200-267 (teacher: num_chunks = num_inputs_to_lstm,
    chunk_len = max_num_frames/num_chunks) and :269-338 (student: num_chunks =
    num_inputs_L1 = 5, chunk_len = (max_num_frames/every_n)/5), py2 int division.
    x: [B, T, F] with T = num_chunks*chunk_len.  The C weight-shared chunk loops
    (:243-250) are folded into the batch dimension (row = b*C + chunk), which is
    the same arithmetic per row.
    Returns (state [B, 2*L*H], predictions [B, V], cache).
    """
    B, T, F = x.shape
    assert T % num_chunks == 0
    Lc = T // num_chunks
    len_l1, len_l2 = hlstm_chunk_lengths(num_frames, num_chunks, Lc)
    l1 = [(params["RNN_L1/rnn/multi_rnn_cell/cell_%d/basic_lstm_cell/kernel" % l],
           params["RNN_L1/rnn/multi_rnn_cell/cell_%d/basic_lstm_cell/bias" % l])
          for l in range(num_layers)]
    l2 = [(params["RNN_L2/rnn/multi_rnn_cell/cell_%d/basic_lstm_cell/kernel" % l],
           params["RNN_L2/rnn/multi_rnn_cell/cell_%d/basic_lstm_cell/bias" % l])
          for l in range(num_layers)]
    xc = x.reshape(B * num_chunks, Lc, F)
    s1, cache1 = multi_rnn_seq_fwd(xc, len_l1.reshape(-1), l1, keep_cache)
    l2_in = s1.reshape(B, num_chunks, s1.shape[1])               # tf.stack(L1_outputs, axis=1) :252
    state, cache2 = multi_rnn_seq_fwd(l2_in, len_l2, l2, keep_cache)
    pred, cache_moe = moe_fwd(state, params["classifier/gates/weights"],
                              params["classifier/experts/weights"],
                              params["classifier/experts/biases"], num_mixtures)
    return state, pred, (cache1, cache2, cache_moe, l1, l2, (B, num_chunks))


def hlstm_bwd(dstate, dpred, cache, num_layers: int = 2):
    """Gradients of hlstm_fwd wrt its 11 parameters given d(state), d(pred)."""
    cache1, cache2, cache_moe, l1, l2, (B, C) = cache
    dx_moe, dWg, dWe, dbe = moe_bwd(dpred, cache_moe)
    ds = dx_moe if dstate is None else dstate + dx_moe
    dl2_in, g2 = multi_rnn_seq_bwd(ds, cache2, l2, need_dx=True)
    _, g1 = multi_rnn_seq_bwd(dl2_in.reshape(B * C, -1), cache1, l1, need_dx=False)
    grads = {}
    for l in range(num_layers):
        grads["RNN_L1/rnn/multi_rnn_cell/cell_%d/basic_lstm_cell/kernel" % l] = g1[l][0]
        grads["RNN_L1/rnn/multi_rnn_cell/cell_%d/basic_lstm_cell/bias" % l] = g1[l][1]
        grads["RNN_L2/rnn/multi_rnn_cell/cell_%d/basic_lstm_cell/kernel" % l] = g2[l][0]
        grads["RNN_L2/rnn/multi_rnn_cell/cell_%d/basic_lstm_cell/bias" % l] = g2[l][1]
    grads["classifier/gates/weights"] = dWg
    grads["classifier/experts/weights"] = dWe
    grads["classifier/experts/biases"] = dbe
    return grads


def hlstm_fwd_unfolded(x, num_frames, params, num_chunks, num_layers=2, num_mixtures=2):
    """Literal (un-folded) restatement of cs/frame_level_models.py:237-257:
    C separate dynamic_rnn loops at batch B.  Used only to check the folded
    version on small cases."""
    B, T, F = x.shape
    Lc = T // num_chunks
    len_l1, len_l2 = hlstm_chunk_lengths(num_frames, num_chunks, Lc)
    l1 = [(params["RNN_L1/rnn/multi_rnn_cell/cell_%d/basic_lstm_cell/kernel" % l],
           params["RNN_L1/rnn/multi_rnn_cell/cell_%d/basic_lstm_cell/bias" % l])
          for l in range(num_layers)]
    l2 = [(params["RNN_L2/rnn/multi_rnn_cell/cell_%d/basic_lstm_cell/kernel" % l],
           params["RNN_L2/rnn/multi_rnn_cell/cell_%d/basic_lstm_cell/bias" % l])
          for l in range(num_layers)]
    outs = []
    for i in range(num_chunks):
        s, _ = multi_rnn_seq_fwd(x[:, i * Lc:(i + 1) * Lc, :], len_l1[:, i], l1, False)
        outs.append(s)
    l2_in = np.stack(outs, axis=1)
    state, _ = multi_rnn_seq_fwd(l2_in, len_l2, l2, False)
    pred, _ = moe_fwd(state, params["classifier/gates/weights"],
                      params["classifier/experts/weights"],
                      params["classifier/experts/biases"], num_mixtures)
    return state, pred


# --------------------------------------------------------------------------
# MoeModel (cs/video_level_models.py:394-448)
# --------------------------------------------------------------------------


def moe_fwd(x, Wg, We, be, num_mixtures=2):
    """gates = x@Wg (no bias) -> softmax over M+1; experts = x@We+be -> sigmoid;
    p = sum_{m<M} g_m * e_m.  cs/video_level_models.py:423-448."""
    B = x.shape[0]
    M = num_mixtures
    V = We.shape[1] // M
    ga = (x @ Wg).reshape(B * V, M + 1)
    ea = (x @ We + be).reshape(B * V, M)
    ga = ga - ga.max(axis=1, keepdims=True)
    eg = np.exp(ga)
    g = eg / eg.sum(axis=1, keepdims=True)
    e = sigmoid(ea)
    p = (g[:, :M] * e).sum(axis=1).reshape(B, V)
    return p, (x, Wg, We, g, e, M, V)


def moe_bwd(dp, cache):
    x, Wg, We, g, e, M, V = cache
    B = x.shape[0]
    dpf = dp.reshape(B * V, 1)
    de = dpf * g[:, :M]
    dea = de * e * (1 - e)
    dg = np.zeros_like(g)
    dg[:, :M] = dpf * e
    dga = g * (dg - (dg * g).sum(axis=1, keepdims=True))
    dga = dga.reshape(B, V * (M + 1))
    dea = dea.reshape(B, V * M)
    dWg = x.T @ dga
    dWe = x.T @ dea
    dbe = dea.sum(axis=0)
    dx = dga @ Wg.T + dea @ We.T
    return dx, dWg, dWe, dbe


def moe_regularization(Wg, We, l2_penalty=1e-8):
    """slim.l2_regularizer(1e-8) on gates and experts weights
    (cs/video_level_models.py:428,434): s * sum(w^2)/2 each."""
    return l2_penalty * 0.5 * (np.sum(np.square(Wg)) + np.sum(np.square(We)))


# --------------------------------------------------------------------------
# Losses
# --------------------------------------------------------------------------

CE_EPSILON = 10e-6  # cs/losses.py:92 (= 1e-5)


def cross_entropy_loss(p, y):
    """CrossEntropyLoss.calculate_loss, cs/losses.py:90-97."""
    y = y.astype(p.dtype)
    ce = -(y * np.log(p + CE_EPSILON) + (1 - y) * np.log(1 - p + CE_EPSILON))
    return np.mean(np.sum(ce, axis=1))


def cross_entropy_grad(p, y):
    y = y.astype(p.dtype)
    B = p.shape[0]
    return -(y / (p + CE_EPSILON) - (1 - y) / (1 - p + CE_EPSILON)) / B


def rep_loss(teacher_state, student_state):
    """L_REP, cs/train.py:359-362: mean_b sum_d (sT - sS)^2."""
    return np.mean(np.sum(np.square(teacher_state - student_state), axis=1))


def rep_loss_grad_student(teacher_state, student_state):
    B = teacher_state.shape[0]
    return -2.0 * (teacher_state - student_state) / B


def pred_kl_loss(p_teacher, p_student):
    """L_PRED, cs/train.py:398-402.

    tf.distributions.Categorical(probs=p) takes logits = log(p); kl_divergence
    = sum_c softmax(log p)_c (log_softmax(log p)_c - log_softmax(log q)_c), i.e.
    KL between the renormalised p/sum(p) and q/sum(q) per row; reduce_sum over
    the batch."""
    P = p_teacher / p_teacher.sum(axis=1, keepdims=True)
    Q = p_student / p_student.sum(axis=1, keepdims=True)
    return np.sum(P * (np.log(P) - np.log(Q)))


def pred_kl_grad_student(p_teacher, p_student):
    P = p_teacher / p_teacher.sum(axis=1, keepdims=True)
    sq = p_student.sum(axis=1, keepdims=True)
    return -P / p_student + 1.0 / sq


# --------------------------------------------------------------------------
# Optimiser (slim.learning.create_train_op + tf.train.AdamOptimizer)
# --------------------------------------------------------------------------


def clip_by_norm(g, clip_norm):
    """tf.clip_by_norm per tensor (slim create_train_op(clip_gradient_norm=c),
    cs/train.py:329-334): g * c / max(||g||_2, c)."""
    n = np.sqrt(np.sum(np.square(g)))
    return g * (clip_norm / max(n, clip_norm))


def exponential_decay(lr0, global_step, batch_size, decay_examples, rate):
    """tf.train.exponential_decay(lr0, global_step*B, decay_examples, rate,
    staircase=True), cs/train.py:223-236."""
    return lr0 * rate ** math.floor(global_step * batch_size / decay_examples)


def adam_step(p, g, m, v, t, lr, beta1=0.9, beta2=0.999, eps=1e-8):
    """tf.train.AdamOptimizer update (epsilon outside the bias correction).
    t is the 1-based step count of this optimizer.  m and v are updated in
    place (they are this optimizer's slots); returns (new p, m, v)."""
    one = p.dtype.type(1.0)
    m *= p.dtype.type(beta1)
    m += (one - p.dtype.type(beta1)) * g
    v *= p.dtype.type(beta2)
    gg = g * g
    gg *= one - p.dtype.type(beta2)
    v += gg
    lr_t = lr * math.sqrt(1 - beta2 ** t) / (1 - beta1 ** t)
    den = np.sqrt(v, out=gg)
    den += p.dtype.type(eps)
    np.divide(m, den, out=den)
    den *= p.dtype.type(lr_t)
    return p - den, m, v


# --------------------------------------------------------------------------
# FrameLevelLogisticModel (cs/frame_level_models.py:50-83)
# --------------------------------------------------------------------------


def logistic_fwd(x, num_frames, W, b):
    """sum over all (padded) frames / true n, then sigmoid(avg @ W + b)."""
    n = np.asarray(num_frames).astype(x.dtype)[:, None]
    avg = x.sum(axis=1) / n
    return sigmoid(avg @ W + b), avg


def logistic_bwd(dp, p, avg):
    dz = dp * p * (1 - p)
    return avg.T @ dz, dz.sum(axis=0)


# --------------------------------------------------------------------------
# DbofModel (cs/frame_level_models.py:85-195, cs/model_utils.py:39-83)
# --------------------------------------------------------------------------

BN_EPSILON = 1e-3   # slim.batch_norm default
BN_DECAY = 0.999    # slim.batch_norm default


def sample_random_frames_index(uniform, num_frames):
    """cs/model_utils.py:50-54: int32(U[0,1) * float32(n)) (truncation).
    ``uniform`` [B,S] float32 is injected so CPU and GPU use the same draw."""
    u = np.asarray(uniform, np.float32)
    n = np.asarray(num_frames).astype(np.float32)[:, None]
    return (u * n).astype(np.int32)


def batch_norm_train_fwd(x, gamma, beta):
    """slim.batch_norm(center, scale, is_training=True): biased batch moments
    over axis 0, epsilon 1e-3."""
    mu = x.mean(axis=0)
    var = x.var(axis=0)
    inv = 1.0 / np.sqrt(var + BN_EPSILON)
    xh = (x - mu) * inv
    return xh * gamma + beta, (xh, inv, gamma, mu, var)


def batch_norm_train_bwd(dy, cache):
    xh, inv, gamma, _, _ = cache
    N = dy.shape[0]
    dgamma = (dy * xh).sum(axis=0)
    dbeta = dy.sum(axis=0)
    dxh = dy * gamma
    dx = inv / N * (N * dxh - dxh.sum(axis=0) - xh * (dxh * xh).sum(axis=0))
    return dx, dgamma, dbeta


def relu6(x):
    return np.minimum(np.maximum(x, 0.0), 6.0)


def dbof_fwd(x, num_frames, uniform, params, num_mixtures=2):
    """DbofModel.create_model with add_batch_norm=True, sample_random_frames=True,
    pooling 'max', is_training=True (cs/frame_level_models.py:108-195)."""
    B = x.shape[0]
    idx = sample_random_frames_index(uniform, num_frames)           # model_utils.py:50-54
    S = idx.shape[1]
    g = x[np.arange(B)[:, None], idx, :]                              # gather_nd :55-58
    F = g.shape[2]
    r = g.reshape(B * S, F)
    r_bn, c_in = batch_norm_train_fwd(r, params["input_bn/gamma"], params["input_bn/beta"])
    act = r_bn @ params["cluster_weights"]
    act_bn, c_cl = batch_norm_train_fwd(act, params["cluster_bn/gamma"], params["cluster_bn/beta"])
    a6 = relu6(act_bn)
    C = a6.shape[1]
    a3 = a6.reshape(B, S, C)
    am = a3.argmax(axis=1)
    pooled = a3.max(axis=1)                                          # FramePooling 'max' model_utils.py:77-78
    hid = pooled @ params["hidden1_weights"]
    hid_bn, c_h = batch_norm_train_fwd(hid, params["hidden1_bn/gamma"], params["hidden1_bn/beta"])
    h6 = relu6(hid_bn)
    pred, c_moe = moe_fwd(h6, params["classifier/gates/weights"], params["classifier/experts/weights"],
                          params["classifier/experts/biases"], num_mixtures)
    cache = (idx, r, r_bn, c_in, act_bn, c_cl, am, pooled, hid_bn, c_h, h6, c_moe, (B, S, C), params)
    return pred, cache


def dbof_bwd(dpred, cache, routing=None):
    """routing (tests only): (argmax [B, C] int, cluster relu6 mask at the selected entries [B, C] bool, hidden relu6 mask [B, Hd]
    bool) - the discrete decisions of ANOTHER implementation of the same forward.  relu6 kinks and max-pool ties are decided on
    values that differ by that implementation's rounding; with its decisions imposed the two gradients are the same smooth
    function and can be compared tightly (the decisions themselves are compared separately, against the margins)."""
    idx, r, r_bn, c_in, act_bn, c_cl, am, pooled, hid_bn, c_h, h6, c_moe, (B, S, C), params = cache
    g = {}
    dh6, g["classifier/gates/weights"], g["classifier/experts/weights"], g["classifier/experts/biases"] = \
        moe_bwd(dpred, c_moe)
    if routing is not None:
        am_r, mask_sel, mask_h = routing
        dhid_bn = dh6 * mask_h
        dhid, g["hidden1_bn/gamma"], g["hidden1_bn/beta"] = batch_norm_train_bwd(dhid_bn, c_h)
        g["hidden1_weights"] = pooled.T @ dhid
        dpooled = dhid @ params["hidden1_weights"].T
        da3 = np.zeros((B, S, C), dpooled.dtype)
        bi, ci = np.meshgrid(np.arange(B), np.arange(C), indexing="ij")
        da3[bi, am_r, ci] = dpooled * mask_sel
        dact_bn = da3.reshape(B * S, C)
        dact, g["cluster_bn/gamma"], g["cluster_bn/beta"] = batch_norm_train_bwd(dact_bn, c_cl)
        g["cluster_weights"] = r_bn.T @ dact
        dr_bn = dact @ params["cluster_weights"].T
        _, g["input_bn/gamma"], g["input_bn/beta"] = batch_norm_train_bwd(dr_bn, c_in)
        return g
    dhid_bn = dh6 * ((hid_bn > 0) & (hid_bn < 6))
    dhid, g["hidden1_bn/gamma"], g["hidden1_bn/beta"] = batch_norm_train_bwd(dhid_bn, c_h)
    g["hidden1_weights"] = pooled.T @ dhid
    dpooled = dhid @ params["hidden1_weights"].T
    da3 = np.zeros((B, S, C), dpooled.dtype)
    bi, ci = np.meshgrid(np.arange(B), np.arange(C), indexing="ij")
    da3[bi, am, ci] = dpooled
    da6 = da3.reshape(B * S, C)
    dact_bn = da6 * ((act_bn > 0) & (act_bn < 6))
    dact, g["cluster_bn/gamma"], g["cluster_bn/beta"] = batch_norm_train_bwd(dact_bn, c_cl)
    g["cluster_weights"] = r_bn.T @ dact
    dr_bn = dact @ params["cluster_weights"].T
    _, g["input_bn/gamma"], g["input_bn/beta"] = batch_norm_train_bwd(dr_bn, c_in)
    return g


# --------------------------------------------------------------------------
# NetVLAD (EXTENSION: the reference's NetVLADModel is an empty stub, cs/frame_level_models.py:341-347 - there is no
# reference math for it.  This is the NetVLAD aggregation of "Learnable pooling with Context Gating" (Miech et al.) as
# the YouTube-8M starter code lineage implements it, in the frame of this repository's DBoF tower: sampled frames ->
# input_bn -> cluster assignment (matmul, cluster_bn, softmax) -> residual aggregation against learned centres ->
# intra-normalisation -> l2 normalisation -> hidden layer (matmul, hidden1_bn, relu6) -> MoE.)
# --------------------------------------------------------------------------
def sample_random_sequence_index(uniform, num_frames, num_samples):
    """SampleRandomSequence (cs/model_utils.py:11-36, reached with --sample_random_frames False): num_samples consecutive frames
    from start = int32(U[0,1) * float32(max(n - num_samples, 0) + 1)), each index clipped at n - 1 (:31-32).  ``uniform`` [B]
    (the tf.random_uniform([batch_size, 1]) draw) is injected."""
    u = np.asarray(uniform, np.float32).reshape(-1)
    n = np.asarray(num_frames).astype(np.int64)
    mx = np.maximum(n - num_samples, 0)
    start = (u * (mx + 1).astype(np.float32)).astype(np.int32).astype(np.int64)
    return np.minimum(start[:, None] + np.arange(num_samples)[None, :], (n - 1)[:, None]).astype(np.int32)


def dbof_general_fwd(x, num_frames, uniform, params, pooling="max", add_batch_norm=True, random_frames=True, num_mixtures=2):
    """DbofModel.create_model (cs/frame_level_models.py:108-195) with every flag combination the reference's graph can train:
    sample_random_frames True (SampleRandomFrames) / False (SampleRandomSequence, uniform [B]); dbof_add_batch_norm True (three
    slim.batch_norm) / False (cluster_biases / hidden1_biases, :158-161,181-185); dbof_pooling_method 'max' / 'average'
    (cs/model_utils.py:75-78).  ('none' returns [B*S, C]: its predictions have B*S rows against B label rows - the graph does not
    train.)  Training mode (batch statistics)."""
    B = x.shape[0]
    S = np.asarray(uniform).shape[1] if random_frames else params["_iterations"]
    idx = sample_random_frames_index(uniform, num_frames) if random_frames else sample_random_sequence_index(uniform, num_frames, S)
    r = x[np.arange(B)[:, None], idx, :].reshape(B * S, x.shape[2])
    c_in = c_cl = c_h = None
    if add_batch_norm:
        r_bn, c_in = batch_norm_train_fwd(r, params["input_bn/gamma"], params["input_bn/beta"])
        act = r_bn @ params["cluster_weights"]
        pre, c_cl = batch_norm_train_fwd(act, params["cluster_bn/gamma"], params["cluster_bn/beta"])
    else:
        r_bn = r
        pre = r @ params["cluster_weights"] + params["cluster_biases"]
    a3 = relu6(pre).reshape(B, S, -1)
    am = a3.argmax(axis=1)
    pooled = a3.max(axis=1) if pooling == "max" else a3.mean(axis=1)
    hid = pooled @ params["hidden1_weights"]
    if add_batch_norm:
        hpre, c_h = batch_norm_train_fwd(hid, params["hidden1_bn/gamma"], params["hidden1_bn/beta"])
    else:
        hpre = hid + params["hidden1_biases"]
    h6 = relu6(hpre)
    pred, c_moe = moe_fwd(h6, params["classifier/gates/weights"], params["classifier/experts/weights"],
                          params["classifier/experts/biases"], num_mixtures)
    return pred, (idx, r_bn, c_in, pre, c_cl, am, pooled, hpre, c_h, c_moe, (B, S), params, pooling, add_batch_norm)


def dbof_general_bwd(dpred, cache, routing=None):
    """routing (tests only, as in dbof_bwd): (argmax [B, C] int or None, cluster relu6 mask [B*S, C] bool, hidden relu6 mask [B, Hd]
    bool) - the discrete decisions of another implementation of the same forward, imposed so that both gradients are the same
    smooth function (the decisions themselves are compared separately)."""
    idx, r_bn, c_in, pre, c_cl, am, pooled, hpre, c_h, c_moe, (B, S), params, pooling, bn = cache
    mask_c, mask_h = (pre > 0) & (pre < 6), (hpre > 0) & (hpre < 6)
    if routing is not None:
        am = routing[0] if routing[0] is not None else am
        mask_c, mask_h = routing[1], routing[2]
    g = {}
    dh6, g["classifier/gates/weights"], g["classifier/experts/weights"], g["classifier/experts/biases"] = moe_bwd(dpred, c_moe)
    dhpre = dh6 * mask_h
    if bn:
        dhid, g["hidden1_bn/gamma"], g["hidden1_bn/beta"] = batch_norm_train_bwd(dhpre, c_h)
    else:
        dhid, g["hidden1_biases"] = dhpre, dhpre.sum(0)
    g["hidden1_weights"] = pooled.T @ dhid
    dpooled = dhid @ params["hidden1_weights"].T
    C = dpooled.shape[1]
    if pooling == "max":
        da3 = np.zeros((B, S, C), dpooled.dtype)
        bi, ci = np.meshgrid(np.arange(B), np.arange(C), indexing="ij")
        da3[bi, am, ci] = dpooled
    else:
        da3 = np.repeat(dpooled[:, None, :] / S, S, axis=1)
    dpre = da3.reshape(B * S, C) * mask_c
    if bn:
        dact, g["cluster_bn/gamma"], g["cluster_bn/beta"] = batch_norm_train_bwd(dpre, c_cl)
    else:
        dact, g["cluster_biases"] = dpre, dpre.sum(0)
    g["cluster_weights"] = r_bn.T @ dact
    if bn:
        _, g["input_bn/gamma"], g["input_bn/beta"] = batch_norm_train_bwd(dact @ params["cluster_weights"].T, c_in)
    return g


def netvlad_fwd(x, num_frames, uniform, params, num_mixtures=2):
    B = x.shape[0]
    idx = sample_random_frames_index(uniform, num_frames)
    S = idx.shape[1]
    g = x[np.arange(B)[:, None], idx, :]
    F = g.shape[2]
    r = g.reshape(B * S, F)
    r_bn, c_in = batch_norm_train_fwd(r, params["input_bn/gamma"], params["input_bn/beta"])
    Wc = params["cluster_weights"]                                   # [F, K]
    K = Wc.shape[1]
    act = r_bn @ Wc
    act_bn, c_cl = batch_norm_train_fwd(act, params["cluster_bn/gamma"], params["cluster_bn/beta"])
    z = act_bn - act_bn.max(axis=1, keepdims=True)
    e = np.exp(z)
    a = e / e.sum(axis=1, keepdims=True)                             # soft assignment [B*S, K]
    a3 = a.reshape(B, S, K)
    X = r_bn.reshape(B, S, F)
    a_sum = a3.sum(axis=1)                                           # [B, K]
    C2 = params["cluster_weights2"]                                  # [F, K] (tf: [1, F, K])
    V = np.einsum("bsk,bsf->bkf", a3, X) - a_sum[:, :, None] * C2.T[None]          # [B, K, F]
    n1 = np.sqrt(np.maximum((V * V).sum(axis=2), 1e-12))             # tf.nn.l2_normalize over the feature axis, per cluster
    U = V / n1[:, :, None]
    Uf = U.transpose(0, 2, 1).reshape(B, F * K)                      # reshape of the [B, F, K] tensor: index f*K + k
    n2 = np.sqrt(np.maximum((Uf * Uf).sum(axis=1), 1e-12))
    Y = Uf / n2[:, None]
    hid = Y @ params["hidden1_weights"]                              # [F*K, H]
    hid_bn, c_h = batch_norm_train_fwd(hid, params["hidden1_bn/gamma"], params["hidden1_bn/beta"])
    h6 = relu6(hid_bn)
    pred, c_moe = moe_fwd(h6, params["classifier/gates/weights"], params["classifier/experts/weights"],
                          params["classifier/experts/biases"], num_mixtures)
    cache = (idx, r_bn, c_in, a3, X, a_sum, C2, V, n1, U, n2, Y, hid_bn, c_h, h6, c_moe, c_cl, (B, S, F, K), params)
    return pred, cache


def netvlad_bwd(dpred, cache):
    idx, r_bn, c_in, a3, X, a_sum, C2, V, n1, U, n2, Y, hid_bn, c_h, h6, c_moe, c_cl, (B, S, F, K), params = cache
    g = {}
    dh6, g["classifier/gates/weights"], g["classifier/experts/weights"], g["classifier/experts/biases"] = moe_bwd(dpred, c_moe)
    dhid_bn = dh6 * ((hid_bn > 0) & (hid_bn < 6))
    dhid, g["hidden1_bn/gamma"], g["hidden1_bn/beta"] = batch_norm_train_bwd(dhid_bn, c_h)
    g["hidden1_weights"] = Y.T @ dhid
    dY = dhid @ params["hidden1_weights"].T                          # [B, F*K]
    dUf = (dY - Y * (Y * dY).sum(axis=1, keepdims=True)) / n2[:, None]
    dU = dUf.reshape(B, F, K).transpose(0, 2, 1)                     # [B, K, F]
    dV = (dU - U * (U * dU).sum(axis=2, keepdims=True)) / n1[:, :, None]
    g["cluster_weights2"] = -np.einsum("bk,bkf->fk", a_sum, dV)
    da3 = np.einsum("bkf,bsf->bsk", dV, X) - np.einsum("bkf,fk->bk", dV, C2)[:, None, :]
    dX = np.einsum("bsk,bkf->bsf", a3, dV)
    a = a3.reshape(B * S, K)
    da = da3.reshape(B * S, K)
    dz = a * (da - (a * da).sum(axis=1, keepdims=True))              # softmax backward
    dact, g["cluster_bn/gamma"], g["cluster_bn/beta"] = batch_norm_train_bwd(dz, c_cl)
    g["cluster_weights"] = r_bn.T @ dact
    dr_bn = dact @ params["cluster_weights"].T + dX.reshape(B * S, F)
    _, g["input_bn/gamma"], g["input_bn/beta"] = batch_norm_train_bwd(dr_bn, c_in)
    return g


def init_netvlad_params(rng, feature_size=1152, cluster_size=64, hidden_size=1024, vocab_size=4716, num_mixtures=2,
                        dtype=np.float64):
    F, K, H = feature_size, cluster_size, hidden_size
    return {
        "input_bn/gamma": np.ones(F, dtype), "input_bn/beta": np.zeros(F, dtype),
        "cluster_weights": (rng.standard_normal((F, K)) / math.sqrt(F)).astype(dtype),
        "cluster_bn/gamma": np.ones(K, dtype), "cluster_bn/beta": np.zeros(K, dtype),
        "cluster_weights2": (rng.standard_normal((F, K)) / math.sqrt(F)).astype(dtype),
        "hidden1_weights": (rng.standard_normal((F * K, H)) / math.sqrt(K)).astype(dtype),
        "hidden1_bn/gamma": np.ones(H, dtype), "hidden1_bn/beta": np.zeros(H, dtype),
        "classifier/gates/weights": glorot_uniform(rng, (H, vocab_size * (num_mixtures + 1)), dtype),
        "classifier/experts/weights": glorot_uniform(rng, (H, vocab_size * num_mixtures), dtype),
        "classifier/experts/biases": np.zeros(vocab_size * num_mixtures, dtype),
    }


def bn_moving_update(moving, batch_value, decay=BN_DECAY):
    """slim.batch_norm UPDATE_OPS: moving -= (1-decay)*(moving - batch)."""
    return moving - (1.0 - decay) * (moving - batch_value)


# --------------------------------------------------------------------------
# Parameter initialisation (TF defaults at the reference's call sites)
# --------------------------------------------------------------------------


def glorot_uniform(rng, shape, dtype=np.float64):
    fan_in, fan_out = shape[0], shape[1]
    lim = math.sqrt(6.0 / (fan_in + fan_out))
    return rng.uniform(-lim, lim, size=shape).astype(dtype)


def init_hlstm_params(rng, feature_size=1152, lstm_cells=1024, num_layers=2,
                      vocab_size=4716, num_mixtures=2, dtype=np.float64):
    """The 11 trainable variables of one tower in TF enumeration order
    (README.md:98,105) with TF-default initialisers:
    glorot-uniform kernels (variable-scope default), zero biases."""
    H = lstm_cells
    p = {}
    for scope, in0 in (("RNN_L1", feature_size), ("RNN_L2", 2 * num_layers * H)):
        for l in range(num_layers):
            nin = in0 if l == 0 else H
            base = "%s/rnn/multi_rnn_cell/cell_%d/basic_lstm_cell/" % (scope, l)
            p[base + "kernel"] = glorot_uniform(rng, (nin + H, 4 * H), dtype)
            p[base + "bias"] = np.zeros((4 * H,), dtype)
    K = 2 * num_layers * H
    p["classifier/gates/weights"] = glorot_uniform(rng, (K, vocab_size * (num_mixtures + 1)), dtype)
    p["classifier/experts/weights"] = glorot_uniform(rng, (K, vocab_size * num_mixtures), dtype)
    p["classifier/experts/biases"] = np.zeros((vocab_size * num_mixtures,), dtype)
    return p


HLSTM_PARAM_ORDER = (
    "RNN_L1/rnn/multi_rnn_cell/cell_0/basic_lstm_cell/kernel",
    "RNN_L1/rnn/multi_rnn_cell/cell_0/basic_lstm_cell/bias",
    "RNN_L1/rnn/multi_rnn_cell/cell_1/basic_lstm_cell/kernel",
    "RNN_L1/rnn/multi_rnn_cell/cell_1/basic_lstm_cell/bias",
    "RNN_L2/rnn/multi_rnn_cell/cell_0/basic_lstm_cell/kernel",
    "RNN_L2/rnn/multi_rnn_cell/cell_0/basic_lstm_cell/bias",
    "RNN_L2/rnn/multi_rnn_cell/cell_1/basic_lstm_cell/kernel",
    "RNN_L2/rnn/multi_rnn_cell/cell_1/basic_lstm_cell/bias",
    "classifier/gates/weights",
    "classifier/experts/weights",
    "classifier/experts/biases",
)


def init_dbof_params(rng, feature_size=1152, cluster_size=8192, hidden_size=1024,
                     vocab_size=4716, num_mixtures=2, dtype=np.float64):
    """cs/frame_level_models.py:138-180: random_normal(stddev=1/sqrt(fan_in))
    cluster/hidden weights; BN beta=0, gamma=1."""
    p = {
        "input_bn/gamma": np.ones(feature_size, dtype), "input_bn/beta": np.zeros(feature_size, dtype),
        "cluster_weights": (rng.standard_normal((feature_size, cluster_size)) / math.sqrt(feature_size)).astype(dtype),
        "cluster_bn/gamma": np.ones(cluster_size, dtype), "cluster_bn/beta": np.zeros(cluster_size, dtype),
        "hidden1_weights": (rng.standard_normal((cluster_size, hidden_size)) / math.sqrt(cluster_size)).astype(dtype),
        "hidden1_bn/gamma": np.ones(hidden_size, dtype), "hidden1_bn/beta": np.zeros(hidden_size, dtype),
        "classifier/gates/weights": glorot_uniform(rng, (hidden_size, vocab_size * (num_mixtures + 1)), dtype),
        "classifier/experts/weights": glorot_uniform(rng, (hidden_size, vocab_size * num_mixtures), dtype),
        "classifier/experts/biases": np.zeros(vocab_size * num_mixtures, dtype),
    }
    return p


# --------------------------------------------------------------------------
# The whole training iteration (cs/train.py:253-425, :516-517)
# --------------------------------------------------------------------------


def teacher_student_step(x_raw, num_frames, labels, teacher, student, every_n,
                         num_inputs_to_lstm=20, num_inputs_l1_student=5, num_layers=2,
                         num_mixtures=2, regularization_penalty=2.0, count_rep_twice=True,
                         with_grads=True):
    """One ``sess.run([train_op, train_student_op, ...])`` worth of math, up to
    (not including) the clip+Adam update.

    Returns a dict with the teacher/student states and predictions, the losses
    exactly as the reference logs them (cs/train.py:528-533) and, if
    ``with_grads``, the gradients of final_loss wrt the teacher's variables
    and of total_student_loss wrt the student's variables (teacher tensors are
    constants in the student loss: variables_to_train, cs/train.py:408-417).
    """
    validate_every_n(every_n, num_inputs_l1_student)
    x = l2_normalize(x_raw, axis=2)                                  # :256
    n_s = student_num_frames(num_frames, every_n)                    # :264
    x_s = subsample_frames(x, every_n)                               # :265-272
    y = labels.astype(x.dtype)

    t_state, t_pred, t_cache = hlstm_fwd(x, num_frames, teacher, num_inputs_to_lstm,
                                         num_layers, num_mixtures, keep_cache=with_grads)
    s_state, s_pred, s_cache = hlstm_fwd(x_s, n_s, student, num_inputs_l1_student,
                                         num_layers, num_mixtures, keep_cache=with_grads)
    out = {"x": x, "x_student": x_s, "num_frames_student": n_s,
           "teacher_state": t_state, "teacher_predictions": t_pred,
           "student_state": s_state, "student_predictions": s_pred}
    out["label_loss"] = cross_entropy_loss(t_pred, y)                               # :297
    out["reg_loss"] = moe_regularization(teacher["classifier/gates/weights"],
                                         teacher["classifier/experts/weights"])    # :305-307
    out["final_loss"] = regularization_penalty * out["reg_loss"] + out["label_loss"]  # :324
    out["student_loss_state"] = rep_loss(t_state, s_state)                          # :359-362
    out["student_label_loss"] = cross_entropy_loss(s_pred, y)                       # :372
    out["stud_reg_loss"] = moe_regularization(student["classifier/gates/weights"],
                                              student["classifier/experts/weights"])
    out["pred_loss"] = pred_kl_loss(t_pred, s_pred)                                 # :402
    rep_w = 2.0 if count_rep_twice else 1.0
    out["total_student_loss"] = (rep_w * out["student_loss_state"] + out["pred_loss"]
                                 + out["student_label_loss"]
                                 + regularization_penalty * out["stud_reg_loss"])   # :406
    if not with_grads:
        return out

    tg = hlstm_bwd(None, cross_entropy_grad(t_pred, y), t_cache, num_layers)
    for k in ("classifier/gates/weights", "classifier/experts/weights"):
        tg[k] = tg[k] + regularization_penalty * 1e-8 * teacher[k]
    ds = rep_w * rep_loss_grad_student(t_state, s_state)
    dp = pred_kl_grad_student(t_pred, s_pred) + cross_entropy_grad(s_pred, y)
    sg = hlstm_bwd(ds, dp, s_cache, num_layers)
    for k in ("classifier/gates/weights", "classifier/experts/weights"):
        sg[k] = sg[k] + regularization_penalty * 1e-8 * student[k]
    out["teacher_grads"] = tg
    out["student_grads"] = sg
    return out


def apply_train_op(params, grads, slots, t, lr, clip_norm=1.0):
    """slim create_train_op + Adam for one tower: per-tensor clip_by_norm then
    TF-Adam (cs/train.py:329-334).  ``slots`` maps name -> (m, v); mutated."""
    new = {}
    for k, p in params.items():
        g = clip_by_norm(grads[k], clip_norm) if clip_norm > 0 else grads[k]
        m, v = slots.get(k, (np.zeros_like(p), np.zeros_like(p)))
        new[k], m, v = adam_step(p, g, m, v, t, lr)
        slots[k] = (m, v)
    return new


# --------------------------------------------------------------------------
# Synthetic inputs - shared by tests and bench
# --------------------------------------------------------------------------


def synthetic_batch(batch, seed=1234, max_frames=300, feature_size=1152, vocab_size=4716,
                    all_full=False, dtype=np.float64):
    """uint8-uniform features dequantised as the reader does (cs/utils.py:22-25),
    n ~ U{120..300} with rows t >= n zeroed (cs/readers.py:170-173 pads after
    dequantisation), 3 uniformly drawn positives per video + class 0 w.p. 0.3."""
    rng = np.random.default_rng(seed)
    q = rng.integers(0, 256, size=(batch, max_frames, feature_size), dtype=np.uint8)
    n = (np.full(batch, max_frames) if all_full
         else rng.integers(min(120, max_frames), max_frames + 1, size=batch)).astype(np.int32)
    x = dequantize(q.astype(dtype)).astype(dtype)
    x[np.arange(max_frames)[None, :] >= n[:, None]] = 0.0
    labels = np.zeros((batch, vocab_size), bool)
    for b in range(batch):
        labels[b, rng.choice(vocab_size, size=min(3, vocab_size), replace=False)] = True
        if rng.random() < 0.3:
            labels[b, 0] = True
    return q, x, n, labels
