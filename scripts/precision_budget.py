"""Error budget of the forward pass per contraction (CPU experiment, no GPU needed).  ANALYSIS TOOL, not part of the product path: like the
tests it drives the oracle (oracle/model_math.py, oracle/torch_cpu.py) as the float64 reference.

Which products of the H-LSTM forward need split operands to hold north_star's 1e-3 on logits / states / predictions
at trained-magnitude weights?  Emulates the MFMA operand types on the CPU: both operands of a product are rounded to
the chosen 16-bit type (bf16 or f16), the contraction itself runs in float64 (the MFMA accumulates in f32; its own
rounding is ~1e-6 and not the question here); "x3" = operands kept exact (hi.hi + hi.lo + lo.hi leaves ~2^-17).
Products: L1 cell_0, L1 cell_1 (the teacher's 300 / student's 30 frame steps: 85 % of the forward flops), L2 cell_0
(hoisted input projection + recurrent part), L2 cell_1, the MoE head.

Weights of trained magnitude as in tests/test_gpu_step.py::_trained_magnitude_weights: Adam iterations at lr 2e-3 from
the reference's initialisation (oracle/torch_cpu.py) until |state| > 2 or |gate logit| > 8.

    python scripts/precision_budget.py [--batch 4] [--save /tmp/w.pt]
"""
import argparse
import itertools
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from oracle import model_math as mm          # noqa: E402
from oracle import torch_cpu as tc           # noqa: E402


def fp8(a, scale):
    """OCP e4m3 image of a * scale, back in a's units (what the MX-scaled MFMA contracts: v_mfma_scale_f32_16x16x128_f8f6f4)."""
    return (a.to(torch.float32) * scale).clamp(-448.0, 448.0).to(torch.float8_e4m3fn).to(torch.float64) / scale     # (saturating, as v_cvt_pk_fp8_f32 after the kernels' clamp)


def fp8_scale(a):
    """Power of two that puts max|a| just under e4m3's 448 (the per-tensor e8m0 exponent the kernel gets)."""
    m = float(a.abs().max())
    return 2.0 ** np.floor(np.log2(448.0 / m)) if m > 0 else 1.0


def e2m3(v):
    """OCP FP6 e2m3 (round to nearest even, saturating at 7.5) of v, elementwise (oracle/lowprec.py::e2m3_round in torch)."""
    mag = v.abs().clamp(max=7.5)
    q = torch.where(mag < 2.0, torch.round(mag * 8.0) / 8.0, torch.where(mag < 4.0, torch.round(mag * 4.0) / 4.0, torch.round(mag * 2.0) / 2.0))
    return torch.sign(v) * q


def fp6_rows(a, dim):
    """e2m3 image of a with one power-of-two scale per slice along `dim` (the slice maximum lands in [4, 8)), back in a's units:
    per weight row / per frame - the per-lane scale operands of v_mfma_scale_f32_16x16x128_f8f6f4."""
    m = a.abs().amax(dim=dim, keepdim=True)
    sc = 2.0 ** torch.where(m > 0, 2.0 - torch.floor(torch.log2(torch.where(m > 0, m, torch.ones_like(m)))), torch.zeros_like(m))
    return e2m3(a * sc) / sc


H6_SCALE = float(os.environ.get("EVC_BUDGET_H6_SCALE", "8"))     # fixed scale of the e2m3 image of h (|h| <= 1; 8: |h| > 0.94 saturates)


def e3m2(v):
    """OCP FP6 e3m2 ("bf6": 3 exponent bits, bias 3, 2 mantissa bits; subnormal step 2^-4, max 28), round to nearest even, saturating."""
    mag = v.abs().clamp(max=28.0)
    e = torch.floor(torch.log2(torch.where(mag > 0, mag, torch.ones_like(mag)))).clamp(min=-2.0, max=4.0)
    step = 2.0 ** (e - 2.0)
    return torch.sign(v) * torch.round(mag / step) * step


H6_FORMAT = os.environ.get("EVC_BUDGET_H6_FORMAT", "e3m2")       # format of the e-image of h: its values crowd near 0 with a tail to +-1, which
# wants exponent range, not mantissa (e2m3 under ONE fixed scale leaves 15 % relative error at |h| ~ 0.03: measured 1.5e-5 on the logits
# per corrected term instead of 1e-6)


def fp6_fixed(a, scale):
    if H6_FORMAT == "e3m2":
        return e3m2(a * 16.0) / 16.0
    return e2m3(a * scale) / scale


def rnd(a, kind):
    if kind == "x3":
        return a
    if kind in ("f16d", "f16s"):        # weights only: per-step dithered images (stack_fwd); this is the plain image for callers outside it
        kind = "f16"
    if kind in ("f16+8", "f16+6"):      # weights only: the f16 image; the fp8 / fp6 low-order half is contracted separately (stack_fwd)
        kind = "f16"
    dt = {"bf16": torch.bfloat16, "f16": torch.float16}[kind]
    return a.to(torch.float32).to(dt).to(torch.float64)


def dither_setup(k, seed, T):
    """Time-dithered f16 images of one LSTM kernel (oracle/lowprec.py::f16_dither_images = evc_cast_f32_to_f16_dither, bit for bit): the device
    dithers the master in its stored layout [4H][in + H], so the flat element index - the phase - is taken there.  k: TF layout [in + H][4H]."""
    from oracle import lowprec as lp
    kt = np.ascontiguousarray(k.to(torch.float32).numpy().T)
    im = lp.f16_dither_images(kt, T, seed)                       # [T][4H][C]
    return torch.from_numpy(np.ascontiguousarray(im.transpose(0, 2, 1)).astype(np.float64))       # [T][C][4H]


def stochastic_images(k, T):
    """independent stochastic rounding per step (comparison only)"""
    from oracle import lowprec as lp
    w = k.to(torch.float32).numpy()
    dn, up = lp.f16_neighbours(w)
    gap = up.astype(np.float64) - dn.astype(np.float64)
    frac = np.where(gap > 0, (w - dn.astype(np.float64)) / np.where(gap > 0, gap, 1.0), 0.0)
    out = np.stack([np.where(np.random.default_rng(1000 + t).random(w.shape) < frac, up, dn).astype(np.float64) for t in range(T)], 0)
    return torch.from_numpy(out)


def _parts(kind):
    """A layer's kind: one name for every operand, or a dict ax / ah / wx / wh (activation / weight, x-part / h-part)."""
    if isinstance(kind, dict):
        return kind["ax"], kind["ah"], kind["wx"], kind["wh"]
    return kind, kind, kind, kind


DITHER_ROW_SHIFT = None        # set by tower_fwd: int64 [M] - row r contracts image (t + shift[r]) mod T at step t (the per-chunk rotation candidate)


def stack_fwd(x, lengths, layers, kinds):
    """x [M, T, F] float64, layers [(kernel, bias)], kinds per layer -> final state [M, 2LH] (c0 h0 c1 h1)."""
    M, T, _ = x.shape
    H = layers[0][1].shape[0] // 4
    c = [x.new_zeros((M, H)) for _ in layers]
    h = [x.new_zeros((M, H)) for _ in layers]
    wq, wlo8, wdith = [], [], []
    for l, (k, _) in enumerate(layers):
        ax, ah, wx, wh = _parts(kinds[l])
        nin = k.shape[0] - H
        dith = None
        if "f16d" in (wx, wh):
            dith = dither_setup(k, 1 + 2 * l, T)          # (HLstmTower.dither_seed of the L1 kernels)
        elif "f16s" in (wx, wh):
            dith = stochastic_images(k, T)
        wdith.append((dith, wx in ("f16d", "f16s"), wh in ("f16d", "f16s")))
        wq.append(torch.cat([rnd(k[:nin], wx), rnd(k[nin:], wh)], 0))
        lo = torch.zeros_like(k)       # fp8 low-order halves (kind "f16+8"): W - f16(W) as e4m3 under one power-of-two scale per matrix
        if wx == "f16+8":
            d = k[:nin] - rnd(k[:nin], "f16")
            lo[:nin] = fp8(d, fp8_scale(d))
        if wh == "f16+8":
            d = k[nin:] - rnd(k[nin:], "f16")
            lo[nin:] = fp8(d, fp8_scale(d))
        if wx == "f16+6":               # e2m3, one scale per weight row (= per output column of the TF kernel) and segment
            lo[:nin] = fp6_rows(k[:nin] - rnd(k[:nin], "f16"), 0)
        if wh == "f16+6":
            lo[nin:] = fp6_rows(k[nin:] - rnd(k[nin:], "f16"), 0)
        wlo8.append(lo if (wx in ("f16+8", "f16+6") or wh in ("f16+8", "f16+6")) else None)
    for t in range(T):
        active = (lengths > t).unsqueeze(1)
        if not bool(active.any()):
            break
        inp = x[:, t]
        for l, (_, bias) in enumerate(layers):
            ax, ah, _, _ = _parts(kinds[l])
            a = torch.cat([rnd(inp, ax), rnd(h[l], ah)], 1)
            wl = wq[l]
            dith, dxp, dhp = wdith[l]
            if dith is not None and DITHER_ROW_SHIFT is not None and len(DITHER_ROW_SHIFT) == M:
                nin_l = layers[l][0].shape[0] - H
                z = torch.empty((M, wl.shape[1]), dtype=a.dtype)
                for sft in torch.unique(DITHER_ROW_SHIFT).tolist():
                    sel = DITHER_ROW_SHIFT == sft
                    img = dith[(t + int(sft)) % T]
                    wls = torch.cat([img[:nin_l] if dxp else wl[:nin_l], img[nin_l:] if dhp else wl[nin_l:]], 0)
                    z[sel] = a[sel] @ wls + bias
            else:
                if dith is not None:
                    nin_l = layers[l][0].shape[0] - H
                    wl = torch.cat([dith[t][:nin_l] if dxp else wl[:nin_l], dith[t][nin_l:] if dhp else wl[nin_l:]], 0)
                z = a @ wl + bias
            _, _, wxk, whk = _parts(kinds[l])
            if wlo8[l] is not None and "f16+6" in (wxk, whk):      # e2m3 images of the activations: the input frames under one scale per
                # frame (layer 0; layer 1's input is the h of the layer below), h under a fixed scale
                a6 = torch.cat([fp6_rows(inp, 1) if l == 0 else fp6_fixed(inp, H6_SCALE), fp6_fixed(h[l], H6_SCALE)], 1)
                z = z + a6 @ wlo8[l]
            elif wlo8[l] is not None:      # the low-order term on fp8 images of the activations (x 2^7: |h| <= 1, |x| <= 1)
                z = z + fp8(torch.cat([inp, h[l]], 1), 128.0) @ wlo8[l]
            if ax == "f16+6":            # the INPUT's low-order half in e2m3 against an e2m3 image of the weights, both under per-row scales
                d = inp - rnd(inp, "f16")
                nin = inp.shape[1]
                z = z + fp6_rows(d, 1) @ fp6_rows(layers[l][0][:nin], 0)
            if ax == "f16+8":            # the INPUT's low-order half in fp8 against an fp8 image of the weights: e4m3((x - f16(x)) 2^18) . e4m3(Wx 2^6)
                d = inp - rnd(inp, "f16")
                nin = inp.shape[1]
                z = z + fp8(d, 2.0 ** 18) @ fp8(layers[l][0][:nin], fp8_scale(layers[l][0][:nin]))
            if ah == "f16+8":            # round 6 candidate: the low-order half of h the same way, e4m3((h - f16(h)) 2^18) . e4m3(Wh 2^6)
                d = h[l] - rnd(h[l], "f16")
                nin = inp.shape[1]
                z = z + fp8(d, 2.0 ** 18) @ fp8(layers[l][0][nin:], fp8_scale(layers[l][0][nin:]))
            i, j, f, o = z.split(H, 1)
            cn = c[l] * torch.sigmoid(f + 1.0) + torch.sigmoid(i) * torch.tanh(j)
            hn = torch.tanh(cn) * torch.sigmoid(o)
            c[l] = torch.where(active, cn, c[l])
            h[l] = torch.where(active, hn, h[l])
            inp = hn
    return torch.cat([s for l in range(len(layers)) for s in (c[l], h[l])], 1)


def tower_fwd(x, n, p, num_chunks, kinds):
    """kinds: dict L1c0, L1c1, L2c0, L2c1, moe -> 'bf16' | 'f16' | 'x3'."""
    B, T, F = x.shape
    Lc = T // num_chunks
    l1, l2 = tc._layers(p, "RNN_L1", 2), tc._layers(p, "RNN_L2", 2)
    xc = x.reshape(B, num_chunks, Lc, F).permute(1, 0, 2, 3).reshape(num_chunks * B, Lc, F)       # row = chunk*B + b
    ln = torch.stack([torch.clamp(n - Lc * i, 0, Lc) for i in range(num_chunks)], 0).reshape(-1)
    global DITHER_ROW_SHIFT
    d = kinds.get("chunk_shift")          # candidate: chunk c of every video contracts image (t + c * d) mod Lc - the L2 level then integrates DIFFERENT residuals
    DITHER_ROW_SHIFT = (torch.arange(num_chunks).repeat_interleave(B) * int(d)) % Lc if d else None
    s1 = stack_fwd(xc, ln, l1, (kinds["L1c0"], kinds["L1c1"]))
    DITHER_ROW_SHIFT = None
    l2_in = s1.reshape(num_chunks, B, -1).permute(1, 0, 2)
    len2 = torch.ceil(n.to(torch.float32) / float(Lc)).to(torch.int64)
    state = stack_fwd(l2_in, len2, l2, (kinds["L2c0"], kinds["L2c1"]))
    mk = kinds["moe"] if isinstance(kinds["moe"], tuple) else (kinds["moe"], kinds["moe"])       # (activation kind, weight kind)
    if mk[0] == "f16+8":      # f16 product + both low-order corrections in fp8: e4m3(x 2^6) . e4m3(W_lo 2^18) + e4m3(x_lo 2^17) . e4m3(W 2^7)
        x16, xlo = rnd(state, "f16"), state - rnd(state, "f16")

        from oracle import lowprec as lp
        dr = lp.fp8_range_drop(float(state.abs().max()), 6)        # the head's dynamic e4m3 range (round 6: MoeHead.dynamic_fp8_range)

        def head(w):
            w16 = rnd(w, "f16")
            return x16 @ w16 + fp8(state, 2.0 ** (6 - dr)) @ fp8(w - w16, 2.0 ** 18) + fp8(xlo, 2.0 ** (17 - dr)) @ fp8(w, 128.0)
        gl = head(p["classifier/gates/weights"])
        el = head(p["classifier/experts/weights"]) + p["classifier/experts/biases"]
    else:
        sq = rnd(state, mk[0])
        gl = sq @ rnd(p["classifier/gates/weights"], mk[1])
        el = sq @ rnd(p["classifier/experts/weights"], mk[1]) + p["classifier/experts/biases"]
    V = el.shape[1] // 2
    g = torch.softmax(gl.reshape(B * V, 3), 1)
    pred = (g[:, :2] * torch.sigmoid(el.reshape(B * V, 2))).sum(1).reshape(B, V)
    return dict(state=state, gate_logits=gl, expert_logits=el, pred=pred)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--seed", type=int, default=91)
    ap.add_argument("--init_seed", type=int, default=3)
    ap.add_argument("--lr", type=float, default=2e-3)
    ap.add_argument("--max_steps", type=int, default=16)
    ap.add_argument("--save", default="")
    ap.add_argument("--load", default="")
    ap.add_argument("--only", default="", help="run only the configurations whose label contains this string")
    ap.add_argument("--load_sd", default="", help="a checkpoint of tests/_long_train.py ({'sd': TF-named state dict of both towers}); the inputs are then "
                    "its 4 evaluation videos (round 6: the budget at the 512-step horizon)")
    ap.add_argument("--gpu", action="store_true", help="weights from the GPU training of tests/test_gpu_step.py::_trained_magnitude_weights "
                    "(run on the GPU box); also prints the errors of the real kernels in both precision modes")
    a = ap.parse_args()
    torch.set_num_threads(8)
    B = a.batch
    q, x, n, labels = mm.synthetic_batch(B, seed=a.seed, dtype=np.float32)
    n[0] = 300
    x[np.arange(300)[None, :] >= n[:, None]] = 0.0
    if a.load_sd:
        sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
        import _long_train as lt
        x, n, labels = lt.eval_videos(16)
        B = 4
        sd = torch.load(a.load_sd, weights_only=False)["sd"]
        teacher = {k[len("model/"):]: v.float().cpu() for k, v in sd.items() if k.startswith("model/")}
        student = {k[len("model_student/"):]: v.float().cpu() for k, v in sd.items() if k.startswith("model_student/")}
    elif a.gpu:
        sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
        import test_gpu_step as tgs
        sd = tgs._trained_magnitude_weights(B, x, n, labels, deterministic=os.environ.get("EVC_BUDGET_DETERMINISTIC") == "1")   # default: a fresh draw per run (margin study)
        teacher = {k[len("model/"):]: v.float().cpu() for k, v in sd.items() if k.startswith("model/")}
        student = {k[len("model_student/"):]: v.float().cpu() for k, v in sd.items() if k.startswith("model_student/")}
    elif a.load:
        teacher, student = torch.load(a.load)
    else:
        rng = np.random.default_rng(a.init_seed)
        teacher = tc.to_torch(mm.init_hlstm_params(rng, dtype=np.float32))
        student = tc.to_torch(mm.init_hlstm_params(rng, dtype=np.float32))
        ot, os_ = tc.Adam(teacher, lr=a.lr), tc.Adam(student, lr=a.lr)
        xt, yt = torch.from_numpy(x), torch.from_numpy(labels.astype(np.float32))
        for it in range(a.max_steps):
            t0 = time.time()
            tc.teacher_student_iteration(xt, n, yt, teacher, student, 10, ot, os_)
            with torch.no_grad():
                st, _ = tc.hlstm_fwd(tc.l2_normalize(xt), n, teacher, 20)
                zmax = float((st @ teacher["classifier/gates/weights"]).abs().max())
                smax = float(st.abs().max())
            print("iteration %d: |state| %.2f |gate logit| %.2f (%.0f s)" % (it + 1, smax, zmax, time.time() - t0), flush=True)
            if smax > 2.0 or zmax > 8.0:
                break
        if a.save:
            torch.save((teacher, student), a.save)
    with torch.no_grad():
        xd = tc.l2_normalize(torch.from_numpy(x)).double()
        nt = torch.as_tensor(n, dtype=torch.int64)
        S = 30
        n_s = torch.as_tensor(np.trunc(n.astype(np.float64) / 300.0 * S).astype(np.int64))
        xs = xd[:, ::10][:, :S]
        towers = (("teacher", {k: v.detach().double() for k, v in teacher.items()}, xd, nt, 20),
                  ("student", {k: v.detach().double() for k, v in student.items()}, xs, n_s, 5))
        exact = dict(L1c0="x3", L1c1="x3", L2c0="x3", L2c1="x3", moe="x3")
        ref = {name: tower_fwd(xx, nn, p, ch, exact) for name, p, xx, nn, ch in towers}
        for name in ref:
            print("%s: |state| %.2f |gate logit| %.2f |expert logit| %.2f" % (
                name, ref[name]["state"].abs().max(), ref[name]["gate_logits"].abs().max(), ref[name]["expert_logits"].abs().max()))
        configs = [
            ("all bf16", dict(L1c0="bf16", L1c1="bf16", L2c0="bf16", L2c1="bf16", moe="bf16")),
            ("all f16", dict(L1c0="f16", L1c1="f16", L2c0="f16", L2c1="f16", moe="f16")),
            ("moe only bf16", dict(exact, moe="bf16")),
            ("moe only f16", dict(exact, moe="f16")),
            ("L1 bf16, rest x3", dict(exact, L1c0="bf16", L1c1="bf16")),
            ("L1 f16, rest x3", dict(exact, L1c0="f16", L1c1="f16")),
            ("L1c0 bf16, rest x3", dict(exact, L1c0="bf16")),
            ("L1c1 bf16, rest x3", dict(exact, L1c1="bf16")),
            ("L1c0 f16, rest x3", dict(exact, L1c0="f16")),
            ("L1c1 f16, rest x3", dict(exact, L1c1="f16")),
            ("L2 bf16, rest x3", dict(exact, L2c0="bf16", L2c1="bf16")),
            ("L2 f16, rest x3", dict(exact, L2c0="f16", L2c1="f16")),
            ("L1+L2 f16, moe x3", dict(L1c0="f16", L1c1="f16", L2c0="f16", L2c1="f16", moe="x3")),
        ]
        def c0(**kw):
            d = dict(ax="x3", ah="x3", wx="x3", wh="x3")
            d.update(kw)
            return dict(exact, L1c0=d)
        configs += [
            ("L1c0 f16: x-part only", c0(ax="f16", wx="f16")),
            ("L1c0 f16: h-part only", c0(ah="f16", wh="f16")),
            ("L1c0 f16: weights only", c0(wx="f16", wh="f16")),
            ("L1c0 f16: activ. only", c0(ax="f16", ah="f16")),
            ("L1c0 f16: Wx only", c0(wx="f16")),
            ("L1c0 f16: x only", c0(ax="f16")),
            ("L1c0 f16: Wh only", c0(wh="f16")),
            ("L1c0 f16: h only", c0(ah="f16")),
        ]
        f16h = dict(ax="x3", ah="f16", wx="x3", wh="f16")
        configs += [
            ("PLAN c0 x-part x3 + h f16, c1 f16", dict(exact, L1c0=f16h, L1c1="f16")),
            ("PLAN c0 x split only (Wx f16)", dict(exact, L1c0=dict(f16h, wx="f16"), L1c1="f16")),
        ]
        ex = dict(ax="x3", ah="x3", wx="x3", wh="x3")
        plan = dict(exact, L1c0=f16h, L1c1="f16")
        for kind in ("f16", "bf16"):
            configs += [
                ("L2 parts %s: c0 x-part" % kind, dict(exact, L2c0=dict(ex, ax=kind, wx=kind))),
                ("L2 parts %s: c0 h-part" % kind, dict(exact, L2c0=dict(ex, ah=kind, wh=kind))),
                ("L2 parts %s: c1 x-part" % kind, dict(exact, L2c1=dict(ex, ax=kind, wx=kind))),
                ("L2 parts %s: c1 h-part" % kind, dict(exact, L2c1=dict(ex, ah=kind, wh=kind))),
                ("L2 parts %s: all h-parts" % kind, dict(exact, L2c0=dict(ex, ah=kind, wh=kind), L2c1=dict(ex, ah=kind, wh=kind))),
                ("L2 parts %s: c1 whole" % kind, dict(exact, L2c1=kind)),
            ]
        for lay in ("L2c0", "L2c1"):
            for part in ("ax", "ah", "wx", "wh"):
                configs.append(("L2 fine f16: %s %s only" % (lay, part), dict(exact, **{lay: dict(ex, **{part: "f16"})})))
        # candidate: the whole L2 level on f16 with the ACTIVATION operands exact (K-extended by their low halves), weights f16
        act_ext = dict(ax="x3", ah="x3", wx="f16", wh="f16")
        configs += [
            ("L2 fine PLAN + L2 f16 weights, exact activations", dict(plan, L2c0=act_ext, L2c1=act_ext)),
            ("L2 fine PLAN + L2 f16, only c1 h exact", dict(plan, L2c0="f16", L2c1=dict(ax="f16", ah="x3", wx="f16", wh="f16"))),
            ("L2 fine PLAN + L2 f16, c1 h and c0 h exact", dict(plan, L2c0=dict(ax="f16", ah="x3", wx="f16", wh="f16"),
                                                                 L2c1=dict(ax="f16", ah="x3", wx="f16", wh="f16"))),
            ("L2 fine PLAN + L2 f16, all activations exact but c0 x", dict(plan, L2c0=dict(ax="f16", ah="x3", wx="f16", wh="f16"), L2c1=act_ext)),
        ]
        configs += [("MOE fine: %s" % k, v) for k, v in (
            ("f16 + fp8 low-order corrections of x and W", dict(exact, moe=("f16+8", "f16+8"))), ("f16 both", dict(exact, moe="f16")),
            ("x f16, W exact", dict(exact, moe=("f16", "x3"))), ("x exact, W f16", dict(exact, moe=("x3", "f16"))),
            ("x bf16, W exact", dict(exact, moe=("bf16", "x3"))), ("x exact, W bf16", dict(exact, moe=("x3", "bf16"))))]
        w_ext = dict(ax="f16", ah="f16", wx="x3", wh="x3")
        c0_2 = dict(ax="x3", ah="f16", wx="f16", wh="f16")          # L1 layer 0: input frames exact (2 segments)
        c0_3 = dict(ax="x3", ah="f16", wx="x3", wh="f16")           # ... + Wx exact (3 segments)
        c0_3h = dict(ax="x3", ah="f16", wx="x3", wh="x3")           # ... + Wh exact
        l2_f16 = dict(L2c0="f16", L2c1=w_ext)
        l2_f16x = dict(L2c0=dict(ax="x3", ah="f16", wx="x3", wh="x3"), L2c1=w_ext)      # + layer 0: input and weights exact (shipped)
        configs += [("ROBUST %s" % k, v) for k, v in (
            ("A  L1c0 x-ext | L1c1 f16 | L2 f16+c1 W-ext", dict(exact, L1c0=c0_2, L1c1="f16", **l2_f16)),
            ("B  L1c0 x,Wx-ext | L1c1 f16 | L2 f16+c1 W-ext", dict(exact, L1c0=c0_3, L1c1="f16", **l2_f16)),
            ("C  L1c0 x,Wx-ext | L1c1 f16 | L2 exact", dict(exact, L1c0=c0_3, L1c1="f16")),
            ("D  L1c0 x-ext | L1c1 f16 | L2 exact", dict(exact, L1c0=c0_2, L1c1="f16")),
            ("E  L1c0 x,Wx-ext | L1c1 W-ext | L2 f16+c1 W-ext", dict(exact, L1c0=c0_3, L1c1=w_ext, **l2_f16)),
            ("F  L1c0 x,Wx,Wh-ext | L1c1 f16 | L2 f16+c1 W-ext", dict(exact, L1c0=c0_3h, L1c1="f16", **l2_f16)),
            ("G  L1 weights exact (acts f16, x exact) | L2 f16+c1 W-ext", dict(exact, L1c0=c0_3h, L1c1=w_ext, **l2_f16)),
            ("F2 L1c0 x,Wh-ext (Wx f16) | L1c1 f16 | L2 f16+c1 W-ext", dict(exact, L1c0=dict(ax="x3", ah="f16", wx="f16", wh="x3"), L1c1="f16", **l2_f16)),
            ("F3 L1c0 x,Wx,Wh-ext | L1c1 Wh-ext | L2 f16+c1 W-ext", dict(exact, L1c0=c0_3h, L1c1=dict(ax="f16", ah="f16", wx="f16", wh="x3"), **l2_f16)),
            ("F4 L1c0 x,Wh-ext (Wx f16) | L1c1 Wh-ext | L2 f16+c1 W-ext", dict(exact, L1c0=dict(ax="x3", ah="f16", wx="f16", wh="x3"),
                                                                            L1c1=dict(ax="f16", ah="f16", wx="f16", wh="x3"), **l2_f16)),
            ("F5 L1c0 x,Wh-ext | L1c1 W-ext (Wx,Wh) | L2 f16+c1 W-ext", dict(exact, L1c0=dict(ax="x3", ah="f16", wx="f16", wh="x3"), L1c1=w_ext, **l2_f16)),
            ("I1 only L1c1 x(=h0) f16", dict(exact, L1c1=dict(ax="f16", ah="x3", wx="x3", wh="x3"))),
            ("I2 only L1c1 h f16", dict(exact, L1c1=dict(ax="x3", ah="f16", wx="x3", wh="x3"))),
            ("I3 only L1c1 Wx f16", dict(exact, L1c1=dict(ax="x3", ah="x3", wx="f16", wh="x3"))),
            ("I4 only L1c1 Wh f16", dict(exact, L1c1=dict(ax="x3", ah="x3", wx="x3", wh="f16"))),
            ("FZ FX + L1c1 Wx,Wh exact", dict(exact, L1c0=c0_3h, L1c1=w_ext, **l2_f16x)),
            ("FZ8 FZ with the L1 weights' low-order halves in fp8", dict(exact, L1c0=dict(ax="x3", ah="f16", wx="f16+8", wh="f16+8"),
                                                                         L1c1=dict(ax="f16", ah="f16", wx="f16+8", wh="f16+8"), **l2_f16x)),
            ("FZ9 FZ8 + L2 weights' low-order halves in fp8", dict(exact, L1c0=dict(ax="x3", ah="f16", wx="f16+8", wh="f16+8"),
                                                                         L1c1=dict(ax="f16", ah="f16", wx="f16+8", wh="f16+8"),
                                                                         L2c0=dict(ax="x3", ah="f16", wx="f16+8", wh="f16+8"),
                                                                         L2c1=dict(ax="f16", ah="f16", wx="f16+8", wh="f16+8"))),
            ("X16 only the L1c0 input f16", dict(exact, L1c0=dict(ex, ax="f16"))),
            ("X16+8 only the L1c0 input f16 + fp8 x fp8 low-order term", dict(exact, L1c0=dict(ex, ax="f16+8"))),
            ("FZ8X FZ8 with the input's low-order half in fp8 too", dict(exact, L1c0=dict(ax="f16+8", ah="f16", wx="f16+8", wh="f16+8"),
                                                                         L1c1=dict(ax="f16", ah="f16", wx="f16+8", wh="f16+8"), **l2_f16x)),
            ("DITH W16d only L1 weights f16, dithered in time", dict(exact, L1c0=dict(ex, wx="f16d", wh="f16d"), L1c1=dict(ex, wx="f16d", wh="f16d"))),
            ("DITH W16s only L1 weights f16, stochastic per step", dict(exact, L1c0=dict(ex, wx="f16s", wh="f16s"), L1c1=dict(ex, wx="f16s", wh="f16s"))),
            ("DITH W16  only L1 weights f16 RTN", dict(exact, L1c0=dict(ex, wx="f16", wh="f16"), L1c1=dict(ex, wx="f16", wh="f16"))),
            ("DITH W2d only the L2 weights f16 dithered", dict(exact, L2c0=dict(ex, wx="f16d", wh="f16d"), L2c1=dict(ex, wx="f16d", wh="f16d"))),
            ("DITH W2  only the L2 weights f16 RTN", dict(exact, L2c0=dict(ex, wx="f16", wh="f16"), L2c1=dict(ex, wx="f16", wh="f16"))),
            ("DITH FZD8X L1 W dithered, x_lo fp8 | shipped L2", dict(exact, L1c0=dict(ax="f16+8", ah="f16", wx="f16d", wh="f16d"),
                                                                       L1c1=dict(ax="f16", ah="f16", wx="f16d", wh="f16d"), **l2_f16x)),
            ("DITH FZDD L1 + L2 W dithered, x_lo fp8", dict(exact, L1c0=dict(ax="f16+8", ah="f16", wx="f16d", wh="f16d"),
                                                                       L1c1=dict(ax="f16", ah="f16", wx="f16d", wh="f16d"),
                                                                       L2c0=dict(ax="x3", ah="f16", wx="f16d", wh="f16d"), L2c1=dict(ax="f16", ah="f16", wx="f16d", wh="f16d"))),
            ("DITH FZS8X L1 W stochastic, x_lo fp8 | shipped L2", dict(exact, L1c0=dict(ax="f16+8", ah="f16", wx="f16s", wh="f16s"),
                                                                       L1c1=dict(ax="f16", ah="f16", wx="f16s", wh="f16s"), **l2_f16x)),
            ("DITH FZD1 layer 1 dithered, layer 0 fp8-corrected | shipped L2", dict(exact, L1c0=dict(ax="f16+8", ah="f16", wx="f16+8", wh="f16+8"),
                                                                       L1c1=dict(ax="f16", ah="f16", wx="f16d", wh="f16d"), **l2_f16x)),
            ("DITH FZD0 layer 0 dithered (x_lo fp8), layer 1 fp8-corrected | shipped L2", dict(exact, L1c0=dict(ax="f16+8", ah="f16", wx="f16d", wh="f16d"),
                                                                       L1c1=dict(ax="f16", ah="f16", wx="f16+8", wh="f16+8"), **l2_f16x)),
            ("DITH FZDh Wh parts dithered, Wx parts fp8-corrected | shipped L2", dict(exact, L1c0=dict(ax="f16+8", ah="f16", wx="f16+8", wh="f16d"),
                                                                       L1c1=dict(ax="f16", ah="f16", wx="f16+8", wh="f16d"), **l2_f16x)),
            ("DITH FZDx Wx parts dithered, Wh parts fp8-corrected | shipped L2", dict(exact, L1c0=dict(ax="f16+8", ah="f16", wx="f16d", wh="f16+8"),
                                                                       L1c1=dict(ax="f16", ah="f16", wx="f16d", wh="f16+8"), **l2_f16x)),
            ("DITH FZD8XC1 both layers dithered, chunk c rotated by c steps", dict(exact, L1c0=dict(ax="f16+8", ah="f16", wx="f16d", wh="f16d"),
                                                                       L1c1=dict(ax="f16", ah="f16", wx="f16d", wh="f16d"), chunk_shift=1, **l2_f16x)),
            ("DITH FZD8XC4 both layers dithered, chunk c rotated by 4c steps", dict(exact, L1c0=dict(ax="f16+8", ah="f16", wx="f16d", wh="f16d"),
                                                                       L1c1=dict(ax="f16", ah="f16", wx="f16d", wh="f16d"), chunk_shift=4, **l2_f16x)),
            ("DITH FZD8XC7 both layers dithered, chunk c rotated by 7c steps", dict(exact, L1c0=dict(ax="f16+8", ah="f16", wx="f16d", wh="f16d"),
                                                                       L1c1=dict(ax="f16", ah="f16", wx="f16d", wh="f16d"), chunk_shift=7, **l2_f16x)),
            ("DITH W16dC4 only L1 weights dithered, chunk c rotated by 4c steps", dict(exact, L1c0=dict(ex, wx="f16d", wh="f16d"), L1c1=dict(ex, wx="f16d", wh="f16d"), chunk_shift=4)),
            ("DITH FZD1L2 L1 layer 1 + L2 layer 1 dithered, the rest fp8-corrected", dict(exact, L1c0=dict(ax="f16+8", ah="f16", wx="f16+8", wh="f16+8"),
                                                                       L1c1=dict(ax="f16", ah="f16", wx="f16d", wh="f16d"),
                                                                       L2c0=dict(ax="x3", ah="f16", wx="x3", wh="x3"), L2c1=dict(ax="f16", ah="f16", wx="f16d", wh="f16d"))),
            ("DITH FZD3 layer 1 + layer 0's Wh dithered, Wx0 + x_lo fp8-corrected | shipped L2", dict(exact, L1c0=dict(ax="f16+8", ah="f16", wx="f16+8", wh="f16d"),
                                                                       L1c1=dict(ax="f16", ah="f16", wx="f16d", wh="f16d"), **l2_f16x)),
            ("DITH A8 L1 W RTN uncorrected, x_lo fp8 | shipped L2", dict(exact, L1c0=dict(ax="f16+8", ah="f16", wx="f16", wh="f16"),
                                                                       L1c1="f16", **l2_f16x)),
            ("DBG6 c0 wx", dict(exact, L1c0=dict(ex, wx="f16+6"))), ("DBG6 c0 wh", dict(exact, L1c0=dict(ex, wh="f16+6"))), ("DBG6 c1 wx", dict(exact, L1c1=dict(ex, wx="f16+6"))), ("DBG6 c1 wh", dict(exact, L1c1=dict(ex, wh="f16+6"))),
            ("DBG8 c0 wx", dict(exact, L1c0=dict(ex, wx="f16+8"))), ("DBG8 c1 wh", dict(exact, L1c1=dict(ex, wh="f16+8"))),
            ("W16+6 only the L1 weights f16 + e2m3 low-order halves (per-row scales)", dict(exact, L1c0=dict(ex, wx="f16+6", wh="f16+6"), L1c1=dict(ex, wx="f16+6", wh="f16+6"))),
            ("X16+6 only the L1c0 input f16 + e2m3 x e2m3 low-order term", dict(exact, L1c0=dict(ex, ax="f16+6"))),
            ("FZ6X FZ8X with the L1 level's corrections in e2m3", dict(exact, L1c0=dict(ax="f16+6", ah="f16", wx="f16+6", wh="f16+6"),
                                                                       L1c1=dict(ax="f16", ah="f16", wx="f16+6", wh="f16+6"), **l2_f16x)),
            ("FZ6A FZ6X + L2 weights' low-order halves in e2m3", dict(exact, L1c0=dict(ax="f16+6", ah="f16", wx="f16+6", wh="f16+6"),
                                                                      L1c1=dict(ax="f16", ah="f16", wx="f16+6", wh="f16+6"),
                                                                      L2c0=dict(ax="x3", ah="f16", wx="x3", wh="f16+6"),
                                                                      L2c1=dict(ax="f16", ah="f16", wx="f16+6", wh="f16+6"))),
            ("W16 only the L1 weights f16 (activations exact)", dict(exact, L1c0=dict(ex, wx="f16", wh="f16"), L1c1=dict(ex, wx="f16", wh="f16"))),
            ("W16+8 only the L1 weights f16 + fp8 low-order halves", dict(exact, L1c0=dict(ex, wx="f16+8", wh="f16+8"), L1c1=dict(ex, wx="f16+8", wh="f16+8"))),
            ("FX (shipped) F with L2 layer 0 input + weights extended", dict(exact, L1c0=c0_3h, L1c1="f16", **l2_f16x)),
            ("FY only the shipped L2 level (L1 exact)", dict(exact, **l2_f16x)),
            ("FG L1 weights exact, acts f16, x exact | L2 f16+c1 W-ext", dict(exact, L1c0=c0_3h, L1c1=w_ext, **l2_f16)),
            ("FS all LSTM weights exact, acts f16, x exact", dict(exact, L1c0=c0_3h, L1c1=w_ext, L2c0=w_ext, L2c1=w_ext)),
            ("FT L1 exact | L2 f16+c1 W-ext", dict(exact, **l2_f16)),
            ("H  only L2 f16+c1 W-ext (L1 exact)", dict(exact, **l2_f16)),
            ("I  only L1c1 f16", dict(exact, L1c1="f16")),
            ("J  only L1c0 Wh f16", dict(exact, L1c0=dict(ax="x3", ah="x3", wx="x3", wh="f16"))),
            ("K  only L1c0 h f16", dict(exact, L1c0=dict(ax="x3", ah="f16", wx="x3", wh="x3"))),
            ("L  only L1c0 Wx f16", dict(exact, L1c0=dict(ax="x3", ah="x3", wx="f16", wh="x3"))),
        )]
        # ---- round 6: layouts for the 512-step horizon (budget of the worst deterministic draw: the input's e4m3 x e4m3 correction leaves
        # 1.8e-3, the uncorrected f16 rounding of h in L1 layer 0 1.0e-3, of L2 layer 1's two activation operands 1.2e-3 / 5e-4) ----
        w8 = lambda ax, ah: dict(ax=ax, ah=ah, wx="f16+8", wh="f16+8")
        l1c1_d = dict(ax="f16", ah="f16", wx="f16d", wh="f16d")
        l2c0_s = dict(ax="x3", ah="f16", wx="x3", wh="f16+8")           # shipped: K-extended input product, recurrent weights' lo in fp8
        head8 = ("f16+8", "f16+8")
        configs += [("R6 %s" % k, v) for k, v in (
            ("shipped (x f16+8, h f16, L1c1 dithered, L2c1 W fp8)", dict(L1c0=w8("f16+8", "f16"), L1c1=l1c1_d, L2c0=l2c0_s, L2c1=w8("f16", "f16"), moe=head8)),
            ("A x EXACT (integer frames), rest shipped", dict(L1c0=w8("x3", "f16"), L1c1=l1c1_d, L2c0=l2c0_s, L2c1=w8("f16", "f16"), moe=head8)),
            ("B A + L1c0 h_lo in fp8", dict(L1c0=w8("x3", "f16+8"), L1c1=l1c1_d, L2c0=l2c0_s, L2c1=w8("f16", "f16"), moe=head8)),
            ("C B + L2c1 both activations' lo in fp8", dict(L1c0=w8("x3", "f16+8"), L1c1=l1c1_d, L2c0=l2c0_s, L2c1=w8("f16+8", "f16+8"), moe=head8)),
            ("D C + L2c0 h_lo in fp8", dict(L1c0=w8("x3", "f16+8"), L1c1=l1c1_d, L2c0=dict(l2c0_s, ah="f16+8"), L2c1=w8("f16+8", "f16+8"), moe=head8)),
            ("E D + L1c1 fp8-corrected weights and both activations' lo", dict(L1c0=w8("x3", "f16+8"), L1c1=w8("f16+8", "f16+8"), L2c0=dict(l2c0_s, ah="f16+8"),
                                                                            L2c1=w8("f16+8", "f16+8"), moe=head8)),
            ("F C with the f32-input fallback (x f16+8)", dict(L1c0=w8("f16+8", "f16+8"), L1c1=l1c1_d, L2c0=l2c0_s, L2c1=w8("f16+8", "f16+8"), moe=head8)),
            ("G C without the L1c0 h_lo (x exact + L2c1 activations)", dict(L1c0=w8("x3", "f16"), L1c1=l1c1_d, L2c0=l2c0_s, L2c1=w8("f16+8", "f16+8"), moe=head8)),
        )]
        configs += [
            ("L2 fine PLAN2: c0 f16, c1 f16 acts + exact weights", dict(plan, L2c0="f16", L2c1=w_ext)),
            ("L2 fine PLAN2b: c0 and c1 f16 acts + exact weights", dict(plan, L2c0=w_ext, L2c1=w_ext)),
            ("L2 fine PLAN2c: c0 f16 but Wh exact, c1 exact weights", dict(plan, L2c0=dict(ax="f16", ah="f16", wx="f16", wh="x3"), L2c1=w_ext)),
        ]
        configs += [
            ("L2 parts PLAN+L2 h-parts f16", dict(plan, L2c0=dict(ex, ah="f16", wh="f16"), L2c1=dict(ex, ah="f16", wh="f16"))),
            ("L2 parts PLAN+L2 c1 f16, c0 h f16", dict(plan, L2c0=dict(ex, ah="f16", wh="f16"), L2c1="f16")),
            ("L2 parts PLAN+L2 all f16", dict(plan, L2c0="f16", L2c1="f16")),
        ]
        if a.only:
            configs = [c for c in configs if any(o in c[0] for o in a.only.split("|"))]
        if a.gpu:
            from efficientvideoclassification_youtube8m_amd.distill import DistillGraph
            dev = "cuda:0"
            xg, yg, ng = torch.from_numpy(x).to(dev), torch.from_numpy(labels.astype(np.uint8)).to(dev), torch.from_numpy(n).to(dev)
            for prec in ("bf16", "high"):
                g = DistillGraph(B, every_n=10, device=dev, seed=3, precision=prec)
                g.teacher.load_state_dict(sd)
                g.student.load_state_dict(sd)
                out = g.step(xg, yg, ng, apply=False, num_frames_host=n)
                row = []
                for name, tw, ks, kp in (("teacher", g.teacher, "teacher_state", "predictions"), ("student", g.student, "student_state", "student_predictions")):
                    row.append("%s state %.1e gate %.1e expert %.1e pred %.1e" % (
                        name[0], (out[ks].double().cpu() - ref[name]["state"]).abs().max(),
                        (tw.moe.gate_logits.double().cpu() - ref[name]["gate_logits"]).abs().max(),
                        (tw.moe.expert_logits.double().cpu() - ref[name]["expert_logits"]).abs().max(),
                        (out[kp].double().cpu() - ref[name]["pred"]).abs().max()))
                print("%-22s %s" % ("KERNELS " + prec, " | ".join(row)), flush=True)
                del g
                torch.cuda.empty_cache()
        for label, kinds in configs:
            row = []
            for name, p, xx, nn, ch in towers:
                out = tower_fwd(xx, nn, p, ch, kinds)
                row.append("%s state %.1e gate %.1e expert %.1e pred %.1e" % (
                    name[0], (out["state"] - ref[name]["state"]).abs().max(), (out["gate_logits"] - ref[name]["gate_logits"]).abs().max(),
                    (out["expert_logits"] - ref[name]["expert_logits"]).abs().max(), (out["pred"] - ref[name]["pred"]).abs().max()))
            print("%-22s %s" % (label, " | ".join(row)), flush=True)


if __name__ == "__main__":
    main()
