"""End-to-end parity of the teacher+student training iteration (through the
C ABI) against the float64 numpy oracle.  pytest -m gpu."""
import os

import numpy as np
import pytest
import torch

from oracle import model_math as mm

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _rel(a, b):
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))


L2_TOL = 1.2e-2     # relative L2 of a gradient tensor: bf16 operands in the backward GEMMs, measured 2e-3 .. 8e-3 (printed by the tests)


def _rel2(a, b):
    """Relative L2 error of a whole tensor (the max-based _rel is dominated by the bf16 rounding of single large entries)."""
    return float(np.linalg.norm(np.asarray(a, np.float64) - b) / (np.linalg.norm(b) + 1e-30))


@pytest.mark.parametrize("batch", [6, 8])     # 6: ragged row counts (NT + transposes); 8: T*M % 32 == 0 (TN weight-gradient GEMMs)
def test_small_step_forward_backward_update(batch):
    from efficientvideoclassification_youtube8m_amd import smoke
    g, out, ref, err = smoke.run(batch=batch, feature_size=64, lstm_cells=64, vocab_size=48, every_n=10, seed=1)
    assert g.teacher.l1.use_tn == (batch == 8)
    # gradients of both towers (bf16 GEMM operands, f32 accumulation): relative to each tensor's max
    for tower, key in ((g.teacher, "teacher_grads"), (g.student, "student_grads")):
        got = smoke.tower_grads_numpy(tower)
        for k in mm.HLSTM_PARAM_ORDER:
            gref = ref[key][k]
            if k in ("classifier/gates/weights", "classifier/experts/weights"):
                gref = gref - 2.0 * 1e-8 * smoke.tower_params_numpy(tower)[k]   # l2 term is folded in at apply time
            r = _rel(got[k], gref)
            assert r < 3e-2, (tower.scope, k, r)
            assert _rel2(got[k], gref) < L2_TOL, (tower.scope, k, _rel2(got[k], gref))
    # apply: per-tensor clip + TF-Adam, global_step += 2
    p_before = {t.scope: smoke.tower_params_numpy(t) for t in (g.teacher, g.student)}
    grads = {t.scope: smoke.tower_grads_numpy(t) for t in (g.teacher, g.student)}
    g.apply_gradients(batch)
    assert g.global_step == 2
    for t in (g.teacher, g.student):
        gr = dict(grads[t.scope])
        for k in ("classifier/gates/weights", "classifier/experts/weights"):
            gr[k] = gr[k] + 2.0 * 1e-8 * p_before[t.scope][k]
        want = mm.apply_train_op(p_before[t.scope], gr, {}, 1, 1e-3, 1.0)
        got = smoke.tower_params_numpy(t)
        for k in mm.HLSTM_PARAM_ORDER:
            # first Adam step moves every weight by ~lr*sign(g): compare the step itself
            step_ref = want[k] - p_before[t.scope][k]
            step_got = got[k] - p_before[t.scope][k]
            big = np.abs(grads[t.scope][k]) > 1e-6 * np.abs(grads[t.scope][k]).max() + 1e-12
            assert np.abs(step_got - step_ref)[big].max() < 2e-5, (t.scope, k)
        # bf16 shadows follow the masters
        for k, sh in t.shadow_fwd.items():
            assert torch.equal(sh, t.store.p(k).bfloat16())
            sb = t.shadow_bwd[k]
            ref = t.store.p(k).t().bfloat16()
            if k.endswith("basic_lstm_cell/kernel"):      # LSTM backward shadow: 4H axis gate-interleaved
                Hh = sh.shape[0] // 4
                ref = ref.reshape(-1, 4, Hh).permute(0, 2, 1).reshape(-1, 4 * Hh)
            assert torch.equal(sb[:, :sh.shape[0]], ref)


def test_real_dims_forward_within_1e3():
    """north_star tolerance: outputs within 1e-3 of the CPU path on identical
    inputs at the real model size (F=1152, H=1024x2, V=4716, 300 frames)."""
    from efficientvideoclassification_youtube8m_amd import smoke, ops
    from efficientvideoclassification_youtube8m_amd.distill import DistillGraph
    B = 3
    q, x, n, labels = mm.synthetic_batch(B, seed=77, dtype=np.float32)
    n[0] = 300
    g = DistillGraph(B, every_n=10, device=DEV, seed=3)
    out = g.step(torch.from_numpy(x).to(DEV), torch.from_numpy(labels.astype(np.uint8)).to(DEV),
                 torch.from_numpy(n).to(DEV), apply=False)
    teacher, student = smoke.tower_params_numpy(g.teacher), smoke.tower_params_numpy(g.student)
    ref = mm.teacher_student_step(x.astype(np.float64), n, labels, teacher, student, 10, with_grads=False)
    e_tp = np.abs(out["predictions"].cpu().numpy() - ref["teacher_predictions"]).max()
    e_sp = np.abs(out["student_predictions"].cpu().numpy() - ref["student_predictions"]).max()
    e_ts = np.abs(out["teacher_state"].cpu().numpy() - ref["teacher_state"]).max()
    e_ss = np.abs(out["student_state"].cpu().numpy() - ref["student_state"]).max()
    gl_ref = ref["teacher_state"] @ teacher["classifier/gates/weights"]
    e_gl = np.abs(g.teacher.gate_logits.cpu().numpy() - gl_ref).max()
    rep = g.loss_report()
    print("real-dims errors: teacher pred %.2e state %.2e gate-logits %.2e | student pred %.2e state %.2e"
          % (e_tp, e_ts, e_gl, e_sp, e_ss))
    print("losses got", rep, "ref", {k: float(ref[k]) for k in ("label_loss", "student_loss_state", "pred_loss",
                                                                 "student_label_loss")})
    assert e_tp < 1e-3 and e_sp < 1e-3
    assert e_gl < 1e-3
    assert abs(rep["label_loss"] - ref["label_loss"]) / ref["label_loss"] < 1e-4
    assert abs(rep["label_loss"] - 1914.1) / 1914.1 < 0.005            # README.md:116 known answer
    assert np.array_equal(out["num_frames_student"].cpu().numpy(), ref["num_frames_student"])


def test_three_iterations_track_the_oracle():
    """Several consecutive teacher+student iterations (forward, BPTT, per-tensor
    clip, TF-Adam with the folded l2 term, shadow refresh): the loss trajectory
    must follow the float64 oracle's."""
    from efficientvideoclassification_youtube8m_amd import smoke
    from efficientvideoclassification_youtube8m_amd.distill import DistillGraph
    B, F, H, V, every_n = 5, 64, 64, 40, 30
    q, x, n, labels = mm.synthetic_batch(B, seed=21, feature_size=F, vocab_size=V, dtype=np.float32)
    g = DistillGraph(B, every_n=every_n, feature_size=F, vocab_size=V, lstm_cells=H, device=DEV, seed=5)
    teacher, student = smoke.tower_params_numpy(g.teacher), smoke.tower_params_numpy(g.student)
    slots_t, slots_s = {}, {}
    xd, yd, nd = (torch.from_numpy(x).to(DEV), torch.from_numpy(labels.astype(np.uint8)).to(DEV),
                  torch.from_numpy(n).to(DEV))
    for it in range(3):
        g.step(xd, yd, nd)
        rep = g.loss_report()
        ref = mm.teacher_student_step(x.astype(np.float64), n, labels, teacher, student, every_n)
        for k in ("label_loss", "student_loss_state", "pred_loss", "student_label_loss"):
            assert abs(rep[k] - ref[k]) <= 2e-2 * abs(ref[k]) + 1e-6, (it, k, rep[k], float(ref[k]))
        teacher = mm.apply_train_op(teacher, ref["teacher_grads"], slots_t, it + 1, 1e-3, 1.0)
        student = mm.apply_train_op(student, ref["student_grads"], slots_s, it + 1, 1e-3, 1.0)
    assert g.global_step == 6                      # += 2 per iteration (README.md:116,121)


def test_high_precision_mode_after_training_steps():
    """After a few optimizer steps the LSTM states are O(1)-O(10) and plain bf16 operands no longer
    hold 1e-3 on the probabilities; the split-bf16 "high" precision forward does (same weights, same
    inputs, float64 oracle)."""
    from efficientvideoclassification_youtube8m_amd import smoke
    from efficientvideoclassification_youtube8m_amd.distill import DistillGraph
    B, F, H, V, every_n = 8, 128, 128, 64, 10
    q, x, n, labels = mm.synthetic_batch(B, seed=31, feature_size=F, vocab_size=V, dtype=np.float32)
    xd, yd, nd = (torch.from_numpy(x).to(DEV), torch.from_numpy(labels.astype(np.uint8)).to(DEV), torch.from_numpy(n).to(DEV))
    errs = {}
    for prec in ("bf16", "high"):
        g = DistillGraph(B, every_n=every_n, feature_size=F, vocab_size=V, lstm_cells=H, device=DEV, seed=9, precision=prec,
                         base_learning_rate=0.01)
        for _ in range(6):                      # grow the states / weights
            g.step(xd, yd, nd)
        out = g.step(xd, yd, nd, apply=False)
        teacher, student = smoke.tower_params_numpy(g.teacher), smoke.tower_params_numpy(g.student)
        ref = mm.teacher_student_step(x.astype(np.float64), n, labels, teacher, student, every_n, with_grads=False)
        errs[prec] = (float(np.abs(out["predictions"].cpu().numpy() - ref["teacher_predictions"]).max()),
                      float(np.abs(out["teacher_state"].cpu().numpy() - ref["teacher_state"]).max()),
                      float(np.abs(ref["teacher_state"]).max()))
    print("trained-state parity (pred err, state err, |state| max):", errs)
    assert errs["high"][0] < 1e-3 and errs["high"][1] < 1e-3 * max(1.0, errs["high"][2])
    assert errs["high"][0] < errs["bf16"][0]


_TRAINED = {}


def _train_in_child(tmp, B, seed, lr, max_steps, state_target, logit_target, dims="real", deterministic=True):
    """tests/_det_train.py in a fresh process (the library reads EVC_DETERMINISTIC once per process)."""
    import os
    import subprocess
    import sys
    env = dict(os.environ, EVC_DETERMINISTIC="1" if deterministic else "0")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tests", "_det_train.py"), tmp, str(B), str(seed), str(lr), str(max_steps),
                        str(state_target), str(logit_target), dims], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    return torch.load(tmp, weights_only=False)


def _trained_magnitude_weights(B, x, n, labels, lr=2e-3, max_steps=16, state_target=2.0, logit_target=8.0, deterministic=True):
    """Weights of trained magnitude: Adam iterations on the GPU (plain bf16 training) until the recurrent states have
    left the +-0.05 range of the reference's initialisation (|state| > state_target) or the MoE gate logits have
    grown past logit_target.  Returns the TF-named state dict of both towers.

    Trained in a child process under EVC_DETERMINISTIC=1 (no floating-point atomics: one workgroup per output tile, fixed-order
    reductions), on the batch synthetic_batch(B, seed=91): the SAME weights on every run and every box - the 1e-3 assertions of
    the callers run on ONE draw (scripts/precision_robustness.sh remains the many-draw margin study).  x, n, labels must be that
    batch (checked by the callers' oracle comparison)."""
    import os
    import tempfile
    key = (B, lr, max_steps, state_target, logit_target)
    if not deterministic:      # the margin study: a fresh draw per call (the default-mode training sums with atomics)
        with tempfile.TemporaryDirectory() as d:
            got = _train_in_child(os.path.join(d, "w.pt"), B, 91, lr, max_steps, state_target, logit_target, deterministic=False)
        return {k: v.to(DEV) for k, v in got["sd"].items()}
    if key not in _TRAINED:
        with tempfile.TemporaryDirectory() as d:
            got = _train_in_child(os.path.join(d, "w.pt"), B, 91, lr, max_steps, state_target, logit_target)
        print("trained-magnitude weights after %d deterministic Adam steps at lr %g: |state| %.2f |gate logit| %.2f" % (got["steps"], lr, got["s_max"], got["z_max"]))
        _TRAINED[key] = {k: v.to(DEV) for k, v in got["sd"].items()}
    return _TRAINED[key]


def test_deterministic_mode_gives_identical_bits_run_to_run():
    """EVC_DETERMINISTIC=1 (no split-K joins by atomics, bias gradients as fixed-order column sums of dz, one-block norm / loss
    reductions): two processes training the real-size towers for three iterations on the same batch end on IDENTICAL weights
    and report identical losses; without the switch the same two runs differ (the split-K joins of the weight-gradient products
    sum in arrival order) - which is what made every run of the tolerance test assert on a different draw (round 3)."""
    import os
    import tempfile
    runs = []
    with tempfile.TemporaryDirectory() as d:
        for i in range(2):
            runs.append(_train_in_child(os.path.join(d, "det%d.pt" % i), 64, 17, 1e-3, 3, 1e9, 1e9))
        plain = [_train_in_child(os.path.join(d, "plain%d.pt" % i), 64, 17, 1e-3, 3, 1e9, 1e9, deterministic=False) for i in range(2)]
    a, b = runs
    assert a["deterministic"] == "1" and a["steps"] == b["steps"] == 3
    for k in a["sd"]:
        assert torch.equal(a["sd"][k], b["sd"][k]), k
    assert {k: v for k, v in a["losses"].items() if k != "pred_loss"} == {k: v for k, v in b["losses"].items() if k != "pred_loss"}
    assert abs(a["losses"]["pred_loss"] - b["losses"]["pred_loss"]) <= 1e-5 * abs(b["losses"]["pred_loss"])      # (L_PRED's scalar: one atomic per video row)
    differ = sum(int(not torch.equal(plain[0]["sd"][k], plain[1]["sd"][k])) for k in plain[0]["sd"])
    print("default mode: %d of %d tensors differ between two runs" % (differ, len(plain[0]["sd"])))
    # deterministic and default mode compute the same function: a fraction of one Adam step apart after three iterations
    for k in a["sd"]:
        d = (a["sd"][k] - plain[0]["sd"][k]).abs()
        # (Adam's first steps move every weight by ~lr whatever the gradient's size: an element whose tiny gradient changes sign under
        #  the other summation order ends up to 2 lr per step apart - few elements, bounded by the three steps taken)
        assert float((d > 3e-4).float().mean()) < 2e-2 and d.max().item() < 6.5e-3, (k, float((d > 3e-4).float().mean()), d.max().item())


def test_real_dims_trained_magnitude_weights_both_precision_modes():
    """The north-star tolerance (1e-3 on the logits, absolute) at the REAL model size on weights of trained magnitude,
    in both forward modes, against the float64 oracle on the same weights and inputs:

    * "high" (bench.py times it as `precision_modes.high`: IEEE f16 operands in every forward product with the low-order halves of
      the weights - and of the input frames / the head's input - as e4m3 operands behind the f16 stages of the same launch)
      must hold 1e-3 on the gate logits, the expert logits, the states and the predictions of both towers;
    * "bf16" (one MFMA product: the mode of bench.py's headline figure, which north_star prescribes) is bounded
      RELATIVE to the logit magnitude: 2^-9 operand rounding over a K=4096..5120 contraction gives ~1e-3 * |z|, i.e.
      it meets the absolute 1e-3 only while |logits| <~ 1 (the reference's initialisation: 5e-5) - asserted here
      as 2e-3 * max(1, |z|_max) so that a regression shows (round 3, on run-to-run different weights: 1.2e-3 .. 1.9e-3 of |z|_max on
      the logits; round 4, on the ONE deterministic draw of _trained_magnitude_weights: 1.45e-3), and printed."""
    from efficientvideoclassification_youtube8m_amd import smoke
    from efficientvideoclassification_youtube8m_amd.distill import DistillGraph
    B = 4
    q, x, n, labels = mm.synthetic_batch(B, seed=91, dtype=np.float32)
    n[0] = 300
    x[np.arange(300)[None, :] >= n[:, None]] = 0.0
    sd = _trained_magnitude_weights(B, x, n, labels)
    params = {sc: {k[len(sc) + 1:]: v.double().cpu().numpy() for k, v in sd.items() if k.startswith(sc + "/")}
              for sc in ("model", "model_student")}
    ref = mm.teacher_student_step(x.astype(np.float64), n, labels, params["model"], params["model_student"], 10, with_grads=False)
    ref_logits = {}
    for sc, st in (("model", ref["teacher_state"]), ("model_student", ref["student_state"])):
        ref_logits[sc] = (st @ params[sc]["classifier/gates/weights"],
                          st @ params[sc]["classifier/experts/weights"] + params[sc]["classifier/experts/biases"])
    zmax = max(float(np.abs(a).max()) for pair in ref_logits.values() for a in pair)
    smax = max(float(np.abs(ref[k]).max()) for k in ("teacher_state", "student_state"))
    print("trained-magnitude weights: |logit| max %.2f, |state| max %.2f" % (zmax, smax))
    assert zmax > 1.0 and smax > 0.3, "the weights did not leave the initialisation regime"
    xd, yd, nd = (torch.from_numpy(x).to(DEV), torch.from_numpy(labels.astype(np.uint8)).to(DEV), torch.from_numpy(n).to(DEV))
    errs = {}
    for prec in ("bf16", "high"):
        g = DistillGraph(B, every_n=10, device=DEV, seed=3, precision=prec)
        g.teacher.load_state_dict(sd)
        g.student.load_state_dict(sd)
        out = g.step(xd, yd, nd, apply=False, num_frames_host=n)
        e = {}
        for name, tw, sc, kp, ks in (("teacher", g.teacher, "model", "predictions", "teacher_state"),
                                     ("student", g.student, "model_student", "student_predictions", "student_state")):
            rp = ref["teacher_predictions" if name == "teacher" else "student_predictions"]
            rs = ref[ks]
            e[name + "_pred"] = float(np.abs(out[kp].cpu().numpy() - rp).max())
            e[name + "_state"] = float(np.abs(out[ks].cpu().numpy() - rs).max())
            e[name + "_gate_logits"] = float(np.abs(tw.moe.gate_logits.cpu().numpy() - ref_logits[sc][0]).max())
            e[name + "_expert_logits"] = float(np.abs(tw.moe.expert_logits.cpu().numpy() - ref_logits[sc][1]).max())
        errs[prec] = e
        print("precision %-4s:" % prec, {k: "%.2e" % v for k, v in e.items()})
        del g
        torch.cuda.empty_cache()
    for k, v in errs["high"].items():
        assert v < 1e-3, ("high", k, v)
    for k, v in errs["bf16"].items():
        bound = 2e-3 * max(1.0, zmax if "logits" in k else (smax if "state" in k else 1.0))     # (round 4: one deterministic draw - measured 1.45e-3 |z| on the logits)
        bound = max(bound, 1e-2) if "state" in k else bound          # (cell states integrate the per-step rounding)
        assert v < bound, ("bf16", k, v, bound)
    assert all(errs["high"][k] < errs["bf16"][k] for k in errs["high"])


@pytest.mark.parametrize("frames", ["f32", "uint8"])
def test_headline_batch_256_high_mode_on_trained_magnitude_weights(frames):
    """The mode that carries north_star's tolerance, at the batch bench.py times it on (B = 256: ~3.6 k live L1 rows per step, the
    224 / 256-row f16 + e4m3 forward tiles, the 256-row head products), on weights of trained magnitude: the first 4 videos of the
    batch against the float64 oracle at 1e-3 on logits, states and predictions of both towers; every output of the 256 finite, the
    predictions in [0, 1]; the videos are independent (the B = 4 graph gives the same rows).  frames = "uint8" (round 6): the reader's bytes go in and
    layer 0 of both towers contracts them as exact integers (evc_l2norm_chunk_int, HLstmTower.x_int) - the same oracle values."""
    from efficientvideoclassification_youtube8m_amd.distill import DistillGraph
    q, x, n, labels = mm.synthetic_batch(4, seed=91, dtype=np.float32)
    n[0] = 300
    if frames == "uint8":                 # (video 0 now has 300 real frames: the oracle sees what the kernel dequantises)
        x = mm.dequantize(q.astype(np.float32)).astype(np.float32)
    x[np.arange(300)[None, :] >= n[:, None]] = 0.0
    sd = _trained_magnitude_weights(4, x, n, labels)
    params = {sc: {k[len(sc) + 1:]: v.double().cpu().numpy() for k, v in sd.items() if k.startswith(sc + "/")}
              for sc in ("model", "model_student")}
    ref = mm.teacher_student_step(x.astype(np.float64), n, labels, params["model"], params["model_student"], 10, with_grads=False)
    B = 256
    qb, xb, nb, lb = mm.synthetic_batch(B, seed=92, dtype=np.float32)
    xb[:4], nb[:4], lb[:4] = x, n, labels
    qb[:4] = q
    xd, yd, nd = (torch.from_numpy(qb if frames == "uint8" else xb).to(DEV), torch.from_numpy(lb.astype(np.uint8)).to(DEV), torch.from_numpy(nb).to(DEV))
    g = DistillGraph(B, every_n=10, device=DEV, seed=3, precision="high")
    g.teacher.load_state_dict(sd)
    g.student.load_state_dict(sd)
    out = g.step(xd, yd, nd, apply=False, num_frames_host=nb)
    assert g.teacher.fp8_lo() and g.teacher.l1.plan is not None and g.teacher.l1.Mrun < 20 * B
    assert g.teacher.x_int() and g.student.x_int() and g.teacher.act_lo()
    errs = {}
    for name, tw, sc, kp, ks in (("teacher", g.teacher, "model", "predictions", "teacher_state"),
                                 ("student", g.student, "model_student", "student_predictions", "student_state")):
        assert torch.isfinite(out[kp]).all() and torch.isfinite(out[ks]).all()
        assert float(out[kp].min()) >= 0.0 and float(out[kp].max()) <= 1.0
        st = ref[ks]
        zg = st @ params[sc]["classifier/gates/weights"]
        ze = st @ params[sc]["classifier/experts/weights"] + params[sc]["classifier/experts/biases"]
        errs[name + "_pred"] = float(np.abs(out[kp][:4].cpu().numpy() - ref["teacher_predictions" if name == "teacher" else "student_predictions"]).max())
        errs[name + "_state"] = float(np.abs(out[ks][:4].cpu().numpy() - st).max())
        errs[name + "_gate_logits"] = float(np.abs(tw.moe.gate_logits[:4].cpu().numpy() - zg).max())
        errs[name + "_expert_logits"] = float(np.abs(tw.moe.expert_logits[:4].cpu().numpy() - ze).max())
        assert all(v == 0 for v in tw.fp8_saturation(out[ks]).values()), tw.fp8_saturation(out[ks])     # the fixed e4m3 scales hold on these operands
    print("B = 256 high mode, first 4 videos vs float64:", {k: "%.2e" % v for k, v in errs.items()})
    for k, v in errs.items():
        assert v < 1e-3, (k, v)
    keep = {k: out[k][:4].clone() for k in ("predictions", "student_predictions", "teacher_state", "student_state")}
    g4 = DistillGraph(4, every_n=10, device=DEV, seed=3, precision="high")
    g4.teacher.load_state_dict(sd)
    g4.student.load_state_dict(sd)
    out4 = g4.step(xd[:4], yd[:4], nd[:4], apply=False, num_frames_host=nb[:4])
    for k, v in keep.items():
        d = (out4[k] - v).abs().max().item()
        assert d < 2e-4, (k, d)                          # (tile heights, split-K joins and accumulation order differ with the batch)
    # the teacher's L1 level as two-tile launches (evc_lstm_level2_fwd_high; measured not faster, off by default): the same bits through the engine
    from efficientvideoclassification_youtube8m_amd.engine import LstmStack
    full = {k: out[k].clone() for k in ("predictions", "student_predictions", "teacher_state", "student_state")}
    saved = LstmStack.fwd_walk2_high
    LstmStack.fwd_walk2_high = True
    try:
        out_w = g.step(xd, yd, nd, apply=False, num_frames_host=nb)
    finally:
        LstmStack.fwd_walk2_high = saved
    for k, v in full.items():
        assert torch.equal(out_w[k], v) if k.endswith("state") else float((out_w[k] - v).abs().max()) < 1e-6, k


LONG_STEPS = (16, 128, 512)
_LONG_DIR = {}


def _long_train_child(d, steps, seed):
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tests", "_long_train.py"), d, "16", "1e-3", ",".join(str(v) for v in steps)],
                       env=dict(os.environ, EVC_DETERMINISTIC="1", EVC_LONG_SEED=str(seed)), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    print(r.stdout[-600:])


@pytest.fixture(scope="module")
def long_trained(tmp_path_factory):
    """Real-size towers trained ONCE per session for 512 deterministic iterations (tests/_long_train.py in a child process:
    EVC_DETERMINISTIC=1, B = 16, the reference's lr 1e-3, labels from a fixed function of the input so the logits keep growing),
    checkpoints at 16 / 128 / 512 steps in a session directory."""
    if "dir" not in _LONG_DIR:
        d = str(tmp_path_factory.mktemp("long_horizon"))
        _long_train_child(d, LONG_STEPS, 3)
        _LONG_DIR["dir"] = d
    return _LONG_DIR["dir"]


@pytest.mark.parametrize("steps", LONG_STEPS)
def test_high_mode_holds_1e3_after_long_training(long_trained, steps):
    """The evidence horizon of north_star's tolerance (round 6).  Up to round 5 every "trained-magnitude" assertion used weights <= 16 Adam
    steps from initialisation.  Here: the real-size towers after 16 / 128 / 512 training iterations at the reference's learning rate
    (|state| up to ~16, |logit| up to ~37, |W| up to 0.32 at 512 steps - cell states far outside the +-7 the fixed e4m3 scale of the
    head's input could hold), 4 videos against the float64 oracle:
      * "high": gate logits, expert logits, states and predictions of BOTH towers < 1e-3 (absolute), and no e4m3 operand outside its range
        (fp8_saturation() all zero: the head's input takes its range from the batch, MoeHead.dynamic_fp8_range);
      * "bf16" (printed as a multiple of |z|; bounded at 2e-3 |z| so that a regression shows): ~7e-4 |z| on the logits, i.e. 2.5e-2 at 512 steps.
    Measured on this (deterministic) draw: see profiles/r06_long_horizon.txt; the margin over many draws: scripts/precision_robustness_long.sh."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import _long_train as lt
    ck = torch.load(os.path.join(long_trained, "step%d.pt" % steps), weights_only=False)
    assert ck["steps"] == steps and ck["deterministic"] == "1"
    mags, res = lt.evaluate(ck, ["bf16", "high", "high:u8"], 16)
    zmax, smax = max(mags["z_teacher"], mags["z_student"]), max(mags["s_teacher"], mags["s_student"])
    print("after %d steps: |z| teacher %.1f student %.1f, |state| teacher %.1f student %.1f, |W| %.3f" % (
        steps, mags["z_teacher"], mags["z_student"], mags["s_teacher"], mags["s_student"], mags["w_max"]))
    for mode in ("bf16", "high", "high:u8"):
        print("  %-7s" % mode, {k: "%.2e" % v for k, v in res[mode].items() if not k.endswith("saturated")})
    print("  bf16 logit error as a multiple of |z|: teacher %.2e, student %.2e" % (
        max(res["bf16"]["teacher_gate_logits"], res["bf16"]["teacher_expert_logits"]) / mags["z_teacher"],
        max(res["bf16"]["student_gate_logits"], res["bf16"]["student_expert_logits"]) / mags["z_student"]))
    assert zmax > 8.0 and smax > 4.0, "the towers did not leave the initialisation regime"
    if steps >= 128:
        assert smax > 7.0, "no state element beyond the fixed e4m3 range: the dynamic range is not exercised"
    for mode in ("high", "high:u8"):        # f32 frames, and the reader's uint8 frames (layer 0 on exact integers, HLstmTower.x_int)
        for k, v in res[mode].items():
            if k.endswith("saturated"):
                assert v == {}, (mode, k, v)
            else:
                assert v < 1e-3, (mode, steps, k, v)
    for k, v in res["bf16"].items():
        z = mags["z_teacher"] if k.startswith("teacher") else mags["z_student"]
        if "logits" in k:
            assert v < 2e-3 * max(1.0, z), ("bf16", steps, k, v, z)
    assert all(res["high"][k] < res["bf16"][k] for k in res["bf16"] if "logits" in k or "state" in k)


def test_high_mode_on_the_worst_of_six_long_horizon_draws(tmp_path):
    """Init seed 5 of tests/_long_train.py: the worst of six deterministic 512-step draws (profiles/r06_long_horizon_draws*.txt) - |z| 52.7, an
    ill-conditioned recurrence on which even the split-bf16 mode leaves 7e-4 on the logits and the round-5 "high" layout 2.2e-3.  Its error budget
    (profiles/r06_budget_worst_draw.txt) named three terms, each fixed in round 6:
      * the input frames' f16 rounding, whose e4m3 x e4m3 correction still left 1.8e-3  ->  the reader's uint8 frames as exact integers
        (evc_l2norm_chunk_int + the rescale behind the x-part of layer 0's K walk): asserted < 1e-3 on everything, both towers;
      * the uncorrected f16 rounding of h in L1 layer 0 and of both activation operands of L2 layer 1 (1.0e-3 / 1.2e-3)  ->  the h_lo forms;
      * (the student's plain-f16 L1 level: 9.5e-4 on the good draw)  ->  the teacher's layout for both towers.
    f32 frames keep the e4m3 correction of the input: 1.5e-3 on this draw - printed and bounded at 2.5e-3, the documented limit of feeding the
    "high" mode dequantised floats instead of the reader's bytes."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import _long_train as lt
    _long_train_child(str(tmp_path), (512,), 5)
    ck = torch.load(os.path.join(str(tmp_path), "step512.pt"), weights_only=False)
    mags, res = lt.evaluate(ck, ["high:u8", "high"], 16)
    print("worst draw: |z| teacher %.1f student %.1f, |state| teacher %.1f" % (mags["z_teacher"], mags["z_student"], mags["s_teacher"]))
    for mode in res:
        print("  %-7s" % mode, {k: "%.2e" % v for k, v in res[mode].items() if not k.endswith("saturated")})
    assert mags["z_teacher"] > 45.0, "not the draw this test was written for"
    for k, v in res["high:u8"].items():
        assert (v == {}) if k.endswith("saturated") else (v < 1e-3), ("high:u8", k, v)
    for k, v in res["high"].items():
        if not k.endswith("saturated"):
            assert v < (1e-3 if k.startswith("student") else (5e-3 if "state" in k else 2.5e-3)), ("high, f32 frames", k, v)


@pytest.mark.parametrize("frames", [[1, 14, 15, 16, 150, 299, 300], [300], [1], [0, 300, 0, 7], [300] * 8, [3] * 8])
def test_extreme_frame_counts_with_row_plans(frames):
    """Row plans at the edges: single video, every row alive, almost every row dead, zero-length videos."""
    from efficientvideoclassification_youtube8m_amd import smoke
    from efficientvideoclassification_youtube8m_amd.distill import DistillGraph
    B, F, H, V = len(frames), 64, 64, 24
    q, x, n, labels = mm.synthetic_batch(B, seed=5 + B, feature_size=F, vocab_size=V, dtype=np.float32)
    n[:] = frames
    x[np.arange(300)[None, :] >= n[:, None]] = 0.0
    g = DistillGraph(B, every_n=10, feature_size=F, vocab_size=V, lstm_cells=H, device=DEV, seed=2)
    assert g.row_plans
    out = g.step(torch.from_numpy(x).to(DEV), torch.from_numpy(labels.astype(np.uint8)).to(DEV), torch.from_numpy(n).to(DEV),
                 apply=False, num_frames_host=n)
    teacher, student = smoke.tower_params_numpy(g.teacher), smoke.tower_params_numpy(g.student)
    ref = mm.teacher_student_step(x.astype(np.float64), n, labels, teacher, student, 10, with_grads=True)
    assert np.abs(out["predictions"].cpu().numpy() - ref["teacher_predictions"]).max() < 1e-3
    assert np.abs(out["student_predictions"].cpu().numpy() - ref["student_predictions"]).max() < 1e-3
    assert np.abs(out["teacher_state"].cpu().numpy() - ref["teacher_state"]).max() < 2e-2
    for tower, key in ((g.teacher, "teacher_grads"), (g.student, "student_grads")):
        got = smoke.tower_grads_numpy(tower)
        for k in mm.HLSTM_PARAM_ORDER:
            gref = ref[key][k]
            if k in ("classifier/gates/weights", "classifier/experts/weights"):
                gref = gref - 2.0 * 1e-8 * smoke.tower_params_numpy(tower)[k]   # l2 term is folded in at apply time
            assert np.isfinite(got[k]).all()
            if np.abs(gref).max() > 1e-12:
                assert _rel(got[k], gref) < 4e-2, (tower.scope, k, _rel(got[k], gref))
            else:
                assert np.abs(got[k]).max() < 1e-9
    # the same step without row plans gives the same states (same per-row arithmetic)
    g2 = DistillGraph(B, every_n=10, feature_size=F, vocab_size=V, lstm_cells=H, device=DEV, seed=2)
    g2.row_plans = False
    out2 = g2.step(torch.from_numpy(x).to(DEV), torch.from_numpy(labels.astype(np.uint8)).to(DEV), torch.from_numpy(n).to(DEV),
                   apply=False, num_frames_host=n)
    # (equal up to accumulation order: without a plan a small L1 stack takes the wavefront form, whose second layer adds
    # the x-projection inside the MFMA accumulator instead of from a hoisted f32 product)
    for k in ("teacher_state", "student_state"):
        assert (out[k] - out2[k]).abs().max().item() < 1e-4 * max(1.0, out2[k].abs().max().item()), k


def test_fused_moe_update_matches_materialised_gradient_path():
    """evc_moe_grad_update (gradient tile recomputed inside the clip + Adam epilogue, both bf16 shadows written
    from it) against the plain path (weight-gradient GEMMs -> grad_sqnorm -> clip_adam -> transposes): same
    weights, moments, shadows and norms after two iterations."""
    from efficientvideoclassification_youtube8m_amd.distill import DistillGraph
    B, F, H, V = 8, 64, 64, 100                      # V*3 = 300, V*2 = 200: ragged last tiles, not multiples of 64
    q, x, n, labels = mm.synthetic_batch(B, seed=31, feature_size=F, vocab_size=V, dtype=np.float32)
    xd, nd, yd = torch.from_numpy(q).to(DEV), torch.from_numpy(n).to(DEV), torch.from_numpy(labels.astype(np.uint8)).to(DEV)
    graphs = []
    for fused in (True, False):
        g = DistillGraph(B, every_n=10, feature_size=F, vocab_size=V, lstm_cells=H, device=DEV, seed=4)
        g.teacher.fused_moe_update = g.student.fused_moe_update = fused
        for _ in range(2):
            g.step(xd, yd, nd, num_frames_host=n)
        torch.cuda.synchronize()
        graphs.append(g)
    a, b = graphs
    for ta, tb in ((a.teacher, b.teacher), (a.student, b.student)):
        assert ta.moe.can_fuse_update()
        for k in ta.names:
            pa, pb = ta.store.p(k), tb.store.p(k)
            assert (pa - pb).abs().max().item() < 2e-6, (ta.scope, k, (pa - pb).abs().max().item())
            ma, mb = ta.store.view(ta.store.m, k), tb.store.view(tb.store.m, k)
            assert (ma - mb).abs().max().item() <= 1e-4 * mb.abs().max().item() + 1e-12
        for k in (ta.GATES, ta.EXPERTS):
            assert torch.equal(ta.shadow_fwd[k], ta.store.p(k).bfloat16())
            sb = ta.shadow_bwd[k]
            assert torch.equal(sb[:, :ta.store.p(k).shape[0]], ta.store.p(k).t().bfloat16())
            assert bool((sb[:, ta.store.p(k).shape[0]:] == 0).all())
        assert torch.allclose(ta.sums, tb.sums, rtol=1e-4, atol=1e-12)


def test_moe_clip_norm_route_is_chosen_by_shape():
    """MoeHead.use_gram_norms (round 5): the Gram route at the headline's head (256 rows, K = 4096), pass 1 over the weights at cfg 4's
    (512 rows, K = 1024: the Gram products grow with rows^2, pass 1 with K - round 4 paid +0.10 ms per cfg-4 step for taking the Gram route
    everywhere), never a shape evc_gram_slabs would refuse, and the environment override."""
    from efficientvideoclassification_youtube8m_amd.engine import MoeHead

    class _Tw:
        device = DEV
    for B, K, want in ((256, 4096, True), (512, 1024, False), (256, 1024, False), (512, 4096, True)):
        m = MoeHead(_Tw(), K, 4716, 2)
        m.alloc(B, True)
        got = [m.use_gram_norms(m.Br, Vn, cols) for Vn, cols in ((4716 * 3, m.dgl_full.shape[1]), (4716 * 2, m.del_full.shape[1]))]
        assert got == [want, want], (B, K, got)
    m = MoeHead(_Tw(), 4096, 4716, 2)
    m.alloc(256, True)
    assert m.use_gram_norms(256, 14148, 14176 - 16) is False             # 14160 columns: not a multiple of 32 -> the two-pass form
    assert MoeHead.gram_slab_count(14176, 16) == 16 and MoeHead.gram_slab_count(32 * 10, 9) in range(1, 10) and MoeHead.gram_slab_count(40, 4) == 0
    for cols, want_s in ((32 * 10, 9), (32 * 17, 16), (32 * 33, 16)):
        S = MoeHead.gram_slab_count(cols, want_s)
        nk = cols // 32
        assert S >= 1 and ((nk + S - 1) // S) * (S - 1) < nk                # no empty last slab
    m.gram_force = False
    assert m.use_gram_norms(256, 14148, m.dgl_full.shape[1]) is False
    m.gram_force = True
    assert m.use_gram_norms(256, 14148, m.dgl_full.shape[1]) is True


@pytest.mark.parametrize("dims", [(8, 64, 64, 100), (40, 128, 128, 236)])
def test_gram_matrix_clip_norm_equals_the_pass_over_the_weights(dims):
    """evc_gram_slabs + evc_moe_grad_norms (|g + l2 W|^2 = <A A^T, X X^T> + 2 l2 <A, logits - bias> + l2^2 |W|^2, no pass over W)
    against pass 1 of evc_moe_grad_update (gradient tile recomputed, W streamed): the same norm sums to f32 rounding, the same
    weights after three iterations; the carried |W|^2 equals the norm of the weights it describes; run-to-run identical."""
    from efficientvideoclassification_youtube8m_amd.distill import DistillGraph
    B, F, H, V = dims
    q, x, n, labels = mm.synthetic_batch(B, seed=35, feature_size=F, vocab_size=V, dtype=np.float32)
    xd, nd, yd = torch.from_numpy(q).to(DEV), torch.from_numpy(n).to(DEV), torch.from_numpy(labels.astype(np.uint8)).to(DEV)
    graphs = []
    for gram in (True, False):
        g = DistillGraph(B, every_n=10, feature_size=F, vocab_size=V, lstm_cells=H, device=DEV, seed=4, regularization_penalty=2.0e4)
        for tw in (g.teacher, g.student):
            tw.moe.gram_norms = gram
            tw.moe.gram_force = True if gram else None           # (by shape these small heads would take pass 1: MoeHead.use_gram_norms)
        sums, first = [], {}
        for it in range(3):
            g.step(xd, yd, nd, num_frames_host=n)
            torch.cuda.synchronize()
            sums.append((g.teacher.sums.clone(), g.student.sums.clone()))
            if it == 0:
                first = {(tw.scope, k): tw.store.p(k).clone() for tw in (g.teacher, g.student) for k in (tw.GATES, tw.EXPERTS)}
        graphs.append((g, sums, first))
    (a, sa, fa), (b, sb, fb) = graphs
    for key in fa:            # one iteration from identical weights: the same update (later ones start from weights that carry the LSTM atomics' noise)
        assert (fa[key] - fb[key]).abs().max().item() < 2e-7, (key, (fa[key] - fb[key]).abs().max().item())
    for it in range(3):
        for t in range(2):
            # (regularization_penalty 2e4 makes the l2 terms a visible part of the norm: l2 = 2e-4)
            tw = (a.teacher, a.student)[t]
            rows = [tw.names.index(tw.GATES), tw.names.index(tw.EXPERTS)]       # (the LSTM rows carry the split-K atomics' run-to-run noise)
            tol = 2e-5 if it == 0 else 2e-3                                      # later iterations: on weights that differ by that noise
            assert torch.allclose(sa[it][t][rows], sb[it][t][rows], rtol=tol, atol=1e-12), (it, t, sa[it][t][rows], sb[it][t][rows])
    for ta, tb in ((a.teacher, b.teacher), (a.student, b.student)):
        for k in (ta.GATES, ta.EXPERTS):
            pa, pb = ta.store.p(k), tb.store.p(k)
            assert (pa - pb).abs().max().item() < 3e-4, (ta.scope, k, (pa - pb).abs().max().item())
        for i, k in enumerate((ta.GATES, ta.EXPERTS)):
            assert ta.moe._wsq_valid[i]
            want = float((ta.store.p(k).double() ** 2).sum())
            assert abs(float(ta.moe.wsq[i, 0]) - want) <= 1e-5 * want, (k, float(ta.moe.wsq[i, 0]), want)
        # fixed summation order: the same inputs give the same bits
        from efficientvideoclassification_youtube8m_amd import ops
        moe, outs = ta.moe, []
        for _ in range(2):
            o = torch.zeros(2, dtype=torch.float32, device=DEV)
            S = moe.gram_S
            ops.gram_slabs(moe.x_full, moe.Br, moe.K, S["x"], moe.gram_x)
            ops.gram_slabs(moe.dgl_full, moe.Br, moe.dgl_full.shape[1], 1, moe.gram_a)
            ops.moe_grad_norms(moe.gram_a, 1, moe.gram_x, S["x"], moe.Br, moe.dgl_full, moe.gate_logits, None, moe.B, moe.V * 3, 2e-4,
                               moe.wsq[0], moe.norm_part, o)
            outs.append(o)
        torch.cuda.synchronize()
        assert torch.equal(outs[0], outs[1]) and float(outs[0][0]) > 0
        # anything else that writes the weights drops the carried norm
        ta.load_state_dict(tb.state_dict())
        assert ta.moe._wsq_valid == [False, False]


def test_backward_phases_issued_in_readiness_order_give_the_same_step():
    """HLstmTower.backward_phases is a generator (one phase per MoE head / LSTM layer); DistillGraph issues the two towers' phases
    either tower after tower ("sequential") or interleaved in the order in which they become ready on the GPU ("interleaved": what
    the one-communicator data-parallel placement uses).  Same launches on the same streams: the same weights after
    three iterations (up to the atomics' run-to-run noise), the same global_step, and an unknown order is refused."""
    from efficientvideoclassification_youtube8m_amd.distill import DistillGraph
    B, F, H, V = 8, 128, 128, 100
    q, x, n, labels = mm.synthetic_batch(B, seed=37, feature_size=F, vocab_size=V, dtype=np.float32)
    xd, nd, yd = torch.from_numpy(q).to(DEV), torch.from_numpy(n).to(DEV), torch.from_numpy(labels.astype(np.uint8)).to(DEV)
    sds = []
    for order in ("sequential", "interleaved"):
        g = DistillGraph(B, every_n=10, feature_size=F, vocab_size=V, lstm_cells=H, device=DEV, seed=4)
        assert g.issue_order == "sequential"                      # one process: the default
        g.issue_order = order
        for _ in range(3):
            g.step(xd, yd, nd, num_frames_host=n)
        sd = {}
        for tw in (g.teacher, g.student):
            sd.update(tw.state_dict())
        torch.cuda.synchronize()
        assert g.global_step == 6
        sds.append(sd)
    for k in sds[0]:
        d = (sds[0][k] - sds[1][k]).abs().max().item()
        assert d < 3e-4, (k, d)
    g.issue_order = "alphabetical"
    with pytest.raises(KeyError):
        g.step(xd, yd, nd, num_frames_host=n)


@pytest.mark.parametrize("precision", ["bf16", "high"])
def test_deferred_updates_equal_immediate_updates(precision):
    """DistillGraph.defer_updates (the MoE-head and L2-level updates of step k enqueued at the start of step k+1, under its L1
    forward) against the immediate schedule: same weights, moments and operand shadows after three iterations + flush(); the
    state of a deferred graph is complete only after flush() (state_dict() flushes).  cs/train.py:516-517: both train ops read
    the pre-update weights of the iteration - the deferral only moves the update later in wall time."""
    from efficientvideoclassification_youtube8m_amd.distill import DistillGraph
    B, F, H, V = 8, 128, 128, 100
    q, x, n, labels = mm.synthetic_batch(B, seed=33, feature_size=F, vocab_size=V, dtype=np.float32)
    xd, nd, yd = torch.from_numpy(q).to(DEV), torch.from_numpy(n).to(DEV), torch.from_numpy(labels.astype(np.uint8)).to(DEV)
    graphs = []
    for defer in (True, False):
        g = DistillGraph(B, every_n=10, feature_size=F, vocab_size=V, lstm_cells=H, device=DEV, seed=4, precision=precision)
        g.defer_updates = defer
        for _ in range(3):
            out = g.step(xd, yd, nd, num_frames_host=n)
        if defer:
            assert g.teacher._deferred and g.student._deferred, "the last step's MoE / L2-level updates are still pending"
        sd = {}
        for tw in (g.teacher, g.student):
            sd.update(tw.state_dict())                     # (flushes)
        assert not g.teacher._deferred and g.teacher._deferred_ev is None
        torch.cuda.synchronize()
        graphs.append((g, sd, {k: float(v) for k, v in g.loss_report().items()}))
    (a, sda, la), (b, sdb, lb) = graphs
    assert a.global_step == b.global_step == 6
    for k in la:
        assert abs(la[k] - lb[k]) <= 2e-3 * abs(lb[k]) + 1e-6, (k, la[k], lb[k])      # (atomics in the split-K joins: not bit-equal)
    for k in sda:
        d = (sda[k] - sdb[k]).abs().max().item()
        assert d < 3e-4, (k, d)                            # a fraction of one Adam step (lr 1e-3)
    for ta, tb in ((a.teacher, b.teacher), (a.student, b.student)):
        for k in ta.names:
            ma, mb = ta.store.view(ta.store.m, k), tb.store.view(tb.store.m, k)
            assert (ma - mb).abs().max().item() <= 2e-2 * mb.abs().max().item() + 1e-12, k
        for k, sh in ta.shadow_fwd.items():                # the shadows follow the masters in both schedules
            if k in getattr(ta, "shadow_w16", {}):         # ("high" head on f16 + e4m3 images: its bf16 forward shadow is not kept current, round 5)
                assert torch.equal(ta.shadow_w16[k], ta.store.p(k).half()), k
                continue
            assert torch.equal(sh, ta.store.p(k).bfloat16()), k
        assert ta.adam_t == tb.adam_t == 3


@pytest.mark.parametrize("H", [64, 128])
def test_fused_moe_update_writes_the_forward_operand_images_in_high_precision(H):
    """"high" precision, one process: the fused MoE update's epilogue also writes the forward operand images of the new weights
    (evc_moe_grad_update_wide) instead of separate passes over the f32 weights - H = 64 (K = 256): the wide [hi | lo] split-bf16 image
    (the split-bf16 head: K is below the e4m3 head's 512); H = 128 (K = 512): the f16 image and [e4m3(W_lo 2^18) | e4m3(W 2^7)] of
    ops.gemm_nt_f16_fp8.  After two training iterations they must be exactly what the cast kernels make of the master weights."""
    from efficientvideoclassification_youtube8m_amd import ops
    from efficientvideoclassification_youtube8m_amd.distill import DistillGraph
    B, F, V = 8, 64, 100
    q, x, n, labels = mm.synthetic_batch(B, seed=33, feature_size=F, vocab_size=V, dtype=np.float32)
    xd, nd, yd = torch.from_numpy(q).to(DEV), torch.from_numpy(n).to(DEV), torch.from_numpy(labels.astype(np.uint8)).to(DEV)
    g = DistillGraph(B, every_n=10, feature_size=F, vocab_size=V, lstm_cells=H, device=DEV, seed=4, precision="high")
    for _ in range(2):
        g.step(xd, yd, nd, num_frames_host=n)
    torch.cuda.synchronize()
    for tw in (g.teacher, g.student):
        for k in (tw.GATES, tw.EXPERTS):
            p = tw.store.p(k)
            K = p.shape[1]
            hi = p.bfloat16()
            if H == 64:
                assert torch.equal(tw.shadow_fwd[k], hi)
            else:       # (round 5) a head that reads f16 + e4m3 images has no use for the bf16 forward shadow: the update no longer writes
                # it (2 of 34 bytes per parameter); refresh_shadows() / set_precision("bf16") rebuild it from the masters - checked below
                assert not torch.equal(tw.shadow_fwd[k], hi)
            if H == 64:
                assert k in tw.shadow_w and k not in tw.shadow_w8
                assert torch.equal(tw.shadow_w[k][:, :K], hi), (tw.scope, k)
                assert torch.equal(tw.shadow_w[k][:, K:], (p - hi.float()).bfloat16()), (tw.scope, k)
            else:
                assert k in tw.shadow_w8 and k not in tw.shadow_w
                assert torch.equal(tw.shadow_w16[k], p.half()), (tw.scope, k)
                want = torch.empty_like(tw.shadow_w8[k])
                ops.cast_fp8_lo(p, want, hi_cols=K, scale_exp=ops.FP8_MOE["w_lo_exp"], hi_exp=ops.FP8_MOE["w_hi_exp"])
                assert torch.equal(tw.shadow_w8[k], want), (tw.scope, k)
    assert all(np.isfinite(v) for v in g.loss_report().values())
    for tw in (g.teacher, g.student):           # back to bf16: every forward shadow is rebuilt from the masters
        tw.set_precision("bf16")
        for k in (tw.GATES, tw.EXPERTS):
            assert torch.equal(tw.shadow_fwd[k], tw.store.p(k).bfloat16())


def test_high_precision_dithered_layer_images_follow_the_weights():
    """"high" precision at dims where the L1 level runs on f16 + e4m3 stages (F, H multiples of 128, >= 384): by default the TOP layer of the
    teacher's L1 level contracts time-dithered weight images (engine.HLstmTower.dither_layers) - 15 images of its kernel, rebuilt behind every
    update (evc_lstm_adam_fused + evc_cast_f32_to_f16_dither) and on load_state_dict.  After two training iterations they must be the oracle's
    images of the NEW master weights, bit for bit (oracle/lowprec.py::f16_dither_images); layer 0 keeps its f16 + e4m3 images (round 6: [lo | hi] of both
    parts, cast_fp8_lo(hi_tail=True)); the student has the same layout on its 6-step chunks (round 6; before: plain f16); the resolved layout names the
    dithered layer; a dithered layer BELOW a corrected one is refused."""
    from oracle import lowprec as lp
    from efficientvideoclassification_youtube8m_amd import ops
    from efficientvideoclassification_youtube8m_amd.distill import DistillGraph
    B, F, V, H = 8, 384, 100, 384
    q, x, n, labels = mm.synthetic_batch(B, seed=35, feature_size=F, vocab_size=V, dtype=np.float32)
    xd, nd, yd = torch.from_numpy(q).to(DEV), torch.from_numpy(n).to(DEV), torch.from_numpy(labels.astype(np.uint8)).to(DEV)
    g = DistillGraph(B, every_n=10, feature_size=F, vocab_size=V, lstm_cells=H, device=DEV, seed=4, precision="high")
    tw = g.teacher
    if "EVC_HIGH_DITHER_LAYERS" in os.environ:
        pytest.skip("the default layout is under test")
    assert tw.fp8_lo() and tw.dither_layers() == (1,) and g.student.dither_layers() == (1,)       # (round 6: the student takes the teacher's layout)
    assert "layers [1] on 15 time-dithered" in tw.precision_layout()["l1"]
    k0, k1 = (k for k in tw.names if k.startswith("RNN_L1/") and k.endswith("kernel"))
    assert set(tw.shadow16d) == {k1} and k0 in tw.shadow16 and k0 in tw.shadow8 and k1 not in tw.shadow8
    assert tw.act_lo() and tw.shadow8[k0].shape[1] == 2 * tw.store.p(k0).shape[1]            # [lo(Wx) | hi(Wx) | lo(Wh) | hi(Wh)]
    out = None
    for _ in range(2):
        out = g.step(xd, yd, nd, num_frames_host=n)
    torch.cuda.synchronize()
    p1 = tw.store.p(k1)
    want = lp.f16_dither_images(p1.cpu().numpy(), tw.l1_steps(), tw.dither_seed(k1))
    assert np.array_equal(tw.shadow16d[k1].cpu().numpy().view(np.uint16), want.view(np.uint16))
    assert torch.equal(tw.shadow16[k0], tw.store.p(k0).half())
    want8 = torch.empty_like(tw.shadow8[k0])                        # layer 0's e4m3 rows from the update's epilogue = the cast of the new weights
    ops.cast_fp8_lo(tw.store.p(k0), want8, hi_cols=F, hi_tail=True)
    assert torch.equal(tw.shadow8[k0], want8)
    for l2k in (k for k in tw.names if k.startswith("RNN_L2/") and k.endswith("kernel") and k in tw.shadow8):      # (the L2 level's e4m3 form needs H >= 512)
        w8 = torch.empty_like(tw.shadow8[l2k])
        pk = tw.store.p(l2k)
        if "cell_0" in l2k:
            ops.cast_fp8_lo(pk[:, pk.shape[1] - H:], w8, hi_tail=True)
        else:
            ops.cast_fp8_lo(pk, w8, hi_cols=H, hi_tail=True)
        assert torch.equal(tw.shadow8[l2k], w8), l2k
    sd = tw.state_dict()
    tw.shadow16d[k1].zero_()
    tw.load_state_dict(sd)
    torch.cuda.synchronize()
    assert np.array_equal(tw.shadow16d[k1].cpu().numpy().view(np.uint16), want.view(np.uint16))
    assert all(np.isfinite(v) for v in g.loss_report().values())
    with pytest.raises(ValueError):                     # a dithered layer below a corrected one: refused
        tw.f16_dither_layers = (0,)
        tw.dither_layers()
    tw.f16_dither_layers = None


def test_moe_update_in_two_phases_on_row_slabs_equals_the_whole():
    """evc_moe_grad_update_phase: phase 1 on every row slab (norm sums accumulate), then phase 2 on every slab, gives
    the weights / moments / shadows of the one-call update of the whole matrix - what the data-parallel ranks do, each
    on its own slab, with an 8-byte all-reduce of the sums in between."""
    from efficientvideoclassification_youtube8m_amd import ops
    torch.manual_seed(5)
    rows, V, K = 64, 600, 256                       # 5 row tiles: slabs of 384 and 216 rows
    Vp = (V + 63) // 64 * 64
    dlog = torch.zeros(rows, Vp, dtype=torch.bfloat16, device=DEV)
    dlog[:, :V] = (torch.randn(rows, V, device=DEV) * 3e-2).to(torch.bfloat16)      # clipping active: the global norm matters
    x = (torch.randn(rows, K, device=DEV) * 0.5).to(torch.bfloat16)
    p0, m0, v0 = torch.randn(V, K, device=DEV) * 0.05, torch.randn(V, K, device=DEV) * 1e-3, torch.rand(V, K, device=DEV) * 1e-5
    ws = torch.empty(2 * ((V + 127) // 128) * ((K + 127) // 128), device=DEV)

    def fresh():
        return (p0.clone(), m0.clone(), v0.clone(), torch.zeros(V, K, dtype=torch.bfloat16, device=DEV),
                torch.zeros(K, Vp, dtype=torch.bfloat16, device=DEV), torch.zeros(2, device=DEV))

    pa, ma, va, pba, pTa, sa = fresh()
    ops.moe_grad_update(dlog, x, rows, V, K, pa, ma, va, pba, pTa, 2e-8, sa, ws, 1.0, 1e-3)
    pb, mb, vb, pbb, pTb, sb = fresh()
    slabs = [(0, 384), (384, 216)]
    for phase in (1, 2):
        for v_lo, vs in slabs:
            ops.moe_grad_update(dlog[:, v_lo:], x, rows, vs, K, pb[v_lo:], mb[v_lo:], vb[v_lo:], pbb[v_lo:], pTb[:, v_lo:], 2e-8,
                                sb, ws, 1.0, 1e-3, phase=phase)
    assert sa[0].item() > 1.0                                                     # the clip is active
    assert torch.allclose(sa, sb, rtol=1e-5)
    assert (pa - pb).abs().max().item() < 1e-7 and (ma - mb).abs().max().item() < 1e-8 and (va - vb).abs().max().item() < 1e-10
    assert torch.equal(pba, pa.bfloat16()) and torch.equal(pbb, pb.bfloat16())       # each run's shadows are its own weights
    assert torch.equal(pTa[:, :V], pa.t().bfloat16()) and torch.equal(pTb[:, :V], pb.t().bfloat16())
    assert (pba.float() - pbb.float()).abs().max().item() <= 2.0 ** -8 * pb.abs().max().item()
    # a phase outside 0..2 is refused
    with pytest.raises(Exception):
        ops.moe_grad_update(dlog, x, rows, V, K, pb, mb, vb, pbb, pTb, 2e-8, sb, ws, 1.0, 1e-3, phase=3)


def test_fused_upper_layer_gradient_in_bptt_matches_hoisted_path_and_oracle():
    """Two-layer L1 stacks with >= 1024 chunk rows contract the upper layer's dX inside the lower layer's BPTT steps
    ([dz0_{t+1} | dz1_t] . [Wh0 ; Wx1]^T): "fused" (layer after layer, the product's default) and "pair" (wavefront order,
    evc_lstm_stack2_bwd) against the hoisted dX product with its bf16 round trip ("off") and the float64 oracle."""
    from efficientvideoclassification_youtube8m_amd import smoke
    from efficientvideoclassification_youtube8m_amd.distill import DistillGraph
    from efficientvideoclassification_youtube8m_amd.engine import LstmStack
    B, F, H, V = 64, 64, 128, 40                    # teacher L1: 20 x 64 = 1280 chunk rows (row-planned: fewer live)
    q, x, n, labels = mm.synthetic_batch(B, seed=17, feature_size=F, vocab_size=V, dtype=np.float32)
    n[:3] = (300, 1, 0)
    n[3:40] = 300                                   # enough live chunk rows for the fused path (>= 1024)
    x[np.arange(300)[None, :] >= n[:, None]] = 0.0
    xd, yd, nd = (torch.from_numpy(x).to(DEV), torch.from_numpy(labels.astype(np.uint8)).to(DEV), torch.from_numpy(n).to(DEV))
    grads = {}
    saved = LstmStack.bwd_fuse
    try:
        for mode in ("fused", "pair", "off"):
            LstmStack.bwd_fuse = mode
            g = DistillGraph(B, every_n=10, feature_size=F, vocab_size=V, lstm_cells=H, device=DEV, seed=4)
            g.step(xd, yd, nd, apply=False, num_frames_host=n)
            assert g.teacher.l1.Mrun >= 1024 and g.student.l1.Mrun < 1024       # the student's L1 (5 x 64 rows) stays on the hoisted form
            grads[mode] = smoke.tower_grads_numpy(g.teacher)
            if mode == "fused":
                teacher, student = smoke.tower_params_numpy(g.teacher), smoke.tower_params_numpy(g.student)
    finally:
        LstmStack.bwd_fuse = saved
    ref = mm.teacher_student_step(x.astype(np.float64), n, labels, teacher, student, 10)["teacher_grads"]
    for k in mm.HLSTM_PARAM_ORDER:
        if not k.startswith("RNN_L1"):
            continue
        for mode in ("fused", "pair"):
            a, b = grads[mode][k], grads["off"][k]
            # (the hoisted path rounds the upper layer's dX to bf16 before adding it; the fused forms keep it in the f32 accumulator)
            assert _rel(a, b) < 1e-2, (mode, k, _rel(a, b))
            assert _rel(a, ref[k]) < 3e-2 and _rel(a, ref[k]) <= _rel(b, ref[k]) * 1.5 + 1e-3, (mode, k, _rel(a, ref[k]), _rel(b, ref[k]))
        assert _rel(grads["fused"][k], grads["pair"][k]) < 1e-3, k


def test_precision_modes_on_amplifying_weights():
    """The three forward modes against the float64 oracle on weights scaled up (x3 LSTM kernels at H = 128: a recurrence that
    amplifies every rounding, |state| ~ 4) so that plain bf16 is visibly off, with >= 1024 live L1 chunk rows (ring tiles):

    * "split" - split-bf16 operands as K-extensions of the plain loops in EVERY forward GEMM (hoisted split x-projection +
      K = 3H recurrent steps, evc_lstm_layer_fwd_hp; no row plans) - holds f32-operand accuracy: 1e-4 on the states here;
    * "high"  - the budgeted mode (f16 L1 level with the K-extended input part, split-bf16 L2 level and MoE head): its contract is
      north_star's 1e-3 at the real model size on trained-magnitude weights
      (test_real_dims_trained_magnitude_weights_both_precision_modes); on this amplifying recurrence it must still sit an order
      of magnitude below bf16;
    * "bf16"  - one product per contraction."""
    from efficientvideoclassification_youtube8m_amd import smoke
    from efficientvideoclassification_youtube8m_amd.distill import DistillGraph
    B, F, H, V = 64, 64, 128, 40                    # teacher L1: 20 x 64 = 1280 chunk rows
    q, x, n, labels = mm.synthetic_batch(B, seed=23, feature_size=F, vocab_size=V, dtype=np.float32)
    n[3:44] = 300
    x[np.arange(300)[None, :] >= n[:, None]] = 0.0
    xd, yd, nd = (torch.from_numpy(x).to(DEV), torch.from_numpy(labels.astype(np.uint8)).to(DEV), torch.from_numpy(n).to(DEV))
    errs = {}
    for prec in ("split", "high", "bf16"):
        g = DistillGraph(B, every_n=10, feature_size=F, vocab_size=V, lstm_cells=H, device=DEV, seed=6, precision=prec)
        for tw in (g.teacher, g.student):           # O(1) recurrent states: scale the LSTM kernels up
            for k in tw.names:
                if k.endswith("basic_lstm_cell/kernel"):
                    tw.store.p(k).mul_(3.0)
            tw.refresh_shadows()
        out = g.step(xd, yd, nd, apply=False, num_frames_host=n)
        assert g.teacher.l1.Mrun >= 1024
        teacher, student = smoke.tower_params_numpy(g.teacher), smoke.tower_params_numpy(g.student)
        ref = mm.teacher_student_step(x.astype(np.float64), n, labels, teacher, student, 10, with_grads=False)
        errs[prec] = (float(np.abs(out["teacher_state"].cpu().numpy() - ref["teacher_state"]).max()),
                      float(np.abs(out["predictions"].cpu().numpy() - ref["teacher_predictions"]).max()),
                      float(np.abs(ref["teacher_state"]).max()),
                      float(np.abs(out["student_state"].cpu().numpy() - ref["student_state"]).max()))
        if prec != "bf16":                          # the backward pass is the bf16 one in every mode: gradients stay finite and sane
            gt = smoke.tower_grads_numpy(g.teacher)
            assert all(np.isfinite(v).all() for v in gt.values())
    print("precision modes on amplifying weights (teacher state err, pred err, |state|, student state err):", errs)
    assert errs["split"][0] < 1e-4 * max(1.0, errs["split"][2]) and errs["split"][1] < 2e-5 and errs["split"][3] < 1e-4
    assert errs["high"][0] < errs["bf16"][0] / 8 and errs["high"][1] < errs["bf16"][1] / 5
    assert errs["bf16"][0] > 30 * errs["split"][0]
