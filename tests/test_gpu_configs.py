"""The BASELINE.json configurations that round 1 left without a GPU parity test, each against the float64
oracle on the same inputs (pytest -m gpu):

  cfg 1  FrameLevelLogisticModel, 5 tfrecords, batch 32, through train.main           (cs/train.py:129-176,281-334)
  cfg 2  HierarchicalLstmModel teacher only, real dims                                (cs/frame_level_models.py:200-267)
  cfg 4  DbofModel (cluster 8192, hidden 1024) + MoE(2), through train.main           (cs/frame_level_models.py:108-195)
  cfg 5  student-only fine-tune, every_n=30 (10 frames), real dims, B=64 and B=1024   (cs/train_finetune.py:243-318)
  + the reference's default every_n=1 (student on all 300 frames, frame-count quirk)  (cs/train.py:262-272)
"""
import numpy as np
import pytest
import torch

from oracle import model_math as mm

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _rel(a, b):
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))


def _rel2(a, b):
    """Relative L2 error of a whole tensor (the max-based _rel is dominated by the bf16 rounding of single large entries)."""
    return float(np.linalg.norm(np.asarray(a, np.float64) - b) / (np.linalg.norm(b) + 1e-30))


def _dev(x, n, labels):
    return (torch.from_numpy(x).to(DEV), torch.from_numpy(labels.astype(np.uint8)).to(DEV), torch.from_numpy(n).to(DEV))


# ---------------------------------------------------------------------------------------------------------------
# cfg 5: student only, every_n = 30 -> S = 10 frames, 5 chunks of 2; L1 runs at 5*B rows x 2 steps, L2 at B x 5
# ---------------------------------------------------------------------------------------------------------------
def test_cfg5_student_only_every_n_30_real_dims():
    from efficientvideoclassification_youtube8m_amd import smoke
    from efficientvideoclassification_youtube8m_amd.distill import DistillGraph
    B, every_n = 64, 30
    q, x, n, labels = mm.synthetic_batch(B, seed=55, dtype=np.float32)
    n[:4] = (300, 29, 30, 1)                        # n_S = 10, 0, 1, 0: full, empty and one-frame students
    x[np.arange(300)[None, :] >= n[:, None]] = 0.0
    g = DistillGraph(B, every_n=every_n, mode="student", device=DEV, seed=3)
    assert g.teacher is None and g.student.T == 10 and g.student.C == 5
    student = smoke.tower_params_numpy(g.student)
    out = g.step(*_dev(x, n, labels), apply=False, num_frames_host=n)
    # oracle: cs/train_finetune.py:243-318 = l2norm, gather, student tower, CE only
    xn = mm.l2_normalize(x.astype(np.float64), 2)
    n_s = mm.student_num_frames(n, every_n)
    assert np.array_equal(out["num_frames_student"].cpu().numpy(), n_s)                      # bit-exact (int64)
    s_state, s_pred, cache = mm.hlstm_fwd(mm.subsample_frames(xn, every_n), n_s, student, 5)
    e_p = np.abs(out["student_predictions"].cpu().numpy() - s_pred).max()
    e_s = np.abs(out["student_state"].cpu().numpy() - s_state).max()
    print("cfg5 B=64: student pred err %.2e state err %.2e" % (e_p, e_s))
    assert e_p < 1e-3 and e_s < 1e-3
    ce = mm.cross_entropy_loss(s_pred, labels.astype(np.float64))
    assert abs(g.loss_report()["student_label_loss"] - ce) < 1e-4 * ce
    grads = mm.hlstm_bwd(None, mm.cross_entropy_grad(s_pred, labels.astype(np.float64)), cache)
    got = smoke.tower_grads_numpy(g.student)
    l2s = {}
    for k in mm.HLSTM_PARAM_ORDER:
        assert _rel(got[k], grads[k]) < 3e-2, (k, _rel(got[k], grads[k]))
        l2s[k] = _rel2(got[k], grads[k])
        assert l2s[k] < 1.2e-2, (k, l2s[k])
    print("cfg5 gradient relative L2:", {k.split("/")[0] + "/" + k.split("/")[-1] + k.split("/")[-3][-2:]: round(v, 4) for k, v in l2s.items()})
    # the update: one train op -> global_step += 1 (cs/train_finetune.py:316-318)
    g.apply_gradients(B)
    assert g.global_step == 1


def test_cfg5_batch_1024_properties():
    """BASELINE cfg 5 at its full batch (L1 5120 rows x 2 steps, L2 1024 x 5): frame counts bit-exact, finite outputs,
    rows are independent (the first 64 videos give the B=64 result), two iterations run."""
    from efficientvideoclassification_youtube8m_amd.distill import DistillGraph
    B, every_n = 1024, 30
    rng = np.random.default_rng(5)
    q = rng.integers(0, 256, (B, 300, 1152), dtype=np.uint8)
    n = rng.integers(1, 301, B).astype(np.int32)
    labels = np.zeros((B, 4716), np.uint8)
    labels[np.arange(B)[:, None], rng.integers(0, 4716, (B, 3))] = 1
    qd, yd, nd = torch.from_numpy(q).to(DEV), torch.from_numpy(labels).to(DEV), torch.from_numpy(n).to(DEV)
    g = DistillGraph(B, every_n=every_n, mode="student", device=DEV, seed=3)
    out = g.step(qd, yd, nd, apply=False, num_frames_host=n)
    assert np.array_equal(out["num_frames_student"].cpu().numpy(), mm.student_num_frames(n, every_n))
    pred = out["student_predictions"].clone()
    assert torch.isfinite(pred).all() and torch.isfinite(out["student_state"]).all()
    assert float(pred.min()) >= 0.0 and float(pred.max()) <= 1.0
    g64 = DistillGraph(64, every_n=every_n, mode="student", device=DEV, seed=3)
    out64 = g64.step(qd[:64], yd[:64], nd[:64], apply=False, num_frames_host=n[:64])
    assert (out64["student_predictions"] - pred[:64]).abs().max().item() < 1e-5              # (tile shapes differ with B)
    for _ in range(2):
        g.step(qd, yd, nd, num_frames_host=n)
    assert g.global_step == 2 and all(np.isfinite(v) for v in g.loss_report().values())


# ---------------------------------------------------------------------------------------------------------------
# cfg 3 at the headline batch: teacher + student, every_n = 10, B = 256 (what bench.py times)
# ---------------------------------------------------------------------------------------------------------------
def test_cfg3_headline_batch_256_properties():
    """BASELINE cfg 3 at the batch the metric is quoted on (teacher L1 5120 chunk rows x 15 steps with row plans, L2 256 x 20,
    student 1280 x 6 / 256 x 5): frame counts bit-exact, finite outputs in [0, 1], the videos are independent - the first 8 videos
    give the B = 8 graph's predictions and states although every tile shape, row plan and launch geometry differs - the
    losses are the oracle's on a 4-video slice, and two training iterations run (global_step += 2 each)."""
    from efficientvideoclassification_youtube8m_amd import smoke
    from efficientvideoclassification_youtube8m_amd.distill import DistillGraph
    B, every_n = 256, 10
    rng = np.random.default_rng(17)
    q = rng.integers(0, 256, (B, 300, 1152), dtype=np.uint8)
    n = rng.integers(120, 301, B).astype(np.int32)
    n[:6] = (300, 1, 9, 10, 299, 150)
    labels = np.zeros((B, 4716), np.uint8)
    labels[np.arange(B)[:, None], rng.integers(0, 4716, (B, 3))] = 1
    qd, yd, nd = torch.from_numpy(q).to(DEV), torch.from_numpy(labels).to(DEV), torch.from_numpy(n).to(DEV)
    g = DistillGraph(B, every_n=every_n, device=DEV, seed=3)
    out = g.step(qd, yd, nd, apply=False, num_frames_host=n)
    assert g.teacher.l1.plan is not None and g.teacher.l1.Mrun < 20 * B            # the padding rows are gone
    assert np.array_equal(out["num_frames_student"].cpu().numpy(), mm.student_num_frames(n, every_n))
    keep = {k: out[k].clone() for k in ("predictions", "student_predictions", "teacher_state", "student_state")}
    for k, v in keep.items():
        assert torch.isfinite(v).all(), k
    for k in ("predictions", "student_predictions"):
        assert float(keep[k].min()) >= 0.0 and float(keep[k].max()) <= 1.0
    g8 = DistillGraph(8, every_n=every_n, device=DEV, seed=3)
    out8 = g8.step(qd[:8], yd[:8], nd[:8], apply=False, num_frames_host=n[:8])
    for k, v in keep.items():
        d = (out8[k] - v[:8]).abs().max().item()
        assert d < 2e-5, (k, d)                                                     # (accumulation order differs with the tiles)
    # the first 4 videos against the float64 oracle (dequantised as the reader does, cs/utils.py:22-25)
    x4 = mm.dequantize(q[:4].astype(np.float64))
    x4[np.arange(300)[None, :] >= n[:4, None]] = 0.0
    teacher, student = smoke.tower_params_numpy(g.teacher), smoke.tower_params_numpy(g.student)
    ref = mm.teacher_student_step(x4, n[:4], labels[:4].astype(bool), teacher, student, every_n, with_grads=False)
    assert np.abs(keep["predictions"][:4].cpu().numpy() - ref["teacher_predictions"]).max() < 1e-3
    assert np.abs(keep["student_predictions"][:4].cpu().numpy() - ref["student_predictions"]).max() < 1e-3
    assert np.abs(keep["teacher_state"][:4].cpu().numpy() - ref["teacher_state"]).max() < 1e-3
    for _ in range(2):
        g.step(qd, yd, nd, num_frames_host=n)
    assert g.global_step == 4 and all(np.isfinite(v) for v in g.loss_report().values())


# ---------------------------------------------------------------------------------------------------------------
# --lstm_layers 1: the reference's flag DEFAULT (cs/frame_level_models.py:39; its launchers pass 2)
# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("precision", ["bf16", "high", "split"])
def test_single_layer_stacks_all_precisions(precision):
    """One BasicLSTMCell per level (state = [c | h], 2H wide; L2 input and MoE input 2H): teacher + student step against the
    float64 oracle - forward in every precision mode (the "high" mode's f16 wavefront needs two layers: a one-layer L2 level takes
    the split-bf16 layer instead), gradients and the update in bf16."""
    from efficientvideoclassification_youtube8m_amd import smoke
    from efficientvideoclassification_youtube8m_amd.distill import DistillGraph
    B, F, H, V = 8, 128, 128, 40
    q, x, n, labels = mm.synthetic_batch(B, seed=12, feature_size=F, vocab_size=V, dtype=np.float32)
    g = DistillGraph(B, every_n=10, feature_size=F, vocab_size=V, lstm_cells=H, lstm_layers=1, device=DEV, seed=4, precision=precision)
    assert g.teacher.L == 1 and g.teacher.K == 2 * H
    for tw in (g.teacher, g.student):               # O(0.3) states so that the modes differ visibly
        for k in tw.names:
            if k.endswith("basic_lstm_cell/kernel"):
                tw.store.p(k).mul_(2.0)
        tw.refresh_shadows()
    out = g.step(*_dev(x, n, labels), apply=False, num_frames_host=n)
    teacher, student = smoke.tower_params_numpy(g.teacher), smoke.tower_params_numpy(g.student)
    ref = mm.teacher_student_step(x.astype(np.float64), n, labels, teacher, student, 10, num_layers=1, with_grads=True)
    e_s = np.abs(out["teacher_state"].cpu().numpy() - ref["teacher_state"]).max()
    e_p = np.abs(out["predictions"].cpu().numpy() - ref["teacher_predictions"]).max()
    e_ps = np.abs(out["student_predictions"].cpu().numpy() - ref["student_predictions"]).max()
    print("lstm_layers=1 %s: teacher state err %.2e, pred err %.2e, student pred err %.2e" % (precision, e_s, e_p, e_ps))
    tol_s, tol_p = {"bf16": (2e-2, 2e-3), "high": (2e-3, 2e-4), "split": (1e-4, 2e-5)}[precision]
    assert e_s < tol_s and e_p < tol_p and e_ps < tol_p
    for tower, key in ((g.teacher, "teacher_grads"), (g.student, "student_grads")):
        got = smoke.tower_grads_numpy(tower)
        for k in got:
            gref = ref[key][k]
            if k in ("classifier/gates/weights", "classifier/experts/weights"):
                gref = gref - 2.0 * 1e-8 * smoke.tower_params_numpy(tower)[k]
            assert _rel(got[k], gref) < 4e-2, (tower.scope, k, _rel(got[k], gref))
    g.apply_gradients(B)
    assert g.global_step == 2


@pytest.mark.parametrize("L,order", [(3, "sequential"), (4, "sequential"), (4, "interleaved"), (6, "interleaved")])
def test_deep_stacks_run_every_backward_phase(L, order, monkeypatch):
    """--lstm_layers >= 3 (the backward issue loop resumed each tower's phase generator a FIXED number of times - enough for two
    layers per level - and left the lowest layers' BPTT, weight gradients, Adam and the final stream joins undone without an error from
    four layers on, six in the interleaved order): one training step in the overlapped schedule against the float64 oracle's gradients for
    EVERY variable of both towers, and every variable must have moved after the update."""
    from efficientvideoclassification_youtube8m_amd import smoke
    from efficientvideoclassification_youtube8m_amd.distill import DistillGraph
    monkeypatch.setenv("EVC_ISSUE_ORDER", order)
    B, F, H, V = 8, 64, 64, 24
    q, x, n, labels = mm.synthetic_batch(B, seed=14, feature_size=F, vocab_size=V, dtype=np.float32)
    g = DistillGraph(B, every_n=10, feature_size=F, vocab_size=V, lstm_cells=H, lstm_layers=L, device=DEV, seed=4)
    assert g.teacher.L == L and g.teacher.K == 2 * L * H and g.issue_order == order
    for tw in (g.teacher, g.student):
        for k in tw.names:
            if k.endswith("basic_lstm_cell/kernel"):
                tw.store.p(k).mul_(1.5)
        tw.refresh_shadows()
    teacher, student = smoke.tower_params_numpy(g.teacher), smoke.tower_params_numpy(g.student)
    before = {(tw.scope, k): tw.store.p(k).clone() for tw in (g.teacher, g.student) for k in tw.names}
    out = g.step(*_dev(x, n, labels), num_frames_host=n)            # apply=True: the updates run inside the step, on the aux streams
    torch.cuda.synchronize()
    ref = mm.teacher_student_step(x.astype(np.float64), n, labels, teacher, student, 10, num_layers=L, with_grads=True)
    assert np.abs(out["predictions"].cpu().numpy() - ref["teacher_predictions"]).max() < 4e-3
    for tower, key in ((g.teacher, "teacher_grads"), (g.student, "student_grads")):
        got = smoke.tower_grads_numpy(tower)
        assert len([k for k in got if k.endswith("kernel")]) == 2 * L
        for k in got:
            if k in ("classifier/gates/weights", "classifier/experts/weights"):
                continue                                              # (fused update: their gradient is never materialised)
            assert _rel(got[k], ref[key][k]) < 6e-2, (tower.scope, k, _rel(got[k], ref[key][k]))
        for k in tower.names:
            assert not torch.equal(tower.store.p(k), before[(tower.scope, k)]), (tower.scope, k, "was not updated")
    assert g.global_step == 2 and all(np.isfinite(v) for v in g.loss_report().values())


@pytest.mark.parametrize("B,H", [(8, 128), (40, 256)])
def test_l2_level_bptt_wavefront_pair_launches_equal_the_layer_by_layer_chain(B, H):
    """Round 5: the M ~ batch two-layer stacks (the L2 levels) run BPTT in wavefront order - layer 0's step t+1, with the gradient from
    layer 1 contracted in its own K walk, and layer 1's step t in ONE launch of the skinny kernel (evc_lstm_stack2_bwd at M <= 512): T + 1
    dependent launches instead of 2 T + a hoisted dX product.  Same batch, same weights, both forms: every gradient of both towers against
    the float64 oracle and against each other (the layer-by-layer form rounds layer 1's dX to bf16 before layer 0 adds it, the pair form keeps
    it in the f32 accumulator: they differ by that rounding, not more), and the L1 level's gradients - which take the L2 level's dX - too."""
    from efficientvideoclassification_youtube8m_amd import smoke
    from efficientvideoclassification_youtube8m_amd.distill import DistillGraph
    F, V = 128, 40
    q, x, n, labels = mm.synthetic_batch(B, seed=21, feature_size=F, vocab_size=V, dtype=np.float32)
    n[0], n[1] = 300, 7                                               # a full video and one that ends inside the first chunk
    x[np.arange(300)[None, :] >= n[:, None]] = 0.0
    g = DistillGraph(B, every_n=10, feature_size=F, vocab_size=V, lstm_cells=H, device=DEV, seed=4)
    for tw in (g.teacher, g.student):
        for k in tw.names:
            if k.endswith("basic_lstm_cell/kernel"):
                tw.store.p(k).mul_(1.5)
        tw.refresh_shadows()
    teacher, student = smoke.tower_params_numpy(g.teacher), smoke.tower_params_numpy(g.student)
    ref = mm.teacher_student_step(x.astype(np.float64), n, labels, teacher, student, 10, with_grads=True)
    got = {}
    for pair in (True, False):
        for tw in (g.teacher, g.student):
            tw.l2.small_pair = pair                                   # (off by default: faster alone, slower inside the training step - engine.py)
        g.step(*_dev(x, n, labels), apply=False, num_frames_host=n)
        torch.cuda.synchronize()
        got[pair] = {tw.scope: smoke.tower_grads_numpy(tw) for tw in (g.teacher, g.student)}
    for tower, key in ((g.teacher, "teacher_grads"), (g.student, "student_grads")):
        for k in got[True][tower.scope]:
            gref = ref[key][k]
            if k in ("classifier/gates/weights", "classifier/experts/weights"):
                gref = gref - 2.0 * 1e-8 * smoke.tower_params_numpy(tower)[k]
            a, b = got[True][tower.scope][k], got[False][tower.scope][k]
            assert _rel2(a, gref) < 3e-2 and _rel2(b, gref) < 3e-2, (tower.scope, k, _rel2(a, gref), _rel2(b, gref))
            assert _rel2(a, b) < 1e-2, (tower.scope, k, _rel2(a, b))
            if "RNN_L2" in k:
                assert _rel2(a, gref) < 1.5 * _rel2(b, gref) + 2e-3, (tower.scope, k, _rel2(a, gref), _rel2(b, gref))   # never worse than the form it replaces


# ---------------------------------------------------------------------------------------------------------------
# cfg 2: teacher only
# ---------------------------------------------------------------------------------------------------------------
def test_cfg2_teacher_only_real_dims():
    from efficientvideoclassification_youtube8m_amd import smoke
    from efficientvideoclassification_youtube8m_amd.distill import DistillGraph
    B = 4
    q, x, n, labels = mm.synthetic_batch(B, seed=12, dtype=np.float32)
    n[0], n[1] = 300, 121
    x[np.arange(300)[None, :] >= n[:, None]] = 0.0
    g = DistillGraph(B, mode="teacher", device=DEV, seed=3)
    assert g.student is None
    teacher = smoke.tower_params_numpy(g.teacher)
    out = g.step(*_dev(x, n, labels), apply=False, num_frames_host=n)
    assert "student_predictions" not in out
    xn = mm.l2_normalize(x.astype(np.float64), 2)
    t_state, t_pred, cache = mm.hlstm_fwd(xn, n, teacher, 20)
    e_p = np.abs(out["predictions"].cpu().numpy() - t_pred).max()
    e_s = np.abs(out["teacher_state"].cpu().numpy() - t_state).max()
    print("cfg2: teacher pred err %.2e state err %.2e" % (e_p, e_s))
    assert e_p < 1e-3 and e_s < 1e-3
    y = labels.astype(np.float64)
    assert abs(g.loss_report()["label_loss"] - mm.cross_entropy_loss(t_pred, y)) < 1e-4 * mm.cross_entropy_loss(t_pred, y)
    grads = mm.hlstm_bwd(None, mm.cross_entropy_grad(t_pred, y), cache)
    got = smoke.tower_grads_numpy(g.teacher)
    l2s = {}
    for k in mm.HLSTM_PARAM_ORDER:
        assert _rel(got[k], grads[k]) < 3e-2, (k, _rel(got[k], grads[k]))
        l2s[k] = _rel2(got[k], grads[k])
        assert l2s[k] < 1.2e-2, (k, l2s[k])
    print("cfg2 gradient relative L2:", {k.split("/")[0] + "/" + k.split("/")[-1] + k.split("/")[-3][-2:]: round(v, 4) for k, v in l2s.items()})
    g.apply_gradients(B)
    assert g.global_step == 1                       # one train op


# ---------------------------------------------------------------------------------------------------------------
# every_n = 1 (the reference's default): the student runs on all 300 frames in 5 chunks of 60 and its frame count
# goes through float64 (n/300)*300, which truncates to n-1 for n = 55, 79, 97, ...
# ---------------------------------------------------------------------------------------------------------------
def test_every_n_1_builds_teacher_and_student_with_the_float64_count_quirk():
    from efficientvideoclassification_youtube8m_amd import smoke
    from efficientvideoclassification_youtube8m_amd.distill import DistillGraph
    B, F, H, V = 6, 64, 64, 30
    q, x, n, labels = mm.synthetic_batch(B, seed=3, feature_size=F, vocab_size=V, dtype=np.float32)
    n[:] = (55, 79, 300, 97, 56, 1)
    x[np.arange(300)[None, :] >= n[:, None]] = 0.0
    ref_n = mm.student_num_frames(n, 1)
    assert list(ref_n) == [54, 78, 300, 96, 56, 1]                                          # the quirk is in the oracle
    g = DistillGraph(B, every_n=1, feature_size=F, vocab_size=V, lstm_cells=H, device=DEV, seed=2)
    assert g.student is not None and g.student.T == 300 and g.student.C == 5
    teacher, student = smoke.tower_params_numpy(g.teacher), smoke.tower_params_numpy(g.student)
    out = g.step(*_dev(x, n, labels), num_frames_host=n)
    assert np.array_equal(out["num_frames_student"].cpu().numpy(), ref_n)
    ref = mm.teacher_student_step(x.astype(np.float64), n, labels, teacher, student, 1, with_grads=False)
    assert np.abs(out["student_predictions"].cpu().numpy() - ref["student_predictions"]).max() < 1e-3
    assert np.abs(out["predictions"].cpu().numpy() - ref["teacher_predictions"]).max() < 1e-3
    rep = g.loss_report()
    for k in ("label_loss", "student_loss_state", "pred_loss", "student_label_loss"):
        assert abs(rep[k] - ref[k]) <= 2e-2 * abs(ref[k]) + 1e-6, (k, rep[k], float(ref[k]))
    assert g.global_step == 2


# ---------------------------------------------------------------------------------------------------------------
# cfg 1 and cfg 4 through the product's train.main on TFRecord files
# ---------------------------------------------------------------------------------------------------------------
def _write_32_videos_in_5_files(directory, sizes=(1024, 128)):
    """5 files holding 6+6+6+6+8 = 32 videos: ONE batch of 32 is the whole data set, so the oracle can be run on it
    without knowing the shuffle order."""
    from efficientvideoclassification_youtube8m_amd import readers
    files = readers.write_synthetic_frame_dataset(str(directory), 4, 6, feature_sizes=sizes, seed=11)
    files += readers.write_synthetic_frame_dataset(str(directory), 1, 8, feature_sizes=sizes, seed=12, first_file_index=4)
    return files


def _records(files, sizes=(1024, 128)):
    from efficientvideoclassification_youtube8m_amd import readers
    rd = readers.YT8MFrameFeatureReader(feature_names=["rgb", "audio"], feature_sizes=list(sizes), max_frames=300)
    ids, q, y, n = [], [], [], []
    for i, mat, lab, nf in rd.prepare_reader(files):
        ids.append(i[0]); q.append(mat[0]); y.append(lab[0]); n.append(nf[0])
    q, y, n = np.stack(q), np.stack(y), np.asarray(n)
    x = mm.dequantize(q.astype(np.float64)) * (np.arange(300)[None, :, None] < n[:, None, None])
    return ids, x, y.astype(np.float64), n


COMMON = ["--frame_features", "True", "--feature_names", "rgb, audio", "--feature_sizes", "1024, 128", "--gpu", "0",
          "--num_readers", "2", "--num_epochs", "1", "--start_new_model", "True", "--batch_size", "32"]


def test_cfg1_logistic_train_main_on_5_tfrecords(tmp_path):
    from efficientvideoclassification_youtube8m_amd import train
    from efficientvideoclassification_youtube8m_amd.flags import FLAGS
    files = _write_32_videos_in_5_files(tmp_path / "data")
    assert len(files) == 5
    FLAGS.reset()
    # learning rate 0: the checkpoint written after the step still holds the weights the logged loss was computed with
    res = train.main(COMMON + ["--model", "FrameLevelLogisticModel", "--train_data_pattern", str(tmp_path / "data" / "train*.tfrecord"),
                               "--train_dir", str(tmp_path / "m") + "/", "--base_learning_rate", "0.0"])
    FLAGS.reset()
    assert res["iterations"] == 1 and len(res["history"]) == 1
    step, loss, metrics = res["history"][0]
    assert step == 1
    sd = torch.load(train.latest_checkpoint(str(tmp_path / "m") + "/"))
    W, b = sd["model/fully_connected/weights"].double().numpy(), sd["model/fully_connected/biases"].double().numpy()
    assert W.shape == (1152, 4716)
    ids, x, y, n = _records(files)
    assert len(set(ids)) == 32
    p_ref, _ = mm.logistic_fwd(mm.l2_normalize(x, 2), n, W, b)
    want = mm.cross_entropy_loss(p_ref, y)
    print("cfg1 loss: train.main %.4f, oracle %.4f" % (loss["loss"], want))
    assert abs(loss["loss"] - want) < 1e-4 * want
    # the step's predictions against the oracle's, rows matched through the video ids of the product's batch
    perm = [ids.index(i) for i in res["graph"].last_batch_ids]
    pred = res["graph"].tower.pred.double().cpu().numpy()
    assert np.abs(pred - p_ref[perm]).max() < 1e-3


def test_cfg4_dbof_train_main_on_tfrecords(tmp_path):
    """DbofModel at the BASELINE cfg-4 model size (cluster 8192, hidden 1024, MoE 2), batch 32, one step through
    train.main; the oracle gets the same records, the checkpoint's weights and the step's own random-frame draw."""
    from efficientvideoclassification_youtube8m_amd import train
    from efficientvideoclassification_youtube8m_amd.flags import FLAGS
    files = _write_32_videos_in_5_files(tmp_path / "data")
    FLAGS.reset()
    res = train.main(COMMON + ["--model", "DbofModel", "--train_data_pattern", str(tmp_path / "data" / "train*.tfrecord"),
                               "--train_dir", str(tmp_path / "m") + "/", "--base_learning_rate", "0.0", "--dbof_cluster_size", "8192",
                               "--dbof_hidden_size", "1024", "--iterations", "30", "--moe_num_mixtures", "2"])
    FLAGS.reset()
    assert res["iterations"] == 1 and res["history"][0][0] == 1
    g = res["graph"]
    sd = torch.load(train.latest_checkpoint(str(tmp_path / "m") + "/"))
    P = {k[len("model/"):]: v.double().numpy() for k, v in sd.items() if k.startswith("model/") and torch.is_tensor(v)}
    assert P["cluster_weights"].shape == (1152, 8192) and P["hidden1_weights"].shape == (8192, 1024)
    ids, x, y, n = _records(files)
    # the product's batch order: match the oracle's rows to it through the frame counts the step saw + the video ids
    order = g.last_batch_ids
    perm = [ids.index(i) for i in order]
    x, y, n = x[perm], y[perm], n[perm]
    u = g.last_uniform.double().cpu().numpy()
    p_ref, _ = mm.dbof_fwd(mm.l2_normalize(x, 2), n, u, P)
    want = mm.cross_entropy_loss(p_ref, y)
    pred = g.tower.pred.double().cpu().numpy()
    err = np.abs(pred - p_ref).max()
    print("cfg4 through train.main: loss %.4f vs oracle %.4f, pred err %.2e" % (res["history"][0][1]["loss"], want, err))
    assert abs(res["history"][0][1]["loss"] - want) < 2e-3 * want
    assert err < 5e-3                               # bf16 operands on O(1) batch-normalised activations
