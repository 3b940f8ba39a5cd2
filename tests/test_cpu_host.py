"""CPU-side checks (no GPU): flag surface, host logic, C-ABI library exports,
loud failure without a device, data-parallel loss scaling over gloo."""
import ctypes
import os
import re
import shlex
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = "efficientvideoclassification_youtube8m_amd"


def test_flags_reference_launcher_line():
    """run_train.sh:6 verbatim flag string parses to the values the README flag dump shows (README.md:46-86)."""
    from efficientvideoclassification_youtube8m_amd.flags import FlagValues, GetListOfFeatureNamesAndSizes
    F = FlagValues()
    line = ('--train_data_pattern "./yt8m/train*.tfrecord" --train_dir  ./model_HLSTM_TeaStud_every10_train/   '
            '--frame_features True     --feature_names  "rgb, audio" --feature_sizes "1024, 128"   '
            '--model "HierarchicalLstmModel" --gpu 0    --batch_size 256  --num_inputs_to_lstm 20 --lstm_layers 2 '
            '--start_new_model True  --num_epochs 1  --every_n 10')
    unknown = F.parse(shlex.split(line))
    assert unknown == []
    assert (F.batch_size, F.every_n, F.lstm_layers, F.num_inputs_to_lstm, F.num_epochs) == (256, 10, 2, 20, 1)
    assert F.frame_features is True and F.start_new_model is True and F.model == "HierarchicalLstmModel"
    names, sizes = GetListOfFeatureNamesAndSizes(F.feature_names, F.feature_sizes)
    assert names == ["rgb", "audio"] and sizes == [1024, 128] and sum(sizes) == 1152
    # defaults
    assert (F.regularization_penalty, F.base_learning_rate, F.clip_gradient_norm, F.lstm_cells) == (2.0, 0.001, 1.0, 1024)
    assert (F.iterations, F.dbof_cluster_size, F.dbof_hidden_size, F.moe_num_mixtures, F.max_num_frames) == (30, 8192, 1024, 2, 300)
    assert F.video_level_classifier_model == "MoeModel" and F.label_loss == "CrossEntropyLoss" and F.a_rate == "2"


def test_flags_syntax_variants_and_unknown_flags():
    from efficientvideoclassification_youtube8m_amd.flags import FlagValues, GetListOfFeatureNamesAndSizes
    F = FlagValues()
    # run_validate.sh:4 passes flags the binary never defines; they are tolerated
    unknown = F.parse(["--batch_size=128", "--run_once", "--not_a_flag", "7", "--start_new_model", "False",
                       "--nobagging", "--top_k", "20", "--also_unknown=3"])
    assert F.batch_size == 128 and F.run_once is True and F.start_new_model is False and F.bagging is False
    assert unknown == ["--not_a_flag", "7", "--also_unknown=3"]
    with pytest.raises(ValueError):
        F.parse(["--batch_size"])
    with pytest.raises(ValueError):
        GetListOfFeatureNamesAndSizes("rgb, audio", "1024")
    with pytest.raises(AttributeError):
        F.no_such_flag


def test_every_n_host_logic_matches_oracle():
    from efficientvideoclassification_youtube8m_amd import distill
    from oracle import model_math as mm
    for e in range(1, 301):
        assert distill.every_n_indices(e) == mm.every_n_indices(e)
        ok_p = ok_o = True
        try:
            distill.validate_every_n(e)
        except ValueError:
            ok_p = False
        try:
            mm.validate_every_n(e)
        except ValueError:
            ok_o = False
        assert ok_p == ok_o, e
    assert distill.exponential_decay(1e-3, 20000, 256, 4000000, 0.5) == pytest.approx(mm.exponential_decay(1e-3, 20000, 256, 4000000, 0.5))


def _header_functions():
    src = open(os.path.join(ROOT, "include", "evc.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(evc_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    """The in-tree libevc_hip.so loads on a CPU-only box and exports exactly the
    entry points include/evc.h declares; the ctypes table binds all of them."""
    from efficientvideoclassification_youtube8m_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        subprocess.run(["bash", os.path.join(ROOT, PKG, "csrc", "build.sh")], check=True)
    lib = ctypes.CDLL(_lib.LIB_PATH)
    declared = _header_functions()
    assert len(declared) >= 30
    for name in declared:
        assert hasattr(lib, name), "declared in evc.h but not exported: " + name
    assert sorted(_lib.EXPORTS) == declared, (set(_lib.EXPORTS) ^ set(declared))
    l = _lib.load()
    assert l.evc_version() == 107
    assert l.evc_last_error() is not None


def test_ops_fail_loudly_without_gpu_tensors():
    from efficientvideoclassification_youtube8m_amd import _lib, ops
    a = torch.zeros((64, 64), dtype=torch.bfloat16)
    out = torch.zeros((64, 64))
    with pytest.raises(_lib.EvcError):
        ops.gemm_nt(a, a, 64, 64, 64, out)          # no CPU fallback
    if not torch.cuda.is_available():
        with pytest.raises(_lib.EvcError):
            ops.check_device(0)


def test_model_classes_resolve_by_name():
    from efficientvideoclassification_youtube8m_amd import frame_level_models, losses, video_level_models
    from efficientvideoclassification_youtube8m_amd.train import find_class_by_name
    for n in ("HierarchicalLstmModel", "DbofModel", "FrameLevelLogisticModel", "NetVLADModel", "NeXtVLADModel", "MoeModel",
              "LogisticModel"):
        cls = find_class_by_name(n, [frame_level_models, video_level_models])
        assert cls.__name__ == n and hasattr(cls(), "create_model")
    assert find_class_by_name("CrossEntropyLoss", [losses]).__name__ == "CrossEntropyLoss"
    with pytest.raises(StopIteration):
        find_class_by_name("NoSuchModel", [frame_level_models, video_level_models])
    assert frame_level_models.NeXtVLADModel().create_model(None, 1, None) is None     # empty stub, as in the reference
    assert frame_level_models.NetVLADModel().create_model_inference(None, 1, 10, None) is None   # (create_model: extension, GPU tests)


DP_WORKER = r'''
import os, sys
sys.path.insert(0, %(root)r)
import numpy as np, torch, torch.distributed as dist
from oracle import model_math as mm
from efficientvideoclassification_youtube8m_amd.distill import dp_loss_scales, GradReducer

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo")
rng = np.random.default_rng(0)
B, F, H, V, every_n = 4, 6, 4, 5, 30
teacher = mm.init_hlstm_params(rng, F, H, 2, V); student = mm.init_hlstm_params(rng, F, H, 2, V)
x = rng.standard_normal((B, 300, F)); n = np.array([300, 200, 150, 31]); y = rng.random((B, V)) > 0.5
x[np.arange(300)[None] >= n[:, None]] = 0
full = mm.teacher_student_step(x, n, y, teacher, student, every_n)
# this rank's shard, with the product's loss scaling rules, through the product's reducer
sl = slice(rank * B // world, (rank + 1) * B // world)
sc = dp_loss_scales(world)
loc = mm.teacher_student_step(x[sl], n[sl], y[sl], teacher, student, every_n, regularization_penalty=0.0)
# oracle grads are for means over the LOCAL batch; recombine per the scaling rules
xs, ns, ys = x[sl], n[sl], y[sl]
t_state, t_pred, t_cache = mm.hlstm_fwd(mm.l2_normalize(xs, 2), ns, teacher, 20)
s_in = mm.subsample_frames(mm.l2_normalize(xs, 2), every_n)
s_state, s_pred, s_cache = mm.hlstm_fwd(s_in, mm.student_num_frames(ns, every_n), student, 5)
tg = mm.hlstm_bwd(None, sc["ce"] * mm.cross_entropy_grad(t_pred, ys), t_cache)
ds = sc["rep"] * 2.0 * mm.rep_loss_grad_student(t_state, s_state)
dp = sc["kl"] * mm.pred_kl_grad_student(t_pred, s_pred) + sc["ce"] * mm.cross_entropy_grad(s_pred, ys)
sg = mm.hlstm_bwd(ds, dp, s_cache)
for grads, ref, params in ((tg, full["teacher_grads"], teacher), (sg, full["student_grads"], student)):
    flat = torch.cat([torch.from_numpy(grads[k]).reshape(-1) for k in mm.HLSTM_PARAM_ORDER])
    red = GradReducer(None)
    cut = flat.numel() // 3
    red.reduce(flat, cut, flat.numel())      # "MoE segment first"
    red.reduce(flat, 0, cut)
    red.wait()
    off = 0
    for k in mm.HLSTM_PARAM_ORDER:
        g = flat[off:off + grads[k].size].numpy().reshape(grads[k].shape); off += grads[k].size
        want = ref[k]
        if k in ("classifier/gates/weights", "classifier/experts/weights"):
            want = want - 2.0 * 1e-8 * params[k]          # the l2 term is added after the reduce, once
        assert np.allclose(g, want, rtol=1e-9, atol=1e-12), (rank, k)
dist.destroy_process_group()
sys.stdout.write("rank" + str(rank) + "-ok\n"); sys.stdout.flush()
'''


def test_data_parallel_gradient_equals_global_batch_gloo(tmp_path):
    """world_size 2 over gloo: per-rank gradients scaled by the product's rules
    (CE and L_REP are batch means -> 1/world; L_PRED is a batch sum -> 1) and summed by
    the product's bucketed reducer equal the single-process gradient at the global batch."""
    script = tmp_path / "dp_worker.py"
    script.write_text(DP_WORKER % {"root": ROOT})
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="2")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                        "--master-addr", "127.0.0.1", "--master-port", "29533", str(script)],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    assert r.stdout.count("-ok") == 2, r.stdout


SHARD_WORKER = r'''
import os, sys
sys.path.insert(0, %(root)r)
import numpy as np, torch, torch.distributed as dist
from efficientvideoclassification_youtube8m_amd.distill import GradReducer

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo")
red = GradReducer(None)
assert red.active and red.world == 2 and red.rank == rank
# factor all-gather: rows of every rank stacked in rank order (bf16 moved as raw bytes)
t = (torch.arange(6, dtype=torch.float32).reshape(3, 2) + 100 * rank).to(torch.bfloat16)
g = red.all_gather_rows(t)
want = torch.cat([(torch.arange(6, dtype=torch.float32).reshape(3, 2) + 100 * r).to(torch.bfloat16) for r in range(world)])
assert g.dtype == torch.bfloat16 and torch.equal(g, want)
# the partial norm sums of a row-sharded tensor
s = torch.tensor([1.0 + rank, 10.0 * (rank + 1)])
red.all_reduce_small(s)
assert s.tolist() == [3.0, 30.0]
# row slabs: V = 600 rows in 128-row tiles over 2 ranks -> 384 + 216 (MoeHead.shard's rule), padded image of 768 rows
V, K = 600, 8
tiles = (V + 127) // 128
slab = (tiles + world - 1) // world * 128
assert slab == 384
full = torch.full((slab * world, K), -1.0)
v0 = rank * slab
vs = min(V, v0 + slab) - v0
assert vs == (384 if rank == 0 else 216)
full[v0:v0 + vs] = torch.arange(v0, v0 + vs, dtype=torch.float32)[:, None].expand(vs, K)     # "this rank's updated rows"
red.all_gather_slabs(full, slab)
assert torch.equal(full[:V, 0], torch.arange(V, dtype=torch.float32))                      # every rank now holds every row
# stream-ordered gradient all-reduce of a segment (synchronous c10d op; returns nothing to wait for)
flat = torch.arange(10, dtype=torch.float32) * (rank + 1)
assert red.reduce_async(flat, 2, 7) is None
assert flat.tolist() == [0, 1 * (rank + 1)] + [3.0 * i for i in range(2, 7)] + [i * (rank + 1.0) for i in range(7, 10)]
# bf16 gradient payload (EVC_DP_GRAD_DTYPE=bf16 / grad_dtype="bf16"): every rank's segment is rounded to bf16 once, the sum
# arrives back in f32 - within 2^-9 of the f32 reduce per element (gloo: exact sum of the rounded values)
redb = GradReducer(None, grad_dtype="bf16")
g = torch.linspace(-3.0, 3.0, 1000) * (1.0 + 0.37 * rank) + 1e-3 * rank
want32 = sum(torch.linspace(-3.0, 3.0, 1000) * (1.0 + 0.37 * r) + 1e-3 * r for r in range(world))
wantb = sum((torch.linspace(-3.0, 3.0, 1000) * (1.0 + 0.37 * r) + 1e-3 * r).bfloat16().float() for r in range(world))
flatb = torch.cat([torch.full((5,), 7.0), g.clone(), torch.full((3,), -7.0)])
assert redb.reduce_async(flatb, 5, 1005) is None
assert torch.equal(flatb[:5], torch.full((5,), 7.0)) and torch.equal(flatb[1005:], torch.full((3,), -7.0))      # outside the segment: untouched
assert torch.allclose(flatb[5:1005], wantb, rtol=0, atol=1e-6)
assert float((flatb[5:1005] - want32).abs().max()) <= 2.0 ** -8 * float(want32.abs().max())
assert GradReducer.stats["all_reduce_grad_bf16"] == [1, 2000] and GradReducer.stats["all_reduce_grad_f32"][1] == 5 * 4
assert GradReducer.wire_bytes("all_reduce_grad_f32", 800.0, 8) == 1400.0 and GradReducer.wire_bytes("all_gather_slabs", 100.0, 8) == 700.0
# iteration-count agreement (train.agree_step_limit): MIN over the ranks; a rank without one whole batch makes it 0 on
# EVERY rank (0 is a limit, not "no limit": the loop is skipped everywhere, nobody is left alone in an all-reduce)
from efficientvideoclassification_youtube8m_amd.train import agree_step_limit
assert agree_step_limit(0, 5 + rank, world) == 5
assert agree_step_limit(3, 5 + rank, world) == 3
assert agree_step_limit(9, 5 + rank, world) == 5
lim = agree_step_limit(0, 4 if rank == 0 else 0, world)
assert lim == 0 and lim is not None
assert agree_step_limit(7, None, world) == 7 and agree_step_limit(0, None, world) is None
dist.destroy_process_group()
sys.stdout.write("rank" + str(rank) + "-ok\n"); sys.stdout.flush()
'''


def test_sharded_update_collectives_gloo(tmp_path):
    """world_size 2 over gloo: the collectives of the row-sharded MoE update (factor all-gather, 8-byte norm
    all-reduce, in-place slab all-gather) and the stream-ordered gradient all-reduce, on CPU tensors."""
    script = tmp_path / "shard_worker.py"
    script.write_text(SHARD_WORKER % {"root": ROOT})
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="2")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                        "--master-addr", "127.0.0.1", "--master-port", "29535", str(script)],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    assert r.stdout.count("-ok") == 2, r.stdout


def test_grad_ranges_skip_synced_batchnorm_gradients():
    """SingleTowerGraph all-reduces the flat gradient buffer in the contiguous ranges that leave out the batch-norm
    gamma/beta gradients (already global after BatchNorm.backward's f64 partial-sum all-reduce: reducing them again
    would multiply them by the world size)."""
    import math
    from collections import OrderedDict
    from efficientvideoclassification_youtube8m_amd.engine import TowerBase, _align
    from efficientvideoclassification_youtube8m_amd.towers import DbofTower

    class Store:
        pass

    tw = TowerBase.__new__(TowerBase)
    shapes = OrderedDict([("input_bn/beta", (10,)), ("input_bn/gamma", (10,)), ("cluster_weights", (7, 10)),
                          ("cluster_bn/beta", (7,)), ("cluster_bn/gamma", (7,)), ("hidden1_weights", (5, 7)),
                          ("hidden1_bn/beta", (5,)), ("hidden1_bn/gamma", (5,)), ("classifier/gates/weights", (9, 5)),
                          ("classifier/experts/weights", (6, 5)), ("classifier/experts/biases", (6,))])
    st = Store()
    st.shapes, st.offsets, off = shapes, {}, 0
    for k, shp in shapes.items():
        st.offsets[k] = off
        off += _align(int(math.prod(shp)))
    tw.store, tw.names = st, list(shapes)
    assert tw.grad_ranges() == [(0, off)]
    rs = tw.grad_ranges(exclude=DbofTower.global_grad_names)
    covered = set()
    for lo, hi in rs:
        covered |= set(range(lo, hi))
    for k, shp in shapes.items():
        inside = set(range(st.offsets[k], st.offsets[k] + int(math.prod(shp)))) <= covered
        assert inside == (k not in DbofTower.global_grad_names), k
    assert len(rs) == 3          # cluster_weights | hidden1_weights | the MoE block


TIMEOUT_WORKER = r'''
import os, sys, time
sys.path.insert(0, %(root)r)
import torch, torch.distributed as dist
from efficientvideoclassification_youtube8m_amd.distill import dp_timeout
rank = int(os.environ["RANK"])
assert dp_timeout().total_seconds() == 4.0
dist.init_process_group("gloo", timeout=dp_timeout())
t = torch.ones(4)
dist.all_reduce(t)                      # first contact works
assert t.tolist() == [2.0] * 4
if rank == 1:
    time.sleep(12)                      # a wedged peer: never joins the second collective in time
    os._exit(0)
t0 = time.time()
try:
    dist.all_reduce(t)
    sys.stdout.write("rank0-no-timeout\n")
except Exception as e:
    sys.stdout.write("rank0-timeout-after-%%.0fs %%s\n" %% (time.time() - t0, type(e).__name__))
sys.stdout.flush()
os._exit(0)
'''


def test_process_group_timeout_ends_a_wedged_collective(tmp_path):
    """bench.py / train.py create their process group with distill.dp_timeout() (EVC_DP_TIMEOUT_S, default 120 s instead of
    c10d's 10 minutes): a collective whose peer never arrives raises after that time instead of holding the launcher's whole
    window.  Two gloo ranks, the second one stalls."""
    script = tmp_path / "timeout_worker.py"
    script.write_text(TIMEOUT_WORKER % {"root": ROOT})
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="2", EVC_DP_TIMEOUT_S="4")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                        "--master-addr", "127.0.0.1", "--master-port", "29537", str(script)],
                       env=env, capture_output=True, text=True, timeout=300)
    assert "rank0-timeout-after" in r.stdout and "no-timeout" not in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]
    waited = float(r.stdout.split("rank0-timeout-after-")[1].split("s")[0])
    assert 2.0 <= waited <= 11.0, r.stdout


def test_bench_rank_supervisor_falls_back_to_the_serial_placement(tmp_path, monkeypatch, capfd):
    """bench.py under N > 1: each rank's supervisor (a process that never touches the GPU) runs the benchmark in a child and, when
    that child dies or overstays its wall limit, starts one more child with EVC_DP_SERIAL_COMM=1 on another rendezvous port -
    exercised here with stand-in children: one that fails fast, one that hangs, one that succeeds at once."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    child = tmp_path / "child.py"
    child.write_text("import os, sys, time\n"
                     "mode = sys.argv[1]\n"
                     "serial = os.environ.get('EVC_DP_SERIAL_COMM') == '1'\n"
                     "assert os.environ['EVC_BENCH_CHILD'] == '1'\n"
                     "print('child attempt', os.environ['EVC_BENCH_ATTEMPT'], 'serial', serial, 'port', os.environ['MASTER_PORT'],\n"
                     "      'agent_store', os.environ.get('TORCHELASTIC_USE_AGENT_STORE'), flush=True)\n"
                     "if mode == 'ok' or serial:\n"
                     "    sys.exit(0)\n"
                     "if mode == 'hang':\n"
                     "    time.sleep(60)\n"
                     "sys.exit(3)\n")
    monkeypatch.setenv("MASTER_PORT", "29600")
    monkeypatch.setenv("RANK", "0")
    monkeypatch.setenv("TORCHELASTIC_USE_AGENT_STORE", "True")
    monkeypatch.setenv("EVC_BENCH_ATTEMPT_S", "3")
    monkeypatch.delenv("EVC_DP_SERIAL_COMM", raising=False)
    assert bench.supervise_ranks(["ok"], script=str(child)) == 0
    out = capfd.readouterr()
    assert out.out.count("child attempt") == 1 and "attempt 0 serial False port 29600 agent_store True" in out.out
    for mode in ("fail", "hang"):
        assert bench.supervise_ranks([mode], script=str(child)) == 0
        out = capfd.readouterr()
        assert "attempt 0 serial False port 29600" in out.out and "attempt 1 serial True port 29617 agent_store None" in out.out, out.out
        assert "retrying with the next placement" in out.err
    monkeypatch.setenv("EVC_DP_SERIAL_COMM", "1")            # already on one communicator: first attempt as given
    assert bench.supervise_ranks(["ok"], script=str(child)) == 0
    assert capfd.readouterr().out.count("child attempt") == 1


def test_bench_rank_supervisors_agree_before_retrying(tmp_path):
    """Two rank supervisors (CPU only): rank 1's child fails on the first placement while rank 0's succeeds - rank 0 must NOT
    return with its child's 0 but retry together with rank 1 (the exit codes are exchanged through a TCPStore), and both end 0
    after the second placement; nobody starts an attempt alone."""
    child = tmp_path / "child.py"
    child.write_text("import os, sys\n"
                     "serial = os.environ.get('EVC_DP_SERIAL_COMM') == '1'\n"
                     "print('rank', os.environ['RANK'], 'attempt', os.environ['EVC_BENCH_ATTEMPT'], 'serial', serial, 'order', os.environ.get('EVC_ISSUE_ORDER'), flush=True)\n"
                     "sys.exit(0 if (serial or os.environ['RANK'] == '0') else 5)\n")
    driver = tmp_path / "driver.py"
    driver.write_text("import importlib.util, sys\n"
                      "spec = importlib.util.spec_from_file_location('bench_mod', %r)\n"
                      "bench = importlib.util.module_from_spec(spec)\n"
                      "spec.loader.exec_module(bench)\n"
                      "sys.exit(bench.supervise_ranks(['x'], script=%r))\n" % (os.path.join(ROOT, "bench.py"), str(child)))
    procs = []
    for rank in (0, 1):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT="29710", EVC_BENCH_ATTEMPT_S="20")
        env.pop("EVC_DP_SERIAL_COMM", None)
        procs.append(subprocess.Popen([sys.executable, str(driver)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=240) for p in procs]
    assert [p.returncode for p in procs] == [0, 0], outs
    for rank, (o, e) in enumerate(outs):
        assert "rank %d attempt 0 serial False" % rank in o and "rank %d attempt 1 serial True order None" % rank in o, (o, e)
        assert "attempt 2" not in o
    assert "ended with code 0 here, 5 over all ranks" in outs[0][1], outs[0][1]


def _load_bench():
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    return bench


def test_bench_figures_flag_a_stall_in_the_timed_window():
    """One 30 ms stall inside a 20-step window of 2 ms steps (round 4's driver read cfg 4 as 3.45 ms that way): the mean moves by
    1.7x, the median does not, and the line says so; a clean window is not flagged; a flagged secondary configuration is timed once
    more with the first attempt kept in the line."""
    bench = _load_bench()
    steps = [1.9] * 19 + [31.0]
    st = bench.step_stats(steps)
    assert st["stall_suspected"] is True and st["ms_per_step_max"] == 31.0 and abs(st["ms_per_step_median"] - 1.9) < 1e-9
    assert sum(steps) / len(steps) > 1.7 * st["ms_per_step_median"]
    clean = bench.step_stats([10.1, 10.3, 10.2, 10.6, 10.0])
    assert clean == {"ms_per_step_median": 10.2, "ms_per_step_max": 10.6, "stall_suspected": False}
    assert bench.step_stats([5.0, 5.0, 14.9])["stall_suspected"] is False and bench.step_stats([5.0, 5.0, 15.1])["stall_suspected"] is True
    assert set(bench.STAT_KEYS) == set(st)
    # time lost outside the per-step events (before the first / behind the last step): the wall-clock mean gives it away
    assert bench.step_stats([2.1] * 20, wall_ms_per_step=2.32)["stall_suspected"] is True
    assert bench.step_stats([2.1] * 20, wall_ms_per_step=2.15)["stall_suspected"] is False
    runs = []

    def fake_run():
        runs.append(1)
        first = len(runs) == 1
        return dict(ms_per_step=3.4 if first else 1.9, **bench.step_stats(steps if first else [1.9] * 20))

    r = bench.retime_on_stall(fake_run)
    assert len(runs) == 2 and r["ms_per_step"] == 1.9 and r["stall_suspected"] is False
    assert r["retimed_after_stall"] == {"ms_per_step": 3.4, "ms_per_step_median": 1.9, "ms_per_step_max": 31.0, "stall_suspected": True}
    runs.clear()
    r = bench.retime_on_stall(lambda: (runs.append(1), dict(ms_per_step=1.9, **bench.step_stats([1.9] * 20)))[1])
    assert len(runs) == 1 and "retimed_after_stall" not in r


def test_bench_gpus_flag_means_n(tmp_path):
    """`--gpus N` is the number of ranks in the line: under a launcher it must equal WORLD_SIZE (exit 2 with the reason otherwise);
    without one, N > 1 makes bench.py start N rank processes itself - fresh children with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*
    set (stand-in children here; the real path runs in tests/test_gpu_dp.py)."""
    env = dict(os.environ, WORLD_SIZE="4", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       text=True, timeout=300)
    assert r.returncode == 2 and "--gpus 2" in r.stderr and "WORLD_SIZE=4" in r.stderr and r.stdout == ""
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert r.returncode == 2 and "--gpus 1" in r.stderr
    bench = _load_bench()
    child = tmp_path / "rank.py"
    child.write_text("import os, sys\n"
                     "print('rank %s/%s local %s at %s:%s self %s args %s' % (os.environ['RANK'], os.environ['WORLD_SIZE'], os.environ['LOCAL_RANK'],\n"
                     "      os.environ['MASTER_ADDR'], os.environ['MASTER_PORT'], os.environ['EVC_BENCH_SELF_LAUNCHED'], ' '.join(sys.argv[1:])), flush=True)\n"
                     "sys.exit(7 if (sys.argv[-1] == 'fail' and os.environ['RANK'] == '2') else 0)\n")
    driver = tmp_path / "driver.py"
    driver.write_text("import importlib.util, os, sys\n"
                      "os.environ.pop('WORLD_SIZE', None); os.environ.pop('MASTER_PORT', None)\n"
                      "spec = importlib.util.spec_from_file_location('bench_mod', %r)\n"
                      "bench = importlib.util.module_from_spec(spec)\n"
                      "spec.loader.exec_module(bench)\n"
                      "sys.exit(bench.self_launch(3, ['--gpus', '3', sys.argv[1]], script=%r))\n" % (os.path.join(ROOT, "bench.py"), str(child)))
    for mode, rc in (("ok", 0), ("fail", 7)):
        r = subprocess.run([sys.executable, str(driver), mode], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
        assert r.returncode == rc, (r.stdout, r.stderr)
        lines = sorted(l for l in r.stdout.splitlines() if l.startswith("rank "))
        assert len(lines) == 3
        ports = {l.split(" at 127.0.0.1:")[1].split()[0] for l in lines}
        assert len(ports) == 1 and 1024 < int(ports.pop()) <= 65000
        for i, l in enumerate(lines):
            assert l.startswith("rank %d/3 local %d at 127.0.0.1:" % (i, i)) and l.endswith("self 1 args --gpus 3 %s" % mode), l


WORLD4_WORKER = r'''
import os, sys
sys.path.insert(0, %(root)r)
import torch, torch.distributed as dist
from efficientvideoclassification_youtube8m_amd.distill import GradReducer, dp_loss_scales
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo")
red = GradReducer(None)
assert red.active and red.world == 4
# MoeHead.shard's slab rule with a rank that owns NO rows: V = 300 rows in 128-row tiles over 4 ranks -> 3 tiles, slab = 128:
# ranks 0, 1 own 128 rows, rank 2 owns 44, rank 3 none (the update kernels are skipped there, the collectives are not)
V, K = 300, 8
tiles = (V + 127) // 128
slab = (tiles + world - 1) // world * 128
assert slab == 128
v0 = rank * slab
vs = min(V, v0 + slab) - v0
assert vs == [128, 128, 44, -84][rank]
full = torch.full((slab * world, K), -1.0)
if vs > 0:
    full[v0:v0 + vs] = torch.arange(v0, v0 + vs, dtype=torch.float32)[:, None].expand(vs, K)
red.all_gather_slabs(full, slab)
assert torch.equal(full[:V, 0], torch.arange(V, dtype=torch.float32))
sums = torch.tensor([float(max(vs, 0)), 1.0])               # partial norm sums: the empty rank contributes zeros / its share
red.all_reduce_small(sums)
assert sums.tolist() == [300.0, 4.0]
# factor all-gather: rank order along the rows
t = torch.full((2, 3), float(rank)).to(torch.bfloat16)
g = red.all_gather_rows(t)
assert g.shape == (8, 3) and g[:, 0].float().tolist() == [0, 0, 1, 1, 2, 2, 3, 3]
# loss scales: batch means 1/world, the L_PRED batch sum 1
sc = dp_loss_scales(world)
assert sc == {"ce": 0.25, "rep": 0.25, "kl": 1.0}
assert GradReducer.wire_bytes("all_reduce_grad_f32", 1000.0, world) == 1500.0 and GradReducer.wire_bytes("all_gather_factors", 10.0, world) == 30.0
dist.destroy_process_group()
sys.stdout.write("rank" + str(rank) + "-ok\n"); sys.stdout.flush()
'''


def test_sharded_update_collectives_world_4_with_an_empty_rank(tmp_path):
    """world_size 4 over gloo: the row-slab rule leaves the last rank without rows when the tiles do not go round (V = 300:
    3 tiles of 128 rows over 4 ranks) - that rank skips the update kernels but takes part in every collective; slabs, norm sums and
    factor images still assemble correctly on all ranks."""
    script = tmp_path / "world4_worker.py"
    script.write_text(WORLD4_WORKER % {"root": ROOT})
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=4",
                        "--master-addr", "127.0.0.1", "--master-port", "29539", str(script)],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    assert r.stdout.count("-ok") == 4, r.stdout


def test_high_precision_dither_layer_selection_rules():
    """engine.HLstmTower.dither_layers / dither_wh0 / dither_col0 (host logic, no GPU): unset -> the TOP layer of an L1 level of two or more
    layers; an explicit list must be a suffix of the stack (a corrected layer reads the e4m3 image of h that a dithered layer below it does
    not write); nothing when the level is not the f16 + e4m3 one (small dims, the student's plain f16); layer 0's recurrent block only with
    every layer above dithered, and then its images dither the columns from the input width on."""
    import types
    from efficientvideoclassification_youtube8m_amd.engine import HLstmTower

    def tower(L, layers=None, fp8=True, wh0=False, nin=1152, H=1024):
        t = types.SimpleNamespace(L=L, H=H, f16_dither_layers=layers, f16_dither_wh0=wh0, fp8_lo=lambda: fp8)
        t.store = types.SimpleNamespace(shapes={"RNN_L1/rnn/multi_rnn_cell/cell_%d/basic_lstm_cell/kernel" % l: (4 * H, (nin if l == 0 else H) + H) for l in range(L)})
        for name in ("dither_layers", "dither_wh0", "dither_col0"):
            setattr(t, name, types.MethodType(getattr(HLstmTower, name), t))
        return t

    assert tower(2).dither_layers() == (1,) and tower(3).dither_layers() == (2,) and tower(1).dither_layers() == ()
    assert tower(2, fp8=False).dither_layers() == () and tower(2, layers=()).dither_layers() == ()
    assert tower(2, layers=(0, 1)).dither_layers() == (0, 1) and tower(3, layers=(1, 2)).dither_layers() == (1, 2)
    assert tower(2, layers=(1, 7)).dither_layers() == (1,)                       # layers the stack does not have are ignored
    for bad in ((0,), (0, 2)):
        with pytest.raises(ValueError):
            tower(3, layers=bad).dither_layers()
    k0, k1 = "RNN_L1/rnn/multi_rnn_cell/cell_0/basic_lstm_cell/kernel", "RNN_L1/rnn/multi_rnn_cell/cell_1/basic_lstm_cell/kernel"
    t = tower(2, wh0=True)
    assert t.dither_wh0() and t.dither_col0(k0) == 1152 and t.dither_col0(k1) == 0
    assert not tower(2, wh0=True, layers=()).dither_wh0() and not tower(1, wh0=True).dither_wh0() and not tower(2, wh0=True, fp8=False).dither_wh0()
    assert tower(2, layers=(0, 1), wh0=True).dither_col0(k0) == 0                # every column when layer 0 itself is dithered


def test_moe_gradient_exchange_is_chosen_by_shape():
    """MoeHead.dp_route (round 6): per rank and step the factor all-gather moves (W-1) Br (N_g + N_e + K) 2 bytes, the bf16 reduce-scatter of the
    locally formed gradient (W-1)/W (N_g + N_e) K 2 - BASELINE cfg 3 (B 256) stays on the factors, cfg 5 (B 1024: 398 vs 176 MB with the slab padding) and cfg 4
    (K 1024: 177 vs 44 MB) take the reduce-scatter; cfg 5's whole step then puts <= 0.68 GB per rank on the wire."""
    from efficientvideoclassification_youtube8m_amd.engine import MoeHead
    import efficientvideoclassification_youtube8m_amd.ops as ops

    def head(B, K, V=4716, Mx=2):
        h = MoeHead(None, K, V, Mx)
        h.B, h.Br = B, ops.round_up(B, 32)
        return h
    os.environ.pop("EVC_DP_MOE_ROUTE", None)
    c3, c5, c4 = head(256, 4096), head(1024, 4096), head(512, 1024)
    b5 = c5.dp_exchange_bytes(8)
    assert abs(b5["factors"] - 7 * 1024 * (14208 + 9472 + 4096) * 2) < 1 and abs(b5["factors"] / 1e6 - 398) < 1
    assert abs(b5["reduce_scatter"] - 7 / 8 * (14 + 10) * 128 * 8 * 4096 * 2) < 1 and abs(b5["reduce_scatter"] / 1e6 - 176.2) < 0.1   # (14 + 10 row tiles of 128 rows per rank: 169 MB unpadded)
    assert c3.dp_route(8) == "factors" and c5.dp_route(8) == "reduce_scatter" and c4.dp_route(8) == "reduce_scatter"
    assert head(448, 4096).dp_route(8) == "factors" and head(480, 4096).dp_route(8) == "reduce_scatter"       # break-even B ~ 453
    assert c5.dp_route(2) == "factors"                                  # two ranks: one peer's factors are fewer bytes than half a gradient
    lstm_allreduce = 2 * 7 / 8 * (17309696 + 29368320) * 4                # the student's L1 + L2 gradient segments, f32 ring all-reduce
    slab_gather = 7 / 8 * (14 * 128 * 8 + 10 * 128 * 8) * 4096 * 2        # the owners' new bf16 rows (either route)
    assert (lstm_allreduce + slab_gather + b5["reduce_scatter"]) / 1e9 <= 0.68 < (lstm_allreduce + slab_gather + b5["factors"]) / 1e9
    os.environ["EVC_DP_MOE_ROUTE"] = "factors"
    try:
        assert c5.dp_route(8) == "factors"
    finally:
        del os.environ["EVC_DP_MOE_ROUTE"]
