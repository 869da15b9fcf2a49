"""CPU tests of the host side (no GPU, no kernel launches): the C-ABI library loads and exports every
symbol include/acimg.h declares, parameter packing is lossless, plans record for every model variant,
error paths return codes + text, and the data-parallel gradient exchange works on 2 gloo ranks."""
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as ge

    ge.build()
    from acimg import _lib

    return _lib.load()


def test_library_exports_every_declared_symbol(lib):
    from acimg import _lib

    hdr = open(os.path.join(ROOT, "include", "acimg.h")).read()
    declared = set(re.findall(r"\b(acimg_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 35
    assert declared == set(_lib.PROTOTYPES), declared ^ set(_lib.PROTOTYPES)
    out = subprocess.check_output(["nm", "-D", "--defined-only", _lib.LIB_PATH]).decode()
    exported = set(re.findall(r" T (acimg_[a-z0-9_]+)", out))
    assert declared <= exported, declared - exported
    assert lib.acimg_version() == 207


def test_host_side_queries_need_no_gpu(lib):
    from acimg import ops

    d = ops.conv_desc(32, 56, 75, 128, 128, 3, 3)
    assert ops.conv2d_fwd_tiling(d) == (128, 128, 1)
    assert ops.conv2d_stats_rows(d) == 1050
    d = ops.conv_desc(32, 14, 19, 2048, 12, 3, 4, 1, "VALID")
    bm, bn, splits = ops.conv2d_fwd_tiling(d)
    assert (bm, bn) == (256, 16) and splits > 1
    assert ops.conv2d_stats_rows(d) == 24          # split-K path: one partial per 256 rows
    # split-K slabs: the larger of the row layout (reduce launch) and the tile-padded layout (in-kernel hand-off)
    rows, tiled = splits * 32 * 12 * 16 * 12 * 4, splits * 24 * 256 * 16 * 4
    assert lib.acimg_conv2d_fwd_workspace(__import__("ctypes").byref(d)) == max(rows, tiled)
    # few-channel 3x3 / stride-1 / SAME layers from 65536 pixels on: the MFMA form leaves one statistics row per workgroup;
    # with an activation, a stride, or below the size rule the direct kernel's 256-pixel blocks remain
    assert ops.conv2d_stats_rows(ops.conv_desc(32, 224, 298, 8, 8, 3, 3, 1, "SAME")) == 512
    assert ops.conv2d_stats_rows(ops.conv_desc(32, 224, 298, 16, 8, 3, 3, 1, "SAME")) == 512
    assert ops.conv2d_stats_rows(ops.conv_desc(32, 112, 149, 8, 32, 3, 3, 1, "SAME")) == 512
    assert ops.conv2d_stats_rows(ops.conv_desc(32, 224, 298, 4, 8, 3, 3, 1, "SAME")) == 512
    assert ops.conv2d_stats_rows(ops.conv_desc(32, 224, 298, 8, 8, 3, 3, 2, "SAME")) == 32 * 112 * 149 // 256
    assert ops.conv2d_stats_rows(ops.conv_desc(32, 224, 298, 8, 8, 3, 3, 1, "SAME", act=1)) == 32 * 224 * 298 // 256
    assert ops.conv2d_stats_rows(ops.conv_desc(2, 100, 100, 8, 8, 3, 3, 1, "SAME")) != 512
    # the host wrapper refuses a statistics tensor sized from another descriptor (the C side cannot see the buffer)
    import pytest as _pt
    import torch as _t
    dd = ops.conv_desc(32, 224, 298, 8, 8, 3, 3, 1, "SAME")
    with _pt.raises(ValueError):
        ops.conv2d_fwd(ops.Plan(_t.device("cpu")), dd, _t.zeros(1), _t.zeros(1), None, _t.zeros(1), stats=_t.zeros(100, 2, 8))
    # which layers read a producer's raw output through its batch norm (forward AND weight gradient stage through registers):
    # precision 0 = fp32-class entries (few channels), 1 = split3, 2 = bf16 (32 / 64 channels in, 32 out)
    assert ops.conv2d_affine_input_ok(ops.conv_desc(32, 224, 298, 8, 8, 3, 3, 1, "SAME"), 0)
    assert ops.conv2d_affine_input_ok(ops.conv_desc(32, 112, 149, 8, 32, 3, 3, 1, "SAME"), 0)
    assert ops.conv2d_affine_input_ok(ops.conv_desc(32, 112, 149, 32, 32, 3, 3, 1, "SAME"), 1)
    assert ops.conv2d_affine_input_ok(ops.conv_desc(32, 112, 149, 64, 32, 3, 3, 1, "SAME"), 2)
    assert not ops.conv2d_affine_input_ok(ops.conv_desc(32, 112, 149, 32, 32, 3, 3, 1, "SAME"), 0)     # 32 channels: not on the fp32 entries
    assert not ops.conv2d_affine_input_ok(ops.conv_desc(32, 112, 149, 32, 64, 3, 3, 1, "SAME"), 1)     # 64 outputs: no halo weight gradient
    assert not ops.conv2d_affine_input_ok(ops.conv_desc(32, 224, 298, 8, 8, 3, 3, 2, "SAME"), 0)       # strided
    assert not ops.conv2d_affine_input_ok(ops.conv_desc(32, 28, 37, 64, 64, 3, 3, 1, "SAME"), 1)       # below the size rule


def test_tf_padding_geometry():
    from acimg import ops

    assert ops.same_out_pad(224, 3, 2) == (112, 0) and ops.same_out_pad(149, 3, 2) == (75, 1)
    d = ops.conv_desc(1, 224, 298, 4, 64, 7, 7, 2, 3)
    assert (d.OH, d.OW, d.pad_t) == (112, 149, 3)
    d = ops.conv_desc(1, 56, 75, 128, 128, 3, 3, 2, 1)
    assert (d.OH, d.OW) == (28, 38)
    d = ops.conv_desc(1, 36, 48, 128, 128, 3, 3, 3, "SAME")
    assert (d.OH, d.OW, d.pad_t, d.pad_l) == (12, 16, 0, 0)
    d = ops.deconv_desc(1, 12, 16, 128, 128, 2, 2, 3)
    assert (d.OH, d.OW) == (36, 48)


def test_error_codes_and_text(lib):
    import ctypes as C

    from acimg import _lib, ops

    d = ops.conv_desc(1, 4, 4, 6, 8, 3, 3)     # C not a multiple of 4: rejected before any launch
    rc = lib.acimg_conv2d_fwd(C.byref(d), 16, 16, None, 16, None, None, 0, None, None, 0, None, None)
    assert rc == -1 and "multiples of 4" in _lib.last_error()
    with pytest.raises(_lib.AcimgError):
        _lib.check(rc, "conv2d_fwd")
    rc = lib.acimg_mfcc_frontend(None, None, None, None, None, 0, 0, None)
    assert rc == -1 and "nframes" in _lib.last_error()


def test_gram_stats_host_side(lib):
    """acimg_gram_stats (include/acimg.h, round 4): workspace sizes are pure host arithmetic and the argument checks fire
    before any launch (no GPU here): unsupported channel counts, a lo plane that overlaps the hi plane, a short workspace"""
    import ctypes as C

    from acimg import _lib

    assert lib.acimg_gram_stats_workspace(134400, 128) > 16 * 2 ** 20          # ~256 pixel ranges x 64 KiB + V + sums
    assert lib.acimg_gram_stats_workspace(134400, 64) > 0 and lib.acimg_gram_stats_workspace(8512, 512) > 0
    assert lib.acimg_gram_stats_workspace(1000, 96) == 0 and lib.acimg_gram_stats_workspace(0, 128) == 0
    buf = (C.c_float * 64)()
    big = (C.c_char * 4096)()
    a = C.addressof(big) + (-C.addressof(big)) % 16
    rc = lib.acimg_gram_stats(a, 2048, 16, 96, C.addressof(buf), 4, 4, None, None, None, None, 0.9, 1e-5, C.addressof(buf),
                              C.addressof(buf), a, 4096, None)
    assert rc == -1 and "multiple of 128" in _lib.last_error()
    rc = lib.acimg_gram_stats(a, 16, 16, 64, C.addressof(buf), 4, 4, None, None, None, None, 0.9, 1e-5, C.addressof(buf),
                              C.addressof(buf), a, 4096, None)
    assert rc == -1 and "overlapping" in _lib.last_error()
    rc = lib.acimg_gram_stats(a, 2048, 16, 64, C.addressof(buf), 4, 4, None, None, None, None, 0.9, 1e-5, C.addressof(buf),
                              C.addressof(buf), a, 64, None)
    assert rc == -1 and "workspace too small" in _lib.last_error()


def _build(num_skip, ae, batch=2):
    from acimg.flags import FLAGS
    from acimg.session import Session
    from acimg.trainer import Trainer
    from acimg.unet_acresnet import UNetAc
    from acimg.vision import ResNet50Model

    FLAGS.model, FLAGS.ae = "UNet", int(ae)
    sess = Session(torch.device("cpu"))
    tr = Trainer(UNetAc(input_shape=[36, 48, 12], embedding=ae, num_skip=num_skip),
                 ResNet50Model(input_shape=[224, 298, 3], num_classes=None), session=sess)
    tr._build_functions(batch_size=batch)
    return tr, sess


@pytest.mark.parametrize("num_skip,ae", [(1, False), (2, False), (0, False), (1, True)])
def test_variables_match_the_reference_inventory(lib, num_skip, ae):
    """same TF variable names / shapes as the oracle's restatement of the reference graph, and a
    lossless round trip TF layout -> padded internal layout -> TF layout"""
    from oracle import trainer as otr

    tr, sess = _build(num_skip, ae)
    orc = otr.Oracle(num_skip=num_skip, embedding=ae, randomize=True)
    ref = orc.state_dict()
    assert set(sess.store.tf_names()) == set(ref.keys())
    loaded = sess.store.load_state(ref, strict=True)
    assert len(loaded) == len(ref)
    sd = sess.store.state_dict()
    for k, v in ref.items():
        assert tuple(sd[k].shape) == tuple(v.shape) and torch.equal(sd[k], v.detach()), k
    assert set(tr.modelac.train_vars + tr.modelimages.train_vars) == set(orc.train_names)
    # pad entries of the internal layouts are zero
    w = sess.store.p("UNetAcRes/layer2/conv_2/kernel")
    assert tuple(w.shape) == (3, 3, 136, 136) and float(w[:, :, 133:, :].abs().max()) == 0
    g = tr.primary
    assert len(g.plan_train) >= 190 and len(g.plan_eval) >= 150
    g.plan_train.finalize()     # every pointer resolves (flat buffers, workspace)


def test_flat_buffer_order_and_buckets(lib):
    from acimg import dp

    tr, sess = _build(1, False)
    ranges = sess.store.train_ranges()
    names = [n for n, _, _ in ranges]
    assert names[0] == "resnet_v1_50/conv_map/weights"
    assert names.index("UNetAcRes/final/kernel") < names.index("UNetAcRes/layer6/conv_1/kernel") < \
        names.index("UNetAcRes/heads/kernel") < names.index("UNetAcRes/layer1/conv_1/kernel")
    for (_, off, n), (_, off2, _) in zip(ranges, ranges[1:]):
        assert off % 64 == 0 and off + n <= off2
    tr.enable_data_parallel()
    b = tr.buckets
    assert len(b) == 5 and b[0][0] == 0 and b[-1][1] <= sess.store.train_numel()
    assert all(x[1] == y[0] for x, y in zip(b, b[1:]))
    big = max(b, key=lambda t: t[1] - t[0])
    assert big[2].endswith("heads/bias") and (big[1] - big[0]) * 4 > 30e6     # the 33 MB mean/std bucket
    assert not tr.comm.enabled and tr.comm.grad_scale == 1.0                  # single process: no-op


def test_flags_and_synthetic_loader():
    from acimg.data import SyntheticDataLoader
    from acimg.flags import _Flags

    f = _Flags()
    assert (f.batch_size, f.learning_rate, f.latent_loss, f.num_skip_conn, f.MSE, f.huber_loss) == \
        (8, 0.001, 0.000001, 1, 1, 1)
    f.parse(["--batch_size", "64", "--learning_rate", "1e-4", "--num_skip_conn", "2", "--mfcc", "1"])
    assert (f.batch_size, f.learning_rate, f.num_skip_conn, f.mfcc) == (64, 1e-4, 2, 1)
    dl = SyntheticDataLoader(10, 4)
    batches = list(dl.data)
    assert dl.total_batches == 3 and [b[0].shape[0] for b in batches] == [4, 4, 2]
    ac, mf, vid = batches[0][:3]
    assert ac.shape == (4, 36, 48, 12) and mf.shape == (4, 12) and vid.shape == (4, 224, 298, 3)
    assert float(ac.amin()) == 0 and float(ac.amax()) == 1 and float(mf.amin(1).max()) == 0


DP_WORKER = r'''
import os, sys
sys.path.insert(0, os.path.join(%(root)r, "acoustic-image-generation_amd"))
import torch, torch.distributed as dist
from acimg import dp
rank = int(os.environ["RANK"])
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%(port)d", rank=rank, world_size=2)
n = 1000
ranges = [("a", 0, 300), ("b", 320, 400), ("c", 768, 232)]
buckets = dp.make_buckets(ranges, ["a", "b"])
assert buckets == [(0, 300, "a"), (300, 720, "b"), (720, 1000, "c")], buckets
flat = torch.arange(n, dtype=torch.float32) * (rank + 1)
comm = dp.GradComm(flat, buckets)
assert comm.enabled and comm.world == 2 and comm.grad_scale == 0.5
for i in (1, 2, 0):            # buckets fire in backward-completion order, not index order
    comm.bucket_ready(i)
comm.wait()
exp = torch.arange(n, dtype=torch.float32) * 3
assert torch.equal(flat, exp), (flat - exp).abs().max()
s = comm.allreduce_scalars(torch.tensor([float(rank)]))
assert abs(float(s) - 0.5) < 1e-6
# gradient-accumulation (strong-scaling) steps exchange the whole flat buffer once
flat2 = torch.full((n,), float(rank + 1))
comm2 = dp.GradComm(flat2, buckets)
comm2.allreduce_all()
assert torch.equal(flat2, torch.full((n,), 3.0))
assert dp.shards_per_rank(256, 32, 2) == 4 and dp.shards_per_rank(256, 32, 8) == 1
for bad in ((250, 32, 2), (96, 32, 2)):
    try:
        dp.shards_per_rank(*bad); raise SystemExit("accepted %%r" %% (bad,))
    except ValueError:
        pass
# checkpoint time: ONE writer; per-replica BN moving statistics averaged, frozen gamma/beta untouched
assert dp.is_writer() == (rank == 0)
state = torch.cat([torch.full((8,), 0.1), torch.full((8,), float(rank)), torch.full((8,), 10.0 * (rank + 1))])
dp.average_moving_statistics(state, [(8, 8), (16, 8)])
assert torch.equal(state[:8], torch.full((8,), 0.1)) and torch.equal(state[8:16], torch.full((8,), 0.5))
assert torch.equal(state[16:], torch.full((8,), 15.0))
# validation under data parallelism (Trainer.train): each rank's loss sum and sample count differ; the reduced totals -
# and with them the best-epoch branch that leads into a collective - are the same on every rank
tot = dp.allreduce_sums([1.5 + rank, 3 + rank])
assert tot == [4.0, 7.0], tot
best = 0.6
took = (tot[0] / tot[1]) <= best
flags = [None, None]
dist.all_gather_object(flags, took)
assert flags[0] == flags[1]
# a forced exchange at world size > 1 is the plain exchange; the `force` argument (bench.py --dp-force) replaces the
# environment variable the constructor used to read
assert dp.GradComm(torch.zeros(n), buckets, force=True).enabled
# bench.py's multi-GPU self-verification (config.dp): the no-exchange leg mutes every collective form, the exchange leg
# restores them; on CPU tensors nothing is timed (the event pairs exist for device buffers only)
flat3 = torch.full((n,), float(rank + 1))
comm3 = dp.GradComm(flat3, buckets)
comm3.start_timing()
assert comm3.timing == [] and comm3.exchange_ms() == 0.0
comm3.muted = True
for i in (0, 1, 2):
    comm3.bucket_ready(i)
comm3.wait(); comm3.allreduce_all()
assert torch.equal(flat3, torch.full((n,), float(rank + 1)))      # nothing was exchanged
comm3.muted = False
comm3.allreduce_all()
assert torch.equal(flat3, torch.full((n,), 3.0)) and comm3.timing == []
import tempfile
marker = os.path.join(%(tmp)r, "writer_%%d" %% rank)
if dp.is_writer():
    open(marker, "w").write("x")
dist.barrier(); dist.destroy_process_group()
print("rank", rank, "ok")
'''


def test_gradient_exchange_two_ranks_gloo(tmp_path):
    """N>1 path on CPU: 2 processes, gloo, bucketed all-reduce of a flat gradient buffer"""
    script = tmp_path / "dp_worker.py"
    script.write_text(DP_WORKER % {"root": ROOT, "port": 29500 + os.getpid() % 2000, "tmp": str(tmp_path)})
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(os.environ, RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(2)]
    outs = [p.communicate(timeout=240)[0].decode() for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o
    assert all("ok" in o for o in outs)
    assert sorted(f for f in os.listdir(tmp_path) if f.startswith("writer_")) == ["writer_0"]


def test_single_process_dp_helpers():
    from acimg import dp
    import torch

    assert dp.is_writer()
    t = torch.arange(6.0)
    assert dp.average_moving_statistics(t, [(0, 6)]) is t and torch.equal(t, torch.arange(6.0))
    assert dp.shards_per_rank(256, 32, 1) == 8


def test_frontend_tables_match_oracle():
    """host-built front-end tables (window, mel bank, folded DCT*norm*lifter) vs the pinned oracle"""
    from acimg import frontend
    from oracle import frontend as ofe

    t = frontend.tables()
    np.testing.assert_array_equal(t["window"], ofe.tukey_window())
    np.testing.assert_array_equal(t["melfb"], ofe.createfilters())
    np.testing.assert_allclose(t["dctl"], ofe.dct_base() * ofe.MFNORM * ofe.lifter()[None, :], rtol=1e-15)
    x = np.random.RandomState(0).rand(5, 12)
    ref = ofe.find_logen(x)
    got = 1.0 / np.exp(x @ t["idct"]).sum(-1)
    np.testing.assert_allclose(got, ref, rtol=1e-12)


def test_bench_contract_on_cpu():
    """bench.py: defaults of the driver contract, the cpu_baseline object on a one-image sample, and NO CPU fallback —
    without a GPU the measured path must fail loudly instead of timing something else"""
    import importlib.util
    import subprocess
    import sys
    import types

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    argv = sys.argv
    try:
        sys.argv = ["bench.py"]
        spec.loader.exec_module(bench)
        args = bench.parse()
    finally:
        sys.argv = argv
    assert (args.gpus, args.batch, args.workload, args.precision) == (1, 32, "trainer_mask", "f16x3")
    assert args.steps * 0.01 < 60 and args.warmup >= 1          # minutes at ~10 ms / step
    assert (args.cpu_batch, args.scaling, args.global_batch) == (32, "weak", 256)    # CPU leg at the GPU leg's batch
    small = types.SimpleNamespace(num_skip=1, cpu_batch=1, cpu_steps=1, cpu_threads=2, cpu_budget=60.0)
    cb = bench.cpu_baseline(small)
    assert set(cb) >= {"value", "unit", "cores", "kind", "sample"} and cb["kind"] == "port" and cb["value"] > 0
    assert cb["unit"] == "images/s" and cb["cores"] == 2 and "1-skip" in cb["sample"]
    import torch
    torch.set_num_threads(bench.host_cores())
    assert 1 <= bench.host_cores() <= len(os.sched_getaffinity(0))
    # the roofline's counter traffic comes from a committed PMC summary, never from a literal in bench.py
    src = open(os.path.join(root, "bench.py")).read()
    assert "252.0e6" not in src and "28.67e9" not in src
    prof = bench.load_traffic_profile("igemm_split3d_kernel<128,128,2,4,512,2,2>")
    if prof is not None:
        assert prof["file"].startswith("profiles/r") and prof["step_bytes"] > 1e9
        assert prof["kernel_bytes_per_launch"] is None or prof["kernel_bytes_per_launch"] > 1e6
    import torch
    if not torch.cuda.is_available():
        r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "1", "--warmup", "0",
                            "--no-cpu-baseline", "--no-secondary"], capture_output=True, text=True, timeout=300)
        assert r.returncode != 0 and "{" not in r.stdout, r.stdout[-500:]


def test_no_wide_buffer_store_with_register_soffset(tmp_path):
    """Enforced invariant: gfx950 takes a data register that a VALU instruction overwrites right
    after a > 8-byte `buffer_store` when the store's soffset is an SGPR - the ISA manual and the compiler's hazard
    recognizer only cover the immediate / null soffset forms (0.03 % wrong lo halves, timing dependent, found in round 3).
    Every 12- / 16-byte buffer store of the shipped code object must therefore carry an immediate or zero soffset: the
    device code of libacimg.so is disassembled and every such store checked."""
    import re
    import shutil

    from acimg import _lib

    objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
    if not os.path.exists(objdump):
        pytest.skip("no llvm-objdump")
    so = tmp_path / "libacimg.so"
    shutil.copy(_lib.LIB_PATH, so)
    subprocess.check_call([objdump, "--offloading", str(so)], stdout=subprocess.DEVNULL, cwd=str(tmp_path))
    objs = [f for f in os.listdir(tmp_path) if "amdgcn" in f and "gfx950" in f]
    assert objs, os.listdir(tmp_path)
    pat = re.compile(r"buffer_store_dwordx[34]\s+[va]\[\d+:\d+\],\s*(?:v\d+|off),\s*s\[\d+:\d+\],\s*(\S+)")
    n_stores, bad = 0, []
    for f in objs:
        dis = subprocess.check_output([objdump, "-d", str(tmp_path / f)]).decode()
        for line in dis.splitlines():
            if "buffer_store_dwordx3" not in line and "buffer_store_dwordx4" not in line:
                continue
            m = pat.search(line)
            assert m, "unparsed store form: " + line.strip()
            n_stores += 1
            soff = m.group(1).rstrip(",")
            if re.fullmatch(r"s\d+|m0|ttmp\d+|vcc_lo|vcc_hi", soff):
                bad.append(line.strip())
    assert n_stores > 100, n_stores          # the trunk / generator kernels are in there
    assert not bad, "16-byte buffer stores with a REGISTER soffset (gfx950 store-data hazard):\n" + "\n".join(bad[:10])


def test_library_reads_no_environment_and_configure_validates(lib):
    """include/acimg.h: the only process-wide state is the tuning record of acimg_configure; the library never calls
    getenv (the Python host maps ACIMG_* variables onto acimg_configure once, at load time)"""
    import ctypes as C

    from acimg import _lib, ops

    und = subprocess.check_output(["nm", "-D", "--undefined-only", _lib.LIB_PATH]).decode()
    assert "getenv" not in und
    assert "acimg_set_ticket_buffer" not in set(_lib.PROTOTYPES)
    cfg = _lib.Config()
    assert lib.acimg_config_default(C.byref(cfg)) == 0
    assert (cfg.splitk_cut, cfg.splitk_target, cfg.splitk_handoff, cfg.wgrad_minpix, cfg.wgrad_halo, cfg.split3_tile_bm,
            cfg.split3_tile_bn, cfg.tail_split, cfg.tail_s, cfg.trunk_persistent, cfg.trunk_bk, cfg.trunk_stagger, cfg.trunk_dma_pos,
            cfg.trunk_ring, cfg.trunk_ring_bm, cfg.trunk_halo) == (320, 768, 1, 128, 1, 0, 0, 1, 0, 1, 0, 0, 0, 1, 0, 0)
    # the ring kernel's row tile decides the statistics rows of a pre-split conv: 14x19 512->512 3x3 at batch 32 is on it
    d14 = ops.conv_desc(32, 14, 19, 512, 512, 3, 3)
    assert ops.conv2d_fwd_split3p_stats_rows(d14) == -(-32 * 14 * 19 // 128)
    d = ops.conv_desc(4, 12, 16, 128, 128, 3, 3)
    base = ops.conv2d_fwd_tiling(d)
    assert base[2] > 1
    try:
        got = _lib.configure_from_env({"ACIMG_SPLITK_CUT": "1", "ACIMG_NO_TAIL_SPLIT": "1", "ACIMG_SPLIT3_TILE": "64x128"})
        assert (got.splitk_cut, got.tail_split, got.split3_tile_bm, got.split3_tile_bn) == (1, 0, 64, 128)
        assert ops.conv2d_fwd_tiling(d)[2] == 1            # the heuristics read the record, not the environment
        os.environ["ACIMG_SPLITK_CUT"] = "320"
        assert ops.conv2d_fwd_tiling(d)[2] == 1
    finally:
        os.environ.pop("ACIMG_SPLITK_CUT", None)
        _lib.configure()
    assert ops.conv2d_fwd_tiling(d) == base
    bad = _lib.Config()
    lib.acimg_config_default(C.byref(bad))
    bad.split3_tile_bm, bad.split3_tile_bn = 96, 96
    assert lib.acimg_configure(C.byref(bad)) == -1 and "tile" in _lib.last_error()
    assert lib.acimg_configure(None) == -1


def test_plan_side_lane_bookkeeping():
    """ops.Plan: calls added with side=True are remembered by index (and re-based by extend); fork / join are host
    hooks; without a GPU (or in eager mode) everything stays on the one stream and the one workspace"""
    from acimg import ops

    a = ops.Plan("cpu")
    a.add("x", lambda *args: 0, 1)
    a.fork()
    a.add("y", lambda *args: 0, 2, side=True)
    a.join()
    assert a.side == set() and len(a.calls) == 2 and a.side_ws is a.ws          # no GPU: one lane

    class FakeCudaPlan(ops.Plan):
        def __init__(self):
            ops.Plan.__init__(self, "cpu")
            self._cuda = True

        @property
        def side_ws(self):
            return self.ws
    b = FakeCudaPlan()
    b.add("x", lambda *args: 0, 1)
    b.add("w", lambda *args: 0, 2, side=True)
    b.add("d", lambda *args: 0, 3)
    c = FakeCudaPlan()
    c.add("head", lambda *args: 0, 0)
    c.extend(b)
    c.add("w2", lambda *args: 0, 4, side=True)
    assert b.side == {1} and c.side == {2, 4} and [n for n, _, _ in c.calls] == ["head", "x", "w", "d", "w2"]
    # calls between a fork and the next join (either lane) may share the chip: bench.py excludes them from
    # `roofline.achieved_exclusive`
    e = FakeCudaPlan()
    e.add("a", lambda *args: 0, 1)
    e.add_hook(lambda: None, "fork")
    e.add("w", lambda *args: 0, 1, side=True)
    e.add("d", lambda *args: 0, 1)
    e.add_hook(lambda: None, "join")
    e.add("z", lambda *args: 0, 1)
    assert e.shared_calls() == {2, 3}


def test_joint_oracle_tables_reduce_to_the_12x16_feature_map():
    """oracle/joint.py: the three split VAEs of trainermulti.py — parameter inventories (UNetSound22 14.7 M, UNetAc2 9.32 M =
    SURVEY A.3's unet_noconc, Unet2's heads 2 x 12*16*512*1024) and, for the two small models, one forward pass: the
    encoders reduce their inputs to 12x16, the decoders return to the input size, losses compose as trainermulti.py:58-81"""
    import torch
    from oracle import joint

    n = {m: sum(int(torch.tensor(s).prod()) for s in joint.param_shapes(m).values()) for m in joint.MODELS}
    assert n["UNetAc2"] == 9317700 and n["UNetSound22"] == 14716449 and n["Unet2"] == 221621699
    assert joint.param_shapes("Unet2")["UNet/mean/kernel"] == (12, 16, 512, 1024)
    assert joint.is_encoder_var("UNetSound22", "UNetAudio/layer4/pool_2/kernel")
    assert not joint.is_encoder_var("UNetSound22", "UNetAudio/layer6/conv_1/kernel")
    g = torch.Generator().manual_seed(3)
    for model in ("UNetAc2", "UNetSound22"):
        cfg = joint.MODELS[model]
        p = joint.init_params(model, seed=5)
        net = joint._Net(model, p, True)
        x = torch.rand(1, cfg["input_hw"][0], cfg["input_hw"][1], cfg["cin"], generator=g)
        f = net.encoder(x)
        assert tuple(f.shape) == (1, 12, 16, joint.feature_channels(model))
        out = net.decoder(f, torch.randn(1, cfg["Z"], generator=g))
        assert tuple(out["output"].shape) == tuple(x.shape) and float(out["std"].min()) > 0
        if cfg["bn"]:
            assert len(net.new_stats) == 2 * sum(1 for k in p if k.endswith("gamma"))
            assert float(joint.regulariser(model, p)) > 0
        else:
            assert not net.new_stats and joint.regulariser(model, p) == 0.0
