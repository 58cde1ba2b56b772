"""CPU: host-side pieces of the section-8f components (no kernel runs): PNG writer, J statistics, affine-matrix host math,
optimizer argument checks."""
import os
import sys

import numpy as np
import pytest
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def test_png_writer_round_trips_through_pil(tmp_path):
    from PIL import Image
    from osvos_pytorch_amd.results import write_png
    rng = np.random.default_rng(0)
    for shape in [(1, 1), (7, 13), (480, 854)]:
        img = rng.integers(0, 256, shape, dtype=np.uint8)
        p = str(tmp_path / ("a%dx%d.png" % shape))
        write_png(p, img)
        back = Image.open(p)
        assert back.mode == "L" and back.size == (shape[1], shape[0]) and np.array_equal(np.array(back), img)
    with pytest.raises(ValueError):
        write_png(str(tmp_path / "bad.png"), np.zeros((2, 2, 3), dtype=np.uint8))


def test_davis_statistics():
    from osvos_pytorch_amd.results import davis_statistics
    st = davis_statistics([1.0, 0.0])
    assert st["mean"] == 0.5 and st["recall"] == 0.5
    st = davis_statistics([0.9] * 4 + [0.5] * 4 + [0.5] * 4 + [0.1] * 4)
    assert abs(st["decay"] - 0.8) < 1e-12 and abs(st["mean"] - 0.5) < 1e-12 and st["recall"] == 0.25
    with pytest.raises(ValueError):
        davis_statistics([])


def test_rotation_matrix_inverse_matches_the_oracle_and_inverts():
    from oracle import augment_ref as A
    from osvos_pytorch_amd.augment import rotation_matrix_inverse
    for (w, h, rot, sc) in [(854, 480, 17.5, 1.1), (31, 24, -30.0, 0.75), (10, 10, 0.0, 1.0)]:
        m = rotation_matrix_inverse(w, h, rot, sc)
        fwd = A.get_rotation_matrix_2d((w / 2, h / 2), rot, sc)
        assert np.array_equal(np.array(m), A.invert_affine(fwd))            # same operations, same order: bit-equal doubles
        full = np.vstack([fwd, [0, 0, 1]]) @ np.vstack([np.array(m).reshape(2, 3), [0, 0, 1]])
        assert np.allclose(full, np.eye(3), atol=1e-9)


def test_fused_sgd_argument_checks_and_cpu_refusal():
    from osvos_pytorch_amd.optim import FusedSGD
    p = [torch.zeros(3, requires_grad=True)]
    for kw in ({"nesterov": True}, {"dampening": 0.1}, {"lr": -1.0}, {"momentum": -0.1}, {"weight_decay": -1.0}):
        with pytest.raises(ValueError):
            FusedSGD(p, **kw)
    o = FusedSGD(p, lr=0.1, momentum=0.9)
    o.step()                                   # no gradients: nothing to do, nothing touched
    p[0].grad = torch.ones(3)
    with pytest.raises(RuntimeError):
        o.step()                               # CPU parameter: refused, no fallback
    assert torch.equal(p[0].detach(), torch.zeros(3))


def test_bench_gpus_n_stands_up_its_own_ranks_or_says_why_not():
    """`python bench.py --gpus N` with no launcher around it re-executes itself under torch.distributed.run with N ranks (VERDICT r02:
    --gpus used to be parsed and ignored).  On a box with fewer GPUs it must refuse with a clear message instead of reporting n_gpus 1."""
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("multi-GPU box: the real launch is the driver's scaling run")
    out = subprocess.run([sys.executable, os.path.join(repo, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                         capture_output=True, text=True, env=env, timeout=300)
    assert out.returncode == 2 and "--gpus 2: this node shows" in out.stderr, (out.returncode, out.stderr[-400:])


def test_bench_gpus_8_launch_command_and_ports():
    """What `python bench.py --gpus 8` would start (no 8-GPU node is reachable from here): ONE torch.distributed.run command
    for 8 ranks on 127.0.0.1 carrying the bench arguments, and an environment with HSA_ENABLE_IPC_MODE_LEGACY=0 and an OSVOS_COMM_PORT of the
    job's own -- different from the rendezvous port and different between successive launches (the driver runs N = 1, 2, 4, 8 back to back)."""
    import importlib.util
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_launch_test", os.path.join(repo, "bench.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    seen = set()
    for n in (2, 4, 8):
        cmd, env = b.rank_launch_command(n, ["--gpus", str(n), "--steps", "20", "--warmup", "5"])
        assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and cmd[cmd.index("--nproc-per-node") + 1] == str(n)
        assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[-6:] == ["--gpus", str(n), "--steps", "20", "--warmup", "5"]
        assert os.path.basename(cmd[cmd.index("--master-port") + 2]) == "bench.py"
        mport, cport = int(cmd[cmd.index("--master-port") + 1]), int(env["OSVOS_COMM_PORT"])
        assert 1024 <= mport <= 65535 and 1024 <= cport <= 65535 and mport != cport
        assert env["HSA_ENABLE_IPC_MODE_LEGACY"] == "0" and "RANK" not in env and "WORLD_SIZE" not in env
        seen.update((mport, cport))
    assert len(seen) == 6
    # ... and the communicator takes that port
    from osvos_pytorch_amd import parallel
    old = os.environ.get("OSVOS_COMM_PORT")
    os.environ["OSVOS_COMM_PORT"] = env["OSVOS_COMM_PORT"]
    try:
        assert parallel.comm_port() == int(env["OSVOS_COMM_PORT"])
    finally:
        if old is None:
            del os.environ["OSVOS_COMM_PORT"]
        else:
            os.environ["OSVOS_COMM_PORT"] = old


def bytescale_scipy11(data):
    """scipy 1.1 misc.bytescale(data, cmin=None, cmax=None, high=255, low=0) as toimage() calls it for mode 'L' (float32 input): the
    restatement the result writer (osvos_mask_to_bytes) is tested against in tests/test_gpu_ops.py"""
    cmin, cmax = data.min(), data.max()
    cscale = cmax - cmin
    if cscale == 0:
        cscale = 1
    scale = np.float32(float(255 - 0) / float(cscale))
    bytedata = (data - cmin) * scale + np.float32(0)
    return (bytedata.clip(0, 255) + np.float32(0.5)).astype(np.uint8)


def test_bytescale_matches_the_scipy_golden():
    """the restated byte scaling of scipy.misc.imsave (train_online.py:183-189) against what scipy <= 1.1 itself produced
    (tools/make_cv2_goldens.py); skipped -- rows a13 / f4 of SURVEY 8 stay "parity unpinned" -- while the golden file is absent"""
    path = os.path.join(REPO, "tests", "golden", "bytescale.npz")
    if not os.path.exists(path):
        pytest.skip("tests/golden/bytescale.npz absent: scipy.misc.imsave / bytescale left scipy in 1.2; run `python tools/make_cv2_goldens.py "
                    "--only bytescale` with scipy <= 1.1 and commit the file")
    g = np.load(path)
    for n in range(3):
        pred = np.squeeze(1 / (1 + np.exp(-g["logits"][n].transpose(1, 2, 0))))
        assert np.array_equal(bytescale_scipy11(pred), g["bytescale%d" % n]), n
        assert np.array_equal(g["imsave%d" % n], g["bytescale%d" % n]), n          # imsave writes exactly those bytes


def test_the_scripts_sgd_group_table_is_the_oracles():
    """ONE parameter-group table: the product's ``train_common.make_sgd`` (what train_online.py / train_parent.py / bench.py build their
    optimizer from) against the oracle's independent restatement of train_online.py:79-88 / train_parent.py:87-103 (``torch_ref.sgd_groups``,
    itself pinned to the real reference's SGD trajectory by tests/test_oracle_golden.py): same groups in the same order, same tensors by
    name, same lr / weight_decay / momentum -- for both loops."""
    import io
    import contextlib
    import networks.vgg_osvos as vo
    from oracle import torch_ref
    from osvos_pytorch_amd.train_common import make_sgd
    with contextlib.redirect_stdout(io.StringIO()):
        net = vo.OSVOS(pretrained=0)
    names = {id(p): k for k, p in net.named_parameters()}
    p = dict(net.named_parameters())
    for mode in ("online", "parent"):
        ours = make_sgd(net, mode, lr=3e-8, fused=False)
        ref = torch.optim.SGD(torch_ref.sgd_groups(p, lr=3e-8, mode=mode), lr=3e-8, momentum=0.9)
        assert len(ours.param_groups) == len(ref.param_groups) == (8 if mode == "online" else 10)
        for a, b in zip(ours.param_groups, ref.param_groups):
            assert [names[id(t)] for t in a["params"]] == [names[id(t)] for t in b["params"]]
            assert a["lr"] == b["lr"] and a["weight_decay"] == b["weight_decay"] and a["momentum"] == b["momentum"] == 0.9
