"""GPU: the two entry scripts run end to end in --synthetic mode (tiny frames, few epochs) and write
the files the reference's scripts write (checkpoint naming, result PNGs)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, tmp_path, extra_env=None):
    env = dict(os.environ, OSVOS_SAVE_ROOT=str(tmp_path), OSVOS_MODELS_DIR=str(tmp_path), PYTHONPATH=REPO)
    env.update(extra_env or {})
    r = subprocess.run([sys.executable] + args, cwd=REPO, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    return r.stdout


def test_train_online_synthetic(tmp_path):
    out = _run(["train_online.py", "--synthetic", "--epochs", "10", "--height", "48", "--width", "64"], tmp_path,
               {"SEQ_NAME": "blackswan"})
    assert "Online training time" in out and "Loss:" in out
    assert os.path.exists(os.path.join(str(tmp_path), "blackswan_epoch-9.pth"))
    assert os.path.exists(os.path.join(str(tmp_path), "Results", "blackswan", "00000.png"))


def test_train_online_test_phase_on_fp16_pairs_writes_the_same_masks(tmp_path):
    """--test-precision fp32h2 (round 6): training in the default arithmetic, the TEST forwards (reference train_online.py:172-189) on two FP16 pieces
    per operand under block exponents -- a re-pack between the phases, same result files; the masks equal the default run's except for pixels whose
    logit sits within rounding of the threshold (allowed: 1 in 1000)."""
    import numpy as np
    from PIL import Image
    a, b = tmp_path / "a", tmp_path / "b"
    a.mkdir(); b.mkdir()
    args = ["train_online.py", "--synthetic", "--epochs", "10", "--height", "48", "--width", "64"]
    _run(args, a, {"SEQ_NAME": "blackswan"})
    out = _run(args + ["--test-precision", "fp32h2"], b, {"SEQ_NAME": "blackswan"})
    assert "Online training time" in out
    pa = np.asarray(Image.open(os.path.join(str(a), "Results", "blackswan", "00000.png"))).astype(np.int32)
    pb = np.asarray(Image.open(os.path.join(str(b), "Results", "blackswan", "00000.png"))).astype(np.int32)
    assert pa.shape == pb.shape and (np.abs(pa - pb) > 1).mean() <= 1e-3, float((np.abs(pa - pb) > 1).mean())


def test_train_parent_synthetic(tmp_path):
    out = _run(["train_parent.py", "--synthetic", "4", "--epochs", "5", "--n-ave-grad", "2", "--height", "40", "--width", "56"], tmp_path)
    assert "Loss 4:" in out and "***Testing (epoch 4, 2 frames) *** Loss 4" in out


def test_data_parallel_path_on_rccl_single_rank(tmp_path):
    """'nccl' (= RCCL) process group with one rank, all-reduce forced: bit-identical to the loop without a process group, and the
    overlapped chunked path is the one that ran."""
    out = _run(["tools/dp_selfcheck.py"], tmp_path, {"DP_H": "60", "DP_W": "107"})
    assert "DP_SELFCHECK_OK" in out, out
    assert "bit-identical" in out


def test_train_parent_device_augment_synthetic(tmp_path):
    """--device-augment: uint8 frames -> pinned staging ring -> H2D -> osvos_augment_frame, through the parent loop."""
    out = _run(["train_parent.py", "--synthetic", "4", "--epochs", "2", "--n-ave-grad", "2", "--height", "40", "--width", "56", "--device-augment"], tmp_path)
    assert "Loss 4:" in out and "optimizer steps taken: 4" in out


def test_train_online_device_augment_synthetic(tmp_path):
    out = _run(["train_online.py", "--synthetic", "--epochs", "10", "--height", "48", "--width", "64", "--device-augment"], tmp_path,
               {"SEQ_NAME": "blackswan"})
    assert "Online training time" in out and os.path.exists(os.path.join(str(tmp_path), "Results", "blackswan", "00000.png"))


def test_train_parent_through_both_rccl_backends_on_one_rank_is_bit_identical(tmp_path):
    """The SCRIPT's data-parallel path on the real backend with one rank (OSVOS_DP_FORCE=1: communicator first, flat gradient arena, one
    all-reduce per optimizer step, epoch statistics + sharded validation through the same communicator): torch.distributed's nccl (= RCCL)
    group, and RCCL through the library's own C ABI with the chunked overlap behind the gradient-ready events (OSVOS_DP_BACKEND=abi,
    OSVOS_DP_OVERLAP=1, its id store on an explicit OSVOS_COMM_PORT).  A one-rank sum is the identity: the weights after 8 optimizer steps
    must equal the run without any communicator BIT FOR BIT, and the printed statistics must be the same lines."""
    import torch
    argv = ["train_parent.py", "--synthetic", "4", "--epochs", "4", "--n-ave-grad", "2", "--height", "40", "--width", "56", "--snapshot", "2",
            "--test-interval", "2", "--init-seed", "11"]      # (the same initial weights in the three processes)
    runs = {}
    for tag, env in {"plain": {}, "torch": {"OSVOS_DP_FORCE": "1", "MASTER_PORT": "29641"},
                     "abi": {"OSVOS_DP_FORCE": "1", "OSVOS_DP_BACKEND": "abi", "OSVOS_DP_OVERLAP": "1", "OSVOS_COMM_PORT": "29655", "MASTER_PORT": "29643"}}.items():
        d = tmp_path / tag
        d.mkdir()
        out = _run(argv, d, env)
        assert "optimizer steps taken: 8" in out and "***Testing (epoch 3, 2 frames) *** Loss 4" in out, out[-1500:]
        runs[tag] = (torch.load(str(d / "parent_epoch-3.pth")), [l for l in out.splitlines() if l.startswith("Loss ") or l.startswith("***Testing")])
    for tag in ("torch", "abi"):
        for k, v in runs["plain"][0].items():
            assert torch.equal(v, runs[tag][0][k]), (tag, k)
        assert runs[tag][1] == runs["plain"][1], tag
