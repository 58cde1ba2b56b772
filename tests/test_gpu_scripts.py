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


def test_train_parent_synthetic(tmp_path):
    out = _run(["train_parent.py", "--synthetic", "4", "--epochs", "5", "--n-ave-grad", "2", "--height", "40", "--width", "56"], tmp_path)
    assert "Loss 4:" in out and "***Testing *** Loss 4" in out
