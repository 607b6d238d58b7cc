"""-m gpu: the end-to-end training example learns (accuracy and explanation ROC-AUC), single process."""
import os
import re
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_train_ba2motifs_example_learns():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "examples", "train_ba2motifs.py"), "--graphs", "600", "--epochs", "25"],
                         capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    last = [l for l in out.stdout.splitlines() if l.startswith("epoch")][-1]
    acc = float(re.search(r"test acc ([0-9.]+)", last).group(1))
    auc = float(re.search(r"ROC-AUC vs motif edges ([0-9.]+)", last).group(1))
    assert acc >= 0.9 and auc >= 0.8, last
