"""A 30-second cut of tools/soak.py in the GPU suite: plans of random image lists (1x1 .. 900x700, 3 / 4 channels, the three
synthetic classes, palette images with long runs, noise with one very long run; both wire formats) under every forcing of
the routes (strips / one workgroup per image, three passes / chained, the wave decoder at 8 / 12 bytes per lane and with the
position-major store path), every stream and every decoded image against the compiled reference.  A process of its own: the soak sets the route variables in its environment."""
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_soak_cut():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "soak.py"), "40", "3003", "40"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    m = re.search(r"soak ok: (\d+) decoded images", r.stdout)
    assert m and int(m.group(1)) >= 100, r.stdout[-500:]
