"""Forty rounds of tools/soak.py in the GPU suite (or 60 seconds of it, whichever ends first -- forty rounds take about fifteen): plans
of random image lists (1x1 .. 900x700, 3 / 4 channels, the three synthetic classes, palette images with long runs, noise with one very
long run; both wire formats) under every forcing of the routes (strips / one workgroup per image, three passes / chained, the wave decoder
at 8 / 12 bytes per lane, with the position-major store path, the balanced kernel), every stream and every decoded image against the
compiled reference.  A process of its own: the soak sets the route variables in its environment."""
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_soak_cut():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "soak.py"), "40", "3003", "60"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    m = re.search(r"soak ok: (\d+) decoded images and their streams identical to the reference \((\d+) rounds", r.stdout)
    assert m, r.stdout[-500:]
    # every round is a plan of 1 .. 700 images x 2 wire formats x 6 route forcings x 3 output channel counts, each image compared: forty
    # rounds of this seed are about 200 000 comparisons; a run that checked far fewer stopped early or skipped routes
    assert int(m.group(2)) >= 20 and int(m.group(1)) >= 50000, r.stdout[-500:]
