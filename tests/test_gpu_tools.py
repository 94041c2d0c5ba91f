"""The C harnesses (plain-C callers of the drop-in header + library): sqoaconv-style file round trip
(sqoa_read / sqoa_write, seqoia.h:336/:350, call shapes of sqoaconv.c:65/:84-90) and the sqoabench-style
benchmark with its round-trip verify (sqoabench.c:448-455)."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "tools", "bin")


def pam(path, px, w, h, ch):
    t = {1: "GRAYSCALE", 2: "GRAYSCALE_ALPHA", 3: "RGB", 4: "RGB_ALPHA"}[ch]
    with open(path, "wb") as f:
        f.write(f"P7\nWIDTH {w}\nHEIGHT {h}\nDEPTH {ch}\nMAXVAL 255\nTUPLTYPE {t}\nENDHDR\n".encode())
        f.write(px.tobytes())


def test_sqoaconv_file_roundtrip(tmp_path):
    from oracle import pyoracle as po

    if not os.path.exists(os.path.join(BIN, "sqoaconv_mi355x")):
        subprocess.run(["make", "-s", "-C", ROOT, "tools"], check=True)
    conv = os.path.join(BIN, "sqoaconv_mi355x")
    for ch, cls in ((4, "mixed"), (3, "icon"), (1, "photo")):
        w, h = 200, 120
        px = po.synth(w, h, 4, cls, 2, 0).reshape(-1, 4)[:, :ch].copy() if ch != 1 else po.synth(w, h, 4, cls, 2, 0).reshape(-1, 4)[:, 1:2].copy()
        src = tmp_path / f"in{ch}.pam"
        pam(src, px, w, h, ch)
        sq, qoi, back = tmp_path / f"a{ch}.sqoa", tmp_path / f"a{ch}.qoi", tmp_path / f"b{ch}.pam"
        assert subprocess.run([conv, str(src), str(sq)]).returncode == 0
        assert sq.read_bytes() == po.best().encode(px.reshape(-1), w, h, ch, 0)       # byte-identical .sqoa file
        if ch >= 3:
            assert subprocess.run([conv, str(sq), str(qoi)]).returncode == 0           # .sqoa -> .qoi
            assert qoi.read_bytes() == po.best().encode(px.reshape(-1), w, h, ch, 1)
            assert subprocess.run([conv, str(qoi), str(back)]).returncode == 0          # .qoi -> raw
        else:
            assert subprocess.run([conv, str(sq), str(qoi)]).returncode != 0           # mono cannot be QOI (seqoia.h:477-480)
            assert subprocess.run([conv, str(sq), str(back)]).returncode == 0
        assert back.read_bytes().split(b"ENDHDR\n", 1)[1] == px.tobytes()
    assert subprocess.run([conv, str(tmp_path / "missing.sqoa"), str(tmp_path / "x.pam")]).returncode != 0


def test_sqoabench_synth_runs_and_verifies():
    bench = os.path.join(BIN, "sqoabench_mi355x")
    r = subprocess.run([bench, "2", "--synth", "mixed", "256", "256", "4", "16"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert "sqoa-batch:" in r.stdout and "decode mpps" in r.stdout


def test_sqoabatch_directory_roundtrip(tmp_path):
    """Directory front-end on the batch C-ABI + sqoa_hip_pack: every file byte-identical to the reference's."""
    from oracle import pyoracle as po

    if not os.path.exists(os.path.join(BIN, "sqoabatch_mi355x")):
        subprocess.run(["make", "-s", "-C", ROOT, "tools"], check=True)
    batch = os.path.join(BIN, "sqoabatch_mi355x")
    src, enc, qoi, back = tmp_path / "src", tmp_path / "enc", tmp_path / "qoi", tmp_path / "back"
    for d in (src, enc, qoi, back):
        d.mkdir()
    imgs = {}
    for k, (w, h, ch, cls) in enumerate([(64, 64, 4, "icon"), (300, 200, 3, "photo"), (513, 7, 4, "mixed"), (1024, 512, 4, "mixed"), (90, 90, 1, "photo")]):
        px = po.synth(w, h, 4, cls, 8, k).reshape(-1, 4)
        px = (px[:, :ch] if ch >= 3 else px[:, 1:2]).copy()
        imgs[f"img{k}"] = (px, w, h, ch)
        pam(src / f"img{k}.pam", px, w, h, ch)
    assert subprocess.run([batch, "encode", str(src), str(enc)]).returncode == 0
    for name, (px, w, h, ch) in imgs.items():
        assert (enc / f"{name}.sqoa").read_bytes() == po.best().encode(px.reshape(-1), w, h, ch, 0), name
    assert subprocess.run([batch, "decode", str(enc), str(back)]).returncode == 0
    for name, (px, w, h, ch) in imgs.items():
        assert (back / f"{name}.pam").read_bytes().split(b"ENDHDR\n", 1)[1] == px.tobytes(), name
    # several sub-batches (1 MB of input each): reader thread, GPU and writer thread overlap; same files
    enc2, back2 = tmp_path / "enc2", tmp_path / "back2"
    enc2.mkdir(); back2.mkdir()
    r = subprocess.run([batch, "encode", str(src), str(enc2), "--batch-mb", "1"], capture_output=True, text=True)
    assert r.returncode == 0 and "sub-batch" in r.stdout and " in 1 sub-batch" not in r.stdout, r.stdout + r.stderr
    for name in imgs:
        assert (enc2 / f"{name}.sqoa").read_bytes() == (enc / f"{name}.sqoa").read_bytes(), name
    assert subprocess.run([batch, "decode", str(enc2), str(back2), "--batch-mb", "1"]).returncode == 0
    for name in imgs:
        assert (back2 / f"{name}.pam").read_bytes() == (back / f"{name}.pam").read_bytes(), name
    r = subprocess.run([batch, "encode", str(src), str(qoi), "--qoi"])  # the mono image cannot be QOI: reported, others written
    assert r.returncode == 1
    for name, (px, w, h, ch) in imgs.items():
        if ch >= 3:
            assert (qoi / f"{name}.qoi").read_bytes() == po.best().encode(px.reshape(-1), w, h, ch, 1), name
        else:
            assert not (qoi / f"{name}.qoi").exists()


def test_bench_rank_path_with_two_ranks_on_the_card():
    """The real rank path of bench.py with WORLD_SIZE 2 on this box's one card (`--rehearsal`: the line says so; gloo for the timing
    barrier / MAX and the gather of the per-image lengths): two rank processes, each with its own plans on the device, the barrier,
    the all-reduce of the step time, the gathered lengths adding up to the bytes the ranks wrote.  (The CPU tests of tests/test_shard.py
    run the launcher with --dry-run: no GPU work under WORLD_SIZE > 1 anywhere else.)"""
    import json

    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--rehearsal", "--backend", "gloo", "--images", "8", "--steps", "2",
                        "--warmup", "1", "--no-secondary", "--no-cpu"], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stdout[-1000:] + r.stderr[-3000:]
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert out["n_gpus"] == 2 and "rehearsal" in out and out["scaling"] == "weak"
    assert out["config"]["encoded_bytes_check"] == out["config"]["encoded_bytes_total"] > 0
    assert out["value"] > 0 and out["steps"] == 2


def test_lds_applies_the_lanes_of_an_atomic_in_ascending_order():
    """The QOI-compat encoder's L8 probes (sqoa_encode.hip, SQOA_ENC_Q_RTN) take "the nearest lower lane of my row on my index slot" from ONE
    `ds_max_rtn_u32` per row, and let the row's last writer of a slot win by plain `ds_write_b32`: both rest on the order in which gfx950's LDS
    applies the lanes of one instruction that meet on an address -- ascending lane number.  tools/ubench/lds_atomic_order.hip puts 8 x 10^8
    atomics and as many plain writes through random, strided, clustered and single-address patterns, eight waves to a workgroup's LDS, and
    counts every lane whose returned value is not the stamp of the highest lower lane on its address, and every address whose final value
    is not the highest lane's: all zero, or the encoder's output is not the reference's (which the parity tests would show as well)."""
    exe = os.path.join(BIN, "lds_atomic_order")
    assert os.path.exists(exe), "tools/bin/lds_atomic_order is built by `make tools`"
    out = subprocess.run([exe], check=True, capture_output=True, text=True, timeout=300).stdout
    lines = [l for l in out.splitlines() if l.startswith("pattern")]
    assert len(lines) == 24, out
    for l in lines:
        assert "out of order 0," in l and l.rstrip().endswith("highest lane 0"), l
