"""The checkers themselves: CPU restatement (oracle/sqoa_oracle.c) and, when built,
the compiled reference (oracle/_ref) against the committed golden vectors that
oracle/make_golden.py captured from /root/reference/seqoia.h."""
import numpy as np
import pytest

from oracle import pyoracle as po


def codecs():
    out = [po.restatement()]
    if po.reference() is not None:
        out.append(po.reference())
    return out


@pytest.mark.parametrize("codec", codecs(), ids=lambda c: c.kind)
def test_encode_kats(codec, kat_encode):
    for c in kat_encode:
        px = np.frombuffer(bytes.fromhex(c["pixels"]), dtype=np.uint8)
        for mode, key in ((0, "sqoa"), (1, "qoi")):
            got = codec.encode(px, c["w"], c["h"], c["channels"], mode, c["colorspace"])
            want = None if c[key] is None else bytes.fromhex(c[key])
            assert got == want, (c["name"], key)


@pytest.mark.parametrize("codec", codecs(), ids=lambda c: c.kind)
def test_decode_kats(codec, kat_decode):
    for c in kat_decode:
        out, d = codec.decode(bytes.fromhex(c["stream"]), c["channels"])
        if c["pixels"] is None:
            assert out is None, c["name"]
        else:
            assert out is not None and out.tobytes() == bytes.fromhex(c["pixels"]), c["name"]
            assert [d.width, d.height, d.channels, d.colorspace, d.qoi_compat] == c["desc"], c["name"]


def test_decode_rejects_negative_channels():
    # reference UB (seqoia.h:664 only rejects > 4); the restatement returns NULL
    px = po.synth(4, 4, 4)
    enc = po.restatement().encode(px, 4, 4, 4)
    assert po.restatement().decode(enc, -1)[0] is None


def test_synth_digests(synth_digests):
    """Generator and encoder pinned by FNV digests of reference output."""
    orc = po.restatement()
    for c in synth_digests["images"]:
        px = po.synth(c["w"], c["h"], c["channels"], c["cls"], c["seed"], c["img"])
        assert f"{po.fnv1a64(px):016x}" == c["raw_fnv"], c
        s = orc.encode(px, c["w"], c["h"], c["channels"], 0)
        q = orc.encode(px, c["w"], c["h"], c["channels"], 1)
        assert (len(s), f"{po.fnv1a64(s):016x}") == (c["sqoa_len"], c["sqoa_fnv"]), c
        assert (len(q), f"{po.fnv1a64(q):016x}") == (c["qoi_len"], c["qoi_fnv"]), c
        # round trip (sqoabench.c:448-455) at native and forced channel counts
        for enc in (s, q):
            out, d = orc.decode(enc, 0)
            assert out.tobytes() == px.tobytes()
            out4, _ = orc.decode(enc, 4)
            assert out4.reshape(-1, 4)[:, : c["channels"]].tobytes() == px.tobytes()


def test_synth_totals(synth_digests):
    orc = po.restatement()
    t = synth_digests["totals"]["icon_512_rgb_0_31"]
    tot = [0, 0]
    for img in range(t["count"]):
        px = po.synth(t["w"], t["h"], t["channels"], t["cls"], t["seed"], img)
        tot[0] += len(orc.encode(px, t["w"], t["h"], t["channels"], 0))
        tot[1] += len(orc.encode(px, t["w"], t["h"], t["channels"], 1))
    assert tot == [t["sqoa_total"], t["qoi_total"]]


@pytest.mark.skipif(po.reference() is None, reason="compiled reference not built")
def test_restatement_equals_reference_on_random_streams():
    """Differential fuzz (successor of sqoafuzz.c:20-32): mutated encoder output and random bodies."""
    orc, ref = po.restatement(), po.reference()
    rng = np.random.default_rng(1234)
    for k in range(400):
        w, h, ch = int(rng.integers(1, 48)), int(rng.integers(1, 12)), int(rng.integers(1, 7))
        px = rng.integers(0, 256, size=w * h * (ch if ch < 5 else ch - 2), dtype=np.uint8)
        if k % 3:
            px = (px // 64) * 64  # few colours: runs and index hits
        for mode in (0, 1):
            a, b = orc.encode(px, w, h, ch, mode), ref.encode(px, w, h, ch, mode)
            assert a == b
            if a is None:
                continue
            m = bytearray(a)
            for _ in range(int(rng.integers(0, 6))):
                m[int(rng.integers(12, len(m)))] = int(rng.integers(0, 256))
            for oc in (0, 1, 2, 3, 4):
                (oa, da), (ob, db) = orc.decode(m, oc), ref.decode(m, oc)
                assert (oa is None) == (ob is None)
                if oa is not None:
                    assert oa.tobytes() == ob.tobytes()
                    assert bytes(da) == bytes(db)


def test_headline_digest_table_is_consistent():
    """tests/golden/headline_digests.json (all 1024 streams of the headline batch, both wire formats, from the compiled
    reference) agrees with the independent fixtures in synth_digests.json and with the restatement on a sample."""
    from conftest import load_golden

    g = load_golden("headline_digests.json")
    d = load_golden("synth_digests.json")
    rows = g["rows"]
    assert len(rows) == g["count"] == 1024
    for c in d["images"]:
        if (c["cls"], c["w"], c["h"], c["channels"], c["seed"]) == ("mixed", 1024, 1024, 4, 1):
            assert rows[c["img"]] == [c["sqoa_len"], c["sqoa_fnv"], c["qoi_len"], c["qoi_fnv"]]
    t = d["totals"]["mixed_1024_rgba_0_15"]
    assert sum(r[0] for r in rows[:16]) == t["sqoa_total"] and sum(r[2] for r in rows[:16]) == t["qoi_total"]
    orc = po.restatement()
    for img in (7, 600):
        px = po.synth(1024, 1024, 4, "mixed", 1, img)
        s, q = orc.encode(px, 1024, 1024, 4, 0), orc.encode(px, 1024, 1024, 4, 1)
        assert rows[img] == [len(s), f"{po.fnv1a64(s):016x}", len(q), f"{po.fnv1a64(q):016x}"]


def test_c4_and_c5_digest_tables_against_the_restatement():
    """tests/golden/c4_digests.npz / c5_digests.npz (every stream of BASELINE configs[3] and configs[4], from the compiled reference:
    oracle/make_golden.py --c4 --c5): shape and totals, and the CPU restatement on a seeded sample of both -- the icon sets at every
    size in both wire formats, and suite images of every directory but the screenshot-sized ones."""
    import os

    from conftest import GOLDEN
    from seqoia_amd import suite

    orc = po.restatement()
    c4 = np.load(os.path.join(GOLDEN, "c4_digests.npz"))
    assert c4["sizes"].tolist() == [64, 128, 256, 512] and c4["lens"].shape == (4, 2, 4096) and c4["fnv"].shape == (4, 2, 4096)
    rng = np.random.default_rng(4)
    for si, size in enumerate(c4["sizes"].tolist()):
        for img in rng.integers(0, 4096, 6).tolist():
            px = po.synth(size, size, 3, "icon", 1, img)
            for mode in (0, 1):
                s = orc.encode(px, size, size, 3, mode)
                assert (len(s), po.fnv1a64(s)) == (int(c4["lens"][si, mode, img]), int(c4["fnv"][si, mode, img])), (size, img, mode)
    c5 = np.load(os.path.join(GOLDEN, "c5_digests.npz"))
    imgs = suite.images(suite.C5_IMAGES)
    assert c5["lens"].shape == (65536,) and c5["fnv"].shape == (65536,) and int(c5["count"]) == len(imgs)
    seen = {}
    for k in rng.permutation(len(imgs)).tolist():
        im = imgs[k]
        if im.pixels > 1_500_000 or seen.get(im.directory, 0) >= 2:
            continue
        seen[im.directory] = seen.get(im.directory, 0) + 1
        s = orc.encode(po.synth(im.width, im.height, im.channels, im.cls, 1, im.index), im.width, im.height, im.channels, 0)
        assert (len(s), po.fnv1a64(s)) == (int(c5["lens"][k]), int(c5["fnv"][k])), im
        if len(seen) >= 11 and all(v >= 2 for v in seen.values()):
            break
    assert len(seen) >= 11


def test_lds_conflict_model_reproduces_its_committed_table():
    """tools/lds_conflict_model.py (the balanced decoder's LDS bank conflicts by access class, on real streams) is an argument
    the design rests on: it has to keep running, and to keep saying what profiles/r06_lds_conflict_model.txt says -- class mixed in the low
    200s of conflict cycles a chunk with the transition table's look-ups the largest share, class photo about twice that."""
    import os
    import re
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "lds_conflict_model.py"), "1"], capture_output=True, text=True, check=True).stdout
    sums = [float(m.group(1)) for m in re.finditer(r"sum \(per chunk and wave\)\s+([0-9.]+)", out)]
    ftab = [float(m.group(1)) for m in re.finditer(r"ftab \(2 passes x 16 reads\)\s+([0-9.]+)", out)]
    assert len(sums) == 2 and len(ftab) == 2, out
    assert 180 < sums[0] < 260 and 400 < sums[1] < 520, sums          # class mixed, class photo (the counters: 254 on the headline batch)
    assert ftab[0] > 0.4 * sums[0], (ftab, sums)                      # data-dependent look-ups are the largest share on class mixed
