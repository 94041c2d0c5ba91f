"""GPU parity: the HIP decoder behind the C-ABI vs golden vectors and the oracle."""
import numpy as np
import os

import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def sq():
    import torch

    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a HIP device; refusing to fall back to anything else")
    import seqoia_amd

    return seqoia_amd


@pytest.fixture(scope="module")
def orc():
    """The checker.  On the GPU box it must be the REAL reference (oracle/_ref/libsqoa_ref.so, /root/reference/seqoia.h compiled
    by oracle/Makefile in the build container; the .so travels with the snapshot): a run that silently fell back to the
    restatement would still be a parity test, but not the one the documents claim."""
    from oracle import pyoracle as po

    assert po.best().kind == "reference", "oracle/_ref/libsqoa_ref.so did not travel: GPU parity must be checked against the compiled reference"
    return po


@pytest.fixture(params=[("0", "0", "0", "0"), ("0", "1", "0", "0"), ("0", "1", "1", "0"), ("0", "1", "0", "1"), ("1", "0", "0", "0"), ("1", "1", "0", "0")],
                ids=["chained", "chained-wide", "chained-wide-sweep", "chained-balanced", "threepass", "threepass-wide"])
def route(request, monkeypatch):
    """Every route of the SQOA wave decoder (the plan reads SQOA_HIP_MULTIPASS / SQOA_HIP_DEC_WIDE / SQOA_HIP_DEC_SWEEP when it is
    filled): one workgroup per stream with the chunks chained through LDS -- eight stream bytes per lane, or twelve (what the plan
    picks by itself when every stream goes to four channels; forced here for every channel count), or twelve with the position-major
    store path for chunks full of runs (the plan's choice for streams of under a byte per pixel; forced here for everything), or the
    balanced kernel (sqoa_decode_bal.hip: sixteen bytes per lane in the parse, the ops dealt out evenly over the lanes; the plan's
    choice for large streams to four channels, forced here for everything) -- and
    the three chip-wide passes the plan picks by itself for few large streams, their tables counted in 512-byte chunks or (plans
    without a QOI-compat stream among them, to four channels; forced here wherever no such stream is) in 768-byte ones."""
    monkeypatch.setenv("SQOA_HIP_MULTIPASS", request.param[0])
    monkeypatch.setenv("SQOA_HIP_DEC_WIDE", request.param[1])
    monkeypatch.setenv("SQOA_HIP_DEC_SWEEP", request.param[2])
    monkeypatch.setenv("SQOA_HIP_DEC_BAL", request.param[3])
    return request.param[0]


def test_legacy_decode_kats(sq, kat_decode):
    """sqoa_decode on every crafted stream captured from the compiled reference: REF ops, ALPHA
    suffixes, leading ALPHA as run, mono / 128-entry index, exhausted streams, header rejects."""
    for c in kat_decode:
        out, d = sq.decode(bytes.fromhex(c["stream"]), c["channels"])
        if c["pixels"] is None:
            assert out is None, c["name"]
        else:
            assert out is not None, c["name"]
            assert out.tobytes() == bytes.fromhex(c["pixels"]), c["name"]
            assert [d.width, d.height, d.channels, d.colorspace, d.qoi_compat] == c["desc"], c["name"]


def test_legacy_decode_rejects(sq, orc):
    enc = orc.restatement().encode(orc.synth(4, 4, 4), 4, 4, 4)
    assert sq.decode(enc, 5)[0] is None
    assert sq.decode(enc, -1)[0] is None  # reference UB (seqoia.h:664); NULL here
    assert sq.decode(enc[:21], 4)[0] is None


def test_decode_encoder_kats_all_channel_args(sq, orc, kat_encode):
    """Decode every golden encoder output at channels = 0,1,2,3,4 and compare with the oracle."""
    o = orc.best()
    for c in kat_encode:
        for key in ("sqoa", "qoi"):
            if c[key] is None:
                continue
            s = bytes.fromhex(c[key])
            for oc in (0, 1, 2, 3, 4):
                want, wd = o.decode(s, oc)
                got, gd = sq.decode(s, oc)
                assert got is not None and got.tobytes() == want.tobytes(), (c["name"], key, oc)
                assert bytes(gd)[:11] == bytes(wd)[:11]


@pytest.mark.parametrize("mode", [0, 1])
@pytest.mark.parametrize("cls,w,h,ch,n", [("mixed", 256, 256, 4, 8), ("icon", 128, 128, 3, 16), ("photo", 200, 150, 3, 4),
                                           ("mixed", 1024, 1024, 4, 3), ("icon", 512, 512, 3, 4), ("mixed", 333, 77, 4, 5),
                                           ("icon", 64, 64, 4, 64), ("mixed", 1, 1, 4, 3), ("mixed", 511, 1, 3, 2), ("mixed", 513, 3, 4, 2)])
def test_batch_roundtrip_and_oracle(sq, orc, mode, cls, w, h, ch, n):
    """sqoabench.c:448-455 on a device-resident batch: decode(encode(px)) == px at the native channel
    count and at 4 channels (the benchmark's timed decode, sqoabench.c:492), plus oracle equality."""
    import torch

    px = torch.empty((n, h, w, ch), dtype=torch.uint8, device="cuda")
    sq.synth(px, cls, seed=1, img0=0)
    enc = sq.EncodePlan(px, qoi_compat=mode)
    enc.launch()
    res = enc.results()
    assert (res["status"] == 0).all()
    for oc in sorted({ch, 4}):
        dec = sq.DecodePlan(enc.out, res["len"], w, h, oc)
        dec.launch()
        dres = dec.results()
        assert (dres["status"] == 0).all() and (dres["len"] == w * h * oc).all()
        assert (dres["width"] == w).all() and (dres["height"] == h).all() and (dres["channels"] == ch).all()
        assert (dres["qoi_compat"] == mode).all()
        if oc == ch:
            assert torch.equal(dec.out, px)
        else:
            assert torch.equal(dec.out[..., :ch], px) and bool((dec.out[..., 3] == 255).all())
        dec.close()
    enc.close()


def test_batch_decode_mutants_match_oracle(sq, orc, route):
    """Differential fuzz on the batch path (successor of sqoafuzz.c:20-32): mutated encoder output must
    decode exactly as the reference decodes it, or be refused exactly when the reference refuses."""
    import torch

    o = orc.best()
    rng = np.random.default_rng(4321)
    streams, wants = [], []
    w, h = 48, 20
    for k in range(96):
        ch = (3, 4)[k & 1]
        px = orc.synth(w, h, ch, ["mixed", "icon", "photo"][k % 3], seed=11, img=k)
        m = bytearray(o.encode(px, w, h, ch, (k >> 1) & 1))
        for _ in range(k % 7):
            m[int(rng.integers(15, len(m) - 8))] = int(rng.integers(0, 256))
        streams.append(bytes(m))
        wants.append(o.decode(bytes(m), 4)[0])
    slot = max(len(s) for s in streams) + 32
    buf = np.zeros((len(streams), slot), dtype=np.uint8)
    for i, s in enumerate(streams):
        buf[i, : len(s)] = np.frombuffer(s, dtype=np.uint8)
    t = torch.from_numpy(buf).cuda()
    dec = sq.DecodePlan(t, [len(s) for s in streams], w, h, 4)
    dec.launch()
    res = dec.results()
    out = dec.out.cpu().numpy()
    for i, want in enumerate(wants):
        if want is None:
            assert res["status"][i] != 0 and res["len"][i] == 0, i
        else:
            assert res["status"][i] == 0, i
            assert out[i].tobytes() == want.tobytes(), i
    dec.close()


def test_wave_decoder_chunk_boundaries(sq, orc, route):
    """Crafted multi-chunk SQOA streams (the wave decoder cuts the stream every 512 bytes, 8 bytes per lane -- or every 768, 12 per
    lane: the `route` fixture):
    every alignment of 1/2/3/4/5/6-byte ops against lane and chunk boundaries, ALPHA suffix as first byte of a
    lane / chunk, op bytes that look like op starts, REF op deep in the stream (hand-over to the serial kernel
    while other waves wait), stream exhausted in the middle of an op."""
    import torch

    o = orc.best()
    rng = np.random.default_rng(77)

    def hdr(w, h, ch=4):
        return b"Sqoa" + w.to_bytes(4, "big") + h.to_bytes(4, "big") + bytes([ch, 0, 0x31])

    def body(n_ops, kinds):
        out = bytearray()
        for _ in range(n_ops):
            k = kinds[int(rng.integers(0, len(kinds)))]
            if k == "luma":
                out += bytes([0x80 | int(rng.integers(0, 64)), int(rng.integers(0, 256))])
            elif k == "luma_a":
                out += bytes([0x80 | int(rng.integers(0, 64)), int(rng.integers(0, 256)), 0x60 | int(rng.integers(0, 32))])
            elif k == "rgb":
                out += bytes([0xfe]) + bytes(rng.integers(0, 256, 3).tolist())
            elif k == "rgb_a":
                out += bytes([0xfe]) + bytes(rng.integers(0, 256, 3).tolist()) + bytes([0x60 | int(rng.integers(0, 32))])
            elif k == "rgba":
                out += bytes([0xff]) + bytes(rng.integers(0, 256, 4).tolist())
            elif k == "rgba_a":
                out += bytes([0xff]) + bytes(rng.integers(0, 256, 4).tolist()) + bytes([0x60 | int(rng.integers(0, 32))])
            elif k == "run":
                out += bytes([0xc0 | int(rng.integers(0, 61))])
            elif k == "run_a":
                out += bytes([0xc0 | int(rng.integers(0, 61)), 0x60 | int(rng.integers(0, 32))])
            elif k == "bigrun":
                out += b"\xfd"
            elif k == "alpha_alpha":  # suffix, then a second 0x60..0x7f byte: a RUN op (seqoia.h:773-775)
                out += bytes([0x80 | int(rng.integers(0, 64)), int(rng.integers(0, 256)), 0x65, 0x6a])
        return bytes(out)

    END = bytes(7) + b"\x01"
    streams = []
    # (["run", "run_a"]: a RUN does not take the RUN behind it into its walk iteration when that one carries an ALPHA suffix -- one
    # byte, then two: eight ops start in twelve bytes, as many as a lane has record registers for)
    mixes = [["luma"], ["luma", "luma_a"], ["rgb", "luma"], ["rgba", "luma_a", "run"], ["rgba_a", "rgb_a", "run_a", "alpha_alpha"],
             ["run", "bigrun", "luma"], ["run", "run_a"], ["bigrun"], ["luma", "luma_a", "rgb", "rgb_a", "rgba", "rgba_a", "run", "run_a", "bigrun", "alpha_alpha"]]
    for shift in range(13):  # leading run bytes move every later op by one byte
        for mix in mixes:
            streams.append(hdr(96, 64) + b"\xc0" * shift + body(700, mix) + END)
    full = mixes[-1]
    streams.append(hdr(96, 64) + body(500, full) + b"\x45" + body(300, full) + END)        # REF op in a later chunk
    streams.append(hdr(96, 64) + body(400, ["luma"]) + b"\x1f" + body(10, full) + END)     # REF reaching back 31 bytes
    streams.append((hdr(96, 64) + body(300, full) + END)[:-9])                             # exhausted: cut inside the tail
    streams.append(hdr(96, 64) + body(300, full) + b"\xff\x01\x02")                        # exhausted inside an RGBA op
    streams.append(hdr(96, 64) + b"\x6a" + body(300, full) + END)                          # leading ALPHA byte is a run
    streams.append(hdr(300, 300) + body(40, ["bigrun", "run"]) + END)                      # runs overshoot the image
    streams.append(hdr(96, 64, 3) + body(700, ["luma", "rgb", "run", "luma_a"]) + END)     # 3-channel file, 4-channel output
    # long streams: several workgroups of 16 chunks each on the three-pass route (entry state, position and pixel of a chunk
    # come from the table the scan kernels fill, not from the neighbour wave)
    lit = ["luma", "luma_a", "rgb", "rgb_a", "rgba", "rgba_a", "alpha_alpha"]
    for shift in range(4):
        streams.append(hdr(200, 160) + b"\xc0" * shift + body(9000, lit + ["run"]) + END)
    streams.append(hdr(200, 160) + body(6000, lit) + b"\x45" + body(3000, lit) + END)       # REF op in the third workgroup
    streams.append((hdr(200, 160) + body(9000, lit) + END)[:-13])                          # exhausted inside the last ops
    streams.append(hdr(200, 160) + body(4000, lit) + END)                                  # stream ends early: the last pixel repeats
    streams.append(hdr(200, 160) + body(3000, lit) + body(6000, ["bigrun", "run", "luma"]) + END)  # runs overshoot from the 2nd workgroup on
    streams.append(hdr(200, 160, 3) + body(9000, ["luma", "rgb", "run", "luma_a"]) + END)
    streams.append(hdr(96, 64) + body(9000, lit + ["run"]) + END)                          # more ops than pixels: the later workgroups store nothing
    wants = [o.decode(s, 4) for s in streams]
    slot = max(len(s) for s in streams) + 32
    buf = np.zeros((len(streams), slot), dtype=np.uint8)
    for i, s in enumerate(streams):
        buf[i, : len(s)] = np.frombuffer(s, dtype=np.uint8)
    t = torch.from_numpy(buf).cuda()
    # every stream has its own geometry: one plan per geometry group would do; here one DecodePlan per stream size class
    groups = {}
    for i, s in enumerate(streams):
        w, h = int.from_bytes(s[4:8], "big"), int.from_bytes(s[8:12], "big")
        groups.setdefault((w, h), []).append(i)
    for (w, h), idx in groups.items():
        sub = t[idx].contiguous()
        dec = sq.DecodePlan(sub, [len(streams[i]) for i in idx], w, h, 4)
        dec.launch()
        res = dec.results()
        out = dec.out.cpu().numpy()
        for k, i in enumerate(idx):
            want, wd = wants[i]
            if want is None:
                assert res["status"][k] != 0, i
            else:
                assert res["status"][k] == 0, (i, int(res["status"][k]))
                assert out[k].tobytes() == want.tobytes(), i
        dec.close()


def test_full_headline_batch_roundtrip(sq, orc, synth_digests):
    """BASELINE configs 2+3 at full size: 1024 x 1024x1024 RGBA, BOTH wire formats.  Every one of the 2 x 1024 streams
    is pinned to the compiled reference (seqoia.h:456-650) by length and FNV-1a-64 (tests/golden/headline_digests.json,
    written by oracle/make_golden.py --headline), the digests computed on the device; plus every status OK,
    decode(encode(px)) == px for the whole batch (sqoabench.c:448-455), and byte-for-byte equality with the oracle on a sample."""
    import torch

    from conftest import load_golden

    gold = load_golden("headline_digests.json")
    n = gold["count"]
    assert (n, gold["w"], gold["h"], gold["channels"], gold["seed"]) == (1024, 1024, 1024, 4, 1)
    px = torch.empty((n, 1024, 1024, 4), dtype=torch.uint8, device="cuda")
    sq.synth(px, "mixed", seed=1, img0=0)
    o = orc.best()
    for mode in (0, 1):
        enc = sq.EncodePlan(px, qoi_compat=mode)
        enc.launch()
        res = enc.results()
        assert (res["status"] == 0).all()
        want_len = [r[2 * mode] for r in gold["rows"]]
        want_fnv = [int(r[2 * mode + 1], 16) for r in gold["rows"]]
        assert res["len"].tolist() == want_len
        base = enc.out.data_ptr()
        got_fnv = sq.fnv1a64([base + i * enc.slot for i in range(n)], want_len)
        bad = [i for i in range(n) if got_fnv[i] != want_fnv[i]]
        assert not bad, (mode, bad[:8])
        if mode == 0:
            assert int(res["len"][:16].sum()) == synth_digests["totals"]["mixed_1024_rgba_0_15"]["sqoa_total"]
        dec = sq.DecodePlan(enc.out, res["len"], 1024, 1024, 4)
        dec.launch()
        dres = dec.results()
        assert (dres["status"] == 0).all() and (dres["len"] == 4 << 20).all()
        assert torch.equal(dec.out, px)
        for k in (5, 1023):
            assert enc.streams([k])[0] == o.encode(orc.synth(1024, 1024, 4, "mixed", 1, k), 1024, 1024, 4, mode)
        dec.close()
        enc.close()
        del dec, enc


@pytest.mark.parametrize("rank", list(range(8)))
def test_c5_rank_of_8_shard(sq, orc, rank):
    """BASELINE configs[4] as EVERY one of its 8 ranks sees it: the qoi-suite-shaped list of 65 536 images (seqoia_amd/suite.py), LPT-sharded
    on w*h, the rank's 8 192 images in ONE encode plan and ONE decode plan (the workload class of `bench.py --config c5 --gpus 8`): every
    status OK, every image round-trips exactly (sqoabench.c:448-455, one result per image as in sqoabench.c:575-628), and EVERY stream is
    pinned to the compiled reference's (seqoia.h:456-650) by length and FNV-1a-64 -- tests/golden/c5_digests.npz, all 65 536 images, written
    by oracle/make_golden.py --c5; the digests here computed on the device -- plus one stream per directory byte for byte."""
    import numpy as np
    import torch

    import bench
    from conftest import GOLDEN
    from seqoia_amd import suite

    gold = np.load(os.path.join(GOLDEN, "c5_digests.npz"))
    assert int(gold["count"]) == suite.C5_IMAGES and int(gold["seed"]) == 1
    shard = suite.shard(suite.C5_IMAGES, 8, rank)
    assert len(shard) > 8000
    wl = bench.Suite(sq, torch, torch.device("cuda", 0), shard)  # asserts statuses and the round trip of every image
    assert wl.pixels == sum(im.pixels for im in shard)
    streams_at = wl.enc.offsets
    idx = np.array([im.index for im in shard])
    assert (wl.lens == gold["lens"][idx].astype("int64")).all(), "stream lengths differ from the reference's"
    base = wl.enc.out.data_ptr()
    got = np.array(sq.fnv1a64([base + o for o in streams_at[:-1]], wl.lens.tolist()), dtype=np.uint64)
    bad = np.nonzero(got != gold["fnv"][idx])[0]
    assert bad.size == 0, [shard[i] for i in bad[:4]]
    o = orc.best()
    seen = set()
    for i, im in enumerate(shard):
        if im.directory in seen or im.pixels > 2_000_000:
            continue
        seen.add(im.directory)
        got1 = bytes(wl.enc.out[streams_at[i] : streams_at[i] + int(wl.lens[i])].cpu().numpy())
        assert got1 == o.encode(orc.synth(im.width, im.height, im.channels, im.cls, 1, im.index), im.width, im.height, im.channels, 0), im
    assert len(seen) >= 12
    wl.step(0)  # the timed step of bench.py is repeatable: same lengths after a second encode + decode
    torch.cuda.synchronize()
    assert (wl.enc.results()["len"].astype("int64") == wl.lens).all()
    wl.close()


@pytest.mark.parametrize("size", [64, 128, 256, 512])
def test_c4_icon_set_at_full_size(sq, orc, size):
    """BASELINE configs[3] at its full size: 4 096 icon-style RGB images of 64^2, 128^2, 256^2 and 512^2 (RUN / BIGRUN dominated; the plan takes
    the wave decoder's position-major store shape for the large ones) in one encode plan and one decode plan per output channel count, in both
    wire formats: every status OK, EVERY stream pinned to the compiled reference's by length and FNV-1a-64 (tests/golden/c4_digests.npz: 4 sizes
    x 4 096 images x 2 formats, written by oracle/make_golden.py --c4; the digests here computed on the device), every image round-trips
    exactly to three channels and to four (sqoabench's timed decode, sqoabench.c:490-494), and 8 streams spread over the set byte for byte."""
    import numpy as np
    import torch

    from conftest import GOLDEN

    gold = np.load(os.path.join(GOLDEN, "c4_digests.npz"))
    si = gold["sizes"].tolist().index(size)
    n, w, h, ch = 4096, size, size, 3
    assert gold["lens"].shape == (4, 2, n) and int(gold["channels"]) == ch and int(gold["seed"]) == 1
    px = torch.empty((n, h, w, ch), dtype=torch.uint8, device="cuda")
    sq.synth(px, "icon", seed=1, img0=0)
    o = orc.best()
    sample = list(range(0, n, n // 8))
    for mode in (0, 1):
        enc = sq.EncodePlan(px, qoi_compat=mode)
        enc.launch()
        res = enc.results()
        assert (res["status"] == 0).all()
        assert (res["len"].astype("int64") == gold["lens"][si, mode].astype("int64")).all(), (size, mode, "stream lengths differ from the reference's")
        base = enc.out.data_ptr()
        got = np.array(sq.fnv1a64([base + i * enc.slot for i in range(n)], res["len"].tolist()), dtype=np.uint64)
        bad = np.nonzero(got != gold["fnv"][si, mode])[0]
        assert bad.size == 0, (size, mode, bad[:8].tolist())
        for k, g in zip(sample, enc.streams(sample)):
            assert g == o.encode(orc.synth(w, h, ch, "icon", 1, k), w, h, ch, mode), (mode, k)
        for oc in (3, 4):
            dec = sq.DecodePlan(enc.out, res["len"], w, h, oc)
            dec.launch()
            dres = dec.results()
            assert (dres["status"] == 0).all() and (dres["len"] == w * h * oc).all()
            if oc == 3:
                assert torch.equal(dec.out, px)
            else:
                assert torch.equal(dec.out[..., :3], px) and bool((dec.out[..., 3] == 255).all())
            dec.close()
        enc.close()


def test_results_wait_for_a_launch_on_the_callers_stream(sq, orc):
    """sqoa_hip_plan_launch(plan, user_stream) followed by sqoa_hip_plan_results with NO other synchronisation: results
    must wait for that launch (the plan records an event behind its kernels on whatever stream they went to)."""
    import torch

    n, w, h = 64, 1024, 1024
    px = torch.empty((n, h, w, 4), dtype=torch.uint8, device="cuda")
    sq.synth(px, "mixed", seed=5, img0=0)
    torch.cuda.synchronize()
    enc = sq.EncodePlan(px, qoi_compat=0)
    side = torch.cuda.Stream()
    want = None
    for rep in range(3):
        with torch.cuda.stream(side):
            filler = torch.empty(1 << 28, dtype=torch.uint8, device="cuda").fill_(rep)  # keeps the side stream busy in front of the launch
        enc.launch(side.cuda_stream)
        res = enc.results()  # no torch.cuda.synchronize() here on purpose
        assert (res["status"] == 0).all() and (res["len"] > 22).all(), rep
        if want is None:
            torch.cuda.synchronize()
            want = enc.results()["len"].copy()
        assert (res["len"] == want).all(), rep
        del filler
    o = orc.best()
    assert enc.streams([3])[0] == o.encode(orc.synth(w, h, 4, "mixed", 5, 3), w, h, 4, 0)
    enc.close()


def test_decode_to_unaligned_rgba_destination_and_shape_check(sq, orc, route):
    """A 4-channel destination that is not dword-aligned takes byte stores (every decoder); a stream whose header does
    not have the shape the job states is refused (SQOA_HIP_ST_BADARG) instead of being written with scrambled rows."""
    import torch

    from seqoia_amd import Desc, Job
    from seqoia_amd._lib import DECODE

    o = orc.best()
    w, h = 300, 200
    cases = []
    for mode, ch, cls in ((0, 4, "mixed"), (0, 3, "photo"), (1, 4, "mixed"), (0, 4, "icon")):
        px = orc.synth(w, h, ch, cls, 6, mode + ch)
        cases.append((o.encode(px, w, h, ch, mode), o))
    blob = b"".join(s + bytes(-len(s) % 16 + 16) for s, _ in cases)
    src = torch.frombuffer(bytearray(blob), dtype=torch.uint8).cuda()
    out = torch.zeros((len(cases), w * h * 4 + 8), dtype=torch.uint8, device="cuda")
    jobs, off = [], 0
    for k, (s, _) in enumerate(cases):
        jobs.append(Job(src.data_ptr() + off, out[k].data_ptr() + 1 + (k & 1) * 2, len(s), w * h * 4, Desc(w, h, 4, 0, 0), 0))
        off += len(s) + (-len(s) % 16 + 16)
    # same streams, but the caller expects another shape with the same pixel count
    jobs.append(Job(src.data_ptr(), out[0].data_ptr(), len(cases[0][0]), w * h * 4, Desc(h, w, 4, 0, 0), 0))
    plan = sq.Plan(DECODE, jobs)
    plan.launch()
    res = plan.results()
    outc = out.cpu().numpy()
    for k, (s, _) in enumerate(cases):
        want, _ = o.decode(s, 4)
        a = 1 + (k & 1) * 2
        assert res["status"][k] == 0, k
        assert outc[k, a : a + w * h * 4].tobytes() == want.tobytes(), k
        assert outc[k, :a].tobytes() == bytes(a) and outc[k, a + w * h * 4 :].tobytes() == bytes(8 - a), k  # nothing outside
    assert res["status"][len(cases)] == 1 and res["len"][len(cases)] == 0
    plan.close()


def test_plans_replay_from_a_hip_graph(sq, orc):
    """sqoa_hip_plan_launch does no allocation and no host sync, so an encode->decode step can be captured
    once into a HIP graph and replayed: new pixels written into the same buffers re-encode / re-decode correctly."""
    import torch

    n, w, h = 32, 256, 256
    px = torch.empty((n, h, w, 4), dtype=torch.uint8, device="cuda")
    sq.synth(px, "mixed", seed=1, img0=0)
    enc = sq.EncodePlan(px, qoi_compat=0)
    enc.launch()
    res = enc.results()
    dec = sq.DecodePlan(enc.out, res["len"], w, h, 4)  # stream lengths are job-table constants: keep the content class
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        enc.launch(side.cuda_stream)  # warm-up on the capture stream
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        enc.launch(side.cuda_stream)
    o = orc.best()
    for seed in (1, 2, 3):
        sq.synth(px, "mixed", seed=seed, img0=0)
        torch.cuda.synchronize()
        g.replay()
        torch.cuda.synchronize()
        r = enc.results()
        assert (r["status"] == 0).all()
        for k in (0, n - 1):
            want = o.encode(orc.synth(w, h, 4, "mixed", seed, k), w, h, 4, 0)
            assert bytes(enc.out[k, : int(r["len"][k])].cpu().numpy()) == want, (seed, k)
    dec.close()
    # the decode plan in a graph of its own, on both routes of the wave decoder (the three-pass route is five kernels and two
    # scan kernels per launch, all of them captured)
    # (and, from QOI-compat streams, the speculative route: passes 1 and 2, the speculative pass, its two checks, the stand-by kernels)
    import os
    sq.synth(px, "mixed", seed=1, img0=0)
    enc.launch()
    res = enc.results()
    encq = sq.EncodePlan(px, qoi_compat=1)
    encq.launch()
    resq = encq.results()
    assert (resq["status"] == 0).all()
    for knob, src, lens in (("0", enc.out, res["len"]), ("1", enc.out, res["len"]), ("1", encq.out, resq["len"]), ("0", encq.out, resq["len"])):
        os.environ["SQOA_HIP_MULTIPASS"] = knob
        try:
            dec = sq.DecodePlan(src, lens, w, h, 4)
        finally:
            del os.environ["SQOA_HIP_MULTIPASS"]
        with torch.cuda.stream(side):
            dec.launch(side.cuda_stream)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        g2 = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g2, stream=side):
            dec.launch(side.cuda_stream)
        for _ in range(2):
            dec.out.zero_()
            torch.cuda.synchronize()
            g2.replay()
            torch.cuda.synchronize()
            assert (dec.results()["status"] == 0).all() and torch.equal(dec.out, px), knob
        dec.close()
    enc.close()
    encq.close()


@pytest.mark.parametrize("mode", [0, 1])
def test_large_single_image(sq, orc, mode, route):
    """One screenshot-sized image (3840x2160, the largest class of SURVEY.md Appendix C's suite table): 16 200
    sub-tiles through one workgroup's chains, output positions beyond 2^24, both wire formats, encode and decode."""
    import torch

    w, h = 3840, 2160
    px = torch.empty((1, h, w, 4), dtype=torch.uint8, device="cuda")
    sq.synth(px, "mixed", seed=3, img0=5)
    enc = sq.EncodePlan(px, qoi_compat=mode)
    enc.launch()
    res = enc.results()
    assert res["status"][0] == 0
    want = orc.best().encode(orc.synth(w, h, 4, "mixed", 3, 5), w, h, 4, mode)
    assert enc.streams()[0] == want
    dec = sq.DecodePlan(enc.out, res["len"], w, h, 4)
    dec.launch()
    assert dec.results()["status"][0] == 0
    assert torch.equal(dec.out, px)
    dec.close()
    enc.close()


def test_batch_decode_random_bodies_match_oracle(sq, orc, route):
    """Differential fuzz with NO structure: random bytes behind a valid header (successor of sqoafuzz.c:20-32 on the
    batch path).  SQOA bodies hit REF / ALPHA / run-as-op paths at random; QOI bodies hit INDEX on an empty table;
    mono headers go to the 128-entry index.  The GPU must agree with the reference on every stream, including
    which ones are refused (REF pointing before the start, seqoia.h:733-736)."""
    import torch

    o = orc.best()
    rng = np.random.default_rng(2025)
    by_geom = {}
    for k in range(384):
        w, h = [(40, 30), (64, 64), (97, 13)][k % 3]
        hdr_ch = [4, 3, 4, 1, 2, 5, 6][k % 7]
        magic = b"qoif" if k % 4 == 1 else b"Sqoa"
        start = b"" if (magic == b"qoif" or k % 16 == 3) else b"\x31"  # k%16==3: "Sqoa" without start byte = QOI mode
        n = int(rng.integers(1, 1400))
        body = rng.integers(0, 256, size=n, dtype=np.uint8)
        if k % 5 == 0:  # bias towards op bytes that keep the stream going
            body = np.where(rng.random(n) < 0.5, rng.integers(0x80, 0x100, size=n), body).astype(np.uint8)
        s = magic + w.to_bytes(4, "big") + h.to_bytes(4, "big") + bytes([hdr_ch, 0]) + start + body.tobytes() + bytes(7) + b"\x01"
        by_geom.setdefault((w, h), []).append(s)
    for (w, h), streams in by_geom.items():
        for oc in (4, 3):
            wants = [o.decode(s, oc)[0] for s in streams]
            slot = max(len(s) for s in streams) + 32
            buf = np.zeros((len(streams), slot), dtype=np.uint8)
            for i, s in enumerate(streams):
                buf[i, : len(s)] = np.frombuffer(s, dtype=np.uint8)
            dec = sq.DecodePlan(torch.from_numpy(buf).cuda(), [len(s) for s in streams], w, h, oc)
            dec.launch()
            res = dec.results()
            out = dec.out.cpu().numpy()
            for i, want in enumerate(wants):
                if want is None:
                    assert res["status"][i] != 0, (w, h, oc, i)
                else:
                    assert res["status"][i] == 0, (w, h, oc, i, int(res["status"][i]))
                    assert out[i].tobytes() == want.tobytes(), (w, h, oc, i)
            dec.close()


def test_qoi_long_random_streams_match_oracle(sq, orc, route):
    """QOI-compat bodies of random ops over dozens of 512-byte steps: INDEX ops (a quarter of random bytes) at every
    position of the 64-op groups and across step boundaries, look-ups of slots nobody has written, runs that pass the
    end of the image.  The wave-parallel QOI decoder (segmented transform scan + stamped index updates) must agree with
    the reference byte for byte -- one workgroup per stream, and on the speculative chip-wide route (`route` = "1"), whose
    guessed decoder states are mostly WRONG on such bodies: its check must find every one of them."""
    import torch

    o = orc.best()
    rng = np.random.default_rng(77)
    w = h = 256
    streams = []
    for k in range(16):
        n = int(rng.integers(9000, 40000))
        body = rng.integers(0, 256, size=n, dtype=np.uint8)
        if k % 4 == 1:   # few runs: the image is not exhausted early
            body = np.where((body >= 0xc0) & (body < 0xfe), body & 0x3f, body).astype(np.uint8)
        if k % 4 == 2:   # two-colour ping-pong through the index between literal pixels
            body[::3] = rng.integers(0, 4, size=len(body[::3]))
        hdr_ch = (4, 3)[k & 1]
        streams.append(b"qoif" + w.to_bytes(4, "big") + h.to_bytes(4, "big") + bytes([hdr_ch, 0]) + body.tobytes() + bytes(7) + b"\x01")
    slot = max(len(s) for s in streams) + 32
    buf = np.zeros((len(streams), slot), dtype=np.uint8)
    for i, s in enumerate(streams):
        buf[i, : len(s)] = np.frombuffer(s, dtype=np.uint8)
    dev = torch.from_numpy(buf).cuda()
    for oc in (4, 3):
        wants = [o.decode(s, oc)[0] for s in streams]
        dec = sq.DecodePlan(dev, [len(s) for s in streams], w, h, oc)
        dec.launch()
        res = dec.results()
        out = dec.out.cpu().numpy()
        for i, want in enumerate(wants):
            if want is None:  # a body that happens to start with 0x31 reads as "qoif" + SQOA start byte: refused (seqoia.h:677, :683)
                assert res["status"][i] != 0, (oc, i)
                continue
            assert res["status"][i] == 0 and res["qoi_compat"][i] == 1, (oc, i)
            assert out[i].tobytes() == want.tobytes(), (oc, i)
        dec.close()


def test_decode_from_packed_blob_at_any_byte_address(sq, orc, route):
    """Streams compacted by sqoa_hip_pack sit back to back, i.e. at arbitrary byte addresses; the wave-parallel decoders
    (SQOA and QOI-compat) take them where they are: exact pixels for every alignment of the stream start."""
    import torch
    from seqoia_amd.api import Plan, DECODE, Job

    shapes = [(96, 40, 4, "mixed", 0), (64, 64, 4, "icon", 1), (333, 21, 3, "photo", 0), (120, 50, 3, "mixed", 1), (257, 9, 4, "mixed", 0),
              (80, 80, 4, "photo", 1), (33, 33, 3, "icon", 0), (512, 6, 4, "mixed", 1)]
    images, raws = [], []
    for k, (w, h, c, cls, mode) in enumerate(shapes):
        raw = orc.synth(w, h, c, cls, 8, k)
        raws.append(raw)
        images.append((torch.from_numpy(raw).cuda(), sq.Desc(w, h, c, 0, mode)))
    enc = sq.EncodeList(images)
    enc.launch()
    blob, offsets = enc.pack()
    off = offsets.cpu().numpy()
    assert len({int(o) % 4 for o in off[:-1]}) > 1  # the blob really holds unaligned streams
    # ... and the pixel outputs sit at odd addresses of one arena as well
    sizes = [w * h * c for (w, h, c, _, _) in shapes]
    arena = torch.zeros(sum(sizes) + 8 * len(sizes) + 8, dtype=torch.uint8, device="cuda")
    outs, at = [], 0
    for k, n in enumerate(sizes):
        at = (at + 7) // 8 * 8 + (k % 4)
        outs.append(arena[at:at + n]); at += n
    assert {o.data_ptr() % 4 for o in outs} == {0, 1, 2, 3}
    jobs = [Job(blob.data_ptr() + int(off[k]), outs[k].data_ptr(), int(off[k + 1] - off[k]), outs[k].numel(), sq.Desc(0, 0, shapes[k][2], 0, 0), 0)
            for k in range(len(shapes))]
    dec = Plan(DECODE, jobs, 0)
    dec.launch()
    res = dec.results()
    for k, (w, h, c, cls, mode) in enumerate(shapes):
        assert res["status"][k] == 0 and res["len"][k] == w * h * c and res["qoi_compat"][k] == mode, k
        assert bytes(outs[k].cpu().numpy()) == raws[k].tobytes(), k
    dec.close(); enc.close()


def test_mono_streams_all_channel_args(sq, orc):
    """MONO / MONO+ALPHA streams (seqoia.h:477-486, :746-748, :795-801) over many 512-byte steps through the record
    pipeline of decode_fallback_kernel, for every `channels` argument; and QOI-compat colour streams to 1 and 2 channels."""
    import torch
    from seqoia_amd.api import Plan, DECODE, Job

    o = orc.best()
    rng = np.random.default_rng(5)
    streams = []
    for k, (w, h, c) in enumerate([(300, 200, 1), (257, 129, 2), (640, 48, 1), (64, 64, 2)]):
        n = w * h * c
        raw = ((np.arange(n) // (7 + k)) + rng.integers(0, 2, n) * rng.integers(0, 60, n)).astype(np.uint8)
        raw[n // 3: n // 3 + 3000 * c] = 17          # a long run (BIGRUN)
        streams.append((o.encode(raw, w, h, c, 0), w, h))
    for k, (w, h, c) in enumerate([(200, 100, 4), (96, 96, 3)]):  # QOI-compat colour -> g / g a
        streams.append((o.encode(orc.synth(w, h, c, "mixed", 2, k), w, h, c, 1), w, h))
    blob = torch.from_numpy(np.frombuffer(b"".join(s for s, _, _ in streams), dtype=np.uint8).copy()).cuda()
    offs = np.cumsum([0] + [len(s) for s, _, _ in streams])
    for arg in (0, 1, 2, 3, 4):
        wants = [o.decode(s, arg) for s, _, _ in streams]
        outs = [torch.zeros(len(wt[0]), dtype=torch.uint8, device="cuda") for wt in wants]
        jobs = [Job(blob.data_ptr() + int(offs[i]), outs[i].data_ptr(), len(streams[i][0]), outs[i].numel(), sq.Desc(0, 0, arg, 0, 0), 0)
                for i in range(len(streams))]
        dec = Plan(DECODE, jobs, 0)
        dec.launch()
        res = dec.results()
        for i, (want, desc) in enumerate(wants):
            assert res["status"][i] == 0 and res["len"][i] == len(want), (arg, i)
            assert bytes(outs[i].cpu().numpy()) == want.tobytes(), (arg, i)
        dec.close()


def test_many_short_streams_take_the_small_workgroup_shapes(sq, orc):
    """More than 2048 short streams in one plan: both decoders run with fewer waves per image (more images at once;
    decode_wave_kernel<SQOA_DEC_W_SMALL, 0>, decode_qoi_kernel<SQOA_QDEC_W_SMALL>).  Random-body streams (INDEX on an empty table, runs, every op length) and encoder
    output, RGB and RGBA, to 3 and 4 channels; every stream against the reference."""
    import torch

    o = orc.best()
    rng = np.random.default_rng(909)
    w, h = 24, 20
    base = []
    for k in range(48):
        ch = (3, 4)[k & 1]
        native = (k // 6) & 1   # both wire formats: the SQOA wave decoder has a small shape too (four waves per stream)
        if k % 3 == 0:   # random bytes behind a valid header (1..3 chunks of 512 bytes)
            body = bytes(rng.integers(0, 256, int(rng.integers(200, 1500)), dtype=np.uint8))
            head = (b"Sqoa", b"qoif")[1 - native] + w.to_bytes(4, "big") + h.to_bytes(4, "big") + bytes([ch, 0])
            s = head + (b"\x31" if native else b"") + body + bytes(7) + b"\x01"
        else:
            s = o.encode(orc.synth(w, h, ch, ["mixed", "icon", "photo"][k % 3], seed=12, img=k), w, h, ch, 1 - native)
        base.append(s)
    n = 2304
    streams = [base[i % len(base)] for i in range(n)]
    for oc in (4, 3):
        wants = [o.decode(s, oc)[0] for s in base]
        slot = max(len(s) for s in base) + 32
        buf = np.zeros((n, slot), dtype=np.uint8)
        for i, s in enumerate(streams):
            buf[i, : len(s)] = np.frombuffer(s, dtype=np.uint8)
        dec = sq.DecodePlan(torch.from_numpy(buf).cuda(), [len(s) for s in streams], w, h, oc)
        dec.launch()
        res = dec.results()
        out = dec.out.cpu().numpy()
        for i in range(n):
            want = wants[i % len(base)]
            if want is None:
                assert res["status"][i] != 0, (oc, i)
            else:
                assert res["status"][i] == 0 and out[i].tobytes() == want.tobytes(), (oc, i)
        dec.close()


def test_waveonly_jobs_skip_the_other_decoders_and_hand_back_what_they_cannot_take(sq, orc):
    """SQOA_HIP_JOB_WAVEONLY (what the legacy sqoa_decode sets after reading the header itself): a plan whose jobs all carry it
    launches only the SQOA wave decoder.  SQOA colour streams decode as ever; a QOI-compat stream and a stream with an
    SQOA_OP_REF in it come back SLOWPATH for the caller to submit again -- which the legacy call does (the REF KATs above)."""
    import torch

    from seqoia_amd import Desc, Job
    from seqoia_amd.api import Plan, DECODE

    o = orc.best()
    w, h = 96, 40
    px = orc.synth(w, h, 4, "mixed", 3, 0)
    s_sqoa, s_qoi = o.encode(px, w, h, 4, 0), o.encode(px, w, h, 4, 1)
    body = bytes.fromhex("fe010203fe09090945a188a288") + bytes(7) + b"\x01"            # a REF op (0x45), SURVEY appendix B
    s_ref = b"Sqoa" + (6).to_bytes(4, "big") + (1).to_bytes(4, "big") + bytes([4, 0, 0x31]) + body
    streams = [s_sqoa, s_qoi, s_ref, s_sqoa]
    blob = torch.zeros((len(streams), 65536), dtype=torch.uint8, device="cuda")
    for k, s in enumerate(streams):
        blob[k, : len(s)] = torch.frombuffer(bytearray(s), dtype=torch.uint8).cuda()
    out = torch.zeros((len(streams), w * h * 4), dtype=torch.uint8, device="cuda")
    flag = 8  # SQOA_HIP_JOB_WAVEONLY
    jobs = [Job(blob[k].data_ptr(), out[k].data_ptr(), len(s), w * h * 4, Desc(0, 0, 4, 0, 0), flag) for k, s in enumerate(streams)]
    plan = Plan(DECODE, jobs)
    plan.launch()
    res = plan.results()
    assert list(res["status"]) == [0, 4, 4, 0]
    assert out[0].cpu().numpy().tobytes() == px.tobytes() and out[3].cpu().numpy().tobytes() == px.tobytes()
    plan.close()
    # one job without the flag: the plan launches everything again
    jobs[1] = Job(blob[1].data_ptr(), out[1].data_ptr(), len(s_qoi), w * h * 4, Desc(0, 0, 4, 0, 0), 0)
    plan = Plan(DECODE, jobs)
    plan.launch()
    res = plan.results()
    assert list(res["status"]) == [0, 0, 0, 0]
    assert out[1].cpu().numpy().tobytes() == px.tobytes()
    want, _ = o.decode(s_ref, 4)
    assert out[2, : 6 * 4].cpu().numpy().tobytes() == want.tobytes()
    plan.close()
    # an unknown flag on a decode job is a bad argument
    bad = Plan(DECODE, [Job(blob[0].data_ptr(), out[0].data_ptr(), len(s_sqoa), w * h * 4, Desc(0, 0, 4, 0, 0), 16)])
    bad.launch()
    assert bad.results()["status"][0] == 1
    bad.close()


def _qoi_stream(w, h, ch, ops):
    return b"qoif" + w.to_bytes(4, "big") + h.to_bytes(4, "big") + bytes([ch, 0]) + b"".join(ops) + bytes(7) + b"\x01"


def _qoi_hash(r, g, b, a=255):
    return (r * 3 + g * 5 + b * 7 + a * 11) % 64   # seqoia.h:785-787 (qoi_compat)


@pytest.mark.parametrize("oc", [4, 3])
def test_streams_with_more_op_starts_than_the_balanced_kernel_has_slots(sq, orc, monkeypatch, oc):
    """The balanced kernel has 512 op slots per 1024-byte chunk and hands on a stream whose chunk holds more starts than that -- which the
    reference's own encoder writes for plain content: a ramp is two-byte LUMA ops throughout, and the one-byte RUN in front of them makes
    513 starts of the first chunk; a flat half below a noisy half is BIGRUN after BIGRUN, every byte a start.  Round 5 found such streams
    at 2 Mpixel/s in the exact decoder (515 ms for a launch of 256 ramps): now the wave decoder takes them in a retry behind the balanced
    kernel (sqoa_launch_decode_standby).  The plan's own choice of kernel for a workgroup per stream (the three chip-wide passes, which a
    plan of 64 streams would take, switched off): the reference's pixels, in a millisecond range -- on the first launch (repaired by
    sqoa_hip_plan_results) and on the relaunches (the kernels behind the first one stay in the launch)."""
    import torch

    monkeypatch.setenv("SQOA_HIP_MULTIPASS", "0")
    monkeypatch.setenv("SQOA_HIP_DEC_CONTENT", "0")   # (the plan would drop the balanced kernel after the first launch of such streams: the next test; here the retry is the subject)
    n, s = 64, 512
    x = torch.arange(s, device="cuda", dtype=torch.int32)
    row = (x & 255).to(torch.uint8)
    ramp = torch.stack([row, row, row, torch.full_like(row, 255)], dim=-1).expand(n // 2, s, s, 4).contiguous()
    g = torch.Generator(device="cuda").manual_seed(5)
    half = torch.randint(0, 256, (n // 2, s, s, 4), device="cuda", dtype=torch.uint8, generator=g)
    half[..., 3] = 255
    half[:, s // 2 :, :, :3] = 200   # (131 072 flat pixels: 256 BIGRUN bytes on end)
    px = torch.cat([ramp, half], dim=0)
    enc = sq.EncodePlan(px, qoi_compat=0)
    enc.launch()
    res = enc.results()
    assert (res["status"] == 0).all()
    streams = enc.streams()
    o = orc.best()
    dec = sq.DecodePlan(enc.out, res["len"], s, s, oc)
    for launch in range(3):
        dec.launch()
        r = dec.results()
        assert (r["status"] == 0).all() and (r["len"] == s * s * oc).all(), (launch, r["status"][:4])
        assert torch.equal(dec.out[..., :3], px[..., :3]) and (oc == 3 or bool((dec.out[..., 3] == 255).all())), launch
    for i in (0, n // 2, n - 1):
        assert dec.out[i].cpu().numpy().tobytes() == o.decode(streams[i], oc)[0].tobytes(), i
    dec.time(2)
    ms = min(dec.time(3) / 3 for _ in range(3))
    assert ms < 5.0, ms   # (16.8 Mpixel: the balanced kernel as far as it gets + the retry; the exact decoder took 130 ms for these)
    dec.close()
    enc.close()


def test_a_decode_plan_learns_from_the_content_its_launches_see(sq, orc, monkeypatch):
    """What a plan can see without reading a stream -- sizes, channel counts -- does not tell photographs from screenshots, and the balanced
    kernel (the plan's choice for large streams to four channels) is the slower one on ops of two bytes throughout (every lane owns exactly
    eight ops: its LDS accesses fall on a quarter of the banks) and has to hand on streams with more op starts in a chunk than it has slots.
    The kernel reports what it saw in the result rows (row_content_hint, sqoa_common.h) and sqoa_hip_plan_results, from a full view of a launch,
    drops the balanced kernel for the NEXT launches where such streams are most of the plan's pixels: class photo and ramps switch, class mixed
    stays; SQOA_HIP_DEC_CONTENT=0 and a forced SQOA_HIP_DEC_BAL keep the first choice; the pixels are the same throughout."""
    import torch

    monkeypatch.setenv("SQOA_HIP_MULTIPASS", "0")
    n, s = 64, 512
    x = torch.arange(s, device="cuda", dtype=torch.int32)
    row = (x & 255).to(torch.uint8)
    ramp = torch.stack([row, row, row, torch.full_like(row, 255)], dim=-1).expand(n, s, s, 4).contiguous()

    def run(px, switches, env=(), splits=False):
        for k, v in env:
            monkeypatch.setenv(k, v)
        enc = sq.EncodePlan(px, qoi_compat=0)
        enc.launch()
        res = enc.results()
        assert (res["status"] == 0).all()
        dec = sq.DecodePlan(enc.out, res["len"], s, s, 4)
        assert dec.shape() & 8, "large streams to four channels: the balanced kernel is the plan's first choice"
        for launch in range(3):
            dec.out.zero_()
            torch.cuda.synchronize()
            dec.launch()
            torch.cuda.synchronize()
            if launch:   # (the first launch of ramps is repaired by results(): the retry)
                assert torch.equal(dec.out, px), launch
            r = dec.results()
            assert (r["status"] == 0).all() and (r["pad"] == 0).all() and torch.equal(dec.out, px), launch
            assert bool(dec.shape() & 8) == (not switches), (launch, dec.shape())
            assert bool(dec.shape() & 0x200) == splits, (launch, hex(dec.shape()))   # (a part of the list launched apart, beside the rest)
            # (the content hint shares its byte with the "stand-by kernel used" mark: a plan of SQOA colour streams that fit the kernel's slots must
            # still learn that it needs none of the kernels behind -- round 6's first version read the whole byte and launched all three for ever)
            if px is not ramp:
                assert dec.shape() & 0x100, (launch, hex(dec.shape()))
        dec.close()
        enc.close()
        for k, _ in env:
            monkeypatch.delenv(k)

    photo = torch.empty((n, s, s, 4), dtype=torch.uint8, device="cuda")
    sq.synth(photo, "photo", seed=1, img0=0)
    mixed = torch.empty((n, s, s, 4), dtype=torch.uint8, device="cuda")
    sq.synth(mixed, "mixed", seed=1, img0=0)
    run(photo, True)
    run(mixed, False)
    run(ramp, True)
    run(photo, False, env=(("SQOA_HIP_DEC_CONTENT", "0"),))
    run(photo, False, env=(("SQOA_HIP_DEC_BAL", "1"),))
    # A list of both kinds: from a twentieth of the pixels on the photographs are split off into a launch of their own beside the others (the
    # balanced kernel keeps the screenshots); from half of the pixels on the whole list changes kernel; two photographs among 62 screenshots stay
    run(torch.cat([photo[:16], mixed[:48]]), False, splits=True)
    run(torch.cat([photo[:48], mixed[:16]]), True)
    run(torch.cat([photo[:2], mixed[:62]]), False)
    run(torch.cat([photo[:16], mixed[:48]]), False, env=(("SQOA_HIP_DEC_CONTENT", "0"),))
    # a split plan in a graph: the launch beside the balanced kernel's is forked from and joined to the capturing stream by two events
    px = torch.cat([photo[:16], mixed[:48]])
    enc = sq.EncodePlan(px, qoi_compat=0)
    enc.launch()
    res = enc.results()
    dec = sq.DecodePlan(enc.out, res["len"], s, s, 4)
    dec.launch()
    assert (dec.results()["status"] == 0).all() and dec.shape() & 0x200
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        dec.launch(side.cuda_stream)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        dec.launch(side.cuda_stream)
    for _ in range(2):
        dec.out.zero_()
        torch.cuda.synchronize()
        g.replay()
        torch.cuda.synchronize()
        assert torch.equal(dec.out, px)
        assert (dec.results()["status"] == 0).all()
    dec.close()
    enc.close()
    # ... and the shape of BASELINE configs[4]: three-channel photographs among four-channel screenshots, one list.  They STAY in the balanced kernel's
    # three-channel instance (what a three-channel job would move to stores bytes: measured slower on that suite, sqoa_runtime.hip)
    o = orc.best()
    imgs = [(photo[k, ..., :3].contiguous(), 3) for k in range(16)] + [(mixed[k], 4) for k in range(48)]
    enc = sq.EncodeList([(t.reshape(-1), sq.Desc(s, s, ch, 0, 0)) for t, ch in imgs])
    enc.launch()
    res = enc.results()
    assert (res["status"] == 0).all()
    base = enc.out.data_ptr()
    dec = sq.DecodeList([(base + enc.offsets[k], int(res["len"][k]), s, s, ch) for k, (_, ch) in enumerate(imgs)])
    assert dec.shape() & 8 and dec.shape() & 16 and not dec.shape() & 0x200
    for launch in range(3):
        dec.out.zero_()
        torch.cuda.synchronize()
        dec.launch()
        torch.cuda.synchronize()
        for k, (t, ch) in enumerate(imgs):
            assert torch.equal(dec.image(k), t.reshape(-1)), (launch, k)
        assert (dec.results()["status"] == 0).all()
        assert dec.shape() & 8 and dec.shape() & 16 and not dec.shape() & 0x200, (launch, hex(dec.shape()))
    assert dec.image(3).cpu().numpy().tobytes() == o.decode(enc.streams()[3], 3)[0].tobytes()
    dec.close()
    enc.close()


def test_qoi_groups_of_few_index_ops_match_oracle(sq, orc, route):
    """The resolver settles a 64-op group that holds up to SQOA_QDEC_SEQ_MAX (4) INDEX ops one op after the other in registers, and a group with
    more by the fixed point: bodies whose INDEX ops are SPARSE, at densities from one in 64 ops to ten in 64 -- groups on both
    sides of the threshold, next to each other in one stream -- between DIFF / LUMA ops (every pixel lands in some slot: most INDEX ops find
    their slot written earlier in their own group, by a relative op or by another INDEX op), literals and short runs.  Byte for byte the
    reference's pixels, one workgroup per stream and on the speculative route."""
    import torch

    o = orc.best()
    rng = np.random.default_rng(4242)
    w = h = 192
    streams = []
    for k, per64 in enumerate((1, 2, 3, 4, 5, 6, 8, 10, 4, 4)):
        n = int(rng.integers(30000, 50000))
        kind = rng.random(n)
        body = rng.integers(0x40, 0x80, size=n).astype(np.uint8)                                   # DIFF
        body = np.where(kind < 0.15, rng.integers(0x80, 0xc0, size=n), body).astype(np.uint8)     # LUMA op bytes (the byte behind is whatever comes)
        body = np.where(kind > 0.985, 0xfe + (rng.random(n) < 0.3), body).astype(np.uint8)        # RGB / RGBA literals
        body = np.where((kind > 0.97) & (kind <= 0.985), rng.integers(0xc0, 0xc4, size=n), body).astype(np.uint8)   # runs of 1..4
        idx = rng.random(n) < per64 / 64.0
        if k >= 8:   # INDEX ops in pairs and triples: one INDEX op's pixel is the next one's "most recent writer of my slot"
            idx = idx | np.roll(idx, 1) | (np.roll(idx, 2) & (rng.random(n) < 0.5))
        body = np.where(idx, rng.integers(0, 64 if k != 9 else 6, size=n), body).astype(np.uint8)
        hdr_ch = (4, 3)[k & 1]
        streams.append(b"qoif" + w.to_bytes(4, "big") + h.to_bytes(4, "big") + bytes([hdr_ch, 0]) + body.tobytes() + bytes(7) + b"\x01")
    slot = max(len(s) for s in streams) + 32
    buf = np.zeros((len(streams), slot), dtype=np.uint8)
    for i, s in enumerate(streams):
        buf[i, : len(s)] = np.frombuffer(s, dtype=np.uint8)
    dev = torch.from_numpy(buf).cuda()
    for oc in (4, 3):
        wants = [o.decode(s, oc)[0] for s in streams]
        assert all(x is not None for x in wants)
        dec = sq.DecodePlan(dev, [len(s) for s in streams], w, h, oc)
        dec.launch()
        res = dec.results()
        out = dec.out.cpu().numpy()
        for i, want in enumerate(wants):
            assert res["status"][i] == 0 and res["qoi_compat"][i] == 1, (oc, i)
            assert out[i].tobytes() == want.tobytes(), (oc, i)
        dec.close()


def test_speculative_qoi_route_finds_stale_index_entries(sq, orc, monkeypatch):
    """The speculative chip-wide QOI-compat decode (decode_qoi_kernel<8, SPEC> + dq_compare_kernel + dq_verify_kernel) guesses the
    colour index in front of every workgroup's range from the 4 KiB before it.  Streams built so that the guess is wrong in exactly
    the ways the check has to catch: an entry written once at the stream's start and read far behind it (never inside any warm-up);
    the same with the entry rewritten to ANOTHER colour two ranges before the read (the warm-up misses the write that counts); an
    entry read through a DIFF chain that starts from it; and the harmless twin of each (entry rewritten inside the warm-up), which
    the route may accept.  Whatever it decides, the pixels are the reference's."""
    import torch

    monkeypatch.setenv("SQOA_HIP_MULTIPASS", "1")
    o = orc.best()
    rng = np.random.default_rng(31)
    w = h = 256
    N = w * h
    key = (200, 100, 50)
    slot = _qoi_hash(*key)
    other = next((r, 7, 9) for r in range(256) if _qoi_hash(r, 7, 9) == slot and (r, 7, 9) != key)

    def rgb(c):
        return bytes([0xfe, c[0], c[1], c[2]])

    def noise(n):   # literal pixels that never touch `slot`
        out = []
        while len(out) < n:
            c = tuple(int(x) for x in rng.integers(0, 256, size=3))
            if _qoi_hash(*c) != slot:
                out.append(rgb(c))
        return out

    streams = []
    for variant in range(6):
        ops = [rgb(key)]
        fill = N - 1
        if variant == 0:     # read at the very end
            ops += noise(fill - 1) + [bytes([slot])]
        elif variant == 1:   # read every 3000 pixels
            body = noise(fill)
            for k in range(2999, fill, 3000):
                body[k] = bytes([slot])
            ops += body
        elif variant == 2:   # rewritten to another colour, read 2500 ops (10 000 bytes: more than a warm-up) later
            body = noise(fill)
            body[20000] = rgb(other)
            body[22500] = bytes([slot])
            ops += body
        elif variant == 3:   # a DIFF chain behind the stale read, then an INDEX op on what the chain wrote
            body = noise(fill)
            body[30000] = bytes([slot])
            for k in range(30001, 30040):
                body[k] = bytes([0x40 | 0x3f])   # +1 +1 +1
            ops += body
        elif variant == 4:   # the harmless twin of 2: rewritten 100 ops in front of the read
            body = noise(fill)
            body[22400] = rgb(other)
            body[22500] = bytes([slot])
            ops += body
        else:                # no literal pixels at all behind the first: DIFF / LUMA ops and reads of the one entry
            body = [bytes([0x40 | int(x)]) for x in rng.integers(0, 64, size=fill)]
            for k in range(4999, fill, 5000):
                body[k] = bytes([slot])
            ops += body
        streams.append(_qoi_stream(w, h, 3, ops))
    slotb = max(len(s) for s in streams) + 32
    buf = np.zeros((len(streams), slotb), dtype=np.uint8)
    for i, s in enumerate(streams):
        buf[i, : len(s)] = np.frombuffer(s, dtype=np.uint8)
    dev = torch.from_numpy(buf).cuda()
    for oc in (3, 4):
        dec = sq.DecodePlan(dev, [len(s) for s in streams], w, h, oc)
        dec.launch()
        res = dec.results()
        out = dec.out.cpu().numpy()
        for i, s in enumerate(streams):
            want = o.decode(s, oc)[0]
            assert res["status"][i] == 0 and res["len"][i] == N * oc, (oc, i)
            assert out[i].tobytes() == want.tobytes(), (oc, i)
        dec.close()


def test_launches_behind_results_keep_decoding(sq, orc):
    """The plan's lazy skip (sqoa_runtime.hip: once a launch's result rows say that no job needed the kernels behind the wave decoder, the
    next launches leave them out) must leave out THOSE kernels and nothing else.  Round 5's version also switched off the speculative
    chip-wide pass of multi-pass QOI-compat streams, whose rows never say "stand-by used": every launch behind a results() call left such a
    stream undecoded, and results() repaired it with the one-workgroup kernel -- invisible to a caller who reads pixels after a stream
    synchronise, or replays a graph.  Here: launch, results(), zero the output, launch again, synchronise, compare the pixels BEFORE
    results() is asked; then the same through plan.time() (what bench.py's secondary rows do)."""
    import torch

    w = h = 1024
    for cls, ch, qoi in (("photo", 3, 1), ("mixed", 4, 1), ("mixed", 4, 0)):
        px = torch.empty((1, h, w, ch), dtype=torch.uint8, device="cuda")
        sq.synth(px, cls, seed=9, img0=2)
        enc = sq.EncodePlan(px, qoi_compat=qoi)
        enc.launch()
        res = enc.results()
        assert res["status"][0] == 0
        dec = sq.DecodePlan(enc.out, res["len"], w, h, ch)
        dec.launch()
        assert dec.results()["status"][0] == 0 and torch.equal(dec.out, px)
        for _ in range(3):   # (round 5 alternated: skip, repair, skip, ...)
            dec.out.zero_()
            torch.cuda.synchronize()
            dec.launch()
            torch.cuda.synchronize()
            assert torch.equal(dec.out, px), (cls, ch, qoi, "a launch behind results() left the stream undecoded")
            assert dec.results()["status"][0] == 0
        dec.out.zero_()
        torch.cuda.synchronize()
        dec.time(3)
        torch.cuda.synchronize()
        assert torch.equal(dec.out, px), (cls, ch, qoi, "timed launches decode nothing")
        dec.close()
        enc.close()


def test_a_graph_captured_behind_results_keeps_the_standby_kernels(sq, orc):
    """A launch captured into a graph AFTER results() has told the plan that nobody needed the kernels behind the wave decoder must still
    carry them: the graph outlives that knowledge.  Replayed over a stream that needs them (an SQOA_OP_REF patched into the same buffer,
    seqoia.h:729-738: the exact decoder's business), the pixels must be the reference's -- and results() must say OK, not SLOWPATH."""
    import torch

    w, h = 256, 256
    px = torch.empty((4, h, w, 4), dtype=torch.uint8, device="cuda")
    sq.synth(px, "mixed", seed=3, img0=0)
    enc = sq.EncodePlan(px, qoi_compat=0)
    enc.launch()
    res = enc.results()
    assert (res["status"] == 0).all()
    o = orc.best()
    streams = enc.streams()
    # a patched copy of stream 2 that the reference still decodes (a REF op somewhere in the middle)
    patched, want = None, None
    for pos in range(len(streams[2]) // 2, len(streams[2]) // 2 + 400):
        cand = bytearray(streams[2]); cand[pos] = 0x45
        got = o.decode(bytes(cand), 4)[0]
        if got is not None and got.tobytes() != px[2].cpu().numpy().tobytes():
            patched, want = bytes(cand), got
            break
    assert patched is not None
    buf = enc.out.clone()
    dec = sq.DecodePlan(buf, res["len"], w, h, 4)
    dec.launch()
    assert (dec.results()["status"] == 0).all() and torch.equal(dec.out, px)   # (from here on the plan's own launches skip the stand-by kernels)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        dec.launch(side.cuda_stream)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        dec.launch(side.cuda_stream)
    buf[2, : len(patched)] = torch.frombuffer(bytearray(patched), dtype=torch.uint8).cuda()
    for _ in range(2):
        dec.out.zero_()
        torch.cuda.synchronize()
        g.replay()
        torch.cuda.synchronize()
        assert dec.out[2].cpu().numpy().tobytes() == want.tobytes(), "the replayed graph left the REF stream to nobody"
        assert torch.equal(dec.out[0], px[0]) and torch.equal(dec.out[3], px[3])
        r = dec.results()
        assert (r["status"] == 0).all() and (r["pad"] == 0).all()
    # ... and the plan's own launches find their way as well (results() repairs the first, the next carries the kernels again)
    for _ in range(2):
        dec.out.zero_()
        torch.cuda.synchronize()
        dec.launch()
        assert (dec.results()["status"] == 0).all()
        assert dec.out[2].cpu().numpy().tobytes() == want.tobytes()
    dec.close()
    enc.close()


@pytest.mark.parametrize("cls,ch", [("mixed", 4), ("mixed", 3), ("photo", 3)])
def test_lone_large_qoi_stream_is_decoded_by_the_whole_chip(sq, orc, cls, ch):
    """One 1024 x 1024 QOI-compat stream alone in a plan: the plan's own choice is the speculative chip-wide route, and on
    photo-like content its guesses hold -- visible as time: one workgroup needs 8.7 ms for such a stream (one compute unit's
    instruction issue), the chip 0.1 ms; a failed guess would cost both.  Pixels against the reference's decode of the same stream."""
    import torch

    w = h = 1024
    px = torch.empty((1, h, w, ch), dtype=torch.uint8, device="cuda")
    sq.synth(px, cls, seed=9, img0=2)
    enc = sq.EncodePlan(px, qoi_compat=1)
    enc.launch()
    res = enc.results()
    assert res["status"][0] == 0
    dec = sq.DecodePlan(enc.out, res["len"], w, h, ch)
    dec.launch()
    assert dec.results()["status"][0] == 0
    want = orc.best().decode(enc.streams()[0], ch)[0]
    assert dec.out[0].cpu().numpy().tobytes() == want.tobytes()
    dec.time(3)
    ms = min(dec.time(3) / 3 for _ in range(3))
    assert ms < 2.0, ms
    dec.close()
    enc.close()


@pytest.mark.parametrize("knob", ["plan", "1", "0"])
def test_run_heavy_streams_take_the_position_major_store_path(sq, orc, monkeypatch, knob):
    """Icon-like streams (BASELINE configs[3]: RUN-heavy) to 3 and to 4 channels: chunks that are full of runs store their pixels 64
    consecutive ones per instruction (sqoa_decode_wave.hip, SWEEP) -- the plan's own choice for streams of under one byte per
    pixel, forced on, forced off: the pixels are the reference's (seqoia.h:723-725, :790-805) every time.  Streams built around
    the path's edges: runs across chunk boundaries, chunks of more pixels than the path takes (runs of hundreds: the fill loop),
    an image that ends inside a chunk whose runs overshoot it, 3-byte output at every alignment of its first byte."""
    import torch

    o = orc.best()
    rng = np.random.default_rng(99)
    if knob != "plan":
        monkeypatch.setenv("SQOA_HIP_DEC_SWEEP", knob)
    cases = []   # (stream bytes, w, h)
    for k, (w, h, ch) in enumerate(((256, 256, 3), (256, 256, 4), (511, 130, 4), (640, 97, 3))):
        px = orc.synth(w, h, ch, "icon", 11, k)
        cases.append((o.encode(px, w, h, ch, 0), w, h))
    # hand-made: rows of short runs of random colours with literal ops in between, long runs, one huge run
    def hdr(w, h, ch=4):
        return b"Sqoa" + w.to_bytes(4, "big") + h.to_bytes(4, "big") + bytes([ch, 0, 0x31])
    END = bytes(7) + b"\x01"
    body = bytearray()
    for _ in range(6000):
        body += bytes([0xfe]) + bytes(rng.integers(0, 256, 3).tolist()) + bytes([0xc0 | int(rng.integers(0, 40))])
        if rng.integers(0, 4) == 0:
            body += bytes([0x80 | int(rng.integers(0, 64)), int(rng.integers(0, 256))])
        if rng.integers(0, 50) == 0:
            body += b"\xfd" * int(rng.integers(1, 6)) + bytes([0xc0 | int(rng.integers(0, 61))])
    cases.append((hdr(700, 300) + bytes(body) + END, 700, 300))          # ends inside the stream: ops behind the image's end
    cases.append((hdr(900, 400) + bytes(body) + END, 900, 400))          # the stream ends early: the last pixel repeats
    cases.append((hdr(640, 480) + bytes(body[:3000]) + b"\xfd" * 700 + END, 640, 480))
    for oc in (3, 4):
        for stream, w, h in cases:
            want, _ = o.decode(stream, oc)
            for shift in ((0,) if oc == 4 else (0, 1, 2, 3)):           # byte address of the first pixel, mod 4
                buf = torch.zeros((len(stream) + 64,), dtype=torch.uint8, device="cuda")
                buf[: len(stream)] = torch.frombuffer(bytearray(stream), dtype=torch.uint8).cuda()
                arena = torch.zeros((w * h * oc + 64,), dtype=torch.uint8, device="cuda")
                out = arena[shift : shift + w * h * oc].view(1, h, w, oc)
                dec = sq.DecodePlan(buf[: len(stream)].view(1, -1), [len(stream)], w, h, oc, out=out)
                dec.launch()
                res = dec.results()
                assert res["status"][0] == 0, (oc, w, h, shift)
                assert bytes(dec.out.reshape(-1).cpu().numpy()) == want.tobytes(), (oc, w, h, shift, knob)
                assert int(arena[shift + w * h * oc :].max()) == 0 and (shift == 0 or int(arena[:shift].max()) == 0), "wrote outside the image"
                dec.close()


def test_a_relaunched_plan_finds_its_result_rows_reset(sq, orc, route):
    """No copy resets the result rows in front of a launch: the first kernel that takes a job resets its row itself (sqoa_runtime.hip,
    plan_enqueue), and a refused job's row is written once when the plan is filled.  So a plan launched AGAIN, over the same buffers
    with other bytes in them and no plan_update in between, must report what it finds now, whatever the launch before left in the row:
    a broken header (BADARG) after a good stream, a good SQOA stream after the broken header (status 0, exact length, the pixels),
    a QOI-compat stream (through SLOWPATH to the QOI-compat kernel) after that, the SQOA stream again; and the rows of jobs the plan
    refused stay as they were through all of it.  On every route of the `route` fixture (one workgroup per stream in its shapes, three
    passes)."""
    import torch

    o = orc.best()
    w, h, ch = 160, 90, 4
    px = [orc.synth(w, h, ch, cls, seed=21, img=k) for k, cls in enumerate(["mixed", "icon", "photo", "mixed"])]
    sqoa = [o.encode(p, w, h, ch, 0) for p in px]
    qoi = [o.encode(p, w, h, ch, 1) for p in px]
    slot = max(len(s) for s in sqoa + qoi) + 64
    buf = torch.zeros((len(px) + 1, slot), dtype=torch.uint8, device="cuda")

    def put(streams):
        host = np.zeros((len(px) + 1, slot), dtype=np.uint8)
        for i, s in enumerate(streams):
            host[i, : len(s)] = np.frombuffer(s, dtype=np.uint8)
        buf.copy_(torch.from_numpy(host).cuda())
        torch.cuda.synchronize()

    # (a job states its stream's exact size: the SQOA and the QOI-compat stream of an image differ in length, so each framing has a plan of its own)
    items = [(buf[i].data_ptr(), len(sqoa[i]), w, h, 4) for i in range(len(px))] + [(buf[len(px)].data_ptr(), 10, w, h, 4)]   # the last job: a stream shorter than a header
    dec = sq.DecodeList(items)

    def check(expect_ok, streams):
        dec.launch()
        res = dec.results()
        for i in range(len(px)):
            if expect_ok[i]:
                assert res["status"][i] == 0 and res["len"][i] == w * h * 4, (i, res["status"][i], res["len"][i])
                want = o.decode(streams[i], 4)[0]
                assert dec.image(i).cpu().numpy().tobytes() == want.tobytes(), i
            else:
                assert res["status"][i] != 0 and res["len"][i] == 0, (i, res["status"][i], res["len"][i])
        assert res["status"][len(px)] != 0 and res["len"][len(px)] == 0   # the refused job's row, untouched by every launch

    put(sqoa)
    check([True] * 4, sqoa)
    broken = [bytes([0x51]) + s[1:] if i % 2 == 0 else s for i, s in enumerate(sqoa)]   # a wrong magic in jobs 0 and 2
    put(broken)
    check([i % 2 == 1 for i in range(4)], broken)
    put(sqoa)
    check([True] * 4, sqoa)
    # an SQOA_OP_REF in job 1 (handed to the exact decoder: SLOWPATH on the way), then the plain stream again
    ref = list(sqoa)
    body = bytearray(ref[1])
    body[15 + 40] = 0x05
    ref[1] = bytes(body)
    if o.decode(ref[1], 4)[0] is not None:
        put(ref)
        check([True] * 4, ref)
        put(sqoa)
        check([True] * 4, sqoa)
    dec.close()
    # the QOI-compat leg: a plan over QOI-compat streams (SLOWPATH in the wave decoder, decoded by the kernel behind it), broken, whole again
    items = [(buf[i].data_ptr(), len(qoi[i]), w, h, 4) for i in range(len(px))] + [(buf[len(px)].data_ptr(), 10, w, h, 4)]
    dec = sq.DecodeList(items)
    put(qoi)
    check([True] * 4, qoi)
    broken = [s[:12] + bytes([9]) + s[13:] if i == 3 else s for i, s in enumerate(qoi)]      # channels byte 9 in job 3
    put(broken)
    check([i != 3 for i in range(4)], broken)
    put(qoi)
    check([True] * 4, qoi)
    dec.close()


@pytest.mark.parametrize("split", ["0", "1"])
def test_a_plan_of_mixed_output_formats_splits_its_launches(sq, orc, monkeypatch, split):
    """A decode plan whose jobs go to different output formats (BASELINE configs[4]: RGB photos to three channels among RGBA images to
    four): the four-channel streams go to the balanced kernel in a launch of their own, the others run beside them on the plan's side
    stream (forked from and joined to the launch's stream) -- every image as the compiled reference decodes it, launched directly, launched
    again, and replayed from a HIP graph (the fork / join is part of the capture); SQOA_HIP_DEC_SPLIT=0 is the single launch."""
    import torch

    monkeypatch.setenv("SQOA_HIP_DEC_SPLIT", split)
    monkeypatch.setenv("SQOA_HIP_MULTIPASS", "0")
    o = orc.best()
    imgs = []
    for k in range(40):
        ch = 3 if k % 5 == 0 else 4
        w, h = (300, 200) if k % 3 else (220, 160)
        cls = ["mixed", "photo", "icon"][k % 3]
        imgs.append((w, h, ch, orc.synth(w, h, ch, cls, seed=9, img=k)))
    streams = [o.encode(px, w, h, ch, 0) for (w, h, ch, px) in imgs]
    slot = max(len(s) for s in streams) + 64
    host = np.zeros((len(streams), slot), dtype=np.uint8)
    for i, s in enumerate(streams):
        host[i, : len(s)] = np.frombuffer(s, dtype=np.uint8)
    buf = torch.from_numpy(host).cuda()
    items = [(buf[i].data_ptr(), len(streams[i]), w, h, ch) for i, (w, h, ch, _) in enumerate(imgs)]
    dec = sq.DecodeList(items)

    def check():
        res = dec.results()
        assert (res["status"] == 0).all()
        for i, (w, h, ch, px) in enumerate(imgs):
            assert dec.image(i).cpu().numpy().tobytes() == px.tobytes(), i

    dec.launch()
    check()
    dec.out.zero_()
    torch.cuda.synchronize()
    dec.launch()
    check()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        dec.launch(side.cuda_stream)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        dec.launch(side.cuda_stream)
    for _ in range(2):
        dec.out.zero_()
        torch.cuda.synchronize()
        g.replay()
        torch.cuda.synchronize()
        check()
    dec.close()
