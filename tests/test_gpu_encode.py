"""GPU parity: the HIP encoder behind the C-ABI vs the oracle, byte for byte.

Every call goes through libsqoa_mi355x.so (seqoia_amd._lib); the oracle
(oracle/) is only the checker.  Mirrors the reference's own check
(sqoabench.c:448-455: decode(encode(px)) == px) and adds what the reference
lacks: byte-identical streams against known answers.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def sq():
    import torch

    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a HIP device; refusing to fall back to anything else")
    import seqoia_amd

    assert seqoia_amd.load().sqoa_hip_device_count() >= 1
    return seqoia_amd


@pytest.fixture(scope="module")
def orc():
    """The checker: on the GPU box it must be the compiled reference (oracle/_ref), not the restatement it would fall back to."""
    from oracle import pyoracle as po

    assert po.best().kind == "reference", "oracle/_ref/libsqoa_ref.so did not travel: GPU parity must be checked against the compiled reference"
    return po


MODES = [0, 1]


@pytest.mark.parametrize("mode", MODES)
def test_legacy_encode_kats(sq, kat_encode, mode):
    """sqoa_encode on every golden input captured from the compiled reference (incl. NULL cases)."""
    key = "qoi" if mode else "sqoa"
    for c in kat_encode:
        px = np.frombuffer(bytes.fromhex(c["pixels"]), dtype=np.uint8)
        got = sq.encode(px, sq.Desc(c["w"], c["h"], c["channels"], c["colorspace"], mode))
        want = None if c[key] is None else bytes.fromhex(c[key])
        assert got == want, (c["name"], key)


def test_legacy_encode_rejects(sq):
    """Argument checks of seqoia.h:465-480."""
    px = np.zeros(64, np.uint8)
    assert sq.encode(px, sq.Desc(0, 4, 4, 0, 0)) is None
    assert sq.encode(px, sq.Desc(4, 0, 4, 0, 0)) is None
    assert sq.encode(px, sq.Desc(4, 4, 0, 0, 0)) is None
    assert sq.encode(px, sq.Desc(4, 4, 7, 0, 0)) is None
    assert sq.encode(px, sq.Desc(4, 4, 4, 2, 0)) is None
    assert sq.encode(px, sq.Desc(20000, 20000, 4, 0, 0)) is None
    assert sq.encode(px, sq.Desc(4, 4, 1, 0, 1)) is None  # mono + qoi_compat
    assert sq.encode(px, sq.Desc(4, 4, 2, 0, 1)) is None


@pytest.mark.parametrize("mode", MODES)
def test_synth_digests(sq, orc, synth_digests, mode):
    """Encoder output FNV digests pinned from the compiled reference."""
    for c in synth_digests["images"]:
        px = orc.synth(c["w"], c["h"], c["channels"], c["cls"], c["seed"], c["img"])
        got = sq.encode(px, sq.Desc(c["w"], c["h"], c["channels"], 0, mode))
        key = "qoi" if mode else "sqoa"
        assert got is not None and len(got) == c[key + "_len"], c
        assert f"{orc.fnv1a64(got):016x}" == c[key + "_fnv"], c


def test_device_synth_matches_oracle(sq, orc):
    import torch

    for cls in ("mixed", "icon", "photo"):
        for (w, h, ch) in [(64, 64, 3), (100, 37, 4), (256, 256, 4)]:
            t = torch.empty((3, h, w, ch), dtype=torch.uint8, device="cuda")
            sq.synth(t, cls, seed=5, img0=7)
            for k in range(3):
                assert t[k].cpu().numpy().tobytes() == orc.synth(w, h, ch, cls, 5, 7 + k).tobytes(), (cls, w, h, ch, k)


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("cls,w,h,ch,n", [("mixed", 256, 256, 4, 8), ("icon", 128, 128, 3, 16), ("photo", 200, 150, 3, 4),
                                           ("mixed", 1024, 1024, 4, 4), ("icon", 512, 512, 3, 4), ("mixed", 333, 77, 4, 5),
                                           ("icon", 64, 64, 4, 64), ("mixed", 1, 1, 4, 3), ("mixed", 511, 1, 3, 2), ("mixed", 513, 3, 4, 2)])
def test_batch_encode_matches_oracle(sq, orc, mode, cls, w, h, ch, n):
    """Batch plan on device-resident synthetic images vs the oracle on the same images."""
    import torch

    px = torch.empty((n, h, w, ch), dtype=torch.uint8, device="cuda")
    sq.synth(px, cls, seed=1, img0=0)
    plan = sq.EncodePlan(px, qoi_compat=mode)
    plan.launch()
    res = plan.results()
    assert (res["status"] == 0).all()
    streams = plan.streams()
    o = orc.best()
    for k in range(n):
        want = o.encode(orc.synth(w, h, ch, cls, 1, k), w, h, ch, mode)
        assert streams[k] == want, (cls, w, h, ch, k, len(streams[k]), len(want))
    plan.close()


@pytest.mark.parametrize("mode", MODES)
def test_batch_encode_random_lowentropy(sq, orc, mode):
    """Random walks with runs, palette repeats and alpha changes: every op kind at arbitrary alignments."""
    import torch

    rng = np.random.default_rng(99)
    o = orc.best()
    for (w, h, ch) in [(640, 9, 4), (1000, 3, 3), (77, 77, 4), (2048, 2, 4)]:
        n = 6
        steps = rng.integers(-3, 4, size=(n, w * h, ch)).astype(np.int64)
        steps[rng.random((n, w * h)) < 0.5] = 0
        jump = rng.random((n, w * h)) < 0.05
        steps[jump] = rng.integers(-128, 128, size=(int(jump.sum()), ch))
        px = ((np.cumsum(steps, axis=1) + 77) % 256).astype(np.uint8)
        pal = rng.random((n, w * h)) < 0.1
        px[pal] = np.array([9, 200, 30, 255][:ch], dtype=np.uint8)
        # long runs, some longer than 512 and 62
        for k in range(n):
            a = int(rng.integers(0, w * h - 1)); b = min(w * h, a + int(rng.integers(1, 1500)))
            px[k, a:b] = px[k, a]
        t = torch.from_numpy(px.reshape(n, h, w, ch)).cuda()
        plan = sq.EncodePlan(t, qoi_compat=mode)
        plan.launch()
        assert (plan.results()["status"] == 0).all()
        for k, s in enumerate(plan.streams()):
            assert s == o.encode(px[k], w, h, ch, mode), (w, h, ch, k)
        plan.close()


def test_headline_batch_properties(sq, orc, synth_digests):
    """BASELINE config 2 at reduced count (64 x 1024x1024 RGBA): totals pinned from the reference
    for images 0-15, plus sampled byte-for-byte checks."""
    import torch

    n = 64
    px = torch.empty((n, 1024, 1024, 4), dtype=torch.uint8, device="cuda")
    sq.synth(px, "mixed", seed=1, img0=0)
    plan = sq.EncodePlan(px, qoi_compat=0)
    plan.launch()
    res = plan.results()
    assert (res["status"] == 0).all()
    assert int(res["len"][:16].sum()) == synth_digests["totals"]["mixed_1024_rgba_0_15"]["sqoa_total"]
    o = orc.best()
    for k in (0, 17, 63):
        assert plan.streams([k])[0] == o.encode(orc.synth(1024, 1024, 4, "mixed", 1, k), 1024, 1024, 4, 0)
    plan.close()


@pytest.mark.parametrize("mode", MODES)
def test_mixed_size_suite_in_one_plan(sq, orc, mode):
    """BASELINE config 5 shape at reduced count: a qoi-suite-like mix of sizes and channel counts
    (SURVEY.md Appendix C size table) in ONE plan, including mono / BGR layouts and a refused job."""
    import torch

    o = orc.best()
    shapes = [(64, 64, 4, "icon"), (512, 512, 4, "icon"), (190, 190, 4, "mixed"), (367, 368, 3, "mixed"), (768, 512, 3, "photo"),
              (553, 553, 4, "mixed"), (800, 800, 4, "mixed"), (1200, 1200, 3, "photo"), (1024, 1024, 3, "photo"), (1, 7, 4, "mixed"),
              (640, 480, 5, "mixed"), (320, 200, 6, "mixed"), (300, 300, 1, "photo"), (300, 300, 2, "mixed")]
    images, raws = [], []
    for k, (w, h, c, cls) in enumerate(shapes):
        stride = sq.stride_channels(c)
        if stride >= 3:
            raw = orc.synth(w, h, stride, cls, 9, k)
        else:  # mono layouts: take channels of a synthetic RGBA image
            raw = orc.synth(w, h, 4, cls, 9, k).reshape(-1, 4)[:, [1, 3][:stride]].reshape(-1).copy()
        raws.append(raw)
        images.append((torch.from_numpy(raw).cuda(), sq.Desc(w, h, c, 0, mode)))
    images.append((torch.zeros(16, dtype=torch.uint8, device="cuda"), sq.Desc(0, 4, 4, 0, mode)))  # invalid: width 0
    plan = sq.EncodeList(images)
    plan.launch()
    res = plan.results()
    got = plan.streams()
    for k, (w, h, c, cls) in enumerate(shapes):
        want = o.encode(raws[k], w, h, c, mode)
        if want is None:  # mono + qoi_compat (seqoia.h:477-480)
            assert res["status"][k] == 1 and got[k] is None, shapes[k]
        else:
            assert res["status"][k] == 0 and got[k] == want, shapes[k]
    assert res["status"][len(shapes)] == 1 and res["len"][len(shapes)] == 0
    plan.close()


def test_legacy_api_is_reentrant_across_threads(sq, orc):
    """The reference keeps no state between calls (seqoia.h:439), so callers may use it from any thread;
    the shim keeps one device context per thread.  16 threads encode + decode different images at once."""
    from concurrent.futures import ThreadPoolExecutor

    o = orc.best()

    def work(k):
        w, h, ch = 100 + 7 * k, 50 + k, (3, 4)[k & 1]
        px = orc.synth(w, h, ch, ("mixed", "icon", "photo")[k % 3], 4, k)
        ok = True
        for mode in (0, 1):
            enc = sq.encode(px, sq.Desc(w, h, ch, 0, mode))
            ok &= enc == o.encode(px, w, h, ch, mode)
            out, d = sq.decode(enc, 0)
            ok &= out is not None and out.tobytes() == px.tobytes()
        return ok

    with ThreadPoolExecutor(16) as ex:
        assert all(ex.map(work, range(48)))


@pytest.mark.parametrize("mode", MODES)
def test_long_runs_across_subtiles(sq, orc, mode):
    """Runs that span many 512-pixel sub-tiles (flat areas): every remainder class of the flush
    (0, 1..61, 62.., 488, 489..511 -> 1..9 bytes), the 0xfd at each 512th pixel, trailing runs,
    runs starting at the image start with the implicit (0,0,0,255) pixel."""
    import torch

    o = orc.best()
    imgs = []
    rng = np.random.default_rng(5)
    for lead in (0, 1, 3, 200, 511, 512, 513):
        for run in (511, 512, 513, 1023, 1024, 1025, 1535, 1536 + 61, 1536 + 62, 2048 + 488, 2048 + 489, 2048 + 511, 4096, 5000):
            for tail in (0, 1, 700):
                px = np.zeros((lead + run + tail, 4), np.uint8)
                px[:lead] = rng.integers(0, 256, size=(lead, 4))
                px[lead:lead + run] = (0, 0, 0, 255) if (lead == 0 and run % 2) else (17, 99, 200, 255)
                px[lead + run:] = rng.integers(0, 256, size=(tail, 4))
                if tail and (px[lead + run] == px[lead + run - 1]).all():
                    px[lead + run, 0] ^= 1
                imgs.append(px)
    images = [(torch.from_numpy(p.reshape(-1)).cuda(), sq.Desc(p.shape[0], 1, 4, 0, mode)) for p in imgs]
    plan = sq.EncodeList(images)
    plan.launch()
    res = plan.results()
    got = plan.streams()
    for k, p in enumerate(imgs):
        want = o.encode(p.reshape(-1), p.shape[0], 1, 4, mode)
        assert res["status"][k] == 0 and got[k] == want, (k, p.shape[0])
    plan.close()


def test_pack_streams_into_exact_size_blob(sq, orc):
    """the worst-case slots compacted on the device into one blob + offset table;
    the blob is the concatenation of the reference streams."""
    import torch

    o = orc.best()
    shapes = [(64, 64, 4, "icon"), (333, 77, 4, "mixed"), (200, 150, 3, "photo"), (1, 1, 4, "mixed"), (512, 512, 3, "icon"), (800, 600, 4, "mixed")]
    images, wants = [], []
    for k, (w, h, c, cls) in enumerate(shapes):
        raw = orc.synth(w, h, c, cls, 6, k)
        images.append((torch.from_numpy(raw).cuda(), sq.Desc(w, h, c, 0, k & 1)))
        wants.append(o.encode(raw, w, h, c, k & 1))
    images.insert(2, (torch.zeros(4, dtype=torch.uint8, device="cuda"), sq.Desc(0, 1, 4, 0, 0)))  # refused job: empty entry
    wants.insert(2, b"")
    plan = sq.EncodeList(images)
    plan.launch()
    blob, offsets = plan.pack()
    off = offsets.cpu().numpy()
    data = bytes(blob.cpu().numpy())
    assert off[0] == 0 and off[-1] == len(data) == sum(len(x) for x in wants)
    for i, wnt in enumerate(wants):
        assert data[off[i]:off[i + 1]] == wnt, i
    plan.close()


@pytest.mark.parametrize("mode", [0, 1])
def test_unaligned_source_and_slot_addresses(sq, orc, mode):
    """Images and output slots at odd byte addresses (sub-allocations of a caller's arena): the pixel loads
    (global -> LDS, scalar neighbour loads) and the stream stores must not assume any alignment."""
    import torch

    o = orc.best()
    arena = torch.zeros(1 << 22, dtype=torch.uint8, device="cuda")
    images, raws, at = [], [], 0
    for k, (w, h, c, off) in enumerate([(300, 7, 4, 1), (64, 40, 4, 2), (300, 7, 3, 3), (64, 40, 3, 5), (1024, 3, 4, 17),
                                         (1024, 3, 3, 6), (129, 9, 2, 1), (129, 9, 1, 3), (512, 4, 4, 4), (512, 4, 3, 8)]):
        if c >= 3:
            raw = orc.synth(w, h, c, ("mixed", "icon", "photo")[k % 3], 9, k)
        else:  # mono: smooth ramp + sparse noise, so that LUMA, raw and run ops all occur
            rng = np.random.default_rng(k)
            raw = ((np.arange(w * h * c) // 5) + rng.integers(0, 2, w * h * c) * rng.integers(0, 40, w * h * c)).astype(np.uint8)
        at = (at + 63) // 64 * 64 + off
        view = arena[at:at + raw.size]
        view.copy_(torch.from_numpy(raw))
        assert view.data_ptr() % 64 == off
        at += raw.size
        images.append((view, sq.Desc(w, h, c, 0, mode)))
        raws.append((raw, w, h, c))
    plan = sq.EncodeList(images)
    plan.launch()
    res = plan.results()
    got = plan.streams()
    for k, (raw, w, h, c) in enumerate(raws):
        want = o.encode(raw, w, h, c, mode)
        if want is None:
            assert res["status"][k] == 1
        else:
            assert res["status"][k] == 0 and got[k] == want, (k, w, h, c)
    plan.close()


def test_padded_batch_arena_matches_contiguous_batch(sq, orc):
    """A batch whose images are spaced further apart than their size (a strided [n, h, w, c] view of an arena):
    streams and decoded pixels are those of the contiguous batch (and of the oracle), and the bytes between the
    images are never written."""
    import torch

    n, h, w, c, gap = 5, 128, 128, 4, 4096
    img = h * w * c

    def arena():
        a = torch.full((n * (img + gap),), 0xA5, dtype=torch.uint8, device="cuda")
        return a, a.as_strided((n, h, w, c), (img + gap, w * c, c, 1)), a.as_strided((n, gap), (img + gap, 1), img)

    flat = torch.empty((n, h, w, c), dtype=torch.uint8, device="cuda")
    sq.synth(flat, "mixed", 3, 0)
    _, pad, guard = arena()
    sq.synth(pad, "mixed", 3, 0)
    assert torch.equal(pad, flat) and not pad.is_contiguous()
    o = orc.best()
    for mode in (0, 1):
        a, b = sq.EncodePlan(flat, qoi_compat=mode), sq.EncodePlan(pad, qoi_compat=mode)
        a.launch(); b.launch()
        sa, sb = a.streams(), b.streams()
        assert sa == sb and sa[0] == o.encode(flat[0].cpu().numpy().reshape(-1), w, h, c, mode)
        _, out, gout = arena()
        dec = sq.DecodePlan(b.out, b.results()["len"], w, h, c, out=out)
        dec.launch()
        assert (dec.results()["status"] == 0).all() and torch.equal(dec.out, flat)
        assert (gout == 0xA5).all() and (guard == 0xA5).all()


def test_one_shot_batch_entry_points(sq, orc):
    """sqoa_hip_encode_batch / sqoa_hip_decode_batch (array of descriptors and device buffers in, slots
    and lengths out) give what the plan API gives: byte-identical streams, exact round trip, refused jobs flagged."""
    import ctypes as C
    import torch
    from seqoia_amd._lib import Job, Result, load

    lib, o = load(), orc.best()
    shapes = [(64, 64, 4, "icon", 0), (200, 150, 3, "photo", 1), (333, 77, 4, "mixed", 0), (0, 4, 4, "mixed", 0)]
    raws, pix, slots, jobs = [], [], [], (Job * len(shapes))()
    for k, (w, h, c, cls, mode) in enumerate(shapes):
        raw = orc.synth(max(w, 1), h, c, cls, 5, k)
        d = sq.Desc(w, h, c, 0, mode)
        t = torch.from_numpy(raw).cuda()
        sl = torch.zeros(int(sq.encode_bound(sq.Desc(max(w, 1), h, c, 0, mode))), dtype=torch.uint8, device="cuda")
        raws.append(raw); pix.append(t); slots.append(sl)
        jobs[k] = Job(t.data_ptr(), sl.data_ptr(), 0, sl.numel(), d, 0)
    res = (Result * len(shapes))()
    assert lib.sqoa_hip_encode_batch(0, jobs, len(shapes), res) == 0
    assert res[3].status == 1 and res[3].len == 0  # width 0: refused like sqoa_encode (seqoia.h:465-473)
    outs, djobs = [], (Job * 3)()
    for k, (w, h, c, cls, mode) in enumerate(shapes[:3]):
        assert res[k].status == 0
        assert bytes(slots[k][: res[k].len].cpu().numpy()) == o.encode(raws[k], w, h, c, mode), k
        out = torch.zeros(w * h * c, dtype=torch.uint8, device="cuda")
        outs.append(out)
        djobs[k] = Job(slots[k].data_ptr(), out.data_ptr(), res[k].len, out.numel(), sq.Desc(0, 0, c, 0, 0), 0)
    dres = (Result * 3)()
    assert lib.sqoa_hip_decode_batch(0, djobs, 3, dres) == 0
    for k, (w, h, c, cls, mode) in enumerate(shapes[:3]):
        assert dres[k].status == 0 and dres[k].len == w * h * c and dres[k].desc.qoi_compat == mode
        assert torch.equal(outs[k].cpu(), torch.from_numpy(raws[k])), k
    assert lib.sqoa_hip_encode_batch(0, None, 0, None) == 0 and lib.sqoa_hip_encode_batch(0, None, 1, None) != 0


@pytest.mark.parametrize("mode", [0, 1])
def test_repeated_launches_are_bit_identical(sq, orc, mode):
    """Race detector for the wave-to-wave chains (run length, output position, colour index): the same plan launched over
    and over must write the same bytes every time, and decode back to the pixels every time."""
    import torch

    n, s = 128, 256
    px = torch.empty((n, s, s, 4), dtype=torch.uint8, device="cuda")
    sq.synth(px, "mixed", 7, 0)
    enc = sq.EncodePlan(px, qoi_compat=mode)
    enc.out.zero_(); torch.cuda.synchronize()
    enc.launch(); res0 = enc.results()
    ref = enc.out.clone()
    assert bytes(ref[0, : int(res0["len"][0])].cpu().numpy()) == orc.best().encode(orc.synth(s, s, 4, "mixed", 7, 0), s, s, 4, mode)
    dec = sq.DecodePlan(ref, res0["len"], s, s, 4)
    for _ in range(12):
        enc.out.zero_(); dec.out.zero_(); torch.cuda.synchronize()
        enc.launch(); res = enc.results()
        assert (res["len"] == res0["len"]).all() and torch.equal(enc.out, ref)
        dec.launch()
        assert (dec.results()["status"] == 0).all() and torch.equal(dec.out, px)
    dec.close(); enc.close()


@pytest.mark.parametrize("mode", [0, 1])
@pytest.mark.parametrize("ch", [3, 4])
def test_l8_hot_loop_runs_of_every_length(sq, orc, ch, mode):
    """The hot loop of colour images gives every lane eight consecutive pixels; runs then start, end and cross the 512th
    repeat at every position inside a lane, across lanes and across sub-tiles (seqoia.h:544-561).  Images large enough
    that nearly all sub-tiles take that loop, filled with runs of random lengths (1 .. 1400, dense around the 61 / 62 /
    511 / 512 boundaries) separated by one to three literal pixels of every op kind."""
    import torch

    rng = np.random.default_rng(1234 + ch + 10 * mode)   # (qoi_compat=1: runs are cut every 62 pixels, seqoia.h:411)
    o = orc.best()
    w, h, n = 512, 160, 6
    imgs = []
    for k in range(n):
        px = np.zeros((w * h, ch), dtype=np.uint8)
        p = 0
        cur = rng.integers(0, 256, size=ch)
        while p < w * h:
            kind = rng.integers(0, 4)
            if kind == 0:
                ln = int(rng.integers(1, 10))
            elif kind == 1:
                ln = int(rng.choice([60, 61, 62, 63, 64, 122, 123, 124, 510, 511, 512, 513, 514, 1023, 1024, 1025])) + int(rng.integers(-1, 2))
            elif kind == 2:
                ln = int(rng.integers(1, 1400))
            else:
                ln = int(rng.integers(1, 70))
            px[p:p + ln] = cur
            p += ln
            for _ in range(int(rng.integers(1, 4))):  # literals: small delta (LUMA), alpha step (ALPHA suffix), jump (RGB / RGBA)
                if p >= w * h:
                    break
                t = rng.integers(0, 4)
                if t == 0:
                    cur = (cur + rng.integers(-3, 4, size=ch)) % 256
                elif t == 1 and ch == 4:
                    cur = cur.copy(); cur[3] = (cur[3] + rng.integers(-16, 16)) % 256
                elif t == 2 and ch == 4:
                    cur = rng.integers(0, 256, size=ch)
                else:
                    cur = cur.copy(); cur[:3] = rng.integers(0, 256, size=3)
                px[p] = cur
                p += 1
        imgs.append(px.reshape(-1))
    t = torch.from_numpy(np.stack(imgs).reshape(n, h, w, ch)).cuda()
    plan = sq.EncodePlan(t, qoi_compat=mode)
    plan.launch()
    assert (plan.results()["status"] == 0).all()
    for k, s in enumerate(plan.streams()):
        want = o.encode(imgs[k], w, h, ch, mode)
        assert s == want, (ch, k, len(s), len(want), next((i for i in range(min(len(s), len(want))) if s[i] != want[i]), None))
    plan.close()


@pytest.mark.parametrize("ch", [3, 4])
def test_qoi_l8_hot_loop_index_hits_everywhere(sq, orc, ch):
    """QOI-compat, the L8 hot loop: few colours that recur at every distance -- inside a lane's eight pixels, across lanes, across
    the 64-pixel rows the probe walks, across sub-tiles and waves -- so that most ops are QOI_OP_INDEX hits and the slots are
    overwritten by colliding colours now and then (seqoia.h:563-582); with (0,0,0,0), whose slot 0 the zeroed table holds from
    the start, and with short runs and small deltas (DIFF / LUMA) mixed in."""
    import torch

    o = orc.best()
    w, h, n = 512, 96, 8
    imgs = []
    for k in range(n):
        rng = np.random.default_rng(4321 + 17 * k + ch)
        ncol = [2, 3, 5, 9, 17, 40, 70, 200][k]
        pal = rng.integers(0, 256, size=(ncol, ch), dtype=np.uint8)
        if k % 2 == 0:
            pal[0] = 0                                              # (0,0,0[,0])
        idx = rng.integers(0, ncol, size=w * h)
        px = pal[idx].astype(np.int32)
        wob = rng.integers(0, 8, size=w * h)                        # small deltas on some pixels: DIFF / LUMA between the hits
        px[:, 0] = (px[:, 0] + (wob == 1) - (wob == 2)) % 256
        px[:, 1] = (px[:, 1] + 3 * (wob == 3)) % 256
        rep = rng.integers(0, 10, size=w * h) == 0                  # short runs
        for i in np.nonzero(rep)[0]:
            px[i:i + int(rng.integers(1, 9))] = px[i]
        imgs.append(px.astype(np.uint8).reshape(-1))
    t = torch.from_numpy(np.stack(imgs).reshape(n, h, w, ch)).cuda()
    plan = sq.EncodePlan(t, qoi_compat=1)
    for _ in range(2):
        plan.launch()
    assert (plan.results()["status"] == 0).all()
    for k, s in enumerate(plan.streams()):
        want = o.encode(imgs[k], w, h, ch, 1)
        assert s == want, (ch, k, len(s), len(want), next((i for i in range(min(len(s), len(want))) if s[i] != want[i]), None))
    plan.close()


def test_legacy_encode_strip_jobs(sq, orc):
    """sqoa_encode cuts a large SQOA-native image into strip jobs (several workgroups; sqoa_hip.h SQOA_HIP_JOB_*): cuts move
    forward to a pixel that differs from its predecessor, the strips' streams are concatenated behind the header.  Byte
    identity with the reference on images whose runs sit exactly where the nominal cuts fall, on a one-colour image (every cut
    runs to the end), on 3-channel, mono and BGR layouts, and through the strip flags on the batch API directly."""
    import torch

    from seqoia_amd import Desc, Job
    from seqoia_amd._lib import ENCODE

    o = orc.best()
    rng = np.random.default_rng(77)
    w, h = 640, 480  # 307 200 pixels: 9 nominal strips
    cases = []
    for ch in (4, 3, 2, 1, 6):
        stride = sq.stride_channels(ch)
        px = rng.integers(0, 256, size=(w * h, stride), dtype=np.uint8)
        # runs across every nominal cut position k/9 (and a few elsewhere), of lengths around the 61 / 512 boundaries
        for k in range(1, 9):
            a = w * h * k // 9 - int(rng.integers(0, 700))
            px[a:a + int(rng.choice([3, 62, 63, 511, 512, 513, 1500, 40000]))] = px[a]
        cases.append((f"runs_at_cuts_ch{ch}", px.reshape(-1), ch))
    cases.append(("one_colour", np.full(w * h * 4, 37, dtype=np.uint8), 4))
    cases.append(("default_pixel", np.tile(np.array([0, 0, 0, 255], dtype=np.uint8), w * h), 4))  # repeats the implicit start pixel throughout
    grad = (np.arange(w * h, dtype=np.uint32)[:, None] >> np.array([3, 9, 14, 30], dtype=np.uint32)).astype(np.uint8)
    cases.append(("gradient_with_long_runs", grad.reshape(-1), 4))
    for name, px, ch in cases:
        got = sq.encode(px, sq.Desc(w, h, ch, 0, 0))
        assert got == o.encode(px, w, h, ch, 0), name
        out, d = sq.decode(got, 0)
        assert out.tobytes() == px.tobytes(), name
    # the flags on the batch API: two strips of one image, cut by hand at a pixel that differs from its predecessor
    name, px, ch = cases[0]
    pxs = px.reshape(-1, 4)
    cut = w * h // 2
    while (pxs[cut] == pxs[cut - 1]).all():
        cut += 1
    t = torch.from_numpy(px).cuda()
    out = torch.zeros((2, sq.encode_bound(sq.Desc(w * h, 1, 4, 0, 0))), dtype=torch.uint8, device="cuda")
    jobs = [Job(t.data_ptr(), out[0].data_ptr(), 0, out.shape[1], Desc(cut, 1, 4, 0, 0), 1 | 2),
            Job(t.data_ptr() + cut * 4, out[1].data_ptr(), 0, out.shape[1], Desc(w * h - cut, 1, 4, 0, 0), 1 | 4),
            Job(t.data_ptr(), out[0].data_ptr(), 0, out.shape[1], Desc(cut, 1, 4, 0, 1), 1 | 2)]  # strip flags with qoi_compat: refused
    plan = sq.Plan(ENCODE, jobs)
    plan.launch()
    res = plan.results()
    assert res["status"].tolist() == [0, 0, 1]
    want = o.encode(px, w, h, 4, 0)
    body = bytes(out[0, : int(res["len"][0])].cpu().numpy()) + bytes(out[1, : int(res["len"][1])].cpu().numpy())
    assert want[:15] + body == want
    plan.close()


@pytest.mark.parametrize("knob", ["1", ""])
def test_batch_encode_cuts_large_images_into_strips_on_the_device(sq, orc, monkeypatch, knob):
    """A plan with few large images cuts them into strips on the device (sqoa_strips.hip): the cut search (forward to a
    pixel that differs from its predecessor, up to the next nominal cut), cuts that do not exist (one-colour stretches),
    the strips' streams packed behind the header at the image's own slot.  Byte identity with the reference; the same
    plan with SQOA_HIP_STRIPS=0 gives the same bytes."""
    import torch

    o = orc.best()
    rng = np.random.default_rng(78)
    w, h = 640, 480  # 307 200 pixels: 18 nominal strips of 16 384 when forced
    cases = []
    for ch in (4, 3, 2, 1, 6):
        stride = sq.stride_channels(ch)
        px = rng.integers(0, 256, size=(w * h, stride), dtype=np.uint8)
        for k in range(1, 18):  # runs across every nominal cut position (and a few elsewhere), around the 61 / 512 boundaries
            a = w * h * k // 18 - int(rng.integers(0, 700))
            px[a:a + int(rng.choice([3, 62, 63, 511, 512, 513, 1500, 40000]))] = px[a]
        cases.append((f"runs_at_cuts_ch{ch}", px.reshape(-1), ch))
    cases.append(("one_colour", np.full(w * h * 4, 37, dtype=np.uint8), 4))
    cases.append(("default_pixel", np.tile(np.array([0, 0, 0, 255], dtype=np.uint8), w * h), 4))
    grad = (np.arange(w * h, dtype=np.uint32)[:, None] >> np.array([3, 9, 14, 30], dtype=np.uint32)).astype(np.uint8)
    cases.append(("gradient_with_long_runs", grad.reshape(-1), 4))
    half = np.concatenate([np.full(w * h * 2, 200, dtype=np.uint8), rng.integers(0, 256, size=w * h * 2, dtype=np.uint8)])
    cases.append(("flat_then_noise", half, 4))
    cases.append(("icon_tile", orc.synth(w, h, 3, "icon", 9, 1), 3))
    cases.append(("small_next_to_large", orc.synth(64, 64, 4, "mixed", 9, 2), 4))   # (too small to cut: one workgroup)
    wants = [o.encode(px, (64 if name.startswith("small") else w), (64 if name.startswith("small") else h), ch, 0) for name, px, ch in cases]
    got = {}
    for setting in (knob, "0"):
        monkeypatch.setenv("SQOA_HIP_STRIPS", setting)
        images = [(torch.from_numpy(px).cuda(), sq.Desc(64 if name.startswith("small") else w, 64 if name.startswith("small") else h, ch, 0, 0))
                  for name, px, ch in cases]
        plan = sq.EncodeList(images)
        for _ in range(2):  # (a second launch of the same plan: the strip tables are rebuilt on the device every time)
            plan.launch()
        res = plan.results()
        got[setting] = plan.streams()
        assert (res["status"] == 0).all(), (setting, res["status"].tolist())
        blob, offsets = plan.pack()
        assert int(offsets[-1]) == sum(len(x) for x in wants)
        plan.close()
    for k, (name, px, ch) in enumerate(cases):
        assert got[knob][k] == wants[k], (name, "strips")
        assert got["0"][k] == wants[k], (name, "one workgroup")


def test_plan_update_switches_between_the_routes(sq, orc):
    """sqoa_hip_plan_update on one encode and one decode plan: a lone large image (strips on the device / three chip-wide
    passes), many small ones (one workgroup each), a mix, and a large one again -- the tables of the chip-wide routes are
    rebuilt (and grown) by every update.  Every stream and every decoded image against the reference."""
    import torch

    from seqoia_amd import Desc, Job
    from seqoia_amd._lib import DECODE, ENCODE

    o = orc.best()
    cap = 24
    rounds = [[(640, 480, 4, "mixed")],
              [(64, 48, 3 + (k & 1), ["icon", "mixed", "photo"][k % 3]) for k in range(cap)],
              [(800, 600, 4, "photo"), (512, 512, 3, "icon")] + [(40, 40, 4, "mixed")] * 6,
              [(1024, 768, 3, "mixed")]]
    slot = sq.encode_bound(Desc(1024, 768, 4, 0, 0))
    src = torch.empty((cap, 1024 * 768 * 4), dtype=torch.uint8, device="cuda")
    enc_out = torch.zeros((cap, slot), dtype=torch.uint8, device="cuda")
    dec_out = torch.zeros((cap, 1024 * 768 * 4), dtype=torch.uint8, device="cuda")
    dummy = [Job(src[i].data_ptr(), enc_out[i].data_ptr(), 0, slot, Desc(8, 8, 4, 0, 0), 0) for i in range(cap)]
    enc = sq.Plan(ENCODE, dummy)
    dec = sq.Plan(DECODE, [Job(enc_out[i].data_ptr(), dec_out[i].data_ptr(), 64, 1024 * 768 * 4, Desc(0, 0, 0, 0, 0), 0) for i in range(cap)])
    for r, shapes in enumerate(rounds):
        raws = []
        for k, (w, h, ch, cls) in enumerate(shapes):
            raw = orc.synth(w, h, ch, cls, 21 + r, k)
            src[k, : raw.size] = torch.from_numpy(raw).cuda()
            raws.append(raw)
        enc.update([Job(src[k].data_ptr(), enc_out[k].data_ptr(), 0, slot, Desc(w, h, ch, 0, 0), 0) for k, (w, h, ch, _) in enumerate(shapes)])
        enc.launch()
        res = enc.results()
        assert (res["status"] == 0).all(), (r, res["status"].tolist())
        for k, (w, h, ch, _) in enumerate(shapes):
            assert bytes(enc_out[k, : int(res["len"][k])].cpu().numpy()) == o.encode(raws[k], w, h, ch, 0), (r, k)
        dec.update([Job(enc_out[k].data_ptr(), dec_out[k].data_ptr(), int(res["len"][k]), w * h * ch, Desc(w, h, ch, 0, 0), 0)
                    for k, (w, h, ch, _) in enumerate(shapes)])
        dec.launch()
        dres = dec.results()
        assert (dres["status"] == 0).all(), (r, dres["status"].tolist())
        for k, (w, h, ch, _) in enumerate(shapes):
            assert bytes(dec_out[k, : w * h * ch].cpu().numpy()) == raws[k].tobytes(), (r, k)
    with pytest.raises(Exception):
        enc.update(dummy + dummy[:1])   # more jobs than the plan was created for
    enc.close()
    dec.close()


@pytest.mark.parametrize("mode", [0, 1])
def test_many_small_images_get_one_wave_each(sq, orc, mode):
    """A plan with at least 512 images of at most 16 384 pixels encodes those with one wave per image
    (encode_kernel<1, ...>), the larger ones of the same plan with eight: every stream against the reference, in the job
    order the caller gave (the plan's own order is largest first, small ones last)."""
    import torch

    o = orc.best()
    rng = np.random.default_rng(31 + mode)
    shapes = []
    for k in range(640):
        w, h = int(rng.integers(1, 129)), int(rng.integers(1, 129))
        shapes.append((w, h, (3, 4)[k & 1], ["icon", "mixed", "photo"][k % 3]))
    shapes[5] = (300, 200, 4, "mixed")       # larger than the small shape: eight waves
    shapes[77] = (640, 480, 3, "photo")      # large enough to be cut into strips when it is SQOA-native
    shapes[300] = (128, 128, 4, "icon")      # exactly at the limit
    raws = [orc.synth(w, h, ch, cls, 41, k) for k, (w, h, ch, cls) in enumerate(shapes)]
    images = [(torch.from_numpy(raw).cuda(), sq.Desc(w, h, ch, 0, mode)) for raw, (w, h, ch, _) in zip(raws, shapes)]
    plan = sq.EncodeList(images)
    plan.launch()
    res = plan.results()
    got = plan.streams()
    assert (res["status"] == 0).all()
    for k, (w, h, ch, _) in enumerate(shapes):
        assert got[k] == o.encode(raws[k], w, h, ch, mode), (k, w, h, ch)
    plan.close()


@pytest.mark.parametrize("knob", ["1", ""])
def test_qoi_compat_images_are_cut_into_strips_with_the_colour_index_handed_on(sq, orc, monkeypatch, knob):
    """QOI-compat images take the strip route too (sqoa_strips.hip): the 64-entry colour index (seqoia.h:563-582) crosses every
    cut -- per strip the last non-run pixel per slot, composed by overwrite in strip order, each strip encoded from the index
    entering it.  Content that lives off the index (few colours recurring across the whole image, so that most ops are
    QOI_OP_INDEX hits on pixels written many strips earlier), long runs across the nominal cuts (62-pixel run chunks), RGBA
    and RGB, next to SQOA-native images in the same plan.  Byte identity with the reference, and with the one-workgroup route."""
    import torch

    o = orc.best()
    rng = np.random.default_rng(79)
    w, h = 640, 480
    cases = []
    for ch in (4, 3):
        pal = rng.integers(0, 256, size=(23, ch), dtype=np.uint8)                      # 23 colours, re-used everywhere: INDEX hits across strips
        idx = rng.integers(0, 23, size=w * h)
        px = pal[idx]
        for k in range(1, 18):
            a = w * h * k // 18 - int(rng.integers(0, 300))
            px[a:a + int(rng.choice([3, 61, 62, 63, 124, 125, 1500, 40000]))] = px[a]
        cases.append((f"palette_ch{ch}", px.reshape(-1).copy(), ch, 1))
        cases.append((f"noise_ch{ch}", rng.integers(0, 256, size=w * h * ch, dtype=np.uint8), ch, 1))
    cases.append(("one_colour", np.full(w * h * 4, 37, dtype=np.uint8), 4, 1))
    cases.append(("default_pixel", np.tile(np.array([0, 0, 0, 255], dtype=np.uint8), w * h), 4, 1))
    cases.append(("zero_pixel", np.zeros(w * h * 4, dtype=np.uint8), 4, 1))            # (0,0,0,0) hashes to slot 0, whose initial content it equals
    cases.append(("mixed", orc.synth(w, h, 4, "mixed", 11, 0), 4, 1))
    cases.append(("icon", orc.synth(w, h, 3, "icon", 11, 1), 3, 1))
    cases.append(("sqoa_neighbour", orc.synth(w, h, 4, "mixed", 11, 2), 4, 0))          # an SQOA-native image in the same plan
    cases.append(("sqoa_icon_neighbour", orc.synth(w, h, 3, "icon", 11, 3), 3, 0))
    wants = [o.encode(px, w, h, ch, mode) for name, px, ch, mode in cases]
    for setting in (knob, "0"):
        monkeypatch.setenv("SQOA_HIP_STRIPS", setting)
        plan = sq.EncodeList([(torch.from_numpy(px).cuda(), sq.Desc(w, h, ch, 0, mode)) for name, px, ch, mode in cases])
        for _ in range(2):
            plan.launch()
        res = plan.results()
        got = plan.streams()
        assert (res["status"] == 0).all(), (setting, res["status"].tolist())
        for k, (name, px, ch, mode) in enumerate(cases):
            assert got[k] == wants[k], (name, "strips" if setting != "0" else "one workgroup")
        plan.close()


@pytest.mark.parametrize("mode", [0, 1])
@pytest.mark.parametrize("ch", [3, 4])
def test_large_single_image(sq, orc, mode, ch):
    """What `sqoaconv x.png y.qoi` / `y.sqoa` produces (sqoaconv.c:84-90): ONE screenshot-sized image, through the legacy call and
    as a plan of one job -- both wire formats take the whole chip (strips; QOI-compat with the colour index handed on)."""
    import torch

    w, h = 3840, 2160
    t = torch.empty((1, h, w, ch), dtype=torch.uint8, device="cuda")
    sq.synth(t, "mixed", seed=4, img0=7)
    px = t.cpu().numpy().reshape(-1)
    want = orc.best().encode(px, w, h, ch, mode)
    enc = sq.EncodePlan(t, qoi_compat=mode)
    enc.launch()
    res = enc.results()
    assert res["status"][0] == 0 and int(res["len"][0]) == len(want)
    assert enc.streams()[0] == want
    dec = sq.DecodePlan(enc.out, res["len"], w, h, ch)
    dec.launch()
    assert (dec.results()["status"] == 0).all() and torch.equal(dec.out, t)
    dec.close()
    enc.close()
    assert sq.encode(px, sq.Desc(w, h, ch, 0, mode)) == want


def test_legacy_calls_take_their_buffers_from_the_callers_allocator(sq, orc, tmp_path):
    """sqoa_hip_set_allocator: the run-time counterpart of the reference's compile-time SQOA_MALLOC / SQOA_FREE
    (seqoia.h:390-396).  Every buffer the four legacy entry points return -- and the file buffer inside sqoa_read, seqoia.h:849 /
    :864; the stream sqoa_write encodes, seqoia.h:823 / :834 -- comes from the caller's malloc and goes back through the
    caller's free; the bytes are the reference's."""
    import ctypes as C

    from seqoia_amd import _lib

    lib = _lib.load()
    libc = C.CDLL(None)
    libc.malloc.restype, libc.malloc.argtypes = C.c_void_p, [C.c_size_t]
    libc.free.restype, libc.free.argtypes = None, [C.c_void_p]
    live, log = {}, []

    @C.CFUNCTYPE(C.c_void_p, C.c_size_t)
    def my_malloc(n):
        p = libc.malloc(n)
        live[p] = n
        log.append(("malloc", n))
        return p

    @C.CFUNCTYPE(None, C.c_void_p)
    def my_free(p):
        assert p in live, "freed what it never allocated"
        log.append(("free", live.pop(p)))
        libc.free(p)

    w, h, ch = 96, 33, 4
    px = orc.synth(w, h, ch, "mixed", 5, 0)
    want = orc.best().encode(px, w, h, ch, 0)
    lib.sqoa_hip_set_allocator(C.cast(my_malloc, C.c_void_p), C.cast(my_free, C.c_void_p))
    try:
        d, n, dd = sq.Desc(w, h, ch, 0, 0), C.c_int(0), sq.Desc()
        p = lib.sqoa_encode(px.ctypes.data, C.byref(d), C.byref(n))
        assert p in live and live[p] == n.value == len(want) and C.string_at(p, n.value) == want
        q = lib.sqoa_decode(p, n.value, C.byref(dd), 0)
        assert q in live and live[q] == w * h * ch and C.string_at(q, w * h * ch) == px.tobytes()
        my_free(p)
        my_free(q)   # (the caller releases what it was handed with the same allocator)
        assert not live
        f = str(tmp_path / "a.sqoa").encode()
        del log[:]
        assert lib.sqoa_write(f, px.ctypes.data, C.byref(d)) == len(want)
        assert log == [("malloc", len(want)), ("free", len(want))] and not live and open(f, "rb").read() == want
        del log[:]
        r = lib.sqoa_read(f, C.byref(dd), 4)
        assert r in live and C.string_at(r, w * h * 4) == px.tobytes()
        assert log == [("malloc", len(want)), ("malloc", w * h * 4), ("free", len(want))]
        my_free(r)
    finally:
        lib.sqoa_hip_set_allocator(None, None)
    # back on malloc / free
    p = lib.sqoa_encode(px.ctypes.data, C.byref(d), C.byref(n))
    assert p and p not in live and C.string_at(p, n.value) == want
    _lib.c_free(p)


def test_plans_run_on_the_last_visible_device(sq, orc):
    """Multi-GPU hygiene (one process per GPU, the images sharded, no collective): nothing in a plan may assume device 0.
    Encode + decode plans on device `device_count() - 1` while the calling thread's current device is 0 -- both wire formats,
    the chip-wide routes included (a lone large image), every result the reference's.  (On a one-GPU box the last device is
    device 0 and this is one more parity test; on any larger box it is the test of everything that takes a device index.)"""
    import torch

    o = orc.best()
    last = torch.cuda.device_count() - 1
    assert sq.load().sqoa_hip_device_count() == torch.cuda.device_count()
    torch.cuda.set_device(0)
    dev = torch.device("cuda", last)
    for (n, w, h, ch) in ((6, 160, 120, 4), (1, 1024, 700, 3)):
        host = np.stack([orc.synth(w, h, ch, "mixed", 9, k).reshape(h, w, ch) for k in range(n)])
        px = torch.from_numpy(host).to(dev)
        for mode in (0, 1):
            enc = sq.EncodePlan(px, qoi_compat=mode)
            assert enc.device == last
            enc.launch()
            res = enc.results()
            assert (res["status"] == 0).all()
            for k in range(n):
                assert bytes(enc.out[k, : int(res["len"][k])].cpu().numpy()) == o.encode(host[k].reshape(-1), w, h, ch, mode), (n, mode, k)
            dec = sq.DecodePlan(enc.out, res["len"], w, h, 4)
            dec.launch()
            dres = dec.results()
            assert (dres["status"] == 0).all() and dec.out.device == dev
            assert np.array_equal(dec.out[..., :ch].cpu().numpy(), host)
            dec.close()
            enc.close()


@pytest.mark.parametrize("strips", ["0", "1"])
@pytest.mark.parametrize("mode", [0, 1])
def test_a_relaunched_encode_plan_reports_the_new_pixels(sq, orc, monkeypatch, strips, mode):
    """The encoder resets its result rows itself (thread 0 of an image's workgroup; the first workgroup of the strip pack): a plan
    launched again over the same pixel buffer with other pixels in it -- no plan_update in between -- reports the new streams' lengths
    and bytes, per image, as one workgroup per image and as strips cut on the device; and many small images on one wave each."""
    import torch

    monkeypatch.setenv("SQOA_HIP_STRIPS", strips)
    o = orc.best()
    for n, w, h, ch in ((3, 320, 200, 4), (600, 40, 30, 3)):
        px = torch.empty((n, h, w, ch), dtype=torch.uint8, device="cuda")
        enc = sq.EncodePlan(px, qoi_compat=mode)
        for cls, seed in (("mixed", 3), ("icon", 4), ("photo", 5), ("mixed", 3)):
            sq.synth(px, cls, seed=seed, img0=0)
            torch.cuda.synchronize()
            enc.launch()
            res = enc.results()
            assert (res["status"] == 0).all()
            sample = list(range(n)) if n <= 8 else [0, 1, n // 2, n - 1]
            for k, got in zip(sample, enc.streams(sample)):
                want = o.encode(orc.synth(w, h, ch, cls, seed, k), w, h, ch, mode)
                assert int(res["len"][k]) == len(want) and got == want, (cls, k)
        enc.close()
