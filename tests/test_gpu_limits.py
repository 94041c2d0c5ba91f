"""GPU parity at the reference's size limits (seqoia.h:428-432 SQOA_PIXELS_MAX = 400 000 000, the guards at :470 and :685, `int` stream
lengths up to 2.0e9): one image of 256 Mpixel and one just under 400 Mpixel, byte-identical to the compiled reference, through a plan
and through the legacy calls; the accept / reject pair around `height >= 400000000 / width`.  Sizes between the largest image of the
other tests (3840 x 2160) and these are where 32-bit positions, byte offsets and slot sizes would first go wrong."""
import numpy as np
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
    from oracle import pyoracle as po

    assert po.best().kind == "reference", "oracle/_ref/libsqoa_ref.so did not travel: GPU parity must be checked against the compiled reference"
    return po


def _device_equal(torch, t, want_bytes):
    """t (uint8 device tensor) == want_bytes, compared on the device in pieces (no second gigabyte on the host)."""
    want = np.frombuffer(want_bytes, dtype=np.uint8)
    assert t.numel() == want.size, (t.numel(), want.size)
    step = 1 << 28
    for a in range(0, want.size, step):
        if not torch.equal(t[a : a + step], torch.from_numpy(want[a : a + step].copy()).cuda()):
            return False
    return True


@pytest.mark.parametrize("w,h,ch,cls", [(16384, 16384, 4, "icon"), (19999, 20000, 3, "mixed")], ids=["256Mpx-rgba-icon", "399.98Mpx-rgb-mixed"])
def test_images_at_the_pixel_limit(sq, orc, w, h, ch, cls):
    """One image of 268 435 456 pixels (RGBA, class icon: RUN / BIGRUN heavy) and one of 399 980 000 (RGB, class mixed: 1.2 GB of pixels,
    a stream of about a gigabyte -- `int` lengths near their end): the plan's stream is the reference's, the decode gives the pixels
    back, to the stream's own channel count and to four; the legacy sqoa_encode / sqoa_decode on host buffers the same."""
    import torch

    o = orc.best()
    t = torch.empty((1, h, w, ch), dtype=torch.uint8, device="cuda")
    sq.synth(t, cls, seed=5, img0=3)
    px = t.cpu().numpy().reshape(-1)
    want = o.encode(px, w, h, ch, 0)
    assert want is not None
    enc = sq.EncodePlan(t, qoi_compat=0)
    enc.launch()
    res = enc.results()
    assert res["status"][0] == 0 and int(res["len"][0]) == len(want), (res["status"][0], int(res["len"][0]), len(want))
    assert _device_equal(torch, enc.out.reshape(-1)[: len(want)], want)
    for oc in sorted({ch, 4}):
        dec = sq.DecodePlan(enc.out, res["len"], w, h, oc)
        dec.launch()
        dres = dec.results()
        assert dres["status"][0] == 0 and int(dres["len"][0]) == w * h * oc
        if oc == ch:
            assert torch.equal(dec.out, t)
        else:
            assert torch.equal(dec.out[..., :ch], t) and bool((dec.out[..., 3] == 255).all())
        dec.close()
    enc.close()
    del t
    torch.cuda.empty_cache()
    # the legacy calls (one H2D, the plan of one job, one exact-size D2H; buffers from malloc)
    got = sq.encode(px, sq.Desc(w, h, ch, 0, 0))
    assert got == want
    out, d = sq.decode(want, ch)
    assert out is not None and (d.width, d.height, d.channels) == (w, h, ch)
    assert np.array_equal(out, px)


def test_the_pixel_limit_itself(sq, orc):
    """seqoia.h:470 / :685: `height >= SQOA_PIXELS_MAX / width` is refused -- 19999 x 20001 (400 000 000 / 19999 = 20001 in integers) is,
    19999 x 20000 is not (the test above) -- by the legacy encode (NULL before a pixel is read), by a plan (BADARG row), and by both
    decoders for a header that says so."""
    import torch

    o = orc.best()
    w, h = 19999, 20001
    assert 400000000 // w == h
    one = np.zeros(4, dtype=np.uint8)   # (refused on the descriptor alone: no 1.2 GB buffer needed)
    assert o.encode(np.zeros(w * 3, dtype=np.uint8), w, 1, 3, 0) is not None
    assert sq.encode(one, sq.Desc(w, h, 3, 0, 0)) is None
    # a stream whose header claims the refused shape: the reference returns NULL (seqoia.h:679-688), so must every decoder here
    small = o.encode(np.zeros(2 * 2 * 3, dtype=np.uint8), 2, 2, 3, 0)
    hdr = bytearray(small)
    hdr[4:8] = w.to_bytes(4, "big")
    hdr[8:12] = h.to_bytes(4, "big")
    assert o.decode(bytes(hdr), 3)[0] is None
    assert sq.decode(bytes(hdr), 3)[0] is None
    buf = torch.from_numpy(np.frombuffer(bytes(hdr), dtype=np.uint8).copy()).cuda().reshape(1, -1)
    dec = sq.DecodeList([(buf.data_ptr(), len(hdr), 0, 0, 3)])
    dec.launch()
    dres = dec.results()
    assert dres["status"][0] != 0 and dres["len"][0] == 0
    dec.close()
