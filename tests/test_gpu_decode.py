"""GPU parity: the HIP decoder behind the C-ABI vs golden vectors and the oracle."""
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

    return po


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


def test_batch_decode_mutants_match_oracle(sq, orc):
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
