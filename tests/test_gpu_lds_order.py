"""The QOI-compat kernels in BOTH forms the library carries, against the compiled reference.

The default QOI-compat encoder probes (seqoia.h:563-572) and the decoder's index update (seqoia.h:785-787) rest on the order in which
gfx950's LDS applies one instruction's lanes -- a property the ISA manual does not state.  The library asks the device when the first plan
is created (seqoia_amd/csrc/sqoa_selftest.hip) and launches the order-free forms of those kernels (`sqoa::fb::encode_kernel`,
`decode_qoi_kernel<.., TABW = false>`) where the device does not pass; `SQOA_HIP_LDS_ORDER=0` forces them.  These tests
  * run the probe through the C-ABI and check what the library decided, and
  * run the QOI-compat parity set -- encoder KATs, synthetic digests, batches against the oracle, random low-entropy pixels, runs of every
    length, INDEX hits everywhere, strips with the colour index handed on, small images, all 1 024 QOI-compat headline digests; decoder
    random bodies, long random streams, groups of few INDEX ops -- on the order-free forms, so that they cannot rot unnoticed: every one of
    them fails if the fallback's output differs from the reference's.
(The same tests run on the default forms in test_gpu_encode.py / test_gpu_decode.py.)"""
import ctypes as C

import pytest

import test_gpu_decode as D
import test_gpu_encode as E
from test_gpu_encode import orc, sq  # noqa: F401 -- fixtures

pytestmark = pytest.mark.gpu


def _params(fn, name):
    """The argument tuples of the `name` parametrisation of a test in the other modules (so that this file runs what they run)."""
    for m in fn.pytestmark:
        if m.name == "parametrize" and m.args[0] == name:
            return list(m.args[1])
    raise KeyError(name)


def test_the_device_passes_the_probe_and_the_library_says_which_form_it_launches(sq, monkeypatch):
    lib = sq.load()
    out = (C.c_uint * 3)()
    assert lib.sqoa_hip_lds_order_probe(0, out) == 0
    assert out[2] > 10_000_000, "the probe must have checked something"
    assert (out[0], out[1]) == (0, 0), f"gfx950's LDS applied the lanes of one instruction out of ascending order: {out[0]} atomic returns, {out[1]} plain writes"
    monkeypatch.delenv("SQOA_HIP_LDS_ORDER", raising=False)
    assert lib.sqoa_hip_lds_order(0) == 1     # asked, passed: the fast forms
    monkeypatch.setenv("SQOA_HIP_LDS_ORDER", "0")
    assert lib.sqoa_hip_lds_order(0) == 0     # forced: the order-free forms
    monkeypatch.setenv("SQOA_HIP_LDS_ORDER", "1")
    assert lib.sqoa_hip_lds_order(0) == 1
    assert lib.sqoa_hip_lds_order(99) == -1


@pytest.fixture
def order_free(sq, monkeypatch):
    """Plans created inside the test launch the order-free QOI-compat kernels."""
    monkeypatch.setenv("SQOA_HIP_LDS_ORDER", "0")
    yield
    monkeypatch.delenv("SQOA_HIP_LDS_ORDER")
    assert sq.load().sqoa_hip_lds_order(0) == 1   # (and the library is back on the fast forms for whoever comes next)


def _forced(sq):
    import torch

    px = torch.zeros((1, 8, 8, 4), dtype=torch.uint8, device="cuda")
    p = sq.EncodePlan(px, qoi_compat=1)   # (a plan creation reads the knob)
    p.close()
    assert sq.load().sqoa_hip_lds_order(0) == 0


def test_order_free_encoder_kats_and_digests(sq, orc, order_free, kat_encode, synth_digests):
    _forced(sq)
    E.test_legacy_encode_kats(sq, kat_encode, 1)
    E.test_synth_digests(sq, orc, synth_digests, 1)


@pytest.mark.parametrize("cls,w,h,ch,n", _params(E.test_batch_encode_matches_oracle, "cls,w,h,ch,n"))
def test_order_free_batch_encode_matches_oracle(sq, orc, order_free, cls, w, h, ch, n):
    E.test_batch_encode_matches_oracle(sq, orc, 1, cls, w, h, ch, n)


def test_order_free_encoder_on_random_and_crafted_pixels(sq, orc, order_free, monkeypatch):
    E.test_batch_encode_random_lowentropy(sq, orc, 1)
    E.test_long_runs_across_subtiles(sq, orc, 1)
    E.test_unaligned_source_and_slot_addresses(sq, orc, 1)
    E.test_many_small_images_get_one_wave_each(sq, orc, 1)
    for ch in (3, 4):
        E.test_l8_hot_loop_runs_of_every_length(sq, orc, ch, 1)
        E.test_qoi_l8_hot_loop_index_hits_everywhere(sq, orc, ch)
        E.test_large_single_image(sq, orc, 1, ch)
    for knob in ("1", ""):
        E.test_qoi_compat_images_are_cut_into_strips_with_the_colour_index_handed_on(sq, orc, monkeypatch, knob)
    _forced(sq)


def test_order_free_headline_batch_all_1024_qoi_digests(sq, orc, order_free, synth_digests):
    """BASELINE configs 2 + 3 in QOI-compat framing at full size on the order-free kernels: every stream's length and FNV-1a-64 against the
    compiled reference's (tests/golden/headline_digests.json), and the decode of all of them back to the pixels."""
    _forced(sq)
    D.test_full_headline_batch_roundtrip(sq, orc, synth_digests)


@pytest.mark.parametrize("multipass", ["0", "1"], ids=["chained", "threepass"])
def test_order_free_qoi_decoder_on_random_and_crafted_streams(sq, orc, order_free, monkeypatch, multipass):
    monkeypatch.setenv("SQOA_HIP_MULTIPASS", multipass)
    _forced(sq)
    D.test_batch_decode_random_bodies_match_oracle(sq, orc, multipass)
    D.test_qoi_long_random_streams_match_oracle(sq, orc, multipass)
    D.test_qoi_groups_of_few_index_ops_match_oracle(sq, orc, multipass)
    D.test_batch_decode_mutants_match_oracle(sq, orc, multipass)
    D.test_a_relaunched_plan_finds_its_result_rows_reset(sq, orc, multipass)


def test_order_free_qoi_roundtrips(sq, orc, order_free):
    for cls, w, h, ch, n in _params(D.test_batch_roundtrip_and_oracle, "cls,w,h,ch,n"):
        D.test_batch_roundtrip_and_oracle(sq, orc, 1, cls, w, h, ch, n)
    D.test_many_short_streams_take_the_small_workgroup_shapes(sq, orc)
