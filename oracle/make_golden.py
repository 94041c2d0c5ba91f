#!/usr/bin/env python3
"""Capture golden vectors from the COMPILED REFERENCE -- TEST INFRASTRUCTURE ONLY.

Runs oracle/_ref/libsqoa_ref.so (built by oracle/Makefile from
/root/reference/seqoia.h; needs the reference tree, so only works in the build
container) and writes inputs + the reference's outputs to tests/golden/*.json.
The fixtures are data only; no reference source text is stored.

    python oracle/make_golden.py
"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import pyoracle as po  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def hx(b):
    return None if b is None else bytes(b).hex()


def rgba(*px):
    return np.array(px, dtype=np.uint8).reshape(-1)


def encode_cases():
    """SURVEY.md Appendix B inputs plus boundary cases of the run rules (seqoia.h:544-561, 640-642)."""
    c = []

    def add(name, pixels, w, h, ch, colorspace=0):
        c.append(dict(name=name, w=w, h=h, channels=ch, colorspace=colorspace, pixels=np.asarray(pixels, np.uint8)))

    add("three_default", rgba(*[(0, 0, 0, 255)] * 3), 3, 1, 4)
    add("luma_run_luma", rgba((1, 1, 1, 255), (1, 1, 1, 255), (2, 2, 2, 255)), 3, 1, 4)
    add("alpha_steps", rgba((10, 20, 30, 255), (10, 20, 30, 250), (10, 20, 30, 100)), 3, 1, 4)
    add("index_pingpong_rgba", rgba((200, 100, 50, 255), (0, 0, 0, 0), (200, 100, 50, 255), (0, 0, 0, 0)), 4, 1, 4)
    add("index_pingpong_rgb", rgba((200, 100, 50), (7, 9, 11), (200, 100, 50), (7, 9, 11)), 4, 1, 3)
    add("mono", rgba(5, 5, 6, 200, 200), 5, 1, 1)
    add("mono_alpha", rgba((5, 255), (5, 255), (6, 250), (200, 0)), 4, 1, 2)
    add("bgr_as_rgb", rgba((1, 2, 3), (4, 5, 6)), 2, 1, 5)
    add("bgra_as_rgba", rgba((1, 2, 3, 4), (5, 6, 7, 8)), 2, 1, 6)
    add("run700", rgba(*[(9, 9, 9, 255)] * 700), 700, 1, 4)
    add("run699_then_diff", rgba(*([(9, 9, 9, 255)] * 699 + [(10, 9, 9, 255)])), 700, 1, 4)
    add("run198_mid", rgba(*([(8, 9, 9, 255)] + [(9, 9, 9, 255)] * 198 + [(1, 9, 9, 255)])), 200, 1, 4)
    add("two_zero", rgba(*[(0, 0, 0, 0)] * 2), 2, 1, 4)
    add("worst_1x1", rgba((100, 5, 200, 7)), 1, 1, 4)
    add("linear_colorspace", rgba((1, 2, 3, 4), (1, 2, 3, 4)), 2, 1, 4, colorspace=1)
    # every run remainder around the 61/62/512 boundaries, ended by a different pixel
    for r in (1, 60, 61, 62, 63, 122, 123, 488, 489, 510, 511, 512, 513, 573, 574, 1023, 1024, 1025, 1536):
        add(f"run{r}_then_px", rgba(*([(7, 7, 7, 255)] * (r + 1) + [(50, 60, 70, 255)])), r + 2, 1, 4)
        add(f"run{r}_tail_rgb", rgba(*([(7, 7, 7)] * (r + 1))), r + 1, 1, 3)
    # leading run of the implicit start pixel, then index-relevant colours
    add("lead_run_then_zero", rgba(*([(0, 0, 0, 255)] * 5 + [(0, 0, 0, 0)] * 3 + [(0, 0, 0, 255)])), 9, 1, 4)
    # wrap-around deltas (signed-char truncation, seqoia.h:585-590)
    add("wrap_deltas", rgba((250, 3, 128, 255), (2, 250, 127, 255), (130, 120, 1, 240), (129, 119, 0, 255),
                            (0, 0, 0, 15), (255, 255, 255, 255), (247, 255, 7, 255), (0, 32, 0, 255),
                            (0, 0, 0, 255), (224, 224, 224, 239), (231, 192, 217, 223)), 11, 1, 4)
    rng = np.random.default_rng(20250228)
    for k, (w, h, ch) in enumerate([(17, 5, 4), (64, 3, 3), (65, 2, 4), (33, 33, 3), (31, 7, 1), (19, 11, 2),
                                    (128, 2, 4), (5, 1, 6), (9, 9, 5)]):
        # low-entropy random walk so LUMA/DIFF/RUN/INDEX all occur
        steps = rng.integers(-3, 4, size=(w * h, ch)).astype(np.int64)
        steps[rng.random(w * h) < 0.35] = 0
        jump = rng.random(w * h) < 0.08
        steps[jump] = rng.integers(-128, 128, size=(int(jump.sum()), ch))
        px = (np.cumsum(steps, axis=0) + 100) % 256
        pal = rng.random(w * h) < 0.15
        px[pal] = np.array([[10, 200, 30, 255, 0, 0][:ch]])
        add(f"walk{k}_{w}x{h}x{ch}", px.astype(np.uint8).reshape(-1), w, h, ch)
    return c


def hdr(magic, w, h, ch, cs=0):
    return magic + w.to_bytes(4, "big") + h.to_bytes(4, "big") + bytes([ch, cs])


END = bytes(7) + b"\x01"


def decode_cases(ref):
    """Crafted streams (seqoia.h:722-806) incl. REF / leading ALPHA / truncation, and header rejects."""
    S, Q = b"Sqoa", b"qoif"
    c = []

    def add(name, stream, channels=0):
        c.append(dict(name=name, stream=bytes(stream), channels=channels))

    add("ref_op", hdr(S, 6, 1, 4) + b"\x31" + bytes.fromhex("fe010203fe09090945a188a288") + END, 4)
    add("alpha_then_run", hdr(S, 6, 1, 4) + b"\x31" + bytes.fromhex("fe0102036570") + END, 4)
    add("alpha_suffixes", hdr(S, 4, 1, 4) + b"\x31" + bytes.fromhex("fe0102036fa18871") + END, 4)
    add("leading_alpha_is_run", hdr(S, 40, 1, 4) + b"\x31" + bytes.fromhex("6afe010203") + END, 4)
    add("ref_before_start", hdr(S, 4, 1, 4) + b"\x31" + bytes.fromhex("5f") + END, 4)
    add("ref_len4_off31", hdr(S, 40, 1, 4) + b"\x31" + bytes(range(0x80, 0xa0)) + bytes.fromhex("fe0a0b0c" "5f" "a188c3") + END, 4)
    add("ref_to_low_byte", hdr(S, 50, 1, 4) + b"\x31" + bytes.fromhex("fe10203005a1884203a090") + END, 4)
    add("exhausted_stream", hdr(S, 9, 1, 4) + b"\x31" + bytes.fromhex("fe0a0b0c") + END, 4)
    add("bigrun_overshoot", hdr(S, 3, 2, 3) + b"\x31" + bytes.fromhex("fe010203fd") + END, 0)
    add("qoi_fd_is_run62", hdr(Q, 70, 1, 4) + bytes.fromhex("fe010203fdc5") + END, 4)
    add("qoi_index_zero_init", hdr(Q, 4, 1, 4) + bytes.fromhex("05fe0a0b0c2405") + END, 4)
    add("qoi_diff_luma", hdr(Q, 5, 1, 3) + bytes.fromhex("fe6464647f40a088c1") + END, 3)
    add("qoi_mono_index128", hdr(Q, 6, 1, 1) + bytes.fromhex("fe09" "7f" "a5" "13" "ff0a0b" "c0") + END, 0)
    add("qoi_mono_alpha_to_rgba", hdr(Q, 4, 1, 2) + bytes.fromhex("ff0a0b" "a5" "c1") + END, 4)
    add("sqoa_mono", hdr(S, 6, 1, 1) + b"\x31" + bytes.fromhex("a5c0a1fec8fd") + END, 0)
    add("sqoa_mono_to_rgb", hdr(S, 6, 1, 1) + b"\x31" + bytes.fromhex("a5c0a1fec8fd") + END, 3)
    add("sqoa_monoa", hdr(S, 4, 1, 2) + b"\x31" + bytes.fromhex("a5c0ff06faffc800") + END, 0)
    add("rgb_file_to_1ch", hdr(S, 3, 1, 3) + b"\x31" + bytes.fromhex("fe0a1420a188a288") + END, 1)
    add("rgba_file_to_2ch", hdr(S, 3, 1, 4) + b"\x31" + bytes.fromhex("ff0a142080a18871") + END, 2)
    add("rgba_file_to_3ch", hdr(S, 3, 1, 4) + b"\x31" + bytes.fromhex("ff0a142080a18871") + END, 3)
    add("hdr_channels5", hdr(S, 2, 1, 5) + b"\x31" + bytes.fromhex("a279a388") + END, 0)
    add("hdr_channels6", hdr(S, 2, 1, 6) + b"\x31" + bytes.fromhex("a27975a48874") + END, 0)
    # rejects (seqoia.h:662-668, 679-688, 705-707)
    add("bad_magic", hdr(b"Sqob", 2, 1, 4) + b"\x31" + b"\xfd" + END, 4)
    add("qoif_with_start_byte", hdr(Q, 2, 1, 4) + b"\x31" + b"\xfd" + END, 4)
    add("sqoa_without_start_byte_is_qoi", hdr(S, 2, 1, 4) + b"\x30\xfd" + END, 4)
    add("zero_width", hdr(S, 0, 1, 4) + b"\x31" + b"\xfd" + END, 4)
    add("zero_height", hdr(S, 1, 0, 4) + b"\x31" + b"\xfd" + END, 4)
    add("channels0_hdr", hdr(S, 1, 1, 0) + b"\x31" + b"\xfd" + END, 4)
    add("channels7_hdr", hdr(S, 1, 1, 7) + b"\x31" + b"\xfd" + END, 4)
    add("colorspace2", hdr(S, 1, 1, 4, 2) + b"\x31" + b"\xfd" + END, 4)
    add("too_many_pixels", hdr(S, 20000, 20000, 4) + b"\x31" + b"\xfd" + END, 4)
    add("too_short", hdr(S, 1, 1, 4)[:13] + END, 4)
    add("min_size_22", hdr(S, 1, 1, 4) + END, 4)
    add("channels_arg_5", hdr(S, 1, 1, 4) + b"\x31" + b"\xfd" + END, 5)
    # mutated valid streams: flip bytes of real encoder output
    rng = np.random.default_rng(7)
    for k in range(24):
        w, h, ch = [(23, 9, 4), (40, 5, 3), (16, 16, 4), (50, 3, 3)][k % 4]
        px = po.synth(w, h, ch, ["mixed", "icon", "photo"][k % 3], seed=3, img=k)
        enc = bytearray(ref.encode(px, w, h, ch, qoi_compat=k & 1))
        for _ in range(1 + k % 5):
            enc[int(rng.integers(15, len(enc) - 8))] = int(rng.integers(0, 256))
        add(f"mutant{k}", enc, [0, 3, 4, 1, 2][k % 5])
    return c


def headline_digests(ref, n=1024, w=1024, h=1024, ch=4, seed=1):
    """Length + FNV-1a-64 of EVERY stream of the headline batch (BASELINE.json configs[1]: class mixed, seed 1,
    img 0..n-1) in both wire formats, from the compiled reference (seqoia.h:456-650).  ctypes calls release the GIL."""
    from concurrent.futures import ThreadPoolExecutor

    def one(img):
        px = po.synth(w, h, ch, "mixed", seed, img)
        s, q = ref.encode(px, w, h, ch, 0), ref.encode(px, w, h, ch, 1)
        return [len(s), f"{po.fnv1a64(s):016x}", len(q), f"{po.fnv1a64(q):016x}"]

    with ThreadPoolExecutor(len(os.sched_getaffinity(0))) as ex:
        rows = list(ex.map(one, range(n)))
    return dict(cls="mixed", w=w, h=h, channels=ch, seed=seed, count=n, columns=["sqoa_len", "sqoa_fnv", "qoi_len", "qoi_fnv"], rows=rows)


def _pool_map(fn, items):
    from concurrent.futures import ThreadPoolExecutor

    with ThreadPoolExecutor(len(os.sched_getaffinity(0))) as ex:   # (the ctypes calls release the GIL)
        return list(ex.map(fn, items))


C4_SIZES = (64, 128, 256, 512)


def c4_digests(ref, n=4096, seed=1):
    """BASELINE.json configs[3] in full: class icon, 3 channels, seed 1, img 0..4095 at 64^2, 128^2, 256^2 and 512^2, BOTH wire formats --
    length and FNV-1a-64 of every one of the 4 x 4 096 x 2 streams from the compiled reference (seqoia.h:456-650).
    Arrays [size][format][image]: lens uint32, fnv uint64."""
    lens = np.zeros((len(C4_SIZES), 2, n), np.uint32)
    fnv = np.zeros((len(C4_SIZES), 2, n), np.uint64)
    for si, size in enumerate(C4_SIZES):
        def one(img, size=size):
            px = po.synth(size, size, 3, "icon", seed, img)
            s, q = ref.encode(px, size, size, 3, 0), ref.encode(px, size, size, 3, 1)
            return len(s), po.fnv1a64(s), len(q), po.fnv1a64(q)

        for img, (ls, fs, lq, fq) in enumerate(_pool_map(one, range(n))):
            lens[si, 0, img], fnv[si, 0, img], lens[si, 1, img], fnv[si, 1, img] = ls, fs, lq, fq
        print(f"  C4 {size}x{size}: SQOA {int(lens[si, 0].sum())} B, QOI {int(lens[si, 1].sum())} B", flush=True)
    return dict(sizes=np.array(C4_SIZES, np.uint32), lens=lens, fnv=fnv, seed=np.uint32(seed), channels=np.uint32(3))


def c5_digests(ref, seed=1):
    """BASELINE.json configs[4] in full: all 65 536 suite-shaped images (seqoia_amd/suite.py: directory -> shape and class, generator index = the
    image's global number, seed 1), SQOA-native -- length and FNV-1a-64 of every stream from the compiled reference.  About 30 Gpixel: a few
    core-minutes.  Arrays [image]: lens uint32, fnv uint64."""
    from seqoia_amd import suite

    imgs = suite.images(suite.C5_IMAGES)

    def one(im):
        px = po.synth(im.width, im.height, im.channels, im.cls, seed, im.index)
        s = ref.encode(px, im.width, im.height, im.channels, 0)
        return len(s), po.fnv1a64(s)

    rows = []
    for a in range(0, len(imgs), 4096):
        rows += _pool_map(one, imgs[a : a + 4096])
        print(f"  C5 {len(rows)} / {len(imgs)}", flush=True)
    lens = np.array([r[0] for r in rows], np.uint32)
    fnv = np.array([r[1] for r in rows], np.uint64)
    return dict(lens=lens, fnv=fnv, seed=np.uint32(seed), count=np.uint32(len(imgs)))


def main():
    ref = po.reference()
    if ref is None:
        sys.exit("oracle/_ref/libsqoa_ref.so missing and /root/reference absent: cannot regenerate goldens")
    os.makedirs(OUT, exist_ok=True)
    if "--c4" in sys.argv or "--c5" in sys.argv:  # the (slow) full-set digest tables of configs[3] / configs[4], as compressed numpy archives
        if "--c4" in sys.argv:
            np.savez_compressed(os.path.join(OUT, "c4_digests.npz"), **c4_digests(ref))
            print("wrote c4_digests.npz")
        if "--c5" in sys.argv:
            np.savez_compressed(os.path.join(OUT, "c5_digests.npz"), **c5_digests(ref))
            print("wrote c5_digests.npz")
        return
    if "--headline" in sys.argv:  # only the (slow) whole-batch digest table
        json.dump(headline_digests(ref), open(os.path.join(OUT, "headline_digests.json"), "w"), separators=(",", ":"))
        print("wrote headline_digests.json")
        return

    enc = []
    for c in encode_cases():
        px = c.pop("pixels")
        c["pixels"] = hx(px)
        c["sqoa"] = hx(ref.encode(px, c["w"], c["h"], c["channels"], 0, c["colorspace"]))
        c["qoi"] = hx(ref.encode(px, c["w"], c["h"], c["channels"], 1, c["colorspace"]))
        enc.append(c)
    json.dump(enc, open(os.path.join(OUT, "kat_encode.json"), "w"), indent=0)

    dec = []
    for c in decode_cases(ref):
        out, d = ref.decode(c["stream"], c["channels"])
        c["stream"] = hx(c["stream"])
        c["pixels"] = hx(out)
        if out is not None:
            c["desc"] = [d.width, d.height, d.channels, d.colorspace, d.qoi_compat]
        dec.append(c)
    json.dump(dec, open(os.path.join(OUT, "kat_decode.json"), "w"), indent=0)

    dig = []
    grid = [("mixed", 256, 256, 4, 1, 0), ("mixed", 256, 256, 4, 1, 1), ("icon", 64, 64, 3, 1, 0), ("icon", 64, 64, 3, 1, 1),
            ("mixed", 100, 37, 4, 7, 0), ("photo", 128, 96, 3, 1, 0), ("photo", 77, 31, 4, 2, 5), ("icon", 512, 512, 3, 1, 0),
            ("icon", 128, 128, 4, 1, 3), ("mixed", 1024, 1024, 4, 1, 0), ("mixed", 1024, 1024, 4, 1, 1023),
            ("mixed", 1024, 1024, 3, 1, 2), ("photo", 1024, 1024, 3, 1, 0), ("mixed", 640, 480, 4, 5, 9),
            ("mixed", 1, 1, 4, 1, 0), ("mixed", 63, 1, 4, 1, 0), ("mixed", 64, 1, 3, 1, 0), ("mixed", 513, 1, 4, 1, 0)]
    for cls, w, h, ch, seed, img in grid:
        px = po.synth(w, h, ch, cls, seed, img)
        s, q = ref.encode(px, w, h, ch, 0), ref.encode(px, w, h, ch, 1)
        dig.append(dict(cls=cls, w=w, h=h, channels=ch, seed=seed, img=img, raw_fnv=f"{po.fnv1a64(px):016x}",
                        sqoa_len=len(s), sqoa_fnv=f"{po.fnv1a64(s):016x}", qoi_len=len(q), qoi_fnv=f"{po.fnv1a64(q):016x}"))
    tot = {}
    for key, (cls, w, h, ch, n) in {"mixed_1024_rgba_0_15": ("mixed", 1024, 1024, 4, 16), "icon_512_rgb_0_31": ("icon", 512, 512, 3, 32),
                                    "photo_1024_rgb_0_3": ("photo", 1024, 1024, 3, 4)}.items():
        t = [0, 0]
        for img in range(n):
            px = po.synth(w, h, ch, cls, 1, img)
            t[0] += len(ref.encode(px, w, h, ch, 0))
            t[1] += len(ref.encode(px, w, h, ch, 1))
        tot[key] = dict(cls=cls, w=w, h=h, channels=ch, seed=1, count=n, sqoa_total=t[0], qoi_total=t[1])
    json.dump(dict(images=dig, totals=tot), open(os.path.join(OUT, "synth_digests.json"), "w"), indent=1)
    print(f"wrote {len(enc)} encode KATs, {len(dec)} decode KATs, {len(dig)} digests, {len(tot)} totals -> {OUT}")


if __name__ == "__main__":
    main()
