"""The drop-in boundary proven with real callers.

* CPU, build container only: the reference's own fuzz harness /root/reference/sqoafuzz.c is
  compiled UNCHANGED against include/seqoia.h and linked with libsqoa_mi355x.so.  `#include
  "seqoia.h"` searches the including file's own directory first, so the caller is reached through
  a symbolic link in a directory that does not hold the reference header (nothing is copied); nm
  must show the four entry points undefined (bound to the library), none defined in the caller.
* GPU: tests/c/fuzz_records.c -- the same call shape, sqoafuzz.c:20-31 -- decodes every golden
  crafted stream through the C-ABI from plain C and must reproduce the reference's pixels.
"""
import os
import struct
import subprocess

import pytest

from oracle import pyoracle as po
from seqoia_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
INC = os.path.join(ROOT, "include")
LIBDIR = os.path.dirname(_lib.LIB_PATH)
REF_FUZZ = "/root/reference/sqoafuzz.c"
HARNESS = os.path.join(ROOT, "tests", "c", "fuzz_records.c")
LINK = ["-L", LIBDIR, "-lsqoa_mi355x", f"-Wl,-rpath,{LIBDIR}"]


def nm_symbols(path):
    out = subprocess.run(["nm", path], check=True, capture_output=True, text=True).stdout
    return {(parts[-2], parts[-1]) for parts in (line.split() for line in out.splitlines()) if len(parts) >= 2}


def write_records(path, kats):
    with open(path, "wb") as f:
        for c in kats:
            rec = struct.pack("=i", c["channels"]) + bytes.fromhex(c["stream"])
            f.write(struct.pack("<I", len(rec)) + rec)


@pytest.mark.skipif(not os.path.exists(REF_FUZZ), reason="/root/reference is only present in the build container")
def test_reference_fuzz_harness_binds_to_the_library(tmp_path, kat_decode):
    link = tmp_path / "sqoafuzz.c"
    os.symlink(REF_FUZZ, link)  # a link, not a copy: the directory must not hold the reference's seqoia.h
    obj = tmp_path / "sqoafuzz.o"
    subprocess.run(["gcc", "-std=gnu99", "-O1", "-Werror=implicit-function-declaration", "-I", INC, "-c", str(link), "-o", str(obj)],
                   check=True)
    syms = nm_symbols(str(obj))
    assert ("U", "sqoa_decode") in syms and ("U", "free") in syms
    assert not [s for s in syms if s[0] in "Tt" and s[1].startswith("sqoa_")], "the caller compiled a CPU codec in"
    # the control: next to the reference's own header the very same command line defines the codec in the caller
    ctl = tmp_path / "control.o"
    subprocess.run(["gcc", "-std=gnu99", "-O1", "-w", "-I", INC, "-c", REF_FUZZ, "-o", str(ctl)], check=True)
    assert ("T", "sqoa_decode") in nm_symbols(str(ctl))
    exe = tmp_path / "fuzz_records"
    subprocess.run(["gcc", "-std=gnu99", "-O1", "-Wall", "-Werror", "-DWITH_FUZZ_ENTRY", "-I", INC, HARNESS, str(obj), "-o", str(exe)]
                   + LINK, check=True)
    syms = nm_symbols(str(exe))
    assert ("U", "sqoa_decode") in syms and not [s for s in syms if s[0] in "Tt" and s[1].startswith("sqoa_")]
    # without a GPU every call must come back (NULL), not crash; with one, test_fuzz_shaped_c_caller checks the pixels
    recs = tmp_path / "recs.bin"
    write_records(recs, kat_decode[:4])
    env = dict(os.environ, HIP_VISIBLE_DEVICES="-1", ROCR_VISIBLE_DEVICES="-1")
    r = subprocess.run([str(exe), str(recs)], capture_output=True, text=True, env=env, timeout=120)
    assert r.returncode == 0, r.stderr
    assert len(r.stdout.splitlines()) == 4


@pytest.mark.gpu
def test_fuzz_shaped_c_caller_decodes_every_golden_stream(tmp_path, kat_decode):
    import torch

    assert torch.cuda.is_available(), "needs a HIP device"
    exe = tmp_path / "fuzz_records"
    subprocess.run(["gcc", "-std=gnu99", "-O1", "-Wall", "-Werror", "-I", INC, HARNESS, "-o", str(exe)] + LINK, check=True)
    recs = tmp_path / "recs.bin"
    write_records(recs, kat_decode)
    r = subprocess.run([str(exe), str(recs)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    lines = r.stdout.splitlines()
    assert len(lines) == len(kat_decode)
    for i, (c, line) in enumerate(zip(kat_decode, lines)):
        if c["pixels"] is None:
            assert line == f"{i} NULL", (c["name"], line)
            continue
        px = bytes.fromhex(c["pixels"])
        w, h, fch, cs, qoi = c["desc"]
        assert line == f"{i} {w} {h} {fch} {cs} {qoi} {len(px)} {po.fnv1a64(px):016x}", (c["name"], line)
