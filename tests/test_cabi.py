"""The C-ABI boundary without a GPU: the library loads, exports every symbol the headers declare,
keeps the reference's struct layout, and refuses bad arguments exactly where the reference does
before touching the device (seqoia.h:465-480, :662-668)."""
import ctypes as C
import os
import re
import subprocess

import numpy as np

import seqoia_amd
from seqoia_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    names = set()
    for hdr in ("seqoia.h", "sqoa_hip.h"):
        text = open(os.path.join(ROOT, "include", hdr)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        names |= set(re.findall(r"\b(sqoa_[a-z0-9_]+)\s*\(", text))
    return names


def test_library_exports_every_declared_symbol():
    lib = seqoia_amd.load()
    names = declared_functions()
    assert {"sqoa_encode", "sqoa_decode", "sqoa_read", "sqoa_write", "sqoa_hip_plan_create", "sqoa_hip_plan_launch"} <= names
    for n in names:
        assert hasattr(lib, n), n
    assert names == set(_lib.SYMBOLS), names ^ set(_lib.SYMBOLS)


def test_struct_layouts_match_the_reference_abi():
    # sqoa_desc: {unsigned, unsigned, uchar, uchar, uchar}, sizeof 12 (seqoia.h:318-324)
    assert C.sizeof(_lib.Desc) == 12
    assert [(_lib.Desc.width.offset), (_lib.Desc.height.offset), _lib.Desc.channels.offset, _lib.Desc.colorspace.offset,
            _lib.Desc.qoi_compat.offset] == [0, 4, 8, 9, 10]
    assert C.sizeof(_lib.Job) == 40 and C.sizeof(_lib.Result) == 20


def test_header_compiles_as_c_and_cpp_and_links(tmp_path):
    """A reference-style caller: #define SQOA_IMPLEMENTATION + #include "seqoia.h" (sqoabench.c:31-32)."""
    src = tmp_path / "caller.c"
    src.write_text(
        '#define SQOA_IMPLEMENTATION\n#include "seqoia.h"\n#include <stdlib.h>\n'
        "int main(void){ int n=0; unsigned char px[4]={1,2,3,4};\n"
        " void *p = sqoa_encode(px, &(sqoa_desc){.width=0,.height=1,.channels=4,.colorspace=SQOA_SRGB,.qoi_compat=0}, &n);\n"
        " sqoa_desc d; void *q = sqoa_decode(px, 4, &d, 4);\n"
        " return (p==NULL && q==NULL && sizeof(sqoa_desc)==12) ? 0 : 1; }\n"
    )
    libdir = os.path.dirname(_lib.LIB_PATH)
    exe = tmp_path / "caller"
    subprocess.run(["gcc", "-std=gnu99", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe), "-L", libdir,
                    "-lsqoa_mi355x", f"-Wl,-rpath,{libdir}"], check=True)
    assert subprocess.run([str(exe)]).returncode == 0
    cpp = tmp_path / "caller.cpp"
    cpp.write_text('#include "seqoia.h"\n#include "sqoa_hip.h"\nint main(){ sqoa_desc d{}; return sqoa_hip_encode_bound(&d) == 0 ? 0 : 1; }\n')
    subprocess.run(["g++", "-I", os.path.join(ROOT, "include"), str(cpp), "-o", str(exe), "-L", libdir, "-lsqoa_mi355x",
                    f"-Wl,-rpath,{libdir}"], check=True)
    assert subprocess.run([str(exe)]).returncode == 0


def test_argument_checks_need_no_device():
    px = np.zeros(64, np.uint8)
    D = seqoia_amd.Desc
    for d in (D(0, 4, 4, 0, 0), D(4, 0, 4, 0, 0), D(4, 4, 0, 0, 0), D(4, 4, 7, 0, 0), D(4, 4, 4, 2, 0), D(20000, 20000, 4, 0, 0),
              D(4, 4, 1, 0, 1), D(4, 4, 2, 0, 1)):
        assert seqoia_amd.encode(px, d) is None
    assert seqoia_amd.decode(b"\0" * 21, 4)[0] is None          # shorter than header + padding
    assert seqoia_amd.decode(b"Sqoa" + b"\0" * 30, 5)[0] is None  # channels > 4
    assert seqoia_amd.decode(b"Sqob" + b"\0" * 30, 4)[0] is None  # bad magic: rejected on the host


def test_lds_order_entry_points_say_no_device_without_one_and_both_kernel_forms_are_in_the_library():
    """include/sqoa_hip.h: sqoa_hip_lds_order / sqoa_hip_lds_order_probe report "no such device" where there is none (no compute, no fallback), and
    the ONE library holds both forms of the QOI-compat encoder -- the order-free one as sqoa::fb::encode_kernel."""
    import ctypes as C
    import torch

    lib = seqoia_amd._lib.load()
    if not torch.cuda.is_available():
        out = (C.c_uint * 3)()
        assert lib.sqoa_hip_lds_order(0) == -1 and lib.sqoa_hip_lds_order_probe(0, out) == -1
    nm = subprocess.run(["nm", "-C", seqoia_amd.LIB_PATH], capture_output=True, text=True, check=True).stdout
    for sym in ("sqoa::encode_kernel<8, true>", "sqoa::fb::encode_kernel<8, true>", "sqoa::fb::encode_kernel<4, true>", "sqoa::fb::encode_kernel<1, true>",
                "sqoa::lds_order_probe_kernel", "sqoa_launch_encode_tab_fb", "sqoa_launch_encode_small_fb"):
        assert sym in nm, sym
    assert "sqoa::fb::encode_kernel<8, false>" not in nm   # (the second build holds the QOI-compat kernels only)


def test_encode_bound_matches_reference_formula():
    # w*h*(ch+1)+14+8 (seqoia.h:487-489) + 1 start byte (seqoia.h:513), rounded up to 256
    for w, h, c, ch in ((1, 1, 4, 4), (1024, 1024, 4, 4), (64, 64, 3, 3), (10, 10, 5, 3), (10, 10, 6, 4), (7, 3, 1, 1), (7, 3, 2, 2)):
        raw = w * h * (ch + 1) + 14 + 8 + 1
        assert seqoia_amd.encode_bound(seqoia_amd.Desc(w, h, c, 0, 0)) == (raw + 255) // 256 * 256
    assert seqoia_amd.encode_bound(seqoia_amd.Desc(0, 1, 4, 0, 0)) == 0


def _route(sizes, cus=256, mode=0):
    import ctypes as C
    lib = seqoia_amd._lib.load()
    n = len(sizes)
    arr = (C.c_uint * max(n, 1))(*sizes)
    order = (C.c_int * max(n, 1))()
    k = C.c_int(-1)
    assert lib.sqoa_hip_decode_route(arr, n, cus, mode, order, C.byref(k)) == 0
    return list(order[:n]), k.value


def test_decode_route_is_a_permutation_and_picks_the_streams_that_would_outlast_the_batch():
    """The host-side choice between one workgroup per stream and the three chip-wide passes."""
    mb = 1 << 20
    # the headline batch: 1024 equal streams stay one workgroup each
    order, k = _route([2600000] * 1024)
    assert k == 0 and sorted(order) == list(range(1024))
    # a lone stream, and a handful: all multi-pass, largest first
    assert _route([2600000]) == ([0], 1)
    order, k = _route([mb, 3 * mb, 2 * mb, 5 * mb])
    assert k == 4 and order == [3, 1, 2, 0]
    # tiny streams never take the passes (the launches cost more than the decode)
    assert _route([4000] * 8)[1] == 0
    # one screenshot among thousands of small images (BASELINE configs[4] shape): only the large ones
    rng = np.random.default_rng(5)
    sizes = [int(x) for x in rng.integers(20_000, 400_000, 4000)] + [20 * mb, 19 * mb, 9 * mb]
    order, k = _route(sizes)
    assert sorted(order) == list(range(len(sizes)))
    assert k == 3 and order[:3] == [4000, 4001, 4002]
    # forced modes
    assert _route(sizes, mode=-1)[1] == 0 and _route(sizes, mode=1)[1] == len(sizes)
    # the one-workgroup-per-stream part: largest first in runs of `cus`, every second run reversed
    order, k = _route(list(range(100000, 100000 + 40 * 1000, 1000)), cus=8, mode=-1)
    s = [100000 + 1000 * i for i in order]
    assert s[:8] == sorted(s[:8], reverse=True) and s[8:16] == sorted(s[8:16]) and s[16:24] == sorted(s[16:24], reverse=True)
    assert max(s[8:16]) <= min(s[:8]) and max(s[16:24]) <= min(s[8:16])
    # empty plan and bad arguments
    assert _route([]) == ([], 0)
    import ctypes as C
    assert seqoia_amd._lib.load().sqoa_hip_decode_route(None, 1, 256, 0, None, None) != 0


def test_encode_route_cuts_the_images_that_would_outlast_the_batch():
    """The host-side choice between one workgroup per image and strips over the whole chip (sqoa_strips.hip)."""
    import ctypes as C
    lib = seqoia_amd._lib.load()

    def route(px, cus=256, mode=0):
        n = len(px)
        arr = (C.c_ulonglong * max(n, 1))(*px)
        order, strips, k = (C.c_int * max(n, 1))(), (C.c_int * max(n, 1))(), C.c_int(-1)
        assert lib.sqoa_hip_encode_route(arr, n, cus, mode, order, strips, C.byref(k)) == 0
        return list(order[:n]), list(strips[:n]), k.value

    mp = 1 << 20
    order, strips, k = route([mp] * 1024)                      # the headline batch: one workgroup per image
    assert k == 0 and sorted(order) == list(range(1024)) and not any(strips)
    order, strips, k = route([mp])                             # a lone image: a strip per compute unit, 256 of 4096 pixels
    assert (order, k) == ([0], 1) and strips == [256]
    order, strips, k = route([mp] * 16)                        # enough images to fill the chip: strips of 16 384 pixels
    assert k == 16 and strips == [64] * 16
    order, strips, k = route([3840 * 2160])                    # a lone screenshot: about 2048 strips / 4 rounds... at least a few hundred
    assert k == 1 and 400 <= strips[0] <= 2048
    order, strips, k = route([64 * 64] * 100)                  # icons: too small to cut
    assert k == 0
    sizes = [300 * 300] * 5000 + [3840 * 2160, 2560 * 1440]    # screenshots among thousands of small images
    order, strips, k = route(sizes)
    assert k == 2 and order[:2] == [5000, 5001] and strips[0] > strips[1] >= 2 and sorted(order) == list(range(len(sizes)))
    assert route(sizes, mode=-1)[2] == 0 and route(sizes, mode=1)[2] == len(sizes)
    assert lib.sqoa_hip_encode_route(None, 1, 256, 0, None, None, None) != 0


def test_no_kernel_restores_exec_from_exec(tmp_path):
    """The hand-written blocks that narrow EXEC (`v_cmpx`) put it back from a copy taken by `exec_saved()` (sqoa_common.h).  Taken through the
    compiler's builtin the copy can come out as the EXEC register itself -- `s_mov_b64 exec, exec`: the narrowed mask stays and a wave whose
    mask is empty never ends (a hang on the GPU box in round 5, found in the disassembly).  Every gfx950 code object of the built library is
    disassembled here and searched for that instruction."""
    import glob
    import shutil

    objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
    if not os.path.exists(objdump):
        import pytest

        pytest.skip("no llvm-objdump in this image")
    lib = shutil.copy(seqoia_amd._lib.LIB_PATH, tmp_path / "lib.so")
    subprocess.run([objdump, "--offloading", str(lib)], check=True, capture_output=True, cwd=tmp_path)
    objs = glob.glob(str(tmp_path / "lib.so.*gfx950"))
    assert len(objs) >= 5, objs
    kernels = 0
    for o in objs:
        text = subprocess.run([objdump, "-d", o], check=True, capture_output=True, text=True).stdout
        kernels += len(re.findall(r"^[0-9a-f]+ <_Z\w*kernel\w*>:", text, flags=re.M))
        bad = [l for l in text.splitlines() if re.search(r"s_mov_b64\s+exec,\s*exec\b", l)]
        assert not bad, (o, bad[:3])
    assert kernels >= 10, kernels


def test_decoders_that_keep_the_parse_entries_in_registers_do_not_spill(tmp_path):
    """The balanced SQOA decoder and the eight-wave QOI-compat decoder keep the parse's transition-table entries in VGPRs for the starts pass
    instead of looking them up in LDS a second time (round 6) -- on the understanding that nothing else wants those
    registers there.  The code objects' notes say whether that still holds with this compiler: no VGPR spill in the four-channel balanced
    kernel and in the eight-wave QOI-compat instances, eight waves a SIMD (at most 64 VGPRs), and a workgroup's LDS within a quarter of a CU's."""
    import glob
    import shutil

    objdump, readelf = "/opt/rocm/lib/llvm/bin/llvm-objdump", "/opt/rocm/lib/llvm/bin/llvm-readelf"
    if not (os.path.exists(objdump) and os.path.exists(readelf)):
        import pytest

        pytest.skip("no llvm-objdump / llvm-readelf in this image")
    lib = shutil.copy(seqoia_amd._lib.LIB_PATH, tmp_path / "lib.so")
    subprocess.run([objdump, "--offloading", str(lib)], check=True, capture_output=True, cwd=tmp_path)
    found = {}
    for o in glob.glob(str(tmp_path / "lib.so.*gfx950")):
        notes = subprocess.run([readelf, "--notes", o], check=True, capture_output=True, text=True).stdout
        for blk in notes.split("- .agpr_count:")[1:]:
            name = re.search(r"\.name:\s+(\S+)", blk)
            if not name:
                continue
            get = lambda key: int(re.search(r"\." + key + r":\s+(\d+)", blk).group(1))
            found[name.group(1)] = (get("vgpr_count"), get("vgpr_spill_count"), get("group_segment_fixed_size"))
    bal = [v for k, v in found.items() if "decode_bal_kernelILi8ELi16ELb0EEE" in k]
    qoi8 = [v for k, v in found.items() if re.search(r"sqoa16?\d*decode_qoi_kernelILi8E", k) or "decode_qoi_kernelILi8E" in k]
    assert len(bal) == 1 and len(qoi8) >= 2, sorted(found)
    for vgpr, spill, lds in bal + qoi8:
        assert spill == 0 and vgpr <= 64 and lds <= 160 * 1024 // 4, (vgpr, spill, lds)
