"""The C-ABI library loads on a CPU-only box and exports what include/*.h declares."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import abracadabra_amd as aa
from abracadabra_amd import dabx
from oracle import binding as ob

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared(header, prefix):
    txt = open(os.path.join(ROOT, "include", header)).read()
    return sorted(set(re.findall(r"\b(" + prefix + r"\w*)\s*\(", txt)) - {"DABX_API", "DABSDR_API"})


def test_library_exports_every_declared_symbol():
    L = aa.load_library()
    names = _declared("dabx.h", "dabx_") + _declared("dabsdr_amd.h", "dabsdr")
    assert len(_declared("dabsdr_amd.h", "dabsdr")) == 24       # the reference's 24 entry points (dabsdr.h:397-429)
    assert sorted(_declared("dabx.h", "dabx_")) == sorted(dabx.DABX_SYMBOLS)
    for n in names:
        assert hasattr(L, n), n


def test_struct_layouts_match_the_reference_abi():
    # sizeof(dabsdrNtfPeriodic_t) is asserted to be the notification length by the host (radiocontrol.cpp:2407)
    assert dabx.SYNC_DTYPE.itemsize == 64 and C.sizeof(dabx.StreamState) == 40 and C.sizeof(dabx.Config) == 24
    L = aa.load_library()
    ver = (C.c_uint8 * 4)()
    L.dabsdrGetVersion(ver)
    assert list(ver)[:3] == [4, 0, 1]


def test_no_gpu_means_loud_failure_not_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(aa.DabxError):
        aa.Context(n_streams=1)
    h = C.c_void_p()
    assert aa.load_library().dabsdrInit(C.byref(h)) != 0 and not h.value


def test_create_rejects_configurations_the_kernels_cannot_index():
    """Argument checks come before any device call, so they run without a GPU: the kernels index one stream's ring with 32 bits
    (dabx.h: (max_frames + 2) frames <= ring_samples <= 2^30), a step holds at most 60 frames, the formats are u8 and s16."""
    L = aa.load_library()
    TF = 196608
    for cfg in (dabx.Config(1, 0, (1 << 30) + 1, 4, 0), dabx.Config(1, 0, 5 * TF, 4, 0), dabx.Config(0, 0, 16 * TF, 4, 0),
                dabx.Config(1, 2, 16 * TF, 4, 0), dabx.Config(1, 0, 80 * TF, 61, 0)):
        h = C.c_void_p()
        assert L.dabx_create(C.byref(cfg), C.byref(h)) == -1 and not h.value         # DABX_E_ARG


def test_fig_database_reads_transmitted_fibs():
    sub = [[0, 0, 3, 64], [48, 1, 4, 32], [100, 2, 17, 0]]
    _, fib, _ = ob.tx_generate(seed=4, eid=0x10AB, n_frames=3, subch=sub, snr_db=100.0)
    L = aa.load_library()
    L.dabsdr_amd_fig_dump.argtypes = [C.c_void_p, C.c_int, C.c_char_p, C.c_int]
    buf = C.create_string_buffer(8192)
    flat = np.ascontiguousarray(fib.reshape(-1, 32))
    n = L.dabsdr_amd_fig_dump(flat.ctypes.data, len(flat), buf, 8192)
    text = buf.value.decode()
    assert n > 0
    assert "eid=10AB ecc=E2 lto=2" in text and "GRAFT ENS" in text and "utc=60587 12:34:56.789" in text
    assert "subch id=0 start=0 size=48 opt=0 level=3 kbps=64" in text
    assert "subch id=1 start=48 size=15 opt=1 level=4 kbps=32" in text
    assert "subch id=2 start=100 size=58 opt=0 level=2 kbps=64" in text          # UEP index 17 via the short form
    assert "service sid=1A01 label='SERVICE 00      ' ncomp=1 [tmid=0 ty=63 subch=0 ps=1]" in text
    assert "service sid=1A03 label='SERVICE 02      ' ncomp=1 [tmid=0 ty=0 subch=2 ps=1]" in text


def test_public_headers_compile_as_plain_c(tmp_path):
    """include/*.h are the boundary a C or cgo/JNI binding would include: they must stand alone under a C compiler,
    and the record layouts the Python bindings mirror must have the documented sizes"""
    import shutil
    import subprocess
    if not shutil.which("gcc"):
        pytest.skip("no gcc")
    inc = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "include")
    src = tmp_path / "h.c"
    src.write_text('#include "dabx.h"\n#include "dabsdr_amd.h"\n'
                   '_Static_assert(sizeof(dabx_superframe_t) == 32, "super frame record");\n'
                   '_Static_assert(sizeof(dabx_sync_rec_t) == 64, "sync record");\n'
                   '_Static_assert(sizeof(dabsdrNtfPeriodic_t) == 32, "periodic notification (radiocontrol.cpp:2407)");\n'
                   'int main(void) { return 0; }\n')
    r = subprocess.run(["gcc", "-std=c11", "-Wall", "-Wextra", "-Werror", "-I", inc, "-c", str(src), "-o", str(tmp_path / "h.o")],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


def test_c_example_builds_against_the_library(tmp_path):
    """examples/decode_rawfiles.c is the batch ABI used from plain C: it must compile and link"""
    import shutil
    import subprocess
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
    if not shutil.which("gcc") or not os.path.exists(os.path.join(root, "abracadabra_amd", "libdabsdr_amd.so")):
        pytest.skip("gcc or the built library missing")
    r = subprocess.run(["gcc", "-std=c11", "-O1", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(root, "include"), "-o", str(tmp_path / "ex"),
                        os.path.join(root, "examples", "decode_rawfiles.c"), "-L", os.path.join(root, "abracadabra_amd"), "-l:libdabsdr_amd.so"],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
