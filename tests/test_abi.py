"""CPU: the C-ABI shared library loads and exports every symbol include/cornetto_accel.h declares; the
no-GPU behaviour is a clean status, never a fallback computation."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    hdr = open(os.path.join(ROOT, "include", "cornetto_accel.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(cornetto_[a-z0-9_]+)\s*\(", hdr)))


@pytest.fixture(scope="module")
def lib():
    import cornetto_amd
    if not os.path.exists(cornetto_amd.LIB_PATH):
        cornetto_amd.build()
    return cornetto_amd.lib()


def test_exports_every_declared_symbol(lib):
    names = declared_symbols()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), n
    assert sorted(lib._declared) == names


def test_the_product_build_has_no_development_switches():
    """every environment name the library sources mention (csrc/: CN_DEV_INT("CORNETTO_...")) is in the development build and NOT in the product
    build, whose only environment variable is CORNETTO_DEVICE: no knob in the shipped library can change what a scan returns (round 5's
    CORNETTO_SIFT_ABL did); the development build exports the same entry points"""
    import cornetto_amd
    srcdir = os.path.join(ROOT, "cornetto_amd", "csrc")
    names = set()
    for f in os.listdir(srcdir):
        if f.endswith((".hip", ".hpp")):
            names |= set(re.findall(r'CN_DEV_INT\("(CORNETTO_[A-Z0-9_]+)"', open(os.path.join(srcdir, f)).read()))
            assert not re.findall(r'getenv\("CORNETTO_(?!DEVICE")', open(os.path.join(srcdir, f)).read()), f
    assert len(names) >= 25 and "CORNETTO_SIFT_ABL" in names and "CORNETTO_SDUST_EST_FORCE" in names
    prod = open(cornetto_amd.LIB_PATH, "rb").read()
    dev = open(cornetto_amd.DEV_LIB_PATH, "rb").read()
    in_prod = set(m.decode() for m in re.findall(rb"CORNETTO_[A-Z0-9_]+", prod))
    assert in_prod == {"CORNETTO_DEVICE"}, in_prod
    for n in names:
        assert n.encode() in dev, n
    d = cornetto_amd.lib(dev=True)
    assert sorted(d._declared) == declared_symbols()


def test_no_device_is_a_status_not_a_fallback(lib):
    import ctypes as C
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    assert lib.cornetto_accel_device_count() == 0
    h = C.c_void_p()
    assert lib.cornetto_accel_open(C.byref(h), 0, None) == -1       # CORNETTO_E_NODEVICE
    n = C.c_int(0)
    assert not lib.cornetto_sdust(None, C.c_char_p(b"ACGT"), 4, 20, 64, C.byref(n))
    assert n.value == -1


def test_pure_host_helpers(lib):
    assert lib.cornetto_n_reg(2551, 2500, 50) == 3
    assert lib.cornetto_n_reg(120, 2500, 50) == 1
    # the asserts of get_regs() (src/boringbits_main.c:353,368): never with inc <= w; with inc > w the last window must reach the end
    assert lib.cornetto_regs_assert(2551, 2500, 50) == 0 and lib.cornetto_regs_assert(120, 2500, 50) == 0
    assert lib.cornetto_regs_assert(15001, 64, 1000) == 0 and lib.cornetto_regs_assert(15661, 64, 1000) == 353
    assert lib.cornetto_regs_assert(2450, 300, 350) == 353 and lib.cornetto_regs_assert(2451, 300, 350) == 0
    assert lib.cornetto_cov_threshold(0.6, 22) == 13
    assert lib.cornetto_cov_threshold(1.6, 22) == 35
    assert abs(lib.cornetto_telowin_threshold(0.4, 99.9) - 0.397606) < 1e-6
    assert lib.cornetto_accel_strerror(-5) == b"parameter outside the supported range"


def test_documents_name_only_declared_entry_points():
    """every cornetto_*() call INTEGRATION.md / DESIGN.md / README.md show is an entry point of the header, and the
    header cites a reference location (file:line) for the units it replaces"""
    names = set(declared_symbols())
    for doc in ("INTEGRATION.md", "DESIGN.md", "README.md"):
        text = open(os.path.join(ROOT, doc)).read()
        for n in set(re.findall(r"\b(cornetto_[a-z0-9_]+)\s*\(", text)):
            if n.endswith("_t") or n in ("cornetto_amd",):
                continue
            assert n in names, (doc, n)
    hdr = open(os.path.join(ROOT, "include", "cornetto_accel.h")).read()
    assert len(re.findall(r"src/[a-z_/]+\.[ch]:\d+", hdr)) >= 25
