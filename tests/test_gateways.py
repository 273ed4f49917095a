"""The drop-in boundary, checked against the reference's own gateway sources.

CPU: mex/mexGPisMap3.cpp and mex/mexGPisMap.cpp of the reference compile UNCHANGED, by path, against include/
with the reference's own flags (mex/make_GPisMap3.m:15) and link against libgpismap_amd.so; the make_*_amd.m twins
exist and name the same gateway files.  GPU (test_gpu_gateways.py): the built gateways are driven like MATLAB
drives them."""
import ctypes as C
import os
import re
import subprocess

import pytest

import mexdrive

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
needs_ref = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "mex")), reason="reference tree not present (GPU box)")


@needs_ref
@pytest.mark.parametrize("gw", ["mexGPisMap3", "mexGPisMap"])
def test_reference_gateway_compiles_unchanged_by_path(gw):
    cmd = ["g++", "-std=c++11", "-pthread", "-fPIC", "-fsyntax-only", "-Wall",
           "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "tests", "cpp", "mexstub"),
           os.path.join(REF, "mex", gw + ".cpp")]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


@needs_ref
def test_gateways_link_against_the_library_and_export_mexFunction():
    mexdrive.build()
    for gw in ("mexGPisMap3", "mexGPisMap"):
        p = mexdrive.gateway_path(gw)
        assert os.path.exists(p)
        L = C.CDLL(p)                       # resolves GPisMap3::update/test/... from libgpismap_amd.so; no GPU needed
        assert hasattr(L, "mexFunction")
        syms = subprocess.check_output(["nm", "-D", "--undefined-only", p], text=True)
        cls = "GPisMap3" if gw.endswith("3") else "GPisMap"
        need = ["update", "test", "reset"] + (["getAllPoints", "resetCam"] if cls == "GPisMap3" else [])
        for m in need:
            assert re.search(r"_ZN\d+%s\d+%s" % (cls, m), syms), (gw, m)


@needs_ref
def test_gateway_without_gpu_fails_loudly_not_silently(capfd):
    """'update' through the real gateway on a box without a HIP device: no crash, no exception across the
    boundary, message on stderr, 'test' creates its zero-filled output and leaves it untouched."""
    import numpy as np
    import gpismap_amd
    if gpismap_amd.device_count() > 0:
        pytest.skip("GPU present")
    mexdrive.build()
    g = mexdrive.Gateway("mexGPisMap3")
    g.call(0, "setCamera", np.array([[1.0]]), "bigbird")
    g.call(0, "update", np.zeros((480, 640), dtype=np.float32), np.zeros((1, 12), dtype=np.float32))
    out = g.call(1, "test", np.zeros((3, 5), dtype=np.float32))
    assert len(out) == 1 and out[0].shape == (8, 5) and not out[0].any()
    assert g.call(1, "getAllPoints") == []
    g.call(0, "reset")
    assert "HIP device unavailable" in capfd.readouterr().err


def test_make_scripts_are_twins_of_the_reference_ones():
    for name, gw in (("make_GPisMap3_amd.m", "mexGPisMap3.cpp"), ("make_GPisMap_amd.m", "mexGPisMap.cpp")):
        txt = open(os.path.join(ROOT, "mex", name)).read()
        assert gw in txt and "-lgpismap_amd" in txt and "CXXFLAGS=-std=c++11 -pthread -fPIC" in txt
        assert "eigen" not in txt.lower().replace("eigen sources", "").replace("eigen include", "")
    if os.path.isdir(os.path.join(REF, "mex")):
        ref = open(os.path.join(REF, "mex", "make_GPisMap3.m")).read()
        assert "CXXFLAGS=-std=c++11 -pthread -fPIC" in ref      # same flags as the twin
