"""Test driver for the reference's UNCHANGED mex gateways.

oracle/_ref/mexGPisMap3_gw.so and mexGPisMap_gw.so are the reference's mex/mexGPisMap3.cpp and
mex/mexGPisMap.cpp compiled by path (oracle/Makefile, target `gateways`) against gpismap_amd's
include/ and a test-only stand-in for MATLAB's mex.h (tests/cpp/mexstub).  This module calls their
mexFunction the way MATLAB would: command string + single/double matrices in, matrices out."""
import ctypes as C
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REFDIR = os.path.join(ROOT, "oracle", "_ref")


def gateway_path(name):
    return os.path.join(REFDIR, name + "_gw.so")


def build():
    """(Re)build the gateway objects where the reference tree is present; no-op elsewhere."""
    import subprocess
    if os.path.isdir("/root/reference/mex"):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "gateways"])


class Gateway:
    def __init__(self, name):
        path = gateway_path(name)
        if not os.path.exists(path):
            raise FileNotFoundError(path)
        L = C.CDLL(path)
        vp = C.c_void_p
        L.mxstub_string.restype = vp; L.mxstub_string.argtypes = [C.c_char_p]
        L.mxstub_single.restype = vp; L.mxstub_single.argtypes = [C.POINTER(C.c_float), C.c_size_t, C.c_size_t]
        L.mxstub_double.restype = vp; L.mxstub_double.argtypes = [C.POINTER(C.c_double), C.c_size_t, C.c_size_t]
        L.mxstub_call.restype = C.c_int
        L.mxstub_call.argtypes = [C.c_int, C.POINTER(vp), C.c_int, vp, vp, vp, vp]
        L.mxGetData.restype = vp; L.mxGetData.argtypes = [vp]
        L.mxGetDimensions.restype = C.POINTER(C.c_size_t); L.mxGetDimensions.argtypes = [vp]
        L.mxGetClassID.restype = C.c_int; L.mxGetClassID.argtypes = [vp]
        L.mxDestroyArray.argtypes = [vp]
        self.L = L

    def _to_mx(self, a):
        if isinstance(a, str):
            return self.L.mxstub_string(a.encode())
        a = np.asarray(a)
        if a.ndim == 1:
            a = a.reshape(1, -1)
        m, n = a.shape                         # MATLAB matrix m x n, column-major storage
        f = np.asfortranarray(a)
        if a.dtype == np.float32:
            return self.L.mxstub_single(f.ctypes.data_as(C.POINTER(C.c_float)), m, n)
        f = np.asfortranarray(a, dtype=np.float64)
        return self.L.mxstub_double(f.ctypes.data_as(C.POINTER(C.c_double)), m, n)

    def call(self, nlhs, *args):
        """mexFunction(nlhs, plhs, len(args), prhs) -> list of numpy arrays (MATLAB shape) for created outputs."""
        mx = [self._to_mx(a) for a in args]
        pad = mx + [None] * (4 - len(mx))
        out = (C.c_void_p * 2)()
        self.L.mxstub_call(nlhs, out, len(mx), *pad)
        res = []
        for i in range(2):
            if out[i]:
                dims = self.L.mxGetDimensions(out[i])
                m, n = int(dims[0]), int(dims[1])
                cls = self.L.mxGetClassID(out[i])
                dt = np.float32 if cls == 7 else np.float64
                buf = (C.c_char * (m * n * np.dtype(dt).itemsize)).from_address(self.L.mxGetData(out[i]))
                res.append(np.frombuffer(buf, dtype=dt).reshape(m, n, order="F").copy())
                self.L.mxDestroyArray(out[i])
        for a in mx:
            self.L.mxDestroyArray(a)
        return res
