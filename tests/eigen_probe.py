"""Run-time probe for a REAL Eigen installation and the driver of
tests/cpp/eigen_probe.cpp.  Test infrastructure; used by tests/test_eigen_probe.py and by bench.py's cpu_baseline leg
(`eigen_on_box`).  Nothing here touches /root/reference."""
import glob
import os
import re
import struct
import subprocess
import sys
import sysconfig

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
CASES = [("f2", "full"), ("f2", "sparse"), ("f3", "small"), ("f3", "medium"), ("f3", "large"), ("f3", "2d")]


def _is_real_eigen(inc):
    """A genuine Eigen include root has Eigen/Dense AND Eigen/src/Core/util/Macros.h with the version macros (a stand-in
    header set that merely offers an `Eigen/Dense` -- such as a survey session's API shim -- is refused)."""
    mac = os.path.join(inc, "Eigen", "src", "Core", "util", "Macros.h")
    if not (os.path.isfile(os.path.join(inc, "Eigen", "Dense")) and os.path.isfile(mac)):
        return None
    txt = open(mac, errors="ignore").read()
    m = [re.search(r"#define\s+EIGEN_%s_VERSION\s+(\d+)" % k, txt) for k in ("WORLD", "MAJOR", "MINOR")]
    if not all(m) or not os.path.isdir(os.path.join(inc, "Eigen", "src", "Cholesky")):
        return None
    return ".".join(x.group(1) for x in m)


def find_eigen():
    """(include dir, version string) of the first real Eigen found, else (None, reason)."""
    cand = []
    if os.environ.get("EIGEN3_INCLUDE_DIR"):
        cand.append(os.environ["EIGEN3_INCLUDE_DIR"])
    cand += ["/usr/include/eigen3", "/usr/local/include/eigen3", "/usr/include", "/usr/local/include", "/opt/local/include/eigen3"]
    for env in ("CONDA_PREFIX", "VIRTUAL_ENV"):
        if os.environ.get(env):
            cand += [os.path.join(os.environ[env], "include", "eigen3"), os.path.join(os.environ[env], "include")]
    for k in ("include", "platinclude"):
        p = sysconfig.get_paths().get(k)
        if p:
            cand += [os.path.join(p, "eigen3"), p, os.path.join(os.path.dirname(p), "eigen3")]
    # python packages that vendor Eigen headers (tensorflow, pybind11's eigen extras, torch third_party trees)
    for sp in {p for p in sys.path if p and os.path.isdir(p) and p.rstrip("/").endswith(("site-packages", "dist-packages"))}:
        for pat in ("tensorflow/include", "tensorflow/include/eigen3", "torch/include", "torch/include/eigen3", "*/include/eigen3", "eigen*/include", "*/third_party/eigen*"):
            cand += glob.glob(os.path.join(sp, pat))
    cand += glob.glob("/opt/*/include/eigen3") + glob.glob("/opt/rocm*/include/eigen3") + glob.glob("/usr/lib/*/include/eigen3")
    seen = set()
    for c in cand:
        c = os.path.realpath(c)
        if c in seen or not os.path.isdir(c):
            continue
        seen.add(c)
        v = _is_real_eigen(c)
        if v:
            return c, v
    return None, "no Eigen include tree with Eigen/Dense + Eigen/src/Core/util/Macros.h among %d candidate directories" % len(seen)


def write_inputs(path):
    z = np.load(os.path.join(GOLDEN, "fixtures_gp.npz"))
    with open(path, "wb") as f:
        f.write(struct.pack("<i", len(CASES)))
        for fam, name in CASES:
            if fam == "f2":
                x, fv, q = z["f2_%s_x" % name], z["f2_%s_f" % name], z["f2_%s_q" % name]
                f.write(struct.pack("<iiif", 0, x.shape[0], q.shape[0], 0.5))
                for a in (x, fv, q):
                    f.write(np.ascontiguousarray(a, dtype=np.float32).tobytes())
            else:
                nd, xq = z["f3_%s_nodes" % name], z["f3_%s_xq" % name]
                dim = 2 if name == "2d" else 3
                f.write(struct.pack("<iiif", dim, nd.shape[0], xq.shape[0], 1.2 if dim == 2 else 0.04))
                f.write(np.ascontiguousarray(nd, dtype=np.float32).tobytes())
                f.write(np.ascontiguousarray(xq, dtype=np.float32).tobytes())


def read_outputs(path):
    z = np.load(os.path.join(GOLDEN, "fixtures_gp.npz"))
    b = open(path, "rb").read()
    ver = struct.unpack_from("<4i", b, 0)
    o = 16
    out = {"version": "%d.%d.%d" % ver[:3], "vectorised": bool(ver[3])}

    def take(n, dt=np.float32):
        nonlocal o
        a = np.frombuffer(b, dtype=dt, count=n, offset=o).copy()
        o += a.nbytes
        return a

    for fam, name in CASES:
        if fam == "f2":
            n = z["f2_%s_x" % name].shape[0]; nq = z["f2_%s_q" % name].shape[0]
            out["f2_%s_L" % name] = np.tril(take(n * n).reshape(n, n).T)
            out["f2_%s_alpha" % name] = take(n); out["f2_%s_val" % name] = take(nq); out["f2_%s_var" % name] = take(nq)
        else:
            dim = 2 if name == "2d" else 3
            nq = z["f3_%s_xq" % name].shape[0]
            K = int(take(1, np.int32)[0])
            out["f3_%s_K" % name] = K
            out["f3_%s_alpha" % name] = take(K); out["f3_%s_pred" % name] = take(nq * 2 * (1 + dim)).reshape(nq, 2 * (1 + dim))
    assert o == len(b), (o, len(b))
    return out


def build_and_run(workdir, inc=None, flags=("-O", "-std=c++11")):
    """Compile tests/cpp/eigen_probe.cpp against the Eigen found (the reference's flags: mex's default -O and CXXFLAGS
    -std=c++11, mex/make_GPisMap3.m:15 -- the oracle headers it borrows the kernel functions from need C++17 features? no:
    gp.hpp is written in the C++11 subset) and run it on the committed F2 / F3 inputs.  Returns the outputs dict."""
    if inc is None:
        inc, why = find_eigen()
        if inc is None:
            raise FileNotFoundError(why)
    exe = os.path.join(workdir, "eigen_probe")
    src = os.path.join(ROOT, "tests", "cpp", "eigen_probe.cpp")
    cmd = ["g++"] + list(flags) + ["-I", inc, "-I", os.path.join(ROOT, "oracle"), src, "-o", exe]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0 and "-std=c++11" in flags:
        # the borrowed oracle headers may need a later language level than the reference's own sources; the arithmetic of
        # Eigen's kernels does not depend on -std
        cmd = ["g++"] + [("-std=c++17" if f == "-std=c++11" else f) for f in flags] + ["-I", inc, "-I", os.path.join(ROOT, "oracle"), src, "-o", exe]
        r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("eigen_probe does not compile against %s:\n%s" % (inc, r.stderr[-2000:]))
    fin, fout = os.path.join(workdir, "in.bin"), os.path.join(workdir, "out.bin")
    write_inputs(fin)
    subprocess.run([exe, fin, fout], check=True, timeout=300)
    return read_outputs(fout)


def compare_with_eigen33(out):
    """Bit-level and tolerance-level comparison of the probe's outputs with the committed eigen33 fixtures.
    Returns dict(name -> (identical elements, elements, max abs difference))."""
    z = np.load(os.path.join(GOLDEN, "samemap.npz"))
    rep = {}
    for fam, name in CASES:
        keys = ("L", "alpha", "val", "var") if fam == "f2" else ("alpha", "pred")
        for k in keys:
            a = out["%s_%s_%s" % (fam, name, k)]; b = z["%s_%s_eigen33_%s" % (fam, name, k)]
            rep["%s_%s_%s" % (fam, name, k)] = (int(np.sum(a.view(np.uint32) == b.view(np.uint32))), int(a.size), float(np.abs(a - b).max()))
    return rep


if __name__ == "__main__":
    inc, v = find_eigen()
    print("Eigen:", inc, v)
    if inc:
        import tempfile
        with tempfile.TemporaryDirectory() as d:
            o = build_and_run(d, inc)
            print("probe ran with Eigen", o["version"], "vectorised" if o["vectorised"] else "scalar")
            for k, r in compare_with_eigen33(o).items():
                print("  %-22s identical %6d / %6d   max |diff| %.3e" % (k, r[0], r[1], r[2]))
