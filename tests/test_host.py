"""CPU tests of the product's host side: C-ABI surface, std::sort emulation, replay harness,
and the multi-rank slab/gather logic over gloo (world_size 2)."""
import ctypes as C
import os
import re
import subprocess
import sys

import numpy as np
import pytest

import replay

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_c_abi_exports_every_declared_symbol():
    import gpismap_amd
    L = C.CDLL(gpismap_amd.LIB_PATH)        # loads without a GPU
    hdr = open(os.path.join(ROOT, "include", "gpismap_amd.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    names = set(re.findall(r"\b(gpis[0-9a-z_]*)\s*\(", hdr))
    assert len(names) >= 28
    for n in sorted(names):
        assert hasattr(L, n), "missing symbol " + n
    L.gpis_version.restype = C.c_char_p
    assert b"gfx950" in L.gpis_version()


def test_no_cpu_fallback_without_gpu():
    """Without a HIP device the compute entry points must fail loudly, never fall back."""
    import gpismap_amd
    if gpismap_amd.device_count() > 0:
        pytest.skip("GPU present")
    with pytest.raises(gpismap_amd.GpisError):
        gpismap_amd.GPisMap3()
    with pytest.raises(gpismap_amd.GpisError):
        gpismap_amd.ObsGP()
    with pytest.raises(gpismap_amd.GpisError):
        gpismap_amd.OnGPIS(3, 0.04)


def test_product_does_not_reference_the_oracle():
    bad = []
    for dp, _, fs in os.walk(os.path.join(ROOT, "gpismap_amd")):
        for f in fs:
            if f.endswith((".py", ".cpp", ".hip", ".h", "Makefile")):
                txt = open(os.path.join(dp, f), errors="ignore").read()
                if re.search(r"oracle[/_]|libgpis_oracle|arbiter64", txt):
                    bad.append(os.path.join(dp, f))
    assert not bad, bad


def test_stdsort_emulation_matches_libstdcxx(tmp_path):
    src = tmp_path / "t.cpp"
    src.write_text(r'''
#include <algorithm>
#include <cstdio>
#include <random>
#include <vector>
#include "%s/gpismap_amd/csrc/stdsort_emul.h"
int main() {
    std::mt19937 rng(5);
    long bad = 0;
    for (int trial = 0; trial < 60000; ++trial) {
        int n = 2 + rng() %% 127;
        int levels = 1 + rng() %% 6;
        std::vector<float> key(n);
        for (auto& k : key) k = (float)(rng() %% levels) * 0.25f;
        std::vector<int> a(n), b(n);
        for (int i = 0; i < n; ++i) a[i] = b[i] = i;
        std::sort(a.begin(), a.end(), [&](int x, int y) { return key[x] < key[y]; });
        if (!gpis::stdsort_emulate(key.data(), b.data(), n) || a != b) ++bad;
    }
    std::printf("%%ld\n", bad);
    return bad != 0;
}
''' % ROOT)
    exe = tmp_path / "t"
    subprocess.check_call(["g++", "-O2", "-std=c++17", str(src), "-o", str(exe)])
    assert subprocess.check_output([str(exe)]).strip() == b"0"


def test_replay_harness_shapes():
    seq = replay.demo3_sequence()
    assert len(seq) == 40 and seq[0] == (93, 1) and seq[1] == (102, 2)
    g = replay.demo3_grid()
    assert g.shape == (21 * 25 * 29, 3)
    assert np.allclose(g[0], [-0.07, -0.10, 0.0]) and np.allclose(g[1], [-0.07, -0.09, 0.0])   # y fastest
    fr = replay.load_bigbird()
    assert len(fr) == 40 and fr[0]["depth"].shape == (307200,)
    d = replay.synthetic_depth(0)
    # column-major: index = col*480 + row; centre pixel (310, 224) is exactly 1 m
    assert d.shape == (307200,) and abs(d[310 * 480 + 224] - 1.0) < 1e-7
    q = replay.synthetic_grid(4)
    assert q.shape == (64, 3) and q[1, 0] > q[0, 0] and q[1, 1] == q[0, 1]                      # x fastest


def _gloo_worker(rank, world, port, out):
    import torch
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    from gpismap_amd import sharding
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n = 1003
    full = torch.arange(n * 8, dtype=torch.float32).reshape(n, 8)
    ok = True
    # block-cyclic cut (ragged last block, more blocks than ranks, fewer blocks than ranks)
    for block in (64, 100, 700, 5000):
        mine = sharding.take_blocks(full, world, rank, block).clone()      # stands for this rank's test() output
        assert mine.shape[0] == sharding.local_count(n, world, rank, block)
        got = sharding.gather_blocks(mine, n, world, rank, dst=0, block=block)
        if rank == 0:
            ok = ok and bool(torch.equal(got, full))
    parts = sharding.shard_clusters([5.0, 1.0, 9.0, 3.0, 3.0, 2.0, 8.0], world)
    ok = ok and sorted(sum(parts, [])) == list(range(7))
    out.put((rank, ok, sharding.local_count(n, world, rank, 64)))
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_query_cut_and_gather_gloo(world):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 500) + world
    ps = [ctx.Process(target=_gloo_worker, args=(r, world, port, q)) for r in range(world)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=120) for _ in ps)
    for p in ps:
        p.join(timeout=60)
    assert all(r[1] for r in res), res
    assert sum(r[2] for r in res) == 1003


def test_block_cyclic_cut_covers_every_query_once_and_balances_z():
    from gpismap_amd import sharding
    n = 256 ** 3
    seen = np.zeros(n // 65536, dtype=np.int32)
    for r in range(8):
        for lo, hi in sharding.cyclic_blocks(n, 8, r):
            assert hi - lo == 65536
            seen[lo // 65536] += 1
    assert np.all(seen == 1)
    # one 64 K block = one x-y sheet of the 256^3 grid: every rank gets every 8th z level
    assert [lo // 65536 for lo, _ in sharding.cyclic_blocks(n, 8, 3)][:3] == [3, 11, 19]
    assert sharding.local_count(1003, 4, 3, 100) == 200 and sharding.local_count(1003, 4, 2, 100) == 203
    # LPT partition: deterministic, balanced
    costs = [float(k) ** 3 for k in (936, 900, 850, 800, 790, 700, 650, 640, 600, 500, 400, 300, 200, 120)]
    parts = sharding.shard_clusters(costs, 4)
    loads = [sum(costs[i] for i in p) for p in parts]
    assert max(loads) / (sum(loads) / 4) < 1.25 and parts == sharding.shard_clusters(costs, 4)


def _build_dropin(tmp_path):
    import subprocess
    exe = str(tmp_path / "dropin_demo")
    libdir = os.path.join(ROOT, "gpismap_amd")
    cmd = ["g++", "-std=c++11", "-pthread", "-fPIC", "-O1", "-I" + os.path.join(ROOT, "include"),
           os.path.join(ROOT, "tests", "cpp", "dropin_demo.cpp"), "-L" + libdir, "-lgpismap_amd",
           "-Wl,-rpath," + libdir, "-o", exe]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return exe


def test_cpp_surface_builds_with_the_gateway_flags(tmp_path):
    """include/GPisMap3.h and include/GPisMap.h used the way mexGPisMap3.cpp / mexGPisMap.cpp use the
    reference classes, compiled with the gateways' own flags (-std=c++11 -pthread -fPIC) and linked against
    the library.  Without a GPU the objects construct, refuse to work loudly and return false -- no fallback."""
    import subprocess
    exe = _build_dropin(tmp_path)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0
    assert "map_dimension 2" in r.stdout
    assert "test_before_update 0" in r.stdout and "test_wrong_dim 0" in r.stdout
    import gpismap_amd
    if gpismap_amd.device_count() < 1:
        assert "map3 ok 0" in r.stdout and "map2 ok 0" in r.stdout
        assert "HIP device unavailable" in r.stderr


def test_flat_tree_matches_oracle_tree(tmp_path):
    """Host spatial index: the product's flat tree (one child visited per level where the reference loops over all
    2^d) against the oracle's restatement of octree.cpp / quadtree.cpp on random insert / remove / range-query
    sequences with near-duplicates, points ON splitting planes and root growth -- every return value and the
    traversal order of the stored points must be identical (tests/cpp/tree_check.cpp)."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "tree_check")
    r = subprocess.run(["g++", "-O2", "-std=c++17", "-I" + os.path.join(root, "gpismap_amd", "csrc"), "-I" + os.path.join(root, "oracle"),
                        os.path.join(root, "tests", "cpp", "tree_check.cpp"), "-o", exe], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    r = subprocess.run([exe, "30000"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.count("identical") == 12 and "MISMATCH" not in r.stdout


def test_flat_tree_under_address_and_ub_sanitizers(tmp_path):
    """The host spatial index and the oracle's tree under AddressSanitizer + UndefinedBehaviorSanitizer (CPU build only: GPU ASan is
    not available on the pool): the same random insert / remove / range-query / frozen-witness sequences as above, any
    out-of-bounds index, use-after-free of a recycled node or signed overflow aborts the run."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "tree_check_san")
    r = subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
                        "-I" + os.path.join(root, "gpismap_amd", "csrc"), "-I" + os.path.join(root, "oracle"),
                        os.path.join(root, "tests", "cpp", "tree_check.cpp"), "-o", exe], capture_output=True, text=True)
    if r.returncode != 0 and ("asan" in r.stderr.lower() or "sanitize" in r.stderr.lower()):
        pytest.skip("sanitizer runtime not available to g++ here: " + r.stderr.strip().splitlines()[-1])
    assert r.returncode == 0, r.stderr
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0")
    r = subprocess.run([exe, "8000"], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert r.stdout.count("identical") == 12 and "MISMATCH" not in r.stdout


def test_no_kernel_spills_or_uses_scratch():
    """the gfx950 code objects of the built library, read through llvm-readelf --notes
    (tools/kernel_resources.py): no kernel may spill a vector register or carry a private (scratch) segment -- round 2's
    factorisation kernels spilled 94-134 registers (hoisted 64-bit addresses of the column-major factor, a by-value
    descriptor copy, volatile LDS accesses through generic pointers), the lookup kernel held 1.5 KB of arrays per lane."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import kernel_resources
    lib = os.path.join(ROOT, "gpismap_amd", "libgpismap_amd.so")
    t = kernel_resources.kernel_table(lib)
    names = kernel_resources.demangle(list(t))
    assert len(t) >= 30, len(t)
    assert any("ongpis_train_fused_kernel" in names[k] for k in t) and any("ongpis_eval_kernel" in names[k] for k in t)
    bad = {names[k]: (r["spill"], r["scratch"]) for k, r in t.items() if r["spill"] or r["scratch"]}   # (SGPRs parked in VGPR lanes are not memory traffic)
    assert not bad, bad
    # the archived experiments (tools/experiments/) are not in the product build
    assert not [names[k] for k in t if "eval_small" in names[k] or "chol_async" in names[k]]
    # the dominant kernel keeps its occupancy: K4 at most 128 VGPRs (4 wavefronts per SIMD)
    assert all(r["vgpr"] <= 128 for k, r in t.items() if "ongpis_eval_kernel" in names[k])


def test_exp_table_of_the_kernels_matches_its_generator():
    """csrc/exp_tab.h (the table-driven double-precision exponential of K4's and K2's generation, round 6) holds a 64-entry table
    of 2^(j/64) as (hi, lo) pairs and the reduction constants; tools/exp_table.py generates them with 80-digit arithmetic and
    checks the scheme against the 80-digit exponential.  The committed header must carry exactly the generator's values, and the
    scheme must stay within 0.52 ulp on its working range."""
    import re
    import subprocess
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "exp_table.py")], capture_output=True, text=True, check=True).stdout
    gen_rows = re.findall(r"\{(-?0x[0-9a-fp.+-]+), (-?0x[0-9a-fp.+-]+)\}", out)
    hdr = open(os.path.join(ROOT, "gpismap_amd", "csrc", "exp_tab.h")).read()
    hdr_rows = re.findall(r"\{(-?0x[0-9a-fp.+-]+), (-?0x[0-9a-fp.+-]+)\}", hdr)
    assert len(gen_rows) == 64 and hdr_rows == gen_rows
    consts = re.search(r"64 / ln 2 = (\S+) ; ln 2 / 64 = (\S+) \+ (\S+)", out).groups()
    for c in consts:
        assert c in hdr or ("-" + c) in hdr, c
    worst = float(re.search(r"worst error ([0-9.]+) ulp", out).group(1))
    assert worst <= 0.52
