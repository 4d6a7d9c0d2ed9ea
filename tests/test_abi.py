"""CPU-side checks of the drop-in boundary: the shared library loads, exports every symbol the
public header declares, and the binding table covers the header (no compute calls without a GPU)."""
import ctypes
import os

import pytest

import causalgpslc_jl_amd as gp
from causalgpslc_jl_amd import _lib


@pytest.fixture(scope="module", autouse=True)
def _built_library():
    """The library is a build product (git-ignored): build it when this checkout has none yet — hipcc
    cross-compiles gfx950 without a GPU (what __graft_entry__.build() does)."""
    if not os.path.exists(_lib.LIB_PATH):
        import subprocess
        subprocess.run(["make", "-C", os.path.dirname(_lib.LIB_PATH)], check=True, capture_output=True, timeout=1500)
    yield


def test_header_and_binding_table_agree():
    hdr = set(_lib.header_symbols())
    assert hdr == set(_lib.SIGNATURES), (hdr ^ set(_lib.SIGNATURES))
    assert len(hdr) >= 15


def test_library_loads_and_exports_every_declared_symbol():
    assert os.path.exists(_lib.LIB_PATH), "build the library first (python -c 'import __graft_entry__ as g; g.build()')"
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in _lib.header_symbols():
        assert hasattr(lib, name), f"{name} declared in include/gpslc_hip.h but not exported"
    gp.load_library()
    assert b"gfx950" in _lib.load().gpslc_version()


def test_host_only_entry_point_sate_samples():
    # gpslc_sate_samples is pure host arithmetic (src/estimation.jl:148-163): callable without a GPU
    import numpy as np
    out = gp.SATEsamples(np.array([1.0, 2.0]), np.array([0.25, 4.0]), 2, z=np.array([1.0, -1.0, 0.5, 2.0]))
    assert np.allclose(out, [1.25, 0.75, 4.0, 10.0])
    import gpslc_oracle as orc
    out = gp.SATEsamples(np.array([0.0]), np.array([1.0]), 6, seed=99)
    ref = orc.philox_normals(99, (1 << 40) + 0, 6)
    assert np.allclose(out, ref, rtol=0, atol=1e-15)


def test_shard_range_is_the_python_drivers_partition():
    """gpslc_shard_range (host-only): the partition gpslc_predict_multi uses = causalgpslc.jl_amd/sharded.py: shard_range, for
    every (S, nblocks) a node would see; blocks are contiguous, ordered, cover [0, S) and differ in size by at most one."""
    import ctypes as C
    from causalgpslc_jl_amd.sharded import shard_range
    lib = _lib.load()
    s0, s1 = C.c_int64(), C.c_int64()
    for S in (0, 1, 2, 5, 7, 8, 64, 1000, 8191, 8192):
        for nb in (1, 2, 3, 4, 7, 8):
            end = 0
            for k in range(nb):
                assert lib.gpslc_shard_range(S, nb, k, C.byref(s0), C.byref(s1)) == 0
                assert (s0.value, s1.value) == shard_range(S, nb, k)
                assert s0.value == end and 0 <= s1.value - s0.value - S // nb <= 1
                end = s1.value
            assert end == S
    assert lib.gpslc_shard_range(-1, 2, 0, C.byref(s0), C.byref(s1)) == -1
    assert lib.gpslc_shard_range(4, 0, 0, C.byref(s0), C.byref(s1)) == -2
    assert lib.gpslc_shard_range(4, 2, 2, C.byref(s0), C.byref(s1)) == -3
    assert lib.gpslc_shard_range(4, 2, 0, None, C.byref(s1)) == -4


def test_argument_validation_mirrors_reference_asserts():
    import numpy as np
    with pytest.raises(AssertionError):   # src/kernel.jl:25 "X1 and X2 are different sizes!"
        gp.rbfKernelLog(np.ones((3, 2)), np.ones((4, 2)), 1.0)
    with pytest.raises(AssertionError):   # src/kernel.jl:14-16
        gp.rbfKernelLog(np.ones((3, 2)), np.ones((3, 2)), np.ones(3))
    with pytest.raises(ValueError):       # zero step: Julia's range constructor throws
        gp.doTRange(1.0, 1.0, 10)


def test_missing_library_fails_loudly(monkeypatch):
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libgpslc_hip.so")
    with pytest.raises(_lib.GPSLCLibraryError):
        _lib.load()


def test_struct_layouts_match_the_c_header(tmp_path):
    """gpslc_node and gpslc_pack_header as ctypes declares them vs. what a C compiler makes of include/gpslc_hip.h
    (size and every field offset): the layout a Julia `struct` / any other FFI has to reproduce."""
    import subprocess
    src = tmp_path / "layout.c"
    src.write_text(
        '#include <stdio.h>\n#include <stddef.h>\n#include "gpslc_hip.h"\n'
        'int main(void) {\n'
        '  printf("%zu %zu %zu %zu %zu %zu %zu\\n", sizeof(gpslc_node), offsetof(gpslc_node, nF), offsetof(gpslc_node, F),\n'
        '         offsetof(gpslc_node, ls), offsetof(gpslc_node, scale), offsetof(gpslc_node, noise), offsetof(gpslc_node, target));\n'
        '  printf("%zu %zu %zu %zu\\n", sizeof(gpslc_pack_header), offsetof(gpslc_pack_header, S),\n'
        '         offsetof(gpslc_pack_header, binary_t), offsetof(gpslc_pack_header, hyper));\n'
        '  return 0; }\n')
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-std=c99", "-I", os.path.dirname(_lib.HEADER_PATH), str(src), "-o", str(exe)], check=True)
    node, pack = [tuple(int(v) for v in line.split()) for line in
                  subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.strip().splitlines()]
    N, P = _lib.Node, _lib.PackHeader
    assert node == (ctypes.sizeof(N), N.nF.offset, N.F.offset, N.ls.offset, N.scale.offset, N.noise.offset, N.target.offset)
    assert pack == (ctypes.sizeof(P), P.S.offset, P.binary_t.offset, P.hyper.offset)
