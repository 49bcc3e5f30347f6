"""CPU-side checks of the drop-in boundary: the C-ABI library is built, loads, and exports every
symbol include/poccala_hip.h declares.  No compute calls (no GPU here)."""
import os
import re

import pytest

import poccala_amd._lib as L

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    text = open(os.path.join(ROOT, 'include', 'poccala_hip.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(pcl_[a-z0-9_]+)\s*\(', text)))


def test_library_is_built():
    assert os.path.exists(L.LIB_PATH), 'run __graft_entry__.build() first'


def test_every_declared_symbol_is_exported_and_bound():
    lib = L.load()
    names = header_symbols()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), n
        assert n in L.PROTOTYPES, 'no ctypes prototype for %s' % n
    assert sorted(L.PROTOTYPES) == names


def test_exported_pcl_symbols_are_exactly_the_header():
    """The converse of the test above: nothing named pcl_* leaves the library that the header does not
    declare -- internal helpers (C++ arguments behind extern "C") stay hidden (-fvisibility=hidden, push(default) in the header)."""
    import subprocess
    out = subprocess.run(['nm', '-D', '--defined-only', L.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = sorted({line.split()[-1] for line in out.splitlines() if line.split() and line.split()[-1].startswith('pcl_')})
    assert exported == header_symbols()


def test_no_gpu_fails_loudly():
    """Without a GPU the product path must raise, never fall back to a CPU implementation."""
    import ctypes
    lib = L.load()
    n = ctypes.c_int(0)
    try:
        hip = ctypes.CDLL('libamdhip64.so')
        rc = hip.hipGetDeviceCount(ctypes.byref(n))
    except OSError:
        rc = 1
    if rc == 0 and n.value > 0:
        pytest.skip('a GPU is present')
    from poccala_amd import Engine, PoccalaHipError
    with pytest.raises(PoccalaHipError):
        Engine(0)


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, 'poccala_amd')
    for root, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(('.py', '.hip', '.h', '.cpp')):
                src = open(os.path.join(root, f), errors='ignore').read()
                assert 'oracle' not in src.replace('"oracle"', '') or f == '__never__', (root, f)
