"""CPU: the C-ABI library builds for gfx950, loads, and exports exactly what include/mtl_hip.h declares."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope='module')
def built():
    import __graft_entry__ as ge
    return ge.build()


def header_functions():
    text = open(os.path.join(ROOT, 'include', 'mtl_hip.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(?:int|long)\s+(mtl_\w+)\s*\(', text)))


def test_header_and_binding_agree(built):
    names = header_functions()
    assert len(names) >= 25
    assert names == sorted(built._lib.SIGNATURES.keys())


def test_library_exports_every_declared_symbol(built):
    h = ctypes.CDLL(built._lib.LIB_PATH)
    for name in header_functions():
        assert hasattr(h, name), name
    assert h.mtl_abi_version() == built._lib.ABI_VERSION


def test_argument_validation_without_gpu(built):
    L = built._lib.lib()
    # bad arguments are rejected before any launch (no device needed)
    assert L.mtl_gemm_f32(None, 0, 1, 0, 4, 4, 1.0, None, 4, None, 4, None, 4, None, None, 0, 0, 1, 1, 0, 0, 0, 0, 0, 0, 0, None, 0) == -22
    assert L.mtl_adam_step(None, None, None, None, None, 1, 1e-3, 0.9, 0.999, 1e-8, 16) == -22
    assert L.mtl_conv3x3_wgrad_workspace(8, 1000, 161, 64, 64, 1) > 0
    assert L.mtl_layernorm_bwd_workspace(2000, 512) > 0


def test_levenshtein_host_helper(built):
    lev = built._lib.levenshtein
    assert lev('kitten', 'sitting') == 3 and lev('', 'abc') == 3 and lev('abc', '') == 3 and lev('', '') == 0
    assert lev('你好世界', '你世界好') == 2

    def dp(a, b):
        prev = list(range(len(b) + 1))
        for i, ca in enumerate(a, 1):
            cur = [i]
            for j, cb in enumerate(b, 1):
                cur.append(min(prev[j] + 1, cur[-1] + 1, prev[j - 1] + (ca != cb)))
            prev = cur
        return prev[-1]
    import random
    rnd = random.Random(0)
    for _ in range(200):
        a = ''.join(rnd.choice('abcd') for _ in range(rnd.randint(0, 30)))
        b = ''.join(rnd.choice('abcd') for _ in range(rnd.randint(0, 30)))
        assert lev(a, b) == dp(a, b)
