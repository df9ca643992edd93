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


def test_probe_library_is_separate_from_the_product(built):
    # bench.py's `roofline.power_limited` aid: its own shared object, one entry point, nothing of it in the product library
    probe = ctypes.CDLL(os.path.join(ROOT, 'meta-transfer-learning_amd', 'libmtl_probe.so'))
    assert hasattr(probe, 'mtl_probe_mfma_f16')
    assert not hasattr(ctypes.CDLL(built._lib.LIB_PATH), 'mtl_probe_mfma_f16')
    assert 'mtl_probe' not in open(os.path.join(ROOT, 'include', 'mtl_hip.h')).read()


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


def test_command_list_dispatch_is_generated_from_the_header(built):
    """csrc/mtl_cmdlist_gen.inc (the switch mtl_cmdlist_run dispatches through) must be what tools/gen_cmdlist.py produces from the
    CURRENT include/mtl_hip.h, and every recordable function must resolve to an opcode."""
    import importlib.util
    spec = importlib.util.spec_from_file_location('gen_cmdlist', os.path.join(ROOT, 'tools', 'gen_cmdlist.py'))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    have = open(os.path.join(ROOT, 'meta-transfer-learning_amd', 'csrc', 'mtl_cmdlist_gen.inc')).read()
    assert have == gen.generate(), 'run `python tools/gen_cmdlist.py` and rebuild'
    L = built._lib.lib()
    for name in ('mtl_gemm_f32_ex', 'mtl_attn_fwd', 'mtl_attn_bwd', 'mtl_layernorm_bwd', 'mtl_event_record', 'mtl_memcpy_d2d'):
        assert L.mtl_cmdlist_opcode(name.encode()) >= 0, name
    for name in ('mtl_colsum_workspace', 'mtl_cmdlist_run', 'mtl_abi_version', 'nope'):
        assert L.mtl_cmdlist_opcode(name.encode()) == -1, name


def test_command_list_records_only_successful_calls_and_reports_failures(built):
    """host side of the command lists without a GPU: a Recorder executes and logs, size queries are not logged, a failing call
    is reported with its index (argument validation happens before any launch)."""
    lib_mod = built._lib
    L = lib_mod.lib()
    cl = lib_mod.CommandList()
    rec = lib_mod.Recorder(L, cl)
    assert rec.mtl_colsum_workspace(100, 64) == L.mtl_colsum_workspace(100, 64)
    assert rec.mtl_adam_step(None, None, None, None, None, 1, 1e-3, 0.9, 0.999, 1e-8, 16) == -22        # rejected -> not recorded
    assert len(cl.entries) == 0
    cl.add(L.mtl_cmdlist_opcode(b'mtl_adam_step'), lib_mod._kinds('mtl_adam_step'), (None, None, None, None, None, 1, 1e-3, 0.9, 0.999, 1e-8, 16))
    cl.finish()
    with pytest.raises(RuntimeError, match='command 0 failed with code -22'):
        cl.run()


def test_product_routing_is_a_pure_host_decision(built):
    """mtl_gemm_f32_ex_route (what bench.py uses to attribute launch timings): the bf16-split engine from its tile-count threshold,
    its split-K form for few-tile / very-long-K products, the small-tile engine for K-batched and row-sum products, the fp32 tile
    engine otherwise; mtl_lstm_*_supported: the shape limits of the persistent LSTM launches.  No GPU involved."""
    L = built._lib.lib()
    old = L.mtl_gemm_x3_min_tiles(-1)
    try:
        L.mtl_gemm_x3_min_tiles(128)
        assert L.mtl_gemm_f32_ex_route(2000, 512, 512, 8, 1, 0) == 2          # 16 x 4 x 8 tiles of 128 x 128
        assert L.mtl_gemm_f32_ex_route(808, 100, 512, 1, 1, 0) == 1           # 7 tiles: small-tile engine
        assert L.mtl_gemm_f32_ex_route(700, 512, 10000, 1, 1, 0) == 2         # 24 tiles x K = 10000: ten K slices
        assert L.mtl_gemm_f32_ex_route(700, 512, 10001, 1, 1, 0) == 2         # (slices are multiples of 32 deep, the last one ragged)
        assert L.mtl_gemm_f32_ex_route(808, 512, 3765, 1, 1, 0) == 2          # one-task vocabulary projection dX: 28 tiles, K >= 2048
        assert L.mtl_gemm_f32_ex_route(808, 512, 1500, 1, 1, 0) != 2          # too shallow to split
        assert L.mtl_gemm_f32_ex_route(700, 512, 10000, 1, 1, 1) == 1         # row sums ride on the small-tile engine
        assert L.mtl_gemm_f32_ex_route(700, 512, 2000, 1, 1, 0) == 1          # K below the split-K threshold (2048)
        L.mtl_gemm_x3_min_tiles(0)
        assert L.mtl_gemm_f32_ex_route(2000, 512, 512, 8, 1, 0) != 2          # engine off
    finally:
        L.mtl_gemm_x3_min_tiles(old)
    assert L.mtl_lstm_layer_supported(20, 512) == 1 and L.mtl_lstm_layer_supported(33, 512) == 0 and L.mtl_lstm_layer_supported(20, 500) == 0
    assert L.mtl_lstm_stack_supported(20, 512, 2) == 1 and L.mtl_lstm_stack_supported(20, 512, 3) == 0      # (2 NL - 1) H / 8 <= 224 workgroups
    assert L.mtl_lstm_stack_supported(20, 256, 3) == 1 and L.mtl_lstm_stack_supported(20, 128, 5) == 0
    assert L.mtl_lstm_stack_scratch(35, 20, 512, 2) == 35 * 64 * 20 * 512 * 4 and L.mtl_lstm_stack_scratch(35, 20, 512, 1) == 0
    assert L.mtl_lstm_layer_workspace() >= 4096


def test_mtl_lib_selects_another_build_of_the_same_abi(tmp_path):
    """MTL_LIB: the package loads the library at that path (A/B runs against the probe builds) and still checks its ABI hash"""
    import shutil
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    built = os.path.join(root, 'meta-transfer-learning_amd', 'libmtl_hip.so')
    other = str(tmp_path / 'libmtl_other.so')
    shutil.copy(built, other)
    code = ("import sys; sys.path.insert(0, %r); import mtl_amd; L = mtl_amd._lib.lib(); "
            "assert mtl_amd._lib.LIB_PATH == %r and L.mtl_abi_version() == mtl_amd._lib.ABI_VERSION; "
            "print(open('/proc/self/maps').read().count('libmtl_other.so') > 0)" % (root, other))
    r = subprocess.run([sys.executable, '-c', code], env=dict(os.environ, MTL_LIB=other), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.strip().endswith('True'), r.stderr[-2000:]
    r = subprocess.run([sys.executable, '-c', code], env=dict(os.environ, MTL_LIB=str(tmp_path / 'missing.so')), capture_output=True, text=True, timeout=300)
    assert r.returncode != 0                              # a missing library fails loudly: no fallback
