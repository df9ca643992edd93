"""ctypes binding of libmtl_hip.so (the C ABI declared in include/mtl_hip.h).

The HIP library is the product's only compute path: there is NO CPU / eager fallback.  If the shared
object is missing, `lib()` raises with the build command instead of silently degrading.
"""
import ctypes
import os
from ctypes import c_float, c_int, c_long, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'libmtl_hip.so')
ABI_VERSION = 1

P, I, L, F = c_void_p, c_int, c_long, c_float

# name -> (restype, argtypes); mirrors include/mtl_hip.h one-to-one (tests/test_abi.py checks both directions)
SIGNATURES = {
    'mtl_abi_version': (I, []),
    'mtl_gemm_f32': (I, [P, I, I, I, I, I, F, P, I, P, I, P, I, P, P, I, I, I, I, L, L, L, L, L, L, L, P, L]),
    'mtl_gemm_f32_ex': (I, [P, I, I, I, I, I, F, P, I, P, I, P, I, P, P, I, I, I, I, L, L, L, L, L, L, L, I, L, L, P, L, P, L]),
    'mtl_conv0_relu_fwd': (I, [P, P, P, P, P, I, I, I]),
    'mtl_conv0_wgrad_workspace': (L, []),
    'mtl_conv0_wgrad': (I, [P, P, P, P, P, P, I, I, I]),
    'mtl_conv3x3_wprep': (I, [P, P, P, P, I, I]),
    'mtl_conv3x3_relu_fwd': (I, [P, P, P, P, P, I, I, I, I, I]),
    'mtl_conv3x3_relu_pool_fwd': (I, [P, P, P, P, P, P, I, I, I, I, I]),
    'mtl_conv3x3_dgrad': (I, [P, P, P, P, P, P, I, I, I, I, I]),
    'mtl_conv3x3_wprep_x3': (I, [P, P, P, P, I, I]),
    'mtl_conv3x3_relu_fwd_x3': (I, [P, P, P, P, P, I, I, I, I, I]),
    'mtl_conv3x3_relu_pool_fwd_x3': (I, [P, P, P, P, P, P, I, I, I, I, I]),
    'mtl_conv3x3_dgrad_x3': (I, [P, P, P, P, P, P, I, I, I, I, I]),
    'mtl_conv3x3_wgrad_workspace': (L, [I, I, I, I, I, I]),
    'mtl_conv3x3_wgrad': (I, [P, P, P, P, P, P, L, I, I, I, I, I]),
    'mtl_conv3x3_wgrad_x3_workspace': (L, [I, I, I, I, I, I]),
    'mtl_conv3x3_wgrad_x3': (I, [P, P, P, P, P, P, L, I, I, I, I, I]),
    'mtl_permute_hc': (I, [P, P, P, I, I, I, I]),
    'mtl_layernorm_fwd': (I, [P, P, P, P, P, P, P, P, F, P, P, P, I, I, I, F]),
    'mtl_layernorm_bwd_workspace': (L, [I, I]),
    'mtl_layernorm_bwd': (I, [P, P, P, P, P, P, P, F, P, P, P, P, P, P, I, I]),
    'mtl_softmax_mask_fwd': (I, [P, P, P, I, F, I, I, I, I, I, P, F, P]),
    'mtl_softmax_bwd': (I, [P, P, P, F, L, I, I, P, F]),
    'mtl_attn_supported': (I, [I, I]),
    'mtl_attn_fwd': (I, [P, P, P, P, I, I, I, P, I, F, I, I, I, I, I, I, P, I, F, P, I, P]),
    'mtl_attn_bwd': (I, [P, P, P, P, I, I, I, P, I, F, I, I, I, I, I, I, P, I, F, P, P, I, P, P, P, P, P, I, I, I]),
    'mtl_embed_pe_fwd': (I, [P, P, P, P, P, I, I, I, P, F]),
    'mtl_embed_bwd': (I, [P, P, P, P, P, P, I, I, L, P, F]),
    'mtl_dropout_mask': (I, [P, P, L, F, P, ctypes.c_ulonglong]),
    'mtl_ce_argmax_fwd': (I, [P, P, P, I, I, I, L, F, I, P, P, P, P, P]),
    'mtl_ce_bwd': (I, [P, P, P, P, I, I, I, L, F, F, P, P, I]),
    'mtl_colsum_workspace': (L, [L, I]),
    'mtl_colsum_accum': (I, [P, P, L, I, L, P, P]),
    'mtl_sgd_theta_prime': (I, [P, P, P, F, P, L]),
    'mtl_axpy': (I, [P, P, P, F, L]),
    'mtl_copy_f32': (I, [P, P, P, L]),
    'mtl_scale': (I, [P, P, F, P, L]),
    'mtl_adam_step': (I, [P, P, P, P, P, I, F, F, F, F, L]),
    'mtl_sumsq': (I, [P, P, L, P, P, I, F]),
    'mtl_spect_logmag': (I, [P, P, I, I, I, P, P, I]),
    'mtl_levenshtein_u32': (I, [P, I, P, I]),
}

_lib = None


class MtlLibraryError(RuntimeError):
    pass


def lib():
    """Load (once) and return the ctypes handle; raise loudly if the HIP library has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise MtlLibraryError(
            'libmtl_hip.so not found at %s -- build it with `python -c "import __graft_entry__ as g; g.build()"` '
            '(hipcc --offload-arch=gfx950).  There is no CPU fallback for the product path.' % LIB_PATH)
    h = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(h, name)          # AttributeError here = header/library drift
        fn.restype = res
        fn.argtypes = args
    if h.mtl_abi_version() != ABI_VERSION:
        raise MtlLibraryError('libmtl_hip.so ABI %d != expected %d; rebuild' % (h.mtl_abi_version(), ABI_VERSION))
    _lib = h
    return h


def check(rc, what):
    if rc != 0:
        raise RuntimeError('%s failed with code %d' % (what, rc))


def levenshtein(a, b):
    """Edit distance between two python strings (host helper, code-point level)."""
    ua = (ctypes.c_uint32 * max(len(a), 1))(*[ord(c) for c in a])
    ub = (ctypes.c_uint32 * max(len(b), 1))(*[ord(c) for c in b])
    d = lib().mtl_levenshtein_u32(ua, len(a), ub, len(b))
    if d < 0:
        raise RuntimeError('mtl_levenshtein_u32 failed: %d' % d)
    return d
