"""ctypes binding of libmtl_hip.so (the C ABI declared in include/mtl_hip.h).

The HIP library is the product's only compute path: there is NO CPU / eager fallback.  If the shared
object is missing, `lib()` raises with the build command instead of silently degrading.
"""
import ctypes
import os
from ctypes import c_float, c_int, c_long, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('MTL_LIB') or os.path.join(_HERE, 'libmtl_hip.so')     # MTL_LIB: another build of the same ABI (A/B measurements)
HEADER_PATH = os.path.join(os.path.dirname(_HERE), 'include', 'mtl_hip.h')


def abi_hash(text):
    """31-bit CRC of the C header without comments / white-space runs (same function as tools/gen_cmdlist.py, which bakes it into
    the library as mtl_abi_version()): any change of a prototype, a struct layout or the opcode order bumps it."""
    import re
    import zlib
    text = re.sub(r'/\*.*?\*/', ' ', text, flags=re.S)
    text = re.sub(r'//[^\n]*', ' ', text)
    return zlib.crc32(' '.join(text.split()).encode()) & 0x7fffffff


ABI_VERSION = abi_hash(open(HEADER_PATH).read()) if os.path.exists(HEADER_PATH) else None

P, I, L, F = c_void_p, c_int, c_long, c_float

# name -> (restype, argtypes); mirrors include/mtl_hip.h one-to-one (tests/test_abi.py checks both directions)
SIGNATURES = {
    'mtl_abi_version': (I, []),
    'mtl_gemm_f32': (I, [P, I, I, I, I, I, F, P, I, P, I, P, I, P, P, I, I, I, I, L, L, L, L, L, L, L, P, L]),
    'mtl_gemm_f32_ex': (I, [P, I, I, I, I, I, F, P, I, P, I, P, I, P, P, I, I, I, I, L, L, L, L, L, L, L, I, L, L, P, L, P, L, L, L]),
    'mtl_gemm_f32_tb': (I, [P, I, I, I, I, I, F, P, I, P, I, P, I, P, P, I, I, I, I, L, L, L, L, L, L, L, I, L, L, P, L, P, L, L, L,
                            I, L, L, L, L, L]),
    'mtl_gemm_f32_ex_route': (I, [I, I, I, I, I, I]),
    'mtl_gemm_x3_min_tiles': (I, [I]),
    'mtl_conv0_relu_fwd': (I, [P, P, P, P, P, I, I, I, P]),
    'mtl_conv0_wgrad_workspace': (L, []),
    'mtl_conv0_wgrad': (I, [P, P, P, P, P, P, I, I, I]),
    'mtl_conv0_relu_fwd_tb': (I, [P, P, P, P, P, I, I, I, P, I, L, L, L, L]),
    'mtl_conv0_wgrad_tb': (I, [P, P, P, P, P, P, I, I, I, I, L, L, L]),
    'mtl_colsum_accum_tb': (I, [P, P, L, I, P, P, P, I, L, L]),
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
    'mtl_conv3x3_wprep_h2_bytes': (L, [I, I]),
    'mtl_conv3x3_wprep_h2': (I, [P, P, P, P, I, I]),
    'mtl_conv3x3_wprep_h2_batch': (I, [P, I, P, P, P, I, I, P, P, P, I, I, P, P, P, I, I]),
    'mtl_conv3x3_wprep_h2_batch_tb': (I, [P, I, P, P, P, I, I, P, P, P, I, I, P, P, P, I, I, I, L, L, L, L]),
    'mtl_conv3x3_relu_fwd_h2': (I, [P, P, P, P, P, P, P, I, I, I, I, I]),
    'mtl_conv3x3_relu_pool_fwd_h2': (I, [P, P, P, P, P, P, P, P, I, I, I, I, I]),
    'mtl_conv3x3_dgrad_h2': (I, [P, P, P, P, P, P, P, P, I, I, I, I, I]),
    'mtl_conv3x3_relu_fwd_h2_tb': (I, [P, P, P, P, P, P, P, I, I, I, I, I, I, L, L, L, L, P, I]),
    'mtl_conv3x3_relu_pool_fwd_h2_tb': (I, [P, P, P, P, P, P, P, P, I, I, I, I, I, I, L, L, L, L, P, I]),
    'mtl_conv3x3_dgrad_h2_tb': (I, [P, P, P, P, P, P, P, P, I, I, I, I, I, I, L, L, L, P, I]),
    'mtl_conv3x3_wgrad_h2': (I, [P, P, P, P, P, P, P, P, P, L, I, I, I, I, I]),
    'mtl_conv3x3_wgrad_h2_tb': (I, [P, P, P, P, P, P, P, P, P, L, I, I, I, I, I, I, L, L, L, L]),
    'mtl_conv3x3_relu_fwd_x3_tb': (I, [P, P, P, P, P, I, I, I, I, I, I, L, L, P, I]),
    'mtl_conv3x3_relu_pool_fwd_x3_tb': (I, [P, P, P, P, P, P, I, I, I, I, I, I, L, L, P, I]),
    'mtl_conv3x3_dgrad_x3_tb': (I, [P, P, P, P, P, P, I, I, I, I, I, I, L, P, I]),
    'mtl_conv3x3_wgrad_x3_tb': (I, [P, P, P, P, P, P, P, L, I, I, I, I, I, I, L, L]),
    'mtl_absmax_f32': (I, [P, P, L, P]),
    'mtl_absmax_f32_tb': (I, [P, P, L, P, I, L, L]),
    'mtl_h2_census': (I, [P, P, L, P, P, I, L, L, L]),
    'mtl_zero_tails': (I, [P, P, I, I, I, P, I, I]),
    'mtl_gemm_h2_tb': (I, [P, I, I, I, I, P, I, P, L, P, I, P, L, P, I, P, P, I, I, L, L, L, L, P, L]),
    'mtl_gemm_h2_tn_tb': (I, [P, I, I, I, P, I, P, L, P, I, P, L, P, I, I, L, L, L]),
    'mtl_permute_hc': (I, [P, P, P, I, I, I, I, P]),
    'mtl_permute_hc_tb': (I, [P, P, P, I, I, I, I, P, I, L, L, L]),
    'mtl_layernorm_fwd': (I, [P, P, P, P, P, P, P, P, F, P, P, P, I, I, I, F]),
    'mtl_layernorm_fwd_g': (I, [P, P, P, P, P, P, P, P, F, P, P, P, I, I, I, F, I, L]),
    'mtl_layernorm_bwd_workspace': (L, [I, I]),
    'mtl_layernorm_bwd_g_waves': (I, [I]),
    'mtl_layernorm_bwd_g_workspace': (L, [I, I, I]),
    'mtl_layernorm_bwd_g': (I, [P, P, P, P, P, P, P, F, P, P, P, P, P, P, P, I, I, I, I, L, L]),
    'mtl_layernorm_bwd': (I, [P, P, P, P, P, P, P, F, P, P, P, P, P, P, P, I, I, I]),
    'mtl_ln_param_reduce_batch': (I, [P, P, I, I]),
    'mtl_softmax_mask_fwd': (I, [P, P, P, I, F, I, I, I, I, I, P, F, P]),
    'mtl_softmax_bwd': (I, [P, P, P, F, L, I, I, P, F]),
    'mtl_attn_supported': (I, [I, I]),
    'mtl_attn_fwd': (I, [P, P, P, P, I, I, I, P, I, F, I, I, I, I, I, I, P, I, F, P, I, P]),
    'mtl_attn_bwd': (I, [P, P, P, P, I, I, I, P, I, F, I, I, I, I, I, I, P, I, F, P, P, I, P, P, P, P, P, I, I, I]),
    'mtl_embed_pe_fwd': (I, [P, P, P, P, P, I, I, I, P, F]),
    'mtl_embed_bwd': (I, [P, P, P, P, P, P, I, I, L, P, F]),
    'mtl_embed_pe_fwd_g': (I, [P, P, P, P, P, I, I, I, P, F, I, L]),
    'mtl_embed_bwd_g': (I, [P, P, P, P, P, P, I, I, L, P, F, I, L]),
    'mtl_dropout_mask': (I, [P, P, L, F, P, ctypes.c_ulonglong]),
    'mtl_ce_argmax_fwd': (I, [P, P, P, I, I, I, L, F, I, P, P, P, P, P]),
    'mtl_ce_bwd': (I, [P, P, P, P, I, I, I, L, F, F, P, P, I]),
    'mtl_ce_argmax_fwd_g': (I, [P, P, P, I, I, I, L, F, P, P, P, P, P, I]),
    'mtl_ce_bwd_g': (I, [P, P, P, P, I, I, I, L, F, F, P, P, I, I]),
    'mtl_colsum_workspace': (L, [L, I]),
    'mtl_colsum_accum': (I, [P, P, L, I, L, P, P, P]),
    'mtl_sgd_theta_prime': (I, [P, P, P, F, P, L]),
    'mtl_sgd_theta_prime_tasks': (I, [P, P, P, F, P, L, I]),
    'mtl_sum_tasks': (I, [P, P, P, L, I, I]),
    'mtl_sum_tasks_strided': (I, [P, P, P, L, I, L, I]),
    'mtl_axpy': (I, [P, P, P, F, L]),
    'mtl_copy_f32': (I, [P, P, P, L]),
    'mtl_scale': (I, [P, P, F, P, L]),
    'mtl_adam_step': (I, [P, P, P, P, P, I, F, F, F, F, L]),
    'mtl_sumsq': (I, [P, P, L, P, P, I, F]),
    'mtl_spect_logmag': (I, [P, P, I, I, I, P, P, I]),
    'mtl_lstm_cell_fwd': (I, [P, P, P, P, P, P, P, P, P, F, I, I]),
    'mtl_lstm_cell_bwd': (I, [P, P, P, F, P, P, P, P, P, P, P, I, I]),
    'mtl_lstm_layer_supported': (I, [I, I]),
    'mtl_lstm_layer_workspace': (L, []),
    'mtl_lstm_layer_fwd': (I, [P, P, P, P, P, P, P, P, P, F, I, I, I, P]),
    'mtl_lstm_layer_bwd': (I, [P, P, P, F, P, P, P, P, I, I, I, P]),
    'mtl_lstm_stack_supported': (I, [I, I, I]),
    'mtl_lstm_stack_scratch': (L, [I, I, I, I]),
    'mtl_lstm_stack_fwd': (I, [P, P, F, I, I, I, I, P]),
    'mtl_lstm_stack_bwd': (I, [P, P, P, F, P, I, I, I, I, P]),
    'mtl_memset_zero': (I, [P, P, L]),
    'mtl_memcpy_d2d': (I, [P, P, P, L]),
    'mtl_event_record': (I, [P, P]),
    'mtl_stream_wait_event': (I, [P, P]),
    'mtl_cmdlist_opcode': (I, [ctypes.c_char_p]),
    'mtl_cmdlist_run': (I, [P, I, P]),
    'mtl_cmdlist_run_timed': (I, [P, I, P, P]),
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
    if ABI_VERSION is not None and h.mtl_abi_version() != ABI_VERSION:
        raise MtlLibraryError('libmtl_hip.so was built from another revision of include/mtl_hip.h (ABI hash %d, header %d): run '
                              '`python tools/gen_cmdlist.py` and rebuild' % (h.mtl_abi_version(), ABI_VERSION))
    _lib = h
    return h


# ---------------------------------------------------------------------------------------------------------------------
# command lists (include/mtl_hip.h "command lists"): record the C calls of one eager run, replay them with ONE ctypes call
# ---------------------------------------------------------------------------------------------------------------------
AMAX_SLOTS = 64 * 32    # MTL_AMAX_FLOATS of include/mtl_hip.h: floats per max|tensor| bound of the h2 kernels (64 slot heads, 128 B apart)


class LstmStack(ctypes.Structure):
    """mtl_lstm_stack of include/mtl_hip.h: a host struct of device pointers, MTL_LSTM_MAX_LAYERS entries per field"""
    _fields_ = [(_n, c_void_p * 4) for _n in ('w_ih', 'b_ih', 'w_hh', 'b_hh', 'gx', 'hall', 'call', 'acts', 'xout', 'dG', 'mask')]


class LnReduceDesc(ctypes.Structure):
    """mtl_ln_reduce_desc of include/mtl_hip.h (40 bytes)"""
    _fields_ = [('part', c_void_p), ('dgamma', c_void_p), ('dbeta', c_void_p), ('dsum', c_void_p), ('nw', c_int), ('d', c_int)]


class _CmdArg(ctypes.Union):
    _fields_ = [('p', c_void_p), ('l', c_long), ('d', ctypes.c_double)]


class MtlCmd(ctypes.Structure):
    _fields_ = [('op', c_int), ('nargs', c_int), ('a', _CmdArg * 48)]


def _kinds(name):
    return ['p' if t is P or t is ctypes.c_char_p else ('d' if t is F else 'l') for t in SIGNATURES[name][1]]


class CommandList:
    """An ordered record of library calls with their arguments.  `finish()` packs it into an mtl_cmd array; `run()` hands that
    array to mtl_cmdlist_run.  `repoint(old, new)` rewrites every POINTER argument equal to `old` (the input batch of a task
    is the only argument that moves between replays)."""

    def __init__(self):
        self.entries, self.buf, self.n = [], None, 0
        self._failed = c_int(-1)
        self._ptr_sites = {}
        self.breaks = []            # (index of the first command AFTER the break, tag): see add_break

    def add(self, op, kinds, args):
        self.entries.append((op, kinds, args))

    def add_break(self, tag):
        """A point where the replaying host must run code of its own between two library calls (run(on_break=...) calls
        on_break(tag) there): the overlapped all-reduce of a finished slice of the meta-gradient is a torch.distributed call."""
        self.breaks.append((len(self.entries), tag))

    def finish(self):
        self.n = len(self.entries)
        self.buf = (MtlCmd * max(self.n, 1))()
        for i, (op, kinds, args) in enumerate(self.entries):
            c = self.buf[i]
            c.op, c.nargs = op, len(args)
            for j, (k, v) in enumerate(zip(kinds, args)):
                if k == 'p':
                    v = int(v or 0)
                    c.a[j].p = v
                    if v:
                        self._ptr_sites.setdefault(v, []).append((i, j))
                elif k == 'd':
                    c.a[j].d = float(v)
                else:
                    c.a[j].l = int(v)
        self.entries = None
        return self

    def _names(self):
        names = {}
        for name in SIGNATURES:
            op = lib().mtl_cmdlist_opcode(name.encode())
            if op >= 0:
                names[op] = name
        return names

    def repoint(self, old, new):
        if old == new:
            return
        sites = self._ptr_sites.pop(old, None)
        if sites is None:
            raise KeyError('pointer %#x is not an argument of this command list' % old)
        for i, j in sites:
            self.buf[i].a[j].p = new
        self._ptr_sites.setdefault(new, []).extend(sites)

    def run(self, on_break=None):
        from . import _trace
        h, start = lib(), 0
        for end, tag in self.breaks + [(self.n, None)]:
            if end > start and _trace.ON:
                us = (c_float * (end - start))()
                rc = h.mtl_cmdlist_run_timed(ctypes.byref(self.buf, start * ctypes.sizeof(MtlCmd)), end - start, ctypes.byref(self._failed), us)
                names = self._names()
                _trace.calls([names[self.buf[i].op] for i in range(start, end)], list(us), start)
                if rc != 0:
                    raise RuntimeError('mtl_cmdlist_run: command %d failed with code %d' % (start + self._failed.value, rc))
            elif end > start:
                rc = h.mtl_cmdlist_run(ctypes.byref(self.buf, start * ctypes.sizeof(MtlCmd)), end - start, ctypes.byref(self._failed))
                if rc != 0:
                    raise RuntimeError('mtl_cmdlist_run: command %d failed with code %d' % (start + self._failed.value, rc))
            start = end
            if tag is not None and on_break is not None:
                on_break(tag)


class Recorder:
    """Stands in for the ctypes library handle while a CommandList is being recorded: every recordable call is executed AND
    logged; size queries (`*_workspace`, ...) are only executed (their results become arguments of later calls)."""

    def __init__(self, handle, cmdlist):
        self._h, self._cl, self._cache = handle, cmdlist, {}

    def segment_break(self, tag):
        self._cl.add_break(tag)

    def __getattr__(self, name):
        fn = self._cache.get(name)
        if fn is None:
            real = getattr(self._h, name)
            op = self._h.mtl_cmdlist_opcode(name.encode()) if name in SIGNATURES else -1
            if op < 0:
                fn = real
            else:
                kinds, cl = _kinds(name), self._cl

                def fn(*args, _real=real, _op=op, _kinds=kinds):
                    rc = _real(*args)
                    if rc == 0:
                        cl.add(_op, _kinds, args)
                    return rc
            self._cache[name] = fn
        return fn


def check(rc, what):
    if rc != 0:
        raise RuntimeError('%s failed with code %d' % (what, rc))


def levenshtein(a, b):
    """Edit distance between two python strings (host helper, code-point level)."""
    # code points as little-endian uint32 straight from the codec (a per-character ord() list was 6 ms of host time per meta-step)
    ua = a.encode('utf-32-le', 'surrogatepass') or b'\0\0\0\0'
    ub = b.encode('utf-32-le', 'surrogatepass') or b'\0\0\0\0'
    d = lib().mtl_levenshtein_u32(ctypes.cast(ctypes.c_char_p(ua), c_void_p), len(a), ctypes.cast(ctypes.c_char_p(ub), c_void_p), len(b))
    if d < 0:
        raise RuntimeError('mtl_levenshtein_u32 failed: %d' % d)
    return d
