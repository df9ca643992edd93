"""Pass executor: one forward + hand-written backward of the VGG-CNN + Transformer speech recogniser,
issued as calls into libmtl_hip.so on the current HIP stream.

PyTorch is used for device memory (a buffer arena of caller-owned tensors) and the stream handle only; every
arithmetic op of the pass is a kernel from include/mtl_hip.h.  Parameters live in ONE flat fp32 buffer
(`theta`, layout from `ParamLayout`); a pass reads parameters from any buffer with that layout (theta0 or the
inner-loop theta') and ACCUMULATES gradients into a second flat buffer (`.backward()` semantics, which the
reference's meta-gradient definition relies on -- SURVEY.md Q1).

What a pass computes follows the reference line by line:
  models/asr/transformer.py:120-149 (forward), modules/encoder.py:53-106, modules/decoder.py:71-115,293-323,
  modules/common_layers.py:110-132,238-331, utils/metrics.py:96-126.
"""
import math

import os

import numpy as np
import torch

import re

from . import _lib, _trace
from ._lib import check

PAD_ID, SOS_ID, EOS_ID = 0, 1, 2
RELU, ACCUM = 1, 2


class ParamLayout:
    """name -> (offset, shape) inside the flat fp32 parameter buffer; offsets are 16-byte aligned."""

    def __init__(self, named_shapes):
        self.entries = {}
        self.order = []
        off = 0
        for name, shape in named_shapes:
            n = 1
            for s in shape:
                n *= int(s)
            self.entries[name] = (off, tuple(int(s) for s in shape), n)
            self.order.append(name)
            off += (n + 3) // 4 * 4
        self.total = off

    def off(self, name):
        return self.entries[name][0]

    def view(self, flat, name):
        off, shape, n = self.entries[name]
        return flat[off:off + n].view(shape)

    def group_bounds(self):
        """{'encoder' | 'decoder' | 'conv': (first, end) offsets in floats} of the three parameter groups in the flat buffers --
        parameters() order of models/asr/transformer.py is encoder, decoder, conv, so each group is one contiguous slice."""
        out, prev = {}, None
        for name in self.order:
            grp = name.split('.')[0]
            if grp != prev:
                if grp in out:
                    raise RuntimeError('parameter group %s is not contiguous in the flat layout' % grp)
                out[grp] = [self.off(name), self.total]
                if prev is not None:
                    out[prev][1] = self.off(name)
                prev = grp
        return {k: tuple(v) for k, v in out.items()}


class Hyper:
    def __init__(self, **kw):
        self.__dict__.update(kw)


def decoder_io(padded_target, pad_id=PAD_ID, sos_id=SOS_ID, eos_id=EOS_ID, width=None):
    """modules/decoder.py:55-69 on the host: seq_in = [SOS, y..] padded with EOS; seq_out = [y.., EOS] padded with PAD.
    width: at least this many decoder positions (a slice of a batch keeps the whole batch's width, so that the labels
    predicted at padded positions -- which the reference's CER strings include -- are the same)."""
    tgt = padded_target.detach().to('cpu', torch.int64).numpy()
    B, L = tgt.shape
    nonpad = tgt != pad_id
    lens = nonpad.sum(1)
    width = max(int(lens.max()) + 1, int(width or 0))
    # every row's non-pad labels moved to the front in their order (the reference drops pads wherever they stand): a stable
    # argsort of the pad flags; numpy on these few hundred integers costs a tenth of the per-row tensor indexing it replaces
    comp = np.take_along_axis(tgt, np.argsort(~nonpad, axis=1, kind='stable'), 1)
    m = min(L, width - 1)
    own = np.arange(m)[None, :] < lens[:, None]
    seq_in = np.full((B, width), eos_id, dtype=np.int64)
    seq_out = np.full((B, width), pad_id, dtype=np.int64)
    seq_in[:, 0] = sos_id
    seq_in[:, 1:m + 1] = np.where(own, comp[:, :m], eos_id)
    seq_out[:, :m] = np.where(own, comp[:, :m], pad_id)
    seq_out[np.arange(B), lens] = eos_id
    return torch.from_numpy(seq_in), torch.from_numpy(seq_out)


_FULL = {'q': 'query', 'k': 'key', 'v': 'value'}


class _DecodeSession:
    """See PassEngine.decode_session."""

    def __init__(self, eng, theta, mem, B, T4, S, shared_memory):
        self.eng, self.B, self.T4, self.S = eng, B, T4, S
        hp = eng.hp
        d, r, h, dk, dv, V = hp.d, hp.r, hp.h, hp.dk, hp.dv, hp.V
        hk, hv = h * dk, h * dv
        if S > eng.pe_dec.shape[0] + 1:
            raise ValueError('tgt_max_len too small for %d decoding positions' % S)
        self.P = theta.data_ptr()
        buf = eng.buf
        self.x = buf('g.x', (B, d))
        # fast path (head sizes of the fused attention kernel, h d_k = h d_v): q / k / v of a layer's self-attention live in ONE (3, B, S, h d_k)
        # block -- row t of [0] is the current query, [1] / [2] are the K / V caches -- so that the three low-rank projections are two
        # strided-batch launches writing row t, and the attention over the cache is ONE flash-attention launch (keys beyond the current
        # position masked by a per-step length: row t of `klen_self`) instead of product + softmax + product
        self.fast = bool(eng.fused_attn and hk == hv and hp.n_dec > 0 and
                         eng._qkv_groups('decoder.layers.0.self_attn.', 1, 1, 1, 1, hk, hv)[0][0] == 'qkv')
        if self.fast:
            self.qkv = [buf('g.qkv%d' % i, (3, B, S, hk)) for i in range(hp.n_dec)]
            for c in self.qkv:
                eng.zero_(c)                                        # (masked keys are multiplied by probability 0: they must be finite)
            self.kc, self.vc = [c[1] for c in self.qkv], [c[2] for c in self.qkv]
            self.klen_self = buf('g.klen', (S, B), torch.int32)
            self.klen_self.copy_(torch.arange(1, S + 1, dtype=torch.int32).view(S, 1).expand(S, B))
            self.klen_cross = buf('g.klenx', (B,), torch.int32)
            self.klen_cross.fill_(T4)
            self.lse = buf('g.lse', (B, h, 1))
            # the weights do not move during a decode: every rank-r pair  y = (x W_a^T) W_b^T  is applied as ONE product with
            # W_b W_a (h d_k x d: 1 MB instead of 0.4 MB streamed per step, but one dependent launch instead of two -- a step is a
            # chain of launches, not a bandwidth problem); merged once per session from the current theta (6 small products per layer)
            self.Wqkv = [buf('g.wqkv%d' % i, (3, hk, d)) for i in range(hp.n_dec)]
            self.Wso = [buf('g.wso%d' % i, (d, hv)) for i in range(hp.n_dec)]
            self.Wcq = [buf('g.wcq%d' % i, (hk, d)) for i in range(hp.n_dec)]
            self.Wco = [buf('g.wco%d' % i, (d, hv)) for i in range(hp.n_dec)]
            for i in range(hp.n_dec):
                pre = 'decoder.layers.%d.' % i
                o = lambda n, pre=pre: self.P + 4 * eng.L.off(pre + n)
                sA_, sB_ = (eng._pstride(pre + 'self_attn.', 'qkv', sfx) for sfx in ('_linear_a.weight', '_linear_b.weight'))
                eng.gemm(0, 0, hk, d, r, o('self_attn.query_linear_b.weight'), r, o('self_attn.query_linear_a.weight'), d, self.Wqkv[i].data_ptr(), d,
                         batch=3, sA=(sB_, 0), sB=(sA_, 0), sC=(hk * d, 0))
                for att, dst in (('self_attn.', self.Wso[i]), ('encoder_attn.', self.Wco[i])):
                    eng.gemm(0, 0, d, hv, r, o(att + 'output_linear_b.weight'), r, o(att + 'output_linear_a.weight'), hv, dst.data_ptr(), hv)
                eng.gemm(0, 0, hk, d, r, o('encoder_attn.query_linear_b.weight'), r, o('encoder_attn.query_linear_a.weight'), d,
                         self.Wcq[i].data_ptr(), d)
        else:
            self.kc = [buf('g.kc%d' % i, (B, S, hk)) for i in range(hp.n_dec)]
            self.vc = [buf('g.vc%d' % i, (B, S, hv)) for i in range(hp.n_dec)]
        Bm = 1 if shared_memory else B
        self.cross_stride = 0 if shared_memory else T4
        self.ck = [buf('g.ck%d' % i, (Bm * T4, hk)) for i in range(hp.n_dec)]
        self.cv = [buf('g.cv%d' % i, (Bm * T4, hv)) for i in range(hp.n_dec)]
        self.ta, self.tq = buf('g.ta', (max(Bm * T4, B), r)), buf('g.q', (B, hk))
        self.to, self.tob = buf('g.o', (B, hv)), buf('g.ob', (B, d))
        self.y1, self.y2, self.y3 = buf('g.y1', (B, d)), buf('g.y2', (B, d)), buf('g.y3', (B, d))
        self.h1, self.h2 = buf('g.h1', (B, hp.inner)), buf('g.h2', (B, d))
        self.xhat, self.rstd = buf('g.xhat', (B, d)), buf('g.rstd', (B,))
        self.ldS = (max(S, T4) + 3) // 4 * 4
        self.Sc = buf('g.S', (B, h, 1, self.ldS))
        self.logits = buf('g.logits', (B, V))
        self.junk = buf('g.junk', (3 * B + 4,))
        self.zero_gold = buf('g.zero', (B,), torch.int64)
        self.zero_gold.zero_()
        self.cur = [buf('g.cur0', (B, d)), buf('g.cur1', (B, d))]
        self.last = None
        for i in range(hp.n_dec):                                   # cross-attention keys / values of the encoder output, once
            pre = 'decoder.layers.%d.encoder_attn.' % i
            self._lowrank(pre, 'key', mem, Bm * T4, self.ck[i].data_ptr())
            self._lowrank(pre, 'value', mem, Bm * T4, self.cv[i].data_ptr(), width=hv)

    def _lowrank(self, pre, name, src, rows, dst, ldc=None, width=None):
        eng, hp = self.eng, self.eng.hp
        hk, hv = hp.h * hp.dk, hp.h * hp.dv
        width = hk if width is None else width
        o = lambda n: self.P + 4 * eng.L.off(pre + n)
        eng.linear_fwd(src, rows, hp.d if name != 'output' else hv, o(name + '_linear_a.weight'), None, self.ta.data_ptr(), hp.r)
        eng.gemm(0, 1, rows, width, hp.r, self.ta.data_ptr(), hp.r, o(name + '_linear_b.weight'), hp.r, dst, ldc or width,
                 bias=o(name + '_linear_b.bias'))

    def _attend(self, q, kbuf, vbuf, kv_rows, kv_stride, out):
        # one query row per (b, h): scores over kv_rows keys, softmax, weighted sum of the values
        eng, hp, B = self.eng, self.eng.hp, self.B
        h, dk, dv = hp.h, hp.dk, hp.dv
        hk, hv, ldS = h * dk, h * dv, self.ldS
        eng.gemm(0, 1, 1, kv_rows, dk, q, hk, kbuf, hk, self.Sc.data_ptr(), ldS, batch=B * h, H=h, sA=(hk, dk), sB=(kv_stride, dk),
                 sC=(h * ldS, ldS))
        check(eng.lib.mtl_softmax_mask_fwd(eng.stream, self.Sc.data_ptr(), None, 0, 1.0 / float(hp.temperature), B, h, 1, kv_rows,
                                           ldS, None, 1.0, None), 'softmax')
        eng.gemm(0, 0, 1, dv, kv_rows, self.Sc.data_ptr(), ldS, vbuf, hv, out, hv, batch=B * h, H=h, sA=(h * ldS, ldS),
                 sB=(kv_stride, dv), sC=(hv, dv))

    def step(self, t, tok_ptr):
        """Feed the B tokens at device address tok_ptr as position t; leaves the logits of that position in self.logits."""
        eng, hp, B, S, T4 = self.eng, self.eng.hp, self.B, self.S, self.T4
        d, V = hp.d, hp.V
        hk, hv = hp.h * hp.dk, hp.h * hp.dv
        L, lib, P = eng.L, eng.lib, self.P
        check(lib.mtl_embed_pe_fwd(eng.stream, tok_ptr, P + 4 * L.off('decoder.trg_embedding.weight'),
                                   eng.pe_dec.data_ptr() + 4 * t * d, self.x.data_ptr(), B, 1, d, None, 1.0), 'embed')
        cur = self.x
        for i in range(hp.n_dec):
            pre = 'decoder.layers.%d.' % i
            o = lambda n: P + 4 * L.off(pre + n)
            sa = pre + 'self_attn.'
            if self.fast:
                sb_ = eng._pstride(sa, 'qkv', '_linear_b.bias')
                qkv = self.qkv[i].data_ptr()
                eng.gemm(0, 1, B, hk, d, cur.data_ptr(), d, self.Wqkv[i].data_ptr(), d, qkv + 4 * t * hk, S * hk,
                         bias=o('self_attn.query_linear_b.bias'), batch=3, sB=(hk * d, 0), sC=(B * S * hk, 0), sbias=sb_)
                check(lib.mtl_attn_fwd(eng.stream, qkv + 4 * t * hk, self.kc[i].data_ptr(), self.vc[i].data_ptr(), S * hk, hk, hv,
                                       self.klen_self.data_ptr() + 4 * t * B, 0, 1.0 / float(hp.temperature), B, hp.h, 1, S, hp.dk, hp.dv, None, 0,
                                       1.0, self.to.data_ptr(), hv, self.lse.data_ptr()), 'mtl_attn_fwd')
            else:
                self._lowrank(sa, 'query', cur.data_ptr(), B, self.tq.data_ptr())
                self._lowrank(sa, 'key', cur.data_ptr(), B, self.kc[i].data_ptr() + 4 * t * hk, ldc=S * hk)      # row t of the cache
                self._lowrank(sa, 'value', cur.data_ptr(), B, self.vc[i].data_ptr() + 4 * t * hv, ldc=S * hv, width=hv)
                self._attend(self.tq.data_ptr(), self.kc[i].data_ptr(), self.vc[i].data_ptr(), t + 1, S * hk, self.to.data_ptr())
            if self.fast:
                eng.gemm(0, 1, B, d, hv, self.to.data_ptr(), hv, self.Wso[i].data_ptr(), hv, self.tob.data_ptr(), d,
                         bias=o('self_attn.output_linear_b.bias'))
            else:
                self._lowrank(sa, 'output', self.to.data_ptr(), B, self.tob.data_ptr(), width=d)
            eng.ln_fwd(self.tob.data_ptr(), cur.data_ptr(), o('self_attn.layer_norm.weight'), o('self_attn.layer_norm.bias'), None, None,
                       self.y1.data_ptr(), self.xhat.data_ptr(), self.rstd.data_ptr(), B, 1)
            ca = pre + 'encoder_attn.'
            if self.fast:
                eng.gemm(0, 1, B, hk, d, self.y1.data_ptr(), d, self.Wcq[i].data_ptr(), d, self.tq.data_ptr(), hk,
                         bias=o('encoder_attn.query_linear_b.bias'))
            else:
                self._lowrank(ca, 'query', self.y1.data_ptr(), B, self.tq.data_ptr())
            if self.fast and self.cross_stride:      # (every row has its own memory: greedy search; beam search shares one with stride 0)
                check(lib.mtl_attn_fwd(eng.stream, self.tq.data_ptr(), self.ck[i].data_ptr(), self.cv[i].data_ptr(), hk, hk, hv,
                                       self.klen_cross.data_ptr(), 0, 1.0 / float(hp.temperature), B, hp.h, 1, T4, hp.dk, hp.dv, None, 0, 1.0,
                                       self.to.data_ptr(), hv, self.lse.data_ptr()), 'mtl_attn_fwd')
            else:
                self._attend(self.tq.data_ptr(), self.ck[i].data_ptr(), self.cv[i].data_ptr(), T4, self.cross_stride * hk, self.to.data_ptr())
            if self.fast:
                eng.gemm(0, 1, B, d, hv, self.to.data_ptr(), hv, self.Wco[i].data_ptr(), hv, self.tob.data_ptr(), d,
                         bias=o('encoder_attn.output_linear_b.bias'))
            else:
                self._lowrank(ca, 'output', self.to.data_ptr(), B, self.tob.data_ptr(), width=d)
            eng.ln_fwd(self.tob.data_ptr(), self.y1.data_ptr(), o('encoder_attn.layer_norm.weight'), o('encoder_attn.layer_norm.bias'), None,
                       None, self.y2.data_ptr(), self.xhat.data_ptr(), self.rstd.data_ptr(), B, 1)
            eng.linear_fwd(self.y2.data_ptr(), B, d, o('pos_ffn.linear_1.weight'), o('pos_ffn.linear_1.bias'), self.h1.data_ptr(), hp.inner,
                           relu=True)
            eng.linear_fwd(self.h1.data_ptr(), B, hp.inner, o('pos_ffn.linear_2.weight'), o('pos_ffn.linear_2.bias'), self.h2.data_ptr(), d)
            nxt = self.cur[i & 1]                  # (the layer's output lands where the next layer reads it: no copy)
            eng.ln_fwd(self.h2.data_ptr(), self.y2.data_ptr(), o('pos_ffn.layer_norm.weight'), o('pos_ffn.layer_norm.bias'), None, None,
                       nxt.data_ptr(), self.xhat.data_ptr(), self.rstd.data_ptr(), B, 1)
            cur = nxt
        eng.gemm(0, 1, B, V, d, cur.data_ptr(), d, P + 4 * L.off('decoder.output_linear.weight'), d, self.logits.data_ptr(), V)
        return self.logits

    def argmax_into(self, dst_ptr):
        """arg-max of the current logits (lowest index on ties) written as B int64 at dst_ptr; log-sum-exp lands in junk[0:B]."""
        eng, B, V = self.eng, self.B, self.eng.hp.V
        check(eng.lib.mtl_ce_argmax_fwd(eng.stream, self.logits.data_ptr(), self.zero_gold.data_ptr(), B, V, V, PAD_ID, 0.0, 1, None,
                                        self.junk.data_ptr(), dst_ptr, self.junk.data_ptr() + 4 * B, self.junk.data_ptr() + 8 * B),
              'argmax')

    def logits_and_lse(self):
        """(B,V) logits and (B,) log-sum-exp of the current position on the host (one synchronising copy each)."""
        hyp = self.eng.buf('g.hyp', (self.B,), torch.int64)
        self.argmax_into(hyp.data_ptr())
        return self.logits.cpu(), self.junk[:self.B].cpu()

    def reorder(self, rows, t):
        """caches[r, :t] <- caches[rows[r], :t] for every layer (beam search: row r continues hypothesis rows[r])."""
        if list(rows) == list(range(self.B)):
            return
        idx = torch.tensor(rows, dtype=torch.int64, device=self.logits.device)
        for c in self.kc + self.vc:
            c[:, :t] = c.index_select(0, idx)[:, :t]


CENSUS_SLOTS = 9    # bound slots of the h2 operands: 0 y1, 1 p1, 2 y5, 3 dp2, 4 dy5, 5 dp1, 6 p2, (7: weights, not counted) 8 de0
CENSUS_NAMES = {0: 'conv0 out', 1: 'pool1', 2: 'conv5 out', 3: 'd pool2', 4: 'd conv5 out', 5: 'd pool1', 6: 'pool2', 8: 'd input-linear out'}
STAGE_RING = 6      # pinned staging buffers per slot (prepare_tasks): > pipeline depth (default 2) + 2

_LAYER_BUF = re.compile(r'^([de])(\d+)\.(.+)$')


def round_width(T, quantum):
    """Frame count rounded up to a repeating width: a multiple of `quantum`, coarser for long batches (at most 1/16 of the width, so the
    padding stays below ~6 % while a 5000-frame workload -- 55 GB of buffers per width -- sees a handful of widths, not dozens)."""
    q = max(int(quantum), 1)
    q = max(q, (T // 16) // q * q)
    return -(-T // q) * q


_POOL_ACCOUNTS = {}


def _pool_account(device):
    """Bytes held by the buffer pools of every engine on one device, and the budget they share (MTL_POOL_GB)."""
    key = (device.type, device.index)
    acc = _POOL_ACCOUNTS.get(key)
    if acc is None:
        gb = os.environ.get('MTL_POOL_GB')
        if gb is not None:
            budget = int(float(gb) * (1 << 30))
        elif device.type == 'cuda':
            budget = max(torch.cuda.get_device_properties(device).total_memory // 4, 4 << 30)
        else:
            budget = 4 << 30
        acc = _POOL_ACCOUNTS[key] = dict(bytes=0, budget=budget, gen=0, engines=[])
    return acc


class PassEngine:
    def __init__(self, layout, hp, device, pe_enc, pe_dec):
        self.pe_enc, self.pe_dec = pe_enc, pe_dec   # (max_len, d) fp32 device tables (non-trainable buffers)
        self.L = layout
        self.hp = hp
        self.device = device
        self.lib = _lib.lib()
        self.arena = {}     # name -> buffer of the current pass (looked up again by the backward)
        self.pool = {}      # (name, shape, dtype) -> allocation, so alternating batch shapes do not re-allocate
        # Real manifests bring a new (T, Td) with almost every batch (collate pads to the batch maximum; padding further would change
        # results: the reference masks with RAW frame counts).  The pool is therefore bounded: entries carry the number
        # of the last pass that touched them, and trim_pool() -- start of every forward -- drops the least recently used ones that
        # neither this pass nor the previous one uses while the pools of ALL engines on this device (a model has one per task lane)
        # together hold more than MTL_POOL_GB (default: a quarter of the device's memory, 72 GiB of the 288; a budget per engine let
        # eight lanes keep 8 x 48 GiB of stale shapes and ran a north-star run on ragged batches out of memory).  An eviction bumps
        # scratch_epoch, which makes the trainer re-record its command lists (they hold raw addresses).
        self.account = _pool_account(device)
        import weakref
        self.account['engines'] = [w for w in self.account['engines'] if w() is not None] + [weakref.ref(self)]
        self._last_pass = self.account['gen']       # the device-wide pass count at this engine's latest pass (trim_pool)
        # stand-alone passes on batches whose width changes from call to call are widened to repeating widths ('auto': from the second
        # width on; '0' never -- Transformer.evaluate; '1' always), to a multiple of MTL_RAGGED_QUANTUM frames
        self.widen, self.widen_quantum = 'auto', int(os.environ.get('MTL_RAGGED_QUANTUM', '64'))
        self._first_width, self._widths_vary = {}, False
        self.pool_budget = self.account['budget']           # (an engine may be given a tighter one of its own: tests)
        self._pool_gen, self._pool_bytes, self._gen = {}, 0, 0
        self.saved = None
        self.gemm_ws = torch.empty(8 << 20, dtype=torch.float32, device=device) if device.type == 'cuda' else None  # split-K slabs
        # weight/bias-gradient kernels are off the critical path (nothing downstream in the backward reads them): they run
        # on a second HIP stream with their own workspaces, forked once per block and joined at the end of the backward
        self.side = torch.cuda.Stream(device) if device.type == 'cuda' else None
        self.gemm_ws_side = torch.empty(8 << 20, dtype=torch.float32, device=device) if device.type == 'cuda' else None
        self.scratch_side = torch.empty(4 << 20, dtype=torch.float32, device=device) if device.type == 'cuda' else None
        self.on_side = False
        self.scratch_epoch = 0
        self._stage, self._stage_turn = {}, {}   # pinned host staging of prepare()
        self._events, self._ev_next = [], 0    # fork / join events of the side stream (raw handles: recordable library calls)
        self.dropout_p = 0.0          # set by the model: hp.dropout when model.training else 0
        self._site = 0                # dropout site counter of the current pass (Philox offset = site << 40)
        self.forward_hook = None      # optional callable(engine) after every forward has been enqueued (tests capture the arena)
        # optional callable(tag), tag in ('decoder', 'encoder', 'conv'): called by backward() -- at ENQUEUE time, eager run or command-list
        # replay alike -- right after the last kernel that writes that parameter group's gradients has been enqueued (on the main or
        # the side stream): the trainer starts the group's share of the meta-gradient all-reduce there (slice_bounds())
        self.slice_hook = None
        # h2 guard: a (tasks, CENSUS_SLOTS, 4) int64 device tensor while the trainer samples the census of the h2 operands against their
        # bounds (mtl_h2_census: non-zero elements / fewer than 22 bits / fewer than 16 bits; TransientTrainer.h2_check_every), else None
        self.census = None
        self.deferred = []
        self._wlog = {}
        # the weight gradients of all layers of a stack run as one strided-batch launch per parameter kind (flush_layer_wgrads); the
        # decoder's prologue (embedding + layer 0's self-attention block) depends on the labels and theta only: it runs on the side
        # stream under the encoder, and its backward (which feeds parameter gradients only) under the encoder's backward
        self.use_side_stream = True
        # 3x3 convolutions (forward, data gradient, weight gradient), MTL_CONV: 'h2' (default) two fp16 pieces per fp32 operand
        # (3 MFMAs per step, per-tensor power-of-two scaling from device scalars the producers deliver; csrc/mtl_h2.h), 'x3' three
        # exact bf16 pieces (6 MFMAs), 'f32' the fp32-MFMA engine
        self.conv_mode = os.environ.get('MTL_CONV', 'h2')
        if self.conv_mode not in ('h2', 'x3', 'f32'):
            raise ValueError('MTL_CONV must be h2, x3 or f32')
        self.conv_x3 = self.conv_mode != 'f32'
        self.conv_h2 = self.conv_mode == 'h2'
        self._ln_pending, self._ln_tables = [], {}
        # the encoder's input Linear (5120 -> 512) with its data / weight gradient: 'h2' with the h2 convolutions (two fp16 pieces, the
        # bounds ride along), else the product engines' own routing (x3 / fp32); bench.py's exact-fp32 leg sets 'f32'
        self.in_linear = 'h2'
        # scaled-dot-product attention as ONE flash-style kernel forward and two backward (no score tensor in HBM); head sizes
        # outside mtl_attn_supported() take the batched-GEMM + softmax path
        self.fused_attn = device.type == 'cuda' and bool(self.lib.mtl_attn_supported(hp.dk, hp.dv))
        # task batching: a pass may carry the batches of `nt` tasks of a meta-step (rows of task t follow those of task t - 1); task t
        # reads its parameters at theta + t * sP floats (0: all tasks share theta0 -- the training passes) and accumulates its
        # gradients at grad + t * sG (see forward_device / backward)
        self.nt, self.sP, self.sG = 1, 0, 0
        self.prof = None    # set (to anything) while a profiling proxy stands in for self.lib: replay / graphs / lane tricks are bypassed
        if device.type != 'cuda':
            raise RuntimeError('PassEngine needs an MI355X device (got %s); there is no CPU product path' % device)

    # ---------------------------------------------------------------- plumbing
    def _took(self, nbytes):
        self._pool_bytes += nbytes
        self.account['bytes'] += nbytes

    def __del__(self):
        try:
            self.account['bytes'] -= self._pool_bytes     # the device-wide account outlives the engine
        except Exception:
            pass

    def buf(self, name, shape, dtype=torch.float32):
        """Named buffer of the current pass.  Per-layer buffers ('d<i>.<what>', 'e<i>.<what>') are slices of ONE allocation per
        <what> with the layers at a constant stride -- 'dec_in.y' / 'enc_in.y' are slot 0 of the '<x>.ff.y' group, so that every
        layer's INPUT is at that stride too -- which lets the weight-gradient products of all layers of a stack run as one
        strided-batch launch (flush_layer_wgrads)."""
        shape = tuple(int(v) for v in shape)
        m = _LAYER_BUF.match(name)
        if m or name in ('dec_in.y', 'enc_in.y'):
            kind, idx, rest = (m.group(1), int(m.group(2)), m.group(3)) if m else (name[0], -1, 'ff.y')
            n = (self.hp.n_dec if kind == 'd' else self.hp.n_enc) + (1 if rest == 'ff.y' else 0)
            slot = idx + 1 if rest == 'ff.y' else idx
            if 0 <= slot < n:
                key = (kind + '*.' + rest, (n,) + shape, dtype)
                grp = self.pool.get(key)
                if grp is None:
                    grp = torch.empty(key[1], dtype=dtype, device=self.device)
                    self.pool[key] = grp
                    self._took(grp.numel() * grp.element_size())
                self._pool_gen[key] = self._gen
                t = grp[slot]
                self.arena[name] = t
                return t
        key = (name, shape, dtype)
        t = self.pool.get(key)
        if t is None:
            t = torch.empty(key[1], dtype=dtype, device=self.device)
            self.pool[key] = t
            self._took(t.numel() * t.element_size())
        self._pool_gen[key] = self._gen
        self.arena[name] = t
        return t

    def _evict(self, evictable):
        """drop the least recently used pool entries `evictable(key)` accepts while the pool / the device-wide account is over its
        budget; -> bytes freed.  An eviction invalidates everything that holds addresses (tables, arena, recorded command lists)."""
        over = lambda: self._pool_bytes > self.pool_budget or self.account['bytes'] > self.account['budget']
        freed = 0
        for key in sorted(self.pool, key=lambda k: self._pool_gen.get(k, 0)):
            if not over() or not evictable(key):
                break
            t = self.pool.pop(key)
            self._pool_gen.pop(key, None)
            nb = t.numel() * t.element_size()
            self._took(-nb)
            freed += nb
        if freed:
            # device tables keyed by buffer addresses (LayerNorm reductions) and the pinned staging of shapes that are gone: rebuilt on
            # demand
            self._ln_tables.clear()
            self.arena = {k: v for k, v in self.arena.items() if not isinstance(v, torch.Tensor) or k == '_scratch'}
            self.scratch_epoch += 1
        return freed

    def trim_pool(self):
        """Start of a pass: count it, and while the pool is over its budget drop least-recently-used buffers that the last two
        passes did not touch (a forward and its backward, and the pass a pipelined host has already enqueued, keep theirs).
        The budget is shared by all engines of the device (a model has one per task lane), and an engine can only give up its OWN
        buffers during its own pass: when the shared account is over, the engines that have been IDLE for a while (no pass among the
        device's last max(8, 4 x engines) passes: lanes a changed schedule no longer uses) are emptied first -- otherwise the one
        engine doing the work would stay over budget for ever, dropping and re-allocating its own shapes (and re-recording the
        trainer's command lists) on every pass.
        Frees go back to torch's caching allocator in stream order: the side stream was joined at the end of the backward, and a
        block returns to the free list of the stream it was allocated on."""
        self._gen += 1
        acc = self.account
        acc['gen'] += 1
        self._last_pass = acc['gen']
        if len(self._ln_tables) > 512:
            # descriptor tables are keyed by addresses AND extents: ragged batches bring new ones with every pass.  Recorded command
            # lists hold the tables' addresses, hence the epoch.
            self._ln_tables.clear()
            self.scratch_epoch += 1
        over = lambda: self._pool_bytes > self.pool_budget or acc['bytes'] > acc['budget']
        if not over():
            return 0
        if torch.cuda.is_current_stream_capturing():
            return 0                      # a free inside a hipGraph capture would be baked into the graph: trim at the next eager pass
        freed = 0
        if acc['bytes'] > acc['budget']:
            others = [e for e in (w() for w in acc['engines']) if e is not None and e is not self]
            idle_after = max(8, 4 * (len(others) + 1))
            for e in sorted(others, key=lambda e_: e_._last_pass):
                if acc['bytes'] <= acc['budget'] or acc['gen'] - e._last_pass < idle_after:
                    break
                freed += e._evict(lambda key: True)
        if over():
            freed += self._evict(lambda key: self._pool_gen.get(key, 0) < self._gen - 2)
        return freed

    def scratch(self, nbytes):
        if self.on_side:
            if nbytes > self.scratch_side.numel() * 4:
                raise RuntimeError('side-stream scratch too small for %d bytes' % nbytes)
            return self.scratch_side.data_ptr()
        t = self.arena.get('_scratch')
        if t is None or t.numel() * 4 < nbytes:
            t = torch.empty((int(nbytes) + 3) // 4 + 1024, dtype=torch.float32, device=self.device)
            self.arena['_scratch'] = t
            self.scratch_epoch += 1          # recorded command lists hold the old address: they are re-recorded (trainer._run_recorded)
        return t.data_ptr()

    @property
    def stream(self):
        return torch.cuda.current_stream(self.device).cuda_stream

    def gemm(self, ta, tb, M, N, K, A, lda, B, ldb, C, ldc, bias=None, gate=None, ldg=0, flags=0, alpha=1.0,
             batch=1, H=1, sA=(0, 0), sB=(0, 0), sC=(0, 0), sbias=0, kbatch=1, sAk=0, sBk=0, rowsum=None, srow=0, sbias_h=0, srow_h=0,
             task=None):
        """kbatch / sAk / sBk: sum over several (A, B) pairs inside one launch; rowsum: += row sums of op(A) (bias gradient);
        sbias / srow stride the OUTER batch index z // H, sbias_h / srow_h the inner one z % H.
        task = (sAt, sBt, sCt, sbias_t, srow_t): strides of the task index of a task-batched pass (the product is issued once with
        `batch` items PER TASK; mandatory when the pass carries several tasks)."""
        wst = self.gemm_ws_side if self.on_side else self.gemm_ws
        if self.nt == 1:
            check(self.lib.mtl_gemm_f32_ex(self.stream, ta, tb, M, N, K, alpha, A, lda, B, ldb, C, ldc, bias, gate, ldg, flags,
                                           batch, H, sA[0], sA[1], sB[0], sB[1], sC[0], sC[1], sbias, kbatch, sAk, sBk, rowsum, srow,
                                           wst.data_ptr(), wst.numel() * 4, sbias_h, srow_h), 'mtl_gemm_f32_ex')
            return
        if task is None:
            raise RuntimeError('task strides missing for a product of a task-batched pass')
        check(self.lib.mtl_gemm_f32_tb(self.stream, ta, tb, M, N, K, alpha, A, lda, B, ldb, C, ldc, bias, gate, ldg, flags,
                                       batch * self.nt, H, sA[0], sA[1], sB[0], sB[1], sC[0], sC[1], sbias, kbatch, sAk, sBk, rowsum,
                                       srow, wst.data_ptr(), wst.numel() * 4, sbias_h, srow_h, self.nt, *task), 'mtl_gemm_f32_tb')

    # ---- byte-level helpers and flat-vector updates as LIBRARY calls (a task body made of library calls only can be recorded
    # into a command list and replayed from C; torch's own fill / copy kernels cannot)
    def zero_(self, t):
        check(self.lib.mtl_memset_zero(self.stream, t.data_ptr(), t.numel() * t.element_size()), 'mtl_memset_zero')

    def copy_(self, dst, src):
        assert dst.numel() * dst.element_size() == src.numel() * src.element_size()
        check(self.lib.mtl_memcpy_d2d(self.stream, dst.data_ptr(), src.data_ptr(), src.numel() * src.element_size()), 'mtl_memcpy_d2d')

    def axpy_(self, y, x, a):
        check(self.lib.mtl_axpy(self.stream, y.data_ptr(), x.data_ptr(), float(a), y.numel()), 'mtl_axpy')

    def sgd_theta_prime(self, theta0, g, lr, out):
        check(self.lib.mtl_sgd_theta_prime(self.stream, theta0.data_ptr(), g.data_ptr(), float(lr), out.data_ptr(), theta0.numel()),
              'mtl_sgd_theta_prime')

    def _event(self):
        """next event handle of a small ring (an event may be re-recorded once its earlier waits have been ENQUEUED: a wait binds
        to the record that precedes it in host order)"""
        if not self._events:
            for _ in range(8):
                ev = torch.cuda.Event()
                ev.record(torch.cuda.current_stream(self.device))       # materialises the hipEvent_t
                self._events.append(ev)
        ev = self._events[self._ev_next]
        self._ev_next = (self._ev_next + 1) % len(self._events)
        return ev.cuda_event

    # ---- weight gradients
    def wgrad(self, dy, x, rows, n_out, k_in, dw, db=None, kind=None):
        """dw (n_out x k_in) += dy^T x ; db += colsum(dy).  With `kind` (the parameter's name without its layer index) the product is
        only LOGGED: flush_layer_wgrads() issues the products of all layers of a stack as one strided-batch launch per kind.
        Otherwise it goes to the side stream as a single call."""
        if kind is not None:
            self.wgrad_job(kind, n_out, k_in, rows, dy, n_out, x, k_in, dw, k_in, rowsum=db)
            return
        self.defer(lambda: self.gemm(1, 0, n_out, k_in, rows, dy, n_out, x, k_in, dw, k_in, flags=ACCUM, rowsum=db,
                                     task=(rows * n_out, rows * k_in, self.sG, 0, self.sG)))

    def wgrad_job(self, kind, M, N, K, A, lda, B, ldb, C, ldc, n=1, sA=0, sB=0, sC=0, rowsum=None, srow=0):
        """log C_z (M x N) += A_z^T B_z (z < n, strides in floats), rowsum_z += row sums of A_z^T"""
        self._wlog.setdefault((kind, M, N, K, lda, ldb, ldc, n, sA, sB, sC, srow, rowsum is not None), []).append(
            (int(C), int(A), int(B), int(rowsum or 0)))

    def flush_layer_wgrads(self):
        """Issue the logged weight-gradient products on the side stream.  The entries of a kind differ only in their four pointers;
        when those advance by constant strides from layer to layer (they do: per-layer buffers and the parameters of a stack sit at
        constant strides) the whole kind is ONE launch with the layer as outer batch index -- 4 x the workgroups of a single layer's
        product (64 tiles: a quarter of the chip), a quarter of the launches and no fork per sub-layer block."""
        log, self._wlog = self._wlog, {}
        for (kind, M, N, K, lda, ldb, ldc, n, sA, sB, sC, srow, has_rs), items in log.items():
            items.sort()
            nl = len(items)
            steps = [tuple(b[j] - a[j] for j in range(4)) for a, b in zip(items, items[1:])]
            merged = nl > 1 and all(st == steps[0] for st in steps) and all(v % 16 == 0 for v in steps[0])
            tk = (K * lda, K * ldb, self.sG, 0, self.sG)       # task t: its K rows of dy / x, its slice of the gradient stack
            if merged:
                dC, dA, dB, dR = (v // 4 for v in steps[0])
                C0, A0, B0, R0 = items[0]
                self.defer(lambda M=M, N=N, K=K, A0=A0, lda=lda, B0=B0, ldb=ldb, C0=C0, ldc=ldc, nl=nl, n=n, dA=dA, sA=sA, dB=dB, sB=sB,
                           dC=dC, sC=sC, R0=R0, dR=dR, srow=srow, has_rs=has_rs, tk=tk: self.gemm(
                    1, 0, M, N, K, A0, lda, B0, ldb, C0, ldc, flags=ACCUM, batch=nl * n, H=n, sA=(dA, sA), sB=(dB, sB), sC=(dC, sC),
                    rowsum=R0 if has_rs else None, srow=dR, srow_h=srow, task=tk))
            else:
                for C0, A0, B0, R0 in items:
                    self.defer(lambda M=M, N=N, K=K, A0=A0, lda=lda, B0=B0, ldb=ldb, C0=C0, ldc=ldc, n=n, sA=sA, sB=sB, sC=sC, R0=R0,
                               srow=srow, has_rs=has_rs, tk=tk: self.gemm(
                        1, 0, M, N, K, A0, lda, B0, ldb, C0, ldc, flags=ACCUM, batch=n, sA=(sA, 0), sB=(sB, 0), sC=(sC, 0),
                        rowsum=R0 if has_rs else None, srow=srow, task=tk))
        self.flush_side()

    # ---- side stream: deferred parameter-gradient work
    def defer(self, fn):
        if self.use_side_stream:
            self.deferred.append(fn)
        else:
            fn()

    def flush_side(self):
        """Everything enqueued on the main stream so far is visible to the deferred jobs, which are now issued on the side
        stream.  Their inputs are per-block buffers that the main stream never rewrites within this backward."""
        if not self.deferred:
            return
        ev = self._event()
        check(self.lib.mtl_event_record(ev, self.stream), 'mtl_event_record')
        jobs, self.deferred = self.deferred, []
        self._issue_side(ev, jobs)

    def _issue_side(self, ev, jobs):
        check(self.lib.mtl_stream_wait_event(self.side.cuda_stream, ev), 'mtl_stream_wait_event')
        with torch.cuda.stream(self.side):
            self.on_side = True
            try:
                for fn in jobs:
                    fn()
            finally:
                self.on_side = False

    def run_on_side(self, fn):
        """fn() with the side stream current, behind everything the main stream has enqueued so far; returns (fn's result, event
        handle recorded on the side stream after it -- wait_side_event() makes the main stream wait for it)"""
        ev = self._event()
        check(self.lib.mtl_event_record(ev, self.stream), 'mtl_event_record')
        check(self.lib.mtl_stream_wait_event(self.side.cuda_stream, ev), 'mtl_stream_wait_event')
        with torch.cuda.stream(self.side):
            self.on_side = True
            try:
                out = fn()
            finally:
                self.on_side = False
            done = self._event()
            check(self.lib.mtl_event_record(done, self.side.cuda_stream), 'mtl_event_record')
        return out, done

    def wait_side_event(self, ev):
        check(self.lib.mtl_stream_wait_event(self.stream, ev), 'mtl_stream_wait_event')

    def join_side(self):
        self.flush_side()
        if self.use_side_stream:
            ev = self._event()
            check(self.lib.mtl_event_record(ev, self.side.cuda_stream), 'mtl_event_record')
            check(self.lib.mtl_stream_wait_event(self.stream, ev), 'mtl_stream_wait_event')

    def slice_bounds(self):
        """ParamLayout.group_bounds(): the three parameter groups' slices of the flat buffers"""
        return self.L.group_bounds()

    def _slice_done(self, tag):
        if self.slice_hook is None:
            return
        # every kernel that writes this group's gradients must have been ENQUEUED
        if self.deferred:
            raise RuntimeError('slice %r handed over with weight-gradient launches still pending' % tag)
        if isinstance(self.lib, _lib.Recorder):
            self.lib.segment_break(tag)        # the replay stops here and hands control to the same hook
        self.slice_hook(tag)

    def linear_fwd(self, x, rows, k_in, w, b, y, n_out, relu=False):
        """rows: per task; the rows of task t are x + t * rows * k_in, its weights w + t * sP"""
        self.gemm(0, 1, rows, n_out, k_in, x, k_in, w, k_in, y, n_out, bias=b, flags=RELU if relu else 0,
                  task=(rows * k_in, self.sP, rows * n_out, self.sP, 0))

    def linear_bwd(self, x, dy, rows, k_in, n_out, w, dw, db, dx, dx_accum, gate=None, kind=None):
        """dw += dy^T x ; db += colsum(dy) (db None: no bias, or already produced by the LayerNorm backward) ;
        dx (=|+=) dy.W  (gate: ReLU mask source for dx)"""
        self.wgrad(dy, x, rows, n_out, k_in, dw, db, kind=kind)     # db rides on the weight-gradient product (row sums of dy^T)
        if dx is not None:
            self.gemm(0, 0, rows, k_in, n_out, dy, n_out, w, k_in, dx, k_in, gate=gate, ldg=k_in,
                      flags=ACCUM if dx_accum else 0, task=(rows * n_out, self.sP, rows * k_in, 0, 0))

    def _census(self, slot, t, amax_ptr):
        """add the census of the stacked h2 operand `t` (tasks x equal shares) against bound slot `slot` of every task"""
        if self.census is None or amax_ptr is None:
            return
        n = t.numel() // self.nt
        check(self.lib.mtl_h2_census(self.stream, t.data_ptr(), n, amax_ptr, self.census.data_ptr() + 32 * slot, self.nt, n,
                                     12 * _lib.AMAX_SLOTS, 4 * CENSUS_SLOTS), 'mtl_h2_census')

    def colsum(self, x, rows, cols, out, amax=None):
        ws = self.scratch(self.lib.mtl_colsum_workspace(rows, cols))
        check(self.lib.mtl_colsum_accum(self.stream, x, rows, cols, cols, out, ws, amax), 'mtl_colsum_accum')

    def drop_mask(self, name, shape):
        """u8 keep-mask for one dropout site of this pass (None when dropout is off); fresh Philox stream per site."""
        if self.dropout_p <= 0.0:
            self.arena.pop(name, None)
            return None
        m = self.buf(name, shape, torch.uint8)
        self._site += 1
        check(self.lib.mtl_dropout_mask(self.stream, m.data_ptr(), m.numel(), float(self.dropout_p), self._seed_ptr,
                                        self._site << 40), 'mtl_dropout_mask')
        return m

    @property
    def drop_scale(self):
        return 1.0 / (1.0 - self.dropout_p)

    def ln_fwd(self, x, res, g, b, pe, keep, y, xhat, rstd, rows, T, xmask=None):
        """rows: per task (the tasks' row blocks follow each other; task t normalises with g / b + t * sP)"""
        check(self.lib.mtl_layernorm_fwd_g(self.stream, x, res, g, b, pe, keep, xmask.data_ptr() if xmask is not None else None,
                                           self.drop_scale, y, xhat, rstd, rows * self.nt, self.hp.d, T, 1e-5, rows, self.sP),
              'mtl_layernorm_fwd_g')

    def ln_bwd(self, dy, xhat, rstd, g, keep, dz, dg, db, rows, dsum=None, xmask=None, dzm=None, dz2=None):
        """LayerNorm backward; the reduction of its per-wave partials into dgamma / dbeta / dsum is DEFERRED: every instance keeps
        its partials in its own buffer and flush_ln_reduce() adds all of them with one launch (17 launches of 5 us before)."""
        d, nt = self.hp.d, self.nt
        nbytes = self.lib.mtl_layernorm_bwd_g_workspace(rows * nt, d, rows)
        part = self.buf('lnpart.%x' % xhat, (nbytes // 4,))
        check(self.lib.mtl_layernorm_bwd_g(self.stream, dy, xhat, rstd, g, keep, xmask.data_ptr() if xmask is not None else None,
                                           self.drop_scale, dz, dzm, dz2, dg, db, dsum, part.data_ptr(), rows * nt, d, 1, rows, self.sP,
                                           self.sG), 'mtl_layernorm_bwd_g')
        if nt == 1:
            self._ln_pending.append((part.data_ptr(), dg, db, dsum or 0, nbytes // (3 * d * 4), d))
        else:       # task t: its partial rows, its slice of the gradient stack
            wpg = self.lib.mtl_layernorm_bwd_g_waves(rows)
            for t in range(nt):
                go = 4 * t * self.sG
                self._ln_pending.append((part.data_ptr() + 4 * t * wpg * 3 * d, dg + go, db + go, (dsum + go) if dsum else 0, wpg, d))

    def flush_ln_reduce(self):
        pend, self._ln_pending = tuple(self._ln_pending), []
        if not pend:
            return
        dev = self._ln_tables.get(pend)
        if dev is None:
            table = (_lib.LnReduceDesc * len(pend))()
            for t, (part, dg, db, dsum, nw, d) in zip(table, pend):
                t.part, t.dgamma, t.dbeta, t.dsum, t.nw, t.d = part, dg, db, dsum or None, nw, d
            dev = torch.frombuffer(bytearray(bytes(table)), dtype=torch.uint8).to(self.device)
            self._ln_tables[pend] = dev
        check(self.lib.mtl_ln_param_reduce_batch(self.stream, dev.data_ptr(), len(pend), max(p[5] for p in pend)),
              'mtl_ln_param_reduce_batch')

    # ---------------------------------------------------------------- attention / ffn blocks
    def mha_fwd(self, tag, P, pre, xq, Bn, Tq, xkv, Tk, klen, causal, keep, kv_ready=None):
        """Bn: samples PER TASK; the pass carries self.nt tasks whose row blocks (Bn * Tq query rows, Bn * Tk key rows each) follow
        each other in every activation buffer."""
        hp, L, nt, sP = self.hp, self.L, self.nt, self.sP
        d, r, h, dk, dv = hp.d, hp.r, hp.h, hp.dk, hp.dv
        Mq, Mk = Bn * Tq, Bn * Tk                      # rows per task
        Rq = nt * Mq                                   # rows of the pass
        hk, hv = h * dk, h * dv
        o = lambda n: P + 4 * L.off(pre + n)
        t = {}
        # The three projections have identical shapes and their parameters sit at a constant stride in the flat theta / G
        # buffers, so projections that share their input run as ONE strided-batch GEMM per low-rank stage (self-attention:
        # q,k,v; encoder-decoder attention: k,v) instead of two launches (+ a split-K reduction) each.
        groups = self._qkv_groups(pre, xq, Mq, xkv, Mk, hk, hv)
        for names, src, rows in groups:
            n, f0 = len(names), _FULL[names[0]]
            wd = hv if names == 'v' else hk                  # groups of several projections exist only when hk == hv
            if kv_ready is not None and names == 'kv':          # projected for all layers at once by cross_kv_fwd
                a_all, b_all = kv_ready
                self.arena[tag + names + 'a'], self.arena[tag + names] = a_all, b_all
                for i, nm in enumerate(names):
                    self.arena[tag + nm + 'a'], self.arena[tag + nm] = a_all[i], b_all[i]
                    t[nm + 'a'], t[nm] = a_all[i], b_all[i]
                continue
            R = nt * rows
            a_all = self.buf(tag + names + 'a', (n, R, r))
            b_all = self.buf(tag + names, (n, R, wd))
            sa, sb, sbias = (self._pstride(pre, names, sfx) for sfx in ('_linear_a.weight', '_linear_b.weight', '_linear_b.bias'))
            self.gemm(0, 1, rows, r, d, src, d, o(f0 + '_linear_a.weight'), d, a_all.data_ptr(), r, batch=n, sB=(sa, 0),
                      sC=(R * r, 0), task=(rows * d, sP, rows * r, 0, 0))
            self.gemm(0, 1, rows, wd, r, a_all.data_ptr(), r, o(f0 + '_linear_b.weight'), r, b_all.data_ptr(), wd,
                      bias=o(f0 + '_linear_b.bias'), batch=n, sA=(R * r, 0), sB=(sb, 0), sC=(R * wd, 0), sbias=sbias,
                      task=(rows * r, sP, rows * wd, sP, 0))
            for i, nm in enumerate(names):
                self.arena[tag + nm + 'a'], self.arena[tag + nm] = a_all[i], b_all[i]
                t[nm + 'a'], t[nm] = a_all[i], b_all[i]
        self.arena[tag + 'groups'] = groups
        ldS = (Tk + 3) // 4 * 4
        Bn = nt * Bn                                   # the attention core sees one batch of all tasks' samples (no parameters)
        mP = self.drop_mask(tag + 'mP', (Bn, h, Tq, ldS))                       # dropout on the probabilities (:328)
        O = self.buf(tag + 'O', (Rq, hv))
        if self.fused_attn:
            lse = self.buf(tag + 'lse', (Bn, h, Tq))
            check(self.lib.mtl_attn_fwd(self.stream, t['q'].data_ptr(), t['k'].data_ptr(), t['v'].data_ptr(), hk, hk, hv, klen, causal,
                                        1.0 / float(hp.temperature), Bn, h, Tq, Tk, dk, dv, mP.data_ptr() if mP is not None else None,
                                        ldS, self.drop_scale, O.data_ptr(), hv, lse.data_ptr()), 'mtl_attn_fwd')
        else:
            if nt > 1:
                raise NotImplementedError('task-batched passes need the fused attention kernel (head sizes of mtl_attn_supported)')
            S = self.buf(tag + 'P', (Bn, h, Tq, ldS))
            self.gemm(0, 1, Tq, Tk, dk, t['q'].data_ptr(), hk, t['k'].data_ptr(), hk, S.data_ptr(), ldS, batch=Bn * h, H=h,
                      sA=(Tq * hk, dk), sB=(Tk * hk, dk), sC=(h * Tq * ldS, Tq * ldS))
            Pd = self.buf(tag + 'Pd', (Bn, h, Tq, ldS)) if mP is not None else S
            check(self.lib.mtl_softmax_mask_fwd(self.stream, S.data_ptr(), klen, causal, 1.0 / float(hp.temperature), Bn, h, Tq,
                                                Tk, ldS, mP.data_ptr() if mP is not None else None, self.drop_scale,
                                                Pd.data_ptr() if mP is not None else None), 'mtl_softmax_mask_fwd')
            self.gemm(0, 0, Tq, dv, Tk, Pd.data_ptr(), ldS, t['v'].data_ptr(), hv, O.data_ptr(), hv, batch=Bn * h, H=h,
                      sA=(h * Tq * ldS, Tq * ldS), sB=(Tk * hv, dv), sC=(Tq * hv, dv))
        self.arena[tag + 'attn'] = (klen, causal)
        oa = self.buf(tag + 'oa', (Rq, r))
        ob = self.buf(tag + 'ob', (Rq, d))
        self.linear_fwd(O.data_ptr(), Mq, hv, o('output_linear_a.weight'), None, oa.data_ptr(), r)
        self.linear_fwd(oa.data_ptr(), Mq, r, o('output_linear_b.weight'), o('output_linear_b.bias'), ob.data_ptr(), d)
        y = self.buf(tag + 'y', (Rq, d))
        xhat = self.buf(tag + 'xhat', (Rq, d))
        rstd = self.buf(tag + 'rstd', (Rq,))
        mo = self.drop_mask(tag + 'mo', (Rq, d))                                # dropout before the residual add (:303)
        self.ln_fwd(ob.data_ptr(), xq, o('layer_norm.weight'), o('layer_norm.bias'), None, keep, y.data_ptr(),
                    xhat.data_ptr(), rstd.data_ptr(), Mq, Tq, xmask=mo)
        return y

    def mha_bwd(self, tag, P, G, pre, dy, xq, Bn, Tq, xkv, Tk, keep, dxq, dxkv, dxkv_accum, dkv_hoisted=None):
        """dy: grad of the block output.  Writes dxq (overwrite) and dxkv (accumulate when dxkv is dxq or flagged).
        dkv_hoisted: (2, rows, h d_k) slice that receives dK / dV when the K / V projections' backward runs once for all decoder
        layers afterwards (cross_kv_bwd); dxkv is not touched then."""
        hp, L, A, nt, sP, sG = self.hp, self.L, self.arena, self.nt, self.sP, self.sG
        d, r, h, dk, dv = hp.d, hp.r, hp.h, hp.dk, hp.dv
        Mq, Mk = Bn * Tq, Bn * Tk                      # rows per task
        Rq = nt * Mq
        hk, hv = h * dk, h * dv
        o = lambda n: P + 4 * L.off(pre + n)
        g = lambda n: G + 4 * L.off(pre + n)
        ldS = (Tk + 3) // 4 * 4
        O, oa = A[tag + 'O'], A[tag + 'oa']
        kd = _LAYER_BUF.sub(r'\1*.\3', tag)              # 'd3.sa.' -> 'd*.sa.': the kind prefix of this block's weight gradients
        # LayerNorm(o + residual) * keep
        dzb = self.buf(tag + '_dz', (Rq, d))       # kept intact for the deferred dW GEMM; dxq = dz + projections
        mo, mP = A.get(tag + 'mo'), A.get(tag + 'mP')
        dzm = self.buf(tag + '_dzm', (Rq, d)) if mo is not None else None
        self.ln_bwd(dy, A[tag + 'xhat'].data_ptr(), A[tag + 'rstd'].data_ptr(), o('layer_norm.weight'), keep, dzb.data_ptr(),
                    g('layer_norm.weight'), g('layer_norm.bias'), Mq, dsum=g('output_linear_b.bias'), xmask=mo,
                    dzm=dzm.data_ptr() if dzm is not None else None, dz2=dxq)        # dxq = dz: the residual path
        dz = dzm.data_ptr() if dzm is not None else dzb.data_ptr()          # gradient of the (dropped) sub-layer branch
        doa = self.buf(tag + '_doa', (Rq, r))
        dO = self.buf(tag + '_dO', (Rq, hv))
        self.linear_bwd(oa.data_ptr(), dz, Mq, r, d, o('output_linear_b.weight'), g('output_linear_b.weight'),
                        None, doa.data_ptr(), False, kind=kd + 'ob')
        self.linear_bwd(O.data_ptr(), doa.data_ptr(), Mq, hv, r, o('output_linear_a.weight'), g('output_linear_a.weight'),
                        None, dO.data_ptr(), False, kind=kd + 'oa')
        q, k, v = A[tag + 'q'], A[tag + 'k'], A[tag + 'v']
        groups = A[tag + 'groups']
        dfull = {}                                       # gradients of the projected q / k / v, grouped like the forward
        for names, _src, rows in groups:
            d_all = (dkv_hoisted if (dkv_hoisted is not None and names == 'kv') else
                     self.buf(tag + '_d' + names, (len(names), nt * rows, hv if names == 'v' else hk)))
            for i, nm in enumerate(names):
                dfull[nm] = d_all[i]
            dfull[names] = d_all
        dq, dkk, dvv = dfull['q'], dfull['k'], dfull['v']
        if self.fused_attn:
            klen, causal = A[tag + 'attn']
            delta = self.buf('_delta.side' if self.on_side else '_delta', (nt * Bn * h * Tq,))
            check(self.lib.mtl_attn_bwd(self.stream, q.data_ptr(), k.data_ptr(), v.data_ptr(), hk, hk, hv, klen, causal,
                                        1.0 / float(hp.temperature), nt * Bn, h, Tq, Tk, dk, dv, mP.data_ptr() if mP is not None else None,
                                        ldS, self.drop_scale, O.data_ptr(), dO.data_ptr(), hv, A[tag + 'lse'].data_ptr(),
                                        delta.data_ptr(), dq.data_ptr(), dkk.data_ptr(), dvv.data_ptr(), hk, hk, hv), 'mtl_attn_bwd')
        else:
            Pm = A[tag + 'P']
            dP = self.buf('_dP', (Bn, h, Tq, ldS))
            sP = (h * Tq * ldS, Tq * ldS)
            # dV = P^T dO ; dP = dO V^T ; dS = softmax'(P, dP)/temp ; dQ = dS K ; dK = dS^T Q
            Pv = A[tag + 'Pd'] if mP is not None else Pm                      # the probabilities that actually multiplied V
            self.gemm(1, 0, Tk, dv, Tq, Pv.data_ptr(), ldS, dO.data_ptr(), hv, dvv.data_ptr(), hv, batch=Bn * h, H=h,
                      sA=sP, sB=(Tq * hv, dv), sC=(Tk * hv, dv))
            self.gemm(0, 1, Tq, Tk, dv, dO.data_ptr(), hv, v.data_ptr(), hv, dP.data_ptr(), ldS, batch=Bn * h, H=h,
                      sA=(Tq * hv, dv), sB=(Tk * hv, dv), sC=sP)
            check(self.lib.mtl_softmax_bwd(self.stream, Pm.data_ptr(), dP.data_ptr(), 1.0 / float(hp.temperature),
                                           Bn * h * Tq, Tk, ldS, mP.data_ptr() if mP is not None else None, self.drop_scale),
                  'mtl_softmax_bwd')
            self.gemm(0, 0, Tq, dk, Tk, dP.data_ptr(), ldS, k.data_ptr(), hk, dq.data_ptr(), hk, batch=Bn * h, H=h,
                      sA=sP, sB=(Tk * hk, dk), sC=(Tq * hk, dk))
            self.gemm(1, 0, Tk, dk, Tq, dP.data_ptr(), ldS, q.data_ptr(), hk, dkk.data_ptr(), hk, batch=Bn * h, H=h,
                      sA=sP, sB=(Tq * hk, dk), sC=(Tk * hk, dk))
        kv_written = False
        for names, src, rows in groups:
            if dkv_hoisted is not None and names == 'kv':
                continue
            n, f0 = len(names), _FULL[names[0]]
            wd = hv if names == 'v' else hk
            a_all, d_all = A[tag + names + 'a'], dfull[names]
            R = nt * rows
            da_all = self.buf(tag + '_da' + names, (n, R, r))
            sa, sb, sbias = (self._pstride(pre, names, sfx) for sfx in ('_linear_a.weight', '_linear_b.weight', '_linear_b.bias'))
            a_ptr, d_ptr, da_ptr = a_all.data_ptr(), d_all.data_ptr(), da_all.data_ptr()

            # dW_b[i] += d[i]^T a[i]  and  db_b[i] += colsum(d[i])
            self.wgrad_job(kd + names + '.b', wd, r, rows, d_ptr, wd, a_ptr, r, g(f0 + '_linear_b.weight'), r, n=n, sA=R * wd,
                           sB=R * r, sC=sb, rowsum=g(f0 + '_linear_b.bias'), srow=sbias)
            # da[i] = d[i] . W_b[i]
            self.gemm(0, 0, rows, r, wd, d_ptr, wd, o(f0 + '_linear_b.weight'), r, da_ptr, r, batch=n, sA=(R * wd, 0),
                      sB=(sb, 0), sC=(R * r, 0), task=(rows * wd, sP, rows * r, 0, 0))

            # dW_a[i] += da[i]^T x
            self.wgrad_job(kd + names + '.a', r, d, rows, da_ptr, r, src, d, g(f0 + '_linear_a.weight'), d, n=n, sA=R * r, sB=0, sC=sa)
            # dx (+)= sum_i da[i] . W_a[i]: the items of a group accumulate into ONE tensor -> one K-batched launch
            if names[0] == 'q':
                dst, accum = dxq, True
            else:
                dst, accum = dxkv, (dxkv_accum or (dxkv == dxq) or kv_written)
                kv_written = True
            self.gemm(0, 0, rows, d, r, da_ptr, r, o(f0 + '_linear_a.weight'), d, dst, d, flags=ACCUM if accum else 0,
                      kbatch=n, sAk=R * r, sBk=sa, task=(rows * r, sP, rows * d, 0, 0))
        self.flush_side()

    # ---- encoder-decoder attention: K / V projections of all decoder layers in one go
    def _cross_kv_plan(self, Mk):
        """(layer stride, projection stride) in floats when the K / V low-rank parameters of the decoder layers' encoder_attn blocks
        sit at constant strides in the flat buffer (they do for the reference's module tree), else None."""
        hp, L = self.hp, self.L
        if hp.n_dec < 2 or hp.dk != hp.dv:
            return None
        plan = None
        for sfx in ('_linear_a.weight', '_linear_b.weight', '_linear_b.bias'):
            k = [L.off('decoder.layers.%d.encoder_attn.key%s' % (i, sfx)) for i in range(hp.n_dec)]
            v = [L.off('decoder.layers.%d.encoder_attn.value%s' % (i, sfx)) for i in range(hp.n_dec)]
            ls = {b - a for a, b in zip(k, k[1:])}
            ps = {b - a for a, b in zip(k, v)}
            if len(ls) != 1 or len(ps) != 1 or min(ls) <= 0 or min(ps) <= 0 or (min(ls) | min(ps)) % 4:
                return None
            if plan is not None and plan != (min(ls), min(ps)):
                return None
            plan = (min(ls), min(ps))
        return plan

    def cross_kv_fwd(self, P, mem, Mk):
        """k_l, v_l = W_b (W_a mem) (+ b) for every decoder layer l: two launches with batch = 2 n_dec (layers outer, k / v inner)."""
        plan = self._cross_kv_plan(Mk)
        self.arena['xkv.plan'] = plan
        if plan is None:
            return None
        hp, L, nt, sP = self.hp, self.L, self.nt, self.sP
        Ls, Ps = plan
        d, r, wd, NL = hp.d, hp.r, hp.h * hp.dk, hp.n_dec
        Rk = nt * Mk                                   # Mk: key rows per task
        o0 = lambda n: P + 4 * L.off('decoder.layers.0.encoder_attn.' + n)
        a_all = self.buf('xkv.a', (NL, 2, Rk, r))
        b_all = self.buf('xkv.b', (NL, 2, Rk, wd))
        self.gemm(0, 1, Mk, r, d, mem, d, o0('key_linear_a.weight'), d, a_all.data_ptr(), r, batch=2 * NL, H=2, sB=(Ls, Ps),
                  sC=(2 * Rk * r, Rk * r), task=(Mk * d, sP, Mk * r, 0, 0))
        self.gemm(0, 1, Mk, wd, r, a_all.data_ptr(), r, o0('key_linear_b.weight'), r, b_all.data_ptr(), wd, bias=o0('key_linear_b.bias'),
                  batch=2 * NL, H=2, sA=(2 * Rk * r, Rk * r), sB=(Ls, Ps), sC=(2 * Rk * wd, Rk * wd), sbias=Ls, sbias_h=Ps,
                  task=(Mk * r, sP, Mk * wd, sP, 0))
        return a_all, b_all

    def cross_kv_bwd(self, P, G, mem, Mk, dmem):
        """backward of cross_kv_fwd from the dK / dV of all layers ('xkv.d'): the two weight-gradient products (batch 2 n_dec, bias
        gradients folded in) on the side stream, da, and dmem = sum over layers and k / v of da . W_a (two K-batched launches)."""
        hp, L, A, nt, sP, sG = self.hp, self.L, self.arena, self.nt, self.sP, self.sG
        Ls, Ps = A['xkv.plan']
        d, r, wd, NL = hp.d, hp.r, hp.h * hp.dk, hp.n_dec
        Rk = nt * Mk
        o0 = lambda n: P + 4 * L.off('decoder.layers.0.encoder_attn.' + n)
        g0 = lambda n: G + 4 * L.off('decoder.layers.0.encoder_attn.' + n)
        a_ptr, d_ptr = A['xkv.a'].data_ptr(), A['xkv.d'].data_ptr()
        da = self.buf('xkv.da', (NL, 2, Rk, r))
        da_ptr = da.data_ptr()
        self.defer(lambda: self.gemm(1, 0, wd, r, Mk, d_ptr, wd, a_ptr, r, g0('key_linear_b.weight'), r, flags=ACCUM, batch=2 * NL, H=2,
                                     sA=(2 * Rk * wd, Rk * wd), sB=(2 * Rk * r, Rk * r), sC=(Ls, Ps), rowsum=g0('key_linear_b.bias'),
                                     srow=Ls, srow_h=Ps, task=(Mk * wd, Mk * r, sG, 0, sG)))
        self.gemm(0, 0, Mk, r, wd, d_ptr, wd, o0('key_linear_b.weight'), r, da_ptr, r, batch=2 * NL, H=2, sA=(2 * Rk * wd, Rk * wd),
                  sB=(Ls, Ps), sC=(2 * Rk * r, Rk * r), task=(Mk * wd, sP, Mk * r, 0, 0))
        self.defer(lambda: self.gemm(1, 0, r, d, Mk, da_ptr, r, mem, d, g0('key_linear_a.weight'), d, flags=ACCUM, batch=2 * NL, H=2,
                                     sA=(2 * Rk * r, Rk * r), sC=(Ls, Ps), task=(Mk * r, Mk * d, sG, 0, 0)))
        for pj, name in enumerate(('key', 'value')):      # dmem (=|+=) sum_l da[l, pj] . W_a[l, pj]
            self.gemm(0, 0, Mk, d, r, da_ptr + 4 * pj * Rk * r, r, o0(name + '_linear_a.weight'), d, dmem, d, flags=ACCUM if pj else 0,
                      kbatch=NL, sAk=2 * Rk * r, sBk=Ls, task=(Mk * r, sP, Mk * d, 0, 0))
        self.flush_side()

    def _pstride(self, pre, names, suffix):
        """Distance (floats) between consecutive projections' parameters `suffix` in the flat buffer (0 for a single one)."""
        if len(names) == 1:
            return 0
        offs = [self.L.off(pre + _FULL[nm] + suffix) for nm in names]
        return offs[1] - offs[0]

    def _qkv_groups(self, pre, xq, Mq, xkv, Mk, hk, hv):
        """[(names, input pointer, rows)]: projections that can share one strided-batch launch."""
        def uniform(names):
            for sfx in ('_linear_a.weight', '_linear_b.weight', '_linear_b.bias'):
                offs = [self.L.off(pre + _FULL[nm] + sfx) for nm in names]
                steps = {b - a for a, b in zip(offs, offs[1:])}
                if len(steps) != 1 or min(steps) <= 0 or min(steps) % 4:
                    return False
            return hk == hv
        if xq == xkv and Mq == Mk and uniform('qkv'):
            return [('qkv', xq, Mq)]
        if uniform('kv'):
            return [('q', xq, Mq), ('kv', xkv, Mk)]
        return [('q', xq, Mq), ('k', xkv, Mk), ('v', xkv, Mk)]

    def ffn_fwd(self, tag, P, pre, x, rows, T, keep):
        hp, L = self.hp, self.L
        o = lambda n: P + 4 * L.off(pre + n)
        R = self.nt * rows                             # rows: per task
        h1 = self.buf(tag + 'h1', (R, hp.inner))
        h2 = self.buf(tag + 'h2', (R, hp.d))
        self.linear_fwd(x, rows, hp.d, o('linear_1.weight'), o('linear_1.bias'), h1.data_ptr(), hp.inner, relu=True)
        self.linear_fwd(h1.data_ptr(), rows, hp.inner, o('linear_2.weight'), o('linear_2.bias'), h2.data_ptr(), hp.d)
        y = self.buf(tag + 'y', (R, hp.d))
        xhat = self.buf(tag + 'xhat', (R, hp.d))
        rstd = self.buf(tag + 'rstd', (R,))
        mf = self.drop_mask(tag + 'mf', (R, hp.d))                              # dropout before the residual add (:130)
        self.ln_fwd(h2.data_ptr(), x, o('layer_norm.weight'), o('layer_norm.bias'), None, keep, y.data_ptr(), xhat.data_ptr(),
                    rstd.data_ptr(), rows, T, xmask=mf)
        return y

    def ffn_bwd(self, tag, P, G, pre, dy, x, rows, keep, dx):
        hp, L, A = self.hp, self.L, self.arena
        o = lambda n: P + 4 * L.off(pre + n)
        g = lambda n: G + 4 * L.off(pre + n)
        R = self.nt * rows
        dzb = self.buf(tag + '_dz', (R, hp.d))
        mf = A.get(tag + 'mf')
        dzm = self.buf(tag + '_dzm', (R, hp.d)) if mf is not None else None
        self.ln_bwd(dy, A[tag + 'xhat'].data_ptr(), A[tag + 'rstd'].data_ptr(), o('layer_norm.weight'), keep, dzb.data_ptr(),
                    g('layer_norm.weight'), g('layer_norm.bias'), rows, dsum=g('linear_2.bias'), xmask=mf,
                    dzm=dzm.data_ptr() if dzm is not None else None, dz2=dx)         # dx = dz: the residual path
        dbr = dzm.data_ptr() if dzm is not None else dzb.data_ptr()
        h1 = A[tag + 'h1']
        dh1 = self.buf(tag + '_dh1', (R, hp.inner))
        kd = _LAYER_BUF.sub(r'\1*.\3', tag)
        self.linear_bwd(h1.data_ptr(), dbr, rows, hp.inner, hp.d, o('linear_2.weight'), g('linear_2.weight'),
                        None, dh1.data_ptr(), False, gate=h1.data_ptr(), kind=kd + 'w2')
        self.linear_bwd(x, dh1.data_ptr(), rows, hp.d, hp.inner, o('linear_1.weight'), g('linear_1.weight'),
                        g('linear_1.bias'), dx, True, kind=kd + 'w1')
        self.flush_side()

    # ---------------------------------------------------------------- the pass
    def prepare(self, lengths, target, B, T, slot=0, norm_count=None, width=None, frames=None):
        """Host-side integer prep of one batch (modules/decoder.py:55-69 target shifting; every mask is derived inside the
        kernels from these few integers) + asynchronous H2D into STATIC per-slot buffers.  Kept separate from the kernels so
        a captured hipGraph of the pass can be replayed for any batch of the same shape."""
        return self.prepare_tasks([(lengths, target)], B, T, slot, norm_count, width, frames=None if frames is None else [frames])

    def prepare_tasks(self, batches, B, T, slot=0, norm_count=None, width=None, frames=None):
        """prepare() for the batches [(lengths, target)] of several tasks that one task-batched pass carries (all B samples x T
        frames; the decoder width is the largest of the tasks' -- positions beyond a task's own width are padding like any other:
        masked as keys, zeroed as rows, ignored by the loss).  Everything per-sample is concatenated in task order; the loss
        normaliser 1 / n_nonpad and the embedding occurrence chains are per task.
        frames: the tasks' OWN frame counts when their batches were padded to different widths (data.py:77 pads a batch to its
        longest utterance) and stacked at the widest, T.  A task's encoder then has (frames // 2) // 2 positions as in its own
        pass: the positions beyond are padding (the reference compares RAW lengths with the positions it has, SURVEY Q2, so a wider
        pad would otherwise turn into live positions), and forward_device clears the convolution outputs beyond each task's frames."""
        hp = self.hp
        nt = len(batches)
        T4 = (T // 2) // 2
        if frames is not None:
            frames = [int(f) for f in frames]
            if len(frames) != nt or max(frames) > T or min(frames) < 4:
                raise ValueError('frames: one count per task, none above the padded width')
            if all(f == T for f in frames):
                frames = None
        ios = [decoder_io(target, width=width) for _lengths, target in batches]
        Td = max(io[0].shape[1] for io in ios)
        if nt > 1 and any(io[0].shape[1] != Td for io in ios):
            ios = [decoder_io(target, width=Td) for _lengths, target in batches]
        if T4 > hp.src_max_len or Td > hp.tgt_max_len:
            raise ValueError('sequence longer than the positional tables')
        pos = np.arange(T4)[None, :]
        dpos = np.arange(Td)[None, :]
        inv, klen_e, klen_d, keep_e, keep_d, firsts, nexts, n_nonpads = [], [], [], [], [], [], [], []
        for ti, ((lengths, _target), (seq_in_t, seq_out_t)) in enumerate(zip(batches, ios)):
            seq_in, seq_out = seq_in_t.numpy(), seq_out_t.numpy()
            if seq_in.shape[0] != B:
                raise ValueError('every task of a batched pass must bring %d samples' % B)
            lens = lengths.detach().to('cpu', torch.int64).numpy()
            if frames is not None:
                lens = np.minimum(lens, (frames[ti] // 2) // 2)     # a position the task's own pass does not have is padding
            is_pad = seq_in == EOS_ID
            dec_len = (~is_pad).sum(1)
            if not bool((is_pad == (dpos >= dec_len[:, None])).all()):
                raise ValueError('EOS inside a target sequence is not supported')
            # occurrence chains of the decoder input ids (deterministic embedding scatter-add, mtl_embed_bwd); row numbers are
            # those of the whole pass, chains stay inside their task
            flat_in = seq_in.reshape(-1)
            order = np.argsort(flat_in, kind='stable')
            srt = flat_in[order]
            same_as_prev = np.zeros(srt.shape, dtype=bool)
            same_as_prev[1:] = srt[1:] == srt[:-1]
            first = np.empty(flat_in.shape, dtype=np.int32)
            first[order] = ~same_as_prev
            nxt = np.full(flat_in.shape, -1, dtype=np.int32)
            base = ti * B * Td
            nxt[order[:-1]] = np.where(same_as_prev[1:], order[1:] + base, -1)
            n_nonpad = int((seq_out != PAD_ID).sum())
            if norm_count is not None:        # this is a slice of a larger batch: normalise the loss by the WHOLE batch's token count
                n_nonpad = int(norm_count)
            n_nonpads.append(n_nonpad)
            inv.append(1.0 / n_nonpad)
            klen_e.append(np.minimum(lens, T4).astype(np.int32))               # klen_enc (B)      (raw lengths)
            klen_d.append(dec_len.astype(np.int32))                            # klen_dec (B)
            keep_e.append((pos < lens[:, None]).astype(np.int32).reshape(-1))  # keep_enc (B*T4)
            keep_d.append((~is_pad).astype(np.int32).reshape(-1))              # keep_dec (B*Td)
            firsts.append(first)
            nexts.append(nxt)
        head = 2 + nt + (nt & 1)              # seed (8 bytes) | 1 / n_nonpad per task | padding to an even count
        # (numpy from here to the upload: a torch CPU kernel on > 32768 elements -- the staging copy was one -- starts an OpenMP
        # region on torch's whole intra-op pool; in a CPU-quota'd container that stalls the enqueueing thread, hostenv.py)
        meta_np = np.concatenate([
            torch.randint(0, 2 ** 62, (1,), dtype=torch.int64).numpy().view(np.int32),   # dropout seed of this pass (torch CPU RNG), 8-byte aligned
            np.asarray(inv, dtype=np.float32).view(np.int32),                            # 1/n_nonpad (fp32 bits), per task
            np.zeros(head - 2 - nt, dtype=np.int32)] + klen_e + klen_d + keep_e + keep_d + firsts + nexts
            + ([np.asarray(frames, dtype=np.int32)] if frames is not None else []))
        n_meta = int(meta_np.shape[0])
        seq_in = torch.cat([io[0] for io in ios]) if nt > 1 else ios[0][0]
        seq_out = torch.cat([io[1] for io in ios]) if nt > 1 else ios[0][1]
        Bt = nt * B
        # page-locked staging (a ring of STAGE_RING buffers per slot, each guarded by an event): the uploads are truly asynchronous
        # and the host never rewrites a staging buffer whose copy has not been consumed yet.  The ring is deeper than the host may
        # run ahead (pipeline depth + the iteration being resolved + 1), so the event wait never waits in the steady state.
        dev_i32 = self.buf('meta_i32.%d' % slot, (n_meta,), torch.int32)
        ids = self.buf('ids.%d' % slot, (2, Bt, Td), torch.int64)
        _trace.mark('prepare_numpy')
        # the ring holds raw pinned bytes sized for the largest batch seen so far (grown geometrically, the whole ring at once: a
        # page-locked allocation serialises against the device, so variable-length data must not bring one per new shape, and none
        # may happen a few iterations later inside somebody's timed region)
        need = 4 * n_meta + 16 + 16 * Bt * Td
        ring = self._stage.get(slot)
        if ring is None or ring[0]['raw'].numel() < need:
            cap = max(need + need // 2, 1 << 16)
            ring = [dict(raw=torch.empty(cap, dtype=torch.uint8).pin_memory(), ev=None) for _ in range(STAGE_RING)]
            self._stage[slot] = ring           # (an old ring's blocks return to torch's pinned allocator once their copies have completed)
            self._stage_turn[slot] = 0
        turn = self._stage_turn.get(slot, 0)
        self._stage_turn[slot] = (turn + 1) % STAGE_RING
        st = ring[turn]
        if st['ev'] is None:
            st['ev'] = torch.cuda.Event()
        else:
            st['ev'].synchronize()
        off_ids = (4 * n_meta + 15) // 16 * 16
        st_i32 = st['raw'][:4 * n_meta].view(torch.int32)
        st_ids = st['raw'][off_ids:off_ids + 16 * Bt * Td].view(torch.int64).view(2, Bt, Td)
        i32_np, ids_np = st_i32.numpy(), st_ids.numpy()          # views of the pinned memory
        _trace.mark('stage_wait')
        np.copyto(i32_np, meta_np)
        np.copyto(ids_np[0], seq_in.numpy())
        np.copyto(ids_np[1], seq_out.numpy())
        dev_i32.copy_(st_i32, non_blocking=True)
        ids.copy_(st_ids, non_blocking=True)
        st['ev'].record(torch.cuda.current_stream(self.device))
        _trace.mark('stage_upload')
        seed = dev_i32.data_ptr()
        inv_count = seed + 8
        klen_enc = seed + 4 * head
        klen_dec = klen_enc + 4 * Bt
        keep_enc = klen_dec + 4 * Bt
        keep_dec = keep_enc + 4 * Bt * T4
        embed_first = keep_dec + 4 * Bt * Td
        embed_next = embed_first + 4 * Bt * Td
        widths = embed_next + 4 * Bt * Td if frames is not None else None
        return dict(frames=frames, widths=widths, seed=seed, B=B, T=T, Td=Td, nt=nt, n_nonpad=n_nonpads[0], n_nonpads=n_nonpads, gold_host=seq_out,
                    gold_hosts=[io[1] for io in ios], ids=ids, klen_enc=klen_enc, klen_dec=klen_dec,
                    keep_enc=keep_enc, keep_dec=keep_dec, embed_first=embed_first, embed_next=embed_next, inv_count=inv_count)

    def forward(self, theta, x, lengths, target, smoothing=0.0, slot=0):
        """x (B,1,F,T) fp32 on device; lengths (B) int; target (B,L) int64 PAD-padded.  Returns a dict with device
        tensors pred (B,Td,V), gold, hyp (B,Td) int64 and `loss` (1,) fp32; keeps what the backward needs."""
        if x.dim() != 4 or x.shape[1] != 1:
            raise ValueError('expected (B,1,F,T) input')
        # stand-alone passes (validation loops, the joint trainer, the drop-in autograd call) on batches whose width changes from call
        # to call: widened to a repeating width like the trainer's lanes (own border and encoder length kept), from the second width on
        # (not while a test's forward hook reads the activations in the pass's own extent).  Exact with dropout off; with dropout on, the
        # keep-masks of a widened pass differ from the own-width pass's (the Philox element indices follow the buffer extents).
        T_own, frames = int(x.shape[3]), None
        if self.widen == 'auto' and not self._widths_vary:
            self._widths_vary = self._first_width.setdefault(slot, T_own) != T_own
        if self.widen_quantum > 1 and (self.widen == '1' or (self.widen == 'auto' and self._widths_vary and self.forward_hook is None)):
            Tq = max(min(round_width(T_own, self.widen_quantum), 4 * self.hp.src_max_len), T_own)
            if Tq != T_own:
                xp = self.buf('fwd.x.%d' % slot, tuple(x.shape[:3]) + (Tq,))
                xp.zero_()
                xp[:, :, :, :T_own].copy_(x, non_blocking=True)
                x, frames = xp, T_own
        meta = self.prepare(lengths, target, x.shape[0], x.shape[3], slot, frames=frames)
        return self.forward_device(theta, x, meta, smoothing)

    def forward_device(self, theta, x, meta, smoothing=0.0, hyp_out=None, loss_out=None, sP=0):
        """Kernel launches only (hipGraph-capturable): everything batch-dependent comes from `meta`'s device buffers.

        Task-batched pass (meta from prepare_tasks with nt > 1 tasks; trainer/asr/transient_trainer.py:178-237: the tasks of a
        meta-step are independent given theta0): x is (nt * B, 1, F, T) -- or (B, 1, F, T) when every task sees the SAME batch (the
        shared validation batch) -- and task t reads its parameters at theta + t * sP floats: sP = 0 for the training passes (all
        at theta0), sP = layout.total for the validation passes at the theta' stack.  Every transformer kernel runs ONCE over the
        rows of all tasks (task = outermost batch index of the products, row group of LayerNorm / embedding / loss); the
        convolutions and the two products around the encoder's input Linear are issued per task.  loss is (nt,), hyp
        (nt * B, Td)."""
        hp, L, lib, st = self.hp, self.L, self.lib, self.stream
        nt = int(meta.get('nt', 1))
        self.trim_pool()
        assert theta.dtype == torch.float32 and theta.is_contiguous()
        assert theta.numel() == L.total * (nt if sP else 1) and (sP == 0 or sP == L.total)
        x = x.contiguous()
        if x.dtype != torch.float32 or x.device != self.device:
            raise ValueError('input must be fp32 on %s' % self.device)
        Bx, _, F, T = x.shape
        B = meta['B']
        if Bx not in (B, nt * B) or T != meta['T']:
            raise ValueError('input batch does not match the prepared batch')
        sX = B * F * T if Bx == nt * B and nt > 1 else 0          # task stride of the input (0: one batch shared by all tasks)
        T2, F2 = T // 2, F // 2
        T4, F4 = T2 // 2, F2 // 2
        if F4 * 128 != hp.d_in:
            raise ValueError('dim_input %d does not match %d frequency bins' % (hp.d_in, F))
        self.nt, self.sP = nt, int(sP)
        try:
            return self._forward_device(theta, x, meta, smoothing, hyp_out, loss_out, nt, sX, F, T)
        finally:
            self.nt, self.sP = 1, 0        # (the backward restores them from `saved`; an exception mid-pass must not leave nt > 1 behind)

    def _forward_device(self, theta, x, meta, smoothing, hyp_out, loss_out, nt, sX, F, T):
        hp, L, lib, st = self.hp, self.L, self.lib, self.stream
        sP = self.sP
        B = meta['B']
        T2, F2 = T // 2, F // 2
        T4, F4 = T2 // 2, F2 // 2
        ntw = nt if sP else 1                                       # distinct parameter sets of this pass
        P = theta.data_ptr()
        o = lambda n, t=0: P + 4 * (L.off(n) + t * self.sP)
        d, V = hp.d, hp.V
        Td, ids, n_nonpad = meta['Td'], meta['ids'], meta['n_nonpad']
        self._seed_ptr, self._site = meta['seed'], 0
        klen_enc, klen_dec, keep_enc, keep_dec = meta['klen_enc'], meta['klen_dec'], meta['keep_enc'], meta['keep_dec']
        Me, Md = B * T4, B * Td                                     # encoder / decoder rows PER TASK
        Bt = nt * B

        # ---- VGG front-end (per task: its own weights in the validation pass, its own tensor bounds) ----
        y1 = self.buf('y1', (Bt, T, F, 64))
        x3, h2 = self.conv_x3, self.conv_h2
        # h2: device bounds max|tensor| (64 slots each) of y1, p1, y5 | dp2, dy5, dp1 -- raised by the producers' epilogues
        # (forward) or written by the bias-gradient column sums (backward)
        # (6, 7, 8: p2, the permuted input_linear weight, de0 -- operands of the two h2 GEMMs around the encoder's input Linear)
        amax = self.buf('amax', (nt, 12, _lib.AMAX_SLOTS))
        am_ = (lambda i, t=0: amax.data_ptr() + 4 * _lib.AMAX_SLOTS * (12 * t + i)) if h2 else (lambda i, t=0: None)
        if h2:
            check(lib.mtl_memset_zero(st, amax.data_ptr(), nt * 12 * 4 * _lib.AMAX_SLOTS), 'mtl_memset_zero')
        xp = lambda t: x.data_ptr() + 4 * t * sX
        if nt > 1:      # every task's samples in one launch (task = grid dimension; sX = 0: the shared validation batch)
            check(lib.mtl_conv0_relu_fwd_tb(st, x.data_ptr(), o('conv.0.weight'), o('conv.0.bias'), y1.data_ptr(), B, T, F, am_(0), nt, sX,
                                            sP, sP, 12 * _lib.AMAX_SLOTS), 'conv0')
        else:
            for t in range(nt):
                check(lib.mtl_conv0_relu_fwd(st, xp(t), o('conv.0.weight', t), o('conv.0.bias', t), y1[t * B:].data_ptr(), B, T, F,
                                             am_(0, t)), 'conv0')
        # tasks of different frame counts stacked at the widest (prepare_tasks(frames=...)): every convolution output that another
        # convolution reads is cleared beyond its task's own frames, which is the zero border the task's own pass has there; p2's tail
        # rows are encoder padding (keep_enc = 0: nothing reads them, no gradient reaches them)
        widths = meta.get('widths')
        tails = (lambda buf_, T_, row_, shift_: check(lib.mtl_zero_tails(st, buf_.data_ptr(), Bt, T_, row_, widths, shift_, B), 'mtl_zero_tails')) \
            if widths is not None else (lambda *a_: None)
        tails(y1, T, F * 64, 0)
        wf, wd = {}, {}
        wprep = lib.mtl_conv3x3_wprep_h2 if h2 else (lib.mtl_conv3x3_wprep_x3 if x3 else lib.mtl_conv3x3_wprep)
        if h2:
            conv_fwd = lambda s_, x_, w_, b_, y_, ai, ao, *dims: lib.mtl_conv3x3_relu_fwd_h2(s_, x_, ai, w_, b_, y_, ao, *dims)
            conv_fwd_pool = lambda s_, x_, w_, b_, y_, a_, ai, ao, *dims: lib.mtl_conv3x3_relu_pool_fwd_h2(
                s_, x_, ai, w_, b_, y_, a_, ao, *dims)
        else:
            f1 = lib.mtl_conv3x3_relu_fwd_x3 if x3 else lib.mtl_conv3x3_relu_fwd
            f2 = lib.mtl_conv3x3_relu_pool_fwd_x3 if x3 else lib.mtl_conv3x3_relu_pool_fwd
            conv_fwd = lambda s_, x_, w_, b_, y_, ai, ao, *dims: f1(s_, x_, w_, b_, y_, *dims)
            conv_fwd_pool = lambda s_, x_, w_, b_, y_, a_, ai, ao, *dims: f2(s_, x_, w_, b_, y_, a_, *dims)
        for idx, cin, cout in ((2, 64, 64), (5, 64, 128), (7, 128, 128)):
            if h2:      # two fp16 pieces of every (scaled) weight + the scale
                nb = lib.mtl_conv3x3_wprep_h2_bytes(cout, cin)
                nb = (nb + 255) // 256 * 256
                wf[idx] = self.buf('wf%d' % idx, (ntw, nb), torch.uint8)
                wd[idx] = self.buf('wd%d' % idx, (ntw, nb), torch.uint8)
            elif x3:    # three exact bf16 pieces of every weight, [piece][tap][cin/32][cout][32]
                wf[idx] = self.buf('wf%d' % idx, (ntw, 3, 9, cin, cout), torch.bfloat16)
                wd[idx] = self.buf('wd%d' % idx, (ntw, 3, 9, cout, cin), torch.bfloat16)
            else:
                wf[idx] = self.buf('wf%d' % idx, (ntw, 9, cin, cout))
                wd[idx] = self.buf('wd%d' % idx, (ntw, 9, cout, cin))
            if not h2:
                for t in range(ntw):
                    check(wprep(st, o('conv.%d.weight' % idx, t), wf[idx][t].data_ptr(), wd[idx][t].data_ptr(), cout, cin), 'wprep')
        if h2 and ntw > 1:      # all three layers of ALL parameter sets (the theta' stack): one call (two launches)
            spec = []
            for idx, cin, cout in ((2, 64, 64), (5, 64, 128), (7, 128, 128)):
                spec += [o('conv.%d.weight' % idx), wf[idx].data_ptr(), wd[idx].data_ptr(), cout, cin]
            check(lib.mtl_conv3x3_wprep_h2_batch_tb(st, 3, *spec, ntw, sP, wf[2].stride(0), wf[5].stride(0), wf[7].stride(0)), 'wprep')
        elif h2:        # all three layers: one call (two launches) per parameter set
            for t in range(ntw):
                spec = []
                for idx, cin, cout in ((2, 64, 64), (5, 64, 128), (7, 128, 128)):
                    spec += [o('conv.%d.weight' % idx, t), wf[idx][t].data_ptr(), wd[idx][t].data_ptr(), cout, cin]
                check(lib.mtl_conv3x3_wprep_h2_batch(st, 3, *spec), 'wprep')
        p1 = self.buf('p1', (Bt, T2, F2, 64))
        am1 = self.buf('am1', (Bt, T2, F2, 64), torch.uint8)
        y5 = self.buf('y5', (Bt, T2, F2, 128))
        p2 = self.buf('p2', (Bt, T4, F4, 128))
        am2 = self.buf('am2', (Bt, T4, F4, 128), torch.uint8)
        # (a single task with frames of its own -- a widened batch on a lane -- takes the several-task launches too: they skip its tail rows)
        x3_only = x3 and not h2
        if x3_only and (nt > 1 or widths is not None):
            # the exact 3 x bf16 split, same structure as the h2 branch below: ONE launch per layer for the samples of all tasks
            swb = lambda idx: wf[idx].stride(0) * wf[idx].element_size() if ntw > 1 else 0
            skip = widths
            check(lib.mtl_conv3x3_relu_pool_fwd_x3_tb(st, y1.data_ptr(), wf[2].data_ptr(), o('conv.2.bias'), p1.data_ptr(), am1.data_ptr(),
                                                      B, T, F, 64, 64, nt, swb(2), sP, skip, 0), 'conv2')
            tails(p1, T2, F2 * 64, 1)
            check(lib.mtl_conv3x3_relu_fwd_x3_tb(st, p1.data_ptr(), wf[5].data_ptr(), o('conv.5.bias'), y5.data_ptr(), B, T2, F2, 64, 128,
                                                 nt, swb(5), sP, skip, 1), 'conv5')
            tails(y5, T2, F2 * 128, 1)
            check(lib.mtl_conv3x3_relu_pool_fwd_x3_tb(st, y5.data_ptr(), wf[7].data_ptr(), o('conv.7.bias'), p2.data_ptr(), am2.data_ptr(),
                                                      B, T2, F2, 128, 128, nt, swb(7), sP, skip, 1), 'conv7')
            if skip is not None:
                tails(p2, T4, F4 * 128, 2)
                tails(am1, T2, F2 * 64 // 4, 1)
                tails(am2, T4, F4 * 128 // 4, 2)
        elif h2 and (nt > 1 or widths is not None):
            # the samples of all tasks in ONE launch per layer (per-task bounds, weights and biases by stride): a persistent grid's
            # prologue, tail and launch boundary are paid once instead of nt times (2-14 % of a layer: tools/probe/conv_batch_tasks.py);
            # per task bitwise the per-task launches (tests/test_ops_gpu.py)
            AS = 12 * _lib.AMAX_SLOTS
            sw = lambda idx: wf[idx].stride(0) if ntw > 1 else 0
            # (tasks with frame counts of their own: the launches leave out the pixel-tile rows beyond a task's frames -- `skip` -- and
            # everything a later kernel reads there is cleared: activations, pooled maps and their arg-max bytes)
            skip = widths
            check(lib.mtl_conv3x3_relu_pool_fwd_h2_tb(st, y1.data_ptr(), am_(0), wf[2].data_ptr(), o('conv.2.bias'), p1.data_ptr(), am1.data_ptr(),
                                                      am_(1), B, T, F, 64, 64, nt, sw(2), sP, AS, AS, skip, 0), 'conv2')
            tails(p1, T2, F2 * 64, 1)
            check(lib.mtl_conv3x3_relu_fwd_h2_tb(st, p1.data_ptr(), am_(1), wf[5].data_ptr(), o('conv.5.bias'), y5.data_ptr(), am_(2),
                                                 B, T2, F2, 64, 128, nt, sw(5), sP, AS, AS, skip, 1), 'conv5')
            tails(y5, T2, F2 * 128, 1)
            check(lib.mtl_conv3x3_relu_pool_fwd_h2_tb(st, y5.data_ptr(), am_(2), wf[7].data_ptr(), o('conv.7.bias'), p2.data_ptr(), am2.data_ptr(),
                                                      am_(6), B, T2, F2, 128, 128, nt, sw(7), sP, AS, AS, skip, 1), 'conv7')
            if skip is not None:
                tails(p2, T4, F4 * 128, 2)
                tails(am1, T2, F2 * 64 // 4, 1)          # (bytes, four to a float)
                tails(am2, T4, F4 * 128 // 4, 2)
        else:
            def c2(t, tw, sl):
                check(conv_fwd_pool(st, y1[sl].data_ptr(), wf[2][tw].data_ptr(), o('conv.2.bias', t), p1[sl].data_ptr(), am1[sl].data_ptr(),
                                    am_(0, t), am_(1, t), B, T, F, 64, 64), 'conv2')

            def c5(t, tw, sl):
                check(conv_fwd(st, p1[sl].data_ptr(), wf[5][tw].data_ptr(), o('conv.5.bias', t), y5[sl].data_ptr(), am_(1, t), am_(2, t),
                               B, T2, F2, 64, 128), 'conv5')

            def c7(t, tw, sl):
                check(conv_fwd_pool(st, y5[sl].data_ptr(), wf[7][tw].data_ptr(), o('conv.7.bias', t), p2[sl].data_ptr(), am2[sl].data_ptr(),
                                    am_(2, t), am_(6, t), B, T2, F2, 128, 128), 'conv7')
            per_task = [(t, (t if sP else 0), slice(t * B, (t + 1) * B)) for t in range(nt)]
            if widths is None:
                for a_ in per_task:          # task by task
                    c2(*a_)
                    c5(*a_)
                    c7(*a_)
            else:                            # layer by layer: a layer's tails are cleared (all tasks at once) before the next layer reads them
                for a_ in per_task:
                    c2(*a_)
                tails(p1, T2, F2 * 64, 1)
                for a_ in per_task:
                    c5(*a_)
                tails(y5, T2, F2 * 128, 1)
                for a_ in per_task:
                    c7(*a_)

        if h2 and self.census is not None:          # (stream order: every producer epilogue has raised its bound by now)
            for slot, t_ in ((0, y1), (1, p1), (2, y5), (6, p2)):
                self._census(slot, t_, am_(slot))

        # ---- encoder ----
        wp = self.buf('wp_in', (ntw, d, hp.d_in))
        # am_(7): max|w| rides along; the theta' stack in one launch
        check(lib.mtl_permute_hc_tb(st, o('encoder.input_linear.weight'), wp.data_ptr(), d, 128, F4, 0, am_(7), ntw, self.sP, d * hp.d_in,
                                    12 * _lib.AMAX_SLOTS), 'permute')
        # decoder prologue: embedding (+ PE, dropout) and layer 0's self-attention block read the labels and theta only
        def dec_prologue():
            d0_ = self.buf('dec_in.y', (nt * Md, d))
            me = self.drop_mask('dec_in.me', (nt * Md, d))                      # dropout(emb + PE) (modules/decoder.py:96)
            check(lib.mtl_embed_pe_fwd_g(self.stream, ids.data_ptr(), o('decoder.trg_embedding.weight'), self.pe_dec.data_ptr(),
                                         d0_.data_ptr(), nt * Md, Td, d, me.data_ptr() if me is not None else None, self.drop_scale,
                                         Md, self.sP), 'embed')
            if hp.n_dec == 0:
                return d0_, None
            return d0_, self.mha_fwd('d0.sa.', P, 'decoder.layers.0.self_attn.', d0_.data_ptr(), B, Td, d0_.data_ptr(), Td, klen_dec, 1,
                                     keep_dec)
        pro_done = None
        self.dec0_on_side = bool(self.use_side_stream and hp.n_dec > 0)
        if self.dec0_on_side:
            (d0, a0), pro_done = self.run_on_side(dec_prologue)      # under the input Linear and the encoder
        e0 = self.buf('e0', (nt * Me, d))
        # the encoder's input Linear (5120 -> 512) and its data gradient: 'x3' = one task-batched launch each on the bf16-split engine
        # (exact 3-piece operands, no bounds needed; MTL_IN_LINEAR=x3), 'h2' (default) = one task-batched launch each on two fp16 pieces
        # (mtl_gemm_h2_tb: the same tile engine with three MFMAs per step)
        self.in_h2 = h2 and self.in_linear == 'h2'
        if self.in_h2:      # the two compute-bound products of the pass on fp16 pairs: e0 = p2 . wp^T here, dp2 = de0 . wp in the backward
            am_st = 12 * _lib.AMAX_SLOTS                          # floats between two tasks' bounds
            # ONE task-batched launch on the tile engine of mtl_gemm_x3.hip (per-task bounds by stride)
            check(lib.mtl_gemm_h2_tb(st, 1, Me, d, hp.d_in, p2.data_ptr(), hp.d_in, am_(6), am_st, wp.data_ptr(), hp.d_in, am_(7),
                                     am_st if sP else 0, e0.data_ptr(), d, o('encoder.input_linear.bias'), None, 0, nt,
                                     Me * hp.d_in, d * hp.d_in if sP else 0, Me * d, self.sP, self.gemm_ws.data_ptr(), self.gemm_ws.numel() * 4),
                  'mtl_gemm_h2_tb')
        else:
            self.gemm(0, 1, Me, d, hp.d_in, p2.data_ptr(), hp.d_in, wp.data_ptr(), hp.d_in, e0.data_ptr(), d,
                      bias=o('encoder.input_linear.bias'), task=(Me * hp.d_in, d * hp.d_in if sP else 0, Me * d, self.sP, 0))
        ex = self.buf('enc_in.y', (nt * Me, d))
        self.ln_fwd(e0.data_ptr(), None, o('encoder.layer_norm_input.weight'), o('encoder.layer_norm_input.bias'),
                    self.pe_enc.data_ptr(), None, ex.data_ptr(), self.buf('enc_in.xhat', (nt * Me, d)).data_ptr(),
                    self.buf('enc_in.rstd', (nt * Me,)).data_ptr(), Me, T4)
        cur = ex
        enc_inputs = []
        for i in range(hp.n_enc):
            pre = 'encoder.layers.%d.' % i
            enc_inputs.append(cur)
            a = self.mha_fwd('e%d.sa.' % i, P, pre + 'self_attn.', cur.data_ptr(), B, T4, cur.data_ptr(), T4, klen_enc, 0, keep_enc)
            cur = self.ffn_fwd('e%d.ff.' % i, P, pre + 'pos_ffn.', a.data_ptr(), Me, T4, keep_enc)
        mem = cur

        # ---- decoder ----
        if pro_done is not None:
            self.wait_side_event(pro_done)
        else:
            d0, a0 = dec_prologue()
        cur = d0
        xkv = self.cross_kv_fwd(P, mem.data_ptr(), Me)
        for i in range(hp.n_dec):
            pre = 'decoder.layers.%d.' % i
            a = a0 if i == 0 else self.mha_fwd('d%d.sa.' % i, P, pre + 'self_attn.', cur.data_ptr(), B, Td, cur.data_ptr(), Td,
                                               klen_dec, 1, keep_dec)
            c = self.mha_fwd('d%d.ca.' % i, P, pre + 'encoder_attn.', a.data_ptr(), B, Td, mem.data_ptr(), T4, klen_enc, 0, keep_dec,
                             kv_ready=(xkv[0][i], xkv[1][i]) if xkv is not None else None)
            cur = self.ffn_fwd('d%d.ff.' % i, P, pre + 'pos_ffn.', c.data_ptr(), Md, Td, keep_dec)
        pred = self.buf('pred', (Bt, Td, V))
        self.gemm(0, 1, Md, V, d, cur.data_ptr(), d, o('decoder.output_linear.weight'), d, pred.data_ptr(), V,
                  task=(Md * d, self.sP, Md * V, 0, 0))

        # ---- loss + arg-max ----
        lse = self.buf('lse', (nt * Md,))
        # hyp_out / loss_out: caller-owned destinations (the trainer's per-pass read-back slots: no device copies afterwards)
        hyp = self.buf('hyp', (Bt, Td), torch.int64) if hyp_out is None else hyp_out
        rowloss = self.buf('rowloss', (nt * Md,))
        loss = self.buf('loss', (nt,)) if loss_out is None else loss_out
        gold_ptr = ids.data_ptr() + 8 * nt * Md
        if nt == 1:
            check(lib.mtl_ce_argmax_fwd(st, pred.data_ptr(), gold_ptr, Md, V, V, PAD_ID, float(smoothing), 0, meta['inv_count'],
                                        lse.data_ptr(), hyp.data_ptr(), rowloss.data_ptr(), loss.data_ptr()), 'ce_fwd')
        else:
            check(lib.mtl_ce_argmax_fwd_g(st, pred.data_ptr(), gold_ptr, nt * Md, V, V, PAD_ID, float(smoothing), meta['inv_count'],
                                          lse.data_ptr(), hyp.data_ptr(), rowloss.data_ptr(), loss.data_ptr(), Md), 'ce_fwd')
        self.saved = dict(theta=theta, x=x, B=B, T=T, F=F, Td=Td, n_nonpad=n_nonpad, smoothing=float(smoothing), meta=meta,
                          klen_enc=klen_enc, klen_dec=klen_dec, keep_enc=keep_enc, keep_dec=keep_dec, dec_last=cur,
                          enc_inputs=enc_inputs, nt=nt, sP=self.sP, sX=sX)
        if self.forward_hook is not None:
            self.forward_hook(self)
        # T4 / frames: the extent of the arena's encoder-side activations (a widened stand-alone pass, PassEngine.forward, carries
        # T // 4 >= own frames // 4 positions per utterance: readers of eng.arena index with T4, the batch's own extent is frames)
        return dict(pred=pred, gold=ids[1], hyp=hyp, loss=loss, gold_host=meta['gold_host'], n_nonpad=n_nonpad, T4=T4,
                    frames=meta.get('frames'))

    # ---------------------------------------------------------------- greedy decoding
    def decode_session(self, theta, mem, B, T4, S, shared_memory=False):
        """K/V-cached incremental decoder over `B` hypothesis rows and up to `S` positions (greedy and beam search share it).
        mem: device pointer of the encoder output, (B, T4, d) -- or (1, T4, d) with shared_memory=True, when all rows are
        hypotheses of ONE utterance (beam search) and address the same cross-attention keys / values with batch stride 0."""
        return _DecodeSession(self, theta, mem, B, T4, S, shared_memory)

    def greedy_decode(self, theta, mem, B, T4, start_token, max_steps=300):
        """Decoder.greedy_search (modules/decoder.py:131-185) on the device: max_steps arg-max steps from `start_token`,
        no padding masks (the reference passes dec_enc_attn_mask=None and an all-ones non_pad_mask), only causality.
        The reference re-runs the whole decoder on the growing prefix at every step; here each layer keeps a K/V cache
        and only the new position is computed (same per-row arithmetic), and the chosen token is fed back through device
        memory, so the 300 steps run without a single host synchronisation.  Returns the (max_steps, B) int64 token ids."""
        if max_steps + 1 > self.hp.tgt_max_len:
            raise ValueError('tgt_max_len too small for %d decoding steps' % max_steps)
        ses = self.decode_session(theta, mem, B, T4, max_steps + 1)          # (eager: cache zeroing, cross-attention keys / values)
        ys = self.buf('g.ys', (max_steps + 1, B), torch.int64)
        ys[0].fill_(int(start_token))

        def steps():
            for t in range(max_steps):
                ses.step(t, ys.data_ptr() + 8 * t * B)
                ses.argmax_into(ys.data_ptr() + 8 * (t + 1) * B)
        # The ~64 launches of a step cost the host more (8 us each through ctypes) than the device (a decode was host-bound at 0.53 ms
        # per step): the calls of all steps are recorded once per (parameters, shapes, buffers) into ONE command list -- every address is
        # fixed: the session's buffers come from the pool by name and shape -- and replayed from C (first sighting: plain eager run,
        # second: recorded, like TransientTrainer._run_recorded).
        key = (theta.data_ptr(), int(mem), B, T4, max_steps, ys.data_ptr(), self.stream, self.scratch_epoch, ses.fast)
        lists = self.__dict__.setdefault('_decode_lists', {})
        ent = lists.get(key)
        if self.prof is not None or isinstance(self.lib, _lib.Recorder):
            steps()
        elif ent is None:
            while len(lists) >= 4:
                lists.pop(next(iter(lists)))
            lists[key] = 'warm'
            steps()
        elif ent == 'warm':
            cl, real = _lib.CommandList(), self.lib
            self.lib = _lib.Recorder(real, cl)
            try:
                steps()
            finally:
                self.lib = real
            if key[7] == self.scratch_epoch:
                lists[key] = cl.finish()
        else:
            ent.run()
        return ys[1:]

    def beam_decode(self, theta, mem_row, T4, start_token, beam_width, nbest, tgt_max_len, num_words, eos_id=EOS_ID, c_weight=1.0):
        """Decoder.beam_search (modules/decoder.py:187-291, no LM rescoring) for ONE utterance: the host keeps the reference's
        hypothesis bookkeeping verbatim in behaviour (expansion order, stable sorts, cumulative truncation to the beam, EOS
        forced at step T' - 1, final_score = score + sqrt(num_words) * c_weight, fp32 score arithmetic); the device runs one
        K/V-cached decoder step for all live hypotheses per iteration (caches re-ordered by parent) and returns the last
        position's logits and log-sum-exp.  num_words(yseq) -> int is the caller's word counter (it needs the vocabulary).
        -> list of (yseq incl. start token and EOS, final_score) sorted best first, at most nbest."""
        import numpy as np
        W = int(beam_width)
        steps = int(tgt_max_len)
        if steps > self.pe_dec.shape[0] or W < 1:
            raise ValueError('bad beam search arguments')
        ses = self.decode_session(theta, mem_row, W, T4, steps, shared_memory=True)
        toks = self.buf('b.tok', (W,), torch.int64)
        hyp_buf = self.buf('g.hyp', (W,), torch.int64)
        # the device side of position i (one decoder step for the W rows + log-sum-exp) is the same call sequence for every utterance:
        # eager at its first sighting, recorded into a command list at the second, replayed from C afterwards (as greedy_decode does for
        # all its steps at once; here the host ranks the hypotheses between two positions, so there is one list per position)
        key = ('beam', theta.data_ptr(), W, T4, steps, toks.data_ptr(), self.stream, ses.fast)
        store = self.__dict__.setdefault('_beam_lists', {})
        ent = store.get(key)
        if ent is None or ent['epoch'] != self.scratch_epoch:
            while len(store) >= 2:
                store.pop(next(iter(store)))
            ent = store[key] = dict(epoch=self.scratch_epoch, seen=set(), lists={})

        def body(i):
            ses.step(i, toks.data_ptr())
            ses.argmax_into(hyp_buf.data_ptr())                   # (leaves the log-sum-exp of the W rows in ses.junk[:W])

        def device_step(i):
            cl = ent['lists'].get(i)
            if self.prof is not None or isinstance(self.lib, _lib.Recorder) or ent['epoch'] != self.scratch_epoch:
                body(i)
            elif cl is not None:
                cl.run()
            elif i in ent['seen']:
                cl, real = _lib.CommandList(), self.lib
                self.lib = _lib.Recorder(real, cl)
                try:
                    body(i)
                finally:
                    self.lib = real
                if ent['epoch'] == self.scratch_epoch:
                    ent['lists'][i] = cl.finish()
            else:
                ent['seen'].add(i)
                body(i)
        hyps = [dict(score=np.float32(0.0), yseq=[int(start_token)], row=0)]
        ended = []
        for i in range(steps):
            n = len(hyps)
            rows = [h['row'] for h in hyps] + [0] * (W - n)
            if i > 0:
                ses.reorder(rows, i)                              # row r of the caches <- its parent's rows 0..i-1
            toks.copy_(torch.tensor([h['yseq'][-1] for h in hyps] + [int(eos_id)] * (W - n), dtype=torch.int64))
            device_step(i)
            logits, lse = ses.logits.cpu(), ses.junk[:W].cpu()
            local = (logits[:n] - lse[:n].unsqueeze(1))           # F.log_softmax of the last position, fp32
            kept = []
            for r, h in enumerate(hyps):
                best, ids = torch.topk(local[r], W)
                for j in range(W):
                    kept.append(dict(score=np.float32(h['score'] + np.float32(best[j].item())), yseq=h['yseq'] + [int(ids[j])], row=r))
                kept = sorted(kept, key=lambda x: x['score'], reverse=True)[:W]
            hyps = kept
            if i == T4 - 1:
                for h in hyps:
                    h['yseq'] = h['yseq'] + [int(eos_id)]
            live = []
            for h in hyps:
                if h['yseq'][-1] == eos_id:
                    h['final_score'] = np.float32(h['score'] + np.float32(math.sqrt(num_words(h['yseq'])) * c_weight))
                    ended.append(h)
                else:
                    live.append(h)
            hyps = live
            if not hyps:
                break
        out = sorted(ended, key=lambda x: x['final_score'], reverse=True)[:min(len(ended), int(nbest))]
        return [(h['yseq'], float(h['final_score'])) for h in out]

    def backward(self, grad, scale=1.0, dpred=None, sG=0):
        """Accumulate `scale` * dLoss/dtheta of the LAST forward into the flat buffer `grad` (+=).
        dpred: optional externally supplied gradient w.r.t. pred (B,Td,V) instead of the fused CE backward.
        Task-batched pass: task t accumulates into grad + t * sG floats (sG = layout.total: a stack of per-task gradients)."""
        S = self.saved
        if S is None:
            raise RuntimeError('backward() without a preceding forward()')
        hp, L, lib, st, A = self.hp, self.L, self.lib, self.stream, self.arena
        nt = S['nt']
        self.nt, self.sP, self.sG = nt, S['sP'], int(sG) if nt > 1 else 0
        if nt > 1 and (sG != L.total or grad.numel() != nt * L.total or dpred is not None):
            raise ValueError('a task-batched backward accumulates into a (tasks, layout.total) gradient stack')
        assert grad.numel() == L.total * nt and grad.is_contiguous()
        sP, sG, sX = self.sP, self.sG, S['sX']
        theta = S['theta']
        P, G = theta.data_ptr(), grad.data_ptr()
        o = lambda n, t=0: P + 4 * (L.off(n) + t * sP)
        g = lambda n, t=0: G + 4 * (L.off(n) + t * sG)
        B, T, F, Td = S['B'], S['T'], S['F'], S['Td']
        T2, F2 = T // 2, F // 2
        T4, F4 = T2 // 2, F2 // 2
        Me, Md, d, V = B * T4, B * Td, hp.d, hp.V                   # rows PER TASK
        keep_enc, keep_dec = S['keep_enc'], S['keep_dec']

        if dpred is None:
            ldd = (V + 3) // 4 * 4        # padded leading dimension -> 16-byte operand loads in the two GEMMs below
            dlog = self.buf('_dpred', (nt * Md, ldd))
            gold_ptr = S['meta']['ids'].data_ptr() + 8 * nt * Md
            check(lib.mtl_ce_bwd_g(st, A['pred'].data_ptr(), A['lse'].data_ptr(), gold_ptr, nt * Md, V, V, PAD_ID, S['smoothing'],
                                   float(scale), S['meta']['inv_count'], dlog.data_ptr(), ldd, Md), 'ce_bwd')
            dlog_ptr = dlog.data_ptr()
        else:
            ldd = V
            dlog = dpred.contiguous()
            if scale != 1.0:
                dlog = dlog * scale
            dlog_ptr = dlog.data_ptr()
        dA = self.buf('_dxA', (nt * Md, d))
        dB = self.buf('_dxB', (nt * Md, d))
        dmem = self.buf('_dmem', (nt * Me, d))
        last = S['dec_last']
        # vocab projection (no bias)
        self.defer(lambda: self.gemm(1, 0, V, d, Md, dlog_ptr, ldd, last.data_ptr(), d, g('decoder.output_linear.weight'), d,
                                     flags=ACCUM, task=(Md * ldd, Md * d, sG, 0, 0)))    # (lda = ldd != V: stays a single call)
        self.gemm(0, 0, Md, d, V, dlog_ptr, ldd, o('decoder.output_linear.weight'), d, dA.data_ptr(), d,
                  task=(Md * ldd, sP, Md * d, 0, 0))
        self.flush_side()
        dcur, dnext = dA, dB
        hoisted = A.get('xkv.plan') is not None
        mem_ptr = A['e%d.ff.y' % (hp.n_enc - 1)].data_ptr() if hp.n_enc else A['enc_in.y'].data_ptr()
        dkv_all = self.buf('xkv.d', (hp.n_dec, 2, nt * Me, hp.h * hp.dk)) if hoisted else None
        def embed_bwd(dx):
            me = A.get('dec_in.me')
            check(lib.mtl_embed_bwd_g(self.stream, S['meta']['ids'].data_ptr(), S['meta']['embed_first'], S['meta']['embed_next'],
                                      dx.data_ptr(), g('decoder.trg_embedding.weight'), nt * Md, d, PAD_ID,
                                      me.data_ptr() if me is not None else None, self.drop_scale, Md, sG), 'embed_bwd')
        embed_done = False
        for i in reversed(range(hp.n_dec)):
            pre = 'decoder.layers.%d.' % i
            x_in = A['dec_in.y'] if i == 0 else A['d%d.ff.y' % (i - 1)]
            sa_y, ca_y = A['d%d.sa.y' % i], A['d%d.ca.y' % i]
            self.ffn_bwd('d%d.ff.' % i, P, G, pre + 'pos_ffn.', dcur.data_ptr(), ca_y.data_ptr(), Md, keep_dec, dnext.data_ptr())
            dcur, dnext = dnext, dcur
            self.mha_bwd('d%d.ca.' % i, P, G, pre + 'encoder_attn.', dcur.data_ptr(), sa_y.data_ptr(), B, Td, mem_ptr, T4, keep_dec,
                         dnext.data_ptr(), dmem.data_ptr(), i != hp.n_dec - 1, dkv_hoisted=dkv_all[i] if hoisted else None)
            dcur, dnext = dnext, dcur
            if i == 0 and getattr(self, 'dec0_on_side', False):
                # layer 0's self-attention block and the embedding feed parameter gradients only (the encoder's backward needs dmem,
                # which is complete): side stream, under the encoder's backward.  dcur / dnext are not touched by the main stream
                # again; the block's weight-gradient products are logged like every layer's and issued behind it on the same stream.
                def dec_epilogue(pre=pre, x_in=x_in, dcur=dcur, dnext=dnext):
                    self.mha_bwd('d0.sa.', P, G, pre + 'self_attn.', dcur.data_ptr(), x_in.data_ptr(), B, Td, x_in.data_ptr(), Td,
                                 keep_dec, dnext.data_ptr(), dnext.data_ptr(), True)
                    embed_bwd(dnext)
                self.run_on_side(dec_epilogue)
                embed_done = True
                continue
            self.mha_bwd('d%d.sa.' % i, P, G, pre + 'self_attn.', dcur.data_ptr(), x_in.data_ptr(), B, Td, x_in.data_ptr(), Td,
                         keep_dec, dnext.data_ptr(), dnext.data_ptr(), True)
            dcur, dnext = dnext, dcur
        if hoisted:
            self.cross_kv_bwd(P, G, mem_ptr, Me, dmem.data_ptr())
        self.flush_layer_wgrads()       # the decoder stack's weight gradients: one launch per parameter kind
        if not embed_done:
            embed_bwd(dcur)
        if self.slice_hook is not None:
            # every decoder-group gradient is enqueued (weights: side stream; LayerNorm partials: main, layer 0's on the side stream):
            # reduce the decoder's LayerNorm / bias partials now, BEHIND both, instead of with the encoder's at the end of the pass
            if self.use_side_stream:
                self.run_on_side(self.flush_ln_reduce)
            else:
                self.flush_ln_reduce()
            self._slice_done('decoder')

        # ---- encoder ----
        eA = self.buf('_deA', (nt * Me, d))
        dcur, dnext = dmem, eA
        for i in reversed(range(hp.n_enc)):
            pre = 'encoder.layers.%d.' % i
            x_in = A['enc_in.y'] if i == 0 else A['e%d.ff.y' % (i - 1)]
            self.ffn_bwd('e%d.ff.' % i, P, G, pre + 'pos_ffn.', dcur.data_ptr(), A['e%d.sa.y' % i].data_ptr(), Me, keep_enc,
                         dnext.data_ptr())
            dcur, dnext = dnext, dcur
            self.mha_bwd('e%d.sa.' % i, P, G, pre + 'self_attn.', dcur.data_ptr(), x_in.data_ptr(), B, T4, x_in.data_ptr(), T4,
                         keep_enc, dnext.data_ptr(), dnext.data_ptr(), True)
            dcur, dnext = dnext, dcur
        # input LayerNorm (+PE: no grad) and input_linear
        de0 = dnext
        self.ln_bwd(dcur.data_ptr(), A['enc_in.xhat'].data_ptr(), A['enc_in.rstd'].data_ptr(), o('encoder.layer_norm_input.weight'),
                    None, de0.data_ptr(), g('encoder.layer_norm_input.weight'), g('encoder.layer_norm_input.bias'), Me,
                    dsum=g('encoder.input_linear.bias'))
        self.flush_layer_wgrads()       # the encoder stack's weight gradients
        p2, y5, p1, y1 = A['p2'], A['y5'], A['p1'], A['y1']
        dwp = self.buf('_dwp', (nt, d, hp.d_in))
        dp2 = self.buf('_dp2', (nt * B, T4, F4, 128))
        h2 = self.conv_h2
        amax = A['amax']
        am_ = (lambda i, t=0: amax.data_ptr() + 4 * _lib.AMAX_SLOTS * (12 * t + i)) if h2 else (lambda i, t=0: None)   # y1, p1, y5 | dp2, dy5, dp1
        am_st = 12 * _lib.AMAX_SLOTS
        if self.in_h2 and nt > 1:
            check(lib.mtl_absmax_f32_tb(st, de0.data_ptr(), Me * d, am_(8), nt, Me * d, am_st), 'mtl_absmax_f32')
        elif self.in_h2:
            for t in range(nt):
                check(lib.mtl_absmax_f32(st, de0[t * Me:].data_ptr(), Me * d, am_(8, t)), 'mtl_absmax_f32')
        if self.in_h2:
            self._census(8, de0, am_(8))
        if self.in_h2:      # dW = de0^T . p2 on fp16 pairs too: both bounds (slots 8, 6) exist for the data gradient below
            check(lib.mtl_gemm_h2_tn_tb(st, d, hp.d_in, Me, de0.data_ptr(), d, am_(8), am_st, p2.data_ptr(), hp.d_in, am_(6), am_st,
                                        dwp.data_ptr(), hp.d_in, nt, Me * d, Me * hp.d_in, d * hp.d_in), 'mtl_gemm_h2_tn_tb')
        else:
            self.gemm(1, 0, d, hp.d_in, Me, de0.data_ptr(), d, p2.data_ptr(), hp.d_in, dwp.data_ptr(), hp.d_in,
                      task=(Me * d, Me * hp.d_in, d * hp.d_in, 0, 0))
        check(lib.mtl_permute_hc_tb(st, dwp.data_ptr(), g('encoder.input_linear.weight'), d, 128, F4, 1, None, nt, d * hp.d_in, sG, 0), 'permute_inv')
        if self.slice_hook is not None:
            self.flush_ln_reduce()         # the encoder's LayerNorms (+ the input LayerNorm): their partials were all produced on this stream
            self._slice_done('encoder')
        if self.in_h2:
            # dp2 = (de0 . wp) gated by p2 > 0, straight from the un-transposed weight, all tasks in one launch
            check(lib.mtl_gemm_h2_tb(st, 0, Me, hp.d_in, d, de0.data_ptr(), d, am_(8), am_st, A['wp_in'].data_ptr(), hp.d_in, am_(7),
                                     am_st if sP else 0, dp2.data_ptr(), hp.d_in, None, p2.data_ptr(), hp.d_in, nt, Me * d,
                                     d * hp.d_in if sP else 0, Me * hp.d_in, 0, None, 0), 'mtl_gemm_h2_tb')
        else:
            self.gemm(0, 0, Me, hp.d_in, d, de0.data_ptr(), d, A['wp_in'].data_ptr(), hp.d_in, dp2.data_ptr(), hp.d_in,
                      gate=p2.data_ptr(), ldg=hp.d_in, task=(Me * d, d * hp.d_in if sP else 0, Me * hp.d_in, 0, 0))

        self.flush_side()
        # ---- VGG front-end (per task) ----
        dgrad_fn = lib.mtl_conv3x3_dgrad_x3 if self.conv_x3 else lib.mtl_conv3x3_dgrad

        def conv_dgrad(t, dy, ai, am, w, act, dx, *dims, ao=None):
            if h2:      # ao: slot that receives the bound of dx (the next layer's amax_dy)
                return lib.mtl_conv3x3_dgrad_h2(st, dy, am_(ai, t), am, w, act, dx, am_(ao, t) if ao is not None else None, *dims)
            return dgrad_fn(st, dy, am, w, act, dx, *dims)

        # h2: the weight-gradient kernel's dy loaders also sum dy (bias gradient) and the data-gradient epilogue delivers the bound of
        # its output, so the column-sum passes over dy5 (164 MB) and dp1 (82 MB) are not needed; dp2 keeps its pass (its bound has no
        # other producer)

        def wgrad(t, xa, axi, dy, adi, am, idx, Bq, Tq, Fq, cin, cout, db=None):
            x3 = self.conv_x3
            wsfn = lib.mtl_conv3x3_wgrad_x3_workspace if x3 else lib.mtl_conv3x3_wgrad_workspace
            need = wsfn(Bq, Tq, Fq, cin, cout, 1 if am else 0)
            ws = self.scratch(need)
            if x3 and h2:
                rc = lib.mtl_conv3x3_wgrad_h2(st, xa, am_(axi, t), dy, am_(adi, t), am, g('conv.%d.weight' % idx, t), db, ws, need, Bq, Tq,
                                              Fq, cin, cout)
            else:
                fn = lib.mtl_conv3x3_wgrad_x3 if x3 else lib.mtl_conv3x3_wgrad
                rc = fn(st, xa, dy, am, g('conv.%d.weight' % idx, t), ws, need, Bq, Tq, Fq, cin, cout)
            check(rc, 'wgrad')

        dy5 = self.buf('_dy5', (nt * B, T2, F2, 128))
        dp1 = self.buf('_dp1', (nt * B, T2, F2, 64))
        dy1 = self.buf('_dy1', (nt * B, T, F, 64))
        x3_only = self.conv_x3 and not h2
        merged = (h2 or x3_only) and (nt > 1 or S['meta'].get('widths') is not None)     # data gradients of conv7 / conv5: ONE launch over the samples of all tasks (see forward)
        # the bias gradients of conv5 / conv2 ride on their weight-gradient launches (sums of dy in the loaders): always with h2 (the
        # data-gradient epilogues also deliver the next bound), with the exact split in the several-task launches
        f5 = f2 = h2 or (x3_only and merged)
        xin = S['x']
        # tasks with frame counts of their own (forward): the data gradients leave out the tile rows beyond a task's frames and those
        # rows of their outputs are cleared right behind them -- bias sums, bounds and weight gradients read whole tensors
        widths_b = S['meta'].get('widths')
        skip = widths_b if merged else None
        tails_b = (lambda buf_, T_, row_, shift_: check(lib.mtl_zero_tails(st, buf_.data_ptr(), nt * B, T_, row_, skip, shift_, B), 'mtl_zero_tails')) \
            if skip is not None else (lambda *a_: None)
        AS = 12 * _lib.AMAX_SLOTS
        swd = lambda name: A[name].stride(0) * A[name].element_size() if (sP and nt > 1) else 0      # (bytes)

        def layer7(t, dgrad):
            tw, sl = (t if sP else 0), slice(t * B, (t + 1) * B)
            am2_t = A['am2'][sl].data_ptr()
            self.colsum(dp2[sl].data_ptr(), B * T4 * F4, 128, g('conv.7.bias', t), am_(3, t))
            wgrad(t, y5[sl].data_ptr(), 2, dp2[sl].data_ptr(), 3, am2_t, 7, B, T2, F2, 128, 128)
            if dgrad:
                check(conv_dgrad(t, dp2[sl].data_ptr(), 3, am2_t, A['wd7'][tw].data_ptr(), y5[sl].data_ptr(), dy5[sl].data_ptr(),
                                 B, T2, F2, 128, 128, ao=4 if f5 else None), 'dgrad7')

        def layer5(t, dgrad):
            tw, sl = (t if sP else 0), slice(t * B, (t + 1) * B)
            if not f5:
                self.colsum(dy5[sl].data_ptr(), B * T2 * F2, 128, g('conv.5.bias', t), am_(4, t))
            wgrad(t, p1[sl].data_ptr(), 1, dy5[sl].data_ptr(), 4, None, 5, B, T2, F2, 64, 128, db=g('conv.5.bias', t) if f5 else None)
            if dgrad:
                check(conv_dgrad(t, dy5[sl].data_ptr(), 4, None, A['wd5'][tw].data_ptr(), p1[sl].data_ptr(), dp1[sl].data_ptr(),
                                 B, T2, F2, 64, 128, ao=5 if f2 else None), 'dgrad5')

        def layer2(t, wg=True, w0=True):
            tw, sl = (t if sP else 0), slice(t * B, (t + 1) * B)
            am1_t = A['am1'][sl].data_ptr()
            if not f2:
                self.colsum(dp1[sl].data_ptr(), B * T2 * F2, 64, g('conv.2.bias', t), am_(5, t))
            if wg:
                wgrad(t, y1[sl].data_ptr(), 0, dp1[sl].data_ptr(), 5, am1_t, 2, B, T, F, 64, 64, db=g('conv.2.bias', t) if f2 else None)
            check(conv_dgrad(t, dp1[sl].data_ptr(), 5, am1_t, A['wd2'][tw].data_ptr(), y1[sl].data_ptr(), dy1[sl].data_ptr(),
                             B, T, F, 64, 64), 'dgrad2')
            if w0:
                ws = self.scratch(lib.mtl_conv0_wgrad_workspace())
                check(lib.mtl_conv0_wgrad(st, xin.data_ptr() + 4 * t * sX, dy1[sl].data_ptr(), g('conv.0.weight', t), g('conv.0.bias', t),
                                          ws, B, T, F), 'wgrad0')

        def wgrad_tb(xa, axi, dy, adi, am, idx, Tq, Fq, cin, cout, db):
            """the weight (+ bias) gradients of all tasks of one layer in ONE launch (its partial slabs are dealt to the tasks: as many
            slabs written and reduced as by one single-task launch)"""
            need = lib.mtl_conv3x3_wgrad_x3_workspace(B, Tq, Fq, cin, cout, 1 if am else 0)
            ws = self.scratch(need)
            if x3_only:
                check(lib.mtl_conv3x3_wgrad_x3_tb(st, xa, dy, am, g('conv.%d.weight' % idx), db, ws, need, B, Tq, Fq, cin, cout, nt, sG, sG),
                      'wgrad_tb')
                return
            check(lib.mtl_conv3x3_wgrad_h2_tb(st, xa, am_(axi), dy, am_(adi), am, g('conv.%d.weight' % idx), db, ws, need, B, Tq, Fq, cin, cout,
                                              nt, AS, AS, sG, sG), 'wgrad_tb')

        if merged:
            per = ((lib.mtl_colsum_workspace(B * T4 * F4, 128) // 4 + 3) // 4 * 4) * 4
            check(lib.mtl_colsum_accum_tb(st, dp2.data_ptr(), B * T4 * F4, 128, g('conv.7.bias'), self.scratch(nt * per + 64), am_(3), nt, sG, AS),
                  'colsum_tb')
            wgrad_tb(y5.data_ptr(), 2, dp2.data_ptr(), 3, A['am2'].data_ptr(), 7, T2, F2, 128, 128, None)
            if x3_only:
                check(lib.mtl_conv3x3_dgrad_x3_tb(st, dp2.data_ptr(), A['am2'].data_ptr(), A['wd7'].data_ptr(), y5.data_ptr(), dy5.data_ptr(),
                                                  B, T2, F2, 128, 128, nt, swd('wd7'), skip, 1), 'dgrad7')
            else:
                check(lib.mtl_conv3x3_dgrad_h2_tb(st, dp2.data_ptr(), am_(3), A['am2'].data_ptr(), A['wd7'].data_ptr(), y5.data_ptr(),
                                                  dy5.data_ptr(), am_(4) if f5 else None, B, T2, F2, 128, 128, nt, swd('wd7'), AS, AS, skip, 1),
                      'dgrad7')
            tails_b(dy5, T2, F2 * 128, 1)
            wgrad_tb(p1.data_ptr(), 1, dy5.data_ptr(), 4, None, 5, T2, F2, 64, 128, g('conv.5.bias'))
            if x3_only:
                check(lib.mtl_conv3x3_dgrad_x3_tb(st, dy5.data_ptr(), None, A['wd5'].data_ptr(), p1.data_ptr(), dp1.data_ptr(),
                                                  B, T2, F2, 64, 128, nt, swd('wd5'), skip, 1), 'dgrad5')
            else:
                check(lib.mtl_conv3x3_dgrad_h2_tb(st, dy5.data_ptr(), am_(4), None, A['wd5'].data_ptr(), p1.data_ptr(), dp1.data_ptr(),
                                                  am_(5) if f2 else None, B, T2, F2, 64, 128, nt, swd('wd5'), AS, AS, skip, 1), 'dgrad5')
            tails_b(dp1, T2, F2 * 64, 1)
            wgrad_tb(y1.data_ptr(), 0, dp1.data_ptr(), 5, A['am1'].data_ptr(), 2, T, F, 64, 64, g('conv.2.bias'))
            # conv2's data gradient: one launch for all tasks too (its matrix kernel covers the even part of the 161-bin axis, the last
            # column goes to the edge kernel: csrc/mtl_mfma.hip conv_dgrad_edge_kernel)
            if x3_only:
                check(lib.mtl_conv3x3_dgrad_x3_tb(st, dp1.data_ptr(), A['am1'].data_ptr(), A['wd2'].data_ptr(), y1.data_ptr(), dy1.data_ptr(),
                                                  B, T, F, 64, 64, nt, swd('wd2'), skip, 0), 'dgrad2')
            else:
                check(lib.mtl_conv3x3_dgrad_h2_tb(st, dp1.data_ptr(), am_(5), A['am1'].data_ptr(), A['wd2'].data_ptr(), y1.data_ptr(),
                                                  dy1.data_ptr(), None, B, T, F, 64, 64, nt, swd('wd2'), AS, 0, skip, 0), 'dgrad2')
            tails_b(dy1, T, F * 64, 0)
            ws = self.scratch(lib.mtl_conv0_wgrad_workspace())
            check(lib.mtl_conv0_wgrad_tb(st, xin.data_ptr(), dy1.data_ptr(), g('conv.0.weight'), g('conv.0.bias'), ws, B, T, F, nt, sX, sG, sG),
                  'wgrad0_tb')
        else:
            for t in range(nt):
                layer7(t, True)
                layer5(t, True)
                layer2(t)
        if h2 and self.census is not None:
            for slot, t_ in ((3, dp2), (4, dy5), (5, dp1)):
                self._census(slot, t_, am_(slot))
        self.join_side()
        self.flush_ln_reduce()     # the parameter / bias gradients of all 17 LayerNorms of the pass: one launch (after the join: one of
                                   # the 17 backward kernels ran on the side stream)
        self._slice_done('conv')
        self.nt, self.sP, self.sG = 1, 0, 0
