"""LM meta-transfer path on the MI355X (BASELINE.json configs[4]): drop-in pieces for `lm/model/rnn_model.py`,
`lm/util/data.py` and the training loop of `lm/main_meta_transfer.py:277-411`.

* `RNNModel('LSTM', ntoken, ninp, nhid, nlayers, dropout)`: same constructor, parameter names (`encoder.weight`,
  `rnn.weight_ih_l0`, ..., `decoder.bias`), `init_weights` and RNG draw order as the reference (bit-identical initialisation); all
  parameters are views into ONE flat fp32 buffer, the compute runs in libmtl_hip.so (`LMEngine`): embedding gather + Philox dropout,
  per layer ONE GEMM for the input contributions of all T steps, per step the recurrent GEMM (small-product engine) + the fused
  LSTM cell kernel, vocabulary projection + fused cross-entropy; hand-written backward (BPTT over the bptt window), weight
  gradients as three products over all T steps with the bias gradients folded in.
* `LMDataset`: `batchify` / `get_batch` / `sample(manifest_id, i)` with the reference's offsets.
* `LMMetaTrainer`: the meta step.  The reference's loop cannot run on torch >= 2 (it back-propagates through parameter storage it
  has overwritten, see oracle/lm_refimpl.py), so the semantics implemented are the documented first-order reading: per task
  train pass at theta0 (hidden state carried, detached), clipped inner SGD step lr / meta_lr_factor, validation pass at theta', G =
  sum_i w_i grad val_i (w = (1-ratio)/(n-1), ..., ratio), clipped, plain SGD outer step.  Tasks are sharded over ranks like the ASR
  loop with ONE all-reduce of the flat G.  **Parity unpinned** against the reference loop; pinned against the oracle restatement.
No CPU fallback: without a GPU / the built library every compute call raises.
"""
import ctypes
import math
import time

import torch
import torch.nn as nn

from . import _lib, dist as mdist
from .engine import ParamLayout

check = _lib.check
ACCUM = 2



def synth_corpus(seed, ntoken, length):
    """Zipf-ish synthetic token stream for benchmarks (the reference's ./data/*.txt corpora do not ship with it)."""
    g = torch.Generator().manual_seed(seed)
    u = torch.rand(length, generator=g)
    return (torch.floor((ntoken - 1) * u * u)).to(torch.int64).clamp_(0, ntoken - 1)


class RNNModel(nn.Module):
    def __init__(self, rnn_type, ntoken, ninp, nhid, nlayers, dropout=0.5, tie_weights=False):
        super().__init__()
        if rnn_type != 'LSTM' or tie_weights:
            raise NotImplementedError("accelerated LM path: rnn_type='LSTM', untied weights")
        self.drop = nn.Dropout(dropout)
        self.encoder = nn.Embedding(ntoken, ninp)
        self.rnn = nn.LSTM(ninp, nhid, nlayers, dropout=dropout)            # parameter container only (same init draws)
        self.decoder = nn.Linear(nhid, ntoken)
        self.rnn_type, self.ntoken, self.ninp, self.nhid, self.nlayers, self.dropout_rate = rnn_type, ntoken, ninp, nhid, nlayers, dropout
        self.init_weights()
        self._layout = ParamLayout([(n, p.shape) for n, p in self.named_parameters()])
        self.engine = None
        self._flatten()

    def init_weights(self):
        initrange = 0.1
        self.encoder.weight.data.uniform_(-initrange, initrange)
        self.decoder.bias.data.fill_(0)
        self.decoder.weight.data.uniform_(-initrange, initrange)

    def _flatten(self):
        params = list(self.named_parameters())
        device = params[0][1].device
        theta = torch.zeros(self._layout.total, dtype=torch.float32, device=device)
        gflat = torch.zeros_like(theta)
        for name, p in params:
            v = self._layout.view(theta, name)
            v.copy_(p.data)
            p.data = v
            p.grad = self._layout.view(gflat, name)
        self._theta, self._gflat = theta, gflat
        self.engine = LMEngine(self, device) if device.type == 'cuda' else None

    def _apply(self, fn, *a, **kw):
        out = super()._apply(fn, *a, **kw)
        self._flatten()
        return out

    @property
    def flat_parameters(self):
        return self._theta

    @property
    def flat_grad(self):
        return self._gflat

    def init_hidden(self, bsz):
        dev = self._theta.device
        return (torch.zeros(self.nlayers, bsz, self.nhid, device=dev), torch.zeros(self.nlayers, bsz, self.nhid, device=dev))

    def _need_engine(self):
        if self.engine is None:
            raise RuntimeError('the LM lives on %s: the product path needs an MI355X (call .cuda()); there is no CPU fallback'
                               % self._theta.device)
        return self.engine

    def forward(self, input, hidden):
        """-> (decoded (T, B, ntoken), hidden) like lm/model/rnn_model.py:52-60 (no autograd graph: use LMEngine.backward)"""
        eng = self._need_engine()
        out = eng.forward(self._theta, input, None, hidden, self.dropout_rate if self.training else 0.0)
        eng.check_handoff()          # the caller consumes the logits: a timed-out persistent launch must not pass as a result
        return out['logits'].view(input.shape[0], input.shape[1], self.ntoken), out['hidden']


class LMEngine:
    """Forward + hand-written backward of the LSTM LM as library calls on the current stream (rows are ordered (t, b))."""

    def __init__(self, model, device):
        self.m, self.device, self.lib, self.L = model, device, _lib.lib(), model._layout
        self.pool, self.saved = {}, None
        self.ws = torch.empty(4 << 20, dtype=torch.float32, device=device)
        # all T steps of a layer in ONE launch per direction (csrc/mtl_lstm.hip) where the shape is supported; False keeps the per-step
        # recurrent product + cell kernel (tests compare the two; unsupported shapes take it anyway)
        self.persistent = True
        # the whole stack as one wavefront launch per direction (layers one step apart); False: one launch per layer and direction
        self.stacked = True
        self.sync_ws = torch.zeros(int(self.lib.mtl_lstm_layer_workspace()) // 4, dtype=torch.int32, device=device)
        self._persistent_issued = False

    def check_handoff(self):
        """The persistent LSTM kernels bound every grid-wide wait and set a sticky error word instead of hanging the device (e.g.
        when their workgroups could not all become resident).  Called wherever results leave the engine: RNNModel.forward
        (evaluation / inference), LMMetaTrainer.run_iteration, and by direct LMEngine users through results().  One 4-byte
        read-back (a host sync) -- skipped when no persistent launch has been issued since the last check."""
        if not self._persistent_issued:
            return
        self._persistent_issued = False
        if int(self.sync_ws[1]) != 0:
            self.sync_ws[1] = 0
            raise RuntimeError('mtl_lstm: a grid-wide wait of a persistent LSTM launch timed out; its results are invalid '
                               '(MTL_LSTM_STACK=0 / MTL_LSTM_PERSISTENT=0 select the per-layer / per-step launches)')

    def buf(self, name, shape, dtype=torch.float32):
        key = (name, tuple(int(v) for v in shape), dtype)
        t = self.pool.get(key)
        if t is None:
            t = torch.empty(key[1], dtype=dtype, device=self.device)
            self.pool[key] = t
        return t

    @property
    def stream(self):
        return torch.cuda.current_stream(self.device).cuda_stream

    def gemm(self, ta, tb, M, N, K, A, lda, B, ldb, C, ldc, bias=None, flags=0, rowsum=None):
        check(self.lib.mtl_gemm_f32_ex(self.stream, ta, tb, M, N, K, 1.0, A, lda, B, ldb, C, ldc, bias, None, 0, flags, 1, 1, 0, 0, 0, 0,
                                       0, 0, 0, 1, 0, 0, rowsum, 0, self.ws.data_ptr(), self.ws.numel() * 4, 0, 0), 'mtl_gemm_f32_ex')

    def _chains(self, ids_host):
        """occurrence chains for the deterministic embedding scatter-add (mtl_embed_bwd); runs on the device `ids_host` lives on"""
        flat = ids_host.reshape(-1)
        order = torch.argsort(flat, stable=True)
        srt = flat[order]
        same = torch.zeros_like(srt, dtype=torch.bool)
        same[1:] = srt[1:] == srt[:-1]
        first = torch.empty_like(flat, dtype=torch.int32)
        first[order] = (~same).to(torch.int32)
        nxt = torch.full_like(flat, -1, dtype=torch.int32)
        nxt[order[:-1]] = torch.where(same[1:], order[1:], torch.full_like(order[1:], -1)).to(torch.int32)
        return torch.stack([first, nxt])

    def forward(self, theta, x, y, hidden, dropout_p=0.0):
        """x (T, B) int64, y (T*B) int64 or None, hidden (h, c) each (L, B, H).  -> dict(logits (T*B, V), loss (1,) or None,
        hidden (detached new state))."""
        m, lib, st, Lo = self.m, self.lib, self.stream, self.L
        T, B = int(x.shape[0]), int(x.shape[1])
        R, H, E, V, NL = T * B, m.nhid, m.ninp, m.ntoken, m.nlayers
        P = theta.data_ptr()
        o = lambda n: P + 4 * Lo.off(n)
        # token ids and the occurrence chains of the embedding scatter-add: where the batch already lives (a device batch is indexed
        # with device ops -- a read-back here would drain the stream once per pass and leave the GPU idle while the host enqueues)
        xh = x.detach().to(torch.int64).contiguous()
        ids = self.buf('ids', (R,), torch.int64)
        ids.copy_(xh.reshape(-1), non_blocking=True)
        chains = self.buf('chains', (2, R), torch.int32)
        chains.copy_(self._chains(xh), non_blocking=True)
        sc = 1.0 / (1.0 - dropout_p) if dropout_p > 0 else 1.0
        seed = self.buf('seed', (1,), torch.int64)
        if dropout_p > 0:
            # (drawn from the generator of the batch's device: a pageable host tensor would make the copy wait for the stream)
            seed.copy_(torch.randint(0, 2 ** 62, (1,), dtype=torch.int64, device=xh.device), non_blocking=True)

        def mask(name, n, site):
            if dropout_p <= 0:
                return None
            t = self.buf(name, (n,), torch.uint8)
            check(lib.mtl_dropout_mask(st, t.data_ptr(), n, float(dropout_p), seed.data_ptr(), site << 40), 'mtl_dropout_mask')
            return t

        zpe = self.buf('zero_pe', (R, E))
        if not getattr(self, '_zpe_ok', None) == (R, E):
            zpe.zero_()
            self._zpe_ok = (R, E)
        # with E == H the inputs of the weight-gradient products -- per layer x_l (R x H) and the previous states hall_l[:T] (R x H)
        # -- live in ONE arena [layer][x, hall] with constant strides, and the gate gradients in one [layer] stack: backward() then
        # forms all 2 NL weight gradients (+ bias gradients) with a single batched product instead of 2 NL small ones
        arena = self.buf('wg_arena', (NL, 2, (T + 1) * B * H)) if (E == H and NL > 1) else None
        emb = arena[0, 0, :R * E].view(R, E) if arena is not None else self.buf('emb', (R, E))
        m_emb = mask('m_emb', R * E, 1)
        check(lib.mtl_embed_pe_fwd(st, ids.data_ptr(), o('encoder.weight'), zpe.data_ptr(), emb.data_ptr(), R, R, E,
                                   m_emb.data_ptr() if m_emb is not None else None, sc), 'embed')
        h0, c0 = hidden
        layers = []
        xin, kin = emb, E
        hn, cn = self.buf('hn', (NL, B, H)), self.buf('cn', (NL, B, H))
        stacked = self.persistent and self.stacked and NL > 1 and bool(lib.mtl_lstm_stack_supported(B, H, NL))
        desc = _lib.LstmStack() if stacked else None
        for l in range(NL):
            hall = arena[l, 1].view(T + 1, B, H) if arena is not None else self.buf('hall%d' % l, (T + 1, B, H))
            call = self.buf('call%d' % l, (T + 1, B, H))
            hall[0].copy_(h0[l])
            call[0].copy_(c0[l])
            acts = self.buf('acts%d' % l, (R, 4 * H))
            # this layer's output after its dropout (next layer's / decoder's input)
            xout = arena[l + 1, 0, :R * H].view(R, H) if (arena is not None and l + 1 < NL) else self.buf('xout%d' % l, (R, H))
            msk = mask('m_l%d' % l, R * H, 2 + l)
            whh, bhh = o('rnn.weight_hh_l%d' % l), o('rnn.bias_hh_l%d' % l)
            layers.append(dict(x=xin, kin=kin, hall=hall, call=call, acts=acts, mask=msk))
            if stacked:
                # every layer in ONE wavefront launch below: only layer 0 takes its input contributions from a product over all steps
                desc.w_ih[l], desc.b_ih[l], desc.w_hh[l], desc.b_hh[l] = o('rnn.weight_ih_l%d' % l), o('rnn.bias_ih_l%d' % l), whh, bhh
                desc.hall[l], desc.call[l], desc.acts[l], desc.xout[l] = hall.data_ptr(), call.data_ptr(), acts.data_ptr(), xout.data_ptr()
                desc.mask[l] = msk.data_ptr() if msk is not None else None
                desc.gx[l] = self.buf('gx%d' % l, (R, 4 * H)).data_ptr()
                xin, kin = xout, H
                continue
            gx = self.buf('gx%d' % l, (R, 4 * H))
            self.gemm(0, 1, R, 4 * H, kin, xin.data_ptr(), kin, o('rnn.weight_ih_l%d' % l), kin, gx.data_ptr(), 4 * H, bias=o('rnn.bias_ih_l%d' % l))
            gh = self.buf('gh', (B, 4 * H))
            fused = self.persistent and bool(lib.mtl_lstm_layer_supported(B, H))
            if fused:
                self._persistent_issued = True
                check(lib.mtl_lstm_layer_fwd(st, gx.data_ptr(), whh, bhh, hall.data_ptr(), call.data_ptr(), acts.data_ptr(), xout.data_ptr(),
                                             msk.data_ptr() if msk is not None else None, sc, T, B, H, self.sync_ws.data_ptr()),
                      'lstm_layer_fwd')
            for t in range(0 if fused else T):
                self.gemm(0, 1, B, 4 * H, H, hall[t].data_ptr(), H, whh, H, gh.data_ptr(), 4 * H, bias=bhh)
                check(lib.mtl_lstm_cell_fwd(st, gx.data_ptr() + 16 * t * B * H, gh.data_ptr(), call[t].data_ptr(),
                                            acts.data_ptr() + 16 * t * B * H, call[t + 1].data_ptr(), hall[t + 1].data_ptr(),
                                            xout.data_ptr() + 4 * t * B * H, msk.data_ptr() + t * B * H if msk is not None else None, sc,
                                            B, H), 'lstm_cell_fwd')
            xin, kin = xout, H
        if stacked:
            gx = self.buf('gx0', (R, 4 * H))
            self.gemm(0, 1, R, 4 * H, E, emb.data_ptr(), E, o('rnn.weight_ih_l0'), E, gx.data_ptr(), 4 * H, bias=o('rnn.bias_ih_l0'))
            self._persistent_issued = True
            check(lib.mtl_lstm_stack_fwd(st, ctypes.byref(desc), sc, T, B, H, NL, self.sync_ws.data_ptr()), 'lstm_stack_fwd')
        for l in range(NL):
            hn[l].copy_(layers[l]['hall'][T])
            cn[l].copy_(layers[l]['call'][T])
        logits = self.buf('logits', (R, V))
        self.gemm(0, 1, R, V, H, xin.data_ptr(), H, o('decoder.weight'), H, logits.data_ptr(), V, bias=o('decoder.bias'))
        loss = None
        if y is not None:
            gold = self.buf('gold', (R,), torch.int64)
            gold.copy_(y.detach().reshape(-1).to(torch.int64), non_blocking=True)
            lse, hyp, rowloss, loss = self.buf('lse', (R,)), self.buf('hyp', (R,), torch.int64), self.buf('rowloss', (R,)), self.buf('loss', (1,))
            check(lib.mtl_ce_argmax_fwd(st, logits.data_ptr(), gold.data_ptr(), R, V, V, -1, 0.0, R, None, lse.data_ptr(), hyp.data_ptr(),
                                        rowloss.data_ptr(), loss.data_ptr()), 'ce_fwd')     # nn.CrossEntropyLoss(): mean over all T*B rows
        self.saved = dict(theta=theta, T=T, B=B, layers=layers, last=xin, m_emb=m_emb, sc=sc, ids=ids, chains=chains, stacked=stacked, arena=arena)
        return dict(logits=logits, loss=loss, hidden=(hn.clone(), cn.clone()))

    def backward(self, grad, scale=1.0):
        """grad (flat) += scale * dLoss/dtheta of the last forward (truncated BPTT over the window: no gradient into the incoming
        hidden state, which the reference detaches with repackage_hidden)."""
        S = self.saved
        if S is None:
            raise RuntimeError('backward() without a preceding forward() with targets')
        m, lib, st, Lo = self.m, self.lib, self.stream, self.L
        T, B, sc = S['T'], S['B'], S['sc']
        R, H, E, V, NL = T * B, m.nhid, m.ninp, m.ntoken, m.nlayers
        P, G = S['theta'].data_ptr(), grad.data_ptr()
        o = lambda n: P + 4 * Lo.off(n)
        g = lambda n: G + 4 * Lo.off(n)
        ldd = (V + 3) // 4 * 4
        dlog = self.buf('dlog', (R, ldd))
        check(lib.mtl_ce_bwd(st, self.pool[('logits', (R, V), torch.float32)].data_ptr(), self.pool[('lse', (R,), torch.float32)].data_ptr(),
                             self.pool[('gold', (R,), torch.int64)].data_ptr(), R, V, V, -1, 0.0, float(scale) / R, None, dlog.data_ptr(), ldd),
              'ce_bwd')
        last = S['last']
        self.gemm(1, 0, V, H, R, dlog.data_ptr(), ldd, last.data_ptr(), H, g('decoder.weight'), H, flags=ACCUM, rowsum=g('decoder.bias'))
        dx = self.buf('dx_out', (R, H))
        self.gemm(0, 0, R, H, V, dlog.data_ptr(), ldd, o('decoder.weight'), H, dx.data_ptr(), H)
        stacked, arena = S['stacked'], S['arena']
        dGs = self.buf('dG_all', (NL, R, 4 * H))
        if stacked:
            desc = _lib.LstmStack()
            for l in range(NL):
                Ly = S['layers'][l]
                desc.w_ih[l], desc.w_hh[l] = o('rnn.weight_ih_l%d' % l), o('rnn.weight_hh_l%d' % l)
                desc.call[l], desc.acts[l] = Ly['call'].data_ptr(), Ly['acts'].data_ptr()
                desc.dG[l] = dGs[l].data_ptr()
                desc.mask[l] = Ly['mask'].data_ptr() if Ly['mask'] is not None else None
            scratch = self.buf('stack_scratch', (int(lib.mtl_lstm_stack_scratch(T, B, H, NL)) // 4,))
            self._persistent_issued = True
            check(lib.mtl_lstm_stack_bwd(st, ctypes.byref(desc), dx.data_ptr(), sc, scratch.data_ptr(), T, B, H, NL, self.sync_ws.data_ptr()),
                  'lstm_stack_bwd')
        for l in reversed(range(NL)):
            Ly = S['layers'][l]
            kin = Ly['kin']
            dG = dGs[l]
            dh_rec, dc = self.buf('dh_rec', (B, H)), [self.buf('dc_a', (B, H)), self.buf('dc_b', (B, H))]
            whh = o('rnn.weight_hh_l%d' % l)
            msk = Ly['mask']
            fused = stacked or (self.persistent and bool(lib.mtl_lstm_layer_supported(B, H)))
            if fused and not stacked:
                self._persistent_issued = True
                check(lib.mtl_lstm_layer_bwd(st, dx.data_ptr(), msk.data_ptr() if msk is not None else None, sc, whh, Ly['acts'].data_ptr(),
                                             Ly['call'].data_ptr(), dG.data_ptr(), T, B, H, self.sync_ws.data_ptr()), 'lstm_layer_bwd')
            for t in reversed(range(0 if fused else T)):
                first = t == T - 1
                check(lib.mtl_lstm_cell_bwd(st, dx.data_ptr() + 4 * t * B * H, msk.data_ptr() + t * B * H if msk is not None else None, sc,
                                            None if first else dh_rec.data_ptr(), None if first else dc[(t + 1) & 1].data_ptr(),
                                            Ly['acts'].data_ptr() + 16 * t * B * H, Ly['call'][t + 1].data_ptr(), Ly['call'][t].data_ptr(),
                                            dG.data_ptr() + 16 * t * B * H, dc[t & 1].data_ptr(), B, H), 'lstm_cell_bwd')
                if t > 0:
                    self.gemm(0, 0, B, H, 4 * H, dG.data_ptr() + 16 * t * B * H, 4 * H, whh, H, dh_rec.data_ptr(), H)
            # parameter gradients over all T steps at once; both bias gradients are colsum(dG)
            if arena is None:
                self.gemm(1, 0, 4 * H, H, R, dG.data_ptr(), 4 * H, Ly['hall'].data_ptr(), H, g('rnn.weight_hh_l%d' % l), H, flags=ACCUM,
                          rowsum=g('rnn.bias_hh_l%d' % l))
                self.gemm(1, 0, 4 * H, kin, R, dG.data_ptr(), 4 * H, Ly['x'].data_ptr(), kin, g('rnn.weight_ih_l%d' % l), kin, flags=ACCUM,
                          rowsum=g('rnn.bias_ih_l%d' % l))
            elif l == 0:
                # all layers' (weight_ih, weight_hh, bias_ih, bias_hh) gradients as ONE product batched over [layer][ih, hh]: A = dG_l
                # (shared by the pair), B = arena[l][x_l | hall_l], C / row sums at the parameters' own offsets (constant strides
                # because every layer has input width H here)
                po = lambda n: Lo.off(n)
                lay = po('rnn.weight_ih_l1') - po('rnn.weight_ih_l0')
                assert all(po('rnn.%s_l%d' % (n, q)) - po('rnn.%s_l0' % n) == q * lay for q in range(NL)
                           for n in ('weight_ih', 'weight_hh', 'bias_ih', 'bias_hh'))
                half = (T + 1) * B * H
                check(lib.mtl_gemm_f32_ex(st, 1, 0, 4 * H, H, R, 1.0, dGs.data_ptr(), 4 * H, arena.data_ptr(), H, g('rnn.weight_ih_l0'), H,
                                          None, None, 0, ACCUM, 2 * NL, 2, R * 4 * H, 0, 2 * half, half, lay,
                                          po('rnn.weight_hh_l0') - po('rnn.weight_ih_l0'), 0, 1, 0, 0, g('rnn.bias_ih_l0'), lay,
                                          self.ws.data_ptr(), self.ws.numel() * 4, 0, po('rnn.bias_hh_l0') - po('rnn.bias_ih_l0')),
                      'lstm weight gradients')
            if stacked and l > 0:
                continue                                          # the stack kernel has handed the input gradient down itself
            dxin = self.buf('dx_in%d' % l, (R, kin))
            self.gemm(0, 0, R, kin, 4 * H, dG.data_ptr(), 4 * H, o('rnn.weight_ih_l%d' % l), kin, dxin.data_ptr(), kin)
            dx = dxin
        m_emb = S['m_emb']
        check(lib.mtl_embed_bwd(st, S['ids'].data_ptr(), S['chains'][0].data_ptr(), S['chains'][1].data_ptr(), dx.data_ptr(),
                                g('encoder.weight'), R, E, -1, m_emb.data_ptr() if m_emb is not None else None, sc), 'embed_bwd')


class LMDataset(object):
    """lm/util/data.py:12-67"""

    def __init__(self, task_list, args):
        self.bptt, self.batch_size, self.args = args.bptt, args.batch_size, args
        self.task_list = [self.batchify(t, self.batch_size) for t in task_list]

    def batchify(self, data, bsz):
        nbatch = data.size(0) // bsz
        data = data.narrow(0, 0, nbatch * bsz).view(bsz, -1).t().contiguous()
        return data.cuda() if getattr(self.args, 'cuda', False) else data

    def get_batch(self, source, i, evaluation=False):
        seq_len = min(self.bptt, len(source) - 1 - i)
        return source[i:i + seq_len], source[i + 1:i + 1 + seq_len].reshape(-1)

    def sample(self, manifest_id, i):
        ids = self.task_list[manifest_id]
        tr_ids = ((i * self.bptt) % len(ids)) - (((i * self.bptt) % len(ids)) % self.bptt)
        val_ids = (((i + 1) * self.bptt) % len(ids)) - ((((i + 1) * self.bptt) % len(ids)) % self.bptt)
        return self.get_batch(ids, tr_ids) + self.get_batch(ids, val_ids)


def task_weights(n, ratio):
    """lm/main_meta_transfer.py:346-349 for n = 3; generalised: the LAST task (the target corpus) weighs `ratio`"""
    return [ratio if i == n - 1 else (1.0 - ratio) / max(n - 1, 1) for i in range(n)]


class LMMetaTrainer:
    """The meta step of lm/main_meta_transfer.py:277-411 in its documented first-order reading (module docstring)."""

    def __init__(self, model, lr=20.0, meta_lr_factor=3.0, clip=0.25, ratio=0.8):
        self.model, self.lr, self.meta_lr_factor, self.clip, self.ratio = model, lr, meta_lr_factor, clip, ratio
        th = model.flat_parameters
        self.g, self.G, self.theta1 = torch.zeros_like(th), torch.zeros_like(th), torch.empty_like(th)
        self.coef = torch.empty(1, dtype=torch.float32, device=th.device)
        self.clip_ws = torch.empty(2048, dtype=torch.float32, device=th.device)
        self.hidden = None

    def _clip(self, grad):
        lib, st = _lib.lib(), torch.cuda.current_stream(grad.device).cuda_stream
        check(lib.mtl_sumsq(st, grad.data_ptr(), grad.numel(), self.coef.data_ptr(), self.clip_ws.data_ptr(), 2, float(self.clip)), 'mtl_sumsq')
        check(lib.mtl_scale(st, grad.data_ptr(), 1.0, self.coef.data_ptr(), grad.numel()), 'mtl_scale')

    def run_iteration(self, task_batches, val_batch, n_tasks=None, task_ids=None):
        """task_batches: this rank's [(x (T,B), y (T*B))] with their GLOBAL task ids; val_batch: (x, y).  One all-reduce of G.
        -> (weighted validation loss, [train losses])   (single device sync at the end)"""
        m = self.model
        eng = m._need_engine()
        lib, dev = _lib.lib(), m.flat_parameters.device
        st = lambda: torch.cuda.current_stream(dev).cuda_stream
        n = n_tasks or len(task_batches)
        task_ids = list(range(len(task_batches))) if task_ids is None else task_ids
        w = task_weights(n, self.ratio)
        p = m.dropout_rate if m.training else 0.0
        theta0 = m.flat_parameters
        if self.hidden is None:
            self.hidden = m.init_hidden(task_batches[0][0].shape[1])
        self.G.zero_()
        tr_losses, val_losses = [], []
        for (x, y), tid in zip(task_batches, task_ids):
            self.g.zero_()
            out = eng.forward(theta0, x, y, self.hidden, p)                                   # meta-train forward (:322)
            tr_losses.append(out['loss'].clone())
            self.hidden = out['hidden']
            eng.backward(self.g, 1.0)
            if self.clip:
                self._clip(self.g)                                                            # (:335-336)
            check(lib.mtl_sgd_theta_prime(st(), theta0.data_ptr(), self.g.data_ptr(), self.lr / self.meta_lr_factor,
                                          self.theta1.data_ptr(), theta0.numel()), 'sgd')     # inner step (:337-338)
            out = eng.forward(self.theta1, val_batch[0], val_batch[1], self.hidden, p)        # meta-validation at theta' (:341)
            val_losses.append(out['loss'].clone())
            eng.backward(self.G, w[tid])                                                      # G += w_i * grad val_i
        mdist.allreduce_sum_(self.G)
        if self.clip:
            self._clip(self.G)                                                                # (:366-367)
        check(lib.mtl_axpy(st(), theta0.data_ptr(), self.G.data_ptr(), -float(self.lr), theta0.numel()), 'outer sgd')   # (:368)
        torch.cuda.synchronize(dev)
        eng.check_handoff()
        batch_loss = sum(w[tid] * float(v) for v, tid in zip(val_losses, task_ids))
        return batch_loss, [float(t) for t in tr_losses]

    def train(self, dataset, start_it, num_it, log_interval=200):
        """loop of lm/main_meta_transfer.py:277-411 without the periodic evaluation: per iteration every task's train batch
        `sample(i, it)`, the shared validation batch `sample(-1, it)`; tasks sharded over ranks"""
        rank, world = mdist.rank(), mdist.world_size()
        n = len(dataset.task_list)
        mine = mdist.shard_tasks(n, rank, world)
        dev = self.model.flat_parameters.device
        total, t0 = 0.0, time.time()
        self.model.train()
        for it in range(start_it, num_it):
            _, _, vx, vy = dataset.sample(-1, it)
            batches = [dataset.sample(i, it)[:2] for i in mine]
            loss, _ = self.run_iteration([(x.to(dev), y.to(dev)) for x, y in batches], (vx.to(dev), vy.to(dev)), n, mine)
            total += mdist.allreduce_scalars([loss], dev)[0]
            if it % log_interval == 0 and it > 0 and rank == 0:
                cur = total / log_interval
                print('| it {:3d} | lr {:02.2f} | ms/batch {:5.2f} | word_loss {:5.2f} | avg ppl {:8.2f}'.format(
                    it, self.lr, (time.time() - t0) * 1000 / log_interval, cur, math.exp(min(cur, 50.0))))
                total, t0 = 0.0, time.time()
