"""Drop-in model for the reference's `models/asr/transformer.py:Transformer` (feat_extractor='vgg_cnn') whose
compute runs in libmtl_hip.so.

The nn.Module tree below is a PARAMETER CONTAINER: same attribute names, registration order, constructors and
initialisers as the reference (so `state_dict()` keys, `parameters()` order and the RNG draw order of the
initialisation are identical -- SURVEY.md Q4), but no module has a torch forward.  All parameters are views into
one flat fp32 buffer; `Transformer.forward` hands that buffer to `engine.PassEngine`.

Reference symbols mirrored: Transformer (models/asr/transformer.py:14-240), Encoder/EncoderLayer
(modules/encoder.py:15-106), Decoder/DecoderLayer (modules/decoder.py:14-115,293-323),
FactorizedMultiHeadAttention / PositionwiseFeedForward / PositionalEncoding (modules/common_layers.py:86-132,238-306).
"""
import math
import os

import numpy as np
import torch
import torch.nn as nn

from . import _lib
from .engine import PassEngine, ParamLayout, Hyper

check = _lib.check


class PositionalEncoding(nn.Module):
    """sin/cos table buffer `pe` (1, max_length, dim_model); modules/common_layers.py:86-108."""

    def __init__(self, dim_model, max_length=2000):
        super().__init__()
        pos = torch.arange(0, max_length).unsqueeze(1).float()
        freq = torch.exp(torch.arange(0, dim_model, 2).float() * -(math.log(10000.0) / dim_model))
        table = torch.zeros(max_length, dim_model)
        table[:, 0::2] = torch.sin(pos * freq)
        table[:, 1::2] = torch.cos(pos * freq)
        self.register_buffer('pe', table.unsqueeze(0))


class FactorizedMultiHeadAttention(nn.Module):
    """Rank-r Q/K/V/O projections (modules/common_layers.py:238-274).  Initialiser calls are kept (they are all
    overwritten by the final xavier sweep) because they consume the RNG stream the 1-D parameters depend on."""

    def __init__(self, num_heads, dim_model, dim_key, dim_value, dropout=0.1, r=100):
        super().__init__()
        self.num_heads, self.dim_model, self.dim_key, self.dim_value, self.r = num_heads, dim_model, dim_key, dim_value, r
        for name, width in (('query', dim_key), ('key', dim_key), ('value', dim_value)):
            setattr(self, name + '_linear_a', nn.Linear(dim_model, r, bias=False))
            setattr(self, name + '_linear_b', nn.Linear(r, num_heads * width))
        for name, width in (('query', dim_key), ('key', dim_key), ('value', dim_value)):
            for half in ('_linear_a', '_linear_b'):
                nn.init.normal_(getattr(self, name + half).weight, mean=0, std=np.sqrt(2.0 / (dim_model + width)))
        self.layer_norm = nn.LayerNorm(dim_model)
        self.output_linear_a = nn.Linear(num_heads * dim_value, r, bias=False)
        self.output_linear_b = nn.Linear(r, dim_model)
        nn.init.xavier_normal_(self.output_linear_a.weight)
        nn.init.xavier_normal_(self.output_linear_b.weight)


class PositionwiseFeedForward(nn.Module):
    def __init__(self, dim_model, dim_ff, dropout=0.1):
        super().__init__()
        self.linear_1 = nn.Linear(dim_model, dim_ff)
        self.linear_2 = nn.Linear(dim_ff, dim_model)
        self.layer_norm = nn.LayerNorm(dim_model)


class EncoderLayer(nn.Module):
    def __init__(self, num_heads, dim_model, dim_inner, dim_key, dim_value, dropout=0.1, is_factorized=False, r=100):
        super().__init__()
        self.self_attn = FactorizedMultiHeadAttention(num_heads, dim_model, dim_key, dim_value, dropout=dropout, r=r)
        self.pos_ffn = PositionwiseFeedForward(dim_model, dim_inner, dropout=dropout)


class Encoder(nn.Module):
    def __init__(self, num_layers, num_heads, dim_model, dim_key, dim_value, dim_input, dim_inner, dropout=0.1,
                 src_max_length=2500, is_factorized=False, r=100):
        super().__init__()
        if is_factorized:
            raise NotImplementedError('--is-factorized (factorized FFN / input projection) is outside the accelerated path')
        self.num_layers, self.num_heads, self.dim_model = num_layers, num_heads, dim_model
        self.dim_key, self.dim_value, self.dim_input, self.dim_inner = dim_key, dim_value, dim_input, dim_inner
        self.src_max_length, self.dropout_rate, self.r = src_max_length, dropout, r
        self.input_linear = nn.Linear(dim_input, dim_model)
        self.layer_norm_input = nn.LayerNorm(dim_model)
        self.positional_encoding = PositionalEncoding(dim_model, src_max_length)
        self.layers = nn.ModuleList([EncoderLayer(num_heads, dim_model, dim_inner, dim_key, dim_value, dropout=dropout, r=r)
                                     for _ in range(num_layers)])


class DecoderLayer(nn.Module):
    def __init__(self, dim_model, dim_inner, num_heads, dim_key, dim_value, dropout=0.1, is_factorized=False, r=100):
        super().__init__()
        self.self_attn = FactorizedMultiHeadAttention(num_heads, dim_model, dim_key, dim_value, dropout=dropout, r=r)
        self.encoder_attn = FactorizedMultiHeadAttention(num_heads, dim_model, dim_key, dim_value, dropout=dropout, r=r)
        self.pos_ffn = PositionwiseFeedForward(dim_model, dim_inner, dropout=dropout)


class Decoder(nn.Module):
    def __init__(self, vocab, num_layers, num_heads, dim_emb, dim_model, dim_inner, dim_key, dim_value, dropout=0.1,
                 trg_max_length=1000, emb_trg_sharing=False, is_factorized=False, r=100):
        super().__init__()
        if is_factorized or emb_trg_sharing:
            raise NotImplementedError('factorized FFN / shared target embedding are outside the accelerated path')
        if dim_emb != dim_model:
            raise ValueError('dim_emb must equal dim_model (the reference adds the embedding to a dim_model table)')
        self.vocab = vocab
        self.num_layers, self.num_heads, self.dim_emb, self.dim_model = num_layers, num_heads, dim_emb, dim_model
        self.dim_inner, self.dim_key, self.dim_value = dim_inner, dim_key, dim_value
        self.trg_max_length, self.dropout_rate, self.r = trg_max_length, dropout, r
        self.trg_embedding = nn.Embedding(len(vocab.label2id), dim_emb, padding_idx=vocab.PAD_ID)
        self.positional_encoding = PositionalEncoding(dim_model, max_length=trg_max_length)
        self.layers = nn.ModuleList([DecoderLayer(dim_model, dim_inner, num_heads, dim_key, dim_value, dropout=dropout, r=r)
                                     for _ in range(num_layers)])
        self.output_linear = nn.Linear(dim_model, len(vocab.label2id), bias=False)
        nn.init.xavier_normal_(self.output_linear.weight)


class _ModelFn(torch.autograd.Function):
    """pred = model(x, lengths, target): forward on the HIP engine; backward accumulates into the flat grad buffer."""

    @staticmethod
    def forward(ctx, anchor, model, x, lengths, target):
        out = model._run_forward(model._theta_for_forward(), x, lengths, target)
        ctx.model = model
        ctx.token = model._pass_token
        # fresh tensor objects over the engine's buffers (the arena re-uses its storage on the next forward)
        pred, gold, hyp = (out[k].view(out[k].shape) for k in ('pred', 'gold', 'hyp'))
        ctx.mark_non_differentiable(gold, hyp)
        return pred, gold, hyp

    @staticmethod
    def backward(ctx, dpred, _g, _h):
        m = ctx.model
        if ctx.token != m._pass_token:
            raise RuntimeError('backward through a stale forward: the engine keeps the activations of the last forward only')
        m._sync_grad_views()
        m.engine.backward(m._gflat, 1.0, dpred=dpred)
        return None, None, None, None, None


class Transformer(nn.Module):
    def __init__(self, encoder, decoder, vocab, feat_extractor='vgg_cnn', train=True, is_factorized=False, r=100):
        super().__init__()
        if feat_extractor != 'vgg_cnn':
            raise NotImplementedError("only feat_extractor='vgg_cnn' is on the accelerated path")
        self.encoder, self.decoder, self.vocab = encoder, decoder, vocab
        self.feat_extractor, self.is_factorized, self.r = feat_extractor, is_factorized, r
        self.copy_grad = None
        if encoder.dropout_rate != decoder.dropout_rate:
            raise ValueError('encoder and decoder are built with one --dropout value in the reference factory')
        print('feat extractor:', feat_extractor)
        # indices 0,2,5,7 hold the convolutions exactly like the reference nn.Sequential (ReLU/MaxPool are parameter-free)
        self.conv = nn.Sequential(nn.Conv2d(1, 64, 3, stride=1, padding=1), nn.ReLU(),
                                  nn.Conv2d(64, 64, 3, stride=1, padding=1), nn.ReLU(), nn.MaxPool2d(2, stride=2),
                                  nn.Conv2d(64, 128, 3, stride=1, padding=1), nn.ReLU(),
                                  nn.Conv2d(128, 128, 3, stride=1, padding=1), nn.ReLU(), nn.MaxPool2d(2, stride=2))
        for p in self.parameters():
            if p.dim() > 1:
                nn.init.xavier_uniform_(p)
        self._layout = ParamLayout([(n, p.shape) for n, p in self.named_parameters()])
        self._theta = self._gflat = self._G = None
        self._theta_override = None
        self.engine = None
        self.engines, self.lane_streams = [], []
        self.n_lanes = max(1, int(os.environ.get('MTL_TASK_LANES', '8')))       # one lane per task of a meta-step, up to 8 (measured: 73.6 vs 76.7 ms at 3)
        self._pass_token = 0
        self._last = None
        self._anchor = None
        self._flatten()

    # ------------------------------------------------------------------ flat storage
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
        self._G = None
        self.copy_grad = None
        self._anchor = torch.zeros((), device=device, requires_grad=True)
        self.engine = None
        if device.type == 'cuda':
            e, d = self.encoder, self.decoder
            hp = Hyper(d=e.dim_model, r=e.r, h=e.num_heads, dk=e.dim_key, dv=e.dim_value, inner=e.dim_inner,
                       d_in=e.dim_input, n_enc=e.num_layers, n_dec=d.num_layers, V=len(self.vocab.label2id),
                       temperature=np.power(e.dim_key, 0.5), src_max_len=e.src_max_length, tgt_max_len=d.trg_max_length)
            pe_e, pe_d = e.positional_encoding.pe[0].contiguous(), d.positional_encoding.pe[0].contiguous()
            # task lanes: independent tasks of a meta-step run concurrently, each on its own stream with its own arena
            self.engines = [PassEngine(self._layout, hp, device, pe_e, pe_d) for _ in range(self.n_lanes)]
            self.lane_streams = [torch.cuda.Stream(device) for _ in range(self.n_lanes)]
            self.engine = self.engines[0]
            for eng in self.engines:
                eng.dropout_p = float(e.dropout_rate) if self.training else 0.0

    def _apply(self, fn, *a, **kw):
        out = super()._apply(fn, *a, **kw)
        self._flatten()
        return out

    def load_state_dict(self, state_dict, strict=True, **kw):
        out = super().load_state_dict(state_dict, strict=strict, **kw)   # copies INTO the flat views
        return out

    @property
    def flat_parameters(self):
        return self._theta

    @property
    def flat_grad(self):
        return self._gflat

    def _sync_grad_views(self):
        """torch optimizers' zero_grad(set_to_none=True) drops the .grad views: treat that as 'gradient is zero'."""
        first = next(iter(self.parameters()))
        if first.grad is None or first.grad.data_ptr() != self._gflat.data_ptr() + 4 * self._layout.off(self._layout.order[0]):
            self._gflat.zero_()
            for name, p in self.named_parameters():
                p.grad = self._layout.view(self._gflat, name)

    def zero_grad(self, set_to_none=False):
        self._sync_grad_views()
        self._gflat.zero_()

    # ------------------------------------------------------------------ forward / backward
    def train(self, mode=True):
        """nn.Module.train/eval: dropout (Philox keep-masks inside the HIP kernels) is active only in training mode."""
        out = super().train(mode)
        p = float(self.encoder.dropout_rate) if mode else 0.0
        for e in getattr(self, 'engines', []):
            e.dropout_p = p
        return out

    def _need_engine(self):
        if self.engine is None:
            raise RuntimeError('the model lives on %s: the product path needs an MI355X (call .cuda()); there is no CPU '
                               'fallback' % self._theta.device)
        return self.engine

    def _theta_for_forward(self):
        return self._theta if self._theta_override is None else self._theta_override

    def _run_forward(self, theta, x, lengths, target, smoothing=0.0):
        eng = self._need_engine()
        if x.device != theta.device:
            x = x.to(theta.device, non_blocking=True)
        out = eng.forward(theta, x.float(), lengths, target, smoothing=smoothing)
        self._pass_token += 1
        self._last = out
        return out

    def forward(self, padded_input, input_lengths, padded_target, verbose=False):
        """(B,1,F,T) fp32, (B) int, (B,L) int64 PAD=0  ->  pred (B,L+1,V), gold (B,L+1), hyp (B,L+1)
        Same contract as models/asr/transformer.py:120-149."""
        return _ModelFn.apply(self._anchor, self, padded_input, input_lengths, padded_target)

    # fast path used by the trainer: no autograd objects at all
    def pass_forward(self, x, lengths, target, theta=None, smoothing=0.0, lane=0):
        if lane == 0:
            return self._run_forward(self._theta if theta is None else theta, x, lengths, target, smoothing)
        self._need_engine()
        th = self._theta if theta is None else theta
        if x.device != th.device:
            x = x.to(th.device, non_blocking=True)
        return self.engines[lane].forward(th, x.float(), lengths, target, smoothing=smoothing)

    def pass_backward(self, grad=None, scale=1.0, lane=0):
        self._need_engine()
        self.engines[lane].backward(self._gflat if grad is None else grad, scale)

    # ------------------------------------------------------------------ copy_grad API (models/asr/transformer.py:205-240)
    def init_copy_grad_(self):
        self._G = torch.zeros_like(self._theta)
        self.copy_grad = [self._layout.view(self._G, n) for n in self._layout.order]

    def zero_copy_grad(self):
        if self._G is None:
            self.init_copy_grad_()
        else:
            self._G.zero_()

    def add_copy_grad(self):
        if self._G is None:
            self.init_copy_grad_()
        self._sync_grad_views()
        self._axpy(self._G, self._gflat, 1.0)

    def to_copy_grad(self):
        if self._G is None:
            self.init_copy_grad_()
        self._sync_grad_views()
        self._G.copy_(self._gflat)

    def from_copy_grad(self):
        if self._G is None:
            self.init_copy_grad_()
        self._sync_grad_views()
        self._gflat.copy_(self._G)

    def _axpy(self, y, x, a):
        if y.device.type == 'cuda':
            check(_lib.lib().mtl_axpy(torch.cuda.current_stream(y.device).cuda_stream, y.data_ptr(), x.data_ptr(), float(a),
                                      y.numel()), 'mtl_axpy')
        else:
            raise RuntimeError('copy_grad accumulation needs the HIP library on an MI355X device')

    def evaluate(self, padded_input, input_lengths, padded_target, args=None, beam_search=False, beam_width=0, beam_nbest=0, lm=None,
                 lm_rescoring=False, lm_weight=0.1, c_weight=1, start_token=-1, verbose=False, max_steps=300):
        """models/asr/transformer.py:162-202: returns (None, strs_hyps, strs_gold).
        The encoder + teacher-forced decoder pass supplies the gold strings exactly like the reference.  Greedy hypotheses come
        from PassEngine.greedy_decode (K/V-cached, device-resident token feedback); beam_search=True runs
        PassEngine.beam_decode per utterance with args.beam_width / args.beam_nbest / args.tgt_max_len like the reference
        (all n-best strings of all utterances, concatenated; falls back to greedy when the best hypothesis is empty).
        LM rescoring is not on the accelerated path."""
        if lm_rescoring:
            raise NotImplementedError('LM rescoring is outside the accelerated path')
        eng = self._need_engine()
        was_training = self.training
        self.eval()
        ids_nbest = None
        try:
            widen, eng.widen = eng.widen, '0'          # the decoders read the encoder output in the batch's own extent (T4 positions per utterance)
            try:
                out = self.pass_forward(padded_input, input_lengths, padded_target)
            finally:
                eng.widen = widen
            B, T = padded_input.shape[0], padded_input.shape[3]
            T4 = (T // 2) // 2
            mem = eng.arena['e%d.ff.y' % (eng.hp.n_enc - 1)] if eng.hp.n_enc else eng.arena['enc_in.y']
            start = self.vocab.SOS_ID if start_token < 0 else start_token      # the reference's callers pass vocab.SOS_ID
            strs_beam = None
            if beam_search:
                mem = mem.clone()                                              # the decode buffers live in the same arena
                ids_nbest, strs_beam = [], []
                for b in range(B):
                    res = eng.beam_decode(self._theta, mem.data_ptr() + 4 * b * T4 * eng.hp.d, T4, start, args.beam_width,
                                          args.beam_nbest, args.tgt_max_len, self._num_words, self.vocab.EOS_ID, c_weight)
                    for yseq, _score in res:
                        ids_nbest.append(yseq)
                        strs_beam.append(self._post_process_hyp(yseq))
                if len(strs_beam) == 0 or len(strs_beam[0].strip()) == 0:
                    strs_beam = None                                           # ">>>>>>> switch to greedy" (:190-196)
            if strs_beam is None:
                ids = eng.greedy_decode(self._theta, mem.data_ptr(), B, T4, start, max_steps).cpu()          # (steps, B)
        finally:
            self.train(was_training)
        if strs_beam is not None:
            strs_gold = [''.join(self.vocab.id2label[int(t)] for t in row) for row in out['gold_host']]
            self.last_beam_ids = ids_nbest
            return None, strs_beam, strs_gold
        strs_gold = [''.join(self.vocab.id2label[int(t)] for t in row) for row in out['gold_host']]
        strs_hyps = []
        for b in range(ids.shape[1]):
            st = ''
            for t in ids[:, b].tolist():
                if t == self.vocab.EOS_ID:
                    break
                st += self.vocab.id2label[t]
            strs_hyps.append(st)
        self.last_greedy_ids = ids
        return None, strs_hyps, strs_gold

    def _num_words(self, yseq):
        """word count of a finished hypothesis as modules/decoder.py:257-259 computes it (label string minus specials, split)"""
        v = self.vocab
        st = ''.join(v.id2label[int(c)] for c in yseq).replace(v.PAD_TOKEN, '').replace(v.SOS_TOKEN, '').replace(v.EOS_TOKEN, '')
        return len(st.replace('  ', ' ').split())

    def _post_process_hyp(self, yseq):
        """modules/decoder.py:117-128"""
        st = ''.join(self.vocab.id2label[int(x)] for x in yseq[1:])
        for tok in self.vocab.special_token_list:
            st = st.replace(tok, '')
        return st.replace('\u2581', ' ')
