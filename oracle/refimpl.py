"""ORACLE (test infrastructure, NOT product code).

CPU restatement, in plain PyTorch fp32, of the reference's meta-transfer hot path:
VGG-CNN front-end + low-rank-attention Transformer encoder/decoder + CE loss + the
`--copy-grad` inner/outer meta step.  Only `tests/`, `__graft_entry__.smoke()` and
`bench.py`'s `cpu_baseline` leg may import this module; the product package
(`meta-transfer-learning_amd/`) never does.

Parity pinning: `oracle/make_golden.py` imports the real reference from /root/reference
(in the build container only) and writes `tests/golden/*.npz`; `tests/test_oracle_golden.py`
checks this restatement against those vectors (bit-exact at fixed thread count).

Every block cites the reference file:line it restates (paths relative to /root/reference).
"""
import math

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

PAD_ID, SOS_ID, EOS_ID, OOV_ID = 0, 1, 2, 3  # utils/data.py:8


# --------------------------------------------------------------------------------------
# masks / small helpers  (modules/common_layers.py:13-84)
# --------------------------------------------------------------------------------------
def length_mask(lengths, width):
    """(B, width) float mask, 1 where j < lengths[b].  common_layers.py:38-52 (input_lengths branch).

    NB: callers pass RAW frame counts against the 4x-pooled width.
    """
    pos = torch.arange(width).unsqueeze(0)
    return (pos < lengths.to(torch.int64).unsqueeze(1)).float()


def sinusoid_table(max_len, d):
    """common_layers.py:90-98 -- written with the same torch expression so it is bit-identical."""
    pe = torch.zeros(max_len, d)
    position = torch.arange(0, max_len).unsqueeze(1).float()
    exp_term = torch.exp(torch.arange(0, d, 2).float() * -(math.log(10000.0) / d))
    pe[:, 0::2] = torch.sin(position * exp_term)
    pe[:, 1::2] = torch.cos(position * exp_term)
    return pe.unsqueeze(0)


def decoder_io(padded_target):
    """modules/decoder.py:55-69: strip PAD, seq_in=[SOS,y] padded with EOS, seq_out=[y,EOS] padded with PAD."""
    rows = [y[y != PAD_ID] for y in padded_target]
    width = max(int(r.numel()) for r in rows) + 1
    seq_in = padded_target.new_full((len(rows), width), EOS_ID)
    seq_out = padded_target.new_full((len(rows), width), PAD_ID)
    for i, r in enumerate(rows):
        n = int(r.numel())
        seq_in[i, 0] = SOS_ID
        seq_in[i, 1:n + 1] = r
        seq_out[i, :n] = r
        seq_out[i, n] = EOS_ID
    return seq_in, seq_out


def relu_replay(x, gate):
    """ReLU whose branch decisions are REPLAYED from `gate` (bool, x.shape, True = pass) instead of taken from sign(x).
    Test infrastructure for the 1e-4 parity bar: two exact-fp32 implementations differ by ~1e-7 in pre-activations, so an
    element within rounding of 0 can be gated differently; replaying the device path's own decisions removes that branch
    noise (exactly like the dropout-mask replay) while every arithmetic op stays the oracle's."""
    return x * gate.to(x.dtype)


def pool_replay(z, argmax, gate):
    """ReLU + MaxPool2d(2, stride 2, floor) of the pre-activation z (B,C,F,T) with replayed decisions: `argmax` (B,C,F//2,T//2)
    in {0..3} = 2*(f&1) + (t&1) names the window element that is routed, `gate` (bool) is the sign of that maximum."""
    B, C, Fz, Tz = z.shape
    Fp, Tp = Fz // 2, Tz // 2
    w = z[:, :, :2 * Fp, :2 * Tp].reshape(B, C, Fp, 2, Tp, 2).permute(0, 1, 2, 4, 3, 5).reshape(B, C, Fp, Tp, 4)
    v = w.gather(4, argmax.long().unsqueeze(-1)).squeeze(-1)
    return v * gate.to(z.dtype)


class PositionTable(nn.Module):
    def __init__(self, d, max_len):
        super().__init__()
        self.register_buffer('pe', sinusoid_table(max_len, d))


# --------------------------------------------------------------------------------------
# layers  (modules/common_layers.py:110-132, 238-331)
# --------------------------------------------------------------------------------------
class LowRankMHA(nn.Module):
    """FactorizedMultiHeadAttention, common_layers.py:238-306 (+ ScaledDotProductAttention :308-331)."""

    def __init__(self, heads, d, dk, dv, r):
        super().__init__()
        self.h, self.dk, self.dv = heads, dk, dv
        # creation + init order matters for the RNG stream
        self.query_linear_a = nn.Linear(d, r, bias=False)
        self.query_linear_b = nn.Linear(r, heads * dk)
        self.key_linear_a = nn.Linear(d, r, bias=False)
        self.key_linear_b = nn.Linear(r, heads * dk)
        self.value_linear_a = nn.Linear(d, r, bias=False)
        self.value_linear_b = nn.Linear(r, heads * dv)
        for lin, width in ((self.query_linear_a, dk), (self.query_linear_b, dk), (self.key_linear_a, dk),
                           (self.key_linear_b, dk), (self.value_linear_a, dv), (self.value_linear_b, dv)):
            nn.init.normal_(lin.weight, mean=0, std=np.sqrt(2.0 / (d + width)))
        self.temperature = np.power(dk, 0.5)
        self.layer_norm = nn.LayerNorm(d)
        self.output_linear_a = nn.Linear(heads * dv, r, bias=False)
        self.output_linear_b = nn.Linear(r, d)
        nn.init.xavier_normal_(self.output_linear_a.weight)
        nn.init.xavier_normal_(self.output_linear_b.weight)

    def forward(self, xq, xkv, blocked, drop=None, tag=''):
        """blocked: (B, Tq, Tk) bool, True = masked out.  drop: optional {site: multiplicative keep-mask already scaled by
        1/(1-p)} replaying a given dropout realisation (sites: common_layers.py:328 -> tag+'mP', :303 -> tag+'mo')."""
        B, Tq, _ = xq.shape
        Tk = xkv.shape[1]
        q = self.query_linear_b(self.query_linear_a(xq)).view(B, Tq, self.h, self.dk).transpose(1, 2)
        k = self.key_linear_b(self.key_linear_a(xkv)).view(B, Tk, self.h, self.dk).transpose(1, 2)
        v = self.value_linear_b(self.value_linear_a(xkv)).view(B, Tk, self.h, self.dv).transpose(1, 2)
        s = torch.matmul(q, k.transpose(2, 3)) / self.temperature          # :321-322 (divide AFTER the product)
        s = s.masked_fill(blocked.unsqueeze(1), -np.inf)                   # :325
        p = torch.softmax(s, dim=-1)                                       # :327
        if drop is not None:
            p = p * drop[tag + 'mP']                                       # :328
        o = torch.matmul(p, v).transpose(1, 2).reshape(B, Tq, self.h * self.dv)
        o = self.output_linear_b(self.output_linear_a(o))                  # :303
        if drop is not None:
            o = o * drop[tag + 'mo']
        return self.layer_norm(o + xq)                                     # :304


class FFN(nn.Module):
    """PositionwiseFeedForward, common_layers.py:110-132."""

    def __init__(self, d, inner):
        super().__init__()
        self.linear_1 = nn.Linear(d, inner)
        self.linear_2 = nn.Linear(inner, d)
        self.layer_norm = nn.LayerNorm(d)

    def forward(self, x, drop=None, tag='', gates=None):
        h = self.linear_1(x)
        h = F.relu(h) if gates is None else relu_replay(h, gates[tag + 'h1'].reshape(h.shape))
        h = self.linear_2(h)
        if drop is not None:
            h = h * drop[tag + 'mf']                                       # common_layers.py:130
        return self.layer_norm(h + x)


class EncLayer(nn.Module):
    """modules/encoder.py:83-106."""

    def __init__(self, heads, d, inner, dk, dv, r):
        super().__init__()
        self.self_attn = LowRankMHA(heads, d, dk, dv, r)
        self.pos_ffn = FFN(d, inner)

    def forward(self, x, keep, blocked, drop=None, tag='', gates=None):
        x = self.self_attn(x, x, blocked, drop, tag + 'sa.') * keep
        return self.pos_ffn(x, drop, tag + 'ff.', gates) * keep


class Enc(nn.Module):
    """modules/encoder.py:15-80 (non-factorized input projection)."""

    def __init__(self, layers, heads, d, dk, dv, d_in, inner, src_max_len, r):
        super().__init__()
        self.input_linear = nn.Linear(d_in, d)
        self.layer_norm_input = nn.LayerNorm(d)
        self.positional_encoding = PositionTable(d, src_max_len)
        self.layers = nn.ModuleList([EncLayer(heads, d, inner, dk, dv, r) for _ in range(layers)])

    def forward(self, feats, lengths, drop=None, gates=None):
        B, T, _ = feats.shape
        keep = length_mask(lengths, T)                                     # encoder.py:64 (Q2)
        blocked = (keep < 1).unsqueeze(1).expand(B, T, T)                  # encoder.py:66
        x = self.layer_norm_input(self.input_linear(feats)) + self.positional_encoding.pe[:, :T]  # :72-73
        for i, layer in enumerate(self.layers):
            x = layer(x, keep.unsqueeze(-1), blocked, drop, 'e%d.' % i, gates)
        return x


class DecLayer(nn.Module):
    """modules/decoder.py:293-323."""

    def __init__(self, d, inner, heads, dk, dv, r):
        super().__init__()
        self.self_attn = LowRankMHA(heads, d, dk, dv, r)
        self.encoder_attn = LowRankMHA(heads, d, dk, dv, r)
        self.pos_ffn = FFN(d, inner)

    def forward(self, x, mem, keep, self_blocked, cross_blocked, drop=None, tag='', gates=None):
        x = self.self_attn(x, x, self_blocked, drop, tag + 'sa.') * keep
        x = self.encoder_attn(x, mem, cross_blocked, drop, tag + 'ca.') * keep
        return self.pos_ffn(x, drop, tag + 'ff.', gates) * keep


class Dec(nn.Module):
    """modules/decoder.py:14-115."""

    def __init__(self, vocab_size, layers, heads, d_emb, d, inner, dk, dv, trg_max_len, r):
        super().__init__()
        self.trg_embedding = nn.Embedding(vocab_size, d_emb, padding_idx=PAD_ID)
        self.positional_encoding = PositionTable(d, trg_max_len)
        self.layers = nn.ModuleList([DecLayer(d, inner, heads, dk, dv, r) for _ in range(layers)])
        self.output_linear = nn.Linear(d, vocab_size, bias=False)
        nn.init.xavier_normal_(self.output_linear.weight)

    def forward(self, padded_target, mem, src_lengths, drop=None, gates=None):
        seq_in, seq_out = decoder_io(padded_target)
        B, L = seq_in.shape
        keep = seq_in.ne(EOS_ID).float().unsqueeze(-1)                     # decoder.py:86 (Q7: keyed on EOS)
        future = torch.triu(torch.ones(L, L, dtype=torch.bool), diagonal=1).unsqueeze(0)
        self_blocked = seq_in.eq(EOS_ID).unsqueeze(1).expand(B, L, L) | future          # :87-90
        cross_blocked = (length_mask(src_lengths, mem.shape[1]) < 1).unsqueeze(1).expand(B, L, mem.shape[1])  # :93-94
        x = self.trg_embedding(seq_in) + self.positional_encoding.pe[:, :L]             # :96 (scale 1.0)
        if drop is not None:
            x = x * drop['dec_in.me']
        for i, layer in enumerate(self.layers):
            x = layer(x, mem, keep, self_blocked, cross_blocked, drop, 'd%d.' % i, gates)
        return self.output_linear(x), seq_out                                          # :108-113


class SpeechTransformer(nn.Module):
    """models/asr/transformer.py:14-149 with feat_extractor='vgg_cnn'."""

    def __init__(self, vocab_size, enc_layers, dec_layers, heads, d, dk, dv, inner, d_emb,
                 src_max_len, trg_max_len, r=100, freq_bins=161):
        super().__init__()
        d_in = (freq_bins // 2 // 2) * 128                                 # utils/functions.py:318-321
        self.encoder = Enc(enc_layers, heads, d, dk, dv, d_in, inner, src_max_len, r)
        self.decoder = Dec(vocab_size, dec_layers, heads, d_emb, d, inner, dk, dv, trg_max_len, r)
        self.conv = nn.Sequential(                                         # transformer.py:47-59
            nn.Conv2d(1, 64, 3, stride=1, padding=1), nn.ReLU(),
            nn.Conv2d(64, 64, 3, stride=1, padding=1), nn.ReLU(),
            nn.MaxPool2d(2, stride=2),
            nn.Conv2d(64, 128, 3, stride=1, padding=1), nn.ReLU(),
            nn.Conv2d(128, 128, 3, stride=1, padding=1), nn.ReLU(),
            nn.MaxPool2d(2, stride=2))
        for p in self.parameters():                                        # transformer.py:74-76 (Q4)
            if p.dim() > 1:
                nn.init.xavier_uniform_(p)

    def conv_stack(self, x, gates=None):
        """transformer.py:48-59; with `gates` the ReLU / max-pool decisions are replayed (relu_replay / pool_replay):
        gates['conv0'], gates['conv5'] bool (B,C,F,T); gates['am1'/'am2'] window indices, gates['pool1'/'pool2'] bool."""
        if gates is None:
            return self.conv(x)
        c = self.conv
        z = relu_replay(c[0](x), gates['conv0'])
        z = pool_replay(c[2](z), gates['am1'], gates['pool1'])
        z = relu_replay(c[5](z), gates['conv5'])
        return pool_replay(c[7](z), gates['am2'], gates['pool2'])

    def forward(self, padded_input, input_lengths, padded_target, drop=None, gates=None):
        f = self.conv_stack(padded_input, gates)                           # :133
        B, C, H, W = f.shape
        f = f.reshape(B, C * H, W).transpose(1, 2).contiguous()            # :136-138
        mem = self.encoder(f, input_lengths, drop, gates)
        pred, gold = self.decoder(padded_target, mem, input_lengths, drop, gates)
        hyp = torch.topk(pred, 1, dim=2)[1].squeeze(2)                     # :146-147
        return pred, gold, hyp


def greedy_search(model, padded_input, input_lengths, start_token, steps):
    """Decoder.greedy_search (modules/decoder.py:131-185) restated: at every step the WHOLE decoder is re-run on the prefix,
    non_pad_mask all ones, causal self-attention mask only, no encoder padding mask; arg-max of the last position."""
    model.eval()
    with torch.no_grad():
        f = model.conv(padded_input)
        B, C, H, W = f.shape
        mem = model.encoder(f.view(B, C * H, W).transpose(1, 2).contiguous(), input_lengths)
        dec = model.decoder
        ys = torch.full((B, 1), int(start_token), dtype=torch.int64)
        for _ in range(steps):
            Lq = ys.shape[1]
            future = torch.triu(torch.ones(Lq, Lq, dtype=torch.bool), diagonal=1).unsqueeze(0).expand(B, Lq, Lq)
            nomask = torch.zeros(B, Lq, mem.shape[1], dtype=torch.bool)
            keep = torch.ones(B, Lq, 1)
            x = dec.trg_embedding(ys) + dec.positional_encoding.pe[:, :Lq]
            for layer in dec.layers:
                x = layer(x, mem, keep, future, nomask)
            nxt = dec.output_linear(x)[:, -1].max(dim=1)[1]
            ys = torch.cat([ys, nxt.unsqueeze(1)], dim=1)
    model.train()
    return ys[:, 1:]


def beam_search(model, padded_input, input_lengths, start_token, beam_width, nbest, tgt_max_len, num_words, c_weight=1.0):
    """Decoder.beam_search (modules/decoder.py:187-291, lm_rescoring=False) restated, utterance by utterance: every live
    hypothesis re-runs the WHOLE decoder on its prefix (no padding masks), log_softmax of the last position, top beam_width
    expansions, cumulative stable sort + truncation to the beam inside the hypothesis loop, EOS forced at step T' - 1,
    final_score = score + sqrt(num_words(yseq)) * c_weight, n-best by final_score.  -> per utterance a list of
    (yseq incl. start token and EOS, final_score)."""
    import math
    model.eval()
    out = []
    with torch.no_grad():
        f = model.conv(padded_input)
        B, C, H, W = f.shape
        mem_all = model.encoder(f.view(B, C * H, W).transpose(1, 2).contiguous(), input_lengths)
        dec = model.decoder
        max_len = mem_all.shape[1]
        for b in range(B):
            mem = mem_all[b:b + 1]
            hyps = [dict(score=0.0, yseq=torch.full((1, 1), int(start_token), dtype=torch.int64))]
            ended = []
            for i in range(tgt_max_len):
                kept = []
                for hyp in hyps:
                    ys = hyp['yseq']
                    Lq = ys.shape[1]
                    future = torch.triu(torch.ones(Lq, Lq, dtype=torch.bool), diagonal=1).unsqueeze(0)
                    x = dec.trg_embedding(ys) + dec.positional_encoding.pe[:, :Lq]
                    for layer in dec.layers:
                        x = layer(x, mem, torch.ones(1, Lq, 1), future, torch.zeros(1, Lq, max_len, dtype=torch.bool))
                    local = F.log_softmax(dec.output_linear(x[:, -1]), dim=1)
                    best, ids = torch.topk(local, beam_width, dim=1)
                    for j in range(beam_width):
                        kept.append(dict(score=hyp['score'] + best[0, j], yseq=torch.cat([ys, ids[0, j].view(1, 1)], dim=1)))
                    kept = sorted(kept, key=lambda h: h['score'], reverse=True)[:beam_width]
                hyps = kept
                if i == max_len - 1:
                    for hyp in hyps:
                        hyp['yseq'] = torch.cat([hyp['yseq'], torch.full((1, 1), EOS_ID, dtype=torch.int64)], dim=1)
                live = []
                for hyp in hyps:
                    if int(hyp['yseq'][0, -1]) == EOS_ID:
                        hyp['final_score'] = hyp['score'] + math.sqrt(num_words(hyp['yseq'][0].tolist())) * c_weight
                        ended.append(hyp)
                    else:
                        live.append(hyp)
                hyps = live
                if not hyps:
                    break
            best_n = sorted(ended, key=lambda h: h['final_score'], reverse=True)[:min(len(ended), nbest)]
            out.append([(h['yseq'][0].tolist(), float(h['final_score'])) for h in best_n])
    model.train()
    return out


def build_model(cfg, seed=123456):
    """cfg: dict with the hyper-parameters of utils/functions.py:307-351; seeds like meta_transfer_train.py:109."""
    torch.manual_seed(seed)
    m = SpeechTransformer(cfg['vocab_size'], cfg['num_enc_layers'], cfg['num_dec_layers'], cfg['num_heads'],
                          cfg['dim_model'], cfg['dim_key'], cfg['dim_value'], cfg['dim_inner'], cfg['dim_emb'],
                          cfg['src_max_len'], cfg['tgt_max_len'], r=cfg.get('r', 100))
    m.train()
    return m


def ce_loss(pred, gold):
    """utils/metrics.py:126 -- mean CE over non-PAD targets of the whole batch."""
    return F.cross_entropy(pred.reshape(-1, pred.size(2)), gold.reshape(-1), ignore_index=PAD_ID, reduction='mean')


# --------------------------------------------------------------------------------------
# the meta step  (trainer/asr/transient_trainer.py:152-255, --copy-grad branch)
# --------------------------------------------------------------------------------------
def flat(tensors):
    return torch.cat([t.reshape(-1) for t in tensors])


def clip_(grads, max_norm):
    """torch.nn.utils.clip_grad_norm_ semantics on a list of gradient tensors (in place)."""
    total = torch.sqrt(sum((g.detach() ** 2).sum() for g in grads))
    coef = torch.clamp(max_norm / (total + 1e-6), max=1.0)
    for g in grads:
        g.mul_(coef)
    return grads


def meta_gradient(model, task_batches, val_batch, alpha, max_norm=None, gates=None, n_tasks=None):
    """G = sum_m [ grad L_tr,m(theta0) + (1/n) grad L_val(theta0 - alpha grad L_tr,m(theta0)) ]  (SURVEY Q1).

    task_batches: list of (x, lengths, y); val_batch: (x, lengths, y).  Restores theta0 before returning.
    max_norm: `--clip` (transient_trainer.py:205-206): the train gradient is clipped BEFORE the inner step and the
    clipped tensor is what stays in .grad.
    gates: optional list of 2n gate dicts (SpeechTransformer.conv_stack), one per forward in execution order (task 0 train,
    task 0 valid, task 1 train, ...): replays the device path's ReLU / max-pool decisions.
    n_tasks: the GLOBAL task count when task_batches are only one rank's share of a sharded meta-step (the 1/n of :226 is global:
    n = len(train_data_list)); default: len(task_batches).
    Returns (G list per parameter, [tr losses], [val losses], [(gold, hyp) of every forward]).
    """
    params = list(model.parameters())
    theta0 = [p.detach().clone() for p in params]
    n = n_tasks or len(task_batches)
    G = [torch.zeros_like(p) for p in params]
    tr_losses, val_losses, labels = [], [], []
    for m_, (x, lens, y) in enumerate(task_batches):
        pred, gold, hyp = model(x, lens, y, gates=gates[2 * m_] if gates is not None else None)
        loss = ce_loss(pred, gold)
        g_tr = torch.autograd.grad(loss, params)                           # :198-199
        if max_norm is not None:
            g_tr = clip_([g.clone() for g in g_tr], max_norm)
        tr_losses.append(float(loss.detach()))
        labels.append((gold.clone(), hyp.clone()))
        with torch.no_grad():
            for p, g in zip(params, g_tr):                                 # inner SGD, :207
                p.add_(g, alpha=-alpha)
        pred, gold, hyp = model(*val_batch, gates=gates[2 * m_ + 1] if gates is not None else None)
        vloss = ce_loss(pred, gold)
        val_losses.append(float(vloss.detach()))
        labels.append((gold.clone(), hyp.clone()))
        g_val = torch.autograd.grad(vloss / n, params)                     # :226-227
        with torch.no_grad():
            for acc, a, b in zip(G, g_tr, g_val):                          # .grad held g_tr, += g_val (Q1); :229
                acc.add_(a + b)
            for p, t0 in zip(params, theta0):                              # :237
                p.copy_(t0)
    return G, tr_losses, val_losses, labels


class AdamState:
    """torch.optim.Adam defaults (transient_trainer.py:109): betas (0.9,0.999), eps 1e-8, no weight decay."""

    def __init__(self, params, lr):
        self.lr, self.t = lr, 0
        self.m = [torch.zeros_like(p) for p in params]
        self.v = [torch.zeros_like(p) for p in params]

    def step(self, params, grads):
        self.t += 1
        b1, b2, eps = 0.9, 0.999, 1e-8
        bc1, bc2 = 1 - b1 ** self.t, 1 - b2 ** self.t
        with torch.no_grad():
            for p, g, m, v in zip(params, grads, self.m, self.v):
                m.mul_(b1).add_(g, alpha=1 - b1)
                v.mul_(b2).addcmul_(g, g, value=1 - b2)
                denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
                p.addcdiv_(m, denom, value=-self.lr / bc1)


def meta_step(model, adam, task_batches, val_batch, alpha, max_norm=None):
    G, tr, va, labels = meta_gradient(model, task_batches, val_batch, alpha, max_norm)
    if max_norm is not None:
        clip_(G, max_norm)                                                 # :253-254
    adam.step(list(model.parameters()), G)                                 # :248-255
    return G, tr, va, labels


def joint_step(model, adam, task_batches, max_norm=None):
    """trainer/asr/joint_trainer.py:182-262 without discriminator: grad of sum_m L_tr,m / n, one Adam(lr) step."""
    params = list(model.parameters())
    n = len(task_batches)
    G = [torch.zeros_like(p) for p in params]
    losses = []
    for (x, lens, y) in task_batches:
        pred, gold, _ = model(x, lens, y)
        loss = ce_loss(pred, gold)
        losses.append(float(loss.detach()))
        for acc, g in zip(G, torch.autograd.grad(loss / n, params)):
            acc.add_(g)
    if max_norm is not None:
        clip_(G, max_norm)
    adam.step(params, G)
    return G, losses


# --------------------------------------------------------------------------------------
# synthetic batches  (SURVEY 8(d) "Synthetic inputs")
# --------------------------------------------------------------------------------------
def synth_batch(seed, k, T, L, vocab_size, variable=False, freq_bins=161):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(k, 1, freq_bins, T, generator=g)
    y = torch.randint(4, vocab_size, (k, L), generator=g)
    lens = torch.full((k,), T, dtype=torch.int32)
    if variable:
        lens = torch.randint(max(T // 8, 1), T + 1, (k,), generator=g).to(torch.int32)
        lens[0] = T
        if k > 1:
            lens[-1] = max(T // 8, 1)   # shorter than T/4: exercises the raw-length-vs-pooled-axis masks (Q2)
        tl = torch.randint(max(L // 2, 1), L + 1, (k,), generator=g)
        tl[0] = L
        for i in range(k):
            x[i, :, :, int(lens[i]):] = 0
            y[i, int(tl[i]):] = PAD_ID
    return x, lens, y
