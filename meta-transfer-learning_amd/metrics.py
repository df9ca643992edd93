"""Loss / CER helpers with the reference's call signatures (utils/metrics.py:38-126), backed by the HIP cross-entropy kernels
and the C Levenshtein helper."""
import torch

from . import _lib

check = _lib.check


def calculate_cer(s1, s2):
    """utils/metrics.py:38-44 (python-Levenshtein replaced by mtl_levenshtein_u32)."""
    return _lib.levenshtein(s1, s2)


class _CrossEntropyFn(torch.autograd.Function):
    """loss, hyp = CE(pred (B,T,V), gold (B,T)) on the device for ANY pred / gold pair (the model's own output, a slice of it,
    a re-used copy): forward = mtl_ce_argmax_fwd (utils/metrics.py:113-126 incl. label smoothing; lowest-index arg-max),
    backward = mtl_ce_bwd into d(pred), which autograd hands on to whatever produced `pred` (the HIP model's backward takes an
    external dpred).  Nothing here depends on the engine's last forward."""

    @staticmethod
    def forward(ctx, pred, gold, pad_id, smoothing):
        if pred.device.type != 'cuda':
            raise RuntimeError('calculate_metrics runs on the MI355X only (pred is on %s); there is no CPU fallback' % pred.device)
        if pred.dim() != 3 or pred.dtype != torch.float32:
            raise ValueError('pred must be (B, T, C) fp32')
        B, T, V = pred.shape
        p2 = pred.contiguous()
        g2 = gold.to(device=pred.device, dtype=torch.int64).contiguous().view(-1)
        rows = B * T
        n_nonpad = int((g2 != pad_id).sum().item())                  # the reference syncs here too (non_pad_mask.sum().item())
        if n_nonpad == 0:
            raise ValueError('cross entropy over a batch without a single non-PAD target')
        lib = _lib.lib()
        st = torch.cuda.current_stream(pred.device).cuda_stream
        lse = torch.empty(rows, dtype=torch.float32, device=pred.device)
        hyp = torch.empty(rows, dtype=torch.int64, device=pred.device)
        rowloss = torch.empty(rows, dtype=torch.float32, device=pred.device)
        loss = torch.empty(1, dtype=torch.float32, device=pred.device)
        check(lib.mtl_ce_argmax_fwd(st, p2.data_ptr(), g2.data_ptr(), rows, V, V, int(pad_id), float(smoothing), n_nonpad, None,
                                    lse.data_ptr(), hyp.data_ptr(), rowloss.data_ptr(), loss.data_ptr()), 'mtl_ce_argmax_fwd')
        ctx.save_for_backward(p2, g2, lse)
        ctx.meta = (int(pad_id), float(smoothing), n_nonpad)
        hyp = hyp.view(B, T)
        ctx.mark_non_differentiable(hyp)
        return loss.reshape(()), hyp

    @staticmethod
    def backward(ctx, gout, _ghyp):
        p2, g2, lse = ctx.saved_tensors
        pad_id, smoothing, n_nonpad = ctx.meta
        B, T, V = p2.shape
        dpred = torch.empty_like(p2)
        g = gout.reshape(1).to(torch.float32).contiguous()
        st = torch.cuda.current_stream(p2.device).cuda_stream
        check(_lib.lib().mtl_ce_bwd(st, p2.data_ptr(), lse.data_ptr(), g2.data_ptr(), B * T, V, V, pad_id, smoothing,
                                    1.0 / n_nonpad, g.data_ptr(), dpred.data_ptr(), V), 'mtl_ce_bwd')
        return dpred, None, None, None


def calculate_metrics(pred, gold, pad_id, input_lengths=None, target_lengths=None, non_pad_mask=None, smoothing=0.0,
                      loss_type='ce'):
    """-> (loss tensor, num_correct) like utils/metrics.py:68-94 for loss_type='ce' (CTC is outside the accelerated path).
    pred (B,T,C) fp32 and gold (B,T) on the device; `non_pad_mask` overrides gold.ne(pad_id) and, like the reference, pads
    `gold` IN PLACE where the mask is False.  `loss.backward()` produces d(pred) with mtl_ce_bwd and continues into the
    producer of `pred`."""
    if loss_type != 'ce':
        raise NotImplementedError("only loss_type='ce' is on the accelerated path")
    if non_pad_mask is None:
        non_pad_mask = gold.ne(pad_id)
    else:
        gold.masked_fill_(torch.logical_not(non_pad_mask), pad_id)
    loss, hyp = _CrossEntropyFn.apply(pred, gold, pad_id, float(smoothing or 0.0))
    num_correct = int((hyp.eq(gold.to(hyp.device)) & non_pad_mask.to(hyp.device)).sum().item())
    return loss, num_correct
