"""Loss / CER helpers with the reference's call signatures (utils/metrics.py:38-94), backed by the HIP CE kernel
and the C Levenshtein helper."""
import torch

from . import _lib


def calculate_cer(s1, s2):
    """utils/metrics.py:38-44 (python-Levenshtein replaced by mtl_levenshtein_u32)."""
    return _lib.levenshtein(s1, s2)


def _model_of(pred):
    fn = getattr(pred, 'grad_fn', None)
    model = getattr(fn, 'model', None)
    if model is None:
        raise RuntimeError('calculate_metrics expects the `pred` tensor returned by the HIP model forward '
                           '(its loss comes from the fused cross-entropy kernel of that forward)')
    return model


def calculate_metrics(pred, gold, pad_id, input_lengths=None, target_lengths=None, non_pad_mask=None, smoothing=0.0,
                      loss_type='ce'):
    """-> (loss tensor, num_correct) like utils/metrics.py:68-94 for loss_type='ce'.
    The loss value was produced by mtl_ce_argmax_fwd during the forward; `.backward()` on it runs the HIP backward."""
    if loss_type != 'ce':
        raise NotImplementedError("only loss_type='ce' is on the accelerated path")
    if smoothing != 0.0:
        raise NotImplementedError('label smoothing goes through TransientTrainer (pass_forward(smoothing=...))')
    model = _model_of(pred)
    loss = model.loss_from_last_forward(pred)
    hyp = model._last['hyp']
    mask = gold.ne(pad_id)
    num_correct = int((hyp.eq(gold) & mask).sum().item())
    return loss, num_correct
