"""Model factory and checkpoint I/O with the reference's names and on-disk format.

Mirrors utils/functions.py: `init_transformer_model` (:307-351), `save_meta_model` (:101-126), `load_meta_model`
(:158-188), `post_process` (:360-364).  Checkpoints are the reference's dict
{'vocab','args','epoch','model_state_dict','inner_opt','outer_opt','metrics'} with torch.optim objects pickled
whole, so files are interchangeable with reference-trained models.
"""
import logging
import math
import os

import torch

from .model import Decoder, Encoder, Transformer


def init_transformer_model(args, vocab, train=True, is_factorized=False, r=100):
    """Builds Encoder/Decoder/Transformer exactly like utils/functions.py:307-351, including the in-place derivation
    of args.dim_input from the sample rate / window (161 bins -> 40 pooled rows x 128 channels = 5120)."""
    if args.feat_extractor != 'vgg_cnn':
        raise NotImplementedError("only feat_extractor='vgg_cnn' is on the accelerated path")
    bins = int(math.floor((args.sample_rate * args.window_size) / 2) + 1)
    args.dim_input = int(math.floor(int(math.floor(bins) / 2) / 2)) * 128
    if getattr(args, 'feat', 'spectrogram') == 'logfbank':
        raise NotImplementedError('logfbank features are outside the accelerated path')
    encoder = Encoder(args.num_enc_layers, num_heads=args.num_heads, dim_model=args.dim_model, dim_key=args.dim_key,
                      dim_value=args.dim_value, dim_input=args.dim_input, dim_inner=args.dim_inner,
                      src_max_length=args.src_max_len, dropout=args.dropout, is_factorized=is_factorized, r=r)
    decoder = Decoder(vocab, num_layers=args.num_dec_layers, num_heads=args.num_heads, dim_emb=args.dim_emb,
                      dim_model=args.dim_model, dim_inner=args.dim_inner, dim_key=args.dim_key, dim_value=args.dim_value,
                      trg_max_length=args.tgt_max_len, dropout=args.dropout, emb_trg_sharing=args.emb_trg_sharing,
                      is_factorized=is_factorized, r=r)
    return Transformer(encoder, decoder, vocab, feat_extractor=args.feat_extractor, train=train)


def post_process(string, special_token_list):
    for tok in special_token_list:
        string = string.replace(tok, '')
    return string.replace('▁', ' ')


def _as_torch_opt(opt):
    return opt.to_torch() if hasattr(opt, 'to_torch') else opt


def save_meta_model(model, vocab, epoch, inner_opt, outer_opt, metrics, args, best_model=False):
    folder = '{}/{}'.format(args.save_folder, args.name)
    save_path = folder + ('/best_model.th' if best_model else '/epoch_{}.th'.format(epoch))
    os.makedirs(folder, exist_ok=True)
    print('SAVE MODEL to', save_path)
    logging.info('SAVE MODEL to ' + save_path)
    state = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    # optimizers are exported as torch.optim objects over CPU copies so the file loads without a GPU
    payload = {'vocab': vocab, 'args': args, 'epoch': epoch, 'model_state_dict': state,
               'inner_opt': _opt_state_only(_as_torch_opt(inner_opt)), 'outer_opt': _opt_state_only(_as_torch_opt(outer_opt)),
               'metrics': metrics}
    torch.save(payload, save_path)
    return save_path


class _OptStateCarrier:
    """Pickle-friendly stand-in exposing `.state_dict()` like the optimizer objects the reference pickles
    (utils/functions.py:185-186 only ever calls `.state_dict()` on them)."""

    def __init__(self, sd):
        self._sd = sd

    def state_dict(self):
        return self._sd


def _opt_state_only(opt):
    sd = opt.state_dict()
    for st in sd['state'].values():
        for k, v in list(st.items()):
            if torch.is_tensor(v):
                st[k] = v.detach().cpu().clone()
    return _OptStateCarrier(sd)


def load_meta_model(load_path, train=True):
    """-> (model, vocab, inner_opt, outer_opt, epoch, metrics, args); optimizers come back as torch.optim objects holding
    the saved state (TransientTrainer.train converts them to its flat-buffer optimizers)."""
    ckpt = torch.load(load_path, map_location=torch.device('cpu'), weights_only=False)
    args, vocab = ckpt['args'], ckpt['vocab']
    model = init_transformer_model(args, vocab, train=train, is_factorized=getattr(args, 'is_factorized', False),
                                   r=getattr(args, 'r', 100))
    model.load_state_dict(ckpt['model_state_dict'])
    model = model.cuda() if getattr(args, 'cuda', False) and torch.cuda.is_available() else model.cpu()
    inner_opt = torch.optim.SGD(model.parameters(), lr=args.lr)
    outer_opt = torch.optim.Adam(model.parameters(), lr=args.meta_lr)
    inner_opt.load_state_dict(ckpt['inner_opt'].state_dict())
    outer_opt.load_state_dict(ckpt['outer_opt'].state_dict())
    return model, vocab, inner_opt, outer_opt, ckpt['epoch'], ckpt['metrics'], args
