"""Model factory and checkpoint I/O with the reference's names and on-disk format.

Mirrors utils/functions.py: `init_transformer_model` (:307-351), `save_meta_model` (:101-126), `load_meta_model`
(:158-188), `post_process` (:360-364).  Checkpoints are the reference's dict
{'vocab','args','epoch','model_state_dict','inner_opt','outer_opt','metrics'} with torch.optim objects pickled
whole, so files are interchangeable with reference-trained models.
"""
import logging
import math
import os
import pickle

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
    """utils/functions.py:101-126: the same `.th` dict ('vocab', 'args', 'epoch', 'model_state_dict', 'inner_opt', 'outer_opt',
    'metrics').  Written so that BOTH stacks can read it: tensors on the CPU; the optimizers as real
    torch.optim.SGD / Adam objects (the reference calls `.state_dict()` on the pickled objects, :185-186) over parameters that
    share storage with the state dict (no duplication in the file); the vocabulary pickled under the reference's class path
    `utils.data.Vocab`, which the reference resolves to its own class and `load_meta_model` here to this package's."""
    folder = '{}/{}'.format(args.save_folder, args.name)
    save_path = folder + ('/best_model.th' if best_model else '/epoch_{}.th'.format(epoch))
    os.makedirs(folder, exist_ok=True)
    print('SAVE MODEL to', save_path)
    logging.info('SAVE MODEL to ' + save_path)
    state = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    params = [torch.nn.Parameter(state[n], requires_grad=True) for n, _ in model.named_parameters()]
    payload = {'vocab': vocab, 'args': args, 'epoch': epoch, 'model_state_dict': state,
               'inner_opt': _export_opt(inner_opt, params), 'outer_opt': _export_opt(outer_opt, params), 'metrics': metrics}
    torch.save(payload, save_path, pickle_module=_ref_path_pickle)
    return save_path


def save_joint_model(model, vocab, epoch, opt, metrics, args, best_model=False):
    """utils/functions.py:43-71: the joint-training `.th` dict ('vocab', 'args', 'epoch', 'model_state_dict', 'opt', 'metrics'),
    written like save_meta_model so that both stacks read it."""
    folder = '{}/{}'.format(args.save_folder, args.name)
    save_path = folder + ('/best_model.th' if best_model else '/epoch_{}.th'.format(epoch))
    os.makedirs(folder, exist_ok=True)
    print('SAVE MODEL to', save_path)
    logging.info('SAVE MODEL to ' + save_path)
    state = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    params = [torch.nn.Parameter(state[n], requires_grad=True) for n, _ in model.named_parameters()]
    payload = {'vocab': vocab, 'args': args, 'epoch': epoch, 'model_state_dict': state, 'opt': _export_opt(opt, params),
               'metrics': metrics}
    torch.save(payload, save_path, pickle_module=_ref_path_pickle)
    return save_path


def _export_opt(opt, params):
    """a torch.optim object of the same kind / hyper-parameters / state over the CPU parameter copies"""
    src = _as_torch_opt(opt)
    sd = src.state_dict()
    for st in sd['state'].values():
        for k, v in list(st.items()):
            if torch.is_tensor(v):
                st[k] = v.detach().cpu().clone()
    out = type(src)(params, lr=src.param_groups[0]['lr'])
    out.load_state_dict(sd)
    return out


class _RefPathPickler(pickle._Pickler):
    """Pickles this package's `Vocab` CLASS under the reference's global name `utils.data.Vocab` (which the reference resolves
    to its own class and `load_*_model` here back to this package's).  The opcode is written directly, so nothing process-global
    (sys.modules, Vocab.__module__) is touched while the trainer's prefetch thread or a user feature_fn may be importing.
    Pure-Python pickler: only the small object graph goes through it, tensor storages are written by torch.save itself."""

    def save_global(self, obj, name=None):
        from .data import Vocab
        if obj is Vocab:
            self.write(pickle.GLOBAL + b'utils.data\nVocab\n')
            self.memoize(obj)
            return
        super().save_global(obj, name)


class _ref_path_pickle:                 # the `pickle_module` duck-type torch.save expects
    __name__ = 'pickle'
    Pickler = _RefPathPickler
    Unpickler = pickle.Unpickler
    load, loads = pickle.load, pickle.loads
    HIGHEST_PROTOCOL, DEFAULT_PROTOCOL = pickle.HIGHEST_PROTOCOL, pickle.DEFAULT_PROTOCOL

    @staticmethod
    def dump(obj, f, protocol=None, **kw):
        _RefPathPickler(f, protocol).dump(obj)

    @staticmethod
    def dumps(obj, protocol=None, **kw):
        import io
        buf = io.BytesIO()
        _RefPathPickler(buf, protocol).dump(obj)
        return buf.getvalue()


class _CompatUnpickler(pickle.Unpickler):
    """Checkpoints written by the reference pickle its `utils.data.Vocab`; resolve it to this package's Vocab (same attributes:
    utils/data.py:1-28) so a reference-trained `.th` file resumes here without the reference on the path."""

    def find_class(self, module, name):
        if (module, name) == ('utils.data', 'Vocab'):
            from .data import Vocab
            return Vocab
        return super().find_class(module, name)


class _compat_pickle:                   # the `pickle_module` duck-type torch.load expects
    __name__ = 'pickle'
    Unpickler = _CompatUnpickler
    load = staticmethod(lambda f, **kw: _CompatUnpickler(f, **kw).load())
    loads = staticmethod(pickle.loads)
    dump, dumps, Pickler = pickle.dump, pickle.dumps, pickle.Pickler
    HIGHEST_PROTOCOL, DEFAULT_PROTOCOL = pickle.HIGHEST_PROTOCOL, pickle.DEFAULT_PROTOCOL


def load_checkpoint_dict(load_path):
    """the raw `.th` dict of either stack on the CPU (reference-written files resolve `utils.data.Vocab` to this package's)"""
    return torch.load(load_path, map_location=torch.device('cpu'), weights_only=False, pickle_module=_compat_pickle)


def load_meta_model(load_path, train=True):
    """-> (model, vocab, inner_opt, outer_opt, epoch, metrics, args); optimizers come back as torch.optim objects holding
    the saved state (TransientTrainer.train converts them to its flat-buffer optimizers)."""
    ckpt = load_checkpoint_dict(load_path)
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


def load_joint_model(load_path, train=True):
    """utils/functions.py:190-218 -> (model, vocab, opt, epoch, metrics, args); the optimizer comes back as torch.optim.Adam."""
    ckpt = load_checkpoint_dict(load_path)
    args, vocab = ckpt['args'], ckpt['vocab']
    model = init_transformer_model(args, vocab, train=train, is_factorized=getattr(args, 'is_factorized', False),
                                   r=getattr(args, 'r', 100))
    model.load_state_dict(ckpt['model_state_dict'])
    model = model.cuda() if getattr(args, 'cuda', False) and torch.cuda.is_available() else model.cpu()
    opt = torch.optim.Adam(model.parameters(), lr=args.lr)
    opt.load_state_dict(ckpt['opt'].state_dict())
    return model, vocab, opt, ckpt['epoch'], ckpt['metrics'], args
