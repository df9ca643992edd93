"""MI355X-native meta-transfer training step (drop-in for the `--copy-grad` hot path of audioku/meta-transfer-learning).

Import as `mtl_amd` (see mtl_amd.py at the repository root; the directory name carries a hyphen).
"""
from . import _lib  # noqa: F401
from .data import (Vocab, SyntheticTask, ManifestTaskDataset, SpectrogramDataset, BucketingSampler, AudioDataLoader, SpectrogramFrontEnd, load_vocab, load_wav_pcm16, synthetic_vocab,  # noqa: F401
                   synth_batch)
from .functions import (init_transformer_model, save_meta_model, load_meta_model, save_joint_model, load_joint_model,  # noqa: F401
                        post_process)
from .metrics import calculate_metrics, calculate_cer  # noqa: F401
from .model import Transformer, Encoder, Decoder  # noqa: F401
from .trainer import TransientTrainer, JointTrainer, FlatAdam, FlatSGD  # noqa: F401
from . import dist  # noqa: F401
from . import hostenv  # noqa: F401
from . import lm  # noqa: F401
