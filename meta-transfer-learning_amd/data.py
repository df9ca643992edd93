"""Vocabulary, label/manifest loaders and the synthetic task source.

Mirrors: `Vocab` (utils/data.py:1-28), the label-JSON loading of meta_transfer_train.py:151-157, the manifest CSV
contract of utils/data_loader.py:191-195 and the batch layout returned by `SpectrogramDataset.sample`
(utils/data_loader.py:245-321).  Audio decoding / STFT (utils/data_loader.py:65-96) is outside the accelerated path
(SURVEY.md 8(f) f1): datasets here get features from a caller-supplied `feature_fn` or synthesize them.
"""
import csv
import json

import numpy as np
import torch


class Vocab(object):
    def __init__(self):
        self.PAD_TOKEN, self.SOS_TOKEN, self.EOS_TOKEN, self.OOV_TOKEN = '<PAD>', '<SOS>', '<EOS>', '<OOV>'
        self.PAD_ID, self.SOS_ID, self.EOS_ID, self.OOV_ID = 0, 1, 2, 3
        self.special_token_list = [self.PAD_TOKEN, self.SOS_TOKEN, self.EOS_TOKEN, self.OOV_TOKEN]
        self.token2id, self.id2token = {}, []
        self.label2id, self.id2label = {}, []
        for tok in self.special_token_list:
            self.add_token(tok)
            self.add_label(tok)

    def add_token(self, token):
        if token not in self.token2id:
            self.token2id[token] = len(self.token2id)
            self.id2token.append(token)

    def add_label(self, label):
        if label not in self.label2id:
            self.label2id[label] = len(self.label2id)
            self.id2label.append(label)


def load_vocab(labels_path):
    """labels JSON (a list of characters, e.g. data/labels/hkust_seame_labels.json) -> Vocab; specials first."""
    with open(labels_path, encoding='utf-8') as f:
        labels = json.load(f)
    vocab = Vocab()
    for label in labels:
        vocab.add_token(label)
        vocab.add_label(label)
    return vocab


def synthetic_vocab(size):
    """`size` ids in total (4 specials + size-4 distinct CJK code points), matching the reference's 3765 when size=3765."""
    vocab = Vocab()
    for i in range(size - 4):
        ch = chr(0x4e00 + i)
        vocab.add_token(ch)
        vocab.add_label(ch)
    return vocab


def read_manifest(path):
    """CSV without header, rows `wav_path,txt_path` (data/manifests/*.csv) -> list of [wav, txt]."""
    with open(path, newline='', encoding='utf-8') as f:
        return [row[:2] for row in csv.reader(f) if row]


def parse_transcript(vocab, transcript_path):
    """utils/data_loader.py:342-361 (char input): ' ' + lower-cased text -> ids of known labels (unknown chars and id 0 dropped)."""
    if transcript_path[-4:] == '.txt':
        with open(transcript_path, 'r', encoding='utf8') as f:
            text = ' ' + f.read().replace('\n', '').lower()
    else:
        text = transcript_path.replace('\n', '').lower()
    return [i for i in (vocab.label2id.get(ch) for ch in text) if i]


def collate(spects, transcripts, pad_id=0):
    """Zero-pad features to the batch max T and PAD-pad targets: the 5-tuple of utils/data_loader.py:284-297."""
    k = len(spects)
    max_t = max(s.size(1) for s in spects)
    freq = spects[0].size(0)
    max_l = max(len(t) for t in transcripts)
    inputs = torch.zeros(k, 1, freq, max_t)
    input_sizes = torch.zeros(k, dtype=torch.int32)
    input_percentages = torch.zeros(k, dtype=torch.float32)
    targets = torch.full((k, max_l), pad_id, dtype=torch.int64)
    target_sizes = torch.zeros(k, dtype=torch.int32)
    for i, (s, t) in enumerate(zip(spects, transcripts)):
        n = s.size(1)
        inputs[i, 0, :, :n] = s
        input_sizes[i] = n
        input_percentages[i] = n / float(max_t)
        targets[i, :len(t)] = torch.tensor(t, dtype=torch.int64)
        target_sizes[i] = len(t)
    return inputs, input_sizes, input_percentages, targets, target_sizes


class ManifestTaskDataset:
    """`.sample(k_train, k_valid, manifest_id)` over manifest CSVs like SpectrogramDataset (utils/data_loader.py:171-321).

    feature_fn(wav_path) -> (F, T) float tensor replaces `parse_audio`; sampling uses np.random.choice with the same
    per-manifest probabilities (uniform, or uniform over the leading `partitions[i]` fraction).
    seed=None draws from the global np.random like the reference (seeded once by the entry script,
    meta_transfer_train.py:109-112); seed=<int> gives the dataset its own RandomState -- with several ranks every rank must
    see the same draws (tasks are sharded AFTER sampling, the validation batch is shared), so pass the same seed on all ranks.
    `__getitem__` / `__len__` follow utils/data_loader.py:323-340 (validation / test use: manifest 0 only unless is_train)."""

    def __init__(self, vocab, args, manifest_filepath_list, feature_fn=None, partitions=None, seed=None, is_train=False):
        if feature_fn is None:
            # default: 16-bit PCM wav -> device spectrogram front-end (SpectrogramParser.parse_audio, normalize=True as in
            # meta_transfer_train.py:161), handed back on the host like the reference's parse_audio output
            fe = SpectrogramFrontEnd(args.sample_rate, args.window_size, args.window_stride, getattr(args, 'window', 'hamming'), True)
            feature_fn = lambda path: fe(load_wav_pcm16(path)).cpu()
        self.vocab, self.args, self.feature_fn = vocab, args, feature_fn
        self.ids_list = [read_manifest(p) for p in manifest_filepath_list]
        self.rng = np.random if seed is None else np.random.RandomState(seed)
        self.is_train = is_train
        self.max_size = max(len(ids) for ids in self.ids_list) * len(self.ids_list)
        if is_train and len(self.ids_list) > 1:
            self.max_size = 30000                                   # utils/data_loader.py:198-203
        self.proba = []
        for i, ids in enumerate(self.ids_list):
            if partitions is not None:
                part = max(int(len(ids) * partitions[i]), 1)
                p = np.zeros(len(ids))
                p[:part] = 1 / part
            else:
                p = np.full(len(ids), 1 / len(ids))
            self.proba.append(p)

    def _rows(self, ids, picks):
        spects, trans = [], []
        for j in picks:
            wav, txt = ids[j][0], ids[j][1]
            spects.append(self.feature_fn(wav)[:, :self.args.src_max_len])
            trans.append(parse_transcript(self.vocab, txt))
        return spects, trans

    def sample(self, k_train, k_val, manifest_id, need=(True, True)):
        """utils/data_loader.py:245-321.  need = (train part, validation part): a part that the caller will not use is still DRAWN
        (the index stream stays what the reference's is, and identical on every rank) but not loaded / featurised / collated --
        it is returned as None.  The meta loop uses only the LAST task's validation batch (transient_trainer.py:168) and, with
        several ranks, only its own tasks' training batches."""
        ids = self.ids_list[manifest_id]
        picks = self.rng.choice(np.arange(0, len(ids)), k_train + k_val, p=self.proba[manifest_id], replace=True)
        tr = collate(*self._rows(ids, picks[:k_train]), pad_id=self.vocab.PAD_ID) if need[0] else None
        va = collate(*self._rows(ids, picks[k_train:k_train + k_val]), pad_id=self.vocab.PAD_ID) if need[1] else None
        return tr, va

    def __len__(self):
        return self.max_size

    def __getitem__(self, index):
        if self.is_train:
            ids = self.ids_list[index % len(self.ids_list)]
            row = ids[(index // len(self.ids_list)) % len(ids)]
        else:
            ids = self.ids_list[0]
            row = ids[index % len(ids)]
        return self.feature_fn(row[0])[:, :self.args.src_max_len], parse_transcript(self.vocab, row[1])


class SpectrogramDataset(ManifestTaskDataset):
    """utils/data_loader.py:171-236 with the reference's OWN constructor, so that its entry script builds the datasets unchanged
    (meta_transfer_train.py:159-175):

        SpectrogramDataset(vocab, args, audio_conf, manifest_filepath_list=..., normalize=True, augment=args.augment,
                           input_type=args.input_type, is_train=True, partitions=args.train_partition_list)

    audio_conf: dict(sample_rate, window_size, window_stride, window, noise_dir, noise_prob, noise_levels) (:141-147).  Features come
    from the device front-end (SpectrogramFrontEnd = SpectrogramParser.parse_audio, :65-96) unless `feature_fn` is given.  Same
    attributes as the reference object (max_size, ids_list, proba, part_len, input_type, manifest_filepath_list, is_train) and the
    same two console lines.  Outside the accelerated path and rejected loudly: augment=True (sox tempo / gain perturbation),
    noise injection (audio_conf['noise_dir']), input_type other than 'char' (the bpe / ipa branches are commented out in the
    reference too)."""

    def __init__(self, vocab, args, audio_conf, manifest_filepath_list, normalize=False, augment=False, input_type='char',
                 is_train=False, partitions=None, feature_fn=None, seed=None):
        if augment:
            raise NotImplementedError('augment=True (sox tempo / gain perturbation, utils/data_loader.py:28-38) is outside the accelerated path')
        if audio_conf.get('noise_dir') is not None:
            raise NotImplementedError("noise injection (audio_conf['noise_dir']) is outside the accelerated path")
        if input_type != 'char':
            raise NotImplementedError("only input_type='char' (utils/data_loader.py:342-361)")
        self.window_stride, self.window_size = audio_conf['window_stride'], audio_conf['window_size']
        self.sample_rate, self.window = audio_conf['sample_rate'], audio_conf.get('window', 'hamming')
        self.normalize, self.augment, self.noise_prob = normalize, augment, audio_conf.get('noise_prob')
        if feature_fn is None:
            fe = []       # built at the first utterance: constructing the dataset must not need the device

            def feature_fn(path):
                if not fe:
                    fe.append(SpectrogramFrontEnd(self.sample_rate, self.window_size, self.window_stride,
                                                  self.window if self.window in ('hamming', 'hann', 'blackman', 'bartlett') else 'hamming',
                                                  self.normalize))
                return fe[0](load_wav_pcm16(path)).cpu()
        super().__init__(vocab, args, manifest_filepath_list, feature_fn=feature_fn, partitions=partitions, seed=seed, is_train=is_train)
        self.manifest_filepath_list, self.input_type = manifest_filepath_list, input_type
        # (the reference leaves part_len at the LAST manifest's partition size, or max_size without partitions: :211-222)
        self.part_len = (max(int(len(self.ids_list[-1]) * partitions[len(self.ids_list) - 1]), 1) if partitions is not None
                         else self.max_size)
        print('max_size:', self.max_size)
        print('input_type:', input_type)

    def parse_transcript(self, transcript_path):
        return parse_transcript(self.vocab, transcript_path)

    def parse_audio(self, audio_path):
        return self.feature_fn(audio_path)


class BucketingSampler(torch.utils.data.Sampler):
    """utils/data_loader.py:480-500: consecutive bins of `batch_size` indices (the data is assumed sorted by length); iterating
    shuffles INSIDE each bin, shuffle(epoch) shuffles the ORDER of the bins -- both with the global np.random like the reference."""

    def __init__(self, data_source, batch_size=1):
        self.data_source = data_source
        ids = list(range(0, len(data_source)))
        self.bins = [ids[i:i + batch_size] for i in range(0, len(ids), batch_size)]

    def __iter__(self):
        for ids in self.bins:
            np.random.shuffle(ids)
            yield ids

    def __len__(self):
        return len(self.bins)

    def shuffle(self, epoch):
        np.random.shuffle(self.bins)


class AudioDataLoader(torch.utils.data.DataLoader):
    """utils/data_loader.py:401-440: a DataLoader over (spectrogram (F,T), transcript ids) items whose batches are sorted by
    descending frame count, zero-padded to the longest utterance / PAD-padded to the longest transcript, and returned as
    (inputs (B,1,F,T), targets (B,L) int64, input_percentages (B) f32, input_sizes (B) i32, target_sizes (B) i32) -- the layout
    the in-loop validation of TransientTrainer / JointTrainer consumes."""

    def __init__(self, pad_token_id, *args, **kwargs):
        self.pad_token_id = pad_token_id
        kwargs['collate_fn'] = self._collate
        super().__init__(*args, **kwargs)

    def _collate(self, batch):
        batch = sorted(batch, key=lambda sample: sample[0].size(1), reverse=True)
        inputs, input_sizes, input_percentages, targets, target_sizes = collate(
            [s for s, _ in batch], [t for _, t in batch], pad_id=self.pad_token_id)
        return inputs, targets, input_percentages, input_sizes, target_sizes


def synth_batch(seed, k, T, L, vocab_size, variable=False, freq_bins=161):
    """Seeded synthetic batch: N(0,1) 'spectrogram', labels in [4, V); optional ragged lengths."""
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(k, 1, freq_bins, T, generator=g)
    y = torch.randint(4, vocab_size, (k, L), generator=g)
    lens = torch.full((k,), T, dtype=torch.int32)
    if variable:
        lens = torch.randint(max(T // 8, 1), T + 1, (k,), generator=g).to(torch.int32)
        lens[0] = T
        if k > 1:
            lens[-1] = max(T // 8, 1)
        tl = torch.randint(max(L // 2, 1), L + 1, (k,), generator=g)
        tl[0] = L
        for i in range(k):
            x[i, :, :, int(lens[i]):] = 0
            y[i, int(tl[i]):] = 0
    return x, lens, y


class SyntheticTask:
    """Duck-types the dataset contract of `TransientTrainer.train`: seeded synthetic (train, valid) batches per call."""

    def __init__(self, task_id, k, T, L, vocab_size, variable=False, pin=False):
        self.task_id, self.k, self.T, self.L, self.V, self.variable, self.pin = task_id, k, T, L, vocab_size, variable, pin
        self.calls = 0

    def sample(self, k_train, k_valid, manifest_id):
        it = self.calls
        self.calls += 1
        out = []
        for part, k in ((0, k_train), (1, k_valid)):
            x, lens, y = synth_batch(1000 * it + 10 * self.task_id + part, k, self.T, self.L, self.V, self.variable)
            if self.pin:
                x = x.pin_memory()
            out.append((x, lens, lens.float() / self.T, y, (y != 0).sum(1).to(torch.int32)))
        return tuple(out)


# ------------------------------------------------------------------------------------------------------------------
# Spectrogram front-end on the device: SpectrogramParser.parse_audio, utils/data_loader.py:65-96
# ------------------------------------------------------------------------------------------------------------------
def load_wav_pcm16(path):
    """16-bit PCM .wav -> float32 mono in [-1, 1) (utils/audio.py:7-15 uses torchaudio.load(normalization=True); channels averaged)."""
    import wave
    with wave.open(path, 'rb') as w:
        if w.getsampwidth() != 2:
            raise ValueError('only 16-bit PCM wav files are supported here')
        raw = np.frombuffer(w.readframes(w.getnframes()), dtype='<i2').astype(np.float32) / 32768.0
        ch = w.getnchannels()
    return raw.reshape(-1, ch).mean(axis=1).astype(np.float32) if ch > 1 else raw


class SpectrogramFrontEnd:
    """wav -> STFT (n_fft = win = sample_rate*window_size, hop = sample_rate*window_stride, symmetric Hamming window,
    center + reflect padding = librosa.stft defaults of the reference era) -> |.| -> log1p -> (x-mean)/std, all on the MI355X:
    the STFT is one fp32-MFMA GEMM (frames = overlapping rows of the padded waveform, lda = hop) against a windowed DFT basis."""

    def __init__(self, sample_rate=16000, window_size=0.02, window_stride=0.01, window='hamming', normalize=True, device='cuda'):
        from scipy.signal import windows as sw
        self.n_fft = int(sample_rate * window_size)
        self.hop = int(sample_rate * window_stride)
        self.F = self.n_fft // 2 + 1
        self.normalize = normalize
        self.device = torch.device(device)
        table = {'hamming': sw.hamming, 'hann': sw.hann, 'blackman': sw.blackman, 'bartlett': sw.bartlett}
        win = table[window](self.n_fft)                       # the reference passes the scipy FUNCTION -> symmetric window
        n = np.arange(self.n_fft)[:, None].astype(np.float64)
        f = np.arange(self.F)[None, :].astype(np.float64)
        ang = 2.0 * np.pi * n * f / self.n_fft
        self.ldb = (2 * self.F + 3) // 4 * 4
        basis = np.zeros((self.n_fft, self.ldb), dtype=np.float32)
        basis[:, :self.F] = (win[:, None] * np.cos(ang)).astype(np.float32)
        basis[:, self.F:2 * self.F] = (-win[:, None] * np.sin(ang)).astype(np.float32)
        self.basis = torch.from_numpy(basis).to(self.device)
        self.partials = torch.empty(512, dtype=torch.float64, device=self.device)

    def __call__(self, y):
        """y: 1-D float waveform (numpy or tensor) -> (F, T) fp32 tensor on the device, T = 1 + len(y) // hop."""
        from . import _lib
        if self.device.type != 'cuda':
            raise RuntimeError('the spectrogram front-end runs on the MI355X only (no CPU fallback)')
        lib = _lib.lib()
        y = np.asarray(y.detach().cpu() if torch.is_tensor(y) else y, dtype=np.float32).reshape(-1)
        pad = self.n_fft // 2
        yp = torch.from_numpy(np.pad(y, (pad, pad), mode='reflect')).to(self.device)
        T = 1 + y.shape[0] // self.hop
        reim = torch.empty(T, self.ldb, device=self.device)
        out = torch.empty(self.F, T, device=self.device)
        st = torch.cuda.current_stream(self.device).cuda_stream
        _lib.check(lib.mtl_gemm_f32(st, 0, 0, T, 2 * self.F, self.n_fft, 1.0, yp.data_ptr(), self.hop, self.basis.data_ptr(), self.ldb,
                                    reim.data_ptr(), self.ldb, None, None, 0, 0, 1, 1, 0, 0, 0, 0, 0, 0, 0, None, 0), 'stft gemm')
        _lib.check(lib.mtl_spect_logmag(st, reim.data_ptr(), self.ldb, T, self.F, out.data_ptr(), self.partials.data_ptr(),
                                        1 if self.normalize else 0), 'mtl_spect_logmag')
        return out
