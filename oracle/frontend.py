"""ORACLE (test infrastructure): numpy restatement of the reference's spectrogram front-end
(`SpectrogramParser.parse_audio`, utils/data_loader.py:65-96).

PARITY UNPINNED: the reference computes the STFT with librosa, which is not installed in the build container and is not
part of /root/reference; this file restates librosa.stft's documented algorithm for the reference's call
(`librosa.stft(y, n_fft=N, hop_length=H, win_length=N, window=scipy.signal.hamming)`, defaults center=True,
pad_mode='reflect' of the librosa releases contemporary with the reference, complex64 output) and cannot be checked against
librosa itself here.  Anchors: the call site above and scipy.signal.hamming(N) (symmetric, because a callable is passed).
"""
import numpy as np
import torch
from scipy.signal import windows


def stft_magnitude(y, n_fft, hop):
    y = np.asarray(y, dtype=np.float32)
    win = windows.hamming(n_fft).astype(np.float32)                 # callable window -> scipy default sym=True
    yp = np.pad(y, n_fft // 2, mode='reflect')                      # center=True, pad_mode='reflect'
    n_frames = 1 + (len(yp) - n_fft) // hop
    frames = np.lib.stride_tricks.as_strided(yp, shape=(n_frames, n_fft), strides=(yp.strides[0] * hop, yp.strides[0]))
    spec = np.fft.rfft(frames * win[None, :], axis=1).astype(np.complex64)   # librosa returns complex64
    return np.abs(spec).T                                            # (1 + n_fft/2, n_frames)


def parse_audio(y, sample_rate=16000, window_size=0.02, window_stride=0.01, normalize=True):
    n_fft = int(sample_rate * window_size)
    hop = int(sample_rate * window_stride)
    spect = torch.FloatTensor(np.log1p(stft_magnitude(y, n_fft, hop)))     # data_loader.py:84-88
    if normalize:                                                          # :90-94
        mean, std = spect.mean(), spect.std()
        spect.add_(-mean)
        spect.div_(std)
    return spect
