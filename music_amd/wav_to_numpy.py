"""Counterpart of the reference's offline data prep ``wavenet/data/wav_to_numpy.py``:
a directory of .wav files -> ``np_audio.pkl`` = a pickled list of int32 mu-law code arrays, the file
``audio_dataset`` reads (wavenet/data/wav_to_numpy.py:25-35).

MI355X-first: the companding runs on the device with the path's CANONICAL encoder
(``audio_func.mu_law_encode`` -> ``wn_mulaw_encode_tbl``, bit-exact against the float32 torch formula of
wavenet/audio_func.py:5-22) instead of a second, float64 numpy formula on the host; whole
files are encoded in one launch each.  librosa (absent here) is replaced by ``scipy.io.wavfile`` +
``scipy.signal.resample_poly`` for reading / mono mix-down / resampling to 16 kHz.  Unlike the
reference module, importing this one has no side effect (the reference runs ``main('/data/...')`` at
import, :37).
"""
import glob
import pickle

import numpy as np
import torch

try:
    from .audio_func import mu_law_encode
except ImportError:
    from music_amd.audio_func import mu_law_encode


def encode_waveforms(waveforms, quantization_channels=256):
    """list of float arrays (any length, values clipped to [-1, 1] by the encoder) -> list of int32 code arrays."""
    out = []
    for w in waveforms:
        a = torch.as_tensor(np.ascontiguousarray(w, dtype=np.float32))
        out.append(mu_law_encode(a, quantization_channels).to(torch.int32).cpu().numpy())
    return out


def load_wav(path, sr=16000):
    """Mono float32 waveform in [-1, 1] at ``sr`` Hz (what ``librosa.load(path, sr=16000, mono=True)[0]`` is used for)."""
    from scipy.io import wavfile
    from scipy.signal import resample_poly
    rate, data = wavfile.read(path)
    if data.dtype.kind == 'i':
        data = data.astype(np.float32) / float(np.iinfo(data.dtype).max + 1)
    elif data.dtype.kind == 'u':                                  # 8-bit PCM is unsigned
        data = (data.astype(np.float32) - 128.0) / 128.0
    else:
        data = data.astype(np.float32)
    if data.ndim == 2:
        data = data.mean(axis=1)
    if rate != sr:
        g = np.gcd(int(rate), int(sr))
        data = resample_poly(data, sr // g, rate // g).astype(np.float32)
    return data


def main(audio_dir, suffix='.wav'):
    """wavenet/data/wav_to_numpy.py:25-35: every ``audio_dir*suffix`` file -> ``audio_dir + "np_audio.pkl"``."""
    file_list = sorted(glob.glob(audio_dir + '*' + suffix))
    audio_list = encode_waveforms([load_wav(f) for f in file_list])
    with open(audio_dir + "np_audio.pkl", 'wb') as output:
        pickle.dump(audio_list, output)
    return audio_list
