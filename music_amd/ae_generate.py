"""Counterpart of the reference's ``wavenet_autoencoder/generate.py``: naive generation by a full
forward on a sliding window of ``receptive_field + 512`` samples per generated sample
(generate.py:13-65).  The reference version cannot run as shipped (``librosa`` used without import
:65, tensors instead of ints appended :48, ``.cuda()`` hard-wired :55); the algorithm is kept, the
output is written with scipy.  It is O(T * rf): a harness, not a kernel (SURVEY 8f3 lists the
cached-queue replacement as future work)."""
import json
import os

import numpy as np
import torch

try:
    from .audio_func import mu_law_decode
    from .model1 import wavenet_autoencoder
    from .train import load_model
except ImportError:
    from music_amd.audio_func import mu_law_decode
    from music_amd.model1 import wavenet_autoencoder
    from music_amd.train import load_model


def predict_next(net, input_wav, quantization_channel=256):
    """generate.py:13-19 — argmax over the last chunk-row of the forward output."""
    with torch.no_grad():
        out = net(input_wav).view(-1, quantization_channel)
    return int(torch.topk(out[-1, :].view(-1), 1)[1])


def generate(model_path, model_name, generate_path, generate_name, start_piece=None, sr=16000, duration=10):
    if os.path.exists(generate_path) is False:
        os.makedirs(generate_path)
    with open('./params/model_params.json') as f:
        model_params = json.load(f)
    net = wavenet_autoencoder(**model_params)
    net = load_model(net, model_path, model_name)
    if net is None:
        raise FileNotFoundError(model_path + model_name)
    net = net.cuda()
    win = net.receptive_field + 512
    if start_piece is None:
        start_piece = torch.zeros(1, 256, win)
        start_piece[:, 128, :] = 1.0
    input_wav = start_piece.cuda()
    generated = []
    for i in range(duration * sr):
        code = predict_next(net, input_wav)
        generated.append(code)
        note = torch.zeros(1, net.quantization_channel, 1, device=input_wav.device)
        note[0, code, 0] = 1.0
        input_wav = torch.cat((input_wav[:, :, -(win - 1):], note), 2)
    audio = mu_law_decode(torch.tensor(generated, dtype=torch.int64), net.quantization_channel).cpu().numpy()
    from scipy.io import wavfile
    wavfile.write(generate_path + generate_name, sr, audio.astype(np.float32))
    return generated
