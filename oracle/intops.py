"""CPU oracle for the integer / indexing side of the WaveNet path (TEST INFRASTRUCTURE).

numpy restatement of wavenet/audio_func.py and wavenet/faster_audio_data.py of the reference.
Pinned against tests/golden/g4_data.npz and g5_mulaw.npz (made by tools/make_golden.py from the
imported reference).  Never imported by music_amd/.
"""
import numpy as np


# --------------------------------------------------------------------------------------------
# wavenet/audio_func.py:5-22  mu_law_encode   (canonical encoder)
# --------------------------------------------------------------------------------------------
def mu_law_encode_formula(audio, q=256):
    """Direct float32 restatement of audio_func.py:16-22:
    trunc((sign(a)*log1p(mu*|clip(a,-1,1)|)/log1p(mu) + 1)/2*mu + 0.5).
    NOT bit-exact against torch on ~ppm of inputs (libm log1p differs in the last ulp, Q12); the
    canonical, bit-exact form is ``mu_law_encode_table`` below."""
    a = np.asarray(audio, dtype=np.float32)
    mu = np.float32(q - 1)
    mag = np.log1p(mu * np.abs(np.clip(a, -1.0, 1.0)).astype(np.float32)).astype(np.float32) \
        / np.log1p(mu).astype(np.float32)
    sig = np.sign(a).astype(np.float32) * mag.astype(np.float32)
    enc = (sig + np.float32(1)) / np.float32(2) * mu + np.float32(0.5)
    return enc.astype(np.float32).astype(np.int64)


def mu_law_encode_table(audio, thresholds):
    """code(a) = #{k : thresholds[k] <= a}.  ``thresholds`` is the (q-1,) float32 table of the
    smallest float32 input that the reference encoder maps to code k+1 (found by bisection over
    the float32 ordering in tools/make_golden.py; the reference encoder is monotone).  This is
    the bit-exact definition both the oracle and the HIP kernel implement."""
    a = np.asarray(audio, dtype=np.float32)
    return np.searchsorted(thresholds, a, side="right").astype(np.int64)


def mu_law_encode_torch(audio, q=256):
    """audio_func.py:16-22 with the ATen float32 ops the reference itself calls, op for op (torch is imported here only):
    the same bits as the reference on the same torch build, for ANY q - what the table form is checked against for
    quantization_channels other than 256 (tests/golden/g5q_mulaw.npz pins it to the reference's own outputs)."""
    import torch
    a = torch.as_tensor(np.asarray(audio, dtype=np.float32))
    mu = torch.Tensor([q - 1]).float()
    magnitude = torch.log1p(mu * torch.abs(torch.clamp(a, -1.0, 1.0))) / torch.log1p(mu)
    return ((torch.sign(a) * magnitude + 1) / 2 * mu + 0.5).long().numpy()


def mu_law_decode_torch(codes, q=256):
    """audio_func.py:35-39 with torch float32 ops, op for op."""
    import torch
    mu = torch.Tensor([q - 1]).float()
    signal = 2.0 * (torch.as_tensor(np.asarray(codes)).float() / mu) - 1.0
    return (torch.sign(signal) * ((1.0 / mu) * ((1.0 + mu) ** torch.abs(signal) - 1.0))).numpy()


def mu_law_decode(codes, q=256):
    """audio_func.py:24-39: s = 2*k/mu - 1 ; sign(s) * ((1+mu)^|s| - 1)/mu  in float32."""
    mu = np.float32(q - 1)
    s = np.float32(2.0) * (np.asarray(codes).astype(np.float32) / mu) - np.float32(1.0)
    mag = (np.float32(1.0) / mu) * (np.power(np.float32(1.0) + mu, np.abs(s)).astype(np.float32)
                                    - np.float32(1.0))
    return (np.sign(s) * mag).astype(np.float32)


# --------------------------------------------------------------------------------------------
# wavenet/faster_audio_data.py
# --------------------------------------------------------------------------------------------
def make_data_pieces(data, receptive_field, window_length):
    """faster_audio_data.py:24-40.  ``data`` = list of 1-D int arrays.
    Returns list of (piece int array (rf+win-1,), target int64 array (win,)).

    While len(item) > rf: if a full rf+win span remains, cut piece=item[:rf+win-1],
    target=item[rf:rf+win], advance by win; otherwise advance by rf and RE-APPEND the previous
    piece/target (the reference's `else` branch does not assign new ones).  A short first item
    raises NameError in the reference; here UnboundLocalError's parent NameError is raised too.
    """
    out = []
    have = False
    piece = target = None
    for item in data:
        item = np.asarray(item)
        while len(item) > receptive_field:
            if len(item) >= receptive_field + window_length:
                piece = item[:receptive_field + window_length - 1]
                target = item[receptive_field:receptive_field + window_length]
                item = item[window_length:]
                have = True
            else:
                item = item[receptive_field:]
            if not have:
                raise NameError("name 'target' is not defined")
            out.append((piece, target.astype(np.int64)))
    return out


def one_hot_scrambled_positions(piece, q=256):
    """faster_audio_data.py:62-83: a (T,q) row-major one-hot is *reshaped* (not
    transposed) to (q,T).  The one for sample s therefore lands at flat offset s*q + piece[s] of
    the (q,T) block, i.e. (channel, time) = divmod(s*q + piece[s], T).  Returns (T,) flat
    offsets (int64)."""
    piece = np.asarray(piece).astype(np.int64)
    return np.arange(len(piece), dtype=np.int64) * q + piece


def one_hot_scrambled(piece, q=256):
    """Dense float32 (q,T) array with ones at one_hot_scrambled_positions()."""
    t = len(piece)
    flat = np.zeros(q * t, dtype=np.float32)
    flat[one_hot_scrambled_positions(piece, q)] = 1.0
    return flat.reshape(q, t)


def one_hot_proper(piece, q=256):
    """Textbook one-hot (q,T): out[piece[s], s] = 1 (what fast_generate.py:159-160,170-171 build)."""
    t = len(piece)
    out = np.zeros((q, t), dtype=np.float32)
    out[np.asarray(piece).astype(np.int64), np.arange(t)] = 1.0
    return out
