"""Counterpart of the reference's ``wavenet/faster_audio_data.py``.

Same names and arguments (``audio_dataset``, ``audio_data_loader``, ``one_hot_encode``), same piece
chopping including the tail re-append quirk (faster_audio_data.py:24-40) and the same
"scrambled" one-hot layout (faster_audio_data.py:62-83) — but MI355X-first: the dataset
keeps the int32 codes, a batch crosses PCIe as B x T int32 (128 KB at 8 x 16000) instead of
B x 256 x T float32 (131 MB), and the one-hot is written directly in HBM by ``wn_onehot``.
``num_workers`` is accepted and ignored (there is no per-item CPU work left to parallelise).
"""
import pickle

import torch
from torch.utils.data import DataLoader, Dataset

try:
    from . import _lib
except ImportError:
    from music_amd import _lib


class audio_dataset(Dataset):

    def __init__(self, audio_path, receptive_field, window_length, cuda_available=False,
                 quantization_channels=256):
        self.audio_path = audio_path
        self.receptive_field = receptive_field
        self.window_length = window_length
        self.cuda_available = cuda_available
        self.quantization_channels = quantization_channels
        with open(self.audio_path, 'rb') as f:
            data = pickle.load(f)
        self.data = self._make_data_pieces(data)

    def _make_data_pieces(self, data):
        """faster_audio_data.py:24-40.  A remainder longer than rf but shorter than rf+window
        advances by rf and appends the PREVIOUS piece/target again (the reference's `else` branch
        assigns nothing new); a first item that short raises NameError, as the reference does."""
        rf, win = self.receptive_field, self.window_length
        pieces = []
        piece = target = None
        for item in data:
            item = torch.from_numpy(item)
            while len(item) > rf:
                if len(item) >= rf + win:
                    piece = item[:rf + win - 1]
                    target = item[rf:rf + win].long()
                    item = item[win:]
                else:
                    item = item[rf:]
                if target is None:
                    raise NameError("name 'target' is not defined")
                pieces.append({'audio_piece': piece, 'audio_target': target})
        return pieces

    def __len__(self):
        return len(self.data)

    def __getitem__(self, idx):
        # integer codes only; the one-hot is built on the device per batch (see _collate)
        return self.data[idx]


def onehot_device(codes, quantization_channels=256, scrambled=True):
    """int (B,T) codes -> float32 (B,Q,T) one-hot in HBM.  scrambled=True is the loader's layout."""
    if not torch.cuda.is_available():
        raise RuntimeError("music_amd.faster_audio_data builds the one-hot on an MI355X only; there is no CPU path")
    c = codes.to(device="cuda", dtype=torch.int32, non_blocking=True).contiguous()
    b, t = c.shape
    out = torch.empty(b, quantization_channels, t, dtype=torch.float32, device="cuda")
    _lib.call("wn_onehot", _lib.ptr(c), _lib.ptr(out), b, quantization_channels, t, 1 if scrambled else 0, _lib.stream())
    # the codes ride along with the tensor: the engine forms the causal layer's weight gradient from them (a scatter)
    # as long as the tensor has not been modified since (music_amd/engine.py: forward_logits)
    out._wn_codes = (c, bool(scrambled), out._version, c._version)
    return out


def one_hot_encode(sample_piece, cuda_available=False, quantization_channels=256):
    """faster_audio_data.py:62-83 for ONE item: dict -> dict with a (Q,T) float32 tensor (on the
    device) in the reference's reshape-not-transpose layout."""
    piece, target = sample_piece['audio_piece'], sample_piece['audio_target']
    return {"audio_piece": onehot_device(piece[None], quantization_channels)[0], "audio_target": target}


def shard_bounds(n_items, rank, world):
    """DataParallel's scatter (torch.chunk): chunks of ceil(n/world) items in rank order, the last
    non-empty chunk takes the remainder and trailing ranks may get nothing.  Returns [lo, hi)."""
    c = -(-n_items // world)
    lo = min(rank * c, n_items)
    return lo, min(lo + c, n_items)


class _Stager:
    """Host -> device for the loader's two small integer tensors without stalling the launch stream: the batch is gathered
    into one of three rotating PINNED buffers and copied on a copy stream; the caller's stream only waits for that copy's
    event.  (A `.cuda()` of pageable memory is a synchronous copy queued BEHIND the previous step's kernels: the host then sits
    out the whole step before it can prepare the next one - 6.1 instead of 4.5 ms per step through train.py at 8 x 16000.)"""

    def __init__(self):
        self.bufs, self.k, self.stream = {}, 0, None

    def to_device(self, parts, dtype):
        """parts: list of equally long 1-D CPU tensors -> (len(parts), n) tensor of `dtype` on the device"""
        n = parts[0].numel()
        key = (len(parts), n, dtype)
        ring = self.bufs.get(key)
        if ring is None:
            ring = self.bufs[key] = [[torch.empty(len(parts), n, dtype=dtype).pin_memory(), None] for _ in range(3)]
        slot = ring[self.k % 3]
        self.k += 1
        pin, ev = slot
        if ev is not None:
            ev.synchronize()                     # the copy that last read this buffer has left it
        pn = pin.numpy()
        for i, t in enumerate(parts):            # plain memcpys (numpy): no OpenMP region
            pn[i] = t.numpy()
        if self.stream is None:
            self.stream = torch.cuda.Stream()
        main = torch.cuda.current_stream()
        with torch.cuda.stream(self.stream):
            dev = pin.to("cuda", non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(self.stream)
        slot[1] = ev
        main.wait_event(ev)
        dev.record_stream(main)
        return dev


class _Collate:
    def __init__(self, q, shard=None):
        self.q = q
        self.shard = shard
        self.stager = _Stager()

    def __call__(self, items):
        scale = 1.0
        if self.shard is not None:
            # DataParallel's scatter: contiguous chunk r of the global batch goes to replica r.  On the ragged last
            # batch of an epoch the chunks differ in size (a rank may get none): "dp_scale" = n_local * world / n_global
            # is the weight that turns `mean over ranks of (scale_r * local-mean gradient)` back into DataParallel's
            # global-batch mean; a rank with no items gets audio_piece = None and must still join the all-reduce.
            r, w = self.shard
            lo, hi = shard_bounds(len(items), r, w)
            scale = (hi - lo) * w / float(len(items))
            items = items[lo:hi]
        if not items:
            return {"audio_piece": None, "audio_target": None, "dp_scale": 0.0}
        same = len({it['audio_piece'].numel() for it in items}) == 1 and len({it['audio_target'].numel() for it in items}) == 1
        if same and torch.cuda.is_available():
            codes = self.stager.to_device([it['audio_piece'].to(torch.int32) for it in items], torch.int32)
            target = self.stager.to_device([it['audio_target'] for it in items], torch.int64)
        else:                                    # (torch.stack raises on ragged items, as in the reference; no device: onehot_device raises)
            codes = torch.stack([it['audio_piece'] for it in items]).to(torch.int32)
            target = torch.stack([it['audio_target'] for it in items])
            target = target.cuda(non_blocking=True) if torch.cuda.is_available() else target
        batch = {"audio_piece": onehot_device(codes, self.q), "audio_target": target}
        if self.shard is not None:
            batch["dp_scale"] = scale
        return batch


def audio_data_loader(batch_size, shuffle, num_workers, pin_memory, shard=None, **kwargs):
    """faster_audio_data.py:51-59.  Yields {"audio_piece": float32 (B,256,T) on the device,
    "audio_target": int64 (B,win) on the device}.  Shuffling is torch's own RandomSampler, i.e. the
    same permutation stream as the reference under the same seed.  ``shard=(rank, world)`` (set by
    train.py under torchrun) makes this process build only its contiguous chunk of every global
    batch; all ranks must share the torch seed so they draw the same permutation."""
    audioDataset = audio_dataset(**kwargs)
    print("{} pieces in total".format(len(audioDataset)))
    return DataLoader(audioDataset, batch_size=batch_size, shuffle=shuffle, num_workers=0,
                      pin_memory=False, collate_fn=_Collate(audioDataset.quantization_channels, shard))
