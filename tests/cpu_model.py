"""TEST-ONLY stand-ins so the host logic (train loop, logs, checkpoints, data parallel averaging)
can be exercised on a machine without a GPU: a `wavenet` whose forward is the CPU oracle, and the
oracle's one-hot builder.  Never used by the product (music_amd has no CPU path)."""
import numpy as np
import torch

from music_amd.model import wavenet as _hip_wavenet
from oracle import intops
from oracle import wavenet_oracle as wo


class OracleWavenet(_hip_wavenet):
    def forward(self, wave_sample):
        if wave_sample.size(2) - self.receptive_field + 1 <= 0:
            raise ValueError("wave sample not long enough")
        return wo.wavenet_forward(dict(self.named_parameters()), self.dilations, wave_sample,
                                  self.filter_width, self.quantization_channels)


def onehot_oracle(codes, quantization_channels=256, scrambled=True):
    fn = intops.one_hot_scrambled if scrambled else intops.one_hot_proper
    return torch.from_numpy(np.stack([fn(r.numpy(), quantization_channels) for r in codes]))


from music_amd.model1 import wavenet_autoencoder as _hip_autoencoder


class OracleAutoencoder(_hip_autoencoder):
    """TEST-ONLY: the autoencoder module with the CPU oracle as its forward (same constructor, same per-forward
    conditioning draws), so that ae_train / ae_generate host logic runs without a GPU."""

    def forward(self, wave_sample):
        if wave_sample.size(2) - self.receptive_field + 1 <= 0:
            raise ValueError("wave sample not long enough")
        cond = self._draw_conditioning()
        probs, enc = wo.autoencoder_forward(dict(self.named_parameters()), self.dilations, wave_sample,
                                            self.en_pool_kernel_size, cond, self.filter_width, self.quantization_channel)
        self.last_encoding = enc.detach()
        return probs

    def cuda(self, *a, **k):
        return self
