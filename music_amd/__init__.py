"""music_amd — MI355X-native (gfx950) implementation of the WaveNet hot path of
deep-art-project/Music, behind the reference's own Python module surface.

    music_amd.model            <- wavenet/model.py            (wavenet, predict_next)
    music_amd.engine           host-side plan over the C ABI (libwavenet_hip.so)
    music_amd._lib             ctypes binding of include/wavenet_hip.h
"""
__all__ = ["model", "engine"]
