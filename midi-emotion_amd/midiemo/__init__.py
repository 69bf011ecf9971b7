"""midiemo -- MI355X-native hot path of serkansulun/midi-emotion.

Host code is Python on PyTorch-ROCm (tensors, streams, torch.distributed);
all compute of the transformer path runs in libmidiemo_hip.so (hand-written
HIP for gfx950, C-ABI in include/midiemo.h).  There is no CPU fallback."""
__all__ = ["build_model", "ops"]


def build_model(*args, **kwargs):
    from .models.build_model import build_model as _bm
    return _bm(*args, **kwargs)
