"""Launch plumbing shared by the ctypes wrappers: the caller's current HIP stream and device, read the cheap way.

`torch.cuda.current_stream(dev).cuda_stream` builds a Stream object (~11 us) and `with torch.cuda.device(dev)` resolves the
device index twice (~8 us) -- seven of those per eager training step were 50-70 us of host time (tools/host_profile.py).  The raw
calls below are what they end up in."""
import ctypes as C

import torch

_get_device = torch._C._cuda_getDevice
_raw_stream = torch._C._cuda_getCurrentRawStream


def stream_of(device):
    """c_void_p of the current stream of `device` (a torch.device of type "cuda"); follows torch.cuda.stream(...) and graph capture."""
    idx = device.index
    return C.c_void_p(_raw_stream(_get_device() if idx is None else idx))


class _Noop:
    def __enter__(self):
        return None

    def __exit__(self, *exc):
        return False


_NOOP = _Noop()


def device_ctx(device):
    """`with device_ctx(dev):` == `with torch.cuda.device(dev):`, free when `dev` already is the current device."""
    idx = device.index
    return _NOOP if (idx is None or idx == _get_device()) else torch.cuda.device(device)
