"""``upfirdn2d`` (mirror of the reference's ``op/upfirdn2d.py:147-158``): same Python signature, the
hand-written gfx950 kernel behind it instead of the JIT-compiled CUDA extension.  Forward only: the
sampling path runs under no_grad (the reference's backward, :21-87, is training code)."""
import torch

from .. import ops


def upfirdn2d(input, kernel, up=1, down=1, pad=(0, 0)):
    """input (N, C, H, W) -> (N, C, H', W'); zero-insert by `up`, pad (pad[0], pad[1]) on both axes,
    FIR with `kernel` (flipped: true convolution), decimate by `down`."""
    if not input.is_cuda:
        raise RuntimeError("upfirdn2d: expected a GPU tensor (this build has no CPU fallback; the reference's "
                           "upfirdn2d_native lives in oracle/resample.py for tests only)")
    batch, channel, in_h, in_w = input.shape
    out = ops.upfirdn2d_raw(input.reshape(-1, in_h, in_w, 1), kernel.to(input.device),
                            up, up, down, down, pad[0], pad[1], pad[0], pad[1])
    return out.view(-1, channel, out.shape[1], out.shape[2])
