"""numpy restatement of the k-space side of the path (oracle; test infrastructure only).

Reference anchors (paths relative to /root/reference):
  generate_mask        ncsn/linear_transforms/__init__.py:60-76
  fft2c / ifft2c       ncsn/linear_transforms/__init__.py:36-57   (i2k_complex / k2i_complex)
  sens_maps            ncsn/linear_transforms/undersampling_fourier.py:100-138
  sense_forward        undersampling_fourier.py:140-150 (+ :77-82)
  sense_adjoint        undersampling_fourier.py:152-160 (+ :84-87: no mask on the adjoint)
  sense_ssos           undersampling_fourier.py:162-170
  l2_penalty           ncsn/models/proximal_op.py:19-51  (one SGD(lr=0.05) step through autograd)
  single_coil          ncsn/models/proximal_op.py:72-94
  get_sigmas           ncsn/models/__init__.py:10-38
  get_lh_weights       ncsn/models/ALD_optimizers.py:23-38
"""
import numpy as np

# (sw, sm, sa) parameter sets of generate_mask.  R20 / R16 / R8 are the reference's
# (undersampling_fourier.py:68-73); R40 is this build's choice -- the reference has none.
MASK_PARAMS = {
    "R20": dict(sw=0.07, sm=0.3, sa=0.01782),
    "R16": dict(sw=0.07926, sm=0.42, sa=0.02),
    "R8": dict(sw=0.196, sm=0.5, sa=0.02),
    "R40": dict(sw=0.07, sm=0.11, sa=0.0065),
}


def generate_mask(T, N, sw=0.3, sm=0.7, sa=0.045, T_max=1000, dev=0.01, seed=None):
    """bool (1, N) if T == 1 else (T, 1, N); legacy numpy MT19937 draw order must be kept."""
    np.random.seed(seed)
    x = np.linspace(-1, 1, N)
    p = np.exp(-np.abs(x) / sw) * sm + sa
    masks = np.random.rand(N, T_max) <= p[:, None]
    masks[N // 2 - 1:N // 2 + 1, :] = True
    keep = np.abs(masks.mean(axis=0) - masks.mean()) < dev
    cand = masks[:, keep]
    idx = np.random.choice(cand.shape[1], T)
    out = cand[:, idx].T
    return out[0:1, :] if T == 1 else out[:, None, :]


def fft2c(x):
    """centred orthonormal 2-D FFT over the last two axes, complex64."""
    x = np.asarray(x).astype(np.complex64)
    k = np.fft.fftn(np.fft.ifftshift(x, axes=(-1, -2)), axes=(-1, -2), norm="ortho")
    return np.fft.fftshift(k, axes=(-1, -2)).astype(np.complex64)


def ifft2c(k):
    k = np.asarray(k).astype(np.complex64)
    x = np.fft.ifftn(np.fft.ifftshift(k, axes=(-1, -2)), axes=(-1, -2), norm="ortho")
    return np.fft.fftshift(x, axes=(-1, -2)).astype(np.complex64)


def coil_anchor(H, W, seed):
    np.random.seed(seed)
    return int(np.random.choice(H)), int(np.random.choice(W))


def sens_maps(num_sens, H, W, seed):
    """float64 (num_sens, H, W): exp(-dist/(2l)), l = max(dist)/2, divided by the root-sum-of-squares.
    The reference builds the coordinate list from np.mgrid[0:W, 0:H] flattened and then reshapes the
    distances to (H, W) (undersampling_fourier.py:131-134), i.e. entry (r, c) of the map is the
    distance of flat index r*W + c in a (W, H)-shaped grid -- reproduced literally here."""
    maps = []
    for i in range(num_sens):
        ah, aw = coil_anchor(H, W, None if seed is None else seed + i)
        ww, hh = np.mgrid[0:W, 0:H]
        d = np.sqrt((ww.flatten() - ah).astype(np.float64) ** 2 + (hh.flatten() - aw).astype(np.float64) ** 2)
        l = d.max() / 2
        maps.append(np.exp(-d.reshape(H, W) / (2 * l)))
    maps = np.stack(maps, 0)
    return maps / np.sqrt((np.abs(maps) ** 2).sum(0))


def sense_forward(x, maps, mask):
    """x (B,C,H,W) c64, maps (n,H,W) f64, mask broadcastable bool -> (n,B,C,H,W) c64."""
    out = []
    for i in range(maps.shape[0]):
        out.append(mask * fft2c(maps[i] * x))            # f64 * c64 -> c128 -> cast c64 inside fft2c
    return np.stack(out, 0).astype(np.complex64)


def sense_adjoint(s, maps, mask=None):
    """sum_i conj(S_i) F^-1 s_i.  mask=None follows conj_op (no mask); a mask gives the true adjoint."""
    acc = np.zeros(s.shape[1:], dtype=np.complex64)
    for i in range(maps.shape[0]):
        si = s[i] if mask is None else mask * s[i]
        acc = (acc + np.conj(maps[i]) * ifft2c(si)).astype(np.complex64)   # c128 product rounded on +=
    return acc


def sense_ssos(s, maps):
    acc = np.zeros(s.shape[1:], dtype=np.float32)
    for i in range(maps.shape[0]):
        acc = acc + (np.abs(ifft2c(s[i])) ** 2).astype(np.float32)
    return np.sqrt(acc)


def l2_penalty_sense(z, y, alpha, lamda, maps, mask):
    """x = z - 0.05*(alpha/lamda) * A^H(A z - y) / K,  K = num_sens * W  (SURVEY.md a7)."""
    r = sense_forward(z, maps, mask) - y
    g = sense_adjoint(r, maps, mask)
    K = maps.shape[0] * z.shape[-1]
    return (z - np.float32(0.05 * (alpha / lamda) / K) * g).astype(np.complex64)


def l2_penalty_single(z, y, alpha, lamda, mask):
    """single-coil operator: K = B (the .mean() runs over the batch only)."""
    r = mask * fft2c(z) - y
    g = ifft2c(mask * r)
    return (z - np.float32(0.05 * (alpha / lamda) / z.shape[0]) * g).astype(np.complex64)


def single_coil(z, y, alpha, lamda, mask):
    a = alpha / lamda
    x = z + a * ifft2c(y)
    k = fft2c(x)
    k = (1.0 / (1.0 + mask * a)) * k
    return ifft2c(k)


def get_sigmas(sigma_begin, sigma_end, num_classes, dist="geometric"):
    if dist == "geometric":
        return np.exp(np.linspace(np.log(sigma_begin), np.log(sigma_end), num_classes)).astype(np.float32)
    if dist == "uniform":
        return np.linspace(sigma_begin, sigma_end, num_classes).astype(np.float32)
    raise NotImplementedError("sigma distribution not supported")


def get_lh_weights(sigmas, start_time):
    w = np.zeros_like(sigmas)
    if start_time == 1:
        return w
    s = int(len(sigmas) * start_time)
    import torch
    w[s:] = torch.linspace(0, 1, len(sigmas) - s).numpy()
    return w
