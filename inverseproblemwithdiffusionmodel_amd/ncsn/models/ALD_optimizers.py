"""Annealed Langevin Dynamics samplers (mirror of the reference's ``ncsn/models/ALD_optimizers.py``:
get_lh_weights :23-38, ALDOptimizer :49-155, ALDUnconditionalSampler :158, ALDInvSegProximalRealImag
:161-327).  Same constructors, ``params`` keys (n_steps_each, step_lr, denoise, final_only), call kwargs
and return value (a list holding the final CPU tensor).

What runs where: the schedule arithmetic (step sizes) is float32 host math identical to the reference;
everything per pixel runs in libipdm.so.  For the SENSE sampler one iteration is
    score network on the (2B, 1, H, W) batch [real planes | imaginary planes]   (InstanceNorm is per image,
                                               so batching the two reference passes is exact)
    one fused kernel: Langevin update of both planes + L2Penalty proximal (FFT in LDS)
and, with ``use_graph=True`` (default), the whole iteration is captured once into a hipGraph and replayed
for all L * n_steps_each steps; per-step scalars live in a device-side ipdm_sched_t.

Extra, optional call kwargs (none changes the defaults' semantics):
    noise_fn(like) -> tensor   injected Langevin noise (parity runs; the reference draws torch.randn_like
                               on the compute device, which is not reproducible across devices)
    seed, sample_offset        Philox key / first global sample id of this shard (default noise source)
    verbose                    per-level progress prints (the reference's per-step prints are host syncs)
    use_graph                  capture/replay one iteration as a hipGraph
Differences kept on purpose: the reference's per-step ``print(max, min)`` syncs and PNG dumps are gone;
``torch.set_grad_enabled(False)`` is scoped to the call instead of leaking globally.
Segmentation guidance (``adjust_grad`` -> compute_seg_grad, :272-286): with a ``seg`` network that offers the fused
forward + input-gradient chain (``seg_unet.UNet.loglh_grad``) the iteration adds ``seg_grad / sigma * lh_weight`` to both
score planes before the Langevin update (skipped altogether when every weight is zero, ``seg_start_time == 1``).
"""
import abc

import numpy as np
import torch

from ... import ops
from .proximal_op import Proximal, L2Penalty, Constrained, SingleCoil  # noqa: F401
from ..linear_transforms.undersampling_fourier import SENSE, RandomUndersamplingFourier
from ...helpers.utils import data_transform, reshape_temporal_dim
from ..linear_transforms.finite_diff import FiniteDiff

SCHED_DTYPE = np.dtype([("step", "<f4"), ("noise_scale", "<f4"), ("coef", "<f4"), ("sigma", "<f4"),
                        ("step_id", "<i8"), ("seg_scale", "<f4"), ("reserved", "<f4")])     # = ipdm_sched_t, 32 bytes


def get_lh_weights(sigmas, start_time, curve_type="linear"):
    assert 0 <= start_time <= 1
    lh_weights = torch.zeros_like(sigmas)
    if start_time == 1:
        return lh_weights
    start_idx = int(len(sigmas) * start_time)
    if curve_type == "linear":
        lh_weights[start_idx:] = torch.linspace(0, 1, len(sigmas) - start_idx, device=sigmas.device)
        return lh_weights
    raise NotImplementedError


def step_schedule(sigmas, step_lr):
    """float32 host tensors (step_size[c], sqrt(2 step_size[c])) computed exactly as the reference's
    ``step_lr * (sigma / sigmas[-1]) ** 2`` and ``torch.sqrt(step_size * 2)`` (:217, :239)."""
    s = sigmas.detach().to("cpu", torch.float32)
    step = step_lr * (s / s[-1]) ** 2
    return step, torch.sqrt(step * 2)


class ALDOptimizer(abc.ABC):
    def __init__(self, x_mod_shape, scorenet, sigmas, params, config,
                 measurement=None, linear_tfm=None, clf=None, seg=None, device=None):
        """params: n_steps_each, step_lr, denoise, final_only"""
        self.x_mod_shape = x_mod_shape
        self.scorenet = scorenet
        self.sigmas = sigmas
        self.params = params
        self.config = config
        self.measurement = measurement
        self.linear_tfm = linear_tfm
        self.clf = clf
        self.seg = seg
        self.device = device if device is not None else torch.device("cuda")

    @torch.no_grad()
    def __call__(self, **kwargs):
        scorenet, sigmas = self.scorenet, self.sigmas
        n_steps_each, step_lr = self.params["n_steps_each"], self.params["step_lr"]
        denoise, final_only = self.params["denoise"], self.params["final_only"]
        noise_fn = kwargs.get("noise_fn")
        seed, sample_offset = kwargs.get("seed", 0), kwargs.get("sample_offset", 0)
        verbose = kwargs.get("verbose", False)

        x_mod = self.init_x_mod()
        x_mod = data_transform(self.config, x_mod).contiguous()
        images = []
        self.preprocessing_steps(**kwargs)
        steps, noise_scales = step_schedule(sigmas, step_lr)
        B = x_mod.shape[0]
        it = 0
        # The score evaluation is launch-bound at small shapes (a 32x32 forward is ~300 launches): when the subclass
        # hooks do not replace x_mod, the forward is captured once as a hipGraph on the in-place updated state.
        plain = (type(self).init_estimation is ALDOptimizer.init_estimation
                 and type(self).adjust_grad is ALDOptimizer.adjust_grad)
        graph = None
        labels = torch.zeros((B,), dtype=torch.long, device=x_mod.device)
        if kwargs.get("use_graph", plain) and plain and x_mod.is_cuda:
            scorenet(x_mod, labels)                              # warm-up: weight packing, allocator
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                grad_static = scorenet(x_mod, labels)
        for c in range(len(sigmas)):
            if verbose and c % max(len(sigmas) // 10, 1) == 0:
                print(f"{c + 1}/{len(sigmas)}")
            labels.fill_(c)
            x_mod = self.init_estimation(x_mod, alpha=steps[c], **kwargs)
            for s in range(n_steps_each):
                if graph is not None:
                    graph.replay()
                    grad = grad_static
                else:
                    grad = scorenet(x_mod, labels)
                grad = self.adjust_grad(grad, x_mod, sigma=sigmas[c], **kwargs)
                noise = None if noise_fn is None else noise_fn(x_mod).to(x_mod.device)
                ops.langevin_step(x_mod, grad, float(steps[c]), float(noise_scales[c]), noise=noise, seed=seed,
                                  sample_offset=sample_offset, step_id=it)
                it += 1
                if not final_only:
                    images.append(x_mod.to('cpu'))
        if denoise:
            labels.fill_(len(sigmas) - 1)
            s2 = float(sigmas.detach().cpu()[-1] ** 2)
            if graph is not None:
                graph.replay()
            ops.langevin_step(x_mod, grad_static if graph is not None else scorenet(x_mod, labels), s2, 0.0,
                              noise=torch.zeros_like(x_mod))
            images.append(x_mod.to('cpu'))
        if final_only:
            return [x_mod.to('cpu')]
        return images

    def preprocessing_steps(self, **kwargs):
        pass

    def init_x_mod(self):
        return torch.rand(*self.x_mod_shape).to(self.device)     # CPU generator, as the reference (:145)

    def init_estimation(self, x_mod, **kwargs):
        return x_mod

    def adjust_grad(self, grad, x_mod, **kwargs):
        return grad


class ALDUnconditionalSampler(ALDOptimizer):
    pass


class ALDInvSegProximalRealImag(ALDOptimizer):
    def __init__(self, proximal: Proximal, seg_start_time, seg_step_type, *args, **kwargs):
        super(ALDInvSegProximalRealImag, self).__init__(*args, **kwargs)
        self.proximal = proximal
        self.seg_start_time = seg_start_time
        self.seg_step_type = seg_step_type
        self.lh_weights = get_lh_weights(self.sigmas, self.seg_start_time, self.seg_step_type)
        self.if_print = False
        self.print_args = {}
        self._graph = None

    # -- one iteration = score net on [real | imag] + fused Langevin/proximal tail ------------------
    def _iteration(self, st):
        grad = self.scorenet(st["x"], st["labels"])
        B = st["B"]
        if st.get("seg_label") is not None:
            # adjust_grad (:272-286): grad + compute_seg_grad(seg, x, label, seg_mode) / sigma * lh_weight on the real and
            # on the imaginary plane (same label); the scalar lh_weight / sigma of this level sits in the device schedule
            gseg = self.seg.loglh_grad(st["x"], st["seg_label"], st["seg_mode"])
            ops.axpy_sched(grad, gseg, dev_sched=st["sched_dev"])
        if st["sc_mode"] is None:            # multi-coil SENSE + L2Penalty
            ops.ald_sense_step(st["x"][:B], st["x"][B:], grad[:B], grad[B:], st["y"], st["sens"], st["mask"], st["work"],
                               noise_re=st["noise_re"], noise_im=st["noise_im"], seed=st["seed"],
                               sample_offset=st["sample_offset"], dev_sched=st["sched_dev"])
        else:                                # single-coil RandomUndersamplingFourier + L2Penalty / SingleCoil
            ops.ald_singlecoil_step(st["x"][:B], st["x"][B:], grad[:B], grad[B:], st["y"], st["mask"], st["sc_mode"],
                                    noise_re=st["noise_re"], noise_im=st["noise_im"], seed=st["seed"],
                                    sample_offset=st["sample_offset"], dev_sched=st["sched_dev"], work=st["work"])

    def _check_fast_path(self, kwargs):
        """-> sc_mode: None for SENSE + L2Penalty, the ipdm_singlecoil_prox_f32 mode for the single-coil operators
        (the reference's acdc_inv_seg_sampling_keep_center_prox_real_imag.py:79-89 / cine_inv_sampling_...:78-88)"""
        if self._seg_active() and not hasattr(self.seg, "loglh_grad"):
            raise NotImplementedError("segmentation-likelihood guidance needs a network with the fused forward + input-gradient "
                                      f"chain (ncsn.models.seg_unet.UNet.loglh_grad); got {type(self.seg).__name__}: there is "
                                      "no autograd on the HIP path")
        if isinstance(self.linear_tfm, SENSE):
            if isinstance(self.proximal, L2Penalty):
                return None
        elif isinstance(self.linear_tfm, RandomUndersamplingFourier):
            if isinstance(self.proximal, L2Penalty):
                return ops.SC_L2PENALTY
            if isinstance(self.proximal, SingleCoil):
                return ops.SC_CLOSED_FORM
        if isinstance(self.proximal, Constrained):
            # the reference's samplers call proximal(x, y, coeff, 1.) (:315); Constrained.__call__ takes (X, S, lamda)
            raise TypeError("Constrained.__call__() takes 4 positional arguments but 5 were given")
        raise NotImplementedError(f"no fused iteration for {type(self.proximal).__name__} + "
                                  f"{type(self.linear_tfm).__name__}")

    def _seg_active(self):
        return self.seg is not None and bool((self.lh_weights != 0).any())

    @torch.no_grad()
    def __call__(self, **kwargs):
        """kwargs: label, lamda, save_dir, lr_scaled, seg_mode (+ noise_fn, seed, sample_offset, verbose, use_graph,
        n_levels/start_level to run a slice of the schedule, snapshot_its: iterations of this call whose incoming state is
        kept in self._snapshots)"""
        if self._hooks_overridden():
            return self._call_with_hooks(**kwargs)
        sc_mode = self._check_fast_path(kwargs)
        sigmas = self.sigmas
        n_steps_each, step_lr = self.params["n_steps_each"], self.params["step_lr"]
        denoise = self.params["denoise"]
        lr_scaled = kwargs.get("lr_scaled", 1.)
        noise_fn = kwargs.get("noise_fn")
        verbose = kwargs.get("verbose", False)
        use_graph = kwargs.get("use_graph", True)
        dev = self.device
        meas = self.measurement.to(dev).to(torch.complex64).contiguous()
        lin = self.linear_tfm

        x0 = kwargs.get("x_init")
        x0 = lin.conj_op(meas) if x0 is None else x0.to(dev)                  # zero-filled SENSE recon
        B, H, W = x0.shape[0], x0.shape[-2], x0.shape[-1]
        x = torch.cat([x0.real, x0.imag], dim=0).contiguous().float()        # (2B, 1, H, W)
        steps, noise_scales = step_schedule(sigmas, step_lr)
        coef = self.proximal.coef(step_lr * lr_scaled, 1., x0.shape)           # alpha = UNscaled lr (:247,313)
        L = len(sigmas)
        lv0 = kwargs.get("start_level", 0)
        lv1 = L if kwargs.get("n_levels") is None else min(L, lv0 + kwargs["n_levels"])

        st = dict(x=x, B=B, y=meas, sc_mode=sc_mode, sens=lin.sens_f32(dev) if sc_mode is None else None,
                  mask=lin.mask_u8(dev),
                  work=ops.sense_workspace(B, lin.sens_maps.shape[0] if sc_mode is None else 1, H, W, dev),
                  labels=torch.zeros(2 * B, dtype=torch.long, device=dev),
                  noise_re=None, noise_im=None, seed=kwargs.get("seed", 0),
                  sample_offset=kwargs.get("sample_offset", 0),
                  sched_dev=torch.zeros(SCHED_DTYPE.itemsize, dtype=torch.uint8, device=dev))
        if noise_fn is not None:
            st["noise_re"] = torch.empty(B, 1, H, W, device=dev)
            st["noise_im"] = torch.empty(B, 1, H, W, device=dev)
        if self._seg_active():
            if kwargs.get("label") is None:
                raise ValueError("segmentation-likelihood guidance is active (seg_start_time < 1): pass label=(B or 1, 1, H, W) int64")
            label = kwargs["label"].to(dev, torch.int64)
            n_cls = getattr(self.seg, "out_channels", None)
            lo_, hi_ = int(label.min()), int(label.max())              # host check, outside the captured graph
            if lo_ < 0 or (n_cls is not None and hi_ >= n_cls):
                raise IndexError(f"label values {lo_}..{hi_} outside the segmentation network's {n_cls} classes "
                                 "(torch.gather in the reference's compute_seg_grad raises here too)")
            if label.shape[0] != B:
                label = label.expand(B, *label.shape[1:])
            st["seg_label"] = torch.cat([label, label], dim=0).contiguous()   # real planes | imaginary planes
            st["seg_mode"] = kwargs.get("seg_mode", "full")
        # all per-step scalars for the run, uploaded once; each step copies its 24-byte record on-stream
        n_it = (lv1 - lv0) * n_steps_each
        table = np.zeros(n_it, dtype=SCHED_DTYPE)
        lv = np.repeat(np.arange(lv0, lv1), n_steps_each)
        table["step"], table["noise_scale"] = steps.numpy()[lv], noise_scales.numpy()[lv]
        table["coef"], table["sigma"] = coef, sigmas.detach().cpu().numpy()[lv]
        table["step_id"] = lv0 * n_steps_each + np.arange(n_it)
        table["seg_scale"] = (self.lh_weights.detach().cpu().float() / sigmas.detach().cpu().float()).numpy()[lv]
        table_dev = torch.from_numpy(table.view(np.uint8).reshape(n_it, -1).copy()).to(dev)
        label_table = torch.from_numpy(np.repeat(lv[:, None], 2 * B, axis=1)).to(dev)

        graph = None
        snap_its = set(kwargs.get("snapshot_its") or ())           # iteration k (of this call) -> state BEFORE it runs
        self._snapshots = {}
        for k in range(n_it):
            c = int(lv[k])
            if verbose and k % n_steps_each == 0 and c % max(L // 10, 1) == 0:
                print(f"{c + 1}/{L}")
            if k in snap_its:
                self._snapshots[k] = torch.complex(x[:B], x[B:]).to("cpu")
            st["sched_dev"].copy_(table_dev[k], non_blocking=True)
            st["labels"].copy_(label_table[k], non_blocking=True)
            if noise_fn is not None:
                st["noise_re"].copy_(noise_fn(x[:B]).to(dev))
                st["noise_im"].copy_(noise_fn(x[B:]).to(dev))
            if not use_graph:
                self._iteration(st)
            elif graph is None:
                self._iteration(st)                      # warm-up (weight packing, LDS attributes, allocator)
                if k + 1 < n_it:
                    graph = self._capture(st)
            else:
                graph.replay()
        if denoise and lv1 == L:
            st["labels"].fill_(L - 1)
            s2 = float(sigmas.detach().cpu()[-1] ** 2)
            ops.langevin_step(x, self.scorenet(x, st["labels"]), s2, 0.0, noise=torch.zeros_like(x))
        out = torch.complex(x[:B], x[B:])
        self._last_state = st
        return [out.to('cpu')]

    def _capture(self, st):
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            self._iteration(st)
        return g

    # -- the reference's hook points.  The fused iteration above does their work inside its kernels and never calls them;
    #    a subclass that OVERRIDES either one gets the reference's loop (:204-252) instead, hook by hook, eagerly --
    def _hooks_overridden(self):
        base, t = ALDInvSegProximalRealImag, type(self)
        return t.adjust_grad is not base.adjust_grad or t.post_processing is not base.post_processing

    def adjust_grad(self, grad, m_mod, **kwargs):
        """reference hook (:272-286): grad + compute_seg_grad(seg, m_mod, label, seg_mode) / sigma * seg_lamda"""
        if not self._seg_active():
            return grad                                             # weight 0 at every level: the term is 0 (:27-28)
        label = kwargs["label"].to(m_mod.device, torch.int64)
        if label.shape[0] != m_mod.shape[0]:
            label = label.expand(m_mod.shape[0], *label.shape[1:])
        gseg = self.seg.loglh_grad(m_mod, label.contiguous(), kwargs.get("seg_mode", "full"))
        return ops.axpby(grad, gseg, 1.0, float(kwargs["seg_lamda"]) / float(kwargs["sigma"]))

    def post_processing(self, x_mod_real, x_mod_imag, **kwargs):
        """reference hook (:288-327): proximal(x, measurement, alpha * lr_scaled, 1.) on separate planes"""
        x = torch.complex(x_mod_real, x_mod_imag)
        x = self.proximal(x, self.measurement.to(x.device), kwargs["alpha"] * kwargs.get("lr_scaled", 1.), 1.)
        return torch.real(x).contiguous(), torch.imag(x).contiguous()

    def _call_with_hooks(self, **kwargs):
        """the reference's loop with its insertion points honoured (a subclass overrides adjust_grad / post_processing):
        score network on both planes as one batch, adjust_grad per plane, the Langevin kernel, post_processing.  Same
        noise keying as the fused iteration (Philox by (seed, sample, iteration, plane), or noise_fn)."""
        if self._seg_active() and not hasattr(self.seg, "loglh_grad"):
            raise NotImplementedError("segmentation-likelihood guidance needs ncsn.models.seg_unet.UNet.loglh_grad")
        sigmas = self.sigmas
        n_steps_each, step_lr = self.params["n_steps_each"], self.params["step_lr"]
        noise_fn, dev = kwargs.get("noise_fn"), self.device
        seed, sample_offset = kwargs.get("seed", 0), kwargs.get("sample_offset", 0)
        meas = self.measurement.to(dev).to(torch.complex64).contiguous()
        x0 = kwargs.get("x_init")
        x0 = self.linear_tfm.conj_op(meas) if x0 is None else x0.to(dev)
        B = x0.shape[0]
        x_re, x_im = x0.real.contiguous().float(), x0.imag.contiguous().float()
        steps, noise_scales = step_schedule(sigmas, step_lr)
        L = len(sigmas)
        lv0 = kwargs.get("start_level", 0)
        lv1 = L if kwargs.get("n_levels") is None else min(L, lv0 + kwargs["n_levels"])
        hook_kw = {k: v for k, v in kwargs.items() if k not in ("sigma", "seg_lamda", "alpha")}
        it = lv0 * n_steps_each
        for c in range(lv0, lv1):
            labels = torch.full((2 * B,), c, dtype=torch.long, device=dev)
            for _ in range(n_steps_each):
                grad = self.scorenet(torch.cat([x_re, x_im], dim=0), labels)
                g_re = self.adjust_grad(grad[:B], x_re, sigma=sigmas[c], seg_lamda=self.lh_weights[c], **hook_kw)
                g_im = self.adjust_grad(grad[B:], x_im, sigma=sigmas[c], seg_lamda=self.lh_weights[c], **hook_kw)
                for plane, (xp, gp) in enumerate(((x_re, g_re), (x_im, g_im))):
                    nz = (noise_fn(xp).to(dev) if noise_fn is not None else
                          ops.philox_normal(tuple(xp.shape), dev, seed=seed, sample_offset=sample_offset, step_id=it, plane=plane))
                    ops.langevin_step(xp, gp.contiguous(), float(steps[c]), float(noise_scales[c]), noise=nz)
                x_re, x_im = self.post_processing(x_re, x_im, alpha=step_lr, sigma=sigmas[c], **hook_kw)
                x_re, x_im = x_re.contiguous().float(), x_im.contiguous().float()
                it += 1
        if self.params["denoise"] and lv1 == L:
            labels = torch.full((2 * B,), L - 1, dtype=torch.long, device=dev)
            x = torch.cat([x_re, x_im], dim=0).contiguous()
            ops.langevin_step(x, self.scorenet(x, labels), float(sigmas.detach().cpu()[-1] ** 2), 0.0, noise=torch.zeros_like(x))
            x_re, x_im = x[:B], x[B:]
        return [torch.complex(x_re, x_im).to("cpu")]


class ALD2DTime(ALDOptimizer):
    """2D+time sampler (mirror of the reference's ALD2DTime, ALD_optimizers.py:330-581): per iteration a spatial
    Langevin step with the 2-D prior on all B*T frames, a temporal step (3-D prior on 8x8xT patches, or temporal
    TV), and the SENSE L2Penalty proximal on every frame.

    x_mod_shape: (B, T, C, H, W); measurement: (num_sens, B, T, C, H, W).
    Extra optional call kwargs: noise_fn(like) (injected noise; default Philox), seed, sample_offset (first global sample
    id of this process' block: sharded runs), verbose, n_levels/start_level.  `if_random_shift` draws ONE host-side
    np.random shift per step for the whole batch (:472): ranks of a sharded run seed numpy identically and stay in step.
    The reference's `_screenshot` PNG dumps are no-ops here."""

    def __init__(self, proximal: Proximal, scorenet_T, sigmas_T, *args, **kwargs):
        super(ALD2DTime, self).__init__(*args, **kwargs)
        self.proximal = proximal
        self.scorenet_T = scorenet_T
        self.sigmas_T_orig = sigmas_T
        n = int((self.sigmas <= sigmas_T[0]).sum())
        self.sigmas_T = torch.ones_like(self.sigmas) * (-1)
        # F.interpolate(..., mode="nearest") of the temporal schedule onto the tail of the spatial one (:342-345)
        idx = torch.floor(torch.arange(n, dtype=torch.float32) * (len(sigmas_T) / n)).long().clamp_(max=len(sigmas_T) - 1)
        self.sigmas_T[-n:] = sigmas_T.to(self.sigmas.device)[idx.to(sigmas_T.device)].to(self.sigmas.device)
        self.scorenet_T.sigmas = self.sigmas_T
        self.win_size = int(np.sqrt(self.scorenet_T.config.data.channels))
        self.finite_diff = None
        self.if_print = False
        self.print_args = {}
        self._it = 0

    def _noise(self, like, plane, kwargs):
        fn = kwargs.get("noise_fn")
        if fn is not None:
            return fn(like).to(like.device)
        # rows of `like` per posterior sample (T frames in the spatial step, patches in the temporal one): the Philox
        # sample id is global, so a sample's noise does not depend on how the batch is sharded over ranks
        per = like.shape[0] // max(self.measurement.shape[1], 1)
        return ops.philox_normal(tuple(like.shape), like.device, seed=kwargs.get("seed", 0),
                                 sample_offset=kwargs.get("sample_offset", 0) * per, step_id=self._it, plane=plane)

    @torch.no_grad()
    def __call__(self, **kwargs):
        """kwargs: save_dir, lr_scaled, mode_T in [tv, diffusion1d, none, diffusion1d-only, tv-only], lamda_T,
        if_random_shift"""
        mode_T = kwargs.get("mode_T", "diffusion1d")
        if_skip_spatial = False
        if mode_T in ["diffusion1d-only", "tv-only"]:
            self.sigmas_T = self.sigmas_T_orig
            self.scorenet_T.sigmas = self.sigmas_T_orig
            self.sigmas = self.sigmas_T_orig
            if_skip_spatial = True
        self.preprocessing_steps(**kwargs)
        x_mod = self.init_x_mod()
        L = self.sigmas.shape[0]
        lv0 = kwargs.get("start_level", 0)
        lv1 = L if kwargs.get("n_levels") is None else min(L, lv0 + kwargs["n_levels"])
        self._steps, self._noise_scales = step_schedule(self.sigmas, self.params["step_lr"])
        sT = self.sigmas_T.detach().to("cpu", torch.float32)
        self._steps_T = self.params["step_lr"] * (sT / sT[-1]) ** 2
        self._it = lv0 * self.params["n_steps_each"] * 4
        for c in range(lv0, lv1):
            if kwargs.get("verbose") and c % max(L // 10, 1) == 0:
                print(f"current: {c + 1}/{L}")
            for s in range(self.params["n_steps_each"]):
                x_mod = self.spatial_step(x_mod, c, if_skip_spatial, kwargs)
                x_mod = self.temporal_step(x_mod, c, mode_T, kwargs.get("lamda_T", 1.),
                                           kwargs.get("if_random_shift", False), kwargs)
                x_mod = self.proximal_step(x_mod, self.params["step_lr"], kwargs["lr_scaled"])
        return [x_mod.to("cpu")]

    def init_x_mod(self):
        num_sens, B, T, C, H, W = self.measurement.shape
        measurement = self.measurement.to(self.device).reshape(num_sens, -1, C, H, W)
        return self.linear_tfm.conj_op(measurement).reshape(B, T, C, H, W)

    def _langevin_pair(self, x_re, x_im, net, labels, step, noise_scale, kwargs):
        """both planes through `net` as one batch, then the in-place Langevin kernel"""
        n = x_re.shape[0]
        x = torch.cat([x_re, x_im], dim=0).contiguous()
        grad = net(x, torch.cat([labels, labels]))
        nz = torch.cat([self._noise(x_re, 0, kwargs), self._noise(x_im, 1, kwargs)], dim=0)
        self._it += 1
        ops.langevin_step(x, grad.contiguous(), float(step), float(noise_scale), noise=nz)
        return x[:n], x[n:]

    def spatial_step(self, x_mod, c, if_skip_spatial, kwargs=None):
        kwargs = kwargs or {}
        if if_skip_spatial:
            return x_mod
        B, T, C, H, W = x_mod.shape
        x = x_mod.reshape(-1, C, H, W)
        labels = torch.full((x.shape[0],), c, dtype=torch.long, device=x.device)
        re, im = self._langevin_pair(x.real.contiguous().float(), x.imag.contiguous().float(), self.scorenet, labels,
                                     self._steps[c], self._noise_scales[c], kwargs)
        return torch.complex(re, im).reshape(B, T, C, H, W)

    def temporal_step(self, x_mod, c, mode_T, lamda_T, if_random_shift, kwargs=None):
        kwargs = kwargs or {}
        if "tv" in mode_T:
            if self.finite_diff is None:
                self.finite_diff = FiniteDiff(dims=1)
            re = x_mod.real + self.finite_diff.log_lh_grad(x_mod.real, lamda=lamda_T)
            im = x_mod.imag + self.finite_diff.log_lh_grad(x_mod.imag, lamda=lamda_T)
            return torch.complex(re, im)
        if "diffusion1d" in mode_T:
            if float(self.sigmas_T[c]) == -1:
                return x_mod
            B, T, C, H, W = x_mod.shape
            x = x_mod.permute(0, 2, 1, 3, 4).reshape(-1, T, H, W)              # (BC, T, H, W)
            if if_random_shift:
                shifts_np = np.random.randint(0, self.win_size, (2,))            # host RNG, shared by the batch (:472)
                x = torch.roll(x, shifts=tuple(shifts_np.tolist()), dims=(-2, -1))
            x = reshape_temporal_dim(x, self.win_size, self.win_size, "forward")   # (B', kx*ky, T)
            labels = torch.full((x.shape[0],), c, dtype=torch.long, device=x.device)
            step = self._steps_T[c] * lamda_T
            re, im = self._langevin_pair(x.real.contiguous().float(), x.imag.contiguous().float(), self.scorenet_T,
                                         labels, step, torch.sqrt(step * 2), kwargs)
            x = reshape_temporal_dim(torch.complex(re, im), self.win_size, self.win_size, "backward", img_size=(H, W))
            if if_random_shift:
                x = torch.roll(x, shifts=tuple((-shifts_np).tolist()), dims=(-2, -1))
            return x.reshape(B, C, T, H, W).permute(0, 2, 1, 3, 4)
        return x_mod

    def proximal_step(self, x_mod, alpha, lr_scaled):
        B, T, C, H, W = x_mod.shape
        num_sens = self.measurement.shape[0]
        measurement = self.measurement.to(x_mod.device).reshape(num_sens, -1, *self.measurement.shape[3:])
        x = self.proximal(x_mod.reshape(-1, C, H, W).contiguous(), measurement.contiguous(), alpha * lr_scaled, 1.)
        return x.reshape(B, T, C, H, W)

    def _screenshot(self, x_mod, print_args: dict):
        return None
