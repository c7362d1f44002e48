"""MAP baselines on the gfx950 kernels (mirror of the reference's ``ncsn/models/MAP_optimizers.py``: ``MAPModel`` :26-52
(the TV baseline), ``MAPOptimizer`` :55-116 and its aliases ``Inpainting`` :119, ``SENSEMAP`` :123 -- SURVEY.md 8f rank 2).

Same constructor and call as the reference: ``SENSEMAP(x_init, measurement, scorenet, linear_tfm, lamda, config,
logger, device=None, opt_class=None, opt_params=None)``; ``opt()`` runs ``config.MAP.n_iters`` iterations of
    grad = log_lh_grad(x) + lamda * (s(Re x, 1) + i s(Im x, 1));   Adam(lr = config.MAP.lr, betas (0.5, 0.5)) on -grad
and returns x.  Every arithmetic step is a kernel: the data gradient is the closed-form SENSE proximal kernel
with unit coefficient (z - A^H(A z - y)) minus z, the two score evaluations are one batch, the combination is
``ipdm_axpby_f32`` and the update ``ipdm_adam_ascent_f32`` on the planar (real, imaginary) state.

Differences: only the default optimiser (Adam, optionally with other betas / eps through ``opt_params``) is built --
``opt_class`` must be None or torch.optim.Adam; ``logger`` may be None (scalars are logged when given, the image panel
every n_iters // 50 iterations as the reference); ``UndersamplingFourier`` (magnitude-only variant) is not built;
``MAPOptimizer2DTime`` (below) logs only the gradient norm and writes no GIF / screenshot panels."""
import torch

from ... import ops


class TotalVariation:
    """kornia.losses.TotalVariation as the reference's TV baseline uses it (scripts/acdc_SENSE_TV.py:76): sum of the moduli
    of the vertical and horizontal first differences, one value per image; value and gradient are HIP kernels (tv.hip)"""

    def __call__(self, x):
        v = ops.tv_value(x.to(torch.complex64).contiguous())
        return v.reshape(x.shape[:-2])

    def grad(self, x):
        return ops.tv_grad(x.to(torch.complex64).contiguous())


class MAPModel:
    """The regularised least-squares baseline (reference ncsn/models/MAP_optimizers.py:26-52 MAPModel, trained by
    helpers/pl_helpers.py:402-442 TrainMAPModel: Adam(lr) on the complex image X initialised with A^H S, one step per epoch):
        loss = |A X - S|^2 / 2 + reg_weight * reg(X)
    `reg` needs `__call__(X)` and `grad(X)` (TotalVariation above).  fit(num_epochs, lr) is the training loop (no Lightning:
    the closed-form data gradient A^H(A X - S) is the SENSE proximal kernel with unit coefficient, the update
    ipdm_adam_ascent_f32 on the planar (real, imaginary) state -- torch's Adam treats a complex parameter the same way)."""

    def __init__(self, S, lin_tfm, reg, reg_weight, device=None):
        self.device = torch.device("cuda") if device is None else device
        self.S = S.to(self.device).to(torch.complex64).contiguous()
        self.lin_tfm = lin_tfm
        self.X = lin_tfm.conj_op(self.S).to(torch.complex64).contiguous()
        self.reg, self.reg_weight = reg, reg_weight

    @torch.no_grad()
    def forward(self):
        AX = self.lin_tfm(self.X)
        data_loss = (torch.abs(AX - self.S) ** 2).sum() / 2
        reg_loss = self.reg(self.X).sum()
        return data_loss, reg_loss, data_loss + self.reg_weight * reg_loss

    __call__ = forward

    @torch.no_grad()
    def fit(self, num_epochs, lr, betas=(0.9, 0.999), eps=1e-8, log_fn=None):
        tfm, dev = self.lin_tfm, self.device
        B, H, W = self.X.shape[0], self.X.shape[-2], self.X.shape[-1]
        x = torch.cat([self.X.real, self.X.imag], dim=0).contiguous().float()        # planar state (2B, 1, H, W)
        sens, mask = tfm.sens_f32(dev), tfm.mask_u8(dev)
        m, v = torch.zeros_like(x), torch.zeros_like(x)
        px = torch.empty_like(x)
        work = ops.sense_workspace(B, sens.shape[0], H, W, dev)
        for it in range(num_epochs):
            # x - A^H(A x - S): the L2 proximal kernel with unit coefficient; minus x = the descent direction of the data term
            ops.sense_l2prox(x[:B], x[B:], self.S, sens, mask, 1.0, out_re=px[:B], out_im=px[B:], work=work)
            g = ops.axpby(px, x, 1.0, -1.0)
            gr = self.reg.grad(torch.complex(x[:B], x[B:]))
            g = ops.axpby(g, torch.cat([gr.real, gr.imag], dim=0).contiguous(), 1.0, -float(self.reg_weight))
            ops.adam_ascent(x, g, m, v, lr, it + 1, betas=betas, eps=eps)
            if log_fn is not None:
                self.X = torch.complex(x[:B], x[B:])
                log_fn(it, *self.forward())
        self.X = torch.complex(x[:B], x[B:])
        return self.get_reconstruction()

    def get_reconstruction(self):
        return self.X.detach().cpu()


class MAPOptimizer(object):
    def __init__(self, x_init, measurement, scorenet, linear_tfm, lamda, config, logger=None, device=None, opt_class=None,
                 opt_params=None):
        if opt_class is not None and opt_class is not torch.optim.Adam:
            raise NotImplementedError("only torch.optim.Adam semantics are built (ipdm_adam_ascent_f32)")
        self.x_init = x_init
        self.measurement = measurement
        self.scorenet = scorenet
        self.linear_tfm = linear_tfm
        self.lamda = lamda
        self.config = config
        self.device = torch.device("cuda") if device is None else device
        self.logger = logger
        self.plot_interval = max(self.config.MAP.n_iters // 50, 1)
        self.lr = self.config.MAP.lr
        opt_params = {"betas": (0.5, 0.5)} if opt_params is None else dict(opt_params)
        unknown = set(opt_params) - {"betas", "eps"}
        if unknown:
            raise NotImplementedError(f"Adam options {sorted(unknown)} are not built")
        self.betas = tuple(opt_params.get("betas", (0.9, 0.999)))
        self.eps = opt_params.get("eps", 1e-8)

    @torch.no_grad()
    def __call__(self):
        tfm = self.linear_tfm
        dev = self.device
        x0 = self.x_init.to(dev).to(torch.complex64)
        B, H, W = x0.shape[0], x0.shape[-2], x0.shape[-1]
        x = torch.cat([x0.real, x0.imag], dim=0).contiguous().float()                 # planar state (2B, 1, H, W)
        y = self.measurement.to(dev).to(torch.complex64).contiguous()
        sens, mask = tfm.sens_f32(dev), tfm.mask_u8(dev)
        m, v = torch.zeros_like(x), torch.zeros_like(x)
        px = torch.empty_like(x)
        work = ops.sense_workspace(B, sens.shape[0], H, W, dev)
        labels = torch.ones(2 * B, dtype=torch.long, device=dev)
        n_iters = self.config.MAP.n_iters
        for it in range(n_iters):
            # z - A^H(A z - y): the L2 proximal kernel with unit coefficient; grad_data = that minus z
            ops.sense_l2prox(x[:B], x[B:], y, sens, mask, 1.0, out_re=px[:B], out_im=px[B:], work=work)
            score = self.scorenet(x, labels)
            grad = ops.axpby(px, x, 1.0, -1.0)
            grad = ops.axpby(grad, score, 1.0, float(self.lamda))
            ops.adam_ascent(x, grad, m, v, self.lr, it + 1, betas=self.betas, eps=self.eps)
            if self.logger is not None:
                xc = torch.complex(x[:B], x[B:])
                data_error = 0.5 * float(torch.sum(torch.abs(tfm(xc) - y) ** 2))
                self.logger.add_scalar("data_error", data_error, global_step=it)
                self.logger.add_scalar("grad", float(torch.linalg.vector_norm(grad)), global_step=it)
                if it % self.plot_interval == 0 or it == n_iters - 1:
                    self.logger.add_image("recons_img", xc.abs().cpu()[0], global_step=it, dataformats="CHW")
        out = torch.complex(x[:B], x[B:])
        if isinstance(self.x_init, torch.Tensor) and self.x_init.shape == out.shape and self.x_init.is_complex():
            self.x_init.copy_(out.to(self.x_init.device))                            # the reference updates x_init in place
        return out


class Inpainting(MAPOptimizer):
    pass


class SENSEMAP(MAPOptimizer):
    pass


class MAPOptimizer2DTime(object):
    """2D+time MAP baseline (reference :154-365): x (B, T, C, H, W) complex; per iteration
        grad = data + prior_weight * (spatial_step_weight * grad_S + temporal_step_weight * grad_T)
    and one Adam step on the real and on the imaginary part (the reference's two optimisers evaluate the same gradient
    from the same x, so they are one elementwise Adam on the planar state).  ``params`` keys as the reference: lr,
    opt_class (None / torch.optim.Adam), opt_params, device, num_iters, num_plot_times, win_size, prior_weight,
    spatial_step_weight, temporal_step_weight, save_dir, mode_T in [diffusion1d, tv], if_random_shift."""

    def __init__(self, x_init, measurement, scorenet_S, scorenet_T, linear_tfm, logger, params):
        import numpy as np
        self.params = params
        oc = params.get("opt_class")
        if oc is not None and oc is not torch.optim.Adam:
            raise NotImplementedError("only torch.optim.Adam semantics are built (ipdm_adam_ascent_f32)")
        op = dict(params.get("opt_params", {}))
        if set(op) - {"betas", "eps"}:
            raise NotImplementedError(f"Adam options {sorted(set(op) - {'betas', 'eps'})} are not built")
        self.betas, self.eps = tuple(op.get("betas", (0.9, 0.999))), op.get("eps", 1e-8)
        self.x = x_init
        self.measurement = measurement
        self.scorenet_S, self.scorenet_T = scorenet_S, scorenet_T
        self.win_size = int(np.sqrt(self.scorenet_T.config.data.channels))
        self.linear_tfm = linear_tfm
        self.device = params.get("device") or torch.device("cuda")
        self.logger = logger
        self.finite_diff = None

    def temporal_grad(self, x, mode_T, if_random_shift):
        """x (B, T, C, H, W) complex64 on the device -> temporal prior gradient, same shape"""
        import numpy as np
        from ..linear_transforms.finite_diff import FiniteDiff
        from ...helpers.utils import reshape_temporal_dim
        if mode_T == "tv":
            if self.finite_diff is None:
                self.finite_diff = FiniteDiff(dims=1)
            return torch.complex(self.finite_diff.log_lh_grad(x.real), self.finite_diff.log_lh_grad(x.imag))
        if mode_T != "diffusion1d":
            raise ValueError(f"mode_T {mode_T!r}: expected 'diffusion1d' or 'tv'")
        B, T, C, H, W = x.shape
        win = self.params["win_size"]
        v = x.permute(0, 2, 1, 3, 4).reshape(B * C, T, H, W)
        if if_random_shift:
            shifts_np = np.random.randint(0, self.win_size, (2,))             # host RNG, one shift per iteration (:309)
            v = torch.roll(v, shifts=tuple(shifts_np.tolist()), dims=(-2, -1))
        p = reshape_temporal_dim(v, win, win, "forward")                      # (B', kx*ky, T)
        n = p.shape[0]
        labels = torch.ones(2 * n, dtype=torch.long, device=x.device)
        s = self.scorenet_T(torch.cat([p.real, p.imag], dim=0).contiguous().float(), labels)
        gT = reshape_temporal_dim(torch.complex(s[:n], s[n:]), win, win, "backward", img_size=(H, W))
        if if_random_shift:
            gT = torch.roll(gT, shifts=tuple((-shifts_np).tolist()), dims=(-2, -1))
        return gT.reshape(B, C, T, H, W).permute(0, 2, 1, 3, 4)

    @torch.no_grad()
    def __call__(self):
        P, dev, tfm = self.params, self.device, self.linear_tfm
        x0 = self.x.to(dev).to(torch.complex64)
        B, T, C, H, W = x0.shape
        N = B * T * C
        xs = torch.cat([x0.real.reshape(N, 1, H, W), x0.imag.reshape(N, 1, H, W)], dim=0).contiguous().float()
        y = self.measurement.to(dev).to(torch.complex64).reshape(self.measurement.shape[0], N, 1, H, W).contiguous()
        sens, mask = tfm.sens_f32(dev), tfm.mask_u8(dev)
        m, v = torch.zeros_like(xs), torch.zeros_like(xs)
        px = torch.empty_like(xs)
        work = ops.sense_workspace(N, sens.shape[0], H, W, dev)
        labels = torch.ones(2 * N, dtype=torch.long, device=dev)
        pw, wS, wT = P["prior_weight"], P["spatial_step_weight"], P["temporal_step_weight"]
        for it in range(P["num_iters"]):
            ops.sense_l2prox(xs[:N], xs[N:], y, sens, mask, 1.0, out_re=px[:N], out_im=px[N:], work=work)
            grad = ops.axpby(px, xs, 1.0, -1.0)                                        # data term: -A^H(A x - y)
            grad = ops.axpby(grad, self.scorenet_S(xs, labels), 1.0, float(pw * wS))   # + spatial prior
            xc = torch.complex(xs[:N], xs[N:]).reshape(B, T, C, H, W)
            gT = self.temporal_grad(xc, P["mode_T"], P.get("if_random_shift", False))
            gTp = torch.cat([gT.real.reshape(N, 1, H, W), gT.imag.reshape(N, 1, H, W)], dim=0).contiguous().float()
            grad = ops.axpby(grad, gTp, 1.0, float(pw * wT))                           # + temporal prior
            ops.adam_ascent(xs, grad, m, v, P["lr"], it + 1, betas=self.betas, eps=self.eps)
            if self.logger is not None:
                self.logger.add_scalar("grad", float(torch.linalg.vector_norm(grad)), global_step=it)
        self.x = torch.complex(xs[:N], xs[N:]).reshape(B, T, C, H, W)
        return self.get_reconstruction()

    def get_reconstruction(self):
        return self.x.detach().cpu()
