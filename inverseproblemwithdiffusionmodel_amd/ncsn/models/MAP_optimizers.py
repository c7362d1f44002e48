"""MAP baselines on the gfx950 kernels (mirror of the reference's ``ncsn/models/MAP_optimizers.py``: ``MAPOptimizer``
:55-116 and its aliases ``Inpainting`` :119, ``SENSEMAP`` :123 -- SURVEY.md 8f rank 2).

Same constructor and call as the reference: ``SENSEMAP(x_init, measurement, scorenet, linear_tfm, lamda, config,
logger, device=None, opt_class=None, opt_params=None)``; ``opt()`` runs ``config.MAP.n_iters`` iterations of
    grad = log_lh_grad(x) + lamda * (s(Re x, 1) + i s(Im x, 1));   Adam(lr = config.MAP.lr, betas (0.5, 0.5)) on -grad
and returns x.  Every arithmetic step is a kernel: the data gradient is the closed-form SENSE proximal kernel
with unit coefficient (z - A^H(A z - y)) minus z, the two score evaluations are one batch, the combination is
``ipdm_axpby_f32`` and the update ``ipdm_adam_ascent_f32`` on the planar (real, imaginary) state.

Differences: only the default optimiser (Adam, optionally with other betas / eps through ``opt_params``) is built --
``opt_class`` must be None or torch.optim.Adam; ``logger`` may be None (scalars are logged when given, the image panel
every n_iters // 50 iterations as the reference); ``UndersamplingFourier`` (magnitude-only variant) and
``MAPOptimizer2DTime`` are not built."""
import torch

from ... import ops


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
        work = torch.empty(B * H * W * 2, dtype=torch.float32, device=dev)
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
