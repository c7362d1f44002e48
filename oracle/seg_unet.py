"""torch-CPU restatement of the segmentation network and of compute_seg_grad (oracle; test infrastructure only).

Reference anchors: ncsn/configs/general_config.yml:1-6 (Seg: MONAI UNet arguments), helpers/load_model.py:30,140-141
(``UNet(**net_params)``), ncsn/models/__init__.py:197-215 (compute_seg_grad: autograd of sum log softmax(seg(X))[label]),
ncsn/models/ALD_optimizers.py:272-286 (adjust_grad: grad + grad_log_lh_seg / sigma * lamda).

MONAI is NOT installed in this image and is not vendored by the reference (version unpinned): the module
below restates ``monai.networks.nets.UNet`` for the reference's arguments (num_res_units = 0, kernel 3, InstanceNorm,
PReLU, 'NDA' ordering, bias) from its published source, with MONAI's parameter names -- PARITY UNPINNED against MONAI
itself.  What it pins is the product's hand-written forward / input-gradient chain against plain torch modules and
torch autograd.
"""
import torch
import torch.nn as nn


class _ADN(nn.Module):
    def __init__(self, ch):
        super().__init__()
        self.N = nn.InstanceNorm2d(ch)            # affine=False, eps=1e-5: no parameters, no buffers
        self.A = nn.PReLU()

    def forward(self, x):
        return self.A(self.N(x))


class _Convolution(nn.Module):
    def __init__(self, cin, cout, stride, transposed=False, conv_only=False):
        super().__init__()
        if transposed:
            self.conv = nn.ConvTranspose2d(cin, cout, 3, stride=stride, padding=1, output_padding=stride - 1)
        else:
            self.conv = nn.Conv2d(cin, cout, 3, stride=stride, padding=1)
        if not conv_only:
            self.adn = _ADN(cout)

    def forward(self, x):
        x = self.conv(x)
        return self.adn(x) if hasattr(self, "adn") else x


class _Skip(nn.Module):
    def __init__(self, submodule):
        super().__init__()
        self.submodule = submodule

    def forward(self, x):
        return torch.cat([x, self.submodule(x)], dim=1)


class UNet(nn.Module):
    def __init__(self, in_channels=1, out_channels=2, channels=(64, 128, 256, 512, 1024), strides=(2, 2, 2, 2)):
        super().__init__()

        def block(inc, outc, chans, strs, is_top):
            c, s = chans[0], strs[0]
            if len(chans) > 2:
                sub, upc = block(c, c, chans[1:], strs[1:], False), c * 2
            else:
                sub, upc = _Convolution(c, chans[1], 1), c + chans[1]
            return nn.Sequential(_Convolution(inc, c, s), _Skip(sub), _Convolution(upc, outc, s, True, is_top))

        self.model = block(in_channels, out_channels, list(channels), list(strides), True)

    def forward(self, x):
        return self.model(x)


def compute_seg_grad(seg, X, label, mode="full"):
    """ncsn/models/__init__.py:197-215"""
    assert mode in ["full", "FG"]
    with torch.enable_grad():
        X = X.detach().clone().requires_grad_(True)
        y = torch.softmax(seg(X), dim=1)
        sel = torch.gather(y, dim=1, index=label)
        torch.log(sel).sum(dim=(1, 2, 3)).sum().backward()
    g = X.grad.detach()
    return g * label if mode == "FG" else g
