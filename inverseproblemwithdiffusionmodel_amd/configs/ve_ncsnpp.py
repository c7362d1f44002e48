"""VE-SDE NCSN++ configs (schema and values of the reference's ``configs/default_lsun_configs.py:5-72`` and
``configs/ve/celebahq_256_ncsnpp_continuous.py:22-68``; config 5 of BASELINE.json)."""
import torch

from . import ConfigDict


def get_default_configs():
    config = ConfigDict()
    config.training = ConfigDict(batch_size=64, continuous=True, reduce_mean=False, likelihood_weighting=False,
                                 sde='vesde')
    config.sampling = ConfigDict(n_steps_each=1, noise_removal=True, probability_flow=False, snr=0.075, method='pc',
                                 predictor='reverse_diffusion', corrector='langevin')
    config.data = ConfigDict(dataset='LSUN', image_size=256, random_flip=True, uniform_dequantization=False,
                             centered=False, num_channels=3)
    config.model = ConfigDict(sigma_max=378, sigma_min=0.01, num_scales=2000, beta_min=0.1, beta_max=20., dropout=0.,
                              embedding_type='fourier')
    config.seed = 42
    config.device = torch.device('cuda:0') if torch.cuda.is_available() else torch.device('cpu')
    return config


def celebahq_256_ncsnpp_continuous():
    config = get_default_configs()
    config.data.dataset = 'CelebAHQ'
    config.data.image_size = 256
    m = config.model
    m.name = 'ncsnpp'
    m.sigma_max = 348
    m.scale_by_sigma = True
    m.ema_rate = 0.999
    m.normalization = 'GroupNorm'
    m.nonlinearity = 'swish'
    m.nf = 128
    m.ch_mult = (1, 1, 2, 2, 2, 2, 2)
    m.num_res_blocks = 2
    m.attn_resolutions = (16,)
    m.resamp_with_conv = True
    m.conditional = True
    m.fir = True
    m.fir_kernel = [1, 3, 3, 1]
    m.skip_rescale = True
    m.resblock_type = 'biggan'
    m.progressive = 'output_skip'
    m.progressive_input = 'input_skip'
    m.progressive_combine = 'sum'
    m.attention_type = 'ddpm'
    m.init_scale = 0.
    m.fourier_scale = 16
    m.conv_size = 3
    return config


get_config = celebahq_256_ncsnpp_continuous
