#!/usr/bin/env python3
"""Single-coil ALD reconstruction of a CINE frame -- counterpart of the reference's
``scripts/cine_inv_sampling_keep_center_prox_real_imag.py`` (same flags; no segmentation label): see
``acdc_inv_seg_sampling_keep_center_prox_real_imag.py``, which this runs with ``--dataset CINE64``."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from acdc_inv_seg_sampling_keep_center_prox_real_imag import main  # noqa: E402

if __name__ == '__main__':
    main("CINE64")
