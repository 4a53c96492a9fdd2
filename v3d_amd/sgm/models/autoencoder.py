"""AutoencodingEngine — first-stage wrapper (reference: sgm/models/autoencoder.py:100-212).  Inference surface only:
encode / decode / forward with the reference's constructor params; the decoder runs on the HIP kernels."""
from __future__ import annotations

from typing import Dict, List, Optional, Tuple, Union

import torch
import torch.nn as nn

from ..util import default, instantiate_from_config


class AutoencodingEngine(nn.Module):
    def __init__(self, *args, encoder_config: Dict, decoder_config: Dict, loss_config: Dict, regularizer_config: Dict,
                 optimizer_config: Union[Dict, None] = None, lr_g_factor: float = 1.0,
                 trainable_ae_params: Optional[List[List[str]]] = None, ae_optimizer_args: Optional[List[dict]] = None,
                 trainable_disc_params: Optional[List[List[str]]] = None, disc_optimizer_args: Optional[List[dict]] = None,
                 disc_start_iter: int = 0, diff_boost_factor: float = 3.0, ckpt_engine: Union[None, str, dict] = None,
                 ckpt_path: Optional[str] = None, additional_decode_keys: Optional[List[str]] = None,
                 input_key: str = "jpg", monitor=None, ema_decay=None, **kwargs):
        super().__init__()
        self.input_key = input_key
        self.encoder: nn.Module = instantiate_from_config(encoder_config)
        self.decoder: nn.Module = instantiate_from_config(decoder_config)
        self.loss: nn.Module = instantiate_from_config(loss_config)
        self.regularization = instantiate_from_config(regularizer_config)
        if default(ckpt_path, ckpt_engine) is not None:
            raise NotImplementedError("AutoencodingEngine: load weights through DiffusionEngine.init_from_ckpt / load_state_dict")
        self.additional_decode_keys = set(default(additional_decode_keys, []))

    def get_last_layer(self):
        return self.decoder.get_last_layer()

    def encode(self, x: torch.Tensor, return_reg_log: bool = False, unregularized: bool = False):
        z = self.encoder(x)
        if unregularized:
            return z, dict()
        z, reg_log = self.regularization(z)
        return (z, reg_log) if return_reg_log else z

    def decode(self, z: torch.Tensor, **kwargs) -> torch.Tensor:
        return self.decoder(z, **kwargs)

    def forward(self, x: torch.Tensor, **additional_decode_kwargs) -> Tuple[torch.Tensor, torch.Tensor, dict]:
        z, reg_log = self.encode(x, return_reg_log=True)
        return z, self.decode(z, **additional_decode_kwargs), reg_log
