"""Same seed -> same initial weights as the reference's Encoder/Decoder (golden init/*)."""
import numpy as np
import pytest
import torch

from helpers import MODEL_CASES, GoldenCase
from cae_tools_amd.models.model_sizer import ModelSpec
from cae_tools_amd.models.encoder import Encoder
from cae_tools_amd.models.decoder import Decoder


@pytest.mark.parametrize("name", MODEL_CASES)
def test_initial_state_is_bit_identical(name):
    case = GoldenCase(name)
    spec = ModelSpec()
    spec.load(case.spec)
    torch.manual_seed(case.meta["seed"])
    enc = Encoder(spec.get_input_layers(), encoded_space_dim=case.meta["latent"], fc_size=case.meta["fc"])
    dec = Decoder(spec.get_output_layers(), encoded_space_dim=case.meta["latent"], fc_size=case.meta["fc"])
    for side, mod in (("enc", enc), ("dec", dec)):
        ref = case.group(f"init/{side}/")
        sd = mod.state_dict()
        assert list(sd) == list(ref)
        for k in ref:
            np.testing.assert_array_equal(sd[k].numpy(), ref[k], err_msg=k)
    assert sum(p.numel() for p in enc.parameters()) + sum(p.numel() for p in dec.parameters()) > 0
