"""UNET host side without a GPU: bit-identical initialisation against the reference-generated vectors, the
layer-definition helper, the C-ABI plan (tensor table in the reference's state_dict order)."""
import json
import os

import numpy as np
import pytest
import torch

from unet_helpers import GOLDEN, UNET_CASES, UnetCase

from cae_tools_amd.models.model_sizer import ModelSpec
from cae_tools_amd.models.unet import Decoder, Encoder, unet_layer_spec


@pytest.mark.parametrize("name", UNET_CASES)
def test_initialisation_is_bit_identical_to_the_reference(name):
    c = UnetCase(name)
    spec = ModelSpec()
    spec.load(c.meta["spec"])
    torch.manual_seed(c.meta["seed"])
    enc = Encoder(spec.get_input_layers(), encoded_space_dim=c.meta["latent"], fc_size=c.meta["fc"])
    dec = Decoder(spec.get_output_layers(), encoded_space_dim=c.meta["latent"], fc_size=c.meta["fc"])
    for (mod, keys, pre) in ((enc, c.meta["enc_keys"], "init/enc/"), (dec, c.meta["dec_keys"], "init/dec/")):
        sd = mod.state_dict()
        assert list(sd.keys()) == keys
        for k in keys:
            want = c.z[pre + k]
            assert sd[k].dtype == torch.from_numpy(want).dtype and tuple(sd[k].shape) == want.shape, k
            assert np.array_equal(sd[k].numpy(), want), k


def test_layer_spec_helper_reproduces_the_golden_specs():
    for name in UNET_CASES:
        m = UnetCase(name).meta
        enc, dec = m["spec"]["input_layers"], m["spec"]["output_layers"]
        (ic, ih, iw) = enc[0]["input_dimensions"]
        spec = unet_layer_spec(ic, dec[-1]["output_dimensions"][0], (ih, iw), [l["output_dimensions"][0] for l in enc],
                               kernel_size=enc[0]["kernel_size"], stride=enc[0]["stride"], padding=enc[0]["output_padding"])
        assert spec.save() == m["spec"]
    with pytest.raises(ValueError, match="skip connection"):
        unet_layer_spec(1, 1, (9, 9), [8, 8])         # 9 -> 4 -> 2, but 2 -> 4 -> 8 != 9


def test_plan_matches_reference_state_dict():
    from cae_tools_amd.unet_engine import UnetPlan
    for name in UNET_CASES:
        m = UnetCase(name).meta
        plan = UnetPlan(m["spec"], m["fc"], m["latent"], 4)
        enc = [n[4:] for n in plan.tensors if n.startswith("enc/")]
        dec = [n[4:] for n in plan.tensors if n.startswith("dec/")]
        assert enc == [k for k in m["enc_keys"] if "num_batches" not in k]
        assert dec == [k for k in m["dec_keys"] if "num_batches" not in k]
        assert sum(t[2] for t in plan.tensors.values() if t[0] == 0) == m["params"]
        assert plan.workspace_bytes > 0
