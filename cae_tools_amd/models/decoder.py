"""Decoder: parameter container with the reference's custom initialisation.

Mirrors src/cae_tools/models/decoder.py:24-78: Linear(latent, fc) -> ReLU -> Linear(fc, C*y*x),
unflatten, per layer ConvTranspose2d(k, stride, output_padding) (-> BatchNorm2d -> ReLU except
after the last), sigmoid.  Initialisation (:55-71): every layer first draws its PyTorch default
init, then ConvTranspose2d weights are re-drawn kaiming_normal(fan_out, relu), Linear weights
kaiming_normal(fan_out, relu) or xavier_normal when out_features == C*y*x, biases zeroed, in
module registration order.
Calling the module on its own (decoder.py:73-78) runs the attached engine through cae_decode, EVAL
mode; training and scoring run encoder and decoder fused.  An unattached module raises.
"""
from torch.nn import init

from ._params import ParamBag, add_batchnorm, add_conv


class Decoder(ParamBag):

    def __init__(self, layers, encoded_space_dim, fc_size):
        super().__init__()
        self.layers = list(layers)
        (self.chan, self.y, self.x) = self.layers[0].get_input_dimensions()
        flat = self.chan * self.y * self.x
        # construction order = RNG order: both Linears, then every ConvTranspose2d
        add_conv(self, "decoder_lin.0", (fc_size, encoded_space_dim), fc_size)
        add_conv(self, "decoder_lin.2", (flat, fc_size), flat)
        last = len(self.layers) - 1
        for i, layer in enumerate(self.layers):
            (cin, _, _) = layer.get_input_dimensions()
            (cout, _, _) = layer.get_output_dimensions()
            (kh, kw) = layer.kernel_hw()
            add_conv(self, f"decoder_conv.{3 * i}", (cin, cout, kh, kw), cout)
            if i != last:
                add_batchnorm(self, f"decoder_conv.{3 * i + 1}", cout)
        self._reinitialise(flat)
        self._engine = None

    def _reinitialise(self, flat):
        for name in ("decoder_lin.0", "decoder_lin.2"):
            w = self.get(name + ".weight")
            if w.shape[0] == flat:
                init.xavier_normal_(w)
            else:
                init.kaiming_normal_(w, mode="fan_out", nonlinearity="relu")
            init.constant_(self.get(name + ".bias"), 0)
        for i in range(len(self.layers)):
            init.kaiming_normal_(self.get(f"decoder_conv.{3 * i}.weight"), mode="fan_out", nonlinearity="relu")
            init.constant_(self.get(f"decoder_conv.{3 * i}.bias"), 0)

    def attach(self, engine):
        self._engine = engine

    def forward(self, z):
        """z (B, encoded_space_dim) fp32 CUDA tensor -> y (B, C, H, W) in [0, 1]; eval mode, on the attached engine
        (whose parameter arena holds the live weights: HipEngine.load_state / ConvAEModel put them there)"""
        if self._engine is None:
            raise RuntimeError("Decoder.forward needs an attached HipEngine (ConvAEModel attaches one when it builds or "
                               "loads a model): there is no CPU path")
        return self._engine.decode(z)

    __call__ = forward
