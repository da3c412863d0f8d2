"""DSDataset — the xarray -> (N,C,H,W) loader of the reference, with the arithmetic on the GPU.

Public surface and semantics follow src/cae_tools/models/ds_dataset.py:
  __init__ :22-75    NaN check of every variable (ValueError with the reference's messages), global
                     min / max per input variable and of the output as python floats
  normalisation :78-113,131-135   (x - min) / (max - min); 0.0 when an input's range is 0
  __getitem__ :137-159           (input (C,H,W) f32, output, mask of ones shaped like the input, label)
The scans run in cae_scan_f32 and the normalise + channel-concat in cae_normalise_pack
(include/cae_hip.h) on device copies of the variables; nothing is computed with numpy here.
"""
import numpy as np
import torch

from ..data.arrays import as_numpy
from .. import engine as _eng


def _device():
    if not torch.cuda.is_available():
        raise _eng.CaeError("DSDataset needs a ROCm GPU: the loader arithmetic runs in libcae_hip (no CPU fallback)")
    return torch.device("cuda", torch.cuda.current_device())


class DSDataset(torch.utils.data.Dataset):

    def __init__(self, ds, input_variable_names, output_variable_name=None, normalise_in=True, normalise_out=True,
                 mask_variable_name=None):
        self.ds = ds
        self.input_variable_names = list(input_variable_names)
        self.output_variable_name = output_variable_name
        self.normalise_in = normalise_in
        self.normalise_out = normalise_out
        self.mask_variable_name = mask_variable_name
        self.transform = None  # assigned by callers of the reference, never applied there either
        self.input_das = [ds[name] for name in self.input_variable_names]
        first = self.input_das[0]
        self.n = int(first.shape[0])
        self.input_chan = int(sum(da.shape[1] for da in self.input_das))
        (self.input_y, self.input_x) = (int(first.shape[2]), int(first.shape[3]))
        self.mask_da = ds[mask_variable_name] if mask_variable_name is not None else None

        dev = _device()
        self._raw_in = [self._upload(da, dev) for da in self.input_das]
        self._raw_out = None
        # the reference scans the output first (and requires it: :42-46)
        out_da = ds[output_variable_name]
        self._raw_out = self._upload(out_da, dev)
        (nans, omin, omax) = _eng.scan_f32(self._raw_out)
        if nans > 0:
            raise ValueError(f"output variable contains {nans} NaN values")
        self.min_inputs, self.max_inputs, self.input_spec = {}, {}, []
        for name, da, raw in zip(self.input_variable_names, self.input_das, self._raw_in):
            (nans, lo, hi) = _eng.scan_f32(raw)
            self.min_inputs[name] = lo
            self.max_inputs[name] = hi
            if nans > 0:
                raise ValueError(f"input variable {name} contains {nans} NaN values")
            self.input_spec.append({"name": name, "shape": [int(s) for s in da.shape[1:]]})
        if output_variable_name:
            self.output_da = out_da
            (self.output_chan, self.output_y, self.output_x) = (int(s) for s in out_da.shape[1:4])
            (self.min_output, self.max_output) = (omin, omax)
            self.output_spec = {"name": output_variable_name, "shape": [int(s) for s in out_da.shape[1:]]}
        else:
            self.output_da = None
            self.output_chan = self.output_y = self.output_x = None
            self.min_output = self.max_output = None
            self.output_spec = None
        self._x = None
        self._t = None
        self._t_norm_flag = None

    @staticmethod
    def _upload(da, dev):
        arr = as_numpy(da)
        if arr.ndim != 4:
            raise ValueError(f"variables must be 4-D (case, channel, y, x); got shape {arr.shape}")
        return _eng.upload_f32(arr, dev)

    # -- reference accessors ---------------------------------------------------------------
    def set_normalise_output(self, normalise_out):
        self.normalise_out = normalise_out

    def get_normalisation_parameters(self):
        return [self.min_inputs, self.max_inputs, self.min_output, self.max_output]

    def set_normalisation_parameters(self, parameters):
        (self.min_inputs, self.max_inputs, self.min_output, self.max_output) = tuple(parameters)
        print(self.min_inputs, self.max_inputs, self.min_output, self.max_output)
        self._x = self._t = None

    def get_input_shape(self):
        return (self.input_chan, self.input_y, self.input_x)

    def get_input_spec(self):
        return self.input_spec

    def get_output_shape(self):
        return (self.output_chan, self.output_y, self.output_x)

    def get_output_spec(self):
        return self.output_spec

    def denormalise_output(self, arr, force=False):
        """numpy fp64 array (host) in, fp64 out: min + arr*(max-min) — host scalar arithmetic on the
        already-scored array, as the reference does; the bulk path is denormalise_device()"""
        if force or self.normalise_out:
            return self.min_output + (arr * (self.max_output - self.min_output))
        return arr

    def denormalise_device(self, y):
        """fp32 CUDA scores -> fp64 CUDA tensor (cae_denormalise_f64)"""
        return _eng.denormalise_f64(y, self.min_output, self.max_output)

    # -- device-resident normalised arrays (what the training loop consumes) ---------------------
    def device_inputs(self):
        if self._x is None:
            x = torch.empty((self.n, self.input_chan, self.input_y, self.input_x), dtype=torch.float32,
                            device=self._raw_in[0].device)
            off = 0
            for name, raw in zip(self.input_variable_names, self._raw_in):
                _eng.normalise_pack(raw, x, off, self.min_inputs[name], self.max_inputs[name], enable=self.normalise_in)
                off += raw.shape[1]
            torch.cuda.synchronize(x.device)
            self._x = x
        return self._x

    def device_outputs(self):
        if self._raw_out is None:
            return None
        if not self.normalise_out:
            return self._raw_out        # the un-normalised target IS the uploaded variable (one variable, no concat)
        if self._t is None or self._t_norm_flag != self.normalise_out:
            t = torch.empty_like(self._raw_out)
            _eng.normalise_pack(self._raw_out, t, 0, self.min_output, self.max_output, enable=self.normalise_out)
            torch.cuda.synchronize(t.device)
            self._t, self._t_norm_flag = t, self.normalise_out
        return self._t

    def device_batches(self, order):
        """(x, t) normalised and laid out in the frozen sample order `order` (host sequence of sample indices): row i
        is sample order[i], so a batch of the frozen shuffle is a contiguous run of rows.  This is the reference's
        DataLoader(shuffle=True) + default collate, whose stacked batches are built once and reused every epoch
        (conv_ae_model.py:291-292, 315-325); here the stacking is the normalisation pass itself (cae_normalise_pack_rows
        writes each sample to its destination row), straight from the raw variables: no second normalised copy."""
        rows = _eng.inverse_permutation(order, self._raw_in[0].device)
        x = torch.empty((self.n, self.input_chan, self.input_y, self.input_x), dtype=torch.float32,
                        device=self._raw_in[0].device)
        off = 0
        for name, raw in zip(self.input_variable_names, self._raw_in):
            _eng.normalise_pack(raw, x, off, self.min_inputs[name], self.max_inputs[name], enable=self.normalise_in,
                                dst_rows=rows)
            off += raw.shape[1]
        t = None
        if self._raw_out is not None:
            t = torch.empty_like(self._raw_out)
            _eng.normalise_pack(self._raw_out, t, 0, self.min_output, self.max_output, enable=self.normalise_out,
                                dst_rows=rows)
        torch.cuda.synchronize(x.device)
        return x, t

    def device_mask(self):
        """mask variable as an fp32 CUDA tensor broadcastable to the output, or None (= every pixel)"""
        if self.mask_da is None or self.mask_da.size == 0 or self._raw_out is None:
            return None
        m = torch.from_numpy(np.ascontiguousarray(as_numpy(self.mask_da), dtype=np.float32)).to(self._raw_out.device)
        while m.dim() < self._raw_out.dim():
            m = m.unsqueeze(1)
        return m

    # -- torch Dataset protocol ---------------------------------------------------------------
    def __getitem__(self, index):
        in_arr = self.device_inputs()[index].cpu().numpy()
        out = self.device_outputs()
        out_arr = out[index].cpu().numpy() if out is not None else None
        if self.mask_da is not None and self.mask_da.size > 0:
            mask = as_numpy(self.mask_da[index]).astype(np.float32)
        else:
            mask = np.ones((self.input_chan, self.input_y, self.input_x), dtype=np.float32)
        return (in_arr, out_arr, mask, f"image{index}")

    def __len__(self):
        return self.n
