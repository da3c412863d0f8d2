"""Minimal named-array containers and NetCDF-3 I/O for the ConvAE drop-in.

The reference passes xarray Datasets around (cli/train_cae.py:58-59, base_model.py:151-152) and
touches only a small duck-typed surface of them: ds[name] -> .shape/.dims/.values/.data, item
assignment, ds.dims, to_netcdf, open_mfdataset(concat_dim=..., combine="nested").  xarray is not
a dependency of this package (and is absent from the build image): Dataset/DataArray below
provide exactly that surface over numpy arrays, and every model entry point accepts either these
or real xarray objects.  Files are NetCDF-3 classic / 64-bit-offset through the package's own reader /
writer (netcdf3.py: struct-parsed header, memory-mapped big-endian slabs); NetCDF-4 (HDF5) files need
xarray + netCDF4/h5netcdf, which open_dataset uses when importable.
"""
import os
from collections import OrderedDict

import numpy as np


class DataArray:

    def __init__(self, data, dims=None, attrs=None, coords=None):
        self._data = np.asarray(data)
        if dims is None:
            dims = tuple(f"dim_{i}" for i in range(self._data.ndim))
        if len(dims) != self._data.ndim:
            raise ValueError(f"{len(dims)} dimension names for a {self._data.ndim}-d array")
        self.dims = tuple(dims)
        self.attrs = dict(attrs or {})

    # the attributes the model code reads
    @property
    def shape(self):
        return self._data.shape

    @property
    def values(self):
        """native-endian numpy array (a file-backed big-endian view is converted once, on first use)"""
        if self._data.dtype.byteorder == ">":
            self._data = self._data.astype(self._data.dtype.newbyteorder("="))
        return self._data

    @property
    def data(self):
        return self.values

    @property
    def raw_values(self):
        """the array as stored (possibly a big-endian read-only view of a NetCDF-3 file mapping): what DSDataset
        uploads, so that file bytes go to the GPU untouched"""
        return self._data

    @property
    def dtype(self):
        return self._data.dtype.newbyteorder("=")

    @property
    def size(self):
        return self._data.size

    @property
    def ndim(self):
        return self._data.ndim

    def __getitem__(self, key):
        sub = self._data[key]       # stays a view of the file mapping for file-backed variables
        if not isinstance(key, tuple):
            key = (key,)
        dims = [d for d, k in zip(self.dims, list(key) + [slice(None)] * (self.ndim - len(key)))
                if not isinstance(k, (int, np.integer))]
        return DataArray(sub, dims=dims if len(dims) == np.ndim(sub) else None, attrs=self.attrs)

    def __len__(self):
        return self.shape[0]

    def __array__(self, dtype=None, copy=None):
        return self.values if dtype is None else self.values.astype(dtype)

    def __repr__(self):
        return f"<DataArray {dict(zip(self.dims, self.shape))} {self.dtype}>"


class _Dims(dict):
    """ds.dims['y'] -> length (cli/train_cae.py:77-78)"""


class Dataset:

    def __init__(self, data_vars=None, attrs=None):
        self._vars = OrderedDict()
        self.attrs = dict(attrs or {})
        for k, v in (data_vars or {}).items():
            self[k] = v

    def __getitem__(self, name):
        if name in self._vars:
            return self._vars[name]
        if name in self.dims:  # a bare dimension: index coordinate, like xarray
            return DataArray(np.arange(self.dims[name]), dims=(name,))
        raise KeyError(name)

    def __setitem__(self, name, value):
        if not isinstance(value, DataArray):
            if hasattr(value, "dims") and hasattr(value, "values"):
                value = DataArray(np.asarray(value.values), dims=tuple(value.dims), attrs=getattr(value, "attrs", {}))
            else:
                raise TypeError("assign a DataArray (data + dims)")
        for d, n in zip(value.dims, value.shape):
            known = self.dims.get(d)
            if known is not None and known != n:
                raise ValueError(f"dimension '{d}' has length {known}, variable '{name}' wants {n}")
        self._vars[name] = value

    def __contains__(self, name):
        return name in self._vars

    def __iter__(self):
        return iter(self._vars)

    def keys(self):
        return self._vars.keys()

    @property
    def data_vars(self):
        return self._vars

    @property
    def dims(self):
        out = _Dims()
        for v in self._vars.values():
            for d, n in zip(v.dims, v.shape):
                out.setdefault(d, n)
        return out

    def to_netcdf(self, path):
        write_netcdf3(self, path)

    def __repr__(self):
        rows = [f"  {k}: {v!r}" for k, v in self._vars.items()]
        return "<Dataset\n" + "\n".join(rows) + "\n>"


# ---------------------------------------------------------------------------------------------
# NetCDF-3
# ---------------------------------------------------------------------------------------------

def write_netcdf3(ds, path):
    """Dataset -> one NetCDF-3 (64-bit offset) file through cae_tools_amd.data.netcdf3"""
    from . import netcdf3
    variables = OrderedDict((name, (da.dims, np.asarray(da.raw_values), da.attrs)) for name, da in ds.data_vars.items())
    netcdf3.write(path, {d: int(n) for d, n in ds.dims.items()}, variables, attributes=ds.attrs, version=2)


def _is_hdf5(path):
    with open(path, "rb") as f:
        return f.read(4) == b"\x89HDF"


def open_dataset(path, mask_and_scale=True):
    """One NetCDF file -> Dataset (arrays are read-only views of the file mapping).  mask_and_scale: CF decoding of
    _FillValue / missing_value / scale_factor / add_offset as xarray applies by default (decode_cf below)."""
    if _is_hdf5(path):
        try:
            import xarray as xr
        except ImportError as ex:
            raise RuntimeError(f"{path} is NetCDF-4/HDF5; reading it needs xarray with netCDF4 or h5netcdf "
                               "(not available) - convert it to NetCDF-3 (ncks -3 / nccopy -k classic)") from ex
        return from_xarray(xr.open_dataset(path))
    from . import netcdf3
    f = netcdf3.File(path)
    out = Dataset(attrs=dict(f.attributes))
    for name, var in f.variables.items():
        # big-endian views of the file mapping (kept alive by the arrays): nothing is read until it is used, and
        # DSDataset uploads contiguous float32 slabs as raw bytes and swaps them on the GPU
        (data, attrs) = decode_cf(var.data, dict(var.attributes)) if mask_and_scale else (var.data, dict(var.attributes))
        out[name] = DataArray(data, dims=var.dimensions, attrs=attrs)
    return out


_CF_KEYS = ("_FillValue", "missing_value", "scale_factor", "add_offset")


def decode_cf(data, attrs):
    """What xarray's default `mask_and_scale=True` does to a variable on open (the reference reads every file through
    xr.open_mfdataset, cli/train_cae.py:58-59): values equal to `_FillValue` / `missing_value` become NaN, then
    `decoded = raw * scale_factor + add_offset`; the four attributes leave `attrs` (xarray moves them to .encoding).
    So a filled pixel reaches DSDataset as NaN and trips its 'contains N NaN values' ValueError (ds_dataset.py:43-46)
    instead of silently becoming the normalisation minimum, and packed integer variables are trained on in their
    physical units.  Returns (array, attrs).  A float variable whose fill value never occurs keeps its file-backed
    big-endian view (the raw-bytes upload path); everything else is decoded into a native-endian array.
    Result dtype as xarray chooses it: floats keep theirs; integers of <= 16 bits give float32 unless an add_offset
    is present (float64, unless both scale_factor and add_offset are float32); wider integers give float64."""
    if data.dtype.kind not in "iuf" or not any(k in attrs for k in _CF_KEYS):
        return data, attrs
    fills = []
    for key in ("_FillValue", "missing_value"):
        if key in attrs:
            fills.extend(np.atleast_1d(np.asarray(attrs[key])).tolist())
    scale, offset = attrs.get("scale_factor"), attrs.get("add_offset")
    rest = {k: v for k, v in attrs.items() if k not in _CF_KEYS}
    native = data.dtype.newbyteorder("=")
    if native.kind == "f":
        out_dtype = native
    elif native.itemsize <= 2 and (offset is None or (np.asarray(scale).dtype == np.float32
                                                      and np.asarray(offset).dtype == np.float32)):
        out_dtype = np.dtype(np.float32)
    else:
        out_dtype = np.dtype(np.float64)
    mask = None
    for fv in fills:
        if isinstance(fv, float) and fv != fv:
            continue   # a NaN fill value: those pixels already read as NaN
        hit = data == np.asarray(fv).astype(native)
        mask = hit if mask is None else (mask | hit)
    if scale is None and offset is None and native.kind == "f" and (mask is None or not mask.any()):
        return data, rest
    out = np.array(data, dtype=out_dtype)
    if mask is not None and mask.any():
        out[mask] = np.nan
    if scale is not None:
        out *= np.asarray(scale).astype(out_dtype).reshape(-1)[0]
    if offset is not None:
        out += np.asarray(offset).astype(out_dtype).reshape(-1)[0]
    return out, rest


def from_xarray(xds):
    out = Dataset(attrs=dict(xds.attrs))
    for name in xds.data_vars:
        v = xds[name]
        out[name] = DataArray(np.asarray(v.values), dims=tuple(v.dims), attrs=dict(v.attrs))
    return out


def open_mfdataset(paths, concat_dim="box", combine="nested", mask_and_scale=True, **_ignored):
    """Nested concatenation of several files along a NEW or existing dimension, the way the CLIs
    call xr.open_mfdataset(paths, concat_dim="box", combine="nested") (cli/train_cae.py:58-59):
    variables that carry `concat_dim` are concatenated along it; if no variable has that dimension
    (the reference's test data uses "n"), every variable gains it as a new leading axis - for a
    single file xarray then leaves the data unchanged, which is what this does too."""
    if isinstance(paths, (str, os.PathLike)):
        paths = [paths]
    parts = [open_dataset(p, mask_and_scale=mask_and_scale) for p in paths]
    if len(parts) == 1:
        return parts[0]
    out = Dataset(attrs=parts[0].attrs)
    for name in parts[0].data_vars:
        das = [p[name] for p in parts]
        dims = das[0].dims
        if concat_dim in dims:
            axis = dims.index(concat_dim)
            out[name] = DataArray(np.concatenate([d.values for d in das], axis=axis), dims=dims, attrs=das[0].attrs)
        else:
            out[name] = DataArray(np.stack([d.values for d in das], axis=0), dims=(concat_dim,) + dims,
                                  attrs=das[0].attrs)
    return out


def as_numpy(var):
    """values of a DataArray-like (ours, xarray's, dask-backed) as a numpy array; ours are handed over as stored
    (a NetCDF-3 variable stays a big-endian view of the file mapping until something converts it)"""
    v = var.raw_values if hasattr(var, "raw_values") else (var.values if hasattr(var, "values") else var)
    return np.asarray(v)
