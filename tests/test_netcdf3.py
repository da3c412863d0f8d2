"""cae_tools_amd.data.netcdf3: the package's own NetCDF-3 reader / writer, checked against
(1) a file assembled byte by byte from the classic-format specification, (2) scipy.io.netcdf_file, an independent
implementation, in both directions (fixed and record variables, CDF-1 and CDF-2), (3) its own round trip.
The reference ships no .nc fixtures (test/data blobs are absent), so these are the pins for the file layer."""
import struct

import numpy as np
import pytest

from cae_tools_amd.data import netcdf3
from cae_tools_amd.data.arrays import DataArray, Dataset, open_dataset, open_mfdataset


def _name(s):
    b = s.encode()
    return struct.pack(">I", len(b)) + b + b"\0" * (-len(b) % 4)


def test_reads_a_file_assembled_from_the_specification(tmp_path):
    # header: CDF\x01, numrecs 0, dims {n: 2, x: 3}, global att title="hi", one variable v(n, x) float, att scale=2.5
    dims = struct.pack(">II", 0x0A, 2) + _name("n") + struct.pack(">I", 2) + _name("x") + struct.pack(">I", 3)
    gatt = struct.pack(">II", 0x0C, 1) + _name("title") + struct.pack(">II", 2, 2) + b"hi\0\0"
    vatt = struct.pack(">II", 0x0C, 1) + _name("scale") + struct.pack(">II", 6, 1) + struct.pack(">d", 2.5)
    var_head = struct.pack(">II", 0x0B, 1) + _name("v") + struct.pack(">III", 2, 0, 1) + vatt + struct.pack(">II", 5, 24)
    fixed = b"CDF\x01" + struct.pack(">I", 0) + dims + gatt + var_head
    begin = len(fixed) + 4
    data = struct.pack(">6f", 1.0, -2.0, 3.5, 4.25, 1e-3, 6e6)
    path = tmp_path / "spec.nc"
    path.write_bytes(fixed + struct.pack(">I", begin) + data)
    with netcdf3.File(path) as f:
        assert (f.version, dict(f.dimensions), dict(f.attributes)) == (1, {"n": 2, "x": 3}, {"title": "hi"})
        v = f.variables["v"]
        assert (v.dimensions, v.shape, v.dtype, v.attributes["scale"]) == (("n", "x"), (2, 3), np.dtype(">f4"), 2.5)
        assert v.data.dtype.byteorder == ">" and not v.data.flags.writeable
        np.testing.assert_array_equal(v.native(), np.array([[1.0, -2.0, 3.5], [4.25, 1e-3, 6e6]], dtype=np.float32))


def _sample_arrays(rng):
    return {
        "f4": (("case", "channel", "y", "x"), rng.standard_normal((5, 2, 3, 7)).astype(np.float32)),
        "f8": (("case", "y"), rng.standard_normal((5, 3))),
        "i4": (("x",), rng.integers(-2 ** 31, 2 ** 31 - 1, 7).astype(np.int32)),
        "i2": (("y", "x"), rng.integers(-30000, 30000, (3, 7)).astype(np.int16)),
        "i1": (("odd",), rng.integers(-128, 127, 5).astype(np.int8)),       # 5 bytes: padded to 8 in the file
    }


@pytest.mark.parametrize("version", [1, 2])
def test_round_trip_and_scipy_reads_our_files(tmp_path, version):
    netcdf_file = pytest.importorskip("scipy.io").netcdf_file
    rng = np.random.default_rng(3)
    arrays = _sample_arrays(rng)
    dims = {"case": 5, "channel": 2, "y": 3, "x": 7, "odd": 5}
    variables = {k: (d, a, {"units": "K", "valid_range": np.array([0.5, 2.5], dtype=np.float32)}) for k, (d, a) in arrays.items()}
    path = str(tmp_path / "ours.nc")
    netcdf3.write(path, dims, variables, attributes={"title": "round trip", "n": 3}, version=version)
    with netcdf3.File(path) as f:
        assert f.version == version and dict(f.dimensions) == dims
        assert f.attributes["title"] == "round trip" and f.attributes["n"] == 3
        for k, (d, a) in arrays.items():
            v = f.variables[k]
            assert v.dimensions == d and v.attributes["units"] == "K"
            np.testing.assert_array_equal(v.attributes["valid_range"], [0.5, 2.5])
            np.testing.assert_array_equal(v.native(), a)
    with netcdf_file(path, "r", mmap=False) as f:
        assert f.version_byte == version and {k: v for k, v in f.dimensions.items()} == dims
        assert f.title == b"round trip"
        for k, (d, a) in arrays.items():
            assert f.variables[k].dimensions == d
            np.testing.assert_array_equal(np.array(f.variables[k][...]), a)
            assert f.variables[k].units == b"K"


@pytest.mark.parametrize("version", [1, 2])
@pytest.mark.parametrize("nrec_vars", [1, 2])
def test_we_read_scipy_files_with_record_variables(tmp_path, version, nrec_vars):
    netcdf_file = pytest.importorskip("scipy.io").netcdf_file
    rng = np.random.default_rng(4)
    path = str(tmp_path / "scipy.nc")
    t = rng.standard_normal((4, 3)).astype(np.float32)
    b = rng.integers(-100, 100, (4, 5)).astype(np.int8)          # 5-byte record slab: padded unless it is alone
    fixed = rng.standard_normal((3, 5))
    with netcdf_file(path, "w", version=version) as f:
        f.createDimension("time", None)
        f.createDimension("x", 3)
        f.createDimension("odd", 5)
        f.history = "made by scipy"
        if nrec_vars == 2:
            vt = f.createVariable("t", "f", ("time", "x"))
            vt[:] = t
            vt.long_name = "temperature"
        vb = f.createVariable("b", "b", ("time", "odd"))
        vb[:] = b
        vf = f.createVariable("fixed", "d", ("x", "odd"))
        vf[:] = fixed
    with netcdf3.File(path) as f:
        assert f.record_dimension == "time" and f.numrecs == 4 and f.dimensions["time"] == 4
        assert f.attributes["history"] == "made by scipy"
        if nrec_vars == 2:
            assert f.variables["t"].is_record and f.variables["t"].attributes["long_name"] == "temperature"
            np.testing.assert_array_equal(f.variables["t"].native(), t)
        np.testing.assert_array_equal(f.variables["b"].native(), b)
        assert not f.variables["fixed"].is_record
        np.testing.assert_array_equal(f.variables["fixed"].native(), fixed)


def test_dataset_file_round_trip_and_multi_file_concat(tmp_path):
    rng = np.random.default_rng(5)
    parts = []
    for i, n in enumerate((3, 2)):
        ds = Dataset(attrs={"source": f"part{i}"})
        ds["lowres"] = DataArray(rng.random((n, 1, 4, 4)).astype(np.float32), dims=("box", "chan", "y", "x"), attrs={"units": "K"})
        ds["flag"] = DataArray(np.arange(n, dtype=np.int64), dims=("box",))
        path = str(tmp_path / f"p{i}.nc")
        ds.to_netcdf(path)
        parts.append((path, ds))
    one = open_dataset(parts[0][0])
    assert one.attrs["source"] == "part0" and one["lowres"].attrs["units"] == "K" and one["lowres"].dims == ("box", "chan", "y", "x")
    assert one["lowres"].raw_values.dtype.byteorder == ">" and not one["lowres"].raw_values.flags.writeable   # the file's bytes
    assert one["lowres"].dtype == np.float32 and one["lowres"].values.dtype.isnative
    np.testing.assert_array_equal(one["lowres"].values, parts[0][1]["lowres"].values)
    assert one["flag"].dtype == np.int32                        # int64 is stored as NetCDF int
    both = open_mfdataset([p for p, _ in parts], concat_dim="box", combine="nested")
    assert both["lowres"].shape == (5, 1, 4, 4)
    np.testing.assert_array_equal(both["lowres"].values, np.concatenate([d["lowres"].values for _, d in parts]))


def test_errors(tmp_path):
    p = tmp_path / "h5.nc"
    p.write_bytes(b"\x89HDF\r\n\x1a\n" + b"\0" * 64)
    with pytest.raises(netcdf3.NetCDFError, match="not a NetCDF-3 file"):
        netcdf3.File(p)
    with pytest.raises(RuntimeError, match="NetCDF-4/HDF5"):
        open_dataset(str(p))
    p = tmp_path / "cdf5.nc"
    p.write_bytes(b"CDF\x05" + b"\0" * 64)
    with pytest.raises(netcdf3.NetCDFError, match="not supported"):
        netcdf3.File(p)
    good = tmp_path / "good.nc"
    netcdf3.write(str(good), {"x": 1000}, {"v": (("x",), np.arange(1000, dtype=np.float32), {})})
    cut = tmp_path / "cut.nc"
    cut.write_bytes(good.read_bytes()[:-100])
    with pytest.raises(netcdf3.NetCDFError, match="past the end"):
        netcdf3.File(cut)
    cut.write_bytes(good.read_bytes()[:20])
    with pytest.raises(netcdf3.NetCDFError, match="truncated"):
        netcdf3.File(cut)
    with pytest.raises(netcdf3.NetCDFError, match="does not match dimensions"):
        netcdf3.write(str(tmp_path / "bad.nc"), {"x": 3}, {"v": (("x",), np.zeros(4, dtype=np.float32), {})})
    with pytest.raises(netcdf3.NetCDFError, match="cannot be stored"):
        netcdf3.write(str(tmp_path / "bad.nc"), {"x": 2}, {"v": (("x",), np.zeros(2, dtype=np.complex64), {})})


def _cf_file(tmp_path):
    """one file with the four CF-packed shapes xarray decodes on open (the reference opens everything through
    xr.open_mfdataset, whose default is mask_and_scale=True: cli/train_cae.py:58-59)"""
    rng = np.random.default_rng(8)
    f = rng.random((3, 1, 4, 4)).astype(np.float32)
    f[1, 0, 2, 3] = -9999.0
    clean = rng.random((3, 1, 4, 4)).astype(np.float32)
    packed = rng.integers(-1000, 1000, (3, 1, 4, 4)).astype(np.int16)
    packed[0, 0, 0, 0] = -32768
    counts = rng.integers(0, 50, (3, 1, 4, 4)).astype(np.int32)
    counts[2, 0, 1, 1] = -1
    dims = {"n": 3, "c": 1, "y": 4, "x": 4}
    d = ("n", "c", "y", "x")
    path = str(tmp_path / "cf.nc")
    netcdf3.write(path, dims, {
        "filled": (d, f, {"_FillValue": np.float32(-9999.0), "units": "K"}),
        "clean": (d, clean, {"_FillValue": np.float32(-9999.0), "missing_value": np.float32(-8888.0)}),
        "packed": (d, packed, {"_FillValue": np.int16(-32768), "scale_factor": np.float64(0.01), "add_offset": np.float64(273.15)}),
        "scaled": (d, packed, {"scale_factor": np.float32(0.5)}),
        "counts": (d, counts, {"missing_value": np.int32(-1)}),
    })
    return path, f, clean, packed, counts


def test_cf_mask_and_scale_decoding(tmp_path):
    (path, f, clean, packed, counts) = _cf_file(tmp_path)
    ds = open_dataset(path)
    filled = ds["filled"]
    assert np.isnan(filled.values[1, 0, 2, 3]) and np.isnan(filled.values).sum() == 1 and filled.dtype == np.float32
    keep = ~np.isnan(filled.values)
    np.testing.assert_array_equal(filled.values[keep], f[keep])
    assert filled.attrs == {"units": "K"}                    # the CF keys move out of attrs, as in xarray
    # a fill value that never occurs leaves the variable as the file's own big-endian bytes (raw upload path)
    assert ds["clean"].raw_values.dtype.byteorder == ">" and ds["clean"].attrs == {}
    np.testing.assert_array_equal(ds["clean"].values, clean)
    p = ds["packed"]
    assert p.dtype == np.float64 and np.isnan(p.values[0, 0, 0, 0]) and np.isnan(p.values).sum() == 1
    ok = ~np.isnan(p.values)
    np.testing.assert_allclose(p.values[ok], packed[ok].astype(np.float64) * 0.01 + 273.15, rtol=0, atol=1e-12)
    s = ds["scaled"]
    assert s.dtype == np.float32
    np.testing.assert_array_equal(s.values, packed.astype(np.float32) * np.float32(0.5))
    c = ds["counts"]
    assert c.dtype == np.float64 and np.isnan(c.values[2, 0, 1, 1]) and np.isnan(c.values).sum() == 1
    # stored values on request
    raw = open_dataset(path, mask_and_scale=False)
    assert raw["filled"].values[1, 0, 2, 3] == -9999.0 and raw["packed"].dtype == np.int16
    assert raw["packed"].attrs["scale_factor"] == 0.01
    # and through the multi-file entry point the CLIs use
    assert np.isnan(open_mfdataset([path, path], concat_dim="box")["filled"].values).sum() == 2


def test_integers_are_never_narrowed_silently(tmp_path):
    path = str(tmp_path / "ints.nc")
    big = np.array([2 ** 40, -5, 7], dtype=np.int64)
    netcdf3.write(path, {"x": 3, "y": 2}, {
        "u8": (("y",), np.array([200, 255], dtype=np.uint8), {}),
        "u16": (("y",), np.array([40000, 65535], dtype=np.uint16), {}),
        "u32": (("y",), np.array([3, 2 ** 31 - 1], dtype=np.uint32), {}),
        "u32big": (("y",), np.array([3, 2 ** 32 - 1], dtype=np.uint32), {}),
        "i64": (("x",), np.array([1, -2, 3], dtype=np.int64), {}),
        "i64big": (("x",), big, {}),
    })
    with netcdf3.File(path) as f:
        got = {k: v.native() for k, v in f.variables.items()}
    np.testing.assert_array_equal(got["u8"], [200, 255])
    assert got["u8"].dtype == np.int16
    np.testing.assert_array_equal(got["u16"], [40000, 65535])
    assert got["u16"].dtype == np.int32
    assert got["u32"].dtype == np.int32 and got["u32"][1] == 2 ** 31 - 1
    assert got["u32big"].dtype == np.float64 and got["u32big"][1] == 2 ** 32 - 1
    assert got["i64"].dtype == np.int32
    assert got["i64big"].dtype == np.float64
    np.testing.assert_array_equal(got["i64big"], big.astype(np.float64))
    with pytest.raises(netcdf3.NetCDFError, match="fit neither"):
        netcdf3.write(path, {"x": 1}, {"v": (("x",), np.array([2 ** 60 + 1], dtype=np.int64), {})})
    with pytest.raises(netcdf3.NetCDFError, match="zero-length"):
        netcdf3.write(path, {"x": 0}, {"v": (("x",), np.zeros(0, dtype=np.float32), {})})


@pytest.mark.gpu
def test_filled_pixel_reaches_dsdataset_as_nan(tmp_path):
    """a _FillValue pixel must raise the reference's NaN error (ds_dataset.py:43-46,56-58), not be trained on"""
    from cae_tools_amd.models.ds_dataset import DSDataset
    (path, *_rest) = _cf_file(tmp_path)
    ds = open_dataset(path)
    with pytest.raises(ValueError, match="input variable filled contains 1 NaN values"):
        DSDataset(ds, ["filled"], "clean")
    with pytest.raises(ValueError, match="output variable contains 1 NaN values"):
        DSDataset(ds, ["clean"], "filled")
    ok = DSDataset(ds, ["clean", "scaled"], "clean")            # decoded (scaled) variables load like any other
    assert ok.get_input_shape() == (2, 4, 4)


@pytest.mark.gpu
def test_big_endian_slab_is_swapped_on_the_gpu(tmp_path, monkeypatch):
    import torch
    from cae_tools_amd import engine as eng
    rng = np.random.default_rng(6)
    a = rng.standard_normal((7, 3, 33, 35)).astype(np.float32)       # 24,255 words: not a multiple of 4
    a.reshape(-1)[:3] = [np.nan, np.inf, -0.0]
    path = str(tmp_path / "v.nc")
    netcdf3.write(path, {"n": 7, "c": 3, "y": 33, "x": 35}, {"v": (("n", "c", "y", "x"), a, {})})
    monkeypatch.setattr(eng, "_STAGE_BYTES", 40000)                      # several staging trips
    with netcdf3.File(path) as f:
        v = f.variables["v"].data
        assert v.dtype == np.dtype(">f4") and v.flags.c_contiguous
        got = eng.upload_f32(v, torch.device("cuda:0"))
        torch.cuda.synchronize()
        assert got.dtype == torch.float32 and tuple(got.shape) == a.shape
        np.testing.assert_array_equal(got.cpu().numpy().view(np.uint32), a.view(np.uint32))     # bit-exact, NaN included
        # anything that is not a contiguous big-endian float32 slab takes the numpy conversion
        sl = eng.upload_f32(v[:, :, ::2], torch.device("cuda:0"))
        np.testing.assert_array_equal(sl.cpu().numpy().view(np.uint32), np.ascontiguousarray(a[:, :, ::2]).view(np.uint32))
