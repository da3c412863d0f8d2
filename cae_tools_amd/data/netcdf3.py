"""Dependency-free NetCDF-3 reader / writer (classic CDF-1 and 64-bit-offset CDF-2).

The reference reads its training / scoring files with xarray.open_mfdataset and writes predictions
with Dataset.to_netcdf (cli/train_cae.py:58-59, cli/apply_cae.py:70-90, gen.py:137-148).  This module is
the file layer under cae_tools_amd.data.arrays: it parses the header with `struct`, and hands every
variable out as a numpy view of ONE read-only memory map of the file - big-endian, exactly as stored.
Fixed-size variables are contiguous slabs, so a (case, channel, y, x) float32 variable goes
file -> pinned host buffer -> HBM as raw bytes and is byte-swapped on the GPU (cae_bswap32,
include/cae_hip.h); nothing touches the values on the host.  Record variables (unlimited dimension) are
strided views over the interleaved records and are gathered by numpy when used.

Format, from the NetCDF classic format specification ("The NetCDF Classic Format Specification",
Unidata): header = magic numrecs dim_list gatt_list var_list; all integers big-endian; names and values
padded to 4 bytes; tags NC_DIMENSION 0x0A, NC_VARIABLE 0x0B, NC_ATTRIBUTE 0x0C; types BYTE 1, CHAR 2,
SHORT 3, INT 4, FLOAT 5, DOUBLE 6; `begin` is 32-bit in CDF-1 and 64-bit in CDF-2.
NetCDF-4 files are HDF5 containers and are not handled here.
"""
import mmap
import os
import struct
from collections import OrderedDict

import numpy as np

NC_DIMENSION, NC_VARIABLE, NC_ATTRIBUTE = 0x0A, 0x0B, 0x0C
_TYPES = {1: ">i1", 2: "S1", 3: ">i2", 4: ">i4", 5: ">f4", 6: ">f8"}
_TYPE_OF = {"i1": 1, "u1": 1, "S1": 2, "i2": 3, "i4": 4, "f4": 5, "f8": 6}
STREAMING = 0xFFFFFFFF


class NetCDFError(ValueError):
    pass


def _pad4(n):
    return (n + 3) & ~3


class Variable:
    """one variable of an open file: .data is a big-endian numpy view of the file mapping"""

    def __init__(self, name, dimensions, shape, dtype, attributes, data, is_record):
        self.name = name
        self.dimensions = tuple(dimensions)
        self.shape = tuple(shape)
        self.dtype = dtype
        self.attributes = attributes
        self.data = data
        self.is_record = is_record

    def native(self):
        """a native-endian, writable, contiguous copy (host byte swap)"""
        return np.ascontiguousarray(self.data).astype(self.data.dtype.newbyteorder("="))


class _Cursor:

    def __init__(self, buf):
        self.buf = buf
        self.pos = 0

    def take(self, fmt):
        size = struct.calcsize(fmt)
        if self.pos + size > len(self.buf):
            raise NetCDFError("truncated NetCDF header")
        vals = struct.unpack_from(fmt, self.buf, self.pos)
        self.pos += size
        return vals if len(vals) > 1 else vals[0]

    def raw(self, n):
        if self.pos + n > len(self.buf):
            raise NetCDFError("truncated NetCDF header")
        b = bytes(self.buf[self.pos:self.pos + n])
        self.pos += _pad4(n)
        return b

    def name(self):
        return self.raw(self.take(">I")).decode("utf-8")


def _read_attributes(cur):
    (tag, count) = cur.take(">II")
    if tag == 0 and count == 0:
        return OrderedDict()
    if tag != NC_ATTRIBUTE:
        raise NetCDFError(f"expected an attribute list, found tag {tag:#x}")
    out = OrderedDict()
    for _ in range(count):
        name = cur.name()
        (nc_type, nelems) = cur.take(">II")
        if nc_type not in _TYPES:
            raise NetCDFError(f"attribute {name}: unknown type {nc_type}")
        dt = np.dtype(_TYPES[nc_type])
        raw = cur.raw(nelems * dt.itemsize)
        if nc_type == 2:
            out[name] = raw.rstrip(b"\x00").decode("utf-8", errors="replace")
        else:
            vals = np.frombuffer(raw, dtype=dt).astype(dt.newbyteorder("="))
            out[name] = vals[0].item() if nelems == 1 else vals
    return out


class File:
    """read-only NetCDF-3 file: .dimensions (name -> length), .attributes, .variables (name -> Variable)"""

    def __init__(self, path):
        self.path = os.fspath(path)
        self._fh = open(self.path, "rb")
        size = os.fstat(self._fh.fileno()).st_size
        if size < 8:
            self._fh.close()
            raise NetCDFError(f"{self.path}: not a NetCDF file (too short)")
        self._map = mmap.mmap(self._fh.fileno(), 0, access=mmap.ACCESS_READ)
        try:
            self._parse(size)
        except Exception:
            self.close()
            raise

    def _parse(self, size):
        cur = _Cursor(self._map)
        magic = cur.raw(4)
        if magic[:3] != b"CDF":
            raise NetCDFError(f"{self.path}: not a NetCDF-3 file (magic {magic!r})")
        self.version = magic[3]
        if self.version not in (1, 2):
            raise NetCDFError(f"{self.path}: NetCDF version byte {self.version} is not supported (CDF-1 / CDF-2 only)")
        off_fmt = ">I" if self.version == 1 else ">Q"
        numrecs = cur.take(">I")
        # dimensions
        (tag, count) = cur.take(">II")
        self.dimensions = OrderedDict()
        dim_names = []
        self.record_dimension = None
        if not (tag == 0 and count == 0):
            if tag != NC_DIMENSION:
                raise NetCDFError(f"expected a dimension list, found tag {tag:#x}")
            for _ in range(count):
                name = cur.name()
                length = cur.take(">I")
                if length == 0:
                    self.record_dimension = name
                dim_names.append(name)
                self.dimensions[name] = length
        self.attributes = _read_attributes(cur)
        # variables
        (tag, count) = cur.take(">II")
        metas = []
        if not (tag == 0 and count == 0):
            if tag != NC_VARIABLE:
                raise NetCDFError(f"expected a variable list, found tag {tag:#x}")
            for _ in range(count):
                name = cur.name()
                ndims = cur.take(">I")
                dimids = [cur.take(">I") for _ in range(ndims)]
                attrs = _read_attributes(cur)
                (nc_type, vsize) = cur.take(">II")
                begin = cur.take(off_fmt)
                if nc_type not in _TYPES:
                    raise NetCDFError(f"variable {name}: unknown type {nc_type}")
                if any(d >= len(dim_names) for d in dimids):
                    raise NetCDFError(f"variable {name}: dimension id out of range")
                metas.append((name, [dim_names[d] for d in dimids], attrs, np.dtype(_TYPES[nc_type]), vsize, begin))
        self.header_bytes = cur.pos
        rec_vars = [m for m in metas if m[1] and m[1][0] == self.record_dimension and self.record_dimension is not None]
        # per-record slab sizes are computed from the shapes (vsize overflows for slabs >= 4 GiB, and the padding rule
        # has the single-record-variable exception)
        def slab_bytes(m):
            inner = [self.dimensions[d] for d in m[1][1:]]
            return int(np.prod(inner, dtype=np.int64)) * m[3].itemsize
        if len(rec_vars) == 1:
            recsize = slab_bytes(rec_vars[0])
        else:
            recsize = sum(_pad4(slab_bytes(m)) for m in rec_vars)
        if numrecs == STREAMING:
            first = min((m[5] for m in rec_vars), default=size)
            numrecs = (size - first) // recsize if recsize else 0
        self.numrecs = numrecs
        if self.record_dimension is not None:
            self.dimensions[self.record_dimension] = numrecs
        self.variables = OrderedDict()
        for m in metas:
            (name, dims, attrs, dt, _, begin) = m
            is_rec = m in rec_vars
            shape = tuple(self.dimensions[d] for d in dims)
            if is_rec:
                inner = shape[1:]
                if numrecs == 0:
                    data = np.empty(shape, dtype=dt)
                else:
                    last = begin + (numrecs - 1) * recsize + slab_bytes(m)
                    if last > size:
                        raise NetCDFError(f"variable {name}: data extends past the end of the file")
                    data = np.ndarray((numrecs,) + inner, dtype=dt, buffer=self._map, offset=begin,
                                      strides=(recsize,) + tuple(np.array(_strides(inner, dt.itemsize), dtype=np.int64)))
            else:
                nbytes = int(np.prod(shape, dtype=np.int64)) * dt.itemsize
                if begin + nbytes > size:
                    raise NetCDFError(f"variable {name}: data extends past the end of the file")
                data = np.ndarray(shape, dtype=dt, buffer=self._map, offset=begin)
            self.variables[name] = Variable(name, dims, shape, dt, attrs, data, is_rec)

    def close(self):
        # numpy views keep the mapping alive; drop ours and let the GC unmap when the last view dies
        self.variables = OrderedDict()
        try:
            self._map.close()
        except BufferError:
            pass  # exported views still exist
        self._fh.close()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


def _strides(shape, itemsize):
    out = []
    acc = itemsize
    for n in reversed(shape):
        out.append(acc)
        acc *= n
    return tuple(reversed(out))


# ---------------------------------------------------------------------------------------------
# writer
# ---------------------------------------------------------------------------------------------

def _enc_name(name):
    b = name.encode("utf-8")
    return struct.pack(">I", len(b)) + b + b"\x00" * (_pad4(len(b)) - len(b))


def _enc_attributes(attrs):
    items = []
    for (key, value) in (attrs or {}).items():
        if isinstance(value, (bytes, str)):
            raw = value.encode("utf-8") if isinstance(value, str) else value
            (nc_type, nelems) = (2, len(raw))
        else:
            arr = np.atleast_1d(np.asarray(value))
            if arr.dtype == bool:
                arr = arr.astype(np.int8)
            if arr.dtype.kind == "i" and arr.dtype.itemsize == 8:
                arr = arr.astype(np.int32) if np.all(np.abs(arr) < 2 ** 31) else arr.astype(np.float64)
            if arr.dtype.kind == "f" and arr.dtype.itemsize == 2:
                arr = arr.astype(np.float32)
            code = arr.dtype.kind + str(arr.dtype.itemsize)
            if code not in _TYPE_OF:
                continue  # objects, unicode arrays ...: not representable in NetCDF-3
            nc_type = _TYPE_OF[code]
            raw = arr.astype(arr.dtype.newbyteorder(">")).tobytes()
            nelems = arr.size
        items.append(_enc_name(key) + struct.pack(">II", nc_type, nelems) + raw + b"\x00" * (_pad4(len(raw)) - len(raw)))
    if not items:
        return struct.pack(">II", 0, 0)
    return struct.pack(">II", NC_ATTRIBUTE, len(items)) + b"".join(items)


def _storable(arr, name="?"):
    """the array in a type NetCDF-3 can hold, never changing a value: bool -> byte, float16 -> float, uint8 -> short,
    uint16 -> int, and uint32 / 64-bit integers -> int when every value fits, double when that is exact, else an error
    (the classic format has no unsigned and no 64-bit integer types)."""
    arr = np.asarray(arr)
    if arr.dtype == bool:
        return arr.astype(np.int8)
    if arr.dtype.kind == "f" and arr.dtype.itemsize == 2:
        return arr.astype(np.float32)
    if arr.dtype.kind == "u" and arr.dtype.itemsize == 1:
        return arr.astype(np.int16)     # NC_BYTE is signed: 200 would read back as -56
    if arr.dtype.kind == "u" and arr.dtype.itemsize == 2:
        return arr.astype(np.int32)
    if arr.dtype.kind in "iu" and arr.dtype.itemsize >= 4 and arr.dtype != np.dtype(np.int32):
        if arr.size == 0:
            return arr.astype(np.int32)
        (lo, hi) = (int(arr.min()), int(arr.max()))
        if -2 ** 31 <= lo and hi < 2 ** 31:
            return arr.astype(np.int32)
        if -2 ** 53 <= lo and hi <= 2 ** 53:
            return arr.astype(np.float64)   # exact for |v| <= 2^53
        raise NetCDFError(f"variable {name}: {arr.dtype} values in [{lo}, {hi}] fit neither NC_INT nor NC_DOUBLE exactly")
    return arr


def write(path, dimensions, variables, attributes=None, version=2):
    """dimensions: name -> length;  variables: name -> (dims, array, attrs).  All variables are fixed-size
    (no record dimension), so each is one contiguous big-endian slab."""
    if version not in (1, 2):
        raise NetCDFError("version must be 1 (classic) or 2 (64-bit offset)")
    dim_names = list(dimensions)
    for (d, n) in dimensions.items():
        if int(n) <= 0:   # length 0 in the header MEANS "the record dimension": a zero-length fixed one cannot be written
            raise NetCDFError(f"dimension {d}: length {n} (NetCDF-3 reads a zero-length dimension as the unlimited one)")
    prepared = []
    for (name, (dims, arr, attrs)) in variables.items():
        arr = _storable(arr, name)
        code = ("S1" if arr.dtype.kind == "S" else arr.dtype.kind + str(arr.dtype.itemsize))
        if code not in _TYPE_OF:
            raise NetCDFError(f"variable {name}: dtype {arr.dtype} cannot be stored in NetCDF-3")
        if len(dims) != arr.ndim or any(dimensions[d] != n for d, n in zip(dims, arr.shape)):
            raise NetCDFError(f"variable {name}: shape {arr.shape} does not match dimensions {dims}")
        prepared.append((name, dims, arr, attrs, _TYPE_OF[code]))
    off_fmt = ">I" if version == 1 else ">Q"

    def header(begins):
        h = b"CDF" + bytes([version]) + struct.pack(">I", 0)
        if dim_names:
            h += struct.pack(">II", NC_DIMENSION, len(dim_names))
            for d in dim_names:
                h += _enc_name(d) + struct.pack(">I", int(dimensions[d]))
        else:
            h += struct.pack(">II", 0, 0)
        h += _enc_attributes(attributes)
        if prepared:
            h += struct.pack(">II", NC_VARIABLE, len(prepared))
            for ((name, dims, arr, attrs, nc_type), begin) in zip(prepared, begins):
                vsize = min(_pad4(arr.nbytes), 0xFFFFFFFF)
                h += _enc_name(name) + struct.pack(">I", len(dims))
                h += b"".join(struct.pack(">I", dim_names.index(d)) for d in dims)
                h += _enc_attributes(attrs) + struct.pack(">II", nc_type, vsize) + struct.pack(off_fmt, begin)
        else:
            h += struct.pack(">II", 0, 0)
        return h

    hlen = len(header([0] * len(prepared)))
    begins = []
    pos = hlen
    for (_, _, arr, _, _) in prepared:
        begins.append(pos)
        pos += _pad4(arr.nbytes)
    if version == 1 and pos > 0xFFFFFFFF:
        raise NetCDFError("file too large for classic format: use version=2")
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    with open(path, "wb") as f:
        f.write(header(begins))
        for (_, _, arr, _, _) in prepared:
            flat = np.ascontiguousarray(arr).reshape(-1)
            be = flat.dtype if flat.dtype.kind == "S" else flat.dtype.newbyteorder(">")
            step = max(1, (64 << 20) // max(flat.dtype.itemsize, 1))      # byte-swap and write 64 MiB at a time
            for lo in range(0, flat.size, step):
                f.write(flat[lo:lo + step].astype(be, copy=False).tobytes())
            f.write(b"\x00" * (_pad4(arr.nbytes) - arr.nbytes))
