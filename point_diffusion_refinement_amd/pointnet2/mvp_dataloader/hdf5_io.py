"""Minimal HDF5 dataset I/O over the HDF5 C library (libhdf5) through ctypes.

The reference reads and writes its data with h5py (mvp_dataset.py:70-127, completion_eval.py:277-318,
generate_samples_distributed.py:40-77): plain N-dimensional numeric datasets at the root of a file
('incomplete_pcds', 'labels', 'complete_pcds', 'data', ...).  h5py is not part of this image, the C library
it wraps is; this module binds the dozen calls needed for exactly that subset -- whole-dataset read and
write of contiguous numeric arrays -- so files are interchangeable with the reference's.  Chunked /
compressed datasets are READ transparently (the library decodes them); datasets are WRITTEN contiguous and
uncompressed, like `h5py.File.create_dataset(name, data=array)` does.

`available()` tells whether libhdf5 could be loaded; shard_io falls back to .npz containers otherwise.
"""
import ctypes
import ctypes.util
import glob
import os

import numpy as np

_LIB = None
_TRIED = False

H5F_ACC_RDONLY, H5F_ACC_TRUNC = 0, 2
H5P_DEFAULT, H5S_ALL = 0, 0
H5T_INTEGER, H5T_FLOAT = 0, 1
hid_t = ctypes.c_int64
hsize_t = ctypes.c_uint64


def _candidates():
    env = os.environ.get("PDR_LIBHDF5")
    if env:
        yield env
    found = ctypes.util.find_library("hdf5")
    if found:
        yield found
    for pat in ("/opt/conda/lib/libhdf5.so*", "/usr/lib/x86_64-linux-gnu/libhdf5*.so*", "/usr/lib/libhdf5.so*",
                "/usr/local/lib/libhdf5.so*"):
        for p in sorted(glob.glob(pat)):
            if "_cpp" not in p and "_hl" not in p and "fortran" not in p:
                yield p


def _load():
    global _LIB, _TRIED
    if _TRIED:
        return _LIB
    _TRIED = True
    for path in _candidates():
        try:
            lib = ctypes.CDLL(path)
            if lib.H5open() < 0:
                continue
        except OSError:
            continue
        for name, res, args in (
                ("H5Fopen", hid_t, [ctypes.c_char_p, ctypes.c_uint, hid_t]),
                ("H5Fcreate", hid_t, [ctypes.c_char_p, ctypes.c_uint, hid_t, hid_t]),
                ("H5Fclose", ctypes.c_int, [hid_t]),
                ("H5Dopen2", hid_t, [hid_t, ctypes.c_char_p, hid_t]),
                ("H5Dcreate2", hid_t, [hid_t, ctypes.c_char_p, hid_t, hid_t, hid_t, hid_t, hid_t]),
                ("H5Dclose", ctypes.c_int, [hid_t]),
                ("H5Dget_space", hid_t, [hid_t]),
                ("H5Dget_type", hid_t, [hid_t]),
                ("H5Dread", ctypes.c_int, [hid_t, hid_t, hid_t, hid_t, hid_t, ctypes.c_void_p]),
                ("H5Dwrite", ctypes.c_int, [hid_t, hid_t, hid_t, hid_t, hid_t, ctypes.c_void_p]),
                ("H5Screate_simple", hid_t, [ctypes.c_int, ctypes.POINTER(hsize_t), ctypes.POINTER(hsize_t)]),
                ("H5Sget_simple_extent_ndims", ctypes.c_int, [hid_t]),
                ("H5Sget_simple_extent_dims", ctypes.c_int, [hid_t, ctypes.POINTER(hsize_t), ctypes.POINTER(hsize_t)]),
                ("H5Sclose", ctypes.c_int, [hid_t]),
                ("H5Tget_class", ctypes.c_int, [hid_t]),
                ("H5Tget_size", ctypes.c_size_t, [hid_t]),
                ("H5Tget_sign", ctypes.c_int, [hid_t]),
                ("H5Tclose", ctypes.c_int, [hid_t]),
                ("H5Lexists", ctypes.c_int, [hid_t, ctypes.c_char_p, hid_t]),
        ):
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = res, args
        _LIB = lib
        break
    return _LIB


def available():
    return _load() is not None


def _native(lib, dtype):
    sym = {"float32": "H5T_NATIVE_FLOAT_g", "float64": "H5T_NATIVE_DOUBLE_g", "int8": "H5T_NATIVE_INT8_g",
           "uint8": "H5T_NATIVE_UINT8_g", "int16": "H5T_NATIVE_INT16_g", "uint16": "H5T_NATIVE_UINT16_g",
           "int32": "H5T_NATIVE_INT32_g", "uint32": "H5T_NATIVE_UINT32_g", "int64": "H5T_NATIVE_INT64_g",
           "uint64": "H5T_NATIVE_UINT64_g"}.get(np.dtype(dtype).name)
    if sym is None:
        raise TypeError("unsupported dtype for HDF5 I/O: %s" % dtype)
    return hid_t.in_dll(lib, sym).value


def _need():
    lib = _load()
    if lib is None:
        raise ImportError("libhdf5 not found (set PDR_LIBHDF5=/path/to/libhdf5.so) -- use .npz shards instead")
    return lib


def read(path, name):
    """Whole dataset `name` of file `path` as a numpy array in the file's own numeric type."""
    lib = _need()
    f = lib.H5Fopen(os.fsencode(path), H5F_ACC_RDONLY, H5P_DEFAULT)
    if f < 0:
        raise OSError("cannot open HDF5 file %s" % path)
    try:
        if lib.H5Lexists(f, name.encode(), H5P_DEFAULT) <= 0:
            raise KeyError("%s has no dataset %r" % (path, name))
        d = lib.H5Dopen2(f, name.encode(), H5P_DEFAULT)
        if d < 0:
            raise KeyError("%s: cannot open dataset %r" % (path, name))
        try:
            sp, tp = lib.H5Dget_space(d), lib.H5Dget_type(d)
            nd = lib.H5Sget_simple_extent_ndims(sp)
            dims = (hsize_t * max(nd, 1))()
            if nd > 0:
                lib.H5Sget_simple_extent_dims(sp, dims, None)
            cls, size, sign = lib.H5Tget_class(tp), lib.H5Tget_size(tp), lib.H5Tget_sign(tp)
            lib.H5Sclose(sp)
            lib.H5Tclose(tp)
            if cls == H5T_FLOAT:
                dt = {4: np.float32, 8: np.float64}.get(size)
            elif cls == H5T_INTEGER:
                dt = {(1, 0): np.uint8, (1, 1): np.int8, (2, 0): np.uint16, (2, 1): np.int16, (4, 0): np.uint32,
                      (4, 1): np.int32, (8, 0): np.uint64, (8, 1): np.int64}.get((size, 1 if sign else 0))
            else:
                dt = None
            if dt is None:
                raise TypeError("%s/%s: only integer / float datasets are supported" % (path, name))
            out = np.empty(tuple(int(dims[i]) for i in range(nd)), dtype=dt)
            if out.size and lib.H5Dread(d, _native(lib, dt), H5S_ALL, H5S_ALL, H5P_DEFAULT,
                                        out.ctypes.data_as(ctypes.c_void_p)) < 0:
                raise OSError("H5Dread failed on %s/%s" % (path, name))
            return out
        finally:
            lib.H5Dclose(d)
    finally:
        lib.H5Fclose(f)


def write(path, arrays):
    """Create / truncate `path` with one contiguous dataset per (name, array) item."""
    lib = _need()
    f = lib.H5Fcreate(os.fsencode(path), H5F_ACC_TRUNC, H5P_DEFAULT, H5P_DEFAULT)
    if f < 0:
        raise OSError("cannot create HDF5 file %s" % path)
    try:
        for name, a in arrays.items():
            a = np.ascontiguousarray(a)
            dims = (hsize_t * max(a.ndim, 1))(*a.shape)
            sp = lib.H5Screate_simple(a.ndim, dims, None)
            t = _native(lib, a.dtype)
            d = lib.H5Dcreate2(f, name.encode(), t, sp, H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT)
            if d < 0:
                lib.H5Sclose(sp)
                raise OSError("cannot create dataset %r in %s" % (name, path))
            rc = lib.H5Dwrite(d, t, H5S_ALL, H5S_ALL, H5P_DEFAULT, a.ctypes.data_as(ctypes.c_void_p)) if a.size else 0
            lib.H5Dclose(d)
            lib.H5Sclose(sp)
            if rc < 0:
                raise OSError("H5Dwrite failed on %s/%s" % (path, name))
    finally:
        lib.H5Fclose(f)
