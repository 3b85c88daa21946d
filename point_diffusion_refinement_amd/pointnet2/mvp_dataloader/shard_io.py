"""Array containers of the data path: '.h5' files through libhdf5 (hdf5_io, interchangeable with the
reference's h5py files) or '.npz' files (numpy, always available).  `resolve(path)` implements the fallback
the dataset uses: the name the reference would open if it exists, else the same name with '.npz'."""
import os

import numpy as np

from . import hdf5_io


def resolve(path):
    if os.path.exists(path):
        return path
    alt = os.path.splitext(path)[0] + ".npz"
    if os.path.exists(alt):
        return alt
    raise FileNotFoundError("%s (or %s)" % (path, alt))


def load_array(path, name):
    path = resolve(path)
    if path.endswith(".npz"):
        with np.load(path) as z:
            if name not in z:
                raise KeyError("%s has no array %r" % (path, name))
            return np.array(z[name])
    return hdf5_io.read(path, name)


def save_arrays(path, arrays):
    """Write {name: array}; an '.h5' target without libhdf5 is written as '.npz' next to it (returned path)."""
    if path.endswith(".npz") or not hdf5_io.available():
        if not path.endswith(".npz"):
            path = os.path.splitext(path)[0] + ".npz"
        np.savez(path, **{k: np.ascontiguousarray(v) for k, v in arrays.items()})
        return path
    hdf5_io.write(path, arrays)
    return path
