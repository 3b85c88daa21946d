#!/bin/bash
# h5import_written.h5: an HDF5 file produced by the HDF Group's own `h5import` tool (/opt/conda/bin in the build image;
# NOT this repo's writer) from raw little-endian arrays: a contiguous float32 (4,6,3), a contiguous int64 (4,) and a
# chunked + gzip-compressed float64 (2,5) inside a group.  expect.npz holds the same arrays.  tests/test_data_path.py
# reads the committed file with the product's libhdf5 binding.
set -e
cd "$(dirname "$0")"
python - <<'P'
import numpy as np
rng = np.random.default_rng(5)
a = (rng.random((4, 6, 3)) - 0.5).astype(np.float32)
l = np.arange(4, dtype=np.int64) * 3 - 2
d = rng.random((2, 5)).astype(np.float64)
a.tofile("/tmp/a.bin"); l.tofile("/tmp/l.bin"); d.tofile("/tmp/d.bin")
np.savez("expect.npz", incomplete_pcds=a, labels=l, data=d)
P
/opt/conda/bin/h5import /tmp/a.bin -c a.cfg /tmp/l.bin -c l.cfg /tmp/d.bin -c d.cfg -o h5import_written.h5
