#!/usr/bin/env python3
"""Generates tests/golden/white_sea_hdf5.npz: the datasets of the reference's white_sea_data.nc (netCDF-4 = HDF5) as the REAL HDF5
library reads them -- `h5dump -b LE` of the image's /opt/conda (libhdf5 1.10) in the build container.  The own minimal HDF5 reader
(host/Hdf5Min.h, hdf5_min.py; the image has no libnetcdf and the GPU box gets no libhdf5) is held to it bit for bit
(tests/test_seanetcdf.py).  Run in the build container:  python tests/golden/make_hdf5_golden.py
"""
import hashlib
import os
import subprocess
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "inputs", "white_sea_data.nc")
H5DUMP = "/opt/conda/bin/h5dump"
out = {}
hdr = subprocess.run([H5DUMP, "-H", SRC], check=True, capture_output=True, text=True).stdout
import re
for name in re.findall(r'DATASET "([^"]+)"', hdr):
    dtype = None
    with tempfile.NamedTemporaryFile(suffix=".bin") as t:
        subprocess.run([H5DUMP, "-d", "/" + name, "-b", "LE", "-o", t.name, SRC], check=True, capture_output=True)
        raw = open(t.name, "rb").read()
    # element type as h5dump reports it
    blk = hdr[hdr.index('DATASET "%s"' % name):]
    dtype = "<f4" if "H5T_IEEE_F32LE" in blk[:blk.index("DATASPACE")] else "<f8"
    a = np.frombuffer(raw, dtype)
    out[name] = a
    out[name + "_sha256"] = np.array(hashlib.sha256(raw).hexdigest())
    print(name, dtype, a.shape, a[:3], out[name + "_sha256"])
np.savez_compressed(os.path.join(HERE, "white_sea_hdf5.npz"), **out)
print(os.path.getsize(os.path.join(HERE, "white_sea_hdf5.npz")), "bytes")
