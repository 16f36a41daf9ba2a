#!/usr/bin/env python3
"""npz -> HDF5 in the layout the DLRM driver's --dataset expects: the counterpart of the reference's
examples/cpp/DLRM/preprocess_hdf.py (same three datasets and conversions: X_cat -> int64,
X_int -> log(float32(x) + 1), y -> float32), written through libhdf5 directly since h5py is not
in this image."""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dlrm_flexflow_amd import hdf5_lite  # noqa: E402


def convert(src: str, dst: str) -> None:
    f = np.load(src)
    x_cat = f["X_cat"].astype(np.int64)
    x_int = np.log(f["X_int"].astype(np.float32) + 1)
    y = f["y"].astype(np.float32)
    hdf5_lite.write(dst, {"X_cat": x_cat, "X_int": x_int, "y": y})


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("-i", "--input", help="Path to input numpy file", required=True)
    ap.add_argument("-o", "--output", help="Path to output HDF file", required=True)
    a = ap.parse_args()
    convert(a.input, a.output)
