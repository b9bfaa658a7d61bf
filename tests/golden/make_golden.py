#!/usr/bin/env python3
"""Regenerates the golden fixtures in this directory from the reference's own
known-answer images.  Run in the build container only (it reads /root/reference,
which does not exist on the GPU box):

    python tests/golden/make_golden.py

Source of truth: /root/reference/rgbbox.png and /root/reference/irreg.png -- the
500x500 renders the reference README embeds (README.md:21,25).  They are the only
known-answer fixtures the reference holds for the render path.

Output: <scene>_500.npy.gz -- the image decoded to the reference's packed pixel
format (ray.fut:158-162: (r<<16)|(g<<8)|b as int32), row-major from the top row,
shape (500, 500), gzip-compressed .npy.
"""
import gzip
import io
import os
import sys

import numpy as np
from PIL import Image

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


def decode(path):
    img = np.asarray(Image.open(path).convert("RGB")).astype(np.int32)
    return (img[..., 0] << 16) | (img[..., 1] << 8) | img[..., 2]


def main():
    for scene in ("rgbbox", "irreg"):
        px = decode(os.path.join(REF, scene + ".png"))
        assert px.shape == (500, 500), px.shape
        buf = io.BytesIO()
        np.save(buf, px.astype(np.int32))
        out = os.path.join(HERE, scene + "_500.npy.gz")
        with gzip.GzipFile(out, "wb", mtime=0) as f:
            f.write(buf.getvalue())
        print(out, os.path.getsize(out), "bytes")


if __name__ == "__main__":
    sys.exit(main())
