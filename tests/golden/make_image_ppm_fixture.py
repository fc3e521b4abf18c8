"""Cuts tests/golden/image_ppm_fixture.npz out of the reference's only known-answer file,
cpu-implementation/image.ppm (P3 header, 256x144, unrounded float body = 255.999 * colour).

The image was written by an early state of the CPU tracer: sky gradient plus one sphere
(centre (0,0,-1), radius 0.5) shaded by its surface normal, 0.5*(N+1).  The fixture keeps
  * the full sphere silhouette (which pixels are not sky) as packed bits,
  * every 5th pixel's three float values, plus every pixel of the rows/columns that cross the
    silhouette edge (rows 30, 72, 113; columns 86, 128, 169).
Run in the build container (needs /root/reference):  python tests/golden/make_image_ppm_fixture.py
"""
import os
import numpy as np

SRC = "/root/reference/cpu-implementation/image.ppm"
DST = os.path.join(os.path.dirname(os.path.abspath(__file__)), "image_ppm_fixture.npz")

lines = open(SRC).read().split("\n")
assert lines[0] == "P3" and lines[1] == "256 144" and lines[2] == "255"
W, H = 256, 144
vals = np.array([[float(t) for t in l.split()] for l in lines[3:3 + W * H]], dtype=np.float64)
assert vals.shape == (W * H, 3)
sphere = vals[:, 2] != 255.999           # sky pixels have blue == 255.999 exactly
keep = np.zeros(W * H, bool)
keep[::5] = True
grid = keep.reshape(H, W)
for r in (30, 72, 113):
    grid[r, :] = True
for c in (86, 128, 169):
    grid[:, c] = True
idx = np.nonzero(keep)[0].astype(np.int32)
np.savez_compressed(DST, width=W, height=H, sphere_bits=np.packbits(sphere), index=idx, values=vals[idx])
print(DST, os.path.getsize(DST), "bytes;", len(idx), "pixels;", int(sphere.sum()), "sphere pixels")
