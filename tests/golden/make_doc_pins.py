"""Derive small pin fixtures from the rendered images the reference ships (doc/out0..4.png).

Run here (the reference checkout is not present on the GPU box):
    python tests/golden/make_doc_pins.py /root/reference

The reference has no tests and no golden vectors; its README renders are the only outputs it
holds.  Fixtures are DATA derived from those PNGs (sub-sampled pixels / block means), never
source text:
  out0_grid.npz    doc/out0.png (README.md:127, = example/Default.json, 1280x720, 16 spp)
                   every 4th pixel in x and y, u8.           deterministic pin
  out1_grid.npz    doc/out1.png (README.md:135, 1920x1080, --ssaa 2) every 6th pixel, u8.
                                                             deterministic pin incl. Lanczos3
  out2_blocks.npz  doc/out2.png (README.md:143-154, Cornell box, bounce 16, 1024 spp):
                   inverse-tone-mapped linear radiance averaged over 8x8 blocks centred on
                   (8x, 8y) + a validity mask (no saturated u8 in the block).  statistical pin
  out3_blocks.npz  doc/out3.png (README.md:16-27, CornellBox2 geometry, 1080x1080 ssaa 2) same.
  out4_blocks.npz  doc/out4.png (README.md:11, = example/dof.json, 1280x720, default gamma 0.8 / exp 0.2) same:
                   plane-UV texture lookup, thin-lens DoF with a real aperture, the rolled camera
                   (rotate_y), box + spheres under a point light with shadows.  statistical pin
"""
import os
import sys

import numpy as np
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))


def inv_tonemap(k, gamma, exp):
    """Inverse of Sampler::img's per-channel map (src/sampler.rs:84-96) at the bin centre."""
    t = (k.astype(np.float64) + 0.5) / 255.0
    w = (1.0 - exp) ** 2
    g = (-(1 - t) + np.sqrt((1 - t) ** 2 + 4 * t / w)) * w / 2
    return g ** (1.0 / gamma)


def blocks(path, f, gamma, exp):
    ref = np.asarray(Image.open(path).convert("RGB"))
    hh, ww, _ = ref.shape
    lin = inv_tonemap(ref, gamma, exp)
    sat = (ref >= 254) | (ref <= 1)
    h, w = hh // f, ww // f
    out = np.zeros((h, w, 3), np.float32)
    ok = np.zeros((h, w), bool)
    for y in range(h):
        for x in range(w):
            y0, x0 = max(0, y * f - f // 2), max(0, x * f - f // 2)
            blk = lin[y0:y * f + f // 2, x0:x * f + f // 2]
            if blk.size == 0:
                continue
            out[y, x] = blk.reshape(-1, 3).mean(0)
            ok[y, x] = not sat[y0:y * f + f // 2, x0:x * f + f // 2].any()
    return out, ok


def main(ref_root):
    doc = os.path.join(ref_root, "doc")
    a = np.asarray(Image.open(os.path.join(doc, "out0.png")).convert("RGB"))
    np.savez_compressed(os.path.join(HERE, "out0_grid.npz"), step=4, shape=a.shape, px=a[::4, ::4])
    a = np.asarray(Image.open(os.path.join(doc, "out1.png")).convert("RGB"))
    np.savez_compressed(os.path.join(HERE, "out1_grid.npz"), step=6, shape=a.shape, px=a[::6, ::6])
    b, ok = blocks(os.path.join(doc, "out2.png"), 8, 0.5, 0.75)
    np.savez_compressed(os.path.join(HERE, "out2_blocks.npz"), f=8, lin=b, ok=ok)
    b, ok = blocks(os.path.join(doc, "out3.png"), 8, 0.6, 0.8)
    np.savez_compressed(os.path.join(HERE, "out3_blocks.npz"), f=8, lin=b, ok=ok)
    b, ok = blocks(os.path.join(doc, "out4.png"), 8, 0.8, 0.2)
    np.savez_compressed(os.path.join(HERE, "out4_blocks.npz"), f=8, lin=b, ok=ok)


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "/root/reference")
