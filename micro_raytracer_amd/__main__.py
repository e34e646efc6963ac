"""Harness entry point (not the reference's CLI, which stays Rust): render a reference-format JSON description.

    python -m micro_raytracer_amd scene.json -o out.png [--sample N] [--bounce N] [--seed S] [--update]

Mirrors CLI::raytrace (src/cli.rs:155-177): per-sample loop with optional --update saves, then the final image.
"""
import argparse
import sys
import time

from . import _lib, load_render
from .sampler import Sampler


def main(argv=None):
    ap = argparse.ArgumentParser(prog="python -m micro_raytracer_amd")
    ap.add_argument("full", help="full render description (JSON, the reference's schema)")
    ap.add_argument("-o", "--output", default="out.png", help=".png or .ppm")
    ap.add_argument("--sample", type=int)
    ap.add_argument("--bounce", type=int)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("-u", "--update", action="store_true", help="save the image after every sample pass")
    a = ap.parse_args(argv)
    render = load_render(a.full)
    if a.sample is not None:
        render.rt.sample = a.sample
    if a.bounce is not None:
        render.rt.bounce = a.bounce
    s = Sampler(seed=a.seed)
    t0 = time.perf_counter()
    if a.update:
        for _ in range(render.rt.sample):
            s.execute(render)
            _lib.save_image(a.output, s.img())
    else:
        s.execute(render, n_samples=render.rt.sample)
    _lib.save_image(a.output, s.img())
    st = s.stats()
    print(f"done: {s.nw}x{s.nh} x {render.rt.sample} spp in {time.perf_counter() - t0:.3f} s -> {a.output} "
          f"({st['block_threads']}-thread workgroups, {st['lds_bytes']} B LDS)", file=sys.stderr)


if __name__ == "__main__":
    main()
