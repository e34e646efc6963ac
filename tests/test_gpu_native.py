"""The C++ mirror of Sampler (csrc/sampler.hpp) driving the C ABI like the reference's CLI and HTTP callers."""
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT, make_holder

pytestmark = pytest.mark.gpu


def _fnv(b):
    h = 1469598103934665603
    for x in bytes(b):
        h = ((h ^ x) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    return h


def test_cpp_sampler_matches_python_sampler_and_is_thread_safe():
    exe = os.path.join(ROOT, "tests", "native", "harness")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", os.path.dirname(exe)])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    kv = dict(line.split(" ", 1) for line in out.stdout.strip().splitlines())
    assert kv["updates"] == "4"
    assert kv["img_update"] == kv["img_batched"]                    # per-pass execute == one batched launch
    assert len(set(kv["img_threads"].split())) == 1 and kv["img_threads"].split()[0] == kv["img_update"]
    # the shim's default flags: per-sample calls booked under MRT_FLAG_DEFER give the eager loop's image, and a changed
    # description rebuilds the context while the sums are kept (src/sampler.rs:28, 60-70)
    assert kv["deferred"] == "1 eager 0 same 1" and kv["img_deferred"] == kv["img_update"]
    assert kv["rebuild"] == "contexts 2 count 4 -> 5"
    assert kv["invalidate"] == "contexts 3 count 6"                 # invalidate(): rebuilt although the description did not change
    assert "emit" in kv["error_path"]                               # Err(String) instead of the reference's panic
    from micro_raytracer_amd import Sampler, scenes
    render, _ = make_holder(scenes.default_scene(res=(96, 54), sample=4))
    s = Sampler(seed=7)
    s.execute(render, n_samples=4)
    assert f"{_fnv(s.img().tobytes()):016x}" == kv["img_update"]
