"""bench.py's bookkeeping that needs no GPU: which committed rocprofv3 summary may ride along in a bench line."""
import json
import os
import sys

import pytest

from conftest import ROOT

sys.path.insert(0, ROOT)


@pytest.fixture()
def bench(tmp_path, monkeypatch):
    import bench as b
    (tmp_path / "profiles").mkdir()
    monkeypatch.setattr(b, "ROOT", str(tmp_path))
    return b


def _summary(tmp_path, tag, workload, kernel, avg_ms, n_gpus=1):
    d = {"workload": workload, "n_gpus": n_gpus, "kernel": f"void mrt::{kernel}(mrt::Params, unsigned int const*)", "avg_ms": avg_ms,
         "derived": {"hbm_traffic_bytes": 123.0, "valu_wave_instr": 2.0e11, "lane_utilisation": 0.5, "mix": {}}}
    json.dump(d, open(os.path.join(str(tmp_path), "profiles", f"{tag}_summary.json"), "w"))


def test_pmc_fields_are_replayed_only_for_the_kernel_that_was_timed(bench, tmp_path):
    """`traffic`, `valu_issue_frac`, `lane_utilisation` come from a committed profile; a profile of another
    instantiation, or of the same one at another speed (> 3 %), must not ride along: null fields + pmc_stale."""
    k = "pt_megakernel<true, 64, 0u>"
    _summary(tmp_path, "r1a", "w", k, 400.0)                 # older, slower
    _summary(tmp_path, "r3a", "w", k, 286.0)                 # the newest one by tag is the one that counts
    r = bench.pmc_replay("w", 1, k, 284.9, False)
    assert not r["pmc_stale"] and r["traffic"] == 123.0 and r["traffic_source"] == "r3a_summary.json"
    assert r["lane_utilisation"] == 0.5 and abs(r["valu_issue_frac"] - 2.0e11 / 0.286 / 1e9 / bench.VALU_ISSUE_ARCH_GINSTR) < 1e-12
    # the same kernel, 5 % faster than the profile: stale
    r = bench.pmc_replay("w", 1, k, 272.0, False)
    assert r["pmc_stale"] and r["traffic"] is None and r["valu_issue_frac"] is None and r["lane_utilisation"] is None and r["valu_pmc"] is None
    assert r["pmc_stale_why"]["profile"] == "r3a_summary.json"
    # another instantiation at the same speed: stale
    r = bench.pmc_replay("w", 1, "pt_megakernel<true, 256, 0u>", 286.0, False)
    assert r["pmc_stale"] and r["traffic"] is None
    # no profile of this workload / world size, or an --spp override: nothing replayed, nothing stale
    for args in (("other", 1, k, 286.0, False), ("w", 2, k, 286.0, False), ("w", 1, k, 286.0, True)):
        r = bench.pmc_replay(*args)
        assert not r["pmc_stale"] and r["traffic"] is None


def test_committed_profiles_name_a_kernel_and_a_duration():
    """Every committed summary bench.py may replay carries what the staleness check needs."""
    import glob
    n = 0
    for f in glob.glob(os.path.join(ROOT, "profiles", "r3*_summary.json")):
        d = json.load(open(f))
        assert "pt_megakernel<" in d["kernel"] and d["avg_ms"] > 0 and d["workload"], f
        n += 1
    assert n >= 3
