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
    for f in glob.glob(os.path.join(ROOT, "profiles", "r[34]*_summary.json")):
        d = json.load(open(f))
        assert "pt_megakernel<" in d["kernel"] and d["avg_ms"] > 0 and d["workload"], f
        n += 1
    assert n >= 3


def test_roofline_names_the_bound_that_binds(bench, tmp_path):
    """the kernel is VALU-issue bound, so `roofline.bound` says so; `frac` is the
    lane-weighted issue fraction when a fresh PMC profile of this kernel exists, the section-8d flop model otherwise; the
    HBM view stays as a sub-block."""
    k = "pt_megakernel<true, 256, 0u>"
    _summary(tmp_path, "r4a", "w", k, 250.0)
    fresh = bench.pmc_replay("w", 1, k, 250.0, False)
    r = bench.roofline_block(fresh, 20.0, 6.0, 1.5e9, 5e7, k, 250.0, 250.0, 250.0, 0.0, 0.4)
    issue = 2.0e11 / 0.250 / 1e9 / bench.VALU_ISSUE_ARCH_GINSTR
    assert r["bound"] == "valu_issue" and r["frac_source"] == "pmc" and abs(r["frac"] - issue * 0.5) < 1e-12
    assert abs(r["achieved"] / r["peak"] - r["frac"]) < 1e-12 and r["peak"] == bench.VALU_ISSUE_ARCH_GINSTR
    assert r["hbm"]["bound"] == "hbm" and r["hbm"]["peak"] == 8000.0 and r["hbm"]["traffic"] == 123.0
    assert abs(r["valu_model_frac"] - 20.0 / 157.3) < 1e-12
    stale = bench.pmc_replay("w", 1, k, 200.0, False)
    r = bench.roofline_block(stale, 20.0, 6.0, 1.5e9, 5e7, k, 200.0, 200.0, 200.0, 0.0, 0.4)
    assert r["bound"] == "valu_issue" and r["frac_source"] == "flop_model" and r["frac"] == r["valu_model_frac"] and r["pmc_stale"]
    assert r["unit"] == "TFLOP/s" and r["peak"] == 157.3 and r["traffic"] is None


def test_a_multi_rank_line_explains_its_process_group_or_refuses(bench):
    """backend, world size as the group reports it, one device per rank -- and no bench line from a
    layout that is not a scaling measurement (two RCCL ranks on one device, missing ranks)."""
    def rk(r, dev, uuid):
        return {"rank": r, "local_rank": r, "device_index": dev, "device_uuid": uuid, "pci_bus_id": None, "device_name": "x"}
    ok = bench.check_ranks("nccl", 4, [rk(2, 2, "c"), rk(0, 0, "a"), rk(3, 3, "d"), rk(1, 1, "b")])
    assert ok["backend"] == "nccl" and ok["world_size"] == 4 and ok["distinct_devices"] == 4 and [d["rank"] for d in ok["devices"]] == [0, 1, 2, 3]
    with pytest.raises(SystemExit):
        bench.check_ranks("nccl", 2, [rk(0, 0, "a"), rk(1, 0, "a")])            # two RCCL ranks on one device
    with pytest.raises(SystemExit):
        bench.check_ranks("nccl", 3, [rk(0, 0, "a"), rk(1, 1, "b"), rk(1, 2, "c")])   # a rank missing / duplicated
    shared = bench.check_ranks("gloo", 2, [rk(0, 0, "a"), rk(1, 0, "a")])       # the one-GPU rehearsal says what it is
    assert shared["distinct_devices"] == 1
    # devices without a uuid (older torch) are told apart by their index
    assert bench.check_ranks("nccl", 2, [rk(0, 0, ""), rk(1, 1, "")])["distinct_devices"] == 2
