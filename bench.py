#!/usr/bin/env python3
"""bench.py — throughput of the path-tracing hot path on N MI355X (one process per GPU, RCCL).

    python bench.py --gpus 1 --steps 3 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Step = one pass of the hot path over one batch: every supersampled pixel of the frame gets `spp`
more path samples (mrt_execute, i.e. `spp` consecutive Sampler::execute passes of the reference,
src/sampler.rs:28-78) and, for N > 1, the one RCCL gather of the shard accumulators to rank 0.
Inputs (the packed scene, the accumulators) are resident in HBM before the timed region.

Workload (BASELINE.json north_star headline): Cornell box (example/CornellBox.json geometry), 1920x1080,
1024 spp, 8 bounces, rows sharded over the N GPUs (strong scaling: the frame is fixed as N grows).
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# dmabuf IPC only on this pool: must be in the environment before the HIP runtime comes up (import torch), under an
# external torchrun as well as under our own launcher
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s
VALU_PEAK_TFLOPS = 157.3       # FP32 vector peak (FMA = 2 flop); 78.6 without fusion, which the math contract forbids
FLOP_PER_SEGMENT = 400.0       # SURVEY.md §8d algorithmic estimate for the Cornell box (10 primitives)

WORKLOADS = {
    # name: (builder kwargs, spp per step)
    "cornell_1080p_1024spp_b8": (dict(kind="cornell_box", res=(1920, 1080), bounce=8), 1024),
    # the same frame driven the way the reference's unmodified callers drive the boundary: one Sampler::execute per sample
    # (src/cli.rs:162-170, src/http.rs:141-144) = 1024 x mrt_execute(ctx, 1) per step
    "cornell_1080p_percall": (dict(kind="cornell_box", res=(1920, 1080), bounce=8), 1024),
    # the same loop with MRT_FLAG_DEFER (what MRT_DEFER=1 gives the unmodified binary): calls only book their sample,
    # the frame is traced in 1024-sample batches when it is observed
    "cornell_1080p_percall_deferred": (dict(kind="cornell_box", res=(1920, 1080), bounce=8), 1024),
    "cornell_512_64spp_b8": (dict(kind="cornell_box", res=(512, 512), bounce=8), 64),           # BASELINE.json configs[1]
    "cornell2_4k_64spp_b16": (dict(kind="cornell_box2", res=(1920, 1080), ssaa=2, bounce=16), 64),  # configs[2] geometry
    # the BASELINE.json configs at their full sizes (one step = the whole render)
    "c1_default_256_1spp_b1": (dict(kind="default_scene", res=(256, 256), ssaa=1, bounce=1), 1),
    "c3_cornell2_1080p_ssaa2_1024spp_b16": (dict(kind="cornell_box2", res=(1920, 1080), ssaa=2, bounce=16), 1024),
    "c4_cornell2_2160p_1024spp_b16": (dict(kind="cornell_box2", res=(3840, 2160), ssaa=1, bounce=16), 1024),
    "c5_mesh_1080p_512spp": (dict(kind="mesh_scene", res=(1920, 1080), ssaa=1, bounce=8), 512),
    "c5_minecraft_1080p_ssaa2_512spp": (dict(kind="minecraft_like", res=(1920, 1080), ssaa=2, bounce=8), 512),
    # meshes the LDS cannot hold (any mesh size is legal input, src/parser.rs:805-824): the Mesh.json scene around 5120 / 20480 triangles
    "mesh5k_1080p_64spp": (dict(kind="mesh_scene", res=(1920, 1080), ssaa=1, bounce=8, n_tris=5120), 64),
    "mesh20k_540p_64spp": (dict(kind="mesh_scene", res=(960, 540), ssaa=1, bounce=8, n_tris=20480), 64),
}


def build_render(spec, spp):
    from micro_raytracer_amd import load_render, scenes
    spec = dict(spec)
    kind = spec.pop("kind")
    return load_render(getattr(scenes, kind)(sample=spp, **spec))


def cpu_baseline(render, seconds_target=12.0):
    """The CPU oracle (a C restatement of the reference's algorithm, kind "port": the Rust reference cannot be built
    here) timed on this host's cores on a bounded sample of the same workload, with the reference's own structure:
    FULL-FRAME sample passes, each cut into n_dim^2 = 4096 tile jobs drawn by a pool of threads that joins after every
    pass (src/sampler.rs:39-74) -- at reduced spp.

    The thread count is measured, not assumed: a container may show 256 CPUs in its affinity mask and be throttled to a
    16-CPU share, where 256 threads meeting at a barrier per pass waste most of their slices.  A short probe on a row
    band picks the fastest of a few pool sizes; a single-thread run of the same band gives the per-core reference
    figure, so the line carries `per_thread` next to `single_thread` (they should agree within ~2x)."""
    from micro_raytracer_amd import _abi
    from oracle import oracle
    aff = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    h = _abi.build_desc(render)
    o = oracle.Oracle(h, seed=1)
    nh, nw = o.nh, o.nw
    band = (nh // 2 - 16, nh // 2 + 16)                     # probe band: 32 rows through the middle of the frame
    band_px = (band[1] - band[0]) * nw
    o.execute(1, threads=1, rows=band)                      # warm-up (page faults)
    t1 = o.execute(1, threads=1, rows=band)
    single = band_px / t1 / 1e6                             # Msamples/s of ONE thread
    spp_probe = max(1, int(0.5 / t1))                       # ~0.5 s of single-thread work per probe, scaled by the pool
    probes = {1: round(single, 4)}
    for t in sorted({min(aff, x) for x in (4, 8, 16, 32, 64, 128, aff)}):
        if t <= 1:
            continue
        n = spp_probe * min(t, 32)
        o.execute(1, threads=t, rows=band)                  # thread creation
        dt = o.execute(n, threads=t, rows=band)
        probes[t] = round(band_px * n / dt / 1e6, 4)
    # `value` is measured with the FASTEST probed pool (the baseline must not be understated); the smallest pool within 15 %
    # of it is reported next to it for information: past the container's CPU share more threads only add SMT siblings and
    # time slicing (GPU box: 16 threads 5.70, 32 threads 6.37, 256 threads 4.88 Msamples/s on the probe band)
    top = max(probes.values())
    best_t = max(probes, key=lambda t: (probes[t], -t))
    lean_t = min(t for t, r in probes.items() if r >= 0.85 * top)
    best_rate = probes[best_t]
    # the measurement: full-frame passes (n_dim = 64 -> 4096 jobs, join per pass) with the best pool, ~seconds_target
    o.reset()
    frame_px = nh * nw
    spp = max(1, min(4096, int(seconds_target * best_rate * 1e6 / frame_px)))
    dt = o.execute(spp, threads=best_t, n_dim=64)
    o.close()
    value = frame_px * spp / dt / 1e6
    return {"value": value, "unit": "Msamples/s", "cores": best_t, "threads": best_t, "affinity_cpus": aff, "kind": "port",
            "per_thread": value / best_t, "single_thread": single, "pool_probe_Msamples_s": probes,
            "smallest_pool_within_15pct": lean_t,
            "sample": f"{spp} full-frame pass(es) of the {nw}x{nh} frame (n_dim 64: 4096 tile jobs, join per pass), {dt:.1f} s, "
                      f"{best_t} threads (the fastest probed pool; {aff} CPUs in the affinity mask)"}


def check_ranks(backend, pg_world, everyone):
    """The `dist` block of a multi-rank bench line, from what every rank reported about itself (all_gather_object).
    Under RCCL ("nccl") every rank must sit on a device of its own: two ranks on one device is a rehearsal layout
    (MRT_SHARE_DEVICE=1 with gloo), not a scaling measurement -- refused with a non-zero exit."""
    ranks = sorted(everyone, key=lambda r: r["rank"])
    if [r["rank"] for r in ranks] != list(range(pg_world)):
        raise SystemExit(f"process group of {pg_world} rank(s) reported ranks {[r['rank'] for r in ranks]}")
    ident = [(r.get("device_uuid") or r.get("pci_bus_id") or "", r["device_index"]) for r in ranks]
    if backend == "nccl" and len(set(ident)) != pg_world:
        raise SystemExit(f"backend nccl with two ranks on one device: {ident}")
    return {"backend": backend, "world_size": pg_world, "devices": ranks, "distinct_devices": len(set(ident))}


VALU_ISSUE_PEAK_GINSTR = 1171.0   # G wave-instructions/s of independent v_mul/v_add/v_fma measured on MI355X
VALU_ISSUE_ARCH_GINSTR = 1228.8   # architectural: 256 CU x 4 SIMD x 2.4 GHz / 2 cycles per wave64 instruction


def pmc_profile(workload, world):
    """The newest committed rocprofv3 summary of this workload (profiles/*_summary.json, written by profiles/summarize.py:
    kernel-trace average + separate --pmc passes; FETCH_SIZE doubled per the gfx950 note of MI355X_MICROARCH.md), or None."""
    import glob
    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_summary.json"))):      # r1a < r2s < r3b: tags sort by round
        try:
            d = json.load(open(f))
        except Exception:
            continue
        if d.get("workload") == workload and d.get("n_gpus", 1) == world:
            best = (d, os.path.basename(f))
    return best


PMC_STALE_TOLERANCE = 0.03


def roofline_block(replay, valu_tflops, hbm_achieved, alg_bytes, alg_bytes_8d, kernel_name, k_ms, k_min, k_max, gather_ms, reduce_ms):
    """`roofline` of the dominant kernel.  The bound that binds is FP32 VALU issue (neither HBM nor MFMA --
    the scene lives in LDS and there is no dense contraction), so `bound` says so and `frac` is
      * with a fresh PMC profile of this very kernel (pmc_replay): the share of the architectural VALU issue rate spent on
        ACTIVE lanes = valu_issue_frac x lane_utilisation (achieved / peak in G wave-instructions/s, lane-weighted);
      * otherwise the section-8d flop model: segments x 400 flop / kernel time against the 157.3 TFLOP/s vector peak.
    The HBM view BASELINE.json asks for stays as the `hbm` sub-block (by construction << 1 %)."""
    hbm = {"bound": "hbm", "achieved": hbm_achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": hbm_achieved / HBM_PEAK_GBS,
           "algorithmic_bytes_per_launch": alg_bytes, "algorithmic_bytes_8d": alg_bytes_8d,
           "traffic": replay["traffic"], "traffic_source": replay["traffic_source"],
           "note": "by construction not HBM-bound (SURVEY.md section 8d): HBM sees 24 B per pixel per launch, or with the sample split one 12 B "
                   "chunk sum per pixel per 16 samples, which reduce_chunks (reduce_ms) folds into the accumulator"}
    model_frac = valu_tflops / VALU_PEAK_TFLOPS
    if replay["valu_issue_frac"] is not None and replay["lane_utilisation"] is not None:
        g = replay["valu_issue_frac"] * VALU_ISSUE_ARCH_GINSTR
        top = {"bound": "valu_issue", "achieved": g * replay["lane_utilisation"], "peak": VALU_ISSUE_ARCH_GINSTR,
               "unit": "G wave-instr/s (active-lane weighted)", "frac": replay["valu_issue_frac"] * replay["lane_utilisation"],
               "frac_source": "pmc"}
    else:
        top = {"bound": "valu_issue", "achieved": valu_tflops, "peak": VALU_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": model_frac,
               "frac_source": "flop_model"}
    top.update({"traffic": replay["traffic"], "valu_model_frac": model_frac, "valu_issue_frac": replay["valu_issue_frac"],
                "lane_utilisation": replay["lane_utilisation"], "pmc_stale": replay["pmc_stale"], "kernel": kernel_name,
                "kernel_ms": k_ms, "kernel_ms_rank_min": k_min, "kernel_ms_rank_max": k_max, "gather_ms": gather_ms,
                "reduce_ms": reduce_ms, "hbm": hbm,
                "note": "kernel_ms is pt_megakernel alone (HIP events on the launch stream)"})
    return top


def pmc_replay(workload, world, kernel_name, kernel_ms, spp_overridden):
    """PMC-derived fields replayed from the committed profile of this workload -- ONLY when that profile is of the kernel
    that was just timed: the same instantiation (pt_megakernel<scene_in_lds, block_threads, FEAT>) and an average launch
    duration within 3 % of the live HIP-event time.  Otherwise every replayed field is null and "pmc_stale" is true:
    a stale profile must not ride along after a kernel change."""
    out = {"traffic": None, "traffic_source": None, "valu_issue_frac": None, "lane_utilisation": None, "valu_pmc": None, "pmc_stale": False}
    prof = None if spp_overridden else pmc_profile(workload, world)
    if not prof:
        return out
    d, src = prof
    ms = d.get("avg_ms") or 0.0
    same_kernel = kernel_name in (d.get("kernel") or "")
    fresh = ms > 0 and kernel_ms > 0 and abs(ms - kernel_ms) <= PMC_STALE_TOLERANCE * kernel_ms
    if not (same_kernel and fresh):
        out.update(pmc_stale=True, pmc_stale_why={"profile": src, "profile_kernel": d.get("kernel"), "profile_avg_ms": ms,
                                                   "timed_kernel": kernel_name, "timed_kernel_ms": kernel_ms})
        return out
    dv = d.get("derived", {})
    if "hbm_traffic_bytes" in dv:
        out.update(traffic=dv["hbm_traffic_bytes"], traffic_source=src)
    if "valu_wave_instr" in dv:
        g = dv["valu_wave_instr"] / (ms * 1e-3) / 1e9
        out.update(valu_issue_frac=g / VALU_ISSUE_ARCH_GINSTR, lane_utilisation=dv.get("lane_utilisation"),
                   valu_pmc={"source": src, "valu_wave_instr_per_launch": dv["valu_wave_instr"], "kernel_ms": ms,
                             "G_wave_instr_per_s": g, "frac_of_measured_issue_peak": g / VALU_ISSUE_PEAK_GINSTR,
                             "lane_utilisation": dv.get("lane_utilisation"), "fp32_tflops_issued": dv.get("fp32_tflops_issued"),
                             "fp32_tflops_useful": dv.get("fp32_tflops_useful"), "instruction_mix": dv.get("mix")})
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="cornell_1080p_1024spp_b8", choices=sorted(WORKLOADS))
    ap.add_argument("--spp", type=int, default=0, help="override samples per step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # Plain `python bench.py --gpus N`: start the N ranks ourselves.  The parent has made no GPU call (torch is not
        # even imported yet), so the children are fresh processes; rank 0's JSON line goes straight to our stdout and
        # the exit code is the launcher's (non-zero if any rank failed).
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        raise SystemExit(subprocess.run(cmd, env=env).returncode)

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the backend has no CPU path)")
    # MRT_DIST_BACKEND=gloo + MRT_SHARE_DEVICE=1 rehearse the N > 1 path on a one-GPU box (all ranks on device 0,
    # the gather staged through host memory); the default is RCCL with one device per rank.
    backend = os.environ.get("MRT_DIST_BACKEND", "nccl")
    if os.environ.get("MRT_SHARE_DEVICE") == "1":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            try:
                dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
            except TypeError:          # a torch without the device_id argument: lazy communicator creation
                dist.init_process_group(backend="nccl")
        else:
            dist.init_process_group(backend=backend)

    # What the process group really is (world > 1): backend, world size as the GROUP reports it, the device of every rank.
    # A launcher that started fewer ranks than --gpus, or two RCCL ranks on one device, must not produce a bench line.
    dist_info = None
    if world > 1:
        pg_world = dist.get_world_size()
        if pg_world != args.gpus:
            raise SystemExit(f"--gpus {args.gpus} but the process group has {pg_world} rank(s)")
        props = torch.cuda.get_device_properties(local_rank)
        mine = {"rank": rank, "local_rank": int(os.environ.get("LOCAL_RANK", "0")), "device_index": local_rank,
                "device_uuid": str(getattr(props, "uuid", "")), "pci_bus_id": getattr(props, "pci_bus_id", None),
                "device_name": props.name}
        everyone = [None] * pg_world
        dist.all_gather_object(everyone, mine)
        dist_info = check_ranks(dist.get_backend(), pg_world, everyone)      # raises SystemExit on a bad layout

    from micro_raytracer_amd.dist import ShardedSampler

    spec, spp = WORKLOADS[args.workload]
    if args.spp:
        spp = args.spp
    render = build_render(spec, spp)
    percall = "_percall" in args.workload
    deferred = args.workload.endswith("_deferred")
    if deferred and world > 1:
        # a sharded rank renders into a bound torch tensor (the RCCL gather sends it without a copy) and MRT_FLAG_DEFER is
        # ignored while the accumulator is visible to the caller: the line would be the eager loop under another name
        raise SystemExit("cornell_1080p_percall_deferred is a single-GPU workload: deferred execution is inactive on a sharded rank")
    # MRT_FLAG_COUNT_SEGMENTS (1): the VALU model needs segments.  Not in the per-call loops: one atomic per wavefront on one
    # counter word (32 400 short wavefronts per 1080p pass) would be what is measured; their segment count comes from one
    # batched pass outside the timed region.  MRT_FLAG_DEFER (4) for the deferred loop.
    flags = (4 if deferred else 0) if percall else 1
    ss = ShardedSampler(render, rank, world, local_rank, seed=1, flags=flags)
    nw, nh = ss.nw, ss.nh

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    seg_per_pass = 0.0
    if percall:                                  # segments of one sample pass, from a counted context (not timed)
        probe = ShardedSampler(render, rank, world, local_rank, seed=1, flags=1)
        probe.execute(16, gather=False)
        seg_per_pass = probe.s.stats()["segments"] / 16.0
        probe.close()

    def step():
        """One step.  Batched: one mrt_execute(ctx, spp).  Per call: spp x mrt_execute(ctx, 1), each synchronous like
        the scoped-pool join of src/sampler.rs:39-74, gathered once at the end of the step; returns (kernel ms, segments)."""
        if not percall:
            ss.execute(spp)
            st = ss.s.stats()
            return st["kernel_ms"], st["segments"]
        for i in range(spp):
            ss.execute(1, gather=(i == spp - 1))
        st = ss.s.stats()                       # stats of the LAST call only (deferred: of the one batch this observation triggers)
        return st["kernel_ms"] * (1 if deferred else spp), seg_per_pass * spp

    for _ in range(args.warmup):
        step()
    sync()
    kernel_ms, gather_ms, segments = [], [], 0
    t0 = time.perf_counter()
    for _ in range(args.steps):
        k, sg = step()
        kernel_ms.append(k)
        gather_ms.append(ss.last_gather_ms)
        segments += sg
    sync()
    elapsed = time.perf_counter() - t0
    k_mean = sum(kernel_ms) / max(1, len(kernel_ms))
    k_min = k_max = k_mean
    if world > 1:
        red_dev = "cuda" if backend == "nccl" else "cpu"
        t = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        seg_t = torch.tensor([float(segments)], dtype=torch.float64, device=red_dev)
        dist.all_reduce(seg_t, op=dist.ReduceOp.SUM)
        total_segments = float(seg_t.item())
        km = torch.tensor([k_mean, -k_mean], dtype=torch.float64, device=red_dev)      # per-rank kernel time: max and min
        dist.all_reduce(km, op=dist.ReduceOp.MAX)
        k_max, k_min = float(km[0].item()), -float(km[1].item())
    else:
        total_segments = float(segments)

    if rank == 0:
        st = ss.s.stats()
        samples = float(nw) * nh * spp * args.steps
        # roofline of the dominant kernel (pt_megakernel) on this rank: algorithmic HBM bytes per launch
        # = accumulator read + write (12 B + 12 B per owned pixel) + one read of the packed scene
        px_local = ss.s.local_rows * nw
        alg_bytes = 24.0 * px_local + st["scene_bytes"]
        alg_bytes_8d = alg_bytes
        if percall and not deferred:
            pass                                 # every pass is a launch: 24 B per pixel + the scene, per launch
        elif st["k_split"] > 1:          # sample split: pt_megakernel writes one f32x3 chunk sum per pixel per 16 samples (reduce_chunks,
            alg_bytes = 12.0 * px_local * ((spp + 15) // 16) + st["scene_bytes"]     # timed apart, folds them into the accumulator)
        k_ms = sum(kernel_ms) / max(1, len(kernel_ms))
        if percall and not deferred:
            k_ms /= spp                          # per launch (estimated from the last call of each step)
        achieved = alg_bytes / (k_ms * 1e-3) / 1e9 if k_ms > 0 else 0.0
        seg_local = segments / max(1, args.steps) / (spp if (percall and not deferred) else 1)
        valu_tflops = seg_local * FLOP_PER_SEGMENT / (k_ms * 1e-3) / 1e12 if k_ms > 0 else 0.0
        kernel_name = f"pt_megakernel<{'true' if st['scene_in_lds'] else 'false'}, {st['block_threads']}, {st['kernel_features']}u>"
        # PMC-derived fields come from the committed profile of this workload and are only shown when that profile is of
        # this very kernel at this very speed (pmc_replay); per-call loops time a different launch than the profile's
        replay = pmc_replay(args.workload, world, kernel_name, k_ms, bool(args.spp))
        line = {
            "metric": "Msamples/sec (res x spp)", "value": samples / elapsed / 1e6, "unit": "Msamples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": args.workload, "scene": spec["kind"],
                       "res": [render.frame.res[0], render.frame.res[1]], "ssaa": render.frame.ssaa, "spp_per_step": spp,
                       "bounce": render.rt.bounce, "samples_per_step": float(nw) * nh * spp,
                       "calls_per_step": spp if percall else 1,
                       "sharding": f"rows, block-cyclic x{ss.shard_rows}, {world} rank(s), 1 RCCL gather/step" if world > 1 else "single GPU"},
            "roofline": roofline_block(replay, valu_tflops, achieved, alg_bytes, alg_bytes_8d, kernel_name, k_ms, k_min, k_max,
                                       sum(gather_ms) / max(1, len(gather_ms)) if world > 1 else 0.0, st.get("reduce_ms")),
            "valu": {"achieved_tflops": valu_tflops, "peak_tflops": VALU_PEAK_TFLOPS, "frac": valu_tflops / VALU_PEAK_TFLOPS,
                     "segments_per_sample": total_segments / samples, "flop_per_segment_model": FLOP_PER_SEGMENT,
                     "Gsegments_per_s": total_segments / elapsed / 1e9},
            "kernel": {"block_threads": st["block_threads"], "lds_bytes": st["lds_bytes"], "scene_bytes": st["scene_bytes"]},
        }
        # Sampler::img on the accumulated frame (outside the timed region): tone map (+ Lanczos3 when ssaa != 1)
        img = ss.img()
        ist = ss.s.stats()
        img_bytes = 15.0 * nw * nh + (0 if (nw, nh) == tuple(render.frame.res) else 3.0 * nw * nh + 2 * 12.0 * nw * render.frame.res[1] + 3.0 * render.frame.res[0] * render.frame.res[1])
        line["img"] = {"kernels_ms": ist["img_ms"], "algorithmic_bytes": img_bytes, "GBps": img_bytes / (ist["img_ms"] * 1e-3) / 1e9 if ist["img_ms"] > 0 else None,
                       "out": [int(img.shape[1]), int(img.shape[0])]}
        if replay["valu_pmc"]:
            line["roofline"]["valu_pmc_source"] = replay["valu_pmc"]["source"]
            line["valu"]["pmc"] = replay["valu_pmc"]
        if replay["pmc_stale"]:
            line["roofline"]["pmc_stale_why"] = replay["pmc_stale_why"]
        if dist_info is not None:
            pr = ss.s.padded_rows()
            dist_info.update(gather_bytes_per_rank=pr * nw * 12, use_all_gather=bool(ss.use_all_gather), shard_rows=ss.shard_rows,
                             exchange="one gather of the padded shard accumulators to rank 0 per step (reference src/sampler.rs:60-70)")
            line["dist"] = dist_info
        if deferred:
            line["config"]["deferred_active"] = bool(st.get("deferred"))
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(render)
        print(json.dumps(line), flush=True)
    ss.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
