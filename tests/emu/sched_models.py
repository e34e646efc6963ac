"""Schedule models for the BVH scenes (not collected by pytest).

    make -C tests/emu libsched.so && python tests/emu/sched_models.py

Per-path work sequences (closest-hit and shadow work per segment) are recorded on the CPU (sched_probe.cpp) for 25 wave
tiles per scene and replayed through models of how a wavefront could schedule them:
  current      the loop as it is: an iteration costs max(closest) + max(shadow) + shade
  mappings     what a wavefront's 64 lanes stand for: 64 pixels with sequential samples (as built), 64 sample streams of
               one pixel, 4 pixels x 16 streams, 16 pixels x 4 streams
  merged       a lane walks its shadow ray and its closest-hit ray in one loop: max(closest + shadow)
  state machine  bounded walk slices of B work units, shading only when T lanes are ready (or nobody walks)
  union        packet traversal: the wavefront visits the union of the lanes' nodes
Numbers are speed-ups over `current` (ideal = every lane busy all the time).
"""
import ctypes as C
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from micro_raytracer_amd import _abi, load_render, scenes  # noqa: E402

L = C.CDLL(os.path.join(HERE, "libsched.so"))
SHADE = 30          # work units of the shading block


def paths(desc, x0, y0, w, h, S, cap=24):
    hd = _abi.build_desc(load_render(desc))
    out = np.zeros((w * h * S, cap), np.uint32)
    nit = np.zeros(w * h * S, np.uint32)
    L.probe_paths(C.cast(hd.ptr(), C.c_void_p), C.c_uint64(1), S, x0, y0, w, h, cap,
                  out.ctypes.data_as(C.POINTER(C.c_uint32)), nit.ctypes.data_as(C.POINTER(C.c_uint32)))
    return out.reshape(h * w, S, cap), nit.reshape(h * w, S)


def lane_seq(out, nit, stream):
    a = []
    for p, s in stream:
        a.extend(int(v) for v in out[p, s, :nit[p, s]])
    return a


def cost_current(seqs):
    mx = max(len(a) for a in seqs)
    return sum(max(a[k] & 0xffff for a in seqs if k < len(a)) + max(a[k] >> 16 for a in seqs if k < len(a)) + SHADE for k in range(mx))


def cost_merged(seqs):
    mx = max(len(a) for a in seqs)
    return sum(max((a[k] & 0xffff) + (a[k] >> 16) for a in seqs if k < len(a)) + SHADE for k in range(mx))


def ideal(seqs):
    return sum(sum((v & 0xffff) + (v >> 16) + SHADE for v in a) for a in seqs) / 64.0


def cost_state_machine(seqs, B, T):
    qs = []
    for a in seqs:
        q = []
        for v in a:
            q.append(v & 0xffff)
            if v >> 16:
                q.append(-(v >> 16))        # a shadow query: no shading pass of its own before the next closest query
        qs.append(q)
    n = len(qs)
    pos = [0] * n
    rem = [abs(q[0]) if q else 0 for q in qs]
    state = ["walk" if q else "done" for q in qs]
    cost = 0
    while any(s != "done" for s in state):
        walking = [i for i in range(n) if state[i] == "walk"]
        if walking:
            sl = max(1, min(B, max(rem[i] for i in walking)))
            cost += sl
            for i in walking:
                rem[i] -= sl
                if rem[i] <= 0:
                    state[i] = "ready"
        ready = [i for i in range(n) if state[i] == "ready"]
        walking = [i for i in range(n) if state[i] == "walk"]
        if ready and (len(ready) >= T or not walking):
            cost += SHADE
            for i in ready:
                while True:
                    pos[i] += 1
                    if pos[i] >= len(qs[i]):
                        state[i] = "done"
                        break
                    rem[i] = abs(qs[i][pos[i]])
                    if rem[i] > 0:
                        state[i] = "walk"
                        break
    return cost


def run(name, desc, tiles, S=48):
    tot = {}
    for tx, ty in tiles:
        out, nit = paths(desc, tx * 8, ty * 8, 8, 8, S)
        seqs = [lane_seq(out, nit, [(p, s) for s in range(S)]) for p in range(64)]
        r = {"current": cost_current(seqs), "merged": cost_merged(seqs), "ideal": ideal(seqs)}
        c = 0                                               # a wavefront = 64 sample streams of one pixel
        for p in range(64):
            c += cost_current([lane_seq(out, nit, [(p, s) for s in range(l, S, 64)]) for l in range(64)])
        r["map 1 pixel x 64"] = c
        c = 0
        for g in range(16):
            px = [(g % 4) * 2 + (g // 4) * 16, (g % 4) * 2 + 1 + (g // 4) * 16, (g % 4) * 2 + 8 + (g // 4) * 16, (g % 4) * 2 + 9 + (g // 4) * 16]
            c += cost_current([lane_seq(out, nit, [(px[l // 16], s) for s in range(l % 16, S, 16)]) for l in range(64)])
        r["map 4 pixels x 16"] = c
        for B in (8, 16, 32):
            for T in (16, 32, 48):
                r[f"state machine B{B} T{T}"] = cost_state_machine(seqs, B, T)
        for k, v in r.items():
            tot[k] = tot.get(k, 0) + v
    print(name, {k: round(tot["current"] / v, 2) for k, v in tot.items()})


def union(name, desc, tiles, S=32):
    hd = _abi.build_desc(load_render(desc))
    out = (C.c_double * 7)()
    tot = np.zeros(7)
    for tx, ty in tiles:
        L.probe_union(C.cast(hd.ptr(), C.c_void_p), C.c_uint64(1), S, tx, ty, out)
        tot += np.array(list(out))
    it = tot[0]
    print(f"{name}: triangle-BVH nodes per wave iteration: closest max {tot[1] / it:.1f} union {tot[2] / it:.1f} mean {tot[5] / it:.1f} | "
          f"shadow max {tot[3] / it:.1f} union {tot[4] / it:.1f} mean {tot[6] / it:.1f}")


if __name__ == "__main__":
    tiles = [(x, y) for x in range(20, 240, 45) for y in range(10, 135, 25)]
    run("mesh", scenes.mesh_scene(res=(1920, 1080), sample=64), tiles)
    run("minecraft-shaped", scenes.minecraft_like(res=(1920, 1080), ssaa=2, sample=64), [(2 * x, 2 * y) for x, y in tiles])
    union("mesh", scenes.mesh_scene(res=(1920, 1080), sample=32), [(x, y) for x in range(20, 240, 30) for y in range(10, 135, 18)])
