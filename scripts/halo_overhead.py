#!/usr/bin/env python3
"""One-GPU estimate of what the halo machinery costs a rank of the 8-way strong-scaled c4 run.

Takes rank 3's share of the c4 mesh split 8 ways (1.24 M elements, ~3 k ghosts) and times the native C++
stepper (a) without any exchange and (b) with the full per-stage sequence of the multi-rank driver
(csrc/hip/stepper.hip: RCCL grouped send/recv -> ghost-reading tiles on the comm lane beside the interior
tiles on the deep lane; T8GPU_STEPPER=legacy T8GPU_PLAN_CLASSES=3: the three-stream pipeline of rounds 1-3
with its pack / unpack kernels) -- where every peer is mapped onto this rank itself (RCCL self send/recv,
message sizes made symmetric). The payload does not cross xGMI, so (b) - (a) is the launch / synchronisation
overhead of the overlap scheme, not link time; the ghost VALUES are meaningless here and the result of (b) is
not checked. T8GPU_STEPPER_PROFILE=1 adds the host cost of (c) by call category.
usage: halo_overhead.py [world=8] [rank=3] [steps=200] [one_gpu_ms_per_step=0.92]"""
import sys
import time
import types

import numpy as np
import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from t8gpu_amd import native  # noqa: E402
from t8gpu_amd.solver import PlainSolver  # noqa: E402
from t8gpu_amd.synth import SynthMesh  # noqa: E402


def main():
    world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    rank = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 200
    # T8GPU_HALO_WORKLOAD=c5: the 3D hexahedral AMR mesh of bench.py's c5 instead of c4 (pass its one-GPU ms/step as argument 4)
    if __import__("os").environ.get("T8GPU_HALO_WORKLOAD") == "c5":
        mesh = SynthMesh(3, 6, 8, band=0.05)
    else:
        mesh = SynthMesh(2, 7, 12, band=0.1472)
    part = mesh.partition(rank, world)
    dt = 0.1 * 2.0 ** -mesh.finest_level
    print(f"rank {rank}/{world}: N={part.N} G={part.G} peers={part.peers.tolist()}", flush=True)

    def timed(solver, many=False):
        for _ in range(20):
            solver.iterate(dt)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        if many:
            solver.iterate_steps(steps, dt)
        else:
            for _ in range(steps):
                solver.iterate(dt)
        timed.host_ms = (time.perf_counter() - t0) / steps * 1e3     # time to ENQUEUE a step
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / steps * 1e3

    a = PlainSolver(part, torch.float64, mode="fused")
    a.use_native_stepper()
    ta = timed(a)
    print(f"(a) no exchange            : {ta:.4f} ms/step  ({a.plan.host.ntiles} tiles, {a.plan.host.n_interior} interior)", flush=True)

    # symmetric self-exchange: message j carries min(send_j, recv_j) elements both ways
    m = np.minimum(np.diff(part.send_off), np.diff(part.recv_off))
    fake = types.SimpleNamespace(N=part.N, G=int(m.sum()), cells_per_element=1, peers=np.zeros(len(m), np.int32),
                                 send_off=np.concatenate([[0], np.cumsum(m)]).astype(np.int32),
                                 recv_off=np.concatenate([[0], np.cumsum(m)]).astype(np.int32),
                                 send_idx=np.concatenate([part.send_idx[part.send_off[j]: part.send_off[j] + m[j]] for j in range(len(m))]).astype(np.int32))
    comm = native.NativeComm(0, 1, lambda b, src: b)
    halo = native.NativeHalo(fake, torch.float64, comm)
    # the exchange alone: latency of one (pack, grouped send/recv, unpack) sequence, and back-to-back rate
    planes = a.step_planes(a.next)
    for _ in range(10):
        halo.exchange(planes)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(100):
        halo.exchange(planes)
        torch.cuda.synchronize()
    lat = (time.perf_counter() - t0) / 100 * 1e6
    t0 = time.perf_counter()
    for _ in range(300):
        halo.exchange(planes)
    torch.cuda.synchronize()
    rate = (time.perf_counter() - t0) / 300 * 1e6
    print(f"exchange alone: {lat:.1f} us with a sync after each, {rate:.1f} us back to back", flush=True)
    b = PlainSolver(part, torch.float64, mode="fused")
    b.use_native_stepper(halo)
    tb = timed(b)
    if native.stream_wait(torch.cuda.current_stream(), 30.0) != 0:
        comm.abort()
        sys.exit("exchange did not drain")
    print(f"(b) self-exchange per stage: {tb:.4f} ms/step  ({len(m)} messages of {m.tolist()} elements; "
          f"{b.plan.host.n_deep} deep / {b.plan.host.n_interior - b.plan.host.n_deep} near-boundary / "
          f"{b.plan.host.ntiles - b.plan.host.n_interior} ghost-reading tiles)", flush=True)
    import ctypes as C
    from t8gpu_amd import hip as _hip
    _hip.lib().t8gpu_hip_stepper_host_profile(1, None, None)
    tc = timed(b, many=True)
    ns, calls = (C.c_double * 4)(), (C.c_longlong * 4)()
    if _hip.lib().t8gpu_hip_stepper_host_profile(1, ns, calls):
        for c, name in enumerate(("kernel launches", "RCCL groups", "event records", "stream waits")):
            print(f"    host profile of (c): {name:16s} {calls[c]:6d} calls, {ns[c] / max(1, calls[c]) * 1e-3:6.2f} us each, "
                  f"{ns[c] * 1e-3 / (steps + 20):7.2f} us per step", flush=True)
    print(f"(c) same, all steps in one call: {tc:.4f} ms/step; the host needs {timed.host_ms:.4f} ms to enqueue a step", flush=True)
    if __import__("os").environ.get("T8GPU_HALO_ONLY") == "c":   # (kernel traces: the window ends with the direct enqueue)
        del a, b
        comm.destroy()
        return
    # (d) the same call through a hipGraph (opt-in: T8GPU_GRAPH_RCCL=1 -- by default a stepper with a halo enqueues directly):
    # captured once (RCCL groups included, on the capture's origin stream), then ONE hipGraphLaunch per call
    td = float("nan")
    if __import__("os").environ.get("T8GPU_GRAPH_RCCL") == "1":
        b.stepper.graph(True)
        b.iterate_steps(steps, dt)                       # capture
        torch.cuda.synchronize()
        td = timed(b, many=True)
        print(f"(d) same through a hipGraph replay: {td:.4f} ms/step; the host needs {timed.host_ms:.4f} ms per step; "
              f"captures / replays = {b.stepper.graph()}", flush=True)
        b.stepper.graph(False)
    print(f"overhead of the overlap scheme: {tb - ta:+.4f} ms/step = {(tb - ta) / 3 * 1e3:+.1f} us/stage; "
          f"8-way ideal would be {1.0:.2f}x of (a), this is {tb / ta:.3f}x", flush=True)
    one = float(sys.argv[4]) if len(sys.argv) > 4 else 0.92   # ms/step of the whole mesh on one GPU (bench.py c4, round 3)
    print(f"projected strong-scaling speedup at {world} ranks if every rank behaves like this one: "
          f"{one / tb:.2f}x direct enqueue per step, {one / tc:.2f}x all steps in one call"
          + (f", {one / td:.2f}x graph replay" if td == td else "") +
          f" (no exchange: {one / ta:.2f}x) against {one} ms/step on one GPU -- a PROJECTION from one GPU: no byte crosses xGMI here", flush=True)
    del a, b                                        # (T8GPU_STEPPER_PROFILE=1: the steppers print their host-cost profile here)
    import gc
    gc.collect()
    comm.destroy()
    print("done", flush=True)


if __name__ == "__main__":
    if __import__("os").environ.get("T8GPU_HALO_OWN_STREAM") == "1":     # the caller's stream is not the legacy default stream
        with torch.cuda.stream(torch.cuda.Stream()):
            main()
    else:
        main()
