"""ctypes mirrors of the native (C++) step driver and RCCL transport of include/t8gpu_hip.h."""
import ctypes as C

import numpy as np
import torch

from . import hip


class T8gpuHalo(C.Structure):
    _fields_ = [("num_elements", C.c_int32), ("num_ghosts", C.c_int32), ("n_peers", C.c_int32), ("n_send", C.c_int32),
                ("cells_per_element", C.c_int32), ("reserved", C.c_int32), ("peers", C.c_void_p), ("send_off", C.c_void_p), ("recv_off", C.c_void_p), ("send_idx", C.c_void_p),
                ("sendbuf", C.c_void_p), ("recvbuf", C.c_void_p), ("comm", C.c_void_p)]


class NativeComm:
    """One RCCL communicator per process. `broadcast_bytes(b, src)` distributes rank 0's unique id
    (torch.distributed, MPI, ...)."""

    def __init__(self, rank, nranks, broadcast_bytes, timeout_s=180.0):
        """Collective. Never leaves a peer waiting: rank 0 broadcasts even when it could not make an id (an
        empty one, on which every rank raises), and ncclCommInitRank runs in a helper thread so that a rank
        whose peers never arrive raises after `timeout_s` instead of blocking forever."""
        import threading
        lib = hip.lib()
        self.handle = C.c_void_p()
        self.rank, self.nranks = rank, nranks
        idbuf = C.create_string_buffer(128)
        uid, err = b"", None
        if rank == 0:
            try:
                hip.check(lib.t8gpu_hip_comm_unique_id(idbuf))
                uid = bytes(idbuf.raw)
            except Exception as exc:  # noqa: BLE001  (reported after the broadcast, which must happen)
                err = exc
        uid = broadcast_bytes(uid, 0)
        if err is not None:
            raise err
        if not isinstance(uid, (bytes, bytearray)) or len(uid) != 128:
            raise hip.T8gpuHipError("rank 0 could not create an RCCL unique id")
        result = {}
        import torch
        device = torch.cuda.current_device()     # the helper thread starts on device 0: hand it this rank's GPU

        def init():
            torch.cuda.set_device(device)
            h = C.c_void_p()
            result["rc"] = lib.t8gpu_hip_comm_create(C.create_string_buffer(bytes(uid), 128), rank, nranks, C.byref(h))
            result["handle"] = h

        th = threading.Thread(target=init, daemon=True)
        th.start()
        th.join(timeout_s)
        if th.is_alive():
            raise hip.T8gpuHipError(f"ncclCommInitRank did not return within {timeout_s:.0f} s")
        hip.check(result["rc"])
        self.handle = result["handle"]

    def abort(self):
        if self.handle:
            hip.lib().t8gpu_hip_comm_abort(self.handle)
            self.handle = C.c_void_p()

    def destroy(self):
        if self.handle:
            hip.lib().t8gpu_hip_comm_destroy(self.handle)
            self.handle = C.c_void_p()


class NativeHalo:
    """T8gpuHalo descriptor of one partition (host index arrays + device buffers)."""

    def __init__(self, part, dtype, comm):
        self.dtype = dtype
        self.peers = np.ascontiguousarray(part.peers, np.int32)
        self.send_off = np.ascontiguousarray(part.send_off, np.int32)
        self.recv_off = np.ascontiguousarray(part.recv_off, np.int32)
        self.send_idx = torch.from_numpy(np.ascontiguousarray(part.send_idx, np.int32)).cuda()
        n_send = int(part.send_idx.size)
        cells = part.cells_per_element
        self.sendbuf = torch.zeros(max(1, 5 * n_send * cells), dtype=dtype, device="cuda")
        self.recvbuf = torch.zeros(max(1, 5 * part.G * cells), dtype=dtype, device="cuda")
        c = T8gpuHalo()
        c.num_elements, c.num_ghosts, c.n_peers, c.n_send = part.N, part.G, len(self.peers), n_send
        c.cells_per_element = cells
        c.peers = self.peers.ctypes.data
        c.send_off = self.send_off.ctypes.data
        c.recv_off = self.recv_off.ctypes.data
        c.send_idx = self.send_idx.data_ptr()
        c.sendbuf, c.recvbuf = self.sendbuf.data_ptr(), self.recvbuf.data_ptr()
        c.comm = comm.handle
        self.c, self.comm = c, comm

    def exchange(self, planes5, stream=None):
        hip.call("t8gpu_hip_halo_exchange", self.dtype, C.byref(self.c), hip.vars_of(planes5), hip.stream_ptr(stream))


class NativeStepper:
    """t8gpu_hip_plain_stepper_*: the whole iterate() enqueued by one C call."""

    def __init__(self, plan, halo=None):
        self.plan, self.halo = plan, halo
        self.handle = C.c_void_p()
        hip.check(hip.lib().t8gpu_hip_plain_stepper_create(C.byref(plan.c), C.byref(halo.c) if halo is not None else None,
                                                           C.byref(self.handle)))

    def __del__(self):
        if getattr(self, "handle", None):
            hip.lib().t8gpu_hip_plain_stepper_destroy(self.handle)
            self.handle = None

    def iterate(self, solver, delta_t, stream=None):
        hip.call("t8gpu_hip_plain_stepper_iterate", solver.dtype, self.handle, solver.kind, hip.ptr(solver.planes),
                 C.c_size_t(solver.stride), solver.prev, solver.next, hip.fscalar(solver.dtype, delta_t),
                 hip.ptr(solver.speed), hip.stream_ptr(stream))

    def iterate_steps(self, solver, delta_t, n_steps, prev, next, stream=None):
        """n_steps steps in one call; prev / next are the roles of the first step."""
        hip.call("t8gpu_hip_plain_stepper_iterate_steps", solver.dtype, self.handle, solver.kind, hip.ptr(solver.planes),
                 C.c_size_t(solver.stride), prev, next, hip.fscalar(solver.dtype, delta_t), hip.ptr(solver.speed),
                 C.c_int(n_steps), hip.stream_ptr(stream))

    # -- shared by both step drivers (one handle type in the library) --------------------------------------
    def graph(self, enable=None):
        """hipGraph replay of iterate_steps(); returns (captures, replays) so far."""
        counts = (C.c_int * 2)()
        hip.check(hip.lib().t8gpu_hip_plain_stepper_graph(self.handle, -1 if enable is None else int(bool(enable)), counts))
        return counts[0], counts[1]

    def timing(self, enable):
        hip.check(hip.lib().t8gpu_hip_plain_stepper_timing(self.handle, int(enable)))

    def host_time(self, reset=False):
        """(milliseconds, steps): host time the multi-rank driver spent enqueueing and the steps that covers (0, 0 for a
        single-rank stepper)."""
        ms, n = C.c_double(), C.c_longlong()
        hip.check(hip.lib().t8gpu_hip_plain_stepper_host_time(self.handle, int(bool(reset)), C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def timed_stages(self):
        return int(hip.lib().t8gpu_hip_plain_stepper_timed_stages(self.handle))

    def elapsed(self):
        ms, n = C.c_double(), C.c_int()
        hip.check(hip.lib().t8gpu_hip_plain_stepper_elapsed(self.handle, C.byref(ms), C.byref(n)))
        return ms.value, n.value


class NativeSubgridStepper(NativeStepper):
    """t8gpu_hip_subgrid_stepper_*: SubgridCompressibleEulerSolver::iterate enqueued by one C call (block classes on
    three streams, one RCCL exchange of whole ghost blocks per stage)."""

    def __init__(self, plan, halo=None):
        self.plan, self.halo = plan, halo
        self.handle = C.c_void_p()
        hip.check(hip.lib().t8gpu_hip_subgrid_stepper_create(C.byref(plan.c), C.byref(halo.c) if halo is not None else None,
                                                             C.byref(self.handle)))

    def iterate(self, solver, delta_t, stream=None):
        self.iterate_steps(solver, delta_t, 1, solver.prev, solver.next, stream)

    def iterate_steps(self, solver, delta_t, n_steps, prev, next, stream=None):
        hip.call("t8gpu_hip_subgrid_stepper_iterate_steps", solver.dtype, self.handle, solver.kind, hip.ptr(solver.planes),
                 C.c_size_t(solver.stride), hip.ptr(solver.volumes), prev, next, hip.fscalar(solver.dtype, delta_t),
                 C.c_int(n_steps), hip.stream_ptr(stream))


def runtime_versions():
    """{"rccl": (compiled against, bound at run time), "hip": (...)}: t8gpu_hip_runtime_versions (include/t8gpu_hip.h)."""
    v = (C.c_int * 4)()
    hip.check(hip.lib().t8gpu_hip_runtime_versions(v))
    return {"rccl": (v[0], v[1]), "hip": (v[2], v[3])}


def stream_wait(stream, timeout_s):
    """0 = idle, 1 = still busy after timeout_s."""
    return hip.lib().t8gpu_hip_stream_wait(C.c_void_p(stream.cuda_stream), C.c_double(timeout_s))
