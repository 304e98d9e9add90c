"""Ghost-layer exchange: one rank per GPU, RCCL send/recv over xGMI (torch.distributed's communicator).

Replaces the reference's single-GPU CUDA-IPC pointer sharing + `cudaDeviceSynchronize(); MPI_Barrier`
pairs (examples/compressible_euler/solver.cu:98-99,111-112,130-131,143-144,162-163): per RK stage

    comm stream :  wait(state ready) -> pack kernel -> grouped isend/irecv per neighbour -> unpack kernel
    main stream :  interior tiles ............................ wait(ghosts) -> tiles that read ghost slots

No host barrier inside the step. The halo plan (peers, send lists, ghost ranges) comes from the mesh
provider (csrc/host/synth_mesh.cpp; with real t8code: the forest's ghost layer).

On CUDA tensors the pack/unpack are the HIP kernels of the C-ABI; the torch-indexing fallback exists
only so that the same exchange logic can be exercised by the world_size-2 gloo tests on CPU.
"""
import torch

from . import hip


class HaloExchange:
    def __init__(self, part, dtype, dist, device="cuda", overlap=True, stage_through_host=False):
        self.part, self.dist, self.dtype = part, dist, dtype
        # rehearsal mode: several ranks share one GPU over gloo, which moves host memory only
        self.stage_through_host = stage_through_host
        self.N, self.G = part.N, part.G
        self.rank = part.rank
        self.peers = [int(p) for p in part.peers]
        self.send_off = [int(x) for x in part.send_off]
        self.recv_off = [int(x) for x in part.recv_off]
        self.n_send = int(part.send_idx.size)
        self.cells = S = part.cells_per_element          # 1 (plain) or Subgrid::size: a ghost block mirrors all its cells
        self.on_gpu = str(device).startswith("cuda")
        self.send_idx = torch.from_numpy(part.send_idx.copy()).to(device)
        self.send_cells64 = (self.send_idx.long()[:, None] * S + torch.arange(S, device=device)[None, :]).reshape(-1)
        self.sendbuf = torch.zeros(max(1, 5 * self.n_send * S), dtype=dtype, device=device)
        self.recvbuf = torch.zeros(max(1, 5 * self.G * S), dtype=dtype, device=device)
        self.comm_stream = torch.cuda.Stream() if (self.on_gpu and overlap) else None
        self.ev_state = torch.cuda.Event() if self.on_gpu else None
        self.ev_ghost = torch.cuda.Event() if self.on_gpu else None
        self._reqs = []
        if self.on_gpu:
            hip.lib()

    # -- device-side gather / scatter ------------------------------------------------------------
    def _pack(self, planes5):
        if self.n_send == 0:
            return
        if self.on_gpu:
            hip.call("t8gpu_hip_halo_pack", self.dtype, self.n_send, self.cells, hip.ptr(self.send_idx), hip.vars_of(planes5),
                     hip.ptr(self.sendbuf), hip.stream_ptr())
        else:
            n = self.n_send * self.cells
            self.sendbuf[:5 * n].view(n, 5).copy_(planes5[:, self.send_cells64].t())

    def _unpack(self, planes5):
        if self.G == 0:
            return
        if self.on_gpu:
            hip.call("t8gpu_hip_halo_unpack", self.dtype, self.G, self.N, self.cells, hip.ptr(self.recvbuf),
                     hip.vars_of(planes5), hip.stream_ptr())
        else:
            S = self.cells
            planes5[:, self.N * S:(self.N + self.G) * S] = self.recvbuf[:5 * self.G * S].view(self.G * S, 5).t()

    def _transport(self):
        if self.stage_through_host:
            return self._transport_host_staged()
        ops = []
        for j, p in enumerate(self.peers):
            w = 5 * self.cells
            r0, r1 = w * self.recv_off[j], w * self.recv_off[j + 1]
            s0, s1 = w * self.send_off[j], w * self.send_off[j + 1]
            if r1 > r0:
                ops.append(self.dist.P2POp(self.dist.irecv, self.recvbuf[r0:r1], p))
            if s1 > s0:
                ops.append(self.dist.P2POp(self.dist.isend, self.sendbuf[s0:s1], p))
        if ops:
            for req in self.dist.batch_isend_irecv(ops):
                req.wait()   # on CUDA: enqueues a stream wait, does not block the host

    def _transport_host_staged(self):
        torch.cuda.current_stream().synchronize()
        send_h, recv_h = self.sendbuf.cpu(), torch.empty_like(self.recvbuf, device="cpu")
        ops = []
        w = 5 * self.cells
        for j, p in enumerate(self.peers):
            r0, r1 = w * self.recv_off[j], w * self.recv_off[j + 1]
            s0, s1 = w * self.send_off[j], w * self.send_off[j + 1]
            if r1 > r0:
                ops.append(self.dist.P2POp(self.dist.irecv, recv_h[r0:r1], p))
            if s1 > s0:
                ops.append(self.dist.P2POp(self.dist.isend, send_h[s0:s1], p))
        if ops:
            for req in self.dist.batch_isend_irecv(ops):
                req.wait()
        self.recvbuf.copy_(recv_h)

    # -- one exchange, split so that compute can run between start() and finish() ----------------
    @property
    def overlapped(self):
        """True when start() runs on a second stream (so work can be queued behind the unpack there)."""
        return bool(self.peers) and self.comm_stream is not None

    def start(self, planes5, then=None):
        """planes5: the [5, stride] view of the step whose ghost slots must be refreshed. `then` (optional) is
        called right behind the unpack with the comm stream current: the ghost-reading tiles go there, so
        that they start as soon as the ghosts are in, beside the interior tiles on the main stream."""
        if not self.peers:
            if then is not None:
                then()
            return
        if self.comm_stream is not None:
            self.ev_state.record()                       # everything that produced planes5 is on the main stream
            with torch.cuda.stream(self.comm_stream):
                self.comm_stream.wait_event(self.ev_state)
                self._pack(planes5)
                self._transport()
                self._unpack(planes5)
                if then is not None:
                    then()
                self.ev_ghost.record()
        else:
            self._pack(planes5)
            self._transport()
            self._unpack(planes5)
            if then is not None:
                then()

    def finish(self):
        if self.peers and self.comm_stream is not None:
            torch.cuda.current_stream().wait_event(self.ev_ghost)

    def exchange(self, planes5):
        self.start(planes5)
        self.finish()
