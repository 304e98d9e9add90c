"""Host-side mirrors of the reference solvers' hot loop, driving the C-ABI of include/t8gpu_hip.h.

PlainSolver   <-> t8gpu::CompressibleEulerSolver::iterate   (examples/compressible_euler/solver.cu:75-175)
SubgridSolver <-> SubgridCompressibleEulerSolver::iterate   (examples/subgrid/solver.inl:152-266)

Same member names and step bookkeeping (`next`/`prev` swap, Step0..Step3 + Fluxes planes,
plane = step*5 + var, stride = capacity). torch is used for device memory and streams only; every
flux / RK computation is a HIP kernel behind the C-ABI. There is no CPU path.
"""
import ctypes as C

import numpy as np
import torch

from . import hip

STEP0, STEP1, STEP2, STEP3, FLUXES = range(5)


def _timer_begin(solver):
    """bench.py sets solver.kernel_timer to a list to get (start, end) HIP events around the
    dominant kernel, recorded on the stream the kernel is launched on (torch's current stream)."""
    if getattr(solver, "kernel_timer", None) is None:
        return None
    ev = torch.cuda.Event(enable_timing=True)
    ev.record()
    return ev


def _timer_end(solver, ev):
    if ev is not None:
        end = torch.cuda.Event(enable_timing=True)
        end.record()
        solver.kernel_timer.append((ev, end))


def _dev(a, dtype=None):
    from . import hostmem
    return hostmem.to_device(a, dtype)          # (tensor.cuda(), or the application's pinned staging buffer: hostmem.py)


class PlainSolver:
    """Plain elements. mode = "compat": reference data flow (face kernel + atomics, RK kernel);
    mode = "fused": tile kernels (flux + RK in one pass, no flux planes in HBM)."""

    def __init__(self, part, dtype=torch.float64, flux_kind=hip.KEPES, mode="compat", capacity=None, state=None,
                 device=None, plan_options=None):
        if not torch.cuda.is_available():
            raise hip.T8gpuHipError("PlainSolver needs a GPU: the hot path has no CPU implementation")
        hip.lib()
        self.part, self.dtype, self.kind, self.mode = part, dtype, flux_kind, mode
        tot = part.N + part.G
        self.N, self.G, self.F, self.B, self.ndim = part.N, part.G, part.F, part.B, part.normal_dim
        self.stride = capacity or tot
        self.planes = torch.zeros((26, self.stride), dtype=dtype, device="cuda")
        if not (isinstance(state, str) and state == "zeros"):     # "zeros": the caller fills the planes on the device (amr.adapt)
            ic = part.kh_initial_state() if state is None else state
            self.planes[0:5, :tot] = _dev(ic, dtype)
        self.planes[25, :tot] = _dev(part.volumes, dtype)
        self.fn = _dev(part.face_neighbors)
        self.indices = None  # ghosts already resolve to local slots
        self._normals = self._areas = None     # per-face geometry of the compat kernels: uploaded when first asked for
        self.speed = torch.zeros(max(1, part.F + part.B), dtype=dtype, device="cuda")
        self.next, self.prev = STEP0, STEP3  # solver.h:100-101
        self.plan = None
        if mode == "fused":
            from . import fused
            import time
            t0 = time.perf_counter()
            self.plan = fused.PlainPlan(part, dtype, **dict(dict(flux_kind=flux_kind), **(plan_options or {})))
            self.plan_build_s = time.perf_counter() - t0          # host tile plan + its upload (amr.adapt reports it)
        elif mode != "compat":
            raise ValueError(mode)

    @property
    def normals(self):
        """device copy of face_normals (the fused kernels read the tile plan's geometry instead: an adaptive fused run never
        uploads these 24 bytes per face)"""
        if self._normals is None:
            self._normals = _dev(self.part.normals, self.dtype)
        return self._normals

    @property
    def areas(self):
        if self._areas is None:
            self._areas = _dev(self.part.areas, self.dtype)
        return self._areas

    # -- accessors named after the reference API ------------------------------------------------
    def get_own_variables(self, step):
        return hip.vars_of(self.planes, step)

    def get_own_volume(self):
        return self.planes[25]

    def state(self, step=None):
        s = self.next if step is None else step
        return self.planes[5 * s:5 * s + 5, :self.N]

    # -- one flux evaluation + RK stage in the reference's data flow -----------------------------
    def _stage_compat(self, stage, src, dst, dt, stream):
        st, fl = self.get_own_variables(src), self.get_own_variables(FLUXES)
        ev = _timer_begin(self)
        hip.call("t8gpu_hip_flux_faces", self.dtype, self.kind, self.F, self.ndim, hip.ptr(self.fn), None,
                 hip.ptr(self.normals), hip.ptr(self.areas), st, fl, hip.ptr(self.speed), stream)
        _timer_end(self, ev)
        if self.B > 0:
            hip.call("t8gpu_hip_flux_boundary", self.dtype, self.kind, self.F, self.B, self.ndim, hip.ptr(self.fn),
                     hip.ptr(self.normals), hip.ptr(self.areas), st, fl, hip.ptr(self.speed), stream)
        hip.call("t8gpu_hip_rk3_stage", self.dtype, stage, self.N, self.get_own_variables(self.prev), st,
                 self.get_own_variables(dst), fl, hip.ptr(self.planes[25]), hip.fscalar(self.dtype, dt), stream)

    # -- scalar diagnostics of the reference solver, computed on the device -----------------------
    def _reduce_buffers(self):
        if not hasattr(self, "_ws"):
            n = hip.lib().t8gpu_hip_reduce_workspace_bytes
            n.restype = C.c_size_t
            self._ws = torch.zeros(n() // 8, dtype=torch.float64, device="cuda")
            self._scalar = torch.zeros(1, dtype=torch.float64, device="cuda")
        return self._ws, self._scalar

    def compute_integral(self, variable=0, step=None):
        """sum(volume * variable) over the owned elements (CompressibleEulerSolver::compute_integral)."""
        ws, res = self._reduce_buffers()
        s = self.next if step is None else step
        hip.call("t8gpu_hip_integral", self.dtype, C.c_size_t(self.N), 1, hip.ptr(self.planes[5 * s + variable]),
                 hip.ptr(self.planes[25]), hip.ptr(ws), hip.ptr(res), hip.stream_ptr())
        return float(res.item())

    def max_speed(self):
        """max of the per-face wave-speed estimates of the last stage (input of compute_timestep)."""
        ws, res = self._reduce_buffers()
        hip.call("t8gpu_hip_max_speed", self.dtype, C.c_size_t(self.F + self.B), hip.ptr(self.speed), hip.ptr(ws),
                 hip.ptr(res), hip.stream_ptr())
        return float(res.item())

    def compute_timestep(self, cfl=0.7, max_level=None, dist=None):
        """cfl * 0.5^max_level / max speed (solver.cu:213-229); `dist` all-reduces the maximum over ranks."""
        speed = torch.tensor([self.max_speed()], dtype=torch.float64, device="cuda" if dist is None or dist.get_backend() == "nccl" else "cpu")
        if dist is not None:
            dist.all_reduce(speed, op=dist.ReduceOp.MAX)
        level = self.part.mesh.finest_level if max_level is None else max_level
        return cfl * 0.5 ** level / float(speed.item())

    def begin_step(self):
        self.next, self.prev = self.prev, self.next  # solver.cu:76

    def stage_steps(self, k):
        """(source step, destination step) of RK stage k = 0, 1, 2 (solver.cu:81-174)."""
        return (self.prev, STEP1, STEP2)[k], (STEP1, STEP2, self.next)[k]

    def step_planes(self, step):
        return self.planes[5 * step:5 * step + 5]

    def run_stage(self, k, delta_t, stream=None, halo=None, split=False):
        """Flux evaluation on the stage's source state + RK update. With a halo exchange (or split=True)
        the fused kernels run the interior tiles first and the ghost-reading tiles after finish()."""
        s = hip.stream_ptr(stream)
        src, dst = self.stage_steps(k)
        ni, nt = (self.plan.host.n_interior, self.plan.host.ntiles) if self.mode == "fused" else (0, 0)
        if halo is not None and halo.overlapped and 0 < ni < nt:
            # boundary pipeline on the comm stream: ghosts, then the tiles that read them; interior tiles beside it
            halo.start(self.step_planes(src), then=lambda: self.plan.stage(self, k + 1, src, dst, delta_t, hip.stream_ptr(), ni, nt - ni))
            self.plan.stage(self, k + 1, src, dst, delta_t, s, 0, ni)
            halo.finish()
            return
        if halo is not None:
            halo.start(self.step_planes(src))
        if self.mode == "compat":
            if halo is not None:
                halo.finish()
            self._stage_compat(k + 1, src, dst, delta_t, s)
            return
        ni, nt = self.plan.host.n_interior, self.plan.host.ntiles
        if (halo is None and not split) or ni == nt or ni == 0:
            if halo is not None:
                halo.finish()
            self.plan.stage(self, k + 1, src, dst, delta_t, s)
        else:
            self.plan.stage(self, k + 1, src, dst, delta_t, s, 0, ni)
            if halo is not None:
                halo.finish()
            self.plan.stage(self, k + 1, src, dst, delta_t, s, ni, nt - ni)

    def use_native_stepper(self, native_halo=None):
        """Drive iterate() through the C++ stepper (one C call per step, RCCL called natively)."""
        from . import native
        assert self.mode == "fused"
        self.stepper = native.NativeStepper(self.plan, native_halo)
        return self.stepper

    def iterate_steps(self, n_steps, delta_t, stream=None, halo=None):
        """n_steps steps with a fixed delta_t. With the native stepper this is ONE call: the exchange stream and
        the compute stream then meet only at its entry and exit (see csrc/hip/stepper.hip)."""
        if n_steps <= 0:
            return
        if getattr(self, "stepper", None) is None:
            for _ in range(n_steps):
                self.iterate(delta_t, stream, halo)
            return
        self.begin_step()
        first_prev, first_next = self.prev, self.next
        for _ in range(n_steps - 1):
            self.begin_step()
        self.stepper.iterate_steps(self, delta_t, n_steps, first_prev, first_next, stream)

    def iterate(self, delta_t, stream=None, halo=None):
        """One SSP-RK3 step (CompressibleEulerSolver::iterate). `halo` (a halo.HaloExchange) refreshes
        the ghost slots of each stage's source state while the interior tiles are already running."""
        self.begin_step()
        if getattr(self, "stepper", None) is not None:
            self.stepper.iterate(self, delta_t, stream)
            return
        for k in range(3):
            self.run_stage(k, delta_t, stream, halo)


class SubgridSolver:
    """Subgrid<4,4> / Subgrid<4,4,4>: planes[25, (N+G)*S] in subcells + per-block volumes."""

    def __init__(self, part, dtype=torch.float32, flux_kind=hip.KEPES, mode="compat", state=None):
        if not torch.cuda.is_available():
            raise hip.T8gpuHipError("SubgridSolver needs a GPU: the hot path has no CPU implementation")
        hip.lib()
        assert part.subgrid
        self.part, self.dtype, self.kind, self.mode = part, dtype, flux_kind, mode
        self.rank = part.mesh.dim
        self.S = 4 ** self.rank
        tot = part.N + part.G
        self.N, self.G, self.F, self.B = part.N, part.G, part.F, part.B
        self.stride = tot * self.S
        self.planes = torch.zeros((25, self.stride), dtype=dtype, device="cuda")
        ic = part.kh_initial_state() if state is None else state
        self.planes[0:5] = _dev(ic, dtype)
        self.volumes = _dev(part.volumes, dtype)
        self.fn = _dev(part.face_neighbors)
        self.level_diff = _dev(part.level_diff)
        self.nb_offset = _dev(part.nb_offset)
        self.normals = _dev(part.normals, dtype)
        self.areas = _dev(part.areas, dtype)
        self.next, self.prev = STEP0, STEP3
        self.plan = None
        if mode == "fused":
            from . import fused
            self.plan = fused.SubgridPlan(part, dtype)
        elif mode != "compat":
            raise ValueError(mode)

    def get_own_variables(self, step):
        return hip.vars_of(self.planes, step)

    def state(self, step=None):
        s = self.next if step is None else step
        return self.planes[5 * s:5 * s + 5, :self.N * self.S]

    def _stage_compat(self, stage, src, dst, dt, stream):
        st, fl = self.get_own_variables(src), self.get_own_variables(FLUXES)
        ev = _timer_begin(self)
        hip.call("t8gpu_hip_subgrid_inner", self.dtype, self.kind, self.rank, self.N, st, fl, hip.ptr(self.volumes),
                 stream)
        _timer_end(self, ev)
        if self.B > 0:
            hip.call("t8gpu_hip_subgrid_boundary", self.dtype, self.kind, self.rank, self.F, self.B,
                     hip.ptr(self.fn), hip.ptr(self.normals), hip.ptr(self.areas), st, fl, stream)
        hip.call("t8gpu_hip_subgrid_outer", self.dtype, self.kind, self.rank, self.F, hip.ptr(self.fn), None,
                 hip.ptr(self.level_diff), hip.ptr(self.nb_offset), hip.ptr(self.normals), hip.ptr(self.areas),
                 st, fl, stream)
        hip.call("t8gpu_hip_subgrid_rk3_stage", self.dtype, stage, self.rank, self.N,
                 self.get_own_variables(self.prev), st, self.get_own_variables(dst), fl, hip.ptr(self.volumes),
                 hip.fscalar(self.dtype, dt), stream)

    def begin_step(self):
        self.prev, self.next = self.next, self.prev  # solver.inl:154

    def stage_steps(self, k):
        return (self.prev, STEP1, STEP2)[k], (STEP1, STEP2, self.next)[k]

    def step_planes(self, step):
        return self.planes[5 * step:5 * step + 5]

    def run_stage(self, k, delta_t, stream=None, halo=None, split=False):
        """One flux evaluation + RK stage; with a halo exchange (or split=True) the fused kernel runs the
        blocks that touch no ghost block first and the others after the ghosts have arrived."""
        s = hip.stream_ptr(stream)
        src, dst = self.stage_steps(k)
        ni, nt = (self.plan.host.n_interior, self.N) if self.mode == "fused" else (0, 0)
        if halo is not None and halo.overlapped and 0 < ni < nt:
            # boundary pipeline on the comm stream: ghost blocks, then the blocks that read them; the rest beside it
            halo.start(self.step_planes(src), then=lambda: self.plan.stage(self, k + 1, src, dst, delta_t, hip.stream_ptr(), ni, nt - ni))
            self.plan.stage(self, k + 1, src, dst, delta_t, s, 0, ni)
            halo.finish()
            return
        if halo is not None:
            halo.start(self.step_planes(src))
        if self.mode == "compat":
            if halo is not None:
                halo.finish()
            self._stage_compat(k + 1, src, dst, delta_t, s)
            return
        ni, nt = self.plan.host.n_interior, self.N
        if (halo is None and not split) or ni == nt or ni == 0:
            if halo is not None:
                halo.finish()
            self.plan.stage(self, k + 1, src, dst, delta_t, s)
        else:
            self.plan.stage(self, k + 1, src, dst, delta_t, s, 0, ni)
            if halo is not None:
                halo.finish()
            self.plan.stage(self, k + 1, src, dst, delta_t, s, ni, nt - ni)

    def use_native_stepper(self, native_halo=None):
        """Drive iterate() through the C++ stepper (one C call per run of steps, RCCL called natively)."""
        from . import native
        assert self.mode == "fused"
        self.stepper = native.NativeSubgridStepper(self.plan, native_halo)
        return self.stepper

    def iterate_steps(self, n_steps, delta_t, stream=None, halo=None):
        """n_steps steps with a fixed delta_t; ONE call with the native stepper (csrc/hip/stepper.hip)."""
        if n_steps <= 0:
            return
        if getattr(self, "stepper", None) is None:
            for _ in range(n_steps):
                self.iterate(delta_t, stream, halo)
            return
        self.begin_step()
        first_prev, first_next = self.prev, self.next
        for _ in range(n_steps - 1):
            self.begin_step()
        self.stepper.iterate_steps(self, delta_t, n_steps, first_prev, first_next, stream)

    def iterate(self, delta_t, stream=None, halo=None):
        """SubgridCompressibleEulerSolver::iterate; `halo` refreshes the ghost blocks of each stage's source."""
        self.begin_step()
        if getattr(self, "stepper", None) is not None:
            self.stepper.iterate(self, delta_t, stream)
            return
        for k in range(3):
            self.run_stage(k, delta_t, stream, halo)
