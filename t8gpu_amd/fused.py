"""Device-side plan objects of the fused tile kernels (uploads the host plan, fills the C struct)."""
import ctypes as C

import numpy as np
import torch

from . import hip, hostmem
from .plan import HostPlainPlan


class T8gpuPlainPlan(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in HostPlainPlan.FIELDS] + [
        ("ntiles", C.c_int32), ("n_interior_tiles", C.c_int32), ("max_elems", C.c_int32), ("max_halo", C.c_int32),
        ("max_faces", C.c_int32), ("ell_width", C.c_int32), ("ell", C.c_void_p), ("geo_idx", C.c_void_p),
        ("geo_table", C.c_void_p), ("n_geo", C.c_int32), ("max_slots", C.c_int32), ("n_deep_tiles", C.c_int32),
        ("n_slots_addressed", C.c_int32), ("tile_desc", C.c_void_p), ("n_patch_tiles", C.c_int32 * 3), ("patch_dim", C.c_int32),
                ("n_irregular_tiles", C.c_int32 * 3),
        # ABI 7: the ghost window, attached by the multi-rank step driver only (stepper.hip); NULL / 0 here
        ("ghost_buf", C.c_void_p), ("send_map", C.c_void_p), ("send_list", C.c_void_p), ("send_buf", C.c_void_p),
        ("n_owned", C.c_int32), ("reserved7", C.c_int32)]


class PlainPlan:
    def __init__(self, part, dtype, tmax=None, fcap=None, compressed=True, dictionary=True, patches=None, flux_kind=None,
                 irregular=None, fcap_elements=None, two_classes=None):
        """compressed=False: generic kernel (CSR lists, full geometry). dictionary=False: pipelined kernel
        with per-face geometry rows even where the mesh has few distinct ones (what curved meshes get anyway).
        patches: cut structured 16 x 16 patches out of the tiling for the patch kernel (default: yes for the compressed
        plan; T8GPU_PATCH=0 switches it off -- same results bit for bit, every element through the tile kernels)."""
        skip_geo = self._plan_on_host(part, dtype, tmax, fcap, compressed, dictionary, patches, flux_kind, irregular, fcap_elements, two_classes)
        self._upload(dtype, compressed, dictionary, skip_geo, part)

    @classmethod
    def on_host(cls, part, dtype, **kw):
        """The host half alone (no GPU): the tile plan and the caps / patch forms the rules below settle on (`host`, `auto_fcap`,
        `irregular`, `auto_irregular`). Tests pin the rules through this."""
        self = object.__new__(cls)
        args = dict(tmax=None, fcap=None, compressed=True, dictionary=True, patches=None, flux_kind=None, irregular=None, fcap_elements=None,
                    two_classes=None)
        args.update(kw)
        self._plan_on_host(part, dtype, **args)
        return self

    def _plan_on_host(self, part, dtype, tmax, fcap, compressed, dictionary, patches, flux_kind, irregular, fcap_elements=None,
                      two_classes=None):
        import os
        # An inherited face cap (amr._inherited_plan_options) was chosen for a mesh of `fcap_elements` elements: which kernel a
        # plan gets depends on its tile count, so a mesh that has since grown or shrunk by more than a factor two decides again
        # (a run that starts from a coarse mesh would otherwise keep the small mesh's 768-face one-tile plan for good).
        if fcap_elements and not (fcap_elements // 2 <= part.N <= 2 * fcap_elements):
            fcap = None
            irregular = None          # (an inherited patch form was decided on that mesh too: ADVICE r3)
        self.auto_fcap_elements = fcap_elements if (fcap is not None or irregular is not None) and fcap_elements else part.N
        if patches is None:
            patches = compressed and os.environ.get("T8GPU_PATCH", "1") != "0"
        self.patches = patches if compressed else False          # True / False, or 2 / 3 for one kind only
        # 3D: blocks next to periodic wraps / walls / coarser - side neighbours become (irregular) patches too
        irregular_auto = irregular is None and "T8GPU_PATCH_IRREGULAR" not in os.environ   # (nobody asked: the rules below may drop it)
        if irregular is None:
            irregular = {"0": False, "all": "all"}.get(os.environ.get("T8GPU_PATCH_IRREGULAR", "1"), True)
        self.irregular = irregular
        # the patch kernels address a plane by a 32-bit byte offset: meshes whose planes reach 4 GiB keep the tile kernels
        if (part.N + part.G) * (4 if dtype == torch.float32 else 8) >= 2 ** 32:
            self.patches = False
        # tuning knobs of the tiling. A mesh that would give fewer than 512 tiles (two per CU) gets half-size tiles:
        # c1 (65 536 elements) runs 16 % faster on 512 tiles of 128 than on 256 tiles of 256.
        given_fcap = fcap
        small = part.N < 512 * 256 and tmax is None and fcap is None and "T8GPU_TMAX" not in os.environ
        tmax = int(os.environ.get("T8GPU_TMAX", 128 if small else 256)) if tmax is None else tmax
        if small:
            fcap = int(os.environ.get("T8GPU_FCAP", 256))
        # 512 faces = two passes of 256: what the persistent kernel takes, and the default. (Round 1 measured larger tiles
        # 1-8 % slower on 3D meshes because the one-tile kernel then lost its fourth workgroup per CU; since round 2 that
        # kernel holds three either way, and 3D AMR meshes get 768 -- below.) A 512-lane workgroup with one lane per own +
        # halo element and two passes of 512 faces was 5-7 % slower (8 waves per barrier, 2 workgroups per CU).
        # the face cap chosen by the heuristics below, or handed down from the plan of the mesh this one was adapted from
        # (amr._inherited_plan_options); None: the default
        self.auto_fcap = given_fcap
        # Partitioned meshes: interior tiles in ONE class (tile_order = interior | ghost-reading) -- the two-lane step driver
        # (stepper.hip) launches [0, n_interior) as one kernel per stage. two_classes=False / T8GPU_PLAN_CLASSES=3 keeps the
        # deep / near-boundary split (what the three-stream pipeline of rounds 1-3 wants: T8GPU_STEPPER=legacy).
        if two_classes is None:
            two_classes = getattr(part, "nranks", 1) > 1 and os.environ.get("T8GPU_PLAN_CLASSES", "2") != "3"
        self.two_classes = bool(two_classes)
        retry_768 = given_fcap in (384, 480) and dtype == torch.float64      # an inherited cap still has to fit the persistent kernel
        if (fcap is None and "T8GPU_FCAP" not in os.environ and not small and dtype == torch.float64 and
                getattr(part.mesh, "dim", 2) == 3):
            # fp64 on 3D meshes. Curved meshes (no small geometry dictionary) run the one-tile kernel, which holds three
            # workgroups per CU whatever the LDS (146 VGPRs) and does better on tiles of up to 768 faces in three passes
            # (~200 instead of 134 elements per 256-lane workgroup): c5p 4 230 -> 4 545, c5t 4 380 -> 4 540 M cell-updates/s.
            # Cartesian 3D AMR (elements with more than 8 faces) runs the persistent kernel IF its third workgroup per CU
            # fits: 480-face tiles (51.7 KB of LDS) do, 512-face tiles (53.1 KB) do not -- c5 4 110 (one-tile, 512) ->
            # 4 450 (one-tile, 768) -> 4 770 (persistent, 480). 2D meshes keep 512 (their 35 KB tiles are what the DENSE
            # budget needs; 768: -13 %), and so does fp32 (four to five workgroups per CU on 512-face tiles either way).
            # (A partitioned run launches tile classes, i.e. the one-tile kernel: 768 there too.)
            if self._many_geometries(part):
                fcap = self.auto_fcap = 768
            elif self._wide_rows(part):
                if getattr(part, "nranks", 1) > 1:
                    fcap = self.auto_fcap = 768
                else:
                    fcap, retry_768 = 480, True
                    self.auto_fcap = 480
        fcap = int(os.environ.get("T8GPU_FCAP", 512)) if fcap is None else fcap
        # the per-face geometry rows are only read by the kernels without a dictionary (generic kernel, dictionary=False)
        self.host = HostPlainPlan.from_partition(part, two_classes=self.two_classes, tmax=tmax, fcap=fcap, want_face_geo=not (compressed and dictionary), patches=self.patches, irregular=self.irregular)
        if retry_768:
            # 480-face tiles only pay if the persistent kernel takes the plan: the LAUNCHER'S OWN test is asked (C-ABI query;
            # it covers the LDS margin, the tile-count gate for mid-size meshes, the flux kind and T8GPU_PERSISTENT=0 --
            # a partial copy of that test used to live here). Otherwise the one-tile kernel runs, which does
            # better on 768-face tiles.
            if not (compressed and dictionary and self._persistent_accepts(self.host, dtype, flux_kind)):
                fcap = self.auto_fcap = 768
                self.host = HostPlainPlan.from_partition(part, two_classes=self.two_classes, tmax=tmax, fcap=fcap, want_face_geo=not (compressed and dictionary), patches=self.patches, irregular=self.irregular)
        if (given_fcap is None and "T8GPU_FCAP" not in os.environ and compressed and dictionary and fcap in (480, 512)
                and getattr(part.mesh, "dim", 2) == 3 and self.host.n_patches * 256 > part.N // 2
                and self.host.n_patches < self.host.ntiles):
            # Most of a 3D mesh in patches: the generic tiles are what is left BETWEEN patches -- short stretches (c5: 144
            # elements / 520 faces each, a 2-cell slab of fine cells beside coarse ones) that a 480-face cap cuts 120 + 24 and a
            # 512-face cap 140 + 4. Measured on c5, fp64 (scripts/fcap_scan.sh): 256: 5 340, 300: 5 640, 360-400: 5 720-5 760,
            # 440: 5 540, 480: 5 420 M/s; fp32: 512: 9 310, 384: 9 950 (a plan without patches prefers 480: 4 550 against
            # 4 130 at 380).
            trial = HostPlainPlan.from_partition(part, two_classes=self.two_classes, tmax=tmax, fcap=384, want_face_geo=False, patches=self.patches, irregular=self.irregular)
            # (fp64: only if the persistent kernel still takes the plan -- the one-tile kernel wants 768; fp32 runs either kernel well)
            if dtype == torch.float32 or self._persistent_accepts(trial, dtype, flux_kind):
                fcap = self.auto_fcap = 384
                self.host = trial
        # fp32: the irregular patch form pays only where the alternative is a launch of generic tiles too small for the
        # persistent kernel (the uniform box c5u: 10 480 -> 12 680 M/s); where the blocks would simply join a large generic
        # launch it costs (c5: 10 630 -> 9 960) -- in fp32 its selects and signs weigh more against the flux than in fp64
        # (c5 fp64: 5 560 -> 5 760). Asked of the launcher with the tile count the plan would have without the form.
        n_irr = sum(self.host.n_irregular_class)
        if (dtype == torch.float32 and self.irregular is True and irregular_auto and n_irr and compressed and dictionary
                and self._persistent_accepts(self.host, dtype, flux_kind, n_generic=self.host.ntiles - self.host.n_patches + 2 * n_irr)):
            self.irregular = False
            self.host = HostPlainPlan.from_partition(part, two_classes=self.two_classes, tmax=tmax, fcap=fcap, want_face_geo=False, patches=self.patches, irregular=False)
        h = self.host
        # What an ADAPTED mesh's plan inherits (amr._inherited_plan_options): where nearly every patch is an irregular one -- thin
        # refined sheets: c5a has 5 944 irregular and 8 regular patches -- the irregular form buys nothing (it runs at the speed of
        # the persistent tile kernel on such cells) and costs planning time in every cycle; keep it where regular patches carry
        # a good part of the mesh (uniform boxes with wraps or walls: c5u +26 %).
        n_irr = sum(h.n_irregular_class)
        self.auto_irregular = self.irregular if n_irr <= 4 * (h.n_patches - n_irr) else False
        # the pipelined kernel with a geometry dictionary never reads the per-face rows (32 B per face: 700 MB at c4)
        skip_geo = (compressed and dictionary and h.geo_table.shape[0] > 0 and h.max_elems <= 256 and h.max_slots <= 512
                    and h.max_faces <= 1024)
        if not skip_geo and h.face_geo.shape[0] == 0 and h.face_lr.size:     # the kernels this plan gets do read the rows
            self.host = h = HostPlainPlan.from_partition(part, two_classes=self.two_classes, tmax=tmax, fcap=fcap, want_face_geo=True, patches=self.patches, irregular=self.irregular)
        return skip_geo

    def _upload(self, dtype, compressed, dictionary, skip_geo, part):
        """device copies of the host plan's arrays and the T8gpuPlainPlan that points at them"""
        self.dtype = dtype
        self._keep = {}
        c = T8gpuPlainPlan()
        # the per-element CSR lists are what the generic kernel walks; a compressed plan inside the pipelined kernels' limits
        # (the launcher's own condition, kernels_fused.hip: plain_generic_stage) never runs it: 2 + 4 bytes per incidence
        # less to upload (50 MB at 3 M elements in 3D)
        h = self.host
        skip_csr = compressed and not self._needs_csr(h)
        for name in HostPlainPlan.FIELDS:
            a = getattr(self.host, name)
            if skip_csr and name in ("csr_off", "csr_ent"):
                continue
            if name == "face_geo":
                if skip_geo:
                    continue
                a = a.astype(np.float32 if dtype == torch.float32 else np.float64)
            if a.dtype == np.uint32:
                a = a.view(np.int32)
            if a.dtype == np.uint16:
                a = a.view(np.int16)
            t = hostmem.to_device(a)
            self._keep[name] = t
            setattr(c, name, t.data_ptr())
        npf = np.float32 if dtype == torch.float32 else np.float64
        if compressed:
            extra = {"ell": self.host.ell.view(np.int16), "tile_desc": self.host.tile_desc}
            c.ell_width = self.host.ell_width
            if dictionary and self.host.geo_table.shape[0] > 0:
                extra["geo_idx"] = self.host.geo_idx.view(np.int16)
                extra["geo_table"] = self.host.geo_table.astype(npf)
                c.n_geo = self.host.geo_table.shape[0]
            for name, a in extra.items():
                t = hostmem.to_device(a)
                self._keep[name] = t
                setattr(c, name, t.data_ptr())
            if c.n_geo > 0:   # the dictionary's tangent rows by the device's own routine (t8gpu_hip.h: same bits as per-face geometry)
                hip.call("t8gpu_hip_plain_geo_frames", dtype, hip.ptr(self._keep["geo_table"]), int(c.n_geo), hip.stream_ptr())
        c.ntiles, c.n_interior_tiles = self.host.ntiles, self.host.n_interior
        c.max_elems, c.max_halo, c.max_faces = self.host.max_elems, self.host.max_halo, self.host.max_faces
        c.max_slots, c.n_deep_tiles = self.host.max_slots, self.host.n_deep
        for k in range(3):
            c.n_patch_tiles[k] = self.host.n_patch_class[k]
            c.n_irregular_tiles[k] = self.host.n_irregular_class[k]
        c.patch_dim = self.host.patch_dim
        c.n_slots_addressed = part.N + part.G
        self.c = c

    @staticmethod
    def _needs_csr(h):
        """Would the generic tiles of the compressed form of host plan `h` run the generic kernel, which walks the CSR lists?
        (t8gpu_hip_plain_needs_csr: the launcher's own test.)"""
        c = T8gpuPlainPlan()
        one = C.c_void_p(1)                                     # "present": never dereferenced by the query
        c.tile_desc, c.ell = one, one
        c.ell_width, c.max_elems, c.max_halo, c.max_faces, c.max_slots = h.ell_width, h.max_elems, h.max_halo, h.max_faces, h.max_slots
        fn = hip.lib().t8gpu_hip_plain_needs_csr
        fn.restype = C.c_int
        return bool(fn(C.byref(c)))

    @staticmethod
    def _persistent_accepts(h, dtype, flux_kind=None, n_generic=None):
        """Would a whole-plan launch of host plan `h` run the persistent tile kernel? (t8gpu_hip_plain_persistent_accepts:
        the launcher's decision; only integer fields and the presence of the compressed arrays matter, so it can be asked
        before anything is uploaded -- and without a GPU.)"""
        c = T8gpuPlainPlan()
        one = C.c_void_p(1)                                     # "present": never dereferenced by the query
        c.tile_desc, c.ell = one, one
        if h.geo_table.shape[0] > 0:
            c.geo_idx, c.geo_table = one, one
        c.n_geo, c.ell_width = h.geo_table.shape[0], h.ell_width
        c.ntiles, c.max_elems, c.max_halo, c.max_faces, c.max_slots = h.ntiles, h.max_elems, h.max_halo, h.max_faces, h.max_slots
        fn = hip.lib().t8gpu_hip_plain_persistent_accepts
        fn.restype = C.c_int
        n_generic = h.ntiles - h.n_patches if n_generic is None else n_generic
        return bool(fn(C.byref(c), int(hip.KEPES if flux_kind is None else flux_kind), 4 if dtype == torch.float32 else 8, int(n_generic)))

    @staticmethod
    def _wide_rows(part):
        """True if some owned element has more than 8 faces (ELL rows of 16 or 24 entries: tile_plan.cpp)."""
        fn = np.asarray(part.face_neighbors).reshape(-1)
        F, N = part.F, part.N
        l, r = fn[0:2 * F:2], fn[1:2 * F:2]
        deg = np.bincount(l[l < N], minlength=N) + np.bincount(r[(r < N) & (r != l)], minlength=N)
        if part.B:
            lb = fn[2 * F:2 * F + part.B]
            deg = deg + np.bincount(lb[lb < N], minlength=N)
        return bool(deg.size) and int(deg.max()) > 8

    @staticmethod
    def _many_geometries(part, sample=20000, limit=128):
        """True if already a sample of the faces shows more distinct {normal, area} rows than the persistent kernel's
        LDS dictionary holds (curved meshes: every face its own)."""
        nd = part.normal_dim
        nrm = np.asarray(part.normals, np.float64).reshape(-1, nd)[:sample]
        rows = np.concatenate([nrm, np.asarray(part.areas, np.float64).reshape(-1, 1)[:sample]], axis=1)
        return np.unique(rows, axis=0).shape[0] > limit

    def stage(self, solver, stage, src, dst, dt, stream, tile_begin=0, tile_count=None):
        from .solver import _timer_begin, _timer_end
        n = self.host.ntiles - tile_begin if tile_count is None else tile_count
        ev = _timer_begin(solver)
        # speed estimates: rewritten by every stage in the reference, read only between steps -> written by stage 3 only
        hip.call("t8gpu_hip_plain_fused_stage", self.dtype, solver.kind, stage, C.byref(self.c), tile_begin, n,
                 solver.get_own_variables(solver.prev), solver.get_own_variables(src), solver.get_own_variables(dst),
                 hip.ptr(solver.planes[25]), hip.fscalar(self.dtype, dt), hip.ptr(solver.speed) if stage == 3 else None, stream)
        _timer_end(solver, ev)


class T8gpuSubgridPlan(C.Structure):
    _fields_ = [("block_rec", C.c_void_p), ("bf_rec", C.c_void_p),
                ("num_elements", C.c_int32), ("rank", C.c_int32), ("max_faces_per_block", C.c_int32),
                ("n_interior_blocks", C.c_int32), ("n_deep_blocks", C.c_int32), ("n_blocks_addressed", C.c_int32),
                ("fam_rec", C.c_void_p), ("rest_rec", C.c_void_p), ("n_families", C.c_int32), ("n_rest", C.c_int32)]


class SubgridPlan:
    """Device copy of the joined per-block face records for the fused Subgrid<4,4> / Subgrid<4,4,4> kernels."""

    def __init__(self, part, dtype):
        from .plan import HostSubgridPlan
        self.host = HostSubgridPlan(part)
        self.dtype = dtype
        block_rec, bf_rec = self.host.records(part.areas, 4 if dtype == torch.float32 else 8)
        self._keep = {"block_rec": torch.from_numpy(block_rec).cuda(), "bf_rec": torch.from_numpy(bf_rec).cuda()}
        c = T8gpuSubgridPlan()
        if self.host.n_families > 0:
            fam_rec, rest_rec = self.host.family_records(part.areas, 4 if dtype == torch.float32 else 8)
            self._keep["fam_rec"], self._keep["rest_rec"] = torch.from_numpy(fam_rec).cuda(), torch.from_numpy(rest_rec).cuda()
            c.n_families, c.n_rest = self.host.n_families, self.host.n_rest
        for k, t in self._keep.items():
            setattr(c, k, t.data_ptr())
        c.num_elements, c.rank, c.max_faces_per_block = part.N, part.mesh.dim, self.host.max_bf
        c.n_interior_blocks = self.host.n_interior
        c.n_deep_blocks = self.host.n_deep
        c.n_blocks_addressed = self.host.n_addressed
        self.c = c

    def stage(self, solver, stage, src, dst, dt, stream, block_begin=0, block_count=None):
        from .solver import _timer_begin, _timer_end
        n = self.host.N - block_begin if block_count is None else block_count
        ev = _timer_begin(solver)
        hip.call("t8gpu_hip_subgrid_fused_stage", self.dtype, solver.kind, stage, C.byref(self.c), block_begin, n,
                 solver.get_own_variables(solver.prev), solver.get_own_variables(src), solver.get_own_variables(dst),
                 hip.ptr(solver.volumes), hip.fscalar(self.dtype, dt), stream)
        _timer_end(solver, ev)
