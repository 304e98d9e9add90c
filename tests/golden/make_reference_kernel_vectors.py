#!/usr/bin/env python3
"""Generates tests/golden/reference_kernel_vectors.npz: outputs of the REFERENCE's own KERNEL BODIES on tiny meshes.

Build container only (needs /root/reference; the GPU box has neither the tree nor any use for this script -- tests read the
committed .npz). Where make_reference_vectors.py executes the reference's flux FUNCTIONS, this script executes its
`__global__` kernels -- the index logic around those functions: face -> element gathers and scatter-adds, the reflective
wall, the 2:1 hanging-face map of the Subgrid outer kernel, the inner-face pattern of a block, the RK stage kernels -- so
that the oracle's reading of them is checked against an EXECUTION of the reference's text and not
only against a reading of it.

What is taken from the reference, into a TEMPORARY directory only (nothing of it is written into this repository; the .npz
holds input and output arrays):
  * t8gpu/utils/meta.h                                   whole (needs <type_traits> only)
  * t8gpu/memory/memory_manager.h:18-326                 variable_traits, step_traits, MemoryAccessorOwn, MemoryAccessorAll
  * t8gpu/memory/subgrid_memory_manager.h:22-424         Subgrid<>, its Accessor, SubgridMemoryAccessorOwn / All
  * t8gpu/mesh/mesh_manager.h:19-182                     MeshConnectivityAccessor
  * t8gpu/mesh/subgrid_mesh_manager.h:19-216             SubgridMeshConnectivityAccessor
  * t8gpu/timestepping/ssp_runge_kutta.h + .inl          declarations and the six stage kernels
  * examples/compressible_euler/kernels.h (declarations) and kernels.cu:1-469 (kepes_compute_fluxes,
    reflective_boundary_condition); examples/subgrid/kernels.inl:1-1107 (compute_inner / outer / boundary_fluxes)
  (`#include` lines dropped; fp64: the `= float;` of variable_traits and of step_traits replaced by `= double;`, the edit
  make_reference_vectors.py makes).
What stands in for CUDA and for the classes that own the accessors (this is why the result does NOT count as a reference
build: parity stays "unpinned"):
  * __global__ / __device__ / __host__ empty, `__shared__` = static, threadIdx (thread_local) / blockIdx / blockDim / gridDim
    plain structs, atomicAdd = read-modify-write, t8_locidx_t = int32_t;
  * the accessors' constructors are private and friend MemoryManager / MeshManager / SubgridMemoryManager /
    SubgridMeshManager: the driver defines classes of those names whose only members build an accessor from host pointers;
  * the example solvers are two-line structs giving `dim` / `float_type`;
  * a kernel launch is a loop over blocks; the threads of a block run ONE AFTER ANOTHER IN DESCENDING THREAD ORDER
    between two __syncthreads() (real threads handed a baton). The order matters in exactly one place: compute_inner_fluxes
    reads its lower neighbour's slot of `shared_fluxes` and rewrites its own slot for the next direction without a barrier
    in between (kernels.inl:413-419 / 454-458); on lock-step hardware every thread reads before any thread
    writes, and running the threads in descending order gives exactly that (a thread's lower neighbours run after it).
Cases (synthetic provider, reference array formats; uniform forests with one or two elements refined, 15 - 80 elements, hanging
faces on every side of the refined ones): a walled 2D quad mesh, a periodic and a walled 3D hex mesh for the plain kernels; a
periodic and a walled 3D Subgrid<4,4,4> forest, a periodic and a walled 2D Subgrid<4,4> forest for the Subgrid kernels; seeded
perturbed states; fp32 and fp64. Per case: flux planes after each flux kernel of a stage (starting
from zero), the speed estimates, and the three RK stage outputs on given inputs.
"""
import os
import subprocess
import sys
import tempfile

import numpy as np

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

COMMON = r"""
#include <algorithm>
#include <array>
#include <cmath>
#include <condition_variable>
#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <tuple>
#include <type_traits>
#include <utility>
#include <vector>
#define __device__
#define __host__
#define __global__
#define __shared__ static
using std::abs; using std::cbrt; using std::log; using std::max; using std::min; using std::sqrt;
typedef int32_t t8_locidx_t;
struct dim3 {   // (brace-initialised from the Subgrid extents: subgrid_memory_manager.h, block_size)
  unsigned x, y, z;
  constexpr dim3(unsigned a = 1, unsigned b = 1, unsigned c = 1) : x(a), y(b), z(c) {}
};
typedef dim3 Dim3;
static thread_local Dim3 threadIdx;
static Dim3 blockIdx, blockDim, gridDim;
template<class T> static T atomicAdd(T* p, T v) { T o = *p; *p = o + v; return o; }

// ---- a block's threads, one after another in descending thread order between barriers (see the docstring) ----------
static std::mutex sch_m;
static std::condition_variable sch_cv;
static int sch_turn = 0, sch_T = 1, sch_phase = 0;
static thread_local int my_id = 0, my_phase = 0;
static void sch_acquire() {
  std::unique_lock<std::mutex> lk(sch_m);
  sch_cv.wait(lk, [] { return sch_phase == my_phase && sch_turn == my_id; });
}
static void sch_pass() {
  std::unique_lock<std::mutex> lk(sch_m);
  if (my_id == 0) { sch_phase++; sch_turn = sch_T - 1; } else { sch_turn = my_id - 1; }
  my_phase++;
  sch_cv.notify_all();
}
static void __syncthreads() { sch_pass(); sch_acquire(); }
// launch<<<grid, block>>>(f): f is called once per thread with threadIdx / blockIdx set
static void launch(unsigned grid, Dim3 block, const std::function<void()>& f, bool threads) {
  gridDim = Dim3{grid, 1, 1};
  blockDim = block;
  const int T = static_cast<int>(block.x * block.y * block.z);
  for (unsigned b = 0; b < grid; b++) {
    blockIdx = Dim3{b, 0, 0};
    if (!threads) {   // kernels without __syncthreads: plain loops
      for (int t = 0; t < T; t++) {
        threadIdx.x = t % block.x; threadIdx.y = (t / block.x) % block.y; threadIdx.z = t / (block.x * block.y);
        f();
      }
      continue;
    }
    sch_T = T; sch_phase = 0; sch_turn = T - 1;
    std::vector<std::thread> th;
    for (int t = 0; t < T; t++)
      th.emplace_back([&, t] {
        threadIdx.x = t % block.x; threadIdx.y = (t / block.x) % block.y; threadIdx.z = t / (block.x * block.y);
        my_id = t; my_phase = 0;
        sch_acquire();
        f();
        sch_pass();
      });
    for (auto& x : th) x.join();
  }
}
template<class T> static std::vector<T> rd(const std::string& path) {
  FILE* f = std::fopen(path.c_str(), "rb"); if (!f) { std::fprintf(stderr, "missing %s\n", path.c_str()); std::exit(2); }
  std::fseek(f, 0, SEEK_END); long n = std::ftell(f); std::fseek(f, 0, SEEK_SET);
  std::vector<T> v(n / sizeof(T)); if (n && std::fread(v.data(), sizeof(T), v.size(), f) != v.size()) std::exit(3);
  std::fclose(f); return v;
}
template<class T> static void wr(const std::string& path, const std::vector<T>& v) {
  FILE* f = std::fopen(path.c_str(), "wb"); std::fwrite(v.data(), sizeof(T), v.size(), f); std::fclose(f);
}
"""

PLAIN = COMMON + r"""
#include "ref_meta.inl"
namespace t8gpu {
#include "ref_mem.inl"
#include "ref_submem.inl"   // (ssp_runge_kutta.h declares the Subgrid stage kernels too)
  // the friends that may construct the accessors (memory_manager.h:90-95, mesh_manager.h:31-32): builders only
  template<typename VT, typename ST> class MemoryManager {
   public:
    using ft = typename variable_traits<VT>::float_type;
    static MemoryAccessorOwn<VT> own(std::array<ft*, 5> a) { return MemoryAccessorOwn<VT>(a); }
    static MemoryAccessorAll<VT> all(std::array<ft* const*, 5> a) { return MemoryAccessorAll<VT>(a); }
  };
#include "ref_conn.inl"
  template<typename VT, typename ST, size_t dim_> class MeshManager {
   public:
    using ft = typename variable_traits<VT>::float_type;
    static MeshConnectivityAccessor<ft, dim_> conn(int const* ranks, t8_locidx_t const* idx, t8_locidx_t const* fn, ft const* nrm,
                                                   ft const* ar, t8_locidx_t F, t8_locidx_t B) {
      return MeshConnectivityAccessor<ft, dim_>(ranks, idx, fn, nrm, ar, F, B);
    }
  };
  enum VariableList { Rho, Rho_v1, Rho_v2, Rho_v3, Rho_e, nb_variables };   // examples/compressible_euler/solver.h:14-21
  struct CompressibleEulerSolver { static constexpr size_t dim = 3; };      // solver.h:36
#include "ref_kernels_h.inl"
}
#include "ref_rk_h.inl"
#include "ref_rk.inl"
#include "ref_kernels_plain.inl"
using namespace t8gpu;
using FT = variable_traits<VariableList>::float_type;
int main(int, char** argv) {
  const std::string d = argv[1];
  auto sz = rd<int32_t>(d + "/sizes.bin");   // N, G, F, B, stride
  const int N = sz[0], F = sz[2], B = sz[3]; const size_t stride = sz[4];
  auto fn = rd<int32_t>(d + "/fn.bin"), idx = rd<int32_t>(d + "/idx.bin");
  auto nrm = rd<FT>(d + "/normals.bin"), ar = rd<FT>(d + "/areas.bin"), st = rd<FT>(d + "/state.bin"), vol = rd<FT>(d + "/volume.bin");
  std::vector<int> ranks(stride, 0);
  std::vector<FT> flux(5 * stride, FT(0)), speed(F + B, FT(0));
  std::array<FT*, 5> sp, fp; std::array<FT* const*, 5> spp, fpp;
  for (int k = 0; k < 5; k++) { sp[k] = st.data() + k * stride; fp[k] = flux.data() + k * stride; spp[k] = &sp[k]; fpp[k] = &fp[k]; }
  const auto conn = MeshManager<VariableList, int, 3>::conn(ranks.data(), idx.data(), fn.data(), nrm.data(), ar.data(), F, B);
  using MM = MemoryManager<VariableList, int>;
  // solver.cu:81-97: interior faces (all-rank accessors), then the reflective wall (own accessors)
  { const auto v = MM::all(spp), f = MM::all(fpp);
    launch((F + 255) / 256, Dim3{256, 1, 1}, [&] { kepes_compute_fluxes(conn, v, f, speed.data()); }, false); }
  wr(d + "/flux_interior.bin", flux);
  if (B > 0) { const auto v = MM::own(sp), f = MM::own(fp);
    launch((B + 255) / 256, Dim3{256, 1, 1}, [&] { reflective_boundary_condition(conn, v, f, speed.data()); }, false); }
  wr(d + "/flux_all.bin", flux);
  wr(d + "/speed.bin", speed);
  // the three RK stages on given inputs: prev = state, mid = rk_mid, fluxes = rk_flux (solver.cu:100-174)
  auto mid = rd<FT>(d + "/rk_mid.bin"), rkf = rd<FT>(d + "/rk_flux.bin");
  const FT dt = rd<FT>(d + "/dt.bin")[0];
  for (int stage = 1; stage <= 3; stage++) {
    std::vector<FT> out(5 * stride, FT(0)), fl = rkf;
    std::array<FT*, 5> mp, op, flp;
    for (int k = 0; k < 5; k++) { mp[k] = mid.data() + k * stride; op[k] = out.data() + k * stride; flp[k] = fl.data() + k * stride; }
    const auto P = MM::own(sp), M = MM::own(mp), O = MM::own(op), Fl = MM::own(flp);
    launch((N + 255) / 256, Dim3{256, 1, 1}, [&] {
      if (stage == 1) timestepping::SSP_3RK_step1<VariableList>(P, O, Fl, vol.data(), dt, N);
      else if (stage == 2) timestepping::SSP_3RK_step2<VariableList>(P, M, O, Fl, vol.data(), dt, N);
      else timestepping::SSP_3RK_step3<VariableList>(P, M, O, Fl, vol.data(), dt, N);
    }, false);
    wr(d + "/rk_out" + std::to_string(stage) + ".bin", out);
    wr(d + "/rk_flux_after" + std::to_string(stage) + ".bin", fl);
  }
  return 0;
}
"""

SUBGRID = COMMON + r"""
#include "ref_meta.inl"
namespace t8gpu {
#include "ref_mem.inl"
#include "ref_submem.inl"
  // the friends that may construct the accessors (subgrid_memory_manager.h:181-184, subgrid_mesh_manager.h:32-33)
  template<typename VT, typename ST, typename SG> class SubgridMemoryManager {
   public:
    using ft = typename variable_traits<VT>::float_type;
    static SubgridMemoryAccessorOwn<VT, SG> own(std::array<ft*, 5> a) { return SubgridMemoryAccessorOwn<VT, SG>(a); }
    static SubgridMemoryAccessorAll<VT, SG> all(std::array<ft* const*, 5> a) { return SubgridMemoryAccessorAll<VT, SG>(a); }
  };
#include "ref_subconn.inl"
  template<typename VT, typename ST, typename SG> class SubgridMeshManager {
   public:
    using ft = typename variable_traits<VT>::float_type;
    static SubgridMeshConnectivityAccessor<ft, SG> conn(int const* ranks, t8_locidx_t const* idx, t8_locidx_t const* fn,
                                                        t8_locidx_t const* ld, t8_locidx_t const* off, ft const* nrm, ft const* ar,
                                                        t8_locidx_t F, t8_locidx_t B) {
      return SubgridMeshConnectivityAccessor<ft, SG>(ranks, idx, fn, ld, off, nrm, ar, F, B);
    }
  };
}
enum VariableList { Rho, Rho_v1, Rho_v2, Rho_v3, Rho_e, nb_variables };       // examples/subgrid/solver.h:12-20
template<typename SG> struct SubgridCompressibleEulerSolver {                 // solver.h:35
  using float_type = typename t8gpu::variable_traits<VariableList>::float_type;
};
#include "ref_rk_h.inl"
#include "ref_rk.inl"
#include "ref_kernels_subgrid.inl"
using FT = t8gpu::variable_traits<VariableList>::float_type;
template<class SG> static int run(const std::string& d) {
  constexpr int S = SG::size, R = SG::rank;
  auto sz = rd<int32_t>(d + "/sizes.bin");   // N, G, F, B, stride (in subcells)
  const int N = sz[0], F = sz[2], B = sz[3]; const size_t stride = sz[4];
  auto fn = rd<int32_t>(d + "/fn.bin"), idx = rd<int32_t>(d + "/idx.bin"), ld = rd<int32_t>(d + "/level_diff.bin"), off = rd<int32_t>(d + "/nb_offset.bin");
  auto nrm = rd<FT>(d + "/normals.bin"), ar = rd<FT>(d + "/areas.bin"), st = rd<FT>(d + "/state.bin"), vol = rd<FT>(d + "/volume.bin");
  std::vector<int> ranks(stride / S, 0);
  std::vector<FT> flux(5 * stride, FT(0));
  std::array<FT*, 5> sp, fp; std::array<FT* const*, 5> spp, fpp;
  for (int k = 0; k < 5; k++) { sp[k] = st.data() + k * stride; fp[k] = flux.data() + k * stride; spp[k] = &sp[k]; fpp[k] = &fp[k]; }
  const auto conn = t8gpu::SubgridMeshManager<VariableList, int, SG>::conn(ranks.data(), idx.data(), fn.data(), ld.data(), off.data(), nrm.data(),
                                                                    ar.data(), F, B);
  using MM = t8gpu::SubgridMemoryManager<VariableList, int, SG>;
  const Dim3 cell = R == 3 ? Dim3{4, 4, 4} : Dim3{4, 4, 1}, face = R == 3 ? Dim3{4, 4, 1} : Dim3{4, 1, 1};
  // examples/subgrid/solver.inl:166-195: inner, boundary, outer
  { const auto v = MM::own(sp), f = MM::own(fp); launch(N, cell, [&] { compute_inner_fluxes<SG>(v, f, vol.data()); }, true); }
  wr(d + "/flux_inner.bin", flux);
  if (B > 0) { const auto v = MM::own(sp), f = MM::own(fp); launch(B, face, [&] { compute_boundary_fluxes<SG>(conn, v, f); }, false); }
  wr(d + "/flux_inner_boundary.bin", flux);
  { const auto v = MM::all(spp), f = MM::all(fpp); launch(F, face, [&] { compute_outer_fluxes<SG>(conn, v, f); }, false); }
  wr(d + "/flux_all.bin", flux);
  auto mid = rd<FT>(d + "/rk_mid.bin"), rkf = rd<FT>(d + "/rk_flux.bin");
  const FT dt = rd<FT>(d + "/dt.bin")[0];
  for (int stage = 1; stage <= 3; stage++) {
    std::vector<FT> out(5 * stride, FT(0)), fl = rkf;
    std::array<FT*, 5> mp, op, flp;
    for (int k = 0; k < 5; k++) { mp[k] = mid.data() + k * stride; op[k] = out.data() + k * stride; flp[k] = fl.data() + k * stride; }
    const auto P = MM::own(sp), M = MM::own(mp), O = MM::own(op), Fl = MM::own(flp);
    launch(N, cell, [&] {
      namespace ts = t8gpu::timestepping::subgrid;
      if (stage == 1) ts::SSP_3RK_step1<VariableList, SG>(P, O, Fl, vol.data(), dt);
      else if (stage == 2) ts::SSP_3RK_step2<VariableList, SG>(P, M, O, Fl, vol.data(), dt);
      else ts::SSP_3RK_step3<VariableList, SG>(P, M, O, Fl, vol.data(), dt);
    }, false);
    wr(d + "/rk_out" + std::to_string(stage) + ".bin", out);
    wr(d + "/rk_flux_after" + std::to_string(stage) + ".bin", fl);
  }
  return 0;
}
int main(int, char** argv) {
  const int rank = std::atoi(argv[2]);
  return rank == 3 ? run<t8gpu::Subgrid<4, 4, 4>>(argv[1]) : run<t8gpu::Subgrid<4, 4>>(argv[1]);
}
"""


def lines(path, a=None, b=None, drop_includes=True):
    """lines a..b (1-based, inclusive) of a reference file, `#include` lines dropped"""
    ls = open(os.path.join(REF, path)).read().split("\n")
    ls = ls[(a - 1 if a else 0):(b if b else len(ls))]
    if drop_includes:
        ls = [x for x in ls if not x.lstrip().startswith("#include")]
    return "\n".join(ls) + "\n"


def between(path, start_marker, end_marker):
    """the text of a reference file from the line that contains start_marker up to (not including) the one with end_marker"""
    ls = open(os.path.join(REF, path)).read().split("\n")
    a = next(i for i, x in enumerate(ls) if start_marker in x)
    b = next(i for i, x in enumerate(ls) if end_marker in x and i > a)
    return "\n".join(x for x in ls[a:b] if not x.lstrip().startswith("#include")) + "\n"


def perturbed(n, seed):
    """conserved states [5, n]: rho in [0.6, 1.8], |v| <= 0.6, p in [0.8, 3] (smooth enough for every branch to be ordinary)"""
    rng = np.random.default_rng(seed)
    rho = rng.uniform(0.6, 1.8, n)
    v = rng.uniform(-0.6, 0.6, (3, n))
    p = rng.uniform(0.8, 3.0, n)
    return np.stack([rho, rho * v[0], rho * v[1], rho * v[2], p / 0.4 + 0.5 * rho * (v ** 2).sum(0)])


def main():
    if not os.path.isdir(REF):
        sys.exit("this generator needs the reference tree at /root/reference (build container only)")
    from t8gpu_amd.synth import SynthMesh

    def adapted(dim, level, refine, periodic):
        """uniform level `level`, the listed elements refined once (2:1 balanced by the provider): a few dozen elements with
        hanging faces on every side of the refined ones"""
        m = SynthMesh(dim, level, level + 1, band=0.0, periodic=periodic)
        marks = np.zeros(m.num_elements, np.int8)
        marks[list(refine)] = 1
        return m.adapt(marks)[0]

    plain_cases = {"plain2d_wall": adapted(2, 2, (3, 9), False).partition(),
                   "plain3d_periodic": adapted(3, 2, (5, 42), True).partition(),
                   "plain3d_wall": adapted(3, 1, (3,), False).partition()}
    sub_cases = {"sub3d_periodic": (adapted(3, 1, (0,), True).partition(subgrid=True), 3),
                 "sub3d_wall": (adapted(3, 2, (5, 42), False).partition(subgrid=True), 3),
                 "sub2d_periodic": (adapted(2, 2, (6,), True).partition(subgrid=True), 2),
                 "sub2d_wall": (adapted(2, 2, (3, 9), False).partition(subgrid=True), 2)}
    res = {}
    with tempfile.TemporaryDirectory(prefix="t8gpu_refkern_") as tmp:
        w = lambda name, text: open(os.path.join(tmp, name), "w").write(text)
        w("ref_meta.inl", between("t8gpu/utils/meta.h", "namespace t8gpu::meta {", "#endif  // UTILS_META_H"))
        mem = between("t8gpu/memory/memory_manager.h", "// Type traits are necessary", "  class MemoryManager {")
        mem = mem[:mem.rindex("  template<typename VariableType, typename StepType>")]
        sub = between("t8gpu/memory/subgrid_memory_manager.h", "// Forward declarations.", "  class SubgridMemoryManager {")
        sub = sub[:sub.rindex("  template<typename VariableType, typename StepType, typename SubgridType>")]
        w("ref_submem.inl", sub)
        w("ref_conn.inl", between("t8gpu/mesh/mesh_manager.h", "/// Forward declaration of MeshManager class", "/// @brief class that represents a distributed mesh"))
        w("ref_subconn.inl", between("t8gpu/mesh/subgrid_mesh_manager.h", "/// Forward declaration of MeshManager class", "/// @brief class that represents a distributed mesh"))
        w("ref_kernels_h.inl", between("examples/compressible_euler/kernels.h", "__global__ void kepes_compute_fluxes", "}  // namespace t8gpu"))
        rkh = open(os.path.join(REF, "t8gpu/timestepping/ssp_runge_kutta.h")).read()
        rkh = rkh[rkh.index("namespace t8gpu::timestepping {"):rkh.index('#include "ssp_runge_kutta.inl"')]
        w("ref_rk_h.inl", rkh)
        w("ref_rk.inl", lines("t8gpu/timestepping/ssp_runge_kutta.inl"))
        w("ref_kernels_plain.inl", lines("examples/compressible_euler/kernels.cu", 1, 469))
        w("ref_kernels_subgrid.inl", lines("examples/subgrid/kernels.inl", 1, 1107))
        w("plain.cpp", PLAIN)
        w("subgrid.cpp", SUBGRID)
        for ft, npdt, tag in (("double", np.float64, "f64"), ("float", np.float32, "f32")):
            text = mem
            if ft == "double":                       # variable_traits (:29) and step_traits (:39): the two `float` tokens
                assert text.count("= float;") == 2
                text = text.replace("= float;", "= double;")
            w("ref_mem.inl", text)
            exes = {}
            for name in ("plain", "subgrid"):
                exes[name] = os.path.join(tmp, f"{name}_{tag}")
                subprocess.check_call(["g++", "-std=c++17", "-O0", "-ffp-contract=off", "-pthread", "-w", "-I", tmp,
                                       os.path.join(tmp, name + ".cpp"), "-o", exes[name]])

            def run_case(cname, part, rank):
                S = 1 if rank is None else 4 ** rank
                tot = part.N + part.G
                stride = tot * S
                d = os.path.join(tmp, f"{cname}_{tag}")
                os.makedirs(d)
                seed = sum(ord(c) for c in cname)                       # (hash() is salted per process)
                ins = {"sizes": np.array([part.N, part.G, part.F, part.B, stride], np.int32),
                       "fn": np.ascontiguousarray(part.face_neighbors, np.int32),
                       "idx": np.arange(tot, dtype=np.int32) if part.indices is None else np.ascontiguousarray(part.indices, np.int32),
                       "normals": np.ascontiguousarray(part.normals, npdt), "areas": np.ascontiguousarray(part.areas, npdt),
                       "state": np.ascontiguousarray(perturbed(stride, seed), npdt),
                       "volume": np.ascontiguousarray(part.volumes, npdt),
                       "rk_mid": np.ascontiguousarray(perturbed(stride, seed + 1), npdt),
                       "rk_flux": np.ascontiguousarray(0.01 * perturbed(stride, seed + 2), npdt),
                       "dt": np.array([0.01 * 2.0 ** -part.mesh.finest_level], npdt)}
                if rank is not None:
                    ins["level_diff"] = np.ascontiguousarray(part.level_diff, np.int32)
                    ins["nb_offset"] = np.ascontiguousarray(part.nb_offset, np.int32)
                for k, v in ins.items():
                    v.tofile(os.path.join(d, k + ".bin"))
                subprocess.check_call([exes["plain" if rank is None else "subgrid"], d] + ([] if rank is None else [str(rank)]))
                outs = (["flux_interior", "flux_all", "speed"] if rank is None else ["flux_inner", "flux_inner_boundary", "flux_all"])
                outs += [f"rk_out{s}" for s in (1, 2, 3)] + [f"rk_flux_after{s}" for s in (1, 2, 3)]
                for k, v in ins.items():
                    res[f"{cname}|{tag}|in|{k}"] = v
                for k in outs:
                    res[f"{cname}|{tag}|out|{k}"] = np.fromfile(os.path.join(d, k + ".bin"), npdt)
                res[f"{cname}|{tag}|in|normal_dim"] = np.array([part.normal_dim], np.int32)
                res[f"{cname}|{tag}|in|rank"] = np.array([0 if rank is None else rank], np.int32)

            for cname, part in plain_cases.items():
                run_case(cname, part, None)
            for cname, (part, rank) in sub_cases.items():
                run_case(cname, part, rank)
    res["meta"] = np.array([
        "reference kernel bodies (examples/compressible_euler/kernels.cu:135-469, examples/subgrid/kernels.inl:335-1107, "
        "t8gpu/timestepping/ssp_runge_kutta.inl:30-221) executed on the host through the reference's own accessor classes; "
        "CUDA built-ins, the accessor-owning friend classes and the example solver structs are stand-ins; threads of a block run "
        "in descending order between barriers (lock-step result for SURVEY quirk Q6). Parity stays 'unpinned' (DESIGN.md section 2). "
        "Meshes: synthetic provider; states seeded per case name."])
    path = os.path.join(HERE, "reference_kernel_vectors.npz")
    np.savez_compressed(path, **res)
    bad = [k for k, v in res.items() if k != "meta" and v.dtype.kind == "f" and not np.isfinite(v).all()]
    print(f"wrote {path}: {len(res) - 1} arrays ({os.path.getsize(path) / 1024:.0f} KiB); non-finite: {bad}")
    for cname, part in list(plain_cases.items()) + [(k, v[0]) for k, v in sub_cases.items()]:
        print(f"  {cname}: N={part.N} F={part.F} B={part.B}")


if __name__ == "__main__":
    main()
