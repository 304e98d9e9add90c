// tests/compat/subgrid_api.hip -- the Subgrid part of the header API, used the way
// examples/subgrid/{solver.inl,kernels.inl} use it: Subgrid<4,4,4> traits, per-variable accessors
// called as rho(e_idx, i, j, k), kernels launched with SubgridType::block_size, the library's
// timestepping::subgrid::SSP_3RK_step kernels, get(rank, vars...) on the "All" accessor.
// Self-checking: prints "subgrid_api OK" and returns 0.
#include <t8gpu/memory/subgrid_memory_manager.h>
#include <t8gpu/mesh/subgrid_mesh_manager.h>
#include <t8gpu/timestepping/ssp_runge_kutta.h>
#include <t8gpu/backend/hip_fast.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

using namespace t8gpu;

enum VariableList { Rho, Rho_v1, Rho_v2, Rho_v3, Rho_e, nb_variables };
enum StepList { Step0, Step1, Step2, Step3, Fluxes, nb_steps };

using Grid3 = Subgrid<4, 4, 4>;
using Grid2 = Subgrid<4, 4>;
static_assert(Grid3::rank == 3 && Grid3::size == 64 && Grid2::rank == 2 && Grid2::size == 16);
static_assert(Grid3::extent<0> == 4 && Grid3::extent<2> == 4);
static_assert(Grid3::stride<0> == 1 && Grid3::stride<1> == 4 && Grid3::stride<2> == 16);
static_assert(Grid3::flat_index(1, 2, 3) == 1 + 4 * 2 + 16 * 3 && Grid2::flat_index(3, 1) == 7);
static_assert(Grid3::block_size.x == 4 && Grid3::block_size.y == 4 && Grid3::block_size.z == 4 && Grid2::block_size.z == 1);
static_assert(meta::log2_v<Grid3::extent<0>> == 2 && meta::all_same_v<int, const int> && !meta::all_same_v<int, long>);
static_assert(meta::argpack_mul_from_v<1, 2, 3, 5> == 15 && meta::argpack_mul_to_v<2, 2, 3, 5> == 6 && meta::argpack_at_v<1, 7, 8, 9> == 8);

using float_type = variable_traits<VariableList>::float_type;

template<typename SubgridType>
__global__ void fill(SubgridMemoryAccessorOwn<VariableList, SubgridType> vars, SubgridMemoryAccessorOwn<VariableList, SubgridType> flux) {
  const int e = blockIdx.x, i = threadIdx.x, j = threadIdx.y, k = threadIdx.z;
  auto [rho, rho_v1, rho_v2, rho_v3, rho_e] = vars.get(Rho, Rho_v1, Rho_v2, Rho_v3, Rho_e);
  rho(e, i, j, k)    = float_type(e * 64 + i + 4 * j + 16 * k);
  rho_v1(e, i, j, k) = 1;
  rho_v2(e, i, j, k) = 2;
  rho_v3(e, i, j, k) = 3;
  rho_e(e, i, j, k)  = 4;
  for (int v = 0; v < nb_variables; v++) flux.get(v)(e, i, j, k) = float_type(v + 1);
}

template<typename SubgridType>
__global__ void read_all(SubgridMemoryAccessorAll<VariableList, SubgridType> vars, float_type* out) {
  auto [rho, rho_e] = vars.get(0, Rho, Rho_e);   // rank 0
  out[0]            = rho(1, 3, 2, 1) + rho_e(1, 0, 0, 0);
}

int main() {
  const int nblocks = 3;
  SubgridMemoryManager<VariableList, StepList, Grid3> mem(nblocks);
  mem.set_volume(std::vector<float_type>(nblocks, float_type(64.0)));   // per-subcell volume 1
  fill<Grid3><<<nblocks, Grid3::block_size>>>(mem.get_own_variables(Step0), mem.get_own_variables(Fluxes));
  const float_type dt = 0.5;
  timestepping::subgrid::SSP_3RK_step1<VariableList, Grid3><<<nblocks, Grid3::block_size>>>(
      mem.get_own_variables(Step0), mem.get_own_variables(Step1), mem.get_own_variables(Fluxes), mem.get_own_volume(), dt);
  float_type* dout = nullptr;
  T8GPU_CUDA_CHECK_ERROR(hipMalloc(&dout, sizeof(float_type)));
  read_all<Grid3><<<1, 1>>>(mem.get_all_variables(Step1), dout);
  float_type got = 0;
  T8GPU_CUDA_CHECK_ERROR(hipMemcpy(&got, dout, sizeof(float_type), hipMemcpyDeviceToHost));
  // Step1 = Step0 + dt / 1 * flux: rho(1,3,2,1) = 64+3+8+16 + 0.5*1, rho_e = 4 + 0.5*5
  const float_type want = float_type(91.5 + 6.5);
  std::vector<float_type> flux(nblocks * 64);
  T8GPU_CUDA_CHECK_ERROR(hipMemcpy(flux.data(), static_cast<float_type*>(mem.get_own_variable(Fluxes, Rho_e)), sizeof(float_type) * flux.size(),
                                   hipMemcpyDeviceToHost));
  bool zeroed = true;
  for (float_type f : flux) zeroed = zeroed && f == 0;
  (void)hipFree(dout);
  if (std::fabs(got - want) > 1e-6 || !zeroed) {
    std::printf("subgrid_api FAILED: got %g want %g zeroed %d\n", double(got), double(want), int(zeroed));
    return 1;
  }
  // ---- the fused block kernel through its C++ face, against the reference-dataflow kernels of the C-ABI ------------
  {
    void*   mesh = t8gpu_synth_mesh_create(3, 2, 3, 0.12, 1.0, 0);     // 3D, levels 2-3, walls
    void*   part = t8gpu_synth_part_create(mesh, 0, 1, 1, 3);
    int64_t cnt[8];
    t8gpu_synth_part_counts(part, cnt);
    const int N = static_cast<int>(cnt[0]), F = static_cast<int>(cnt[2]), B = static_cast<int>(cnt[3]);
    HostSubgridMeshArrays m;
    m.num_local_elements = N; m.num_ghost_elements = static_cast<int32_t>(cnt[1]); m.num_local_faces = F; m.num_local_boundary_faces = B;
    m.rank = 3;
    m.face_neighbors.resize(2 * F + B); m.face_level_difference.resize(F); m.face_neighbor_offset.resize(3 * F);
    m.face_normals.resize(3 * (F + B)); m.face_surfaces.resize(F + B); m.volumes.resize(N);
    t8gpu_synth_part_connectivity(part, m.face_neighbors.data(), m.face_normals.data(), m.face_surfaces.data(),
                                  m.face_level_difference.data(), m.face_neighbor_offset.data());
    m.centres.resize(3 * static_cast<size_t>(N));
    m.levels.resize(N);
    t8gpu_synth_part_elements(part, m.levels.data(), m.volumes.data(), m.centres.data());
    std::vector<double> ic(5 * static_cast<size_t>(N) * 64);
    t8gpu_synth_part_kh_ic(part, 4, ic.data(), static_cast<size_t>(N) * 64);
    auto make = [&](SubgridMemoryManager<VariableList, StepList, Grid3>& mm) {
      mm.set_volume(std::vector<float_type>(m.volumes.begin(), m.volumes.end()));
      for (int s = 0; s < nb_steps; s++)
        for (int v = 0; v < 5; v++) mm.set_variable(static_cast<StepList>(s), static_cast<VariableList>(v), std::vector<float_type>(static_cast<size_t>(N) * 64, 0));
      for (int v = 0; v < 5; v++)
        mm.set_variable(Step0, static_cast<VariableList>(v),
                        std::vector<float_type>(ic.begin() + static_cast<size_t>(v) * N * 64, ic.begin() + static_cast<size_t>(v + 1) * N * 64));
    };
    SubgridMemoryManager<VariableList, StepList, Grid3> a(N), b(N);
    make(a);
    make(b);
    hip::SubgridFusedPlan<float_type> plan(m);
    // device copies of the connectivity for the reference-dataflow kernels
    auto up = [](auto const& v) {
      using T = typename std::decay_t<decltype(v)>::value_type;
      T* d = nullptr;
      T8GPU_CUDA_CHECK_ERROR(hipMalloc(&d, sizeof(T) * (v.empty() ? 1 : v.size())));
      if (!v.empty()) T8GPU_CUDA_CHECK_ERROR(hipMemcpy(d, v.data(), sizeof(T) * v.size(), hipMemcpyHostToDevice));
      return d;
    };
    int32_t *fn = up(m.face_neighbors), *ld = up(m.face_level_difference), *off = up(m.face_neighbor_offset);
    float_type *nrm = up(std::vector<float_type>(m.face_normals.begin(), m.face_normals.end())),
               *ars = up(std::vector<float_type>(m.face_surfaces.begin(), m.face_surfaces.end()));
    const float_type dts = float_type(0.1 * std::pow(0.5, t8gpu_synth_mesh_finest_level(mesh) + 2));
    StepList next = Step0, prev = Step3;
    for (int it = 0; it < 3; it++) {
      std::swap(next, prev);
      hip::iterate_fused(a, plan, prev, next, dts);
      const StepList src[3] = {prev, Step1, Step2}, dst[3] = {Step1, Step2, next};
      for (int k = 0; k < 3; k++) {   // the reference's launch sequence (solver.inl:166-195) on the C-ABI
        auto st = hip::to_vars(b.get_own_variables(src[k])), fl = hip::to_vars(b.get_own_variables(Fluxes));
        T8GPU_HIP_CHECK_ABI(t8gpu_hip_subgrid_inner_f32(T8GPU_FLUX_KEPES, 3, N, st, fl, b.get_own_volume(), nullptr));
        T8GPU_HIP_CHECK_ABI(t8gpu_hip_subgrid_boundary_f32(T8GPU_FLUX_KEPES, 3, F, B, fn, nrm, ars, st, fl, nullptr));
        T8GPU_HIP_CHECK_ABI(t8gpu_hip_subgrid_outer_f32(T8GPU_FLUX_KEPES, 3, F, fn, nullptr, ld, off, nrm, ars, st, fl, nullptr));
        T8GPU_HIP_CHECK_ABI(t8gpu_hip_subgrid_rk3_stage_f32(k + 1, 3, N, hip::to_vars(b.get_own_variables(prev)), st,
                                                            hip::to_vars(b.get_own_variables(dst[k])), fl, b.get_own_volume(), dts, nullptr));
      }
    }
    T8GPU_CUDA_CHECK_ERROR(hipDeviceSynchronize());
    double worst = 0, scale = 0;
    for (int v = 0; v < 5; v++) {
      std::vector<float_type> ha(static_cast<size_t>(N) * 64), hb(ha.size());
      T8GPU_CUDA_CHECK_ERROR(hipMemcpy(ha.data(), static_cast<float_type*>(a.get_own_variable(next, static_cast<VariableList>(v))), sizeof(float_type) * ha.size(), hipMemcpyDeviceToHost));
      T8GPU_CUDA_CHECK_ERROR(hipMemcpy(hb.data(), static_cast<float_type*>(b.get_own_variable(next, static_cast<VariableList>(v))), sizeof(float_type) * hb.size(), hipMemcpyDeviceToHost));
      for (size_t i = 0; i < ha.size(); i++) {
        worst = std::max(worst, std::fabs(double(ha[i]) - double(hb[i])));
        scale = std::max(scale, std::fabs(double(hb[i])));
      }
    }
    // the mesh-manager face of the same arrays: accessor counts and the VTK members (subgrid_mesh_manager.inl:1051-1206)
    {
      SubgridMeshManager<VariableList, StepList, Grid3> mm(m);   // the reference's class name (subgrid_mesh_manager.h:266)
      if (mm.get_num_local_elements() != N || mm.get_connectivity_information().get_num_local_faces() != F) return 3;
      for (int v = 0; v < 5; v++)
        mm.set_variable(Step0, static_cast<VariableList>(v),
                        std::vector<float_type>(ic.begin() + static_cast<size_t>(v) * N * 64, ic.begin() + static_cast<size_t>(v + 1) * N * 64));
      const char* out = std::getenv("T8GPU_TEST_VTK_PREFIX");
      if (out) {
        mm.save_variable_to_vtk(Step0, Rho, std::string(out) + "_rho");
        mm.save_mesh_to_vtk(std::string(out) + "_mesh");
        // get_host_{scalar,vector}_variable + save_variables_to_vtk (subgrid_mesh_manager.h:426-446)
        std::vector<SubgridMeshManager<VariableList, StepList, Grid3>::HostVariableInfo> fields;
        fields.push_back(mm.get_host_scalar_variable(Step0, Rho, "density"));
        fields.push_back(mm.get_host_vector_variable(Step0, {Rho_v1, Rho_v2, Rho_v3}, "momentum"));
        mm.save_variables_to_vtk(std::move(fields), std::string(out) + "_fields");
      }
    }
    for (void* p : {static_cast<void*>(fn), static_cast<void*>(ld), static_cast<void*>(off), static_cast<void*>(nrm), static_cast<void*>(ars)}) (void)hipFree(p);
    t8gpu_synth_part_destroy(part);
    t8gpu_synth_mesh_destroy(mesh);
    if (!(worst < 1e-4 * scale) || !(scale > 0)) {
      std::printf("subgrid_api FAILED: fused vs reference-dataflow kernels differ by %g (scale %g)\n", worst, scale);
      return 1;
    }
    std::printf("fused block kernel vs reference-dataflow kernels over 3 steps: max |diff| = %.3g (scale %.3g)\n", worst, scale);
  }
  // ---- SubgridMeshManager::adapt in C++ (subgrid_mesh_manager.inl:428-558): criteria kernel, forest adapt, block-wise
  //      transfer, new connectivity, fused steps on the new mesh; mass must be kept ---------------------------------
  {
    using SubgridManager = SubgridMeshManager<VariableList, StepList, Grid3>;
    static_assert(SubgridManager::dim == 3 && SubgridManager::nb_variables == 5 && SubgridManager::max_level == 6);
    SubgridManager mm(t8gpu_synth_mesh_create(3, 2, 2, 0.0, 1.0, 1), 1, 3);
    // initialize_variables (subgrid_mesh_manager.inl:144-194): one value per block, broadcast to its 64 subcells
    mm.initialize_variables([](MemoryAccessorOwn<VariableList>& accessor, t8_forest_t, t8_locidx_t, t8_element_t const* element,
                               t8_locidx_t e_idx) {
      auto [rho, rho_e] = accessor.get(Rho, Rho_e);
      rho[e_idx]        = synthetic_element(element).centre[2] < 0.5 ? float_type(2) : float_type(1);
      rho_e[e_idx]      = float_type(6.25) + float_type(synthetic_element(element).level);
    });
    {
      const size_t            n = static_cast<size_t>(mm.get_num_local_elements()) * 64;
      std::vector<float_type> h(n);
      T8GPU_CUDA_CHECK_ERROR(hipMemcpy(h.data(), static_cast<float_type*>(mm.get_own_variable(Step0, Rho)), sizeof(float_type) * n, hipMemcpyDeviceToHost));
      int twos = 0;
      for (size_t i = 0; i < n; i++) {
        if (h[i] != h[i - i % 64] || (h[i] != float_type(1) && h[i] != float_type(2))) return 4;   // constant per block
        twos += h[i] == float_type(2);
      }
      if (2 * static_cast<size_t>(twos) != n) return 5;                                               // lower half of the cube
    }
    auto fill_ic = [&]() {
      void*        part = t8gpu_synth_part_create(mm.forest(), 0, 1, 1, 3);
      const size_t n    = static_cast<size_t>(mm.get_num_local_elements()) * 64;
      std::vector<double> ic(5 * n);
      t8gpu_synth_part_kh_ic(part, 4, ic.data(), n);
      t8gpu_synth_part_destroy(part);
      for (int st = 0; st < nb_steps; st++)
        for (int v = 0; v < 5; v++) mm.set_variable(static_cast<StepList>(st), static_cast<VariableList>(v), std::vector<float_type>(n, 0));
      for (int v = 0; v < 5; v++)
        mm.set_variable(Step0, static_cast<VariableList>(v), std::vector<float_type>(ic.begin() + v * n, ic.begin() + (v + 1) * n));
    };
    fill_ic();
    hip::Reducer reduce;
    auto mass = [&](StepList st) {
      return reduce.integral<float_type>(static_cast<size_t>(mm.get_num_local_elements()) * 64,
                                         static_cast<float_type const*>(mm.get_own_variable(st, Rho)), mm.get_own_volume(), 64);
    };
    const double m0 = mass(Step0);
    const int    n0 = mm.get_num_local_elements();
    float_type*  crit = nullptr;
    T8GPU_CUDA_CHECK_ERROR(hipMalloc(&crit, sizeof(float_type) * n0));
    T8GPU_HIP_CHECK_ABI(t8gpu_hip_subgrid_refinement_criteria_f32(3, n0, static_cast<float_type const*>(mm.get_own_variable(Step0, Rho)),
                                                                  mm.get_own_volume(), crit, nullptr));
    std::vector<float_type> hc(n0);
    T8GPU_CUDA_CHECK_ERROR(hipMemcpy(hc.data(), crit, sizeof(float_type) * n0, hipMemcpyDeviceToHost));
    (void)hipFree(crit);
    mm.adapt(thrust::host_vector<float_type>(hc.begin(), hc.end()), Step0);   // the reference's signature
    mm.partition(Step0);
    mm.compute_connectivity_information();
    const int    n1 = mm.get_num_local_elements();
    const double m1 = mass(Step0);
    hip::SubgridFusedPlan<float_type> plan(mm.host_arrays());
    for (int st = 1; st < nb_steps; st++)
      for (int v = 0; v < 5; v++)
        mm.set_variable(static_cast<StepList>(st), static_cast<VariableList>(v), std::vector<float_type>(static_cast<size_t>(n1) * 64, 0));
    StepList next = Step0, prev = Step3;
    const float_type dts = float_type(0.1 * std::pow(0.5, t8gpu_synth_mesh_finest_level(mm.forest()) + 2));
    for (int it = 0; it < 3; it++) {
      std::swap(next, prev);
      hip::iterate_fused(mm, plan, prev, next, dts);
    }
    T8GPU_CUDA_CHECK_ERROR(hipDeviceSynchronize());
    const double m2 = mass(next);
    std::printf("subgrid adapt: %d -> %d blocks, mass %.9g -> %.9g -> %.9g after 3 steps\n", n0, n1, m0, m1, m2);
    if (n1 == n0 || !(std::fabs(m1 - m0) <= 1e-5 * std::fabs(m0)) || !(std::fabs(m2 - m0) <= 1e-5 * std::fabs(m0))) {
      std::printf("subgrid_api FAILED: adapt\n");
      return 1;
    }
  }
  std::printf("subgrid_api OK\n");
  return 0;
}
