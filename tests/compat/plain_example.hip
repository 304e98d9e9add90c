// tests/compat/plain_example.hip -- a solver written the way the reference's example is written
// (examples/compressible_euler/solver.{h,cu}): VariableList / StepList enums, a mesh manager that IS a
// MemoryManager, accessor PODs passed by value to __global__ kernels launched with <<<>>>, structured
// bindings on get(...), `var[rank][index]` through get_all_variables(), the library's
// timestepping::SSP_3RK_stepK kernels, next/prev swapping. It checks that code in that style compiles
// and runs against the HIP-backed headers of include/t8gpu/, and runs the same steps three ways:
//   A  user-launched kernels through the accessor API            ("examples build against the new backend")
//   B  the C-ABI's reference-dataflow kernels via t8gpu/backend/hip_fast.h
//   C  the fused step driver via t8gpu::hip::iterate_fused
// Results go to a binary file; tests/test_gpu_headers.py compares them with the CPU oracle.
#include <t8gpu/backend/hip_fast.h>
#include <t8gpu/mesh/mesh_manager.h>
#include <t8gpu/timestepping/ssp_runge_kutta.h>
#include <t8gpu/utils/cuda.h>
#include <t8gpu/utils/profiling.h>

#include <cstdio>
#include <cstdlib>
#include <utility>
#include <vector>

#include "flux_math.hpp"

using namespace t8gpu;

enum VariableList { Rho, Rho_v1, Rho_v2, Rho_v3, Rho_e, nb_variables };
enum StepList { Step0, Step1, Step2, Step3, Fluxes, nb_steps };

using float_type         = variable_traits<VariableList>::float_type;
static constexpr size_t dim = 3;

// same shape as the reference's kepes_compute_fluxes (kernels.cu:135-309): one thread per face,
// double indirection through (rank, index), 10 atomicAdd
__global__ void user_compute_fluxes(MeshConnectivityAccessor<float_type, dim> connectivity,
                                    MemoryAccessorAll<VariableList> variables, MemoryAccessorAll<VariableList> fluxes,
                                    float_type* __restrict__ speed_estimates) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= connectivity.get_num_local_faces()) return;
  const float_type face_surface = connectivity.get_face_surface(i);
  auto [l_idx, r_idx]           = connectivity.get_face_neighbor_indices(i);
  const int l_rank = connectivity.get_element_owner_rank(l_idx), l_index = connectivity.get_element_owner_remote_index(l_idx);
  const int r_rank = connectivity.get_element_owner_rank(r_idx), r_index = connectivity.get_element_owner_remote_index(r_idx);
  auto [nx, ny, nz] = connectivity.get_face_normal(i);
  auto [rho, rho_v1, rho_v2, rho_v3, rho_e] = variables.get(Rho, Rho_v1, Rho_v2, Rho_v3, Rho_e);
  float_type sl[5] = {rho[l_rank][l_index], rho_v1[l_rank][l_index], rho_v2[l_rank][l_index], rho_v3[l_rank][l_index], rho_e[l_rank][l_index]};
  float_type sr[5] = {rho[r_rank][r_index], rho_v1[r_rank][r_index], rho_v2[r_rank][r_index], rho_v3[r_rank][r_index], rho_e[r_rank][r_index]};
  float_type n[3] = {nx, ny, nz}, t1[3], t2[3], Ff[5], g[5], spd;
  t8gpu_hip::face_basis<float_type>(n, t1, t2);
  t8gpu_hip::face_frame_flux_ref<float_type, 0>(n, t1, t2, sl, sr, false, Ff, spd);
  speed_estimates[i] = spd;
  for (int k = 0; k < 5; k++) Ff[k] = face_surface * Ff[k];
  t8gpu_hip::from_face_frame<float_type>(n, t1, t2, Ff, g);
  for (int k = 0; k < 5; k++) {
    atomicAdd(&fluxes.get(k)[l_rank][l_index], -g[k]);
    atomicAdd(&fluxes.get(k)[r_rank][r_index], g[k]);
  }
}

__global__ void user_reflective_boundary(MeshConnectivityAccessor<float_type, dim> connectivity,
                                         MemoryAccessorOwn<VariableList> variables, MemoryAccessorOwn<VariableList> fluxes,
                                         float_type* __restrict__ speed_estimates) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= connectivity.get_num_local_boundary_faces()) return;
  const float_type face_surface = connectivity.get_boundary_face_surface(i);
  const auto       e            = connectivity.get_boundary_face_neighbor_index(i);
  auto [nx, ny, nz]             = connectivity.get_boundary_face_normal(i);
  auto [rho, rho_v1, rho_v2, rho_v3, rho_e] = variables.get(Rho, Rho_v1, Rho_v2, Rho_v3, Rho_e);
  float_type s[5] = {rho[e], rho_v1[e], rho_v2[e], rho_v3[e], rho_e[e]};
  float_type n[3] = {nx, ny, nz}, t1[3], t2[3], Ff[5], g[5], spd;
  t8gpu_hip::face_basis<float_type>(n, t1, t2);
  t8gpu_hip::face_frame_flux_ref<float_type, 0>(n, t1, t2, s, s, true, Ff, spd);
  speed_estimates[connectivity.get_num_local_faces() + i] = spd;
  for (int k = 0; k < 5; k++) Ff[k] = face_surface * Ff[k];
  t8gpu_hip::from_face_frame<float_type>(n, t1, t2, Ff, g);
  for (int k = 0; k < 5; k++) atomicAdd(&fluxes.get(k)[e], -g[k]);
}

struct Solver {
  SyntheticMeshManager<VariableList, StepList, dim> mesh;
  float_type*                                        speed = nullptr;
  StepList next = Step0, prev = Step3;  // solver.h:100-101

  explicit Solver(HostMeshArrays const& m, std::vector<double> const& ic) : mesh(m) {
    const size_t tot = static_cast<size_t>(m.num_local_elements) + m.num_ghost_elements;
    for (int k = 0; k < 5; k++)
      for (int s = 0; s < nb_steps; s++) mesh.set_variable(static_cast<StepList>(s), static_cast<VariableList>(k), std::vector<float_type>(tot, 0));
    for (int k = 0; k < 5; k++)
      mesh.set_variable(next, static_cast<VariableList>(k), std::vector<float_type>(ic.begin() + k * tot, ic.begin() + (k + 1) * tot));
    T8GPU_CUDA_CHECK_ERROR(hipMalloc(&speed, sizeof(float_type) * (m.num_local_faces + m.num_local_boundary_faces + 1)));
  }
  ~Solver() { (void)hipFree(speed); }

  // the example's output step, written as the reference writes it (examples/compressible_euler/solver.cu:177-186)
  void save_conserved_variables_to_vtk(std::string prefix) const {
    std::array<VariableList, 3> momentum = {Rho_v1, Rho_v2, Rho_v3};
    std::vector<SyntheticMeshManager<VariableList, StepList, dim>::HostVariableInfo> variables;
    variables.push_back(mesh.get_host_scalar_variable(next, Rho, "density"));
    variables.push_back(mesh.get_host_scalar_variable(next, Rho_e, "energy"));
    variables.push_back(mesh.get_host_vector_variable(next, momentum, "momentum"));
    mesh.save_variables_to_vtk(std::move(variables), prefix);
  }

  // A: the reference's iterate(), launch for launch (solver.cu:75-175)
  void iterate_user_kernels(float_type delta_t) {
    std::swap(next, prev);
    constexpr int tbs = 256;
    const int     nf = mesh.get_num_local_faces(), nb = mesh.get_num_local_boundary_faces(), ne = mesh.get_num_local_elements();
    const StepList src[3] = {prev, Step1, Step2};
    for (int s = 0; s < 3; s++) {
      user_compute_fluxes<<<(nf + tbs - 1) / tbs, tbs>>>(mesh.get_connectivity_information(), mesh.get_all_variables(src[s]),
                                                        mesh.get_all_variables(Fluxes), speed);
      T8GPU_CUDA_CHECK_LAST_ERROR();
      if (nb > 0)
        user_reflective_boundary<<<(nb + tbs - 1) / tbs, tbs>>>(mesh.get_connectivity_information(), mesh.get_own_variables(src[s]),
                                                               mesh.get_own_variables(Fluxes), speed);
      T8GPU_CUDA_CHECK_LAST_ERROR();
      const int blocks = (ne + tbs - 1) / tbs;
      if (s == 0)
        timestepping::SSP_3RK_step1<VariableList><<<blocks, tbs>>>(mesh.get_own_variables(prev), mesh.get_own_variables(Step1),
                                                                  mesh.get_own_variables(Fluxes), mesh.get_own_volume(), delta_t, ne);
      else if (s == 1)
        timestepping::SSP_3RK_step2<VariableList><<<blocks, tbs>>>(mesh.get_own_variables(prev), mesh.get_own_variables(Step1),
                                                                  mesh.get_own_variables(Step2), mesh.get_own_variables(Fluxes),
                                                                  mesh.get_own_volume(), delta_t, ne);
      else
        timestepping::SSP_3RK_step3<VariableList><<<blocks, tbs>>>(mesh.get_own_variables(prev), mesh.get_own_variables(Step2),
                                                                  mesh.get_own_variables(next), mesh.get_own_variables(Fluxes),
                                                                  mesh.get_own_volume(), delta_t, ne);
      T8GPU_CUDA_CHECK_LAST_ERROR();
    }
  }

  // B: same structure, the three launches of a stage replaced by the C-ABI's kernels
  void iterate_cabi_compat(float_type delta_t) {
    std::swap(next, prev);
    const StepList src[3] = {prev, Step1, Step2}, dst[3] = {Step1, Step2, next};
    auto           conn   = mesh.get_connectivity_information();
    for (int s = 0; s < 3; s++) {
      hip::flux_faces<VariableList, dim>(conn, mesh.get_own_variables(src[s]), mesh.get_own_variables(Fluxes), speed);
      hip::flux_boundary<VariableList, dim>(conn, mesh.get_own_variables(src[s]), mesh.get_own_variables(Fluxes), speed);
      hip::rk3_stage<VariableList>(s + 1, mesh.get_num_local_elements(), mesh.get_own_variables(prev), mesh.get_own_variables(src[s]),
                                   mesh.get_own_variables(dst[s]), mesh.get_own_variables(Fluxes), mesh.get_own_volume(), delta_t);
    }
  }

  // C: the fused step driver
  void iterate_fused(hip::PlainFusedPlan<float_type> const& plan, float_type delta_t) {
    std::swap(next, prev);
    hip::iterate_fused(mesh, plan, prev, next, delta_t, speed);
  }

  // C': the whole run in one call of the driver (prev / next alternate inside; an odd count leaves them swapped)
  void iterate_fused_steps(hip::PlainFusedPlan<float_type> const& plan, float_type delta_t, int n) {
    std::swap(next, prev);
    hip::iterate_fused(mesh, plan, prev, next, delta_t, speed, T8GPU_FLUX_KEPES, nullptr, n);
    if (n % 2 == 0) std::swap(next, prev);
  }

  std::vector<float_type> download() {
    const size_t            n = mesh.get_num_local_elements();
    std::vector<float_type> out(5 * n);
    for (int k = 0; k < 5; k++)
      T8GPU_CUDA_CHECK_ERROR(hipMemcpy(out.data() + k * n, mesh.get_own_variable(next, static_cast<VariableList>(k)), sizeof(float_type) * n,
                                       hipMemcpyDeviceToHost));
    return out;
  }
};

int main(int argc, char** argv) {
  if (argc < 8) {
    std::fprintf(stderr, "usage: %s dim base max band periodic steps out.bin\n", argv[0]);
    return 2;
  }
  const int    mdim = std::atoi(argv[1]), base = std::atoi(argv[2]), lmax = std::atoi(argv[3]);
  const double band = std::atof(argv[4]);
  const int    periodic = std::atoi(argv[5]), steps = std::atoi(argv[6]);
  void*        mesh = t8gpu_synth_mesh_create(mdim, base, lmax, band, 1.0, periodic);
  void*        part = t8gpu_synth_part_create(mesh, 0, 1, 0, 3);
  int64_t      cnt[8];
  t8gpu_synth_part_counts(part, cnt);
  // connectivity through the forest-query adapter, the way a t8code build would get it
  T8gpuForestQuery* query = t8gpu_synth_query_create(mesh, 0, 1);
  HostMeshArrays    m     = t8gpu::hip::host_mesh_arrays_from_query(*query);
  t8gpu_synth_query_destroy(query);
  if (m.num_local_elements != cnt[0] || m.num_local_faces != cnt[2] || m.num_local_boundary_faces != cnt[3]) return 4;
  std::vector<int32_t> lev(cnt[0] + cnt[1]);
  std::vector<double>  cen(3 * (cnt[0] + cnt[1]));
  t8gpu_synth_part_elements(part, lev.data(), m.volumes.data(), cen.data());
  m.mesh_dim = mdim;
  m.levels.assign(lev.begin(), lev.begin() + cnt[0]);
  m.centres.assign(cen.begin(), cen.begin() + 3 * cnt[0]);
  std::vector<double> ic(5 * (cnt[0] + cnt[1]));
  t8gpu_synth_part_kh_ic(part, 1, ic.data(), cnt[0] + cnt[1]);
  const float_type delta_t = static_cast<float_type>(0.1 * std::pow(0.5, t8gpu_synth_mesh_finest_level(mesh)));

  Solver a(m, ic), b(m, ic), c(m, ic);
  hip::PlainFusedPlan<float_type> plan(m);
  hip::Reducer                    reduce;
  const double mass0 = reduce.integral<float_type>(m.num_local_elements, c.mesh.get_own_variable(c.next, Rho), c.mesh.get_own_volume());
  T8GPU_TIMER_START(three_variants);
  for (int i = 0; i < steps; i++) {
    a.iterate_user_kernels(delta_t);
    b.iterate_cabi_compat(delta_t);
    c.iterate_fused(plan, delta_t);
  }
  T8GPU_CUDA_CHECK_ERROR(hipDeviceSynchronize());
  T8GPU_TIMER_STOP(three_variants);
  // compute_integral / compute_timestep of the reference solver (solver.cu:190-229) on the device
  const double mass1 = reduce.integral<float_type>(m.num_local_elements, c.mesh.get_own_variable(c.next, Rho), c.mesh.get_own_volume());
  const double vmax  = reduce.max_speed<float_type>(m.num_local_faces + m.num_local_boundary_faces, c.speed);
  if (!(std::fabs(mass1 - mass0) <= (sizeof(float_type) == 8 ? 1e-12 : 1e-5) * std::fabs(mass0)) || !(vmax > 0.5 && vmax < 10.0)) {
    std::fprintf(stderr, "mass %.17g -> %.17g, max speed %g\n", mass0, mass1, vmax);
    return 5;
  }

  // the multi-rank constructor path with a one-rank communicator and no peers, and all steps in one driver call:
  // must reproduce variant C bit for bit
  {
    hip::Communicator                comm(hip::Communicator::unique_id(), 0, 1);
    hip::HostHaloArrays              no_peers;
    hip::PlainFusedPlan<float_type>  plan1(m, 3, 256, 512, &no_peers, &comm);
    Solver                           d(m, ic);
    d.iterate_fused_steps(plan1, delta_t, steps);
    T8GPU_CUDA_CHECK_ERROR(hipDeviceSynchronize());
    if (d.download() != c.download()) {
      std::fprintf(stderr, "iterate_fused with n_steps = %d differs from %d single-step calls\n", steps, steps);
      return 6;
    }
  }
  std::FILE* f = std::fopen(argv[7], "wb");
  if (!f) return 3;
  const int32_t header[3] = {m.num_local_elements, static_cast<int32_t>(sizeof(float_type)), steps};
  std::fwrite(header, sizeof(header), 1, f);
  for (Solver* s : {&a, &b, &c}) {
    auto v = s->download();
    std::fwrite(v.data(), sizeof(float_type), v.size(), f);
  }
  std::fclose(f);
  // the example's output step (examples/compressible_euler/solver.cu:231-262): density + momentum to VTK
  if (argc > 8) c.save_conserved_variables_to_vtk(argv[8]);
  t8gpu_synth_part_destroy(part);
  t8gpu_synth_mesh_destroy(mesh);
  return 0;
}
