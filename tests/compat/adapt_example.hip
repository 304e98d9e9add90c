// tests/compat/adapt_example.hip -- the adaptive part of the reference's main loop in C++ against the backend headers:
// CompressibleEulerSolver::adapt (examples/compressible_euler/solver.cu:243-262) = estimate_gradient + criteria kernels,
// MeshManager::adapt (mesh_manager.inl:196-330) = forest adapt + adapt_variables_and_volume + new connectivity,
// then iterate() on the new mesh. Self-checking: prints "adapt_example OK" and returns 0.
#include <t8gpu/backend/hip_fast.h>
#include <t8gpu/mesh/mesh_manager.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

using namespace t8gpu;

enum VariableList { Rho, Rho_v1, Rho_v2, Rho_v3, Rho_e, nb_variables };
enum StepList { Step0, Step1, Step2, Step3, Fluxes, nb_steps };
using float_type = variable_traits<VariableList>::float_type;
using Manager    = MeshManager<VariableList, StepList, 3>;   // the reference's class name (mesh_manager.h:232)

// MeshManager::initialize_variables with a callable of the reference's signature (solver.cu:17-72); the element
// handle of the synthetic provider carries the centre where the reference asks t8_forest_element_centroid for it.
// Returns the largest deviation from the provider's own evaluation of the same initial condition.
static double set_initial_state(Manager& mm) {
  mm.initialize_variables([](MemoryAccessorOwn<VariableList>& accessor, t8_forest_t /*forest*/, t8_locidx_t /*tree_idx*/,
                             t8_element_t const* element, t8_locidx_t e_idx) {
    auto [rho, rho_v1, rho_v2, rho_v3, rho_e] = accessor.get(Rho, Rho_v1, Rho_v2, Rho_v3, Rho_e);
    double const* c     = synthetic_element(element).centre;
    const double  sigma = 0.05 / std::sqrt(2.0), gamma = 1.4, y = c[1];
    const bool    in    = std::fabs(y - 0.5) < 0.25;
    const double  a = (y - 0.75) / (2 * sigma), b = (y - 0.25) / (2 * sigma), r = in ? 2.0 : 1.0;
    const double  u1 = in ? -0.5 : 0.5, u2 = r * (0.1 * std::sin(4.0 * M_PI * (c[0] - 0.5)) * (std::exp(-a * a) + std::exp(-b * b)));
    rho[e_idx]    = static_cast<float_type>(r);
    rho_v1[e_idx] = static_cast<float_type>(u1);
    rho_v2[e_idx] = static_cast<float_type>(u2);
    rho_v3[e_idx] = float_type(0);
    rho_e[e_idx]  = static_cast<float_type>(2.5 / (gamma - 1.0) + 0.5 * (u1 * u1 + u2 * u2) / r);
  });
  void*               part = t8gpu_synth_part_create(mm.forest(), 0, 1, 0, 3);
  const size_t        n    = mm.get_num_local_elements();
  std::vector<double> ic(5 * n);
  t8gpu_synth_part_kh_ic(part, 1, ic.data(), n);
  t8gpu_synth_part_destroy(part);
  double worst = 0;
  for (int v = 0; v < 5; v++) {
    std::vector<float_type> got(n);
    T8GPU_CUDA_CHECK_ERROR(hipMemcpy(got.data(), mm.get_own_variable(Step0, static_cast<VariableList>(v)), sizeof(float_type) * n, hipMemcpyDeviceToHost));
    for (size_t i = 0; i < n; i++) worst = std::max(worst, std::fabs(double(got[i]) - double(static_cast<float_type>(ic[v * n + i]))));
  }
  return worst;
}

// overloads pick the entry point of the build's float_type
static int estimate_gradient(int nf, const int32_t* fn, const float* rho, float* g) { return t8gpu_hip_estimate_gradient_f32(nf, fn, nullptr, rho, g, nullptr); }
static int estimate_gradient(int nf, const int32_t* fn, const double* rho, double* g) { return t8gpu_hip_estimate_gradient_f64(nf, fn, nullptr, rho, g, nullptr); }
static int refinement_criteria(int n, const float* g, const float* v, float* c) { return t8gpu_hip_refinement_criteria_f32(n, g, v, c, nullptr); }
static int refinement_criteria(int n, const double* g, const double* v, double* c) { return t8gpu_hip_refinement_criteria_f64(n, g, v, c, nullptr); }

static std::vector<float_type> criteria(Manager& mm, StepList step) {   // solver.cu:243-262 on the C-ABI
  const int n = mm.get_num_local_elements(), nf = mm.get_num_local_faces();
  float_type *grad = nullptr, *crit = nullptr;
  T8GPU_CUDA_CHECK_ERROR(hipMalloc(&grad, sizeof(float_type) * n));
  T8GPU_CUDA_CHECK_ERROR(hipMalloc(&crit, sizeof(float_type) * n));
  T8GPU_CUDA_CHECK_ERROR(hipMemset(grad, 0, sizeof(float_type) * n));
  auto conn = mm.get_connectivity_information();
  T8GPU_HIP_CHECK_ABI(estimate_gradient(nf, conn.face_neighbors(), mm.get_own_variable(step, Rho), grad));
  T8GPU_HIP_CHECK_ABI(refinement_criteria(n, grad, mm.get_own_volume(), crit));
  std::vector<float_type> host(n);
  T8GPU_CUDA_CHECK_ERROR(hipMemcpy(host.data(), crit, sizeof(float_type) * n, hipMemcpyDeviceToHost));
  (void)hipFree(grad);
  (void)hipFree(crit);
  return host;
}

int main() {
  const int min_level = 4, max_level = 7;
  Manager mm(t8gpu_synth_mesh_create(2, 5, 5, 0.0, 1.0, 1), min_level, max_level);   // 2D, uniform level 5, periodic
  StepList next = Step0, prev = Step3;
  static_assert(Manager::nb_variables == 5 && Manager::nb_steps == 5 && Manager::min_level == 1 && Manager::max_level == 4);
  const double ic_dev = set_initial_state(mm);
  if (!(ic_dev <= 1e-6)) {
    std::printf("adapt_example FAILED: initialize_variables deviates from the provider's initial condition by %.3g\n", ic_dev);
    return 1;
  }
  hip::Reducer reduce;
  const double mass0 = reduce.integral<float_type>(mm.get_num_local_elements(), mm.get_own_variable(next, Rho), mm.get_own_volume());
  float_type*  speed = nullptr;
  int          sizes[4];
  for (int cycle = 0; cycle < 3; cycle++) {
    sizes[cycle] = mm.get_num_local_elements();
    // the reference's sequence (solver.cu:243-262, main.cu:30-36): adapt, partition, compute_connectivity_information
    const std::vector<float_type>   c = criteria(mm, next);
    thrust::host_vector<float_type> crit(c.begin(), c.end());
    mm.adapt(crit, next);                                               // refine the shear layers, coarsen the rest
    mm.partition(next);
    mm.compute_connectivity_information();
    const double mass = reduce.integral<float_type>(mm.get_num_local_elements(), mm.get_own_variable(next, Rho), mm.get_own_volume());
    if (!(std::fabs(mass - mass0) <= 1e-5 * std::fabs(mass0))) {
      std::printf("adapt_example FAILED: mass %.17g -> %.17g in cycle %d\n", mass0, mass, cycle);
      return 1;
    }
    // a few steps on the new mesh (new tile plan from the manager's host arrays, as after compute_connectivity_information)
    hip::PlainFusedPlan<float_type> plan(mm.host_arrays());
    (void)hipFree(speed);
    T8GPU_CUDA_CHECK_ERROR(hipMalloc(&speed, sizeof(float_type) * (mm.get_num_local_faces() + mm.get_num_local_boundary_faces() + 1)));
    const int   finest  = t8gpu_synth_mesh_finest_level(mm.forest());
    const float_type dt = float_type(0.1 * std::pow(0.5, finest));
    for (int s = 0; s < 1; s++)   // the planes of the other steps are scratch after adapt(): zero them as the reference's RK does
      for (int st = 0; st < nb_steps; st++)
        if (st != next)
          for (int v = 0; v < 5; v++) mm.set_variable(static_cast<StepList>(st), static_cast<VariableList>(v), std::vector<float_type>(mm.get_num_local_elements(), 0));
    for (int it = 0; it < 4; it++) {
      std::swap(next, prev);
      hip::iterate_fused(mm, plan, prev, next, dt, speed);
    }
    T8GPU_CUDA_CHECK_ERROR(hipDeviceSynchronize());
  }
  sizes[3] = mm.get_num_local_elements();
  const double mass = reduce.integral<float_type>(mm.get_num_local_elements(), mm.get_own_variable(next, Rho), mm.get_own_volume());
  const double vmax = reduce.max_speed<float_type>(mm.get_num_local_faces(), speed);
  (void)hipFree(speed);
  std::printf("elements per cycle: %d %d %d %d, mass drift %.3g, max speed %.3g\n", sizes[0], sizes[1], sizes[2], sizes[3],
              std::fabs(mass - mass0) / std::fabs(mass0), vmax);
  if (!(sizes[1] != sizes[0]) || !(std::fabs(mass - mass0) <= 1e-5 * std::fabs(mass0)) || !(vmax > 0.5 && vmax < 10)) {
    std::printf("adapt_example FAILED\n");
    return 1;
  }
  if (const char* prefix = std::getenv("T8GPU_TEST_VTK_PREFIX")) mm.save_variable_to_vtk(next, Rho, prefix);
  std::printf("adapt_example OK\n");
  return 0;
}
