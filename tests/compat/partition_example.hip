// tests/compat/partition_example.hip -- the reference's adaptive main loop (examples/compressible_euler/main.cu:30-36:
// adapt, partition, compute_connectivity_information, then steps) in C++ on SEVERAL RANKS, checked against one rank.
// MeshManager::adapt / partition on N > 1 (t8gpu/mesh/mesh_manager.inl:196-330, 626-723) go through a t8gpu::Transport; here
// every rank is a host thread of this process with the loopback transport of tests/compat/loopback_transport.h (one GPU, no
// second RCCL rank available), the product uses t8gpu::RcclTransport over the same interface. Steps are the fused stage kernels
// with the ghost layer refreshed before every stage, so the k-rank run must equal the single-rank run BIT FOR BIT, state and
// element counts, after every cycle. Self-checking: prints "partition_example OK" and returns 0.
#include <t8gpu/backend/hip_fast.h>
#include <t8gpu/mesh/mesh_manager.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "loopback_transport.h"

using namespace t8gpu;

enum VariableList { Rho, Rho_v1, Rho_v2, Rho_v3, Rho_e, nb_variables };
enum StepList { Step0, Step1, Step2, Step3, Fluxes, nb_steps };
using float_type = variable_traits<VariableList>::float_type;
using Manager    = MeshManager<VariableList, StepList, 3>;

static void set_initial_state(Manager& mm) {   // the 2D Kelvin-Helmholtz state of examples/subgrid/solver.inl:84-103 at the cell centres
  mm.initialize_variables([](MemoryAccessorOwn<VariableList>& accessor, t8_forest_t, t8_locidx_t, t8_element_t const* element, t8_locidx_t e_idx) {
    auto [rho, rho_v1, rho_v2, rho_v3, rho_e] = accessor.get(Rho, Rho_v1, Rho_v2, Rho_v3, Rho_e);
    double const* c     = synthetic_element(element).centre;
    const double  sigma = 0.05 / std::sqrt(2.0), gamma = 1.4, y = c[1];
    const bool    in    = std::fabs(y - 0.5) < 0.25;
    const double  a = (y - 0.75) / (2 * sigma), b = (y - 0.25) / (2 * sigma), r = in ? 2.0 : 1.0;
    const double  u1 = in ? -0.5 : 0.5, u2 = r * (0.1 * std::sin(4.0 * M_PI * (c[0] - 0.5)) * (std::exp(-a * a) + std::exp(-b * b)));
    rho[e_idx] = static_cast<float_type>(r); rho_v1[e_idx] = static_cast<float_type>(u1); rho_v2[e_idx] = static_cast<float_type>(u2);
    rho_v3[e_idx] = float_type(0);
    rho_e[e_idx]  = static_cast<float_type>(2.5 / (gamma - 1.0) + 0.5 * (u1 * u1 + u2 * u2) / r);
  });
}

// overloads pick the entry point of the build's float_type
static int estimate_gradient(int nf, const int32_t* fn, const float* rho, float* g) { return t8gpu_hip_estimate_gradient_f32(nf, fn, nullptr, rho, g, nullptr); }
static int estimate_gradient(int nf, const int32_t* fn, const double* rho, double* g) { return t8gpu_hip_estimate_gradient_f64(nf, fn, nullptr, rho, g, nullptr); }
static int refinement_criteria(int n, const float* g, const float* v, float* c) { return t8gpu_hip_refinement_criteria_f32(n, g, v, c, nullptr); }
static int refinement_criteria(int n, const double* g, const double* v, double* c) { return t8gpu_hip_refinement_criteria_f64(n, g, v, c, nullptr); }
static int fused_stage(int stage, const T8gpuPlainPlan* plan, T8gpuVars_f32 p, T8gpuVars_f32 m, T8gpuVars_f32 o, const float* vol, float dt, float* speed) {
  return t8gpu_hip_plain_fused_stage_f32(T8GPU_FLUX_KEPES, stage, plan, 0, plan->ntiles, p, m, o, vol, dt, speed, nullptr);
}
static int fused_stage(int stage, const T8gpuPlainPlan* plan, T8gpuVars_f64 p, T8gpuVars_f64 m, T8gpuVars_f64 o, const double* vol, double dt, double* speed) {
  return t8gpu_hip_plain_fused_stage_f64(T8GPU_FLUX_KEPES, stage, plan, 0, plan->ntiles, p, m, o, vol, dt, speed, nullptr);
}

static std::vector<float_type> criteria(Manager& mm, StepList step) {   // solver.cu:243-262 on the C-ABI; reads ghost densities
  mm.refresh_ghost_layer(step);
  const int n = mm.get_num_local_elements(), g = mm.get_num_ghost_elements(), nf = mm.get_num_local_faces();
  float_type *grad = nullptr, *crit = nullptr;
  T8GPU_CUDA_CHECK_ERROR(hipMalloc(&grad, sizeof(float_type) * (n + g + 1)));
  T8GPU_CUDA_CHECK_ERROR(hipMalloc(&crit, sizeof(float_type) * (n + 1)));
  T8GPU_CUDA_CHECK_ERROR(hipMemset(grad, 0, sizeof(float_type) * (n + g + 1)));
  auto conn = mm.get_connectivity_information();
  T8GPU_HIP_CHECK_ABI(estimate_gradient(nf, conn.face_neighbors(), mm.get_own_variable(step, Rho), grad));
  T8GPU_HIP_CHECK_ABI(refinement_criteria(n, grad, mm.get_own_volume(), crit));
  std::vector<float_type> host(static_cast<size_t>(n));
  T8GPU_CUDA_CHECK_ERROR(hipMemcpy(host.data(), crit, sizeof(float_type) * n, hipMemcpyDeviceToHost));
  (void)hipFree(grad);
  (void)hipFree(crit);
  return host;
}

// one SSP-RK3 step: the fused stage kernel over the rank's whole plan, ghost slots of the stage's source refreshed first
static void step_once(Manager& mm, hip::PlainFusedPlan<float_type> const& plan, StepList prev, StepList next, float_type dt, float_type* speed) {
  const StepList src[3] = {prev, Step1, Step2}, dst[3] = {Step1, Step2, next};
  for (int k = 0; k < 3; k++) {
    mm.refresh_ghost_layer(src[k]);
    auto vars = [&](StepList s) {
      hip::vars_t<float_type> v;
      for (int q = 0; q < 5; q++) v.p[q] = mm.get_own_variable(s, static_cast<VariableList>(q));
      return v;
    };
    T8GPU_HIP_CHECK_ABI(fused_stage(k + 1, &plan.view(), vars(prev), vars(src[k]), vars(dst[k]), mm.get_own_volume(), dt, k == 2 ? speed : nullptr));
  }
  T8GPU_CUDA_CHECK_ERROR(hipDeviceSynchronize());
}

struct Result {
  std::vector<int>        counts;       // global element count after every cycle
  std::vector<float_type> state;        // this rank's final state [5][N]
  int64_t                 first = 0;
  int                     n     = 0;
};

#define TRACE(what) do { if (std::getenv("T8GPU_TEST_TRACE")) std::fprintf(stderr, "[rank %d/%d] %s\n", rank, nranks, what); } while (0)
static void run_rank(void* forest, int rank, int nranks, Transport* transport, Result* out) {
  // (min_level = the initial level: elements are refined and refined families coarsened again, never the initial families --
  //  a family cut by a rank boundary is not coarsened (t8gpu_synth_mesh_unmark_split_families, as t8code leaves it), which
  //  would make the k-rank forest differ from the single-rank one for reasons that are not this test's business)
  const int min_level = 5, max_level = 7, cycles = 3;
  TRACE("construct");
  Manager   mm(forest, min_level, max_level, sc_MPI_Comm{rank, nranks});
  mm.set_transport(transport);
  StepList next = Step0, prev = Step3;
  TRACE("initial state");
  set_initial_state(mm);
  for (int cycle = 0; cycle < cycles; cycle++) {
    TRACE("criteria");
    const std::vector<float_type>   c = criteria(mm, next);
    thrust::host_vector<float_type> crit(c.begin(), c.end());
    TRACE("adapt");
    mm.adapt(crit, next);
    TRACE("partition");
    mm.partition(next);
    TRACE("connectivity");
    mm.compute_connectivity_information();
    TRACE("plan + steps");
    out->counts.push_back(static_cast<int>(t8gpu_synth_mesh_num_elements(mm.forest())));
    hip::PlainFusedPlan<float_type> plan(mm.host_arrays());
    float_type* speed = nullptr;
    T8GPU_CUDA_CHECK_ERROR(hipMalloc(&speed, sizeof(float_type) * (mm.get_num_local_faces() + mm.get_num_local_boundary_faces() + 1)));
    const size_t tot = static_cast<size_t>(mm.get_num_local_elements()) + mm.get_num_ghost_elements();
    for (int st = 0; st < nb_steps; st++)   // the other steps' planes are scratch after adapt()
      if (st != next)
        for (int v = 0; v < 5; v++) mm.set_variable(static_cast<StepList>(st), static_cast<VariableList>(v), std::vector<float_type>(tot, 0));
    const float_type dt = float_type(0.1 * std::pow(0.5, t8gpu_synth_mesh_finest_level(mm.forest())));
    for (int it = 0; it < 3; it++) {
      std::swap(next, prev);
      step_once(mm, plan, prev, next, dt, speed);
    }
    (void)hipFree(speed);
    TRACE("cycle done");
  }
  TRACE("read back");
  out->n     = mm.get_num_local_elements();
  out->first = mm.host_arrays().first_global_element;
  out->state.resize(5 * static_cast<size_t>(out->n));
  for (int v = 0; v < 5; v++)
    T8GPU_CUDA_CHECK_ERROR(hipMemcpy(out->state.data() + static_cast<size_t>(v) * out->n, mm.get_own_variable(next, static_cast<VariableList>(v)),
                                     sizeof(float_type) * out->n, hipMemcpyDeviceToHost));
}

int main() {
  std::setvbuf(stdout, nullptr, _IONBF, 0);
  auto forest = [] { return t8gpu_synth_mesh_create(2, 5, 5, 0.0, 1.0, 1); };   // 2D, uniform level 5, periodic: every rank its own handle
  Result one;
  run_rank(forest(), 0, 1, nullptr, &one);
  {   // the product transport on one rank (no communicator needed: every run stays on the rank): same bits again
    RcclTransport rccl(nullptr, 0, 1);
    Result        again;
    run_rank(forest(), 0, 1, &rccl, &again);
    if (again.counts != one.counts || again.state != one.state) {
      std::printf("partition_example FAILED: one rank through RcclTransport differs from one rank without a transport\n");
      return 1;
    }
  }
  for (int nranks : {2, 3}) {
    t8gpu_test::LoopbackHub                     hub(nranks);
    std::vector<t8gpu_test::LoopbackTransport>  tr;
    std::vector<Result>                         res(static_cast<size_t>(nranks));
    for (int r = 0; r < nranks; r++) tr.emplace_back(hub, r);
    std::vector<std::thread> th;
    for (int r = 0; r < nranks; r++) th.emplace_back([&, r] { run_rank(forest(), r, nranks, &tr[static_cast<size_t>(r)], &res[static_cast<size_t>(r)]); });
    for (auto& t : th) t.join();
    int total = 0;
    for (auto const& x : res) total += x.n;
    if (res[0].counts != one.counts || total != one.n) {
      std::printf("partition_example FAILED on %d ranks: element counts differ (%d vs %d)\n", nranks, total, one.n);
      return 1;
    }
    int lo = one.n, hi = 0;
    for (auto const& x : res) {
      lo = std::min(lo, x.n);
      hi = std::max(hi, x.n);
      for (int v = 0; v < 5; v++)
        if (std::memcmp(x.state.data() + static_cast<size_t>(v) * x.n, one.state.data() + static_cast<size_t>(v) * one.n + x.first, sizeof(float_type) * x.n) != 0) {
          std::printf("partition_example FAILED on %d ranks: variable %d of the rank that starts at element %lld differs from the single-rank run\n", nranks, v,
                      static_cast<long long>(x.first));
          return 1;
        }
    }
    if (hi - lo > 1) {
      std::printf("partition_example FAILED on %d ranks: shares of %d .. %d elements after partition()\n", nranks, lo, hi);
      return 1;
    }
    std::printf("%d ranks: elements per cycle %d %d %d, shares %d .. %d, state bitwise the single-rank run\n", nranks, res[0].counts[0], res[0].counts[1],
                res[0].counts[2], lo, hi);
  }
  if (!(one.counts[0] > 1024 && one.counts[2] != one.counts[0])) {
    std::printf("partition_example FAILED: the mesh did not change (%d %d %d)\n", one.counts[0], one.counts[1], one.counts[2]);
    return 1;
  }
  std::printf("partition_example OK\n");
  return 0;
}
