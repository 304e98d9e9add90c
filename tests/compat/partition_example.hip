// tests/compat/partition_example.hip -- the reference's adaptive main loop (examples/compressible_euler/main.cu:30-36:
// adapt, partition, compute_connectivity_information, then steps) in C++ on SEVERAL RANKS, checked against one rank.
// MeshManager::adapt / partition on N > 1 (t8gpu/mesh/mesh_manager.inl:196-330, 626-723) go through a t8gpu::Transport; here
// every rank is a host thread of this process with the loopback transport of tests/compat/loopback_transport.h (one GPU, no
// second RCCL rank available), the product uses t8gpu::RcclTransport over the same interface. Steps are the fused stage kernels
// with the ghost layer refreshed before every stage, so the k-rank run must equal the single-rank run BIT FOR BIT, state and
// element counts, after every cycle. Further scenarios (other thresholds, 4 and 5 ranks) are run for what holds even where the k-rank
// forest legitimately differs from the single-rank one (a family cut by a rank boundary is not coarsened): mass is conserved
// through every adapt / partition / step, shares stay balanced to one element, every ghost slot holds its owner's value after
// refresh_ghost_layer(). Self-checking: prints "partition_example OK" and returns 0.
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
  double                  mass0 = 0, mass1 = 0;   // this rank's integral of the density: initial state, final state
  // the ghost layer of the final density after refresh_ghost_layer(): values of the slots [N, N + G), and the lists that say whose they are
  std::vector<float_type> ghost_rho;
  HostHaloArrays          halo;
};

struct Scenario {
  double threshold;   // of the adapt callback (10.0 in the reference, mesh_manager.inl:125-162)
  int    min_level, max_level, cycles;
};

#define TRACE(what) do { if (std::getenv("T8GPU_TEST_TRACE")) std::fprintf(stderr, "[rank %d/%d] %s\n", rank, nranks, what); } while (0)
static void run_rank(void* forest, int rank, int nranks, Transport* transport, Result* out, Scenario sc = {10.0, 5, 7, 3}) {
  // (the default scenario: min_level = the initial level -- elements are refined and refined families coarsened again, never the
  //  initial families. A family cut by a rank boundary is not coarsened (t8gpu_synth_mesh_unmark_split_families, as t8code leaves
  //  it), which makes a k-rank forest differ from the single-rank one; the default scenario has no such family on 2 and 3 ranks)
  const int min_level = sc.min_level, max_level = sc.max_level, cycles = sc.cycles;
  TRACE("construct");
  Manager   mm(forest, min_level, max_level, sc_MPI_Comm{rank, nranks});
  mm.set_transport(transport);
  StepList next = Step0, prev = Step3;
  TRACE("initial state");
  set_initial_state(mm);
  hip::Reducer reduce;
  auto mass = [&](StepList st) { return reduce.integral<float_type>(static_cast<size_t>(mm.get_num_local_elements()), mm.get_own_variable(st, Rho), mm.get_own_volume()); };
  out->mass0 = mass(next);
  for (int cycle = 0; cycle < cycles; cycle++) {
    TRACE("criteria");
    const std::vector<float_type> c = criteria(mm, next);
    TRACE("adapt");
    mm.adapt(c, next, sc.threshold);
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
  out->mass1 = mass(next);
  mm.refresh_ghost_layer(next);
  out->halo = mm.host_halo();
  out->ghost_rho.resize(static_cast<size_t>(mm.get_num_ghost_elements()));
  if (!out->ghost_rho.empty())
    T8GPU_CUDA_CHECK_ERROR(hipMemcpy(out->ghost_rho.data(), mm.get_own_variable(next, Rho) + mm.get_num_local_elements(),
                                     sizeof(float_type) * out->ghost_rho.size(), hipMemcpyDeviceToHost));
  out->n     = mm.get_num_local_elements();
  out->first = mm.host_arrays().first_global_element;
  out->state.resize(5 * static_cast<size_t>(out->n));
  for (int v = 0; v < 5; v++)
    T8GPU_CUDA_CHECK_ERROR(hipMemcpy(out->state.data() + static_cast<size_t>(v) * out->n, mm.get_own_variable(next, static_cast<VariableList>(v)),
                                     sizeof(float_type) * out->n, hipMemcpyDeviceToHost));
}

// What holds for every partitioned run, whatever its forest: mass conserved from the initial state to the end (periodic domain,
// conservative scheme, conservative transfer), shares balanced to one element, every ghost slot = its owner's value.
static bool invariants(std::vector<Result> const& res, char const* what) {
  const int nranks = static_cast<int>(res.size());
  double    m0 = 0, m1 = 0;
  int       lo = res[0].n, hi = res[0].n;
  for (auto const& x : res) { m0 += x.mass0; m1 += x.mass1; lo = std::min(lo, x.n); hi = std::max(hi, x.n); }
  const double tol = sizeof(float_type) == 4 ? 2e-5 : 1e-11;
  if (!(std::fabs(m1 - m0) <= tol * std::fabs(m0)) || !(m0 > 0)) {
    std::printf("partition_example FAILED (%s, %d ranks): mass %.12g -> %.12g\n", what, nranks, m0, m1);
    return false;
  }
  if (hi - lo > 1) {
    std::printf("partition_example FAILED (%s, %d ranks): shares of %d .. %d elements after partition()\n", what, nranks, lo, hi);
    return false;
  }
  size_t checked = 0;
  for (int r = 0; r < nranks; r++) {
    HostHaloArrays const& h = res[static_cast<size_t>(r)].halo;
    for (size_t j = 0; j < h.peers.size(); j++) {
      Result const&         o  = res[static_cast<size_t>(h.peers[j])];
      HostHaloArrays const& oh = o.halo;
      size_t                jj = oh.peers.size();
      for (size_t k = 0; k < oh.peers.size(); k++)
        if (oh.peers[k] == r) jj = k;
      const int n = h.recv_off[j + 1] - h.recv_off[j];
      if (jj == oh.peers.size() || oh.send_off[jj + 1] - oh.send_off[jj] != n) {
        std::printf("partition_example FAILED (%s, %d ranks): the ghost lists of ranks %d and %d do not match\n", what, nranks, r, h.peers[j]);
        return false;
      }
      for (int i = 0; i < n; i++, checked++) {
        const float_type got = res[static_cast<size_t>(r)].ghost_rho[static_cast<size_t>(h.recv_off[j] + i)];
        const float_type own = o.state[static_cast<size_t>(oh.send_idx[static_cast<size_t>(oh.send_off[jj] + i)])];   // (variable 0 = density)
        if (std::memcmp(&got, &own, sizeof got) != 0) {
          std::printf("partition_example FAILED (%s, %d ranks): ghost %d of rank %d from rank %d is stale\n", what, nranks, h.recv_off[j] + i, r, h.peers[j]);
          return false;
        }
      }
    }
  }
  std::printf("%s, %d ranks: mass %.12g kept to %.1e, shares %d .. %d, %zu ghost values = their owners'\n", what, nranks, m0, std::fabs(m1 - m0) / m0, lo, hi, checked);
  return true;
}

static std::vector<Result> run_ranks(int nranks, void* (*forest)(), Scenario sc) {
  t8gpu_test::LoopbackHub                    hub(nranks);
  std::vector<t8gpu_test::LoopbackTransport> tr;
  std::vector<Result>                        res(static_cast<size_t>(nranks));
  for (int r = 0; r < nranks; r++) tr.emplace_back(hub, r);
  std::vector<std::thread> th;
  for (int r = 0; r < nranks; r++) th.emplace_back([&, r] { run_rank(forest(), r, nranks, &tr[static_cast<size_t>(r)], &res[static_cast<size_t>(r)], sc); });
  for (auto& t : th) t.join();
  return res;
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
    std::vector<Result> res = run_ranks(nranks, +forest, Scenario{10.0, 5, 7, 3});
    if (!invariants(res, "default scenario")) return 1;
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
  // further scenarios: lower thresholds (more refinement, families coarsened where the sheet has moved on), coarsening below the
  // initial level (min_level 4: families cut by a rank boundary stay), 4 and 5 ranks. The forests may differ from the single-rank
  // run's here, so only the invariants are demanded -- and that at least one scenario really has a differing forest.
  bool differed = false;
  for (Scenario sc : {Scenario{10.0, 4, 7, 3}, Scenario{4.0, 4, 6, 4}, Scenario{2.0, 5, 7, 3}}) {
    Result ref;
    run_rank(forest(), 0, 1, nullptr, &ref, sc);
    for (int nranks : {2, 3, 4, 5}) {
      char what[96];
      std::snprintf(what, sizeof what, "threshold %g levels %d-%d", sc.threshold, sc.min_level, sc.max_level);
      std::vector<Result> res = run_ranks(nranks, +forest, sc);
      if (!invariants(res, what)) return 1;
      differed = differed || res[0].counts != ref.counts;
    }
  }
  std::printf("a k-rank forest differed from the single-rank one in some scenario: %s\n", differed ? "yes" : "no");
  if (!(one.counts[0] > 1024 && one.counts[2] != one.counts[0])) {
    std::printf("partition_example FAILED: the mesh did not change (%d %d %d)\n", one.counts[0], one.counts[1], one.counts[2]);
    return 1;
  }
  std::printf("partition_example OK\n");
  return 0;
}
