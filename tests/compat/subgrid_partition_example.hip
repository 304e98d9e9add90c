// tests/compat/subgrid_partition_example.hip -- the Subgrid example's adaptive main loop (examples/subgrid/main.cu: adapt,
// partition, compute_connectivity_information, then steps) in C++ on SEVERAL RANKS, checked against one rank.
// SubgridMeshManager::adapt / partition on N > 1 (t8gpu/mesh/subgrid_mesh_manager.inl:428-558, 1217-1369) move whole
// Subgrid<4,4,4> blocks through a t8gpu::Transport; here every rank is a host thread of this process with the loopback transport
// of tests/compat/loopback_transport.h (see partition_example.hip). Steps are the fused block kernel with the ghost BLOCKS
// refreshed before every stage, so the k-rank run must equal the single-rank run BIT FOR BIT after every cycle.
// Other thresholds and 4 ranks are run for what holds even where the k-rank forest legitimately differs (a family cut by a rank
// boundary is not coarsened): mass conserved, shares balanced to one block, every ghost block = its owner's 64 values.
// Self-checking: prints "subgrid_partition_example OK" and returns 0.
#include <t8gpu/backend/hip_fast.h>
#include <t8gpu/mesh/subgrid_mesh_manager.h>

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
using Grid3      = Subgrid<4, 4, 4>;
using Manager    = SubgridMeshManager<VariableList, StepList, Grid3>;
constexpr size_t S = Grid3::size;

static int block_criteria(int n, const float* rho, const float* vol, float* c) { return t8gpu_hip_subgrid_refinement_criteria_f32(3, n, rho, vol, c, nullptr); }
static int block_criteria(int n, const double* rho, const double* vol, double* c) { return t8gpu_hip_subgrid_refinement_criteria_f64(3, n, rho, vol, c, nullptr); }
static int fused_stage(int stage, const T8gpuSubgridPlan* plan, T8gpuVars_f32 p, T8gpuVars_f32 m, T8gpuVars_f32 o, const float* vol, float dt) {
  return t8gpu_hip_subgrid_fused_stage_f32(T8GPU_FLUX_KEPES, stage, plan, 0, plan->num_elements, p, m, o, vol, dt, nullptr);
}
static int fused_stage(int stage, const T8gpuSubgridPlan* plan, T8gpuVars_f64 p, T8gpuVars_f64 m, T8gpuVars_f64 o, const double* vol, double dt) {
  return t8gpu_hip_subgrid_fused_stage_f64(T8GPU_FLUX_KEPES, stage, plan, 0, plan->num_elements, p, m, o, vol, dt, nullptr);
}

static float_type* plane(Manager& mm, StepList s, int v) { return static_cast<float_type*>(mm.get_own_variable(s, static_cast<VariableList>(v))); }

static void set_initial_state(Manager& mm, int rank, int nranks) {   // Kelvin-Helmholtz (examples/subgrid/solver.inl:35-56,84-103), every subcell
  void*        part = t8gpu_synth_part_create(mm.forest(), rank, nranks, 1, 3);
  const size_t n    = (static_cast<size_t>(mm.get_num_local_elements()) + mm.get_num_ghost_elements()) * S;
  std::vector<double> ic(5 * n);
  t8gpu_synth_part_kh_ic(part, 4, ic.data(), n);
  t8gpu_synth_part_destroy(part);
  for (int st = 0; st < nb_steps; st++)
    for (int v = 0; v < 5; v++) mm.set_variable(static_cast<StepList>(st), static_cast<VariableList>(v), std::vector<float_type>(n, 0));
  for (int v = 0; v < 5; v++) mm.set_variable(Step0, static_cast<VariableList>(v), std::vector<float_type>(ic.begin() + v * n, ic.begin() + (v + 1) * n));
}

static std::vector<float_type> criteria(Manager& mm, StepList step) {   // solver.inl:268-287 on the C-ABI: block-local, no ghost values read
  const int   n    = mm.get_num_local_elements();
  float_type* crit = nullptr;
  T8GPU_CUDA_CHECK_ERROR(hipMalloc(&crit, sizeof(float_type) * (n + 1)));
  T8GPU_HIP_CHECK_ABI(block_criteria(n, plane(mm, step, Rho), mm.get_own_volume(), crit));
  std::vector<float_type> host(static_cast<size_t>(n));
  T8GPU_CUDA_CHECK_ERROR(hipMemcpy(host.data(), crit, sizeof(float_type) * n, hipMemcpyDeviceToHost));
  (void)hipFree(crit);
  return host;
}

static void step_once(Manager& mm, hip::SubgridFusedPlan<float_type> const& plan, StepList prev, StepList next, float_type dt) {
  const StepList src[3] = {prev, Step1, Step2}, dst[3] = {Step1, Step2, next};
  for (int k = 0; k < 3; k++) {
    mm.refresh_ghost_layer(src[k]);
    auto vars = [&](StepList s) {
      hip::vars_t<float_type> v;
      for (int q = 0; q < 5; q++) v.p[q] = plane(mm, s, q);
      return v;
    };
    T8GPU_HIP_CHECK_ABI(fused_stage(k + 1, &plan.view(), vars(prev), vars(src[k]), vars(dst[k]), mm.get_own_volume(), dt));
  }
  T8GPU_CUDA_CHECK_ERROR(hipDeviceSynchronize());
}

// Threshold of the adapt callback. The reference's 0.02 (subgrid_mesh_manager.inl:428) leaves this small mesh alone for two
// cycles; 0.002 gives 64 -> 400 -> 2528 -> 2304 blocks: refinement of an adapted mesh on unequal shares AND a coarsening cycle.
// (A family cut by a rank boundary is not coarsened -- t8gpu_synth_mesh_unmark_split_families, as t8code leaves it -- so a
//  k-rank forest may legitimately differ from the single-rank one: 0.02 and 0.01 do on 3 ranks, by 7 blocks. This scenario has no
//  such family, which is what makes the bitwise comparison possible.)
static double threshold() {
  char const* e = std::getenv("T8GPU_TEST_THRESHOLD");
  return e ? std::atof(e) : 0.002;
}

struct Result {
  std::vector<int>        counts;   // global block count after every cycle
  std::vector<float_type> state;    // this rank's final state [5][N * 64]
  std::vector<float_type> volume;   // [N]
  int64_t                 first = 0;
  int                     n     = 0;
  double                  mass0 = 0, mass1 = 0;   // this rank's integral of the density: initial state, final state
  std::vector<float_type> ghost_rho;              // the ghost BLOCKS of the final density after refresh_ghost_layer(): [G * 64]
  HostHaloArrays          halo;
};

static void run_rank(void* forest, int rank, int nranks, Transport* transport, Result* out, double thr = threshold()) {
  // (min_level = the initial level, as in partition_example.hip: families cut by a rank boundary are not coarsened)
  const int min_level = 2, max_level = 4, cycles = 4;
  Manager   mm(forest, min_level, max_level, sc_MPI_Comm{rank, nranks});
  mm.set_transport(transport);
  StepList next = Step0, prev = Step3;
  set_initial_state(mm, rank, nranks);
  hip::Reducer reduce;
  auto mass = [&](StepList st) { return reduce.integral<float_type>(static_cast<size_t>(mm.get_num_local_elements()) * S, plane(mm, st, Rho), mm.get_own_volume(), static_cast<int>(S)); };
  out->mass0 = mass(next);
  for (int cycle = 0; cycle < cycles; cycle++) {
    const std::vector<float_type>   c = criteria(mm, next);
    mm.adapt(c, next, thr);
    mm.partition(next);
    mm.compute_connectivity_information();
    out->counts.push_back(static_cast<int>(t8gpu_synth_mesh_num_elements(mm.forest())));
    hip::SubgridFusedPlan<float_type> plan(mm.host_arrays());
    const size_t tot = (static_cast<size_t>(mm.get_num_local_elements()) + mm.get_num_ghost_elements()) * S;
    for (int st = 0; st < nb_steps; st++)   // the other steps' planes are scratch after adapt()
      if (st != next)
        for (int v = 0; v < 5; v++) mm.set_variable(static_cast<StepList>(st), static_cast<VariableList>(v), std::vector<float_type>(tot, 0));
    const float_type dt = float_type(0.1 * std::pow(0.5, t8gpu_synth_mesh_finest_level(mm.forest()) + 2));
    for (int it = 0; it < 2; it++) {
      std::swap(next, prev);
      step_once(mm, plan, prev, next, dt);
    }
  }
  out->mass1 = mass(next);
  mm.refresh_ghost_layer(next);
  out->halo = mm.host_halo();
  out->ghost_rho.resize(static_cast<size_t>(mm.get_num_ghost_elements()) * S);
  if (!out->ghost_rho.empty())
    T8GPU_CUDA_CHECK_ERROR(hipMemcpy(out->ghost_rho.data(), plane(mm, next, Rho) + static_cast<size_t>(mm.get_num_local_elements()) * S,
                                     sizeof(float_type) * out->ghost_rho.size(), hipMemcpyDeviceToHost));
  out->n     = mm.get_num_local_elements();
  out->first = mm.host_arrays().first_global_element;
  out->state.resize(5 * static_cast<size_t>(out->n) * S);
  out->volume.resize(static_cast<size_t>(out->n));
  for (int v = 0; v < 5; v++)
    T8GPU_CUDA_CHECK_ERROR(hipMemcpy(out->state.data() + static_cast<size_t>(v) * out->n * S, plane(mm, next, v), sizeof(float_type) * out->n * S, hipMemcpyDeviceToHost));
  T8GPU_CUDA_CHECK_ERROR(hipMemcpy(out->volume.data(), mm.get_own_volume(), sizeof(float_type) * out->n, hipMemcpyDeviceToHost));
}

// What holds for every partitioned run, whatever its forest (see partition_example.hip)
static bool invariants(std::vector<Result> const& res, char const* what) {
  const int nranks = static_cast<int>(res.size());
  double    m0 = 0, m1 = 0;
  int       lo = res[0].n, hi = res[0].n;
  for (auto const& x : res) { m0 += x.mass0; m1 += x.mass1; lo = std::min(lo, x.n); hi = std::max(hi, x.n); }
  const double tol = sizeof(float_type) == 4 ? 2e-5 : 1e-11;
  if (!(std::fabs(m1 - m0) <= tol * std::fabs(m0)) || !(m0 > 0)) {
    std::printf("subgrid_partition_example FAILED (%s, %d ranks): mass %.12g -> %.12g\n", what, nranks, m0, m1);
    return false;
  }
  if (hi - lo > 1) {
    std::printf("subgrid_partition_example FAILED (%s, %d ranks): shares of %d .. %d blocks after partition()\n", what, nranks, lo, hi);
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
        std::printf("subgrid_partition_example FAILED (%s, %d ranks): the ghost lists of ranks %d and %d do not match\n", what, nranks, r, h.peers[j]);
        return false;
      }
      for (int i = 0; i < n; i++, checked++)
        if (std::memcmp(&res[static_cast<size_t>(r)].ghost_rho[static_cast<size_t>(h.recv_off[j] + i) * S],
                        &o.state[static_cast<size_t>(oh.send_idx[static_cast<size_t>(oh.send_off[jj] + i)]) * S], sizeof(float_type) * S) != 0) {
          std::printf("subgrid_partition_example FAILED (%s, %d ranks): ghost block %d of rank %d from rank %d is stale\n", what, nranks, h.recv_off[j] + i, r,
                      h.peers[j]);
          return false;
        }
    }
  }
  std::printf("%s, %d ranks: mass %.12g kept to %.1e, shares %d .. %d, %zu ghost blocks = their owners'\n", what, nranks, m0, std::fabs(m1 - m0) / m0, lo, hi, checked);
  return true;
}

static std::vector<Result> run_ranks(int nranks, void* (*forest)(), double thr) {
  t8gpu_test::LoopbackHub                    hub(nranks);
  std::vector<t8gpu_test::LoopbackTransport> tr;
  std::vector<Result>                        res(static_cast<size_t>(nranks));
  for (int r = 0; r < nranks; r++) tr.emplace_back(hub, r);
  std::vector<std::thread> th;
  for (int r = 0; r < nranks; r++) th.emplace_back([&, r] { run_rank(forest(), r, nranks, &tr[static_cast<size_t>(r)], &res[static_cast<size_t>(r)], thr); });
  for (auto& t : th) t.join();
  return res;
}

int main() {
  std::setvbuf(stdout, nullptr, _IONBF, 0);
  auto forest = [] { return t8gpu_synth_mesh_create(3, 2, 2, 0.0, 1.0, 1); };   // 3D, uniform level 2 (64 blocks), periodic
  Result one;
  run_rank(forest(), 0, 1, nullptr, &one);
  for (int nranks : {2, 3}) {
    std::vector<Result> res = run_ranks(nranks, +forest, threshold());
    if (!invariants(res, "default threshold")) return 1;
    int total = 0;
    for (auto const& x : res) total += x.n;
    if (res[0].counts != one.counts || total != one.n) {
      std::printf("subgrid_partition_example FAILED on %d ranks: block counts differ (%d vs %d)\n", nranks, total, one.n);
      return 1;
    }
    int lo = one.n, hi = 0;
    for (auto const& x : res) {
      lo = std::min(lo, x.n);
      hi = std::max(hi, x.n);
      if (std::memcmp(x.volume.data(), one.volume.data() + x.first, sizeof(float_type) * x.n) != 0) {
        std::printf("subgrid_partition_example FAILED on %d ranks: volumes of the rank that starts at block %lld\n", nranks, static_cast<long long>(x.first));
        return 1;
      }
      for (int v = 0; v < 5; v++)
        if (std::memcmp(x.state.data() + static_cast<size_t>(v) * x.n * S, one.state.data() + (static_cast<size_t>(v) * one.n + x.first) * S,
                        sizeof(float_type) * x.n * S) != 0) {
          std::printf("subgrid_partition_example FAILED on %d ranks: variable %d of the rank that starts at block %lld differs from the single-rank run\n",
                      nranks, v, static_cast<long long>(x.first));
          return 1;
        }
    }
    if (hi - lo > 1) {
      std::printf("subgrid_partition_example FAILED on %d ranks: shares of %d .. %d blocks after partition()\n", nranks, lo, hi);
      return 1;
    }
    std::printf("%d ranks: blocks per cycle %d %d %d %d, shares %d .. %d, state bitwise the single-rank run\n", nranks, res[0].counts[0], res[0].counts[1],
                res[0].counts[2], res[0].counts[3], lo, hi);
  }
  // the reference's threshold and half of it: on 3 ranks a family is cut by a rank boundary and stays (the forests differ by 7
  // blocks from the single-rank run's), so only the invariants are demanded there
  bool differed = false;
  if (!std::getenv("T8GPU_TEST_THRESHOLD"))
    for (double thr : {0.02, 0.01}) {
      Result ref;
      run_rank(forest(), 0, 1, nullptr, &ref, thr);
      for (int nranks : {2, 3, 4}) {
        char what[64];
        std::snprintf(what, sizeof what, "threshold %g", thr);
        std::vector<Result> res = run_ranks(nranks, +forest, thr);
        if (!invariants(res, what)) return 1;
        differed = differed || res[0].counts != ref.counts;
      }
    }
  std::printf("a k-rank forest differed from the single-rank one in some scenario: %s\n", differed ? "yes" : "no");
  if (!(one.counts[2] != 64 && one.counts[3] != one.counts[2])) {   // (the last cycle adapts an ADAPTED mesh on unequal shares)
    std::printf("subgrid_partition_example FAILED: the mesh did not change (%d %d %d %d)\n", one.counts[0], one.counts[1], one.counts[2], one.counts[3]);
    return 1;
  }
  std::printf("subgrid_partition_example OK\n");
  return 0;
}
