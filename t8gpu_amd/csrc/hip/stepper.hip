// stepper.hip -- native driver of one SSP-RK3 step: iterate() without any host language in the loop.
//
// Mirrors CompressibleEulerSolver::iterate (examples/compressible_euler/solver.cu:75-175) for the fused
// tier: 3 x [ghost exchange || interior tiles -> ghost-reading tiles]. Where the reference brackets
// every kernel with cudaDeviceSynchronize + MPI_Barrier (5 pairs per step) this driver only enqueues:
// ordering is carried by two HIP streams and two events, the ghost exchange is one RCCL group of
// ncclSend/ncclRecv per neighbour rank over xGMI, and the host returns immediately.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

#include "ranges.hpp"
#include "t8gpu_hip.h"

namespace {

inline int nccl_code(ncclResult_t r) { return r == ncclSuccess ? 0 : 10000 + static_cast<int>(r); }

#define T8_HIP_TRY(expr)                                  \
  do {                                                    \
    hipError_t e__ = (expr);                              \
    if (e__ != hipSuccess) return static_cast<int>(e__);  \
  } while (0)
#define T8_TRY(expr)            \
  do {                          \
    int c__ = (expr);           \
    if (c__ != 0) return c__;   \
  } while (0)

// Host cost of the enqueue by category (T8GPU_STEPPER_PROFILE=1; printed when the stepper is destroyed): where the time of a
// multi-rank stage goes on the host -- kernel launches, the RCCL group, event records / waits.
struct HostProfile {
  bool   on = std::getenv("T8GPU_STEPPER_PROFILE") != nullptr;
  double ns[4] = {0, 0, 0, 0};
  long   calls[4] = {0, 0, 0, 0};
};
HostProfile& host_profile() {
  static HostProfile p;
  return p;
}
struct HostTimer {   // cat: 0 kernel launch, 1 RCCL group, 2 event record, 3 stream wait
  int cat;
  std::chrono::steady_clock::time_point t0;
  explicit HostTimer(int c) : cat(c) {
    if (host_profile().on) t0 = std::chrono::steady_clock::now();
  }
  ~HostTimer() {
    if (!host_profile().on) return;
    static std::mutex m;   // (two host threads enqueue: stepper lanes)
    std::lock_guard<std::mutex> lk(m);
    host_profile().ns[cat] += std::chrono::duration<double, std::nano>(std::chrono::steady_clock::now() - t0).count();
    host_profile().calls[cat]++;
  }
};

struct Stepper {
  T8gpuPlainPlan   plan{};      // plain elements: units = tiles of the plan
  T8gpuSubgridPlan splan{};     // Subgrid blocks: units = blocks in block_order position order
  bool             subgrid = false;
  bool           has_halo = false;
  T8gpuHalo      halo{};
  std::vector<int32_t> peers, send_off, recv_off;
  hipStream_t    comm_stream = nullptr;   // pack, RCCL, unpack, class A tiles
  hipStream_t    near_stream = nullptr;   // class B tiles
  hipEvent_t     ev_state = nullptr, ev_ghost = nullptr;   // step entry / last boundary launch
  hipEvent_t     ev_interior = nullptr;                  // last class B launch
  hipEvent_t     ev_deep = nullptr;                      // last class C launch
  int            timing = 0;        // 0 = off, n = time the stage kernels of every n-th step
  bool           sample = false;    // the step being enqueued is one of those
  std::vector<hipEvent_t> pool;   // start/stop pairs of the stage-kernel launches
  size_t         used = 0;
  int            stages_timed = 0;
  // hipGraph replay of a whole iterate_steps() call (t8gpu_hip_plain_stepper_graph): the enqueue sequence is
  // captured once per distinct argument set on an internal origin stream and replayed with one hipGraphLaunch
  int             graph_mode = 0;       // 0 off, 1 on
  hipStream_t     graph_stream = nullptr;
  hipEvent_t      ev_graph_in = nullptr, ev_graph_out = nullptr;
  struct GraphEntry {               // one executable per argument set; a step loop alternates between two (the roles of
    hipGraphExec_t exec = nullptr;  // prev / next swap with every odd n_steps), so the cache holds a few
    unsigned char  key[96] = {0};
    unsigned long  used = 0;
  };
  GraphEntry      graph_cache[4];
  unsigned long   graph_clock = 0;
  int             graph_captures = 0, graph_replays = 0;
  bool            capturing = false;    // iterate() is being recorded into a graph
  int             capture_variant = 0;  // diagnostics (T8GPU_GRAPH_VARIANT): 2 = global capture mode, 3 = thread-local capture mode,
                                        // 5 = exchange chain on a FORKED stream of the capture (the layout that crashes)
  void*           scratch = nullptr;
  std::vector<hipEvent_t> capture_events;   // one event per (stage, role) of a captured call (see stage_event)

  // ---- two-lane driver (iterate_lanes; round 4) ----------------------------------------------------------------------
  static constexpr int kRing = 4;
  struct Lane {
    hipStream_t        stream = nullptr;
    hipEvent_t         ring[kRing] = {nullptr, nullptr, nullptr, nullptr};   // the record of stage g uses ring[g % kRing]
    std::atomic<long>  recorded{0};     // stages of the current call whose record has been issued
    std::vector<hipEvent_t> pool;       // timing events of this lane
    size_t             used = 0;
  };
  Lane            deep_lane, comm_lane;     // C tiles | RCCL, A tiles, B tiles
  bool            zero_copy = false;        // ghost window: A tiles read the receive buffer and fill the send buffer themselves
  T8gpuPlainPlan  plan_a{};                 // `plan` + the ghost window (launches of the A class)
  void*           d_send_map = nullptr;
  void*           d_send_list = nullptr;
  std::atomic<bool> lane_abort{false};
  // the comm lane's host thread (enqueues its lane while the caller's thread enqueues the deep lane)
  std::thread               worker;
  std::mutex                mu;
  std::condition_variable   cv;
  std::function<int()>      job;            // guarded by mu
  std::atomic<int>          job_state{0};   // 0 idle, 1 posted, 2 done
  int                       job_rc = 0;
  bool                      worker_exit = false;
  int                       device = 0;
  bool                      threads = true;
  double                    host_ns = 0;    // host time inside iterate_lanes (both threads overlap: wall time of the call)
  long                      host_steps = 0;
};

// Events that order the three streams. Direct enqueue: one event per role, re-recorded every stage. Inside a capture
// every (stage, role) gets an event of its own: a captured wait refers to the record node it follows, and the HIP
// runtime of this stack crashes when an event that captured waits already refer to is recorded again in the same
// capture (segmentation fault at the end of the capture, with or without RCCL in it: tests/test_gpu_graph.py).
enum EventRole { kDeep = 0, kGhost = 1, kInterior = 2, kJoin = 3 };
static int stage_event(Stepper* S, int g, EventRole role, hipEvent_t* out) {
  if (!S->capturing) {
    *out = role == kDeep ? S->ev_deep : (role == kGhost ? S->ev_ghost : S->ev_interior);
    return 0;
  }
  const size_t idx = static_cast<size_t>(g) * 4 + static_cast<size_t>(role);
  while (S->capture_events.size() <= idx) {
    hipEvent_t e;
    T8_HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    S->capture_events.push_back(e);
  }
  *out = S->capture_events[idx];
  return 0;
}

template <class T>
ncclDataType_t nccl_type();
template <>
ncclDataType_t nccl_type<float>() { return ncclFloat; }
template <>
ncclDataType_t nccl_type<double>() { return ncclDouble; }

template <class T, class V>
int exchange(const T8gpuHalo& h, const int32_t* peers, const int32_t* send_off, const int32_t* recv_off, V state,
             hipStream_t s) {
  if (h.n_peers <= 0) return 0;
  T*       sb = static_cast<T*>(h.sendbuf);
  T*       rb = static_cast<T*>(h.recvbuf);
  const int cells = h.cells_per_element < 1 ? 1 : h.cells_per_element;
  HostTimer htp(0);
  if constexpr (sizeof(T) == 4) {
    T8_TRY(t8gpu_hip_halo_pack_f32(h.n_send, cells, h.send_idx, state, sb, s));
  } else {
    T8_TRY(t8gpu_hip_halo_pack_f64(h.n_send, cells, h.send_idx, state, sb, s));
  }
  // T8GPU_EXP_NO_RCCL: experiment builds only (build.py variants, tests/test_gpu_graph.py): the RCCL group is left out, to
  // tell a capture that fails because of RCCL from one that fails because of the three-stream fork / join. The ghosts are
  // then stale and the results wrong, so the product library has no run-time switch for it.
  ncclComm_t comm = static_cast<ncclComm_t>(h.comm);
#ifndef T8GPU_EXP_NO_RCCL
  {
  HostTimer ht(1);
  T8_TRY(nccl_code(ncclGroupStart()));
  for (int j = 0; j < h.n_peers; j++) {
    const size_t w  = 5 * static_cast<size_t>(cells);   // values per element on the wire
    const size_t rc = w * static_cast<size_t>(recv_off[j + 1] - recv_off[j]);
    const size_t sc = w * static_cast<size_t>(send_off[j + 1] - send_off[j]);
    if (rc) T8_TRY(nccl_code(ncclRecv(rb + w * static_cast<size_t>(recv_off[j]), rc, nccl_type<T>(), peers[j], comm, s)));
    if (sc) T8_TRY(nccl_code(ncclSend(sb + w * static_cast<size_t>(send_off[j]), sc, nccl_type<T>(), peers[j], comm, s)));
  }
  T8_TRY(nccl_code(ncclGroupEnd()));
  }
#else
  (void)comm;
  (void)peers; (void)send_off; (void)recv_off; (void)rb;
#endif
  if constexpr (sizeof(T) == 4) {
    T8_TRY(t8gpu_hip_halo_unpack_f32(h.num_ghosts, h.num_elements, cells, rb, state, s));
  } else {
    T8_TRY(t8gpu_hip_halo_unpack_f64(h.num_ghosts, h.num_elements, cells, rb, state, s));
  }
  return 0;
}

template <class V, class T>
V step_vars(T* planes, size_t stride, int step) {
  V v;
  for (int k = 0; k < 5; k++) v.p[k] = planes + (static_cast<size_t>(step) * 5 + k) * stride;
  return v;
}

int tick(Stepper* S, hipStream_t s) {
  if (!S->timing || !S->sample) return 0;
  if (S->used == S->pool.size()) {
    hipEvent_t e;
    T8_HIP_TRY(hipEventCreate(&e));
    S->pool.push_back(e);
  }
  T8_HIP_TRY(hipEventRecord(S->pool[S->used++], s));
  return 0;
}


// ---- the two-lane driver (round 4) ------------------------------------------------------------------------------------
// What the three-stream pipeline above costs at 8 ranks (profiles/r04_halo_*.md): the exchange chain
// pack -> RCCL -> unpack -> A tiles is the critical path of a stage (the RCCL kernel alone takes ~50 us beside the tile
// kernels), and the host needs ~55 us to enqueue a stage (13 HIP calls + the RCCL group) that the GPU finishes in ~50.
// This driver shortens both:
//   * GHOST WINDOW (plain 2D / tile plans): the A tiles read their ghosts straight from the receive buffer of the
//     exchange, in its wire format, and write the elements a peer mirrors into the send buffer in their RK epilogue
//     (T8gpuPlainPlan::ghost_buf / send_map, fused_common.hpp). No pack kernel (except for the first stage of a call, whose
//     source state comes from outside) and no unpack kernel: the chain is RCCL -> A tiles.
//   * TWO LANES, each depending only on the OTHER lane's PREVIOUS stage:
//       deep lane (caller's stream)  : [A_(g-1)] -> I_g          I = all interior tiles [0, n_interior): one launch
//       comm lane                    : RCCL_g -> [I_(g-1)] -> A_g
//     I_g reads what I- and A-tiles of stage g-1 wrote; A_g reads the ghosts of stage g and what A- and I-tiles of stage
//     g-1 wrote; RCCL_g sends what A_(g-1) put into the send buffer. Both waits refer to work that ended most of a stage
//     earlier (the chain RCCL + A, ~35 us, is shorter than I, ~45 us, beside it), so neither lane ever stalls on the
//     other and the long launches run back to back. With the three-stream pipeline the deep tiles of stage g + 1 waited
//     for B_g, which sat behind the whole exchange chain, across queues: ~10 us of idle GPU per stage.
//     The RCCL kernel needs 264 VGPRs per lane (rcclGenericKernel<1, false> of RCCL 2.26.6 for gfx950): beside tile kernels
//     that hold 3 x 160 of a SIMD's 512 it is not dispatched before the tile launch has handed out its last workgroup.
//     Here it is queued a whole stage ahead of its deadline and slips in at the drain between two interior launches.
//     (A plan WITH a deep / near-boundary split of its interior tiles -- Subgrid plans, plain plans built without flag 32 -- runs
//     C_g on the deep lane behind [B_(g-1)] and B_g on the comm lane behind A_g and [C_(g-1)].)
//   * The comm lane is enqueued by a HOST THREAD of its own while the caller's thread enqueues the deep lane; the threads
//     meet through two counters (a wait on stage g's event may only be issued once the other thread has recorded it).
// Plans the ghost window does not cover (Subgrid blocks, 3D patch tiles) run the same two lanes with the pack / unpack
// kernels around the group.
template <class T>
int exchange_wire(const T8gpuHalo& h, const int32_t* peers, const int32_t* send_off, const int32_t* recv_off, hipStream_t s) {
#ifndef T8GPU_EXP_NO_RCCL
  HostTimer   ht(1);
  T*          sb    = static_cast<T*>(h.sendbuf);
  T*          rb    = static_cast<T*>(h.recvbuf);
  const int   cells = h.cells_per_element < 1 ? 1 : h.cells_per_element;
  ncclComm_t  comm  = static_cast<ncclComm_t>(h.comm);
  T8_TRY(nccl_code(ncclGroupStart()));
  for (int j = 0; j < h.n_peers; j++) {
    const size_t w  = 5 * static_cast<size_t>(cells);   // values per element on the wire
    const size_t rc = w * static_cast<size_t>(recv_off[j + 1] - recv_off[j]);
    const size_t sc = w * static_cast<size_t>(send_off[j + 1] - send_off[j]);
    if (rc) T8_TRY(nccl_code(ncclRecv(rb + w * static_cast<size_t>(recv_off[j]), rc, nccl_type<T>(), peers[j], comm, s)));
    if (sc) T8_TRY(nccl_code(ncclSend(sb + w * static_cast<size_t>(send_off[j]), sc, nccl_type<T>(), peers[j], comm, s)));
  }
  T8_TRY(nccl_code(ncclGroupEnd()));
#else
  (void)h; (void)peers; (void)send_off; (void)recv_off; (void)s;
#endif
  return 0;
}

inline void cpu_relax() {
#if defined(__x86_64__) || defined(__i386__)
  __builtin_ia32_pause();
#endif
}

int lane_tick(Stepper* S, Stepper::Lane& L) {
  if (!S->timing || !S->sample) return 0;
  if (L.used == L.pool.size()) {
    hipEvent_t e;
    T8_HIP_TRY(hipEventCreate(&e));
    L.pool.push_back(e);
  }
  T8_HIP_TRY(hipEventRecord(L.pool[L.used++], L.stream));
  return 0;
}

// wait until the other lane's host thread has RECORDED stage g (only then may a wait on that event be issued)
inline bool lane_wait_recorded(Stepper* S, Stepper::Lane& other, long g) {
  while (other.recorded.load(std::memory_order_acquire) <= g) {
    if (S->lane_abort.load(std::memory_order_relaxed)) return false;
    cpu_relax();
  }
  return true;
}

void worker_main(Stepper* S) {
  (void)hipSetDevice(S->device);
  std::unique_lock<std::mutex> lk(S->mu);
  for (;;) {
    // spin briefly for the next call (a step loop calls again within microseconds), then sleep
    lk.unlock();
    const auto t0 = std::chrono::steady_clock::now();
    while (S->job_state.load(std::memory_order_acquire) != 1 &&
           std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() < 300.0)
      cpu_relax();
    lk.lock();
    S->cv.wait(lk, [&] { return S->job_state.load(std::memory_order_acquire) == 1 || S->worker_exit; });
    if (S->worker_exit) return;
    std::function<int()> job = std::move(S->job);
    lk.unlock();
    const int rc = job();
    lk.lock();
    S->job_rc = rc;
    S->job_state.store(2, std::memory_order_release);
  }
}

template <class T, class V>
int iterate_lanes(Stepper* S, int kind, T* planes, size_t stride, const T* vol, int prev, int next, T dt, T* speed, int n_steps,
                  hipStream_t s) {
  if (n_steps <= 0) return 0;
  const auto host_t0 = std::chrono::steady_clock::now();
  const int  nt = S->subgrid ? S->splan.num_elements : S->plan.ntiles;
  const int  ni = S->subgrid ? S->splan.n_interior_blocks : S->plan.n_interior_tiles;
  const int  ndeep = S->subgrid ? S->splan.n_deep_blocks : S->plan.n_deep_tiles;
  // The plan decides: one interior class (n_deep = n_interior; t8gpu_plan_plain_create_ex flag 32, what partitioned meshes get)
  // -> the interior tiles are ONE launch per stage on the deep lane; a deep / near-boundary split in the plan (Subgrid plans,
  // plain plans built without the flag) -> C_g on the deep lane, B_g behind A_g on the comm lane. n_deep = 0 ("unknown"):
  // every interior tile counts as near-boundary.
  const int  nd = (ndeep > 0 && ndeep <= ni) ? ndeep : 0;
  const int  G  = 3 * n_steps;
  t8gpu_hip::Range whole("t8gpu.iterate_steps (exchange -> ghost-reading tiles || interior tiles, two lanes)");
  Stepper::Lane& DL = S->deep_lane;
  Stepper::Lane& XL = S->comm_lane;
  DL.stream = s;
  XL.stream = S->comm_stream;
  DL.recorded.store(0, std::memory_order_relaxed);
  XL.recorded.store(0, std::memory_order_relaxed);
  S->lane_abort.store(false, std::memory_order_relaxed);
  const int timing = S->timing;
  if (timing > 0) S->stages_timed += 3 * ((n_steps + timing - 1) / timing);

  // entry: the lanes see everything the caller queued on s
  T8_HIP_TRY(hipEventRecord(S->ev_state, s));
  T8_HIP_TRY(hipStreamWaitEvent(XL.stream, S->ev_state, 0));
  if (DL.stream != s) T8_HIP_TRY(hipStreamWaitEvent(DL.stream, S->ev_state, 0));

  struct StageArgs {
    int k;
    V   pv, sv, ov;
    T*  stage_speed;
  };
  auto stage_args = [=](int g) {
    StageArgs a;
    a.k          = g % 3;
    const int pr = (g / 3) % 2 == 0 ? prev : next, nx = (g / 3) % 2 == 0 ? next : prev;
    const int src = a.k == 0 ? pr : a.k, dst = a.k == 2 ? nx : a.k + 1;   // Step1 = 1, Step2 = 2 (solver.h:24-31)
    a.pv = step_vars<V>(planes, stride, pr);
    a.sv = step_vars<V>(planes, stride, src);
    a.ov = step_vars<V>(planes, stride, dst);
    a.stage_speed = a.k == 2 ? speed : nullptr;   // (speed estimates once per step: see iterate())
    return a;
  };
  auto launch = [=](Stepper::Lane& L, const T8gpuPlainPlan* plan, const StageArgs& a, int b, int n, bool sample) -> int {
    if (n <= 0) return 0;
    HostTimer ht(0);
    if (sample) T8_TRY(lane_tick(S, L));
    if constexpr (sizeof(T) == 4) {
      if (S->subgrid)
        T8_TRY(t8gpu_hip_subgrid_fused_stage_f32(kind, a.k + 1, &S->splan, b, n, a.pv, a.sv, a.ov, vol, dt, L.stream));
      else
        T8_TRY(t8gpu_hip_plain_fused_stage_f32(kind, a.k + 1, plan, b, n, a.pv, a.sv, a.ov, vol, dt, a.stage_speed, L.stream));
    } else {
      if (S->subgrid)
        T8_TRY(t8gpu_hip_subgrid_fused_stage_f64(kind, a.k + 1, &S->splan, b, n, a.pv, a.sv, a.ov, vol, dt, L.stream));
      else
        T8_TRY(t8gpu_hip_plain_fused_stage_f64(kind, a.k + 1, plan, b, n, a.pv, a.sv, a.ov, vol, dt, a.stage_speed, L.stream));
    }
    if (sample) T8_TRY(lane_tick(S, L));
    return 0;
  };
  // (S->sample is read by lane_tick: both threads sample the same steps, so it is set once here, "on", and the per-stage
  //  decision is the `sample` argument)
  S->sample = true;

  // ---- comm lane, stage g: RCCL_g -> A_g -> [C_(g-1)] -> B_g --------------------------------------------------------------
  auto comm_stage = [=, &DL, &XL](int g) -> int {
    const T8gpuHalo& h      = S->halo;
    const int        cells  = h.cells_per_element < 1 ? 1 : h.cells_per_element;
    const StageArgs  a      = stage_args(g);
    const bool       sample = timing > 0 && (g / 3) % timing == 0;
    if (!S->zero_copy || g == 0) {   // (ghost window: the A tiles of stage g-1 have filled the send buffer)
      HostTimer ht(0);
      if constexpr (sizeof(T) == 4)
        T8_TRY(t8gpu_hip_halo_pack_f32(h.n_send, cells, h.send_idx, a.sv, static_cast<T*>(h.sendbuf), XL.stream));
      else
        T8_TRY(t8gpu_hip_halo_pack_f64(h.n_send, cells, h.send_idx, a.sv, static_cast<T*>(h.sendbuf), XL.stream));
    }
    T8_TRY(exchange_wire<T>(h, S->peers.data(), S->send_off.data(), S->recv_off.data(), XL.stream));
    if (!S->zero_copy) {
      HostTimer ht(0);
      if constexpr (sizeof(T) == 4)
        T8_TRY(t8gpu_hip_halo_unpack_f32(h.num_ghosts, h.num_elements, cells, static_cast<const T*>(h.recvbuf), a.sv, XL.stream));
      else
        T8_TRY(t8gpu_hip_halo_unpack_f64(h.num_ghosts, h.num_elements, cells, static_cast<const T*>(h.recvbuf), a.sv, XL.stream));
    }
    auto wait_deep = [&]() -> int {   // the deep lane's stage g-1
      if (g == 0 || nd == 0) return 0;
      if (!lane_wait_recorded(S, DL, g - 1)) return 0;
      HostTimer ht(3);
      T8_HIP_TRY(hipStreamWaitEvent(XL.stream, DL.ring[(g - 1) % Stepper::kRing], 0));
      return 0;
    };
    if (nd == ni) T8_TRY(wait_deep());                                                   // A_g <- interior_(g-1)
    T8_TRY(launch(XL, S->zero_copy ? &S->plan_a : &S->plan, a, ni, nt - ni, sample));   // A_g
    if (nd < ni) {
      T8_TRY(wait_deep());                                                               // B_g <- C_(g-1)
      T8_TRY(launch(XL, &S->plan, a, nd, ni - nd, sample));                              // B_g
    }
    {
      HostTimer ht(2);
      T8_HIP_TRY(hipEventRecord(XL.ring[g % Stepper::kRing], XL.stream));
    }
    XL.recorded.store(g + 1, std::memory_order_release);
    return 0;
  };
  // ---- deep lane, stage g: [B_(g-1)] -> C_g --------------------------------------------------------------------------------
  auto deep_stage = [=, &DL, &XL](int g) -> int {
    const StageArgs a      = stage_args(g);
    const bool      sample = timing > 0 && (g / 3) % timing == 0;
    if (nd > 0) {
      if (g > 0) {
        if (!lane_wait_recorded(S, XL, g - 1)) return 0;
        HostTimer ht(3);
        T8_HIP_TRY(hipStreamWaitEvent(DL.stream, XL.ring[(g - 1) % Stepper::kRing], 0));
      }
      T8_TRY(launch(DL, &S->plan, a, 0, nd, sample));                                    // C_g
      HostTimer ht(2);
      T8_HIP_TRY(hipEventRecord(DL.ring[g % Stepper::kRing], DL.stream));
    }
    DL.recorded.store(g + 1, std::memory_order_release);
    return 0;
  };
  // (a lane that fails must not leave the other one's host thread waiting for its records)
  auto run_lane = [S, G](const std::function<int(int)>& stage) -> int {
    for (int g = 0; g < G; g++) {
      const int rc = stage(g);
      if (rc != 0) {
        S->lane_abort.store(true, std::memory_order_relaxed);
        return rc;
      }
      if (S->lane_abort.load(std::memory_order_relaxed)) return 0;
    }
    return 0;
  };

  int rc_comm = 0, rc_deep = 0;
  if (S->threads) {
    if (!S->worker.joinable()) {
      (void)hipGetDevice(&S->device);
      S->worker = std::thread(worker_main, S);
    }
    {
      std::lock_guard<std::mutex> lk(S->mu);
      S->job = [&]() { return run_lane(comm_stage); };
      S->job_state.store(1, std::memory_order_release);
    }
    S->cv.notify_one();
    rc_deep = run_lane(deep_stage);
    while (S->job_state.load(std::memory_order_acquire) != 2) cpu_relax();
    {
      std::lock_guard<std::mutex> lk(S->mu);
      rc_comm = S->job_rc;
      S->job_state.store(0, std::memory_order_release);
    }
  } else {
    // one host thread: stage by stage, the deep lane first (every wait for a record is then satisfied when it is reached)
    for (int g = 0; g < G && rc_deep == 0 && rc_comm == 0; g++) {
      rc_deep = deep_stage(g);
      if (rc_deep == 0) rc_comm = comm_stage(g);
    }
  }
  // exit: everything is ordered on s again
  T8_HIP_TRY(hipStreamWaitEvent(s, XL.ring[(G - 1) % Stepper::kRing], 0));
  if (DL.stream != s && nd > 0) T8_HIP_TRY(hipStreamWaitEvent(s, DL.ring[(G - 1) % Stepper::kRing], 0));
  S->host_ns += std::chrono::duration<double, std::nano>(std::chrono::steady_clock::now() - host_t0).count();
  S->host_steps += n_steps;
  return rc_deep != 0 ? rc_deep : rc_comm;
}

// n_steps SSP-RK3 steps; (prev, next) are the roles of the FIRST step, they swap from step to step
// (solver.cu:76). Multi-rank pipeline, per stage g (tile classes of tile_plan.cpp: C = deep interior,
// B = interior tiles that read an element owned by an A tile, A = tiles that read ghost slots):
//   caller's stream s : [B_(g-1)] -> C_g
//   near stream       : [C_(g-1), A_(g-1)] -> B_g
//   comm stream       : pack_g -> RCCL_g -> unpack_g -> [B_(g-1)] -> A_g
// C_g reads only what B/C tiles of stage g-1 wrote; pack_g reads only elements next to a cut face, which A
// tiles own (same stream, no wait); A_g reads ghosts, A- and B-owned elements; B_g reads all three classes
// of stage g-1 but nothing of stage g, so it runs beside C_g instead of behind it. The long launch (C) and
// the exchange chain therefore never wait for each other, and every bracketed dependency is normally
// satisfied long before it is reached. Stages two apart are ordered transitively (C_g > B_(g-1) > all of
// g-2, and so on), which is what the reuse of the four step buffers needs. The streams meet only at the
// entry and at the exit of the call.
// Subgrid blocks run through the same pipeline (SubgridCompressibleEulerSolver::iterate, examples/subgrid/solver.inl:
// 152-266): units are blocks in the plan's position order (deep interior, near-boundary, ghost-touching), a ghost
// block mirrors all 4^rank subcells, `vol` is the separate per-block volume array of SubgridMemoryManager.
template <class T, class V>
int iterate(Stepper* S, int kind, T* planes, size_t stride, const T* vol, int prev, int next, T dt, T* speed, int n_steps,
            hipStream_t s) {
  const int  nt = S->subgrid ? S->splan.num_elements : S->plan.ntiles;
  const int  ni = S->subgrid ? S->splan.n_interior_blocks : S->plan.n_interior_tiles;
  const int  ndeep = S->subgrid ? S->splan.n_deep_blocks : S->plan.n_deep_tiles;
  const int  nd = (ndeep > 0 && ndeep <= ni) ? ndeep : 0;
  const bool comm = S->has_halo && S->halo.n_peers > 0;
  // several ranks, direct enqueue: the two-lane driver above. This function keeps the single-rank loop, and the three-stream
  // pipeline for a hipGraph capture (whose fork / join shape the capture rules of this stack dictate) and for A/B
  // measurements (T8GPU_STEPPER=legacy).
  static const bool legacy = std::getenv("T8GPU_STEPPER") && std::strcmp(std::getenv("T8GPU_STEPPER"), "legacy") == 0;
  if (comm && !S->capturing && !legacy) return iterate_lanes<T, V>(S, kind, planes, stride, vol, prev, next, dt, speed, n_steps, s);
  t8gpu_hip::Range whole(comm ? "t8gpu.iterate_steps (exchange + 3 tile classes)" : "t8gpu.iterate_steps");
  static const char* const stage_name[3] = {"t8gpu.rk_stage1", "t8gpu.rk_stage2", "t8gpu.rk_stage3"};
  hipEvent_t last_ghost = nullptr, last_interior = nullptr;
  for (int g = 0; g < 3 * n_steps; g++) {
    const int k  = g % 3;
    t8gpu_hip::Range stage_range(stage_name[k]);
    S->sample    = S->timing > 0 && (g / 3) % S->timing == 0;
    if (S->sample) S->stages_timed++;
    const int pr = (g / 3) % 2 == 0 ? prev : next, nx = (g / 3) % 2 == 0 ? next : prev;
    const int src = k == 0 ? pr : k, dst = k == 2 ? nx : k + 1;   // Step1 = 1, Step2 = 2 (solver.h:24-31)
    const V   pv = step_vars<V>(planes, stride, pr), sv = step_vars<V>(planes, stride, src), ov = step_vars<V>(planes, stride, dst);
    // The per-face speed estimates are rewritten by every stage and read only between steps (compute_timestep uses
    // those "computed at the last step of the last timestepping", solver.h:88-91): only the third stage writes them
    // (same contents after every step, a tenth less HBM traffic per step).
    T* const stage_speed = k == 2 ? speed : nullptr;
    auto launch = [&](int b, int n, hipStream_t on) -> int {
      if (n <= 0) return 0;
      HostTimer ht(0);
      T8_TRY(tick(S, on));
      if constexpr (sizeof(T) == 4) {
        if (S->subgrid)
          T8_TRY(t8gpu_hip_subgrid_fused_stage_f32(kind, k + 1, &S->splan, b, n, pv, sv, ov, vol, dt, on));
        else
          T8_TRY(t8gpu_hip_plain_fused_stage_f32(kind, k + 1, &S->plan, b, n, pv, sv, ov, vol, dt, stage_speed, on));
      } else {
        if (S->subgrid)
          T8_TRY(t8gpu_hip_subgrid_fused_stage_f64(kind, k + 1, &S->splan, b, n, pv, sv, ov, vol, dt, on));
        else
          T8_TRY(t8gpu_hip_plain_fused_stage_f64(kind, k + 1, &S->plan, b, n, pv, sv, ov, vol, dt, stage_speed, on));
      }
      return tick(S, on);
    };
    if (!comm) {
      T8_TRY(launch(0, nt, s));
      continue;
    }
    // Roles of the three streams. Direct enqueue: the caller's stream carries the deep tiles (the long launch), the comm
    // stream the exchange chain. Inside a capture the ORIGIN stream of the capture must carry the exchange chain: an RCCL
    // group issued on a FORKED stream of a capture ends in a segmentation fault inside hipStreamEndCapture on this stack
    // (HIP 7.0.51831 / RCCL 2.26.6; relaxed, global and thread-local capture modes alike -- T8GPU_GRAPH_VARIANT=5 keeps
    // that layout for the opt-in diagnostic of tests/test_gpu_graph.py), issued on the origin stream it is captured and
    // replayed correctly. A graph only knows dependencies, so the replayed pipeline has the same shape either way.
    const bool        swap = S->capturing && S->capture_variant != 5;
    const hipStream_t sc = swap ? S->comm_stream : s;    // C_g: deep tiles
    const hipStream_t sx = swap ? s : S->comm_stream;    // pack_g -> RCCL_g -> unpack_g -> A_g
    const hipStream_t sb = S->near_stream;               // B_g
    if (g == 0) {  // entry: the other streams must see everything the caller queued on s
      if (S->capturing && S->scratch) T8_HIP_TRY(hipMemsetAsync(S->scratch, 0, 64, s));   // a first node for the fork event
      T8_HIP_TRY(hipEventRecord(S->ev_state, s));
      T8_HIP_TRY(hipStreamWaitEvent(S->comm_stream, S->ev_state, 0));
      T8_HIP_TRY(hipStreamWaitEvent(S->near_stream, S->ev_state, 0));
    }
    // every wait on an event of stage g-1 is issued before that event is re-recorded for stage g
    hipEvent_t deep_p = nullptr, ghost_p = nullptr, inter_p = nullptr, deep_c, ghost_c, inter_c;
    if (g > 0) {
      T8_TRY(stage_event(S, g - 1, kDeep, &deep_p));
      T8_TRY(stage_event(S, g - 1, kGhost, &ghost_p));
      T8_TRY(stage_event(S, g - 1, kInterior, &inter_p));
    }
    T8_TRY(stage_event(S, g, kDeep, &deep_c));
    T8_TRY(stage_event(S, g, kGhost, &ghost_c));
    T8_TRY(stage_event(S, g, kInterior, &inter_c));
    // (Measured and dropped in round 3: a TWO-class pipeline -- B and A tiles in one launch behind the unpack, 9 host calls
    //  per stage instead of 13. The host cost fell from 170 to 146 us per step, but C_g then has to wait for the previous
    //  stage's exchange chain and the step rose from 0.175 to 0.256 ms on rank 3 of the 8-way c4 split with an RCCL
    //  self-exchange: the third class IS what keeps the long launch from waiting. profiles/r03_halo_overhead.md)
    if (g > 0 && S->capturing) {
      // Inside a capture every dependency goes through the origin stream: it joins the other streams' previous stage,
      // records one event, and they fork from that (forked streams waiting on each other's events -- what the direct
      // enqueue below does -- crashes the end of the capture as well, with or without RCCL in it). Costs edges the
      // pipeline does not need: stage g starts when all of stage g-1 is done.
      hipEvent_t join;
      T8_TRY(stage_event(S, g, kJoin, &join));
      T8_HIP_TRY(hipStreamWaitEvent(s, inter_p, 0));
      T8_HIP_TRY(hipStreamWaitEvent(s, swap ? deep_p : ghost_p, 0));   // (the third class of g-1 ran on s itself)
      T8_HIP_TRY(hipEventRecord(join, s));
      T8_HIP_TRY(hipStreamWaitEvent(S->near_stream, join, 0));
      T8_HIP_TRY(hipStreamWaitEvent(S->comm_stream, join, 0));
    } else if (g > 0) {
      HostTimer ht(3);
      T8_HIP_TRY(hipStreamWaitEvent(sb, deep_p, 0));                       // B_g <- C_(g-1)
      T8_HIP_TRY(hipStreamWaitEvent(sb, ghost_p, 0));                      // B_g <- A_(g-1)
      T8_HIP_TRY(hipStreamWaitEvent(sc, inter_p, 0));                      // C_g <- B_(g-1)
    }
    T8_TRY(launch(0, nd, sc));                                             // C_g
    { HostTimer ht(2); T8_HIP_TRY(hipEventRecord(deep_c, sc)); }
    T8_TRY((exchange<T, V>(S->halo, S->peers.data(), S->send_off.data(), S->recv_off.data(), sv, sx)));
    if (g > 0 && !S->capturing) { HostTimer ht(3); T8_HIP_TRY(hipStreamWaitEvent(sx, inter_p, 0)); }               // A_g <- B_(g-1)
    T8_TRY(launch(ni, nt - ni, sx));                                       // A_g
    { HostTimer ht(2); T8_HIP_TRY(hipEventRecord(ghost_c, sx)); }
    T8_TRY(launch(nd, ni - nd, sb));                                       // B_g
    { HostTimer ht(2); T8_HIP_TRY(hipEventRecord(inter_c, sb)); }
    last_ghost    = swap ? deep_c : ghost_c;   // (what the exit below joins: the two streams that are not s)
    last_interior = inter_c;
  }
  if (comm && n_steps > 0) {  // exit: everything is ordered on s again
    T8_HIP_TRY(hipStreamWaitEvent(s, last_ghost, 0));
    T8_HIP_TRY(hipStreamWaitEvent(s, last_interior, 0));
  }
  return 0;
}

// iterate() through a hipGraph: capture the enqueue sequence once per argument set, then replay it. The capture runs
// on the stepper's own origin stream (the caller's stream may be the legacy default stream, which cannot capture);
// the comm and near streams join the capture through the events they wait on, and rejoin before it ends. Timing
// events are off in graph mode. With a halo the RCCL group is captured too (RCCL enqueues its kernels on the capturing
// stream); if the runtime refuses any part of the capture the error is returned and the caller falls back.
template <class T, class V>
int iterate_graph(Stepper* S, int kind, T* planes, size_t stride, const T* vol, int prev, int next, T dt, T* speed, int n_steps,
                  hipStream_t s) {
  if (!S->graph_mode || S->timing > 0 || n_steps <= 0) return iterate<T, V>(S, kind, planes, stride, vol, prev, next, dt, speed, n_steps, s);
  // Multi-rank stages have an RCCL group in the middle. Capturing it is OPT-IN (T8GPU_GRAPH_RCCL=1): a replayed RCCL group
  // has only ever run on one GPU exchanging with itself (tests/test_gpu_graph.py), never across xGMI, the capture has to
  // route every dependency through the origin stream (see iterate()), and hipGraphLaunch costs this stack more host time
  // than the two-lane direct enqueue (profiles/r04_halo_overhead.md). By default a stepper with peers enqueues directly.
  static const bool rccl_capture = std::getenv("T8GPU_GRAPH_RCCL") && std::getenv("T8GPU_GRAPH_RCCL")[0] == '1';
  if (S->has_halo && S->halo.n_peers > 0 && !rccl_capture) return iterate<T, V>(S, kind, planes, stride, vol, prev, next, dt, speed, n_steps, s);
  struct Key {
    int kind, prev, next, n_steps, tsize, subgrid;
    const void *planes, *vol, *speed;
    size_t stride;
    double dt;
  } key{kind, prev, next, n_steps, static_cast<int>(sizeof(T)), S->subgrid ? 1 : 0, planes, vol, speed, stride, static_cast<double>(dt)};
  static_assert(sizeof(Key) <= sizeof(S->graph_cache[0].key), "graph key");
  if (!S->graph_stream) {
    T8_HIP_TRY(hipStreamCreateWithFlags(&S->graph_stream, hipStreamNonBlocking));
    T8_HIP_TRY(hipEventCreateWithFlags(&S->ev_graph_in, hipEventDisableTiming));
    T8_HIP_TRY(hipEventCreateWithFlags(&S->ev_graph_out, hipEventDisableTiming));
  }
  // look the argument set up; on a miss the least recently used entry is replaced. (delta_t is part of the key: a
  // CFL-adaptive step size re-captures per value -- use the direct enqueue for such loops.)
  Stepper::GraphEntry* hit = nullptr;
  Stepper::GraphEntry* lru = &S->graph_cache[0];
  for (auto& ge : S->graph_cache) {
    if (ge.exec && std::memcmp(&key, ge.key, sizeof(Key)) == 0) hit = &ge;
    if (!ge.exec || (lru->exec && ge.used < lru->used)) lru = &ge;
  }
  if (!hit) {
    if (lru->exec) {
      (void)hipGraphExecDestroy(lru->exec);
      lru->exec = nullptr;
    }
    hipGraph_t g = nullptr;
    static const bool trace = std::getenv("T8GPU_DEBUG_GRAPH") != nullptr;   // progress marks on stderr (diagnostics)
#define T8_MARK(what) do { if (trace) { std::fprintf(stderr, "[t8gpu graph] %s\n", what); std::fflush(stderr); } } while (0)
    T8_MARK("begin capture");
    S->capture_variant = std::getenv("T8GPU_GRAPH_VARIANT") ? std::atoi(std::getenv("T8GPU_GRAPH_VARIANT")) : 0;
    if (!S->scratch) T8_HIP_TRY(hipMalloc(&S->scratch, 64));
    {   // every event a capture of this length needs exists before the capture starts
      struct Capturing {   // (reset on every way out of this block, early error returns included)
        Stepper* s;
        explicit Capturing(Stepper* p) : s(p) { s->capturing = true; }
        ~Capturing() { s->capturing = false; }
      } guard(S);
      hipEvent_t e;
      for (int g2 = 0; g2 < 3 * n_steps; g2++)
        for (int r = 0; r < 4; r++) T8_TRY(stage_event(S, g2, static_cast<EventRole>(r), &e));
    }
    const hipStreamCaptureMode cmode = S->capture_variant == 2 ? hipStreamCaptureModeGlobal
                                                                : (S->capture_variant == 3 ? hipStreamCaptureModeThreadLocal : hipStreamCaptureModeRelaxed);
    T8_HIP_TRY(hipStreamBeginCapture(S->graph_stream, cmode));
    S->capturing  = true;
    const int  rc = iterate<T, V>(S, kind, planes, stride, vol, prev, next, dt, speed, n_steps, S->graph_stream);
    S->capturing  = false;
    T8_MARK("enqueue recorded");
    hipError_t e  = hipStreamEndCapture(S->graph_stream, &g);
    T8_MARK("capture ended");
    if (rc != 0 || e != hipSuccess || !g) {
      if (g) (void)hipGraphDestroy(g);
      return rc != 0 ? rc : static_cast<int>(e != hipSuccess ? e : hipErrorStreamCaptureInvalidated);
    }
    e = hipGraphInstantiate(&lru->exec, g, nullptr, nullptr, 0);
    T8_MARK("instantiated");
    (void)hipGraphDestroy(g);
    if (e != hipSuccess) {
      lru->exec = nullptr;
      return static_cast<int>(e);
    }
    std::memset(lru->key, 0, sizeof(lru->key));
    std::memcpy(lru->key, &key, sizeof(Key));
    S->graph_captures++;
    hit = lru;
  }
  hit->used = ++S->graph_clock;
  T8_HIP_TRY(hipEventRecord(S->ev_graph_in, s));                         // the graph starts behind the caller's work ...
  T8_HIP_TRY(hipStreamWaitEvent(S->graph_stream, S->ev_graph_in, 0));
  T8_HIP_TRY(hipGraphLaunch(hit->exec, S->graph_stream));
  if (std::getenv("T8GPU_DEBUG_GRAPH")) { std::fprintf(stderr, "[t8gpu graph] launched\n"); std::fflush(stderr); }
  T8_HIP_TRY(hipEventRecord(S->ev_graph_out, S->graph_stream));
  T8_HIP_TRY(hipStreamWaitEvent(s, S->ev_graph_out, 0));                 // ... and the caller's stream continues behind it
  S->graph_replays++;
  return 0;
}


// ---- MeshManager::partition over RCCL ------------------------------------------------------------------------
// The device half of the reference's partition() (t8gpu/mesh/mesh_manager.inl:626-723, subgrid_mesh_manager.inl:1217-1369):
// there the new owner of an element PULLS its variables through CUDA-IPC pointers into the old owner's memory
// (partition_data<<<>>>); here the old owner SENDS. Elements move in contiguous runs of the space-filling curve, and a run is
// contiguous in every variable plane and in the volume array, so a run is six messages straight from the source planes into
// the destination planes -- no gather kernel, no staging buffer. One RCCL group for all runs of a call; a run whose peer is
// this rank is a device copy.
template <class T, class V>
int repartition(void* comm, int my_rank, int n_send, const int32_t* send_peer, const int32_t* send_first, const int32_t* send_count,
                int n_recv, const int32_t* recv_peer, const int32_t* recv_first, const int32_t* recv_count, V src, const T* src_vol, V dst,
                T* dst_vol, int cells, hipStream_t s) {
  if (cells < 1 || n_send < 0 || n_recv < 0) return static_cast<int>(hipErrorInvalidValue);
  const size_t w = static_cast<size_t>(cells);
  // runs that stay on this rank: the i-th local send pairs with the i-th local receive (both lists are in curve order)
  int js = 0;
  for (int jr = 0; jr < n_recv; jr++) {
    if (recv_peer[jr] != my_rank) continue;
    while (js < n_send && send_peer[js] != my_rank) js++;
    if (js >= n_send || send_count[js] != recv_count[jr]) return static_cast<int>(hipErrorInvalidValue);
    const size_t n = static_cast<size_t>(recv_count[jr]);
    for (int k = 0; k < 5; k++)
      T8_HIP_TRY(hipMemcpyAsync(dst.p[k] + w * recv_first[jr], src.p[k] + w * send_first[js], sizeof(T) * w * n, hipMemcpyDeviceToDevice, s));
    T8_HIP_TRY(hipMemcpyAsync(dst_vol + recv_first[jr], src_vol + send_first[js], sizeof(T) * n, hipMemcpyDeviceToDevice, s));
    js++;
  }
  bool remote = false;
  for (int j = 0; j < n_send; j++) remote = remote || send_peer[j] != my_rank;
  for (int j = 0; j < n_recv; j++) remote = remote || recv_peer[j] != my_rank;
  if (!remote) return 0;
  if (!comm) return static_cast<int>(hipErrorInvalidValue);
#ifndef T8GPU_EXP_NO_RCCL
  ncclComm_t c = static_cast<ncclComm_t>(comm);
  T8_TRY(nccl_code(ncclGroupStart()));
  for (int j = 0; j < n_recv; j++) {
    if (recv_peer[j] == my_rank || recv_count[j] <= 0) continue;
    const size_t n = static_cast<size_t>(recv_count[j]);
    for (int k = 0; k < 5; k++) T8_TRY(nccl_code(ncclRecv(dst.p[k] + w * recv_first[j], w * n, nccl_type<T>(), recv_peer[j], c, s)));
    T8_TRY(nccl_code(ncclRecv(dst_vol + recv_first[j], n, nccl_type<T>(), recv_peer[j], c, s)));
  }
  for (int j = 0; j < n_send; j++) {
    if (send_peer[j] == my_rank || send_count[j] <= 0) continue;
    const size_t n = static_cast<size_t>(send_count[j]);
    for (int k = 0; k < 5; k++) T8_TRY(nccl_code(ncclSend(src.p[k] + w * send_first[j], w * n, nccl_type<T>(), send_peer[j], c, s)));
    T8_TRY(nccl_code(ncclSend(src_vol + send_first[j], n, nccl_type<T>(), send_peer[j], c, s)));
  }
  T8_TRY(nccl_code(ncclGroupEnd()));
#endif
  return 0;
}
}  // namespace

extern "C" {

// roctx ranges for host code above the C-ABI (bench.py marks pre-warm / warm-up / timed repetitions with them)
int t8gpu_hip_range_push(const char* name) {
  if (!t8gpu_hip::roctx().on) return 0;
  t8gpu_hip::roctx().push(name ? name : "t8gpu");
  return 1;
}
int t8gpu_hip_range_pop(void) {
  if (!t8gpu_hip::roctx().on) return 0;
  t8gpu_hip::roctx().pop();
  return 1;
}

int t8gpu_hip_comm_unique_id(char* id128) {
  ncclUniqueId id;
  ncclResult_t r = ncclGetUniqueId(&id);
  if (r != ncclSuccess) return nccl_code(r);
  static_assert(sizeof(id) == 128, "ncclUniqueId layout");
  std::memcpy(id128, &id, 128);
  return 0;
}

int t8gpu_hip_comm_create(const char* id128, int rank, int nranks, void** comm) {
  {   // header / runtime skew (t8gpu_hip_runtime_versions): same major version, at least 2.7
    int v[4];
    (void)t8gpu_hip_runtime_versions(v);
    if (v[1] < 20700 || v[1] / 10000 != NCCL_VERSION_CODE / 10000 || v[3] / 10000000 != HIP_VERSION / 10000000) {
      std::fprintf(stderr, "[t8gpu] RCCL %d / HIP %d at run time do not match the headers this library was built with (RCCL %d, HIP %d)\n",
                   v[1], v[3], v[0], v[2]);
      return static_cast<int>(hipErrorNotSupported);
    }
  }
  ncclUniqueId id;
  std::memcpy(&id, id128, 128);
  ncclComm_t c = nullptr;
  ncclResult_t r = ncclCommInitRank(&c, nranks, id, rank);
  if (r != ncclSuccess) return nccl_code(r);
  *comm = c;
  return 0;
}

int t8gpu_hip_comm_destroy(void* comm) { return comm ? nccl_code(ncclCommDestroy(static_cast<ncclComm_t>(comm))) : 0; }
int t8gpu_hip_comm_abort(void* comm) { return comm ? nccl_code(ncclCommAbort(static_cast<ncclComm_t>(comm))) : 0; }

int t8gpu_hip_halo_exchange_f32(const T8gpuHalo* h, T8gpuVars_f32 state, void* stream) {
  if (!h) return static_cast<int>(hipErrorInvalidValue);
  return exchange<float, T8gpuVars_f32>(*h, h->peers, h->send_off, h->recv_off, state, static_cast<hipStream_t>(stream));
}
int t8gpu_hip_halo_exchange_f64(const T8gpuHalo* h, T8gpuVars_f64 state, void* stream) {
  if (!h) return static_cast<int>(hipErrorInvalidValue);
  return exchange<double, T8gpuVars_f64>(*h, h->peers, h->send_off, h->recv_off, state, static_cast<hipStream_t>(stream));
}

int t8gpu_hip_repartition_f32(void* comm, int my_rank, int n_send, const int32_t* send_peer, const int32_t* send_first,
                              const int32_t* send_count, int n_recv, const int32_t* recv_peer, const int32_t* recv_first,
                              const int32_t* recv_count, T8gpuVars_f32 src, const float* src_volume, T8gpuVars_f32 dst, float* dst_volume,
                              int cells_per_element, void* stream) {
  return repartition<float, T8gpuVars_f32>(comm, my_rank, n_send, send_peer, send_first, send_count, n_recv, recv_peer, recv_first, recv_count,
                                           src, src_volume, dst, dst_volume, cells_per_element, static_cast<hipStream_t>(stream));
}
int t8gpu_hip_repartition_f64(void* comm, int my_rank, int n_send, const int32_t* send_peer, const int32_t* send_first,
                              const int32_t* send_count, int n_recv, const int32_t* recv_peer, const int32_t* recv_first,
                              const int32_t* recv_count, T8gpuVars_f64 src, const double* src_volume, T8gpuVars_f64 dst, double* dst_volume,
                              int cells_per_element, void* stream) {
  return repartition<double, T8gpuVars_f64>(comm, my_rank, n_send, send_peer, send_first, send_count, n_recv, recv_peer, recv_first, recv_count,
                                            src, src_volume, dst, dst_volume, cells_per_element, static_cast<hipStream_t>(stream));
}

// Every rank's chunk of a distributed double array to every rank: mine[offsets[rank + 1] - offsets[rank]] -> all[offsets[nranks]]
// (device pointers; offsets on the host). The refinement criteria of a partitioned adapt travel this way (the forest is
// replicated, so every rank evaluates the adapt callback on the whole array).
int t8gpu_hip_comm_allgatherv_f64(void* comm, int rank, int nranks, const double* mine, double* all, const int64_t* offsets, void* stream) {
  if (!mine || !all || !offsets || rank < 0 || rank >= nranks) return static_cast<int>(hipErrorInvalidValue);
  hipStream_t  s = static_cast<hipStream_t>(stream);
  const size_t n = static_cast<size_t>(offsets[rank + 1] - offsets[rank]);
  if (n) T8_HIP_TRY(hipMemcpyAsync(all + offsets[rank], mine, sizeof(double) * n, hipMemcpyDeviceToDevice, s));
  if (nranks == 1) return 0;
  if (!comm) return static_cast<int>(hipErrorInvalidValue);
#ifndef T8GPU_EXP_NO_RCCL
  ncclComm_t c = static_cast<ncclComm_t>(comm);
  T8_TRY(nccl_code(ncclGroupStart()));
  for (int q = 0; q < nranks; q++) {
    if (q == rank) continue;
    const size_t m = static_cast<size_t>(offsets[q + 1] - offsets[q]);
    if (m) T8_TRY(nccl_code(ncclRecv(all + offsets[q], m, ncclDouble, q, c, s)));
    if (n) T8_TRY(nccl_code(ncclSend(mine, n, ncclDouble, q, c, s)));
  }
  T8_TRY(nccl_code(ncclGroupEnd()));
#endif
  return 0;
}

// Polls a stream until it is idle or `timeout_s` elapsed (1 = timed out). Lets a caller bound the
// first exchange on a new communicator and fall back (t8gpu_hip_comm_abort) instead of hanging.
int t8gpu_hip_stream_wait(void* stream, double timeout_s) {
  const auto t0 = std::chrono::steady_clock::now();
  for (;;) {
    hipError_t e = hipStreamQuery(static_cast<hipStream_t>(stream));
    if (e == hipSuccess) return 0;
    if (e != hipErrorNotReady) return static_cast<int>(e);
    if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > timeout_s) return 1;
    std::this_thread::sleep_for(std::chrono::milliseconds(1));
  }
}

static int stepper_halo_setup(Stepper* S, const T8gpuHalo* halo) {
  if (halo && halo->n_peers > 0) {
    S->has_halo = true;
    S->halo     = *halo;
    S->peers.assign(halo->peers, halo->peers + halo->n_peers);
    S->send_off.assign(halo->send_off, halo->send_off + halo->n_peers + 1);
    S->recv_off.assign(halo->recv_off, halo->recv_off + halo->n_peers + 1);
    // Normal priority on purpose: a high-priority comm stream was measured (rocprofv3 kernel trace, one rank
    // of the 8-way c4 split) to make everything slower -- tile kernels 49 -> 90-150 us, the 5 us pack / unpack
    // kernels up to 90 us, 50 us gaps -- the queue preempts the running tile waves instead of waiting for a slot.
    // (round 4, two-lane driver: the highest stream priority for the comm lane changes nothing -- 0.1521 / 0.1522 ms per step on
    //  rank 3 of the 8-way c4 split)
    hipError_t e = hipStreamCreateWithFlags(&S->comm_stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&S->ev_state, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&S->ev_ghost, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&S->ev_interior, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&S->ev_deep, hipEventDisableTiming);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&S->near_stream, hipStreamNonBlocking);
    // the lanes' events order two queues of ONE device: a device-scope release is all they need (the default is a
    // system-scope fence per record). T8GPU_EVENT_SCOPE=system keeps the default (measurements).
    const char*    scope = std::getenv("T8GPU_EVENT_SCOPE");
    const unsigned flags = hipEventDisableTiming | ((scope && scope[0] == 's') ? 0u : static_cast<unsigned>(hipEventReleaseToDevice));
    for (int i = 0; i < Stepper::kRing && e == hipSuccess; i++) {
      e = hipEventCreateWithFlags(&S->deep_lane.ring[i], flags);
      if (e == hipSuccess) e = hipEventCreateWithFlags(&S->comm_lane.ring[i], flags);
    }
    if (e != hipSuccess) return static_cast<int>(e);
    const char* th = std::getenv("T8GPU_STEPPER_THREADS");
    S->threads     = !(th && th[0] == '0');
    // (Measured and dropped: the interior tiles on a stream whose CU mask leaves 8 or 16 compute units to the comm lane
    //  (hipExtStreamCreateWithCUMask) -- 0.200 instead of 0.168 ms per step with the three-class lanes; such a stream is also a
    //  blocking one, which serialises it with the legacy default stream most callers use.)
    // the ghost window (t8gpu_hip.h): plain elements through the tile / 2D patch kernels
    const char* gw = std::getenv("T8GPU_GHOST_WINDOW");
    if (!S->subgrid && halo->cells_per_element <= 1 && S->plan.patch_dim != 3 && halo->n_send > 0 && halo->num_elements > 0 &&
        halo->sendbuf && halo->recvbuf && !(gw && gw[0] == '0')) {
      // send slots per owned element, from the rank's send list: -1 none, t >= 0 one slot, -(2 + i) a run of send_list
      const int            N = halo->num_elements, ns = halo->n_send;
      std::vector<int32_t> idx(ns), map(N, -1), cnt(N, 0), list;
      e = hipMemcpy(idx.data(), halo->send_idx, sizeof(int32_t) * ns, hipMemcpyDeviceToHost);
      if (e != hipSuccess) return static_cast<int>(e);
      for (int t = 0; t < ns; t++) {
        if (idx[t] < 0 || idx[t] >= N) return static_cast<int>(hipErrorInvalidValue);
        cnt[idx[t]]++;
      }
      std::vector<int32_t> first(N, -1);
      for (int t = 0; t < ns; t++) {
        const int el = idx[t];
        if (cnt[el] == 1) {
          map[el] = t;
        } else {
          if (first[el] < 0) {   // reserve the element's run
            first[el] = static_cast<int32_t>(list.size());
            list.resize(list.size() + cnt[el], -1);
            map[el] = -(2 + first[el]);
          }
          int32_t* run = list.data() + first[el];
          int      k   = 0;
          while (run[k] != -1) k++;
          run[k] = (k == cnt[el] - 1) ? static_cast<int32_t>(static_cast<uint32_t>(t) | 0x80000000u) : t;
        }
      }
      e = hipMalloc(&S->d_send_map, sizeof(int32_t) * N);
      if (e == hipSuccess) e = hipMemcpy(S->d_send_map, map.data(), sizeof(int32_t) * N, hipMemcpyHostToDevice);
      if (e == hipSuccess && !list.empty()) {
        e = hipMalloc(&S->d_send_list, sizeof(int32_t) * list.size());
        if (e == hipSuccess) e = hipMemcpy(S->d_send_list, list.data(), sizeof(int32_t) * list.size(), hipMemcpyHostToDevice);
      }
      if (e != hipSuccess) return static_cast<int>(e);
      S->plan_a           = S->plan;
      S->plan_a.ghost_buf = halo->recvbuf;
      S->plan_a.send_map  = static_cast<const int32_t*>(S->d_send_map);
      S->plan_a.send_list = static_cast<const int32_t*>(S->d_send_list);
      S->plan_a.send_buf  = halo->sendbuf;
      S->plan_a.n_owned   = N;
      S->zero_copy        = true;
    }
  }
  return 0;
}

int t8gpu_hip_plain_stepper_create(const T8gpuPlainPlan* plan, const T8gpuHalo* halo, void** out) {
  if (!plan || !out) return static_cast<int>(hipErrorInvalidValue);
  Stepper* S = new Stepper;
  S->plan = *plan;
  const int rc = stepper_halo_setup(S, halo);
  if (rc != 0) {
    t8gpu_hip_plain_stepper_destroy(S);
    return rc;
  }
  *out = S;
  return 0;
}

// Subgrid blocks: the same driver over a T8gpuSubgridPlan (halo->cells_per_element = 4^rank). The handle is used
// with t8gpu_hip_subgrid_stepper_iterate_steps_* and the generic destroy / timing / elapsed entry points above.
int t8gpu_hip_subgrid_stepper_create(const T8gpuSubgridPlan* plan, const T8gpuHalo* halo, void** out) {
  if (!plan || !out) return static_cast<int>(hipErrorInvalidValue);
  if (halo && halo->n_peers > 0 && halo->cells_per_element != (plan->rank == 3 ? 64 : 16)) return static_cast<int>(hipErrorInvalidValue);
  Stepper* S = new Stepper;
  S->subgrid = true;
  S->splan   = *plan;
  const int rc = stepper_halo_setup(S, halo);
  if (rc != 0) {
    t8gpu_hip_plain_stepper_destroy(S);
    return rc;
  }
  *out = S;
  return 0;
}

int t8gpu_hip_plain_stepper_destroy(void* h) {
  Stepper* S = static_cast<Stepper*>(h);
  if (!S) return 0;
  if (S->worker.joinable()) {
    {
      std::lock_guard<std::mutex> lk(S->mu);
      S->worker_exit = true;
    }
    S->cv.notify_one();
    S->worker.join();
  }
  for (Stepper::Lane* L : {&S->deep_lane, &S->comm_lane}) {
    for (hipEvent_t e : L->pool) (void)hipEventDestroy(e);
    for (hipEvent_t e : L->ring)
      if (e) (void)hipEventDestroy(e);
  }
  if (S->d_send_map) (void)hipFree(S->d_send_map);
  if (S->d_send_list) (void)hipFree(S->d_send_list);
  for (hipEvent_t e : S->pool) (void)hipEventDestroy(e);
  if (S->ev_state) (void)hipEventDestroy(S->ev_state);
  if (S->ev_ghost) (void)hipEventDestroy(S->ev_ghost);
  if (S->ev_interior) (void)hipEventDestroy(S->ev_interior);
  if (S->ev_deep) (void)hipEventDestroy(S->ev_deep);
  if (S->comm_stream) (void)hipStreamDestroy(S->comm_stream);
  if (S->near_stream) (void)hipStreamDestroy(S->near_stream);
  for (auto& ge : S->graph_cache)
    if (ge.exec) (void)hipGraphExecDestroy(ge.exec);
  for (hipEvent_t e : S->capture_events) (void)hipEventDestroy(e);
  if (S->scratch) (void)hipFree(S->scratch);
  if (S->ev_graph_in) (void)hipEventDestroy(S->ev_graph_in);
  if (S->ev_graph_out) (void)hipEventDestroy(S->ev_graph_out);
  if (S->graph_stream) (void)hipStreamDestroy(S->graph_stream);
  delete S;
  return 0;
}

int t8gpu_hip_plain_stepper_iterate_f32(void* h, int flux_kind, float* planes, size_t stride, int prev, int next,
                                        float delta_t, float* speed, void* stream) {
  return t8gpu_hip_plain_stepper_iterate_steps_f32(h, flux_kind, planes, stride, prev, next, delta_t, speed, 1, stream);
}
int t8gpu_hip_plain_stepper_iterate_f64(void* h, int flux_kind, double* planes, size_t stride, int prev, int next,
                                        double delta_t, double* speed, void* stream) {
  return t8gpu_hip_plain_stepper_iterate_steps_f64(h, flux_kind, planes, stride, prev, next, delta_t, speed, 1, stream);
}
int t8gpu_hip_plain_stepper_iterate_steps_f32(void* h, int flux_kind, float* planes, size_t stride, int prev, int next,
                                              float delta_t, float* speed, int n_steps, void* stream) {
  if (!h || prev < 0 || prev > 3 || next < 0 || next > 3 || prev == next || n_steps < 0) return static_cast<int>(hipErrorInvalidValue);
  if (static_cast<Stepper*>(h)->subgrid) return static_cast<int>(hipErrorInvalidValue);
  return iterate_graph<float, T8gpuVars_f32>(static_cast<Stepper*>(h), flux_kind, planes, stride, planes + 25 * stride, prev, next,
                                             delta_t, speed, n_steps, static_cast<hipStream_t>(stream));
}
int t8gpu_hip_plain_stepper_iterate_steps_f64(void* h, int flux_kind, double* planes, size_t stride, int prev, int next,
                                              double delta_t, double* speed, int n_steps, void* stream) {
  if (!h || prev < 0 || prev > 3 || next < 0 || next > 3 || prev == next || n_steps < 0) return static_cast<int>(hipErrorInvalidValue);
  if (static_cast<Stepper*>(h)->subgrid) return static_cast<int>(hipErrorInvalidValue);
  return iterate_graph<double, T8gpuVars_f64>(static_cast<Stepper*>(h), flux_kind, planes, stride, planes + 25 * stride, prev, next,
                                              delta_t, speed, n_steps, static_cast<hipStream_t>(stream));
}

int t8gpu_hip_subgrid_stepper_iterate_steps_f32(void* h, int flux_kind, float* planes, size_t stride, const float* volumes, int prev,
                                                int next, float delta_t, int n_steps, void* stream) {
  if (!h || !static_cast<Stepper*>(h)->subgrid || prev < 0 || prev > 3 || next < 0 || next > 3 || prev == next || n_steps < 0)
    return static_cast<int>(hipErrorInvalidValue);
  return iterate_graph<float, T8gpuVars_f32>(static_cast<Stepper*>(h), flux_kind, planes, stride, volumes, prev, next, delta_t, nullptr,
                                             n_steps, static_cast<hipStream_t>(stream));
}
int t8gpu_hip_subgrid_stepper_iterate_steps_f64(void* h, int flux_kind, double* planes, size_t stride, const double* volumes, int prev,
                                                int next, double delta_t, int n_steps, void* stream) {
  if (!h || !static_cast<Stepper*>(h)->subgrid || prev < 0 || prev > 3 || next < 0 || next > 3 || prev == next || n_steps < 0)
    return static_cast<int>(hipErrorInvalidValue);
  return iterate_graph<double, T8gpuVars_f64>(static_cast<Stepper*>(h), flux_kind, planes, stride, volumes, prev, next, delta_t, nullptr,
                                              n_steps, static_cast<hipStream_t>(stream));
}

// hipGraph replay of iterate_steps() (both step drivers): enable = 1 captures the whole call once per argument set and
// replays it with one hipGraphLaunch; 0 enqueues directly. counts (may be NULL) = {captures, replays} so far.
int t8gpu_hip_plain_stepper_graph(void* h, int enable, int* counts) {
  Stepper* S = static_cast<Stepper*>(h);
  if (!S) return static_cast<int>(hipErrorInvalidValue);
  if (enable >= 0) S->graph_mode = enable ? 1 : 0;
  if (counts) {
    counts[0] = S->graph_captures;
    counts[1] = S->graph_replays;
  }
  return 0;
}

// Diagnostics (T8GPU_STEPPER_PROFILE=1 in the environment, else all zeros): host time the step drivers of this process spent
// in {kernel launches, RCCL groups, event records, stream waits} since the last reset. ns4 / calls4 may be NULL.
int t8gpu_hip_stepper_host_profile(int reset, double* ns4, long long* calls4) {
  for (int c = 0; c < 4; c++) {
    if (ns4) ns4[c] = host_profile().ns[c];
    if (calls4) calls4[c] = host_profile().calls[c];
    if (reset) {
      host_profile().ns[c]    = 0;
      host_profile().calls[c] = 0;
    }
  }
  return host_profile().on ? 1 : 0;
}

// Host time the step driver spent enqueueing (wall time of the multi-rank iterate calls; both host threads of the two-lane
// driver overlap inside it) and the steps that covers, since the stepper was created or last reset. 0 / 0 for single-rank
// steppers (their enqueue is a handful of launches).
int t8gpu_hip_plain_stepper_host_time(void* h, int reset, double* total_ms, long long* steps) {
  Stepper* S = static_cast<Stepper*>(h);
  if (!S) return static_cast<int>(hipErrorInvalidValue);
  if (total_ms) *total_ms = S->host_ns * 1e-6;
  if (steps) *steps = S->host_steps;
  if (reset) {
    S->host_ns    = 0;
    S->host_steps = 0;
  }
  return 0;
}

// {RCCL version this library was compiled against (NCCL_VERSION_CODE), RCCL version of the library the process bound,
//  HIP version compiled against (HIP_VERSION), HIP runtime version}. The process may bind another build than the headers
// came from (here: the torch wheel's librccl / libamdhip64 against /opt/rocm's headers): the entry points this library
// uses -- ncclGetUniqueId, ncclCommInitRank, ncclGroupStart / End, ncclSend, ncclRecv, ncclCommDestroy / Abort, all with
// their NCCL 2.7 signatures, no ncclConfig_t -- exist unchanged in every 2.x since 2.7. t8gpu_hip_comm_create refuses a
// runtime with another major version or older than 2.7 (hipErrorNotSupported).
int t8gpu_hip_runtime_versions(int out4[4]) {
  if (!out4) return static_cast<int>(hipErrorInvalidValue);
  int nv = 0, hv = 0;
  out4[0] = NCCL_VERSION_CODE;
  out4[1] = ncclGetVersion(&nv) == ncclSuccess ? nv : -1;
  out4[2] = HIP_VERSION;
  out4[3] = hipRuntimeGetVersion(&hv) == hipSuccess ? hv : -1;
  return 0;
}

int t8gpu_hip_plain_stepper_timing(void* h, int enable) {
  Stepper* S = static_cast<Stepper*>(h);
  if (!S) return static_cast<int>(hipErrorInvalidValue);
  S->timing = enable < 0 ? 0 : enable;
  S->stages_timed = 0;
  S->used   = 0;
  S->deep_lane.used = S->comm_lane.used = 0;
  return 0;
}

// Sum of the stage-kernel durations recorded since timing was enabled (call after a device sync).
int t8gpu_hip_plain_stepper_elapsed(void* h, double* total_ms, int* launches) {
  Stepper* S = static_cast<Stepper*>(h);
  if (!S || !total_ms || !launches) return static_cast<int>(hipErrorInvalidValue);
  double sum = 0;
  int    n   = 0;
  auto   add = [&](const std::vector<hipEvent_t>& pool, size_t used) -> int {
    for (size_t i = 0; i + 1 < used; i += 2) {
      float ms = 0;
      T8_HIP_TRY(hipEventElapsedTime(&ms, pool[i], pool[i + 1]));
      sum += ms;
      n++;
    }
    return 0;
  };
  T8_TRY(add(S->pool, S->used));
  T8_TRY(add(S->deep_lane.pool, S->deep_lane.used));   // (the two-lane driver's launches)
  T8_TRY(add(S->comm_lane.pool, S->comm_lane.used));
  *total_ms = sum;
  *launches = n;
  return 0;
}


// Number of RK stages whose kernels were bracketed by events since timing was enabled (elapsed() / this = the
// average duration of one stage's kernels, however many tile ranges a stage is split into).
int t8gpu_hip_plain_stepper_timed_stages(void* h) {
  Stepper* S = static_cast<Stepper*>(h);
  return S ? S->stages_timed : 0;
}

}  // extern "C"
