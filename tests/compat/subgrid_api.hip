// tests/compat/subgrid_api.hip -- the Subgrid part of the header API, used the way
// examples/subgrid/{solver.inl,kernels.inl} use it: Subgrid<4,4,4> traits, per-variable accessors
// called as rho(e_idx, i, j, k), kernels launched with SubgridType::block_size, the library's
// timestepping::subgrid::SSP_3RK_step kernels, get(rank, vars...) on the "All" accessor.
// Self-checking: prints "subgrid_api OK" and returns 0.
#include <t8gpu/memory/subgrid_memory_manager.h>
#include <t8gpu/mesh/subgrid_mesh_manager.h>
#include <t8gpu/timestepping/ssp_runge_kutta.h>

#include <cmath>
#include <cstdio>
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
  std::printf("subgrid_api OK\n");
  return 0;
}
