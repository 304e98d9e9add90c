// t8gpu/timestepping/ssp_runge_kutta.h (MI355X backend)
//
// Kernel templates with the reference's names and parameter lists
// (t8gpu/timestepping/ssp_runge_kutta.h:18-116, bodies ssp_runge_kutta.inl:30-221):
//   timestepping::SSP_3RK_step{1,2,3}<VariableType><<<ceil(N/256), 256>>>(prev[, stepK-1], out, fluxes, volume, dt, N)
//   timestepping::subgrid::SSP_3RK_step{1,2,3}<VariableType, SubgridType><<<N, block_size>>>(...)
// Each stage also zeroes the flux planes. Coefficients are the reference's truncated decimals
// (ssp_runge_kutta.inl:12-14,23-25), not 1/3 and 2/3.
// These are the accessor-level kernels user code launches itself; the fused path of the C-ABI
// (t8gpu/backend/hip_fast.h) performs the same update inside the flux kernel.
#ifndef T8GPU_HIP_TIMESTEPPING_SSP_RUNGE_KUTTA_H
#define T8GPU_HIP_TIMESTEPPING_SSP_RUNGE_KUTTA_H

#include <t8gpu/memory/memory_manager.h>
#include <t8gpu/memory/subgrid_memory_manager.h>

namespace t8gpu::timestepping {

  template<typename ft>
  struct rk_coeffs;
  template<>
  struct rk_coeffs<float> {
    static constexpr float stage_2_1 = 0.75f, stage_2_2 = 0.25f, stage_2_3 = 0.25f;
    static constexpr float stage_3_1 = 0.33333333333333f, stage_3_2 = 0.66666666666666f, stage_3_3 = 0.66666666666666f;
  };
  template<>
  struct rk_coeffs<double> {
    static constexpr double stage_2_1 = 0.75, stage_2_2 = 0.25, stage_2_3 = 0.25;
    static constexpr double stage_3_1 = 0.33333333333333, stage_3_2 = 0.66666666666666, stage_3_3 = 0.66666666666666;
  };

  namespace detail {
    // out = a * prev + b * mid + c * dt / vol * flux for every variable of one cell; flux := 0
    template<int STAGE, typename Prev, typename Mid, typename Out, typename Flux, typename ft, typename Index>
    __device__ inline void stage_update(Prev& prev, Mid& mid, Out& out, Flux& flux, ft volume, ft dt, size_t nvars, Index at) {
      for (size_t k = 0; k < nvars; k++) {
        if constexpr (STAGE == 1) {
          at(out, k) = at(prev, k) + dt / volume * at(flux, k);
        } else if constexpr (STAGE == 2) {
          at(out, k) = rk_coeffs<ft>::stage_2_1 * at(prev, k) + rk_coeffs<ft>::stage_2_2 * at(mid, k) +
                       rk_coeffs<ft>::stage_2_3 * dt / volume * at(flux, k);
        } else {
          at(out, k) = rk_coeffs<ft>::stage_3_1 * at(prev, k) + rk_coeffs<ft>::stage_3_2 * at(mid, k) +
                       rk_coeffs<ft>::stage_3_3 * dt / volume * at(flux, k);
        }
        at(flux, k) = ft(0.0);
      }
    }
  }  // namespace detail

#define T8GPU_PLAIN_STAGE(STAGE, ...)                                                                              \
  const int i = blockIdx.x * blockDim.x + threadIdx.x;                                                             \
  if (i >= num_elements) return;                                                                                   \
  using ft = typename variable_traits<VariableType>::float_type;                                                   \
  auto at  = [i](auto& acc, size_t k) -> decltype(auto) { return acc.get(k)[i]; };                                 \
  detail::stage_update<STAGE>(__VA_ARGS__, volume[i], delta_t, variable_traits<VariableType>::nb_variables, at)

  template<typename VariableType>
  __global__ void SSP_3RK_step1(MemoryAccessorOwn<VariableType> prev, MemoryAccessorOwn<VariableType> step1,
                                MemoryAccessorOwn<VariableType> fluxes,
                                typename variable_traits<VariableType>::float_type const* __restrict__ volume,
                                typename variable_traits<VariableType>::float_type delta_t, int num_elements) {
    T8GPU_PLAIN_STAGE(1, prev, prev, step1, fluxes);
  }
  template<typename VariableType>
  __global__ void SSP_3RK_step2(MemoryAccessorOwn<VariableType> prev, MemoryAccessorOwn<VariableType> step1,
                                MemoryAccessorOwn<VariableType> step2, MemoryAccessorOwn<VariableType> fluxes,
                                typename variable_traits<VariableType>::float_type const* __restrict__ volume,
                                typename variable_traits<VariableType>::float_type delta_t, int num_elements) {
    T8GPU_PLAIN_STAGE(2, prev, step1, step2, fluxes);
  }
  template<typename VariableType>
  __global__ void SSP_3RK_step3(MemoryAccessorOwn<VariableType> prev, MemoryAccessorOwn<VariableType> step2,
                                MemoryAccessorOwn<VariableType> next, MemoryAccessorOwn<VariableType> fluxes,
                                typename variable_traits<VariableType>::float_type const* __restrict__ volume,
                                typename variable_traits<VariableType>::float_type delta_t, int num_elements) {
    T8GPU_PLAIN_STAGE(3, prev, step2, next, fluxes);
  }
#undef T8GPU_PLAIN_STAGE

  namespace subgrid {
    // one workgroup per block, one thread per subcell (launch with SubgridType::block_size)
#define T8GPU_SUBGRID_STAGE(STAGE, ...)                                                                               \
  using ft        = typename variable_traits<VariableType>::float_type;                                               \
  const int e_idx = blockIdx.x;                                                                                       \
  const ft  vol   = volumes[e_idx] / static_cast<ft>(SubgridType::size);                                              \
  const size_t cell = static_cast<size_t>(e_idx) * SubgridType::size + threadIdx.x +                                  \
                      SubgridType::template extent<0> * (threadIdx.y + (SubgridType::rank == 3 ? SubgridType::template extent<1> * threadIdx.z : 0)); \
  auto at = [cell](auto& acc, size_t k) -> decltype(auto) { return static_cast<ft*>(acc.get(k))[cell]; };             \
  timestepping::detail::stage_update<STAGE>(__VA_ARGS__, vol, delta_t, variable_traits<VariableType>::nb_variables, at)

    template<typename VariableType, typename SubgridType>
    __global__ void SSP_3RK_step1(SubgridMemoryAccessorOwn<VariableType, SubgridType> prev,
                                  SubgridMemoryAccessorOwn<VariableType, SubgridType> step1,
                                  SubgridMemoryAccessorOwn<VariableType, SubgridType> fluxes,
                                  typename variable_traits<VariableType>::float_type const* __restrict__ volumes,
                                  typename variable_traits<VariableType>::float_type delta_t) {
      T8GPU_SUBGRID_STAGE(1, prev, prev, step1, fluxes);
    }
    template<typename VariableType, typename SubgridType>
    __global__ void SSP_3RK_step2(SubgridMemoryAccessorOwn<VariableType, SubgridType> prev,
                                  SubgridMemoryAccessorOwn<VariableType, SubgridType> step1,
                                  SubgridMemoryAccessorOwn<VariableType, SubgridType> step2,
                                  SubgridMemoryAccessorOwn<VariableType, SubgridType> fluxes,
                                  typename variable_traits<VariableType>::float_type const* __restrict__ volumes,
                                  typename variable_traits<VariableType>::float_type delta_t) {
      T8GPU_SUBGRID_STAGE(2, prev, step1, step2, fluxes);
    }
    template<typename VariableType, typename SubgridType>
    __global__ void SSP_3RK_step3(SubgridMemoryAccessorOwn<VariableType, SubgridType> prev,
                                  SubgridMemoryAccessorOwn<VariableType, SubgridType> step2,
                                  SubgridMemoryAccessorOwn<VariableType, SubgridType> next,
                                  SubgridMemoryAccessorOwn<VariableType, SubgridType> fluxes,
                                  typename variable_traits<VariableType>::float_type const* __restrict__ volumes,
                                  typename variable_traits<VariableType>::float_type delta_t) {
      T8GPU_SUBGRID_STAGE(3, prev, step2, next, fluxes);
    }
#undef T8GPU_SUBGRID_STAGE
  }  // namespace subgrid

}  // namespace t8gpu::timestepping

#endif  // T8GPU_HIP_TIMESTEPPING_SSP_RUNGE_KUTTA_H
