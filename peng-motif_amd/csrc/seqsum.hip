// seqsum.hip -- pengk_sequential_sum_f32: left-to-right float32 sums of arrays, bit for bit, by the scan of seqsum.h.
// The EM's serial mode (em.hip) uses the same evaluation on the weights of a PWM cell; this entry point exposes it on
// plain arrays so that it can be checked against a CPU loop on inputs an EM run never produces (denormals, overflow,
// terms spanning the whole exponent range) -- and it is how a caller would fold any float32 table the reference's way.
#include "pengk_internal.h"
#include "seqsum.h"

namespace pengk {
namespace {

struct PlainTerms {
  const float* __restrict__ t;
  uint64_t n;
  // share `part` of NF: terms 64 k + lane of block b, k = part * 64 / NF .., in R[k - part * 64 / NF]; zero behind the
  // end of the chain (s + 0 = s for every s >= +0)
  template <uint32_t NF>
  __device__ __forceinline__ void load(uint32_t b, uint32_t part, uint32_t lane, float (&R)[64 / NF]) const {
    const uint64_t c0 = (uint64_t)b * seqsum::BLOCK + (uint64_t)part * (seqsum::BLOCK / NF) + lane;
#pragma unroll
    for (uint32_t k = 0; k < 64u / NF; ++k) {
      const uint64_t c = c0 + 64u * k;
      R[k] = c < n ? t[c] : 0.0f;
    }
  }
  template <uint32_t NF>
  __device__ __forceinline__ void deposit(uint32_t part, uint32_t lane, const float (&R)[64 / NF], float* lds) const {
#pragma unroll
    for (uint32_t k = 0; k < 64u / NF; ++k) lds[(part * (64u / NF) + k) * seqsum::SEG_STRIDE + lane] = R[k];
  }
  // the plain loop, for chains with a negative or non-finite term (every lane computes the same value)
  __device__ float serial() const {
    float s = 0.0f;
    for (uint64_t i = 0; i < n; ++i) s += t[i];
    return s;
  }
};

__global__ __launch_bounds__(seqsum::CHAIN_THREADS, 4) void sequential_sum_kernel(const float* __restrict__ terms, uint64_t chain_len,
                                                                                float* __restrict__ out) {
  __shared__ __attribute__((aligned(16))) float lds[seqsum::CHAIN_LDS_FLOATS];
  const PlainTerms src{terms + (size_t)blockIdx.x * chain_len, chain_len};
  const uint32_t n_blocks = (uint32_t)((chain_len + seqsum::BLOCK - 1) / seqsum::BLOCK);
  const float s = seqsum::fold_chain<PlainTerms, true>(src, n_blocks, lds, threadIdx.x);
  if (threadIdx.x == 0) out[blockIdx.x] = s;
}

}  // namespace

int launch_sequential_sum(pengk_ctx* ctx, const float* d_terms, uint64_t n_chains, uint64_t chain_len, float* d_out) {
  if (n_chains == 0) return PENGK_OK;
  if (n_chains > 0x7FFFFFFFull) return fail(PENGK_ERR_RANGE, "pengk_sequential_sum_f32: %llu chains in one call", (unsigned long long)n_chains);
  if (chain_len > ((uint64_t)1 << 43)) return fail(PENGK_ERR_RANGE, "pengk_sequential_sum_f32: chain of %llu terms", (unsigned long long)chain_len);
  hipLaunchKernelGGL(sequential_sum_kernel, dim3((unsigned)n_chains), dim3(seqsum::CHAIN_THREADS), 0, ctx->stream, d_terms, chain_len, d_out);
  PENGK_HIP(hipGetLastError());
  return PENGK_OK;
}

}  // namespace pengk
