// Counter-based device random numbers for the diffusion draws (the reference draws on the host with numpy inside the
// sampling loop, src/data/so3_diffuser.py:347-349 / r3_diffuser.py:140-147, and in the loader workers for the forward
// noising, so3_diffuser.py:233-248 / r3_diffuser.py:96-99): Philox4x32-10 (Salmon, Moraes, Dror, Shaw, SC'11; key =
// 64-bit seed, counter = (block index, 64-bit subsequence)), four 32-bit words per counter.
//   uniform:  u_j = (x_j + 0.5) * 2^-32                      in (0, 1), exactly representable in fp64
//   normal :  Box-Muller in fp64 on (u_0, u_1) and (u_2, u_3): sqrt(-2 ln u_a) * {cos, sin}(2 pi u_b)
// Element e of stream (seed, subseq) is word e % 4 of counter block e / 4: streams are reproducible, independent of the
// launch geometry, and addressable (a consumer kernel may regenerate the draw of element e itself).
// The host-numpy path (draws as inputs to dfold_se3_reverse / dfold_se3_forward_marginal) stays for parity with the
// reference's RNG stream; this is the production path (no host round trip inside the sampling loop).
#include "dfold_common.h"
#include "../../include/dfold_hip.h"

__device__ __forceinline__ void philox4x32_10(uint32_t c[4], uint32_t k0, uint32_t k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c[0], p1 = (uint64_t)0xCD9E8D57u * c[2];
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0, n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
    c[1] = (uint32_t)p1;
    c[3] = (uint32_t)p0;
    c[0] = n0;
    c[2] = n2;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
}

template <bool NORMAL>
__global__ __launch_bounds__(256) void philox_fill_kernel(double* __restrict__ out, long n, uint64_t seed, uint64_t subseq) {
  const long blocks = (n + 3) >> 2;
  for (long b = (long)blockIdx.x * blockDim.x + threadIdx.x; b < blocks; b += (long)gridDim.x * blockDim.x) {
    uint32_t c[4] = {(uint32_t)b, (uint32_t)((uint64_t)b >> 32), (uint32_t)subseq, (uint32_t)(subseq >> 32)};
    philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
    double v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = ((double)c[j] + 0.5) * 2.3283064365386963e-10;   // 2^-32
    if (NORMAL) {
      const double r0 = sqrt(-2.0 * log(v[0])), r1 = sqrt(-2.0 * log(v[2]));
      const double a0 = 6.283185307179586476925 * v[1], a1 = 6.283185307179586476925 * v[3];
      v[0] = r0 * cos(a0);
      v[1] = r0 * sin(a0);
      v[2] = r1 * cos(a1);
      v[3] = r1 * sin(a1);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (4 * b + j < n) out[4 * b + j] = v[j];
  }
}

static int philox_fill(double* out, int64_t n, uint64_t seed, uint64_t subseq, bool normal, void* stream) {
  if (!out || n <= 0) return DFOLD_EINVAL;
  long blocks = ((n + 3) / 4 + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  if (normal)
    DFOLD_LAUNCH(philox_fill_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, out, (long)n, seed, subseq);
  else
    DFOLD_LAUNCH(philox_fill_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, out, (long)n, seed, subseq);
  return dfold_check_launch();
}

extern "C" int dfold_philox_normal_f64(double* out, int64_t n, uint64_t seed, uint64_t subseq, void* stream) {
  return philox_fill(out, n, seed, subseq, true, stream);
}

extern "C" int dfold_philox_uniform_f64(double* out, int64_t n, uint64_t seed, uint64_t subseq, void* stream) {
  return philox_fill(out, n, seed, subseq, false, stream);
}
