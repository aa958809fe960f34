// Fused Adam(amsgrad) step over a list of fp32 parameter tensors in ONE launch (reference optimizer:
// torch.optim.Adam(params, lr, amsgrad=True), train_DFOLD_dynamics.py:412; 184 M parameters, 89 % of them the shared conv
// tower).  torch's foreach implementation walks the 4 state tensors in ~10 separate passes; here every element is read
// and written once: p, g, exp_avg, exp_avg_sq, max_exp_avg_sq in, p and the three states out (HBM-bound, 8 x 4 B per
// parameter = 5.9 GB per step at 184 M parameters).
//   m <- m + (g - m)(1 - b1);  v <- b2 v + (1 - b2) g^2;  vmax <- max(vmax, v)
//   p <- p - (lr / (1 - b1^t)) * m / (sqrt(vmax) / sqrt(1 - b2^t) + eps)
// Tensors are described by a device table (5 pointers + element count each) and split into chunks of ADAM_CHUNK elements;
// chunk_start[i] = first chunk of tensor i (exclusive prefix sum, n_tensors + 1 entries): a workgroup binary-searches
// its tensor.
#include "dfold_common.h"
#include "../../include/dfold_hip.h"

#define ADAM_CHUNK 8192

__global__ __launch_bounds__(256) void adam_amsgrad_kernel(const dfold_adam_tensor* __restrict__ tab,
                                                           const int* __restrict__ chunk_start, int n_tensors, float lr_c1,
                                                           float omb1, float b2, float omb2, float rsqrt_c2, float eps) {
  const int chunk = blockIdx.x;
  int lo = 0, hi = n_tensors;           // chunk_start[lo] <= chunk < chunk_start[hi]
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (chunk_start[mid] <= chunk) lo = mid; else hi = mid;
  }
  const dfold_adam_tensor t = tab[lo];
  const long base = (long)(chunk - chunk_start[lo]) * ADAM_CHUNK;
  const long end = min(t.n, base + ADAM_CHUNK);
  float* __restrict__ p = (float*)t.p;
  const float* __restrict__ g = (const float*)t.g;
  float* __restrict__ m = (float*)t.exp_avg;
  float* __restrict__ v = (float*)t.exp_avg_sq;
  float* __restrict__ vm = (float*)t.max_exp_avg_sq;
  const bool vec = ((((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v | (uintptr_t)vm) & 15) == 0);
  for (long i = base + threadIdx.x * 4; i < end; i += 256 * 4) {
    if (vec && i + 3 < end) {
      float4 pp = *(const float4*)(p + i), gg = *(const float4*)(g + i), mm = *(const float4*)(m + i),
             vv = *(const float4*)(v + i), xx = *(const float4*)(vm + i);
      float* P = (float*)&pp; const float* G = (const float*)&gg; float* M = (float*)&mm; float* V = (float*)&vv; float* X = (float*)&xx;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        M[j] = M[j] + (G[j] - M[j]) * omb1;
        V[j] = V[j] * b2 + omb2 * G[j] * G[j];
        X[j] = fmaxf(X[j], V[j]);
        P[j] = P[j] - lr_c1 * (M[j] / (sqrtf(X[j]) * rsqrt_c2 + eps));
      }
      *(float4*)(p + i) = pp; *(float4*)(m + i) = mm; *(float4*)(v + i) = vv; *(float4*)(vm + i) = xx;
    } else {
      for (long k = i; k < min(end, i + 4); ++k) {
        const float gk = g[k];
        const float mk = m[k] + (gk - m[k]) * omb1;
        const float vk = v[k] * b2 + omb2 * gk * gk;
        const float xk = fmaxf(vm[k], vk);
        m[k] = mk; v[k] = vk; vm[k] = xk;
        p[k] = p[k] - lr_c1 * (mk / (sqrtf(xk) * rsqrt_c2 + eps));
      }
    }
  }
}

extern "C" int dfold_adam_amsgrad(const dfold_adam_tensor* table, const int32_t* chunk_start, int32_t n_tensors,
                                  int32_t n_chunks, double lr, double beta1, double beta2, double eps, int64_t step,
                                  void* stream) {
  if (!table || !chunk_start || n_tensors <= 0 || n_chunks <= 0 || step < 1) return DFOLD_EINVAL;
  if (!(lr >= 0.) || !(beta1 >= 0. && beta1 < 1.) || !(beta2 >= 0. && beta2 < 1.) || !(eps >= 0.)) return DFOLD_EINVAL;
  const double c1 = 1.0 - pow(beta1, (double)step), c2 = 1.0 - pow(beta2, (double)step);
  DFOLD_LAUNCH(adam_amsgrad_kernel, dim3((unsigned)n_chunks), dim3(256), 0, (hipStream_t)stream, table, (const int*)chunk_start,
               n_tensors, (float)(lr / c1), (float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2),
               (float)(1.0 / sqrt(c2)), (float)eps);   // hyper-parameters arrive as doubles: 1 - beta is formed in double like
                                                        // the reference optimizer does (1.f - 0.999f is 1.3e-5 off 0.001)
  return dfold_check_launch();
}

extern "C" int dfold_adam_chunk(void) { return ADAM_CHUNK; }
