// Torsion angles + backbone frame -> idealised atom14 / atom37 coordinates, one thread per residue:
//   feats.torsion_angles_to_frames (openfold/utils/feats.py:165-228): 8 rigid groups per residue from the residue-type
//     default frames and 7 torsions (chi2..chi4 chained onto chi1), composed onto the backbone frame,
//   all_atom.frames_to_atom14_pos (src/data/all_atom.py:114-154): integer gather of the group frame per atom + literature
//     positions + mask,
//   atom14_to_atom37 (src/model/Dfold_network_dynamic.py:574-594): integer gather 37 <- 14 + mask.
// The reference runs this as ~60 tiny aten launches (incl. five batched 3x3 matmuls through the BLAS library) and two
// host syncs; here it is one launch, tables (6.5 KB) staged in LDS.  Integer indexing is bit-exact by construction.
#include "dfold_common.h"
#include "../../include/dfold_hip.h"

struct Frame {
  float r[9];
  float t[3];
};

__device__ __forceinline__ Frame compose(const Frame& a, const Frame& b) {  // a o b : x -> Ra (Rb x + tb) + ta
  Frame o;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
#pragma unroll
    for (int j = 0; j < 3; ++j) o.r[3 * i + j] = a.r[3 * i] * b.r[j] + a.r[3 * i + 1] * b.r[3 + j] + a.r[3 * i + 2] * b.r[6 + j];
    o.t[i] = a.r[3 * i] * b.t[0] + a.r[3 * i + 1] * b.t[1] + a.r[3 * i + 2] * b.t[2] + a.t[i];
  }
  return o;
}

// tables: default_frames [21][8][4][4] f32, atom14_group [21][14] i64, atom14_mask [21][14] f32, atom14_pos [21][14][3] f32,
//         atom37_to_atom14 [21][37] i64, atom37_mask [21][37] f32
__global__ __launch_bounds__(128) void frames_to_atoms_kernel(const float* __restrict__ t7, const float* __restrict__ angles,
                                                              const long* __restrict__ aatype,
                                                              const float* __restrict__ default_frames,
                                                              const long* __restrict__ atom14_group,
                                                              const float* __restrict__ atom14_mask,
                                                              const float* __restrict__ atom14_pos,
                                                              const long* __restrict__ atom37_to_atom14,
                                                              const float* __restrict__ atom37_mask, float* __restrict__ atom14,
                                                              float* __restrict__ atom37, long P) {
  __shared__ float s_df[21 * 8 * 16];
  __shared__ float s_pos[21 * 14 * 3];
  __shared__ float s_m14[21 * 14];
  __shared__ float s_m37[21 * 37];
  __shared__ int s_grp[21 * 14];
  __shared__ int s_idx[21 * 37];
  for (int e = threadIdx.x; e < 21 * 8 * 16; e += blockDim.x) s_df[e] = default_frames[e];
  for (int e = threadIdx.x; e < 21 * 14 * 3; e += blockDim.x) s_pos[e] = atom14_pos[e];
  for (int e = threadIdx.x; e < 21 * 14; e += blockDim.x) {
    s_m14[e] = atom14_mask[e];
    s_grp[e] = (int)atom14_group[e];
  }
  for (int e = threadIdx.x; e < 21 * 37; e += blockDim.x) {
    s_m37[e] = atom37_mask[e];
    s_idx[e] = (int)atom37_to_atom14[e];
  }
  __syncthreads();
  const long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= P) return;
  const int aa = (int)aatype[p];
  // backbone frame from the (unnormalised, like the reference) quaternion form
  const float a = t7[p * 7], b = t7[p * 7 + 1], c = t7[p * 7 + 2], d = t7[p * 7 + 3];
  Frame bb;
  bb.r[0] = a * a + b * b - c * c - d * d; bb.r[1] = 2 * (b * c - a * d);         bb.r[2] = 2 * (b * d + a * c);
  bb.r[3] = 2 * (b * c + a * d);         bb.r[4] = a * a - b * b + c * c - d * d; bb.r[5] = 2 * (c * d - a * b);
  bb.r[6] = 2 * (b * d - a * c);         bb.r[7] = 2 * (c * d + a * b);         bb.r[8] = a * a - b * b - c * c + d * d;
  bb.t[0] = t7[p * 7 + 4]; bb.t[1] = t7[p * 7 + 5]; bb.t[2] = t7[p * 7 + 6];
  Frame grp[8];
#pragma unroll
  for (int g = 0; g < 8; ++g) {
    const float* m = s_df + (aa * 8 + g) * 16;   // 4x4 row-major
    const float sn = g == 0 ? 0.f : angles[(p * 7 + g - 1) * 2];
    const float cs = g == 0 ? 1.f : angles[(p * 7 + g - 1) * 2 + 1];
    // default rotation times a rotation about x by the torsion: [1 0 0; 0 c -s; 0 s c]
    Frame f;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const float r0 = m[4 * i], r1 = m[4 * i + 1], r2 = m[4 * i + 2];
      f.r[3 * i] = r0;
      f.r[3 * i + 1] = r1 * cs + r2 * sn;
      f.r[3 * i + 2] = -r1 * sn + r2 * cs;
      f.t[i] = m[4 * i + 3];
    }
    grp[g] = f;
  }
  grp[5] = compose(grp[4], grp[5]);
  grp[6] = compose(grp[5], grp[6]);
  grp[7] = compose(grp[6], grp[7]);
#pragma unroll
  for (int g = 0; g < 8; ++g) grp[g] = compose(bb, grp[g]);
  float a14[14][3];
#pragma unroll
  for (int k = 0; k < 14; ++k) {
    const int gi = s_grp[aa * 14 + k];
    const float* lp = s_pos + (aa * 14 + k) * 3;
    const float mk = s_m14[aa * 14 + k];
    Frame f = grp[0];
#pragma unroll
    for (int g = 1; g < 8; ++g)
      if (gi == g) f = grp[g];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      a14[k][i] = (f.r[3 * i] * lp[0] + f.r[3 * i + 1] * lp[1] + f.r[3 * i + 2] * lp[2] + f.t[i]) * mk;
      atom14[(p * 14 + k) * 3 + i] = a14[k][i];
    }
  }
  for (int k = 0; k < 37; ++k) {
    const int src = s_idx[aa * 37 + k];
    const float mk = s_m37[aa * 37 + k];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      float v = a14[0][i];
#pragma unroll
      for (int q = 1; q < 14; ++q)
        if (src == q) v = a14[q][i];
      atom37[(p * 37 + k) * 3 + i] = v * mk;
    }
  }
}

extern "C" int dfold_frames_to_atoms(const float* t7, const float* angles, const int64_t* aatype, const float* default_frames,
                                     const int64_t* atom14_group, const float* atom14_mask, const float* atom14_pos,
                                     const int64_t* atom37_to_atom14, const float* atom37_mask, float* atom14, float* atom37,
                                     int64_t P, void* stream) {
  if (!t7 || !angles || !aatype || !default_frames || !atom14_group || !atom14_mask || !atom14_pos || !atom37_to_atom14 ||
      !atom37_mask || !atom14 || !atom37 || P <= 0)
    return DFOLD_EINVAL;
  DFOLD_LAUNCH(frames_to_atoms_kernel, dim3((unsigned)((P + 127) / 128)), dim3(128), 0, (hipStream_t)stream, t7, angles,
               (const long*)aatype, default_frames, (const long*)atom14_group, atom14_mask, atom14_pos,
               (const long*)atom37_to_atom14, atom37_mask, atom14, atom37, (long)P);
  return dfold_check_launch();
}
