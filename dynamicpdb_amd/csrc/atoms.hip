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

// ------------------------------------------------------------------------------------------------------------------
// Backward of frames_to_atoms (the reference builds the atoms inside autograd, src/model/Dfold_network_dynamic.py:532-538;
// its bb-atom / dist-mat loss terms read them, train_DFOLD_dynamics.py:1317-1364): one thread per residue re-derives the
// 8 group frames and walks the chain in reverse.  compose(a, b): R_o = Ra Rb, t_o = Ra tb + ta
//   =>  dRa += dR_o Rb^T + dt_o (x) tb,  dta += dt_o,  dRb += Ra^T dR_o,  dtb += Ra^T dt_o.
// ------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void compose_bwd(const Frame& a, const Frame& b, const Frame& go, Frame& ga, Frame& gb) {
#pragma unroll
  for (int i = 0; i < 3; ++i) {
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      // dRa[i][j] += sum_k dR_o[i][k] Rb[j][k] + dt_o[i] tb[j];   dRb[i][j] += sum_k Ra[k][i] dR_o[k][j]
      ga.r[3 * i + j] += go.r[3 * i] * b.r[3 * j] + go.r[3 * i + 1] * b.r[3 * j + 1] + go.r[3 * i + 2] * b.r[3 * j + 2] + go.t[i] * b.t[j];
      gb.r[3 * i + j] += a.r[i] * go.r[j] + a.r[3 + i] * go.r[3 + j] + a.r[6 + i] * go.r[6 + j];
    }
    ga.t[i] += go.t[i];
    gb.t[i] += a.r[i] * go.t[0] + a.r[3 + i] * go.t[1] + a.r[6 + i] * go.t[2];
  }
}

__global__ __launch_bounds__(128) void frames_to_atoms_bwd_kernel(const float* __restrict__ t7, const float* __restrict__ angles,
                                                                  const long* __restrict__ aatype,
                                                                  const float* __restrict__ default_frames,
                                                                  const long* __restrict__ atom14_group,
                                                                  const float* __restrict__ atom14_mask,
                                                                  const float* __restrict__ atom14_pos,
                                                                  const long* __restrict__ atom37_to_atom14,
                                                                  const float* __restrict__ atom37_mask,
                                                                  const float* __restrict__ g14, const float* __restrict__ g37,
                                                                  float* __restrict__ dt7, float* __restrict__ dangles, long P) {
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
  const float a = t7[p * 7], b = t7[p * 7 + 1], c = t7[p * 7 + 2], d = t7[p * 7 + 3];
  Frame bb;
  bb.r[0] = a * a + b * b - c * c - d * d; bb.r[1] = 2 * (b * c - a * d);         bb.r[2] = 2 * (b * d + a * c);
  bb.r[3] = 2 * (b * c + a * d);         bb.r[4] = a * a - b * b + c * c - d * d; bb.r[5] = 2 * (c * d - a * b);
  bb.r[6] = 2 * (b * d - a * c);         bb.r[7] = 2 * (c * d + a * b);         bb.r[8] = a * a - b * b - c * c + d * d;
  bb.t[0] = t7[p * 7 + 4]; bb.t[1] = t7[p * 7 + 5]; bb.t[2] = t7[p * 7 + 6];
  // forward again: local frames f[g] (default x torsion), chained frames ch[g] (chi2..chi4 onto chi1; ch[g] = f[g] for g < 5)
  Frame f[8], ch[8];
#pragma unroll
  for (int g = 0; g < 8; ++g) {
    const float* m = s_df + (aa * 8 + g) * 16;
    const float sn = g == 0 ? 0.f : angles[(p * 7 + g - 1) * 2];
    const float cs = g == 0 ? 1.f : angles[(p * 7 + g - 1) * 2 + 1];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const float r0 = m[4 * i], r1 = m[4 * i + 1], r2 = m[4 * i + 2];
      f[g].r[3 * i] = r0;
      f[g].r[3 * i + 1] = r1 * cs + r2 * sn;
      f[g].r[3 * i + 2] = -r1 * sn + r2 * cs;
      f[g].t[i] = m[4 * i + 3];
    }
    ch[g] = f[g];
  }
  ch[5] = compose(ch[4], f[5]);
  ch[6] = compose(ch[5], f[6]);
  ch[7] = compose(ch[6], f[7]);
  // gradient of the global group frames G_g = bb o ch[g] from the atoms: atom14[k] = (R_G lp + t_G) mk, atom37 = gather x mask
  Frame gG[8];
#pragma unroll
  for (int g = 0; g < 8; ++g) {
#pragma unroll
    for (int e = 0; e < 9; ++e) gG[g].r[e] = 0.f;
    gG[g].t[0] = gG[g].t[1] = gG[g].t[2] = 0.f;
  }
  float da[14][3];
#pragma unroll
  for (int k = 0; k < 14; ++k)
#pragma unroll
    for (int i = 0; i < 3; ++i) da[k][i] = g14 ? g14[(p * 14 + k) * 3 + i] : 0.f;
  if (g37) {
    for (int k = 0; k < 37; ++k) {
      const int src = s_idx[aa * 37 + k];
      const float mk = s_m37[aa * 37 + k];
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        const float v = g37[(p * 37 + k) * 3 + i] * mk;
#pragma unroll
        for (int q = 0; q < 14; ++q)
          if (src == q) da[q][i] += v;
      }
    }
  }
#pragma unroll
  for (int k = 0; k < 14; ++k) {
    const int gi = s_grp[aa * 14 + k];
    const float* lp = s_pos + (aa * 14 + k) * 3;
    const float mk = s_m14[aa * 14 + k];
#pragma unroll
    for (int g = 0; g < 8; ++g)
      if (gi == g) {
#pragma unroll
        for (int i = 0; i < 3; ++i) {
          const float v = da[k][i] * mk;
          gG[g].r[3 * i] += v * lp[0];
          gG[g].r[3 * i + 1] += v * lp[1];
          gG[g].r[3 * i + 2] += v * lp[2];
          gG[g].t[i] += v;
        }
      }
  }
  // G_g = compose(bb, ch[g])
  Frame gbb, gch[8], gf[8];
#pragma unroll
  for (int e = 0; e < 9; ++e) gbb.r[e] = 0.f;
  gbb.t[0] = gbb.t[1] = gbb.t[2] = 0.f;
#pragma unroll
  for (int g = 0; g < 8; ++g) {
#pragma unroll
    for (int e = 0; e < 9; ++e) gch[g].r[e] = gf[g].r[e] = 0.f;
    gch[g].t[0] = gch[g].t[1] = gch[g].t[2] = 0.f;
    gf[g].t[0] = gf[g].t[1] = gf[g].t[2] = 0.f;
    compose_bwd(bb, ch[g], gG[g], gbb, gch[g]);
  }
  // chain in reverse: ch[7] = ch[6] o f[7];  ch[6] = ch[5] o f[6];  ch[5] = ch[4] o f[5];  ch[g] = f[g] otherwise
  compose_bwd(ch[6], f[7], gch[7], gch[6], gf[7]);
  compose_bwd(ch[5], f[6], gch[6], gch[5], gf[6]);
  compose_bwd(ch[4], f[5], gch[5], gch[4], gf[5]);
#pragma unroll
  for (int g = 0; g < 5; ++g) gf[g] = gch[g];
  // torsions: R[i][1] = r1 cs + r2 sn, R[i][2] = -r1 sn + r2 cs
#pragma unroll
  for (int g = 1; g < 8; ++g) {
    const float* m = s_df + (aa * 8 + g) * 16;
    float dsn = 0.f, dcs = 0.f;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const float r1 = m[4 * i + 1], r2 = m[4 * i + 2];
      dcs += gf[g].r[3 * i + 1] * r1 + gf[g].r[3 * i + 2] * r2;
      dsn += gf[g].r[3 * i + 1] * r2 - gf[g].r[3 * i + 2] * r1;
    }
    dangles[(p * 7 + g - 1) * 2] = dsn;
    dangles[(p * 7 + g - 1) * 2 + 1] = dcs;
  }
  // backbone frame: the quadratic quaternion form (no normalisation, like the forward)
  const float* G = gbb.r;
  dt7[p * 7 + 0] = 2 * (a * (G[0] + G[4] + G[8]) + d * (G[3] - G[1]) + c * (G[2] - G[6]) + b * (G[7] - G[5]));
  dt7[p * 7 + 1] = 2 * (b * (G[0] - G[4] - G[8]) + c * (G[1] + G[3]) + d * (G[2] + G[6]) + a * (G[7] - G[5]));
  dt7[p * 7 + 2] = 2 * (c * (-G[0] + G[4] - G[8]) + b * (G[1] + G[3]) + a * (G[2] - G[6]) + d * (G[5] + G[7]));
  dt7[p * 7 + 3] = 2 * (d * (-G[0] - G[4] + G[8]) + a * (G[3] - G[1]) + b * (G[2] + G[6]) + c * (G[5] + G[7]));
  dt7[p * 7 + 4] = gbb.t[0];
  dt7[p * 7 + 5] = gbb.t[1];
  dt7[p * 7 + 6] = gbb.t[2];
}

extern "C" int dfold_frames_to_atoms_bwd(const float* t7, const float* angles, const int64_t* aatype, const float* default_frames,
                                         const int64_t* atom14_group, const float* atom14_mask, const float* atom14_pos,
                                         const int64_t* atom37_to_atom14, const float* atom37_mask, const float* g_atom14,
                                         const float* g_atom37, float* d_t7, float* d_angles, int64_t P, void* stream) {
  if (!t7 || !angles || !aatype || !default_frames || !atom14_group || !atom14_mask || !atom14_pos || !atom37_to_atom14 ||
      !atom37_mask || (!g_atom14 && !g_atom37) || !d_t7 || !d_angles || P <= 0)
    return DFOLD_EINVAL;
  DFOLD_LAUNCH(frames_to_atoms_bwd_kernel, dim3((unsigned)((P + 127) / 128)), dim3(128), 0, (hipStream_t)stream, t7, angles,
               (const long*)aatype, default_frames, (const long*)atom14_group, atom14_mask, atom14_pos,
               (const long*)atom37_to_atom14, atom37_mask, g_atom14, g_atom37, d_t7, d_angles, (long)P);
  return dfold_check_launch();
}

