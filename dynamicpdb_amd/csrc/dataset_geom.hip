// Dataset-side geometry on device: atom37 coordinates -> the 8 rigid-group frames and the 7 torsion
// angles of every residue (reference openfold/data/data_transforms.py:755-893 atom37_to_frames, :923-1088
// atom37_to_torsion_angles, Rigid.from_3_points openfold/utils/rigid_utils.py:1233-1275), which the reference evaluates
// per item in forked DataLoader workers on float64 CPU tensors (src/data/Dfold_data_loader_dynamic.py:229-240).
// One thread per residue, float64 arithmetic; frames leave as fp32 4x4 matrices (the reference's Rotation / Rigid
// classes hold fp32), masks and torsion sin/cos as float64.  The index tables (base atoms of each rigid group, chi atoms,
// masks, ambiguity flags) are the reference's residue constants, passed in as device arrays.
#include "dfold_common.h"
#include "../../include/dfold_hip.h"

struct V3 {
  double x, y, z;
};
__device__ __forceinline__ V3 ld3(const double* p) { return V3{p[0], p[1], p[2]}; }
__device__ __forceinline__ V3 sub(V3 a, V3 b) { return V3{a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ double dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ V3 scale(V3 a, double s) { return V3{a.x * s, a.y * s, a.z * s}; }

// Gram-Schmidt frame: columns e0, e1, e2 (rigid_utils.py:1255-1268), eps inside the square roots
__device__ __forceinline__ void frame3(V3 p_neg_x, V3 origin, V3 p_xy, double eps, V3& e0, V3& e1, V3& e2) {
  e0 = sub(origin, p_neg_x);
  e1 = sub(p_xy, origin);
  e0 = scale(e0, 1.0 / sqrt(dot(e0, e0) + eps));
  e1 = sub(e1, scale(e0, dot(e0, e1)));
  e1 = scale(e1, 1.0 / sqrt(dot(e1, e1) + eps));
  e2 = V3{e0.y * e1.z - e0.z * e1.y, e0.z * e1.x - e0.x * e1.z, e0.x * e1.y - e0.y * e1.x};
}

__global__ __launch_bounds__(128) void atom37_geom_kernel(
    const long* __restrict__ aatype, const double* __restrict__ pos, const double* __restrict__ mask,
    const long* __restrict__ group_base, const float* __restrict__ group_mask, const float* __restrict__ group_amb,
    const long* __restrict__ chi_atoms, const float* __restrict__ chi_mask, const float* __restrict__ chi_pi,
    float* __restrict__ frames, float* __restrict__ alt_frames, double* __restrict__ gt_exists,
    double* __restrict__ group_exists, double* __restrict__ group_is_amb, double* __restrict__ tors, double* __restrict__ alt_tors,
    double* __restrict__ tors_mask, long P, int N, double eps) {
  const long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= P) return;
  const long aa = aatype[p];
  const double* x = pos + p * 37 * 3;
  const double* m = mask + p * 37;
  // ---- rigid-group frames ----
  if (frames != nullptr) {
    for (int k = 0; k < 8; ++k) {
      const long* idx = group_base + (aa * 8 + k) * 3;
      V3 e0, e1, e2;
      const V3 org = ld3(x + idx[1] * 3);
      frame3(ld3(x + idx[0] * 3), org, ld3(x + idx[2] * 3), eps, e0, e1, e2);
      if (k == 0) {  // backbone group: compose with diag(-1, 1, -1)
        e0 = scale(e0, -1.0);
        e2 = scale(e2, -1.0);
      }
      const double ge = (double)group_mask[aa * 8 + k];
      const double amb = (double)group_amb[aa * 8 + k];
      const double sg = 1.0 - 2.0 * amb;   // ambiguous groups: alternative frame = frame * diag(1, -1, -1)
      float* f = frames + (p * 8 + k) * 16;
      float* a = alt_frames + (p * 8 + k) * 16;
      f[0] = (float)e0.x; f[1] = (float)e1.x; f[2] = (float)e2.x; f[3] = (float)org.x;
      f[4] = (float)e0.y; f[5] = (float)e1.y; f[6] = (float)e2.y; f[7] = (float)org.y;
      f[8] = (float)e0.z; f[9] = (float)e1.z; f[10] = (float)e2.z; f[11] = (float)org.z;
      f[12] = 0.f; f[13] = 0.f; f[14] = 0.f; f[15] = 1.f;
      a[0] = f[0]; a[1] = (float)(e1.x * sg); a[2] = (float)(e2.x * sg); a[3] = f[3];
      a[4] = f[4]; a[5] = (float)(e1.y * sg); a[6] = (float)(e2.y * sg); a[7] = f[7];
      a[8] = f[8]; a[9] = (float)(e1.z * sg); a[10] = (float)(e2.z * sg); a[11] = f[11];
      a[12] = 0.f; a[13] = 0.f; a[14] = 0.f; a[15] = 1.f;
      gt_exists[p * 8 + k] = fmin(fmin(m[idx[0]], m[idx[1]]), m[idx[2]]) * ge;
      group_exists[p * 8 + k] = ge;
      group_is_amb[p * 8 + k] = amb;
    }
  }
  // ---- torsion angles: pre-omega, phi, psi, chi1..4 ----
  if (tors != nullptr) {
    const long ac = aa > 20 ? 20 : aa;
    const bool has_prev = (p % N) != 0;   // the chain restarts with every row of N residues
    const double* xp = x - 37 * 3;
    const double* mp = m - 37;
    const V3 zero{0.0, 0.0, 0.0};
    for (int t = 0; t < 7; ++t) {
      V3 a0, a1, a2, a3;
      double tm;
      if (t == 0) {        // prev CA, prev C, N, CA
        a0 = has_prev ? ld3(xp + 3) : zero; a1 = has_prev ? ld3(xp + 6) : zero; a2 = ld3(x); a3 = ld3(x + 3);
        tm = (has_prev ? mp[1] * mp[2] : 0.0) * (m[0] * m[1]);
      } else if (t == 1) { // prev C, N, CA, C
        a0 = has_prev ? ld3(xp + 6) : zero; a1 = ld3(x); a2 = ld3(x + 3); a3 = ld3(x + 6);
        tm = (has_prev ? mp[2] : 0.0) * (m[0] * m[1] * m[2]);
      } else if (t == 2) { // N, CA, C, O
        a0 = ld3(x); a1 = ld3(x + 3); a2 = ld3(x + 6); a3 = ld3(x + 12);
        tm = (m[0] * m[1] * m[2]) * m[4];
      } else {
        const long* ci = chi_atoms + (ac * 4 + (t - 3)) * 4;
        a0 = ld3(x + ci[0] * 3); a1 = ld3(x + ci[1] * 3); a2 = ld3(x + ci[2] * 3); a3 = ld3(x + ci[3] * 3);
        tm = (double)chi_mask[ac * 4 + (t - 3)] * (m[ci[0]] * m[ci[1]] * m[ci[2]] * m[ci[3]]);
      }
      V3 e0, e1, e2;
      frame3(a1, a2, a0, 1e-8, e0, e1, e2);          // from_3_points(p_neg_x = atom 1, origin = atom 2, p_xy = atom 0)
      const V3 r = sub(a3, a2);
      double s = dot(e2, r), c = dot(e1, r);          // fourth atom in the frame: (sin, cos) = (z, y)
      const double inv = 1.0 / sqrt(s * s + c * c + 1e-8);
      s *= inv;
      c *= inv;
      if (t == 2) {                                   // psi is measured to O: flipped
        s = -s;
        c = -c;
      }
      const double mir = t < 3 ? 1.0 : 1.0 - 2.0 * (double)chi_pi[ac * 4 + (t - 3)];
      tors[(p * 7 + t) * 2] = s;
      tors[(p * 7 + t) * 2 + 1] = c;
      alt_tors[(p * 7 + t) * 2] = s * mir;
      alt_tors[(p * 7 + t) * 2 + 1] = c * mir;
      tors_mask[p * 7 + t] = tm;
    }
  }
}

extern "C" int dfold_atom37_geometry(const int64_t* aatype, const double* all_atom_positions, const double* all_atom_mask,
                                     const int64_t* group_base_atom37, const float* group_mask, const float* group_ambiguous,
                                     const int64_t* chi_atom37, const float* chi_mask, const float* chi_pi_periodic,
                                     float* gt_frames, float* alt_gt_frames, double* gt_exists, double* group_exists,
                                     double* group_is_ambiguous, double* torsion_sin_cos, double* alt_torsion_sin_cos,
                                     double* torsion_mask, int64_t P, int32_t N, double eps, void* stream) {
  if (!aatype || !all_atom_positions || !all_atom_mask || P <= 0 || N <= 0 || (P % N) != 0) return DFOLD_EINVAL;
  const bool want_frames = gt_frames != nullptr, want_tors = torsion_sin_cos != nullptr;
  if (!want_frames && !want_tors) return DFOLD_EINVAL;
  if (want_frames && (!alt_gt_frames || !gt_exists || !group_exists || !group_is_ambiguous || !group_base_atom37 ||
                      !group_mask || !group_ambiguous))
    return DFOLD_EINVAL;
  if (want_tors && (!alt_torsion_sin_cos || !torsion_mask || !chi_atom37 || !chi_mask || !chi_pi_periodic)) return DFOLD_EINVAL;
  DFOLD_LAUNCH(atom37_geom_kernel, dim3((unsigned)((P + 127) / 128)), dim3(128), 0, (hipStream_t)stream, (const long*)aatype,
               all_atom_positions, all_atom_mask, (const long*)group_base_atom37, group_mask, group_ambiguous,
               (const long*)chi_atom37, chi_mask, chi_pi_periodic, gt_frames, alt_gt_frames, gt_exists, group_exists,
               group_is_ambiguous, torsion_sin_cos, alt_torsion_sin_cos, torsion_mask, (long)P, N, eps);
  return dfold_check_launch();
}
