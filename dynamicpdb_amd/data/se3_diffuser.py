"""SE(3) diffuser = IGSO(3) x VP-SDE.  API mirror of the reference's
src/data/se3_diffuser.py (SE3Diffuser :31-280): forward_marginal, reverse,
sample_ref, calc_rot_score, calc_trans_score, score_scaling keep their names,
arguments, return dict keys and ValueErrors.  The two score heads evaluated
inside the network forward run as HIP kernels on device tensors."""
import logging

import numpy as np
import torch
from scipy.spatial.transform import Rotation as _SciRot

from ..rigid import Rigid, Rotation
from . import r3_diffuser, so3_diffuser


def _extract_trans_rots(rigid: Rigid):
    rot = rigid.get_rots().get_rot_mats().detach().cpu().numpy()
    shp = rot.shape[:-2]
    rotvec = _SciRot.from_matrix(rot.reshape(-1, 3, 3)).as_rotvec().reshape(shp + (3,))
    return rigid.get_trans().detach().cpu().numpy(), rotvec


def _assemble_rigid(rotvec, trans, device=torch.device('cpu')):
    shp = rotvec.shape[:-1]
    rotmat = _SciRot.from_rotvec(rotvec.reshape(-1, 3)).as_matrix().reshape(shp + (3, 3))
    return Rigid(Rotation(rot_mats=torch.Tensor(rotmat).to(device)), torch.tensor(trans).to(device))


class SE3Diffuser:
    def __init__(self, se3_conf):
        self._log = logging.getLogger(__name__)
        self._se3_conf = se3_conf
        self._diffuse_rot = se3_conf.diffuse_rot
        self._so3_diffuser = so3_diffuser.SO3Diffuser(se3_conf.so3)
        self._diffuse_trans = se3_conf.diffuse_trans
        self._r3_diffuser = r3_diffuser.R3Diffuser(se3_conf.r3)

    @staticmethod
    def _apply_mask(x_diff, x_fixed, diff_mask):
        return diff_mask * x_diff + (1 - diff_mask) * x_fixed

    def forward_marginal(self, rigids_0: Rigid, t: float, diffuse_mask=None, as_tensor_7=True):
        trans_0, rot_0 = _extract_trans_rots(rigids_0)
        if self._diffuse_rot:
            rot_t, rot_score = self._so3_diffuser.forward_marginal(rot_0, t)
            rot_score_scaling = self._so3_diffuser.score_scaling(t)
        else:
            rot_t, rot_score, rot_score_scaling = rot_0, np.zeros_like(rot_0), np.ones_like(t)
        if self._diffuse_trans:
            trans_t, trans_score = self._r3_diffuser.forward_marginal(trans_0, t)
            trans_score_scaling = self._r3_diffuser.score_scaling(t)
        else:
            trans_t, trans_score, trans_score_scaling = trans_0, np.zeros_like(trans_0), np.ones_like(t)
        if diffuse_mask is not None:
            m = diffuse_mask[..., None]
            rot_t = self._apply_mask(rot_t, rot_0, m)
            trans_t = self._apply_mask(trans_t, trans_0, m)
            trans_score = self._apply_mask(trans_score, np.zeros_like(trans_score), m)
            rot_score = self._apply_mask(rot_score, np.zeros_like(rot_score), m)
        rigids_t = _assemble_rigid(rot_t, trans_t)
        if as_tensor_7:
            rigids_t = rigids_t.to_tensor_7()
        return {'rigids_t': rigids_t, 'trans_score': trans_score, 'rot_score': rot_score,
                'trans_score_scaling': trans_score_scaling, 'rot_score_scaling': rot_score_scaling}

    def calc_trans_0(self, trans_score, trans_t, t):
        return self._r3_diffuser.calc_trans_0(trans_score, trans_t, t)

    def calc_trans_score(self, trans_t, trans_0, t, use_torch=False, scale=True):
        return self._r3_diffuser.score(trans_t, trans_0, t, use_torch=use_torch, scale=scale)

    def calc_rot_score(self, rots_t: Rotation, rots_0: Rotation, t):
        """score of q_0^{-1} (x) q_t as a rotation vector (reference :119-125)."""
        quats_t, quats_0 = rots_t.get_quats(), rots_0.get_quats()
        if quats_t.is_cuda:
            return self.calc_rot_score_t7(quats_t, quats_0, t)
        from .. import host_math
        return self._so3_diffuser.torch_score(host_math.quat_to_rotvec(
            host_math.quat_multiply(host_math.invert_quat(quats_0), quats_t)), t)

    def calc_rot_score_t7(self, quats_t, quats_0, t, t_host=None):
        """Device path on raw quaternion tensors; the leading axis enumerates the windows of `t`
        ([F,N,4] with t [1] as in the reference, or [B,F,N,4] with t [B]).  t_host: the same diffusion times as host numbers
        (the sigma of the series is a table look-up on the host, so3_diffuser.py:188-190) -- given by a caller that knows them
        (the sampler, a data loader), it saves the device -> host copy of `t`, i.e. one stream synchronisation per forward,
        and makes the forward capturable in a HIP graph."""
        from ..model import score_heads
        if t_host is not None:
            t_np = np.atleast_1d(np.asarray(t_host, dtype=np.float64))
        else:
            t_np = np.atleast_1d(t.detach().cpu().numpy() if torch.is_tensor(t) else np.asarray(t))
        so3 = self._so3_diffuser
        sigma = so3.discrete_sigma[so3.t_to_idx(t_np)]
        if len(sigma) == 1 and quats_t.dim() == 3:
            return score_heads.rot_score(quats_t[None], quats_0[None], sigma)[0]
        return score_heads.rot_score(quats_t, quats_0, sigma)

    def score(self, rigid_0: Rigid, rigid_t: Rigid, t: float):
        tran_0, rot_0 = _extract_trans_rots(rigid_0)
        tran_t, rot_t = _extract_trans_rots(rigid_t)
        rot_score = self._so3_diffuser.score(rot_t, t) if self._diffuse_rot else np.zeros_like(rot_0)
        trans_score = self._r3_diffuser.score(tran_t, tran_0, t) if self._diffuse_trans else np.zeros_like(tran_0)
        return trans_score, rot_score

    def score_scaling(self, t):
        return self._so3_diffuser.score_scaling(t), self._r3_diffuser.score_scaling(t)

    def reverse(self, rigid_t: Rigid, rot_score, trans_score, t, dt, diffuse_mask=None, center=True,
                noise_scale=1.0, device=torch.device('cpu'), z_rot=None, z_trans=None):
        """One reverse-SDE step t -> t-dt (reference :160-215).  z_rot / z_trans optionally
        inject the normal draws (parity tests); default = numpy global RNG as upstream."""
        if rigid_t.get_trans().is_cuda:      # device frames: one HIP launch, no host round trip
            return Rigid.from_tensor_7(self.reverse_t7(rigid_t.to_tensor_7(), rot_score, trans_score, t, dt, diffuse_mask,
                                                       center, noise_scale, z_rot, z_trans))
        trans_t, rot_t = _extract_trans_rots(rigid_t)
        rot_t_1 = rot_t if not self._diffuse_rot else self._so3_diffuser.reverse(
            rot_t=rot_t, score_t=rot_score, t=t, dt=dt, noise_scale=noise_scale, z=z_rot)
        trans_t_1 = trans_t if not self._diffuse_trans else self._r3_diffuser.reverse(
            x_t=trans_t, score_t=trans_score, t=t, dt=dt, center=center, noise_scale=noise_scale, z=z_trans)
        if diffuse_mask is not None:
            trans_t_1 = self._apply_mask(trans_t_1, trans_t, diffuse_mask[..., None])
            rot_t_1 = self._apply_mask(rot_t_1, rot_t, diffuse_mask[..., None])
        return _assemble_rigid(rot_t_1, trans_t_1, device)

    def reverse_t7(self, rigids_t, rot_score, trans_score, t, dt, diffuse_mask=None, center=True, noise_scale=1.0,
                   z_rot=None, z_trans=None, rng=None):
        """Device form of `reverse` on tensor_7 frames [..., N, 7] (one HIP launch, csrc/diffusion.hip).  The normal draws
        default to numpy's global RNG in the reference's order (rotations first, se3_diffuser.py:184-190) so that a seeded
        run consumes the same random stream as the reference; pass device tensors z_rot / z_trans to inject them, or
        rng = dynamicpdb_amd.rng.DeviceRNG to draw them on the device (Philox4x32-10, no host round trip; rotations
        first as well)."""
        from ctypes import c_double, c_int32, c_int64
        from .. import _lib
        from ..ops import _p
        if not np.isscalar(t):
            raise ValueError(f'{t} must be a scalar.')
        dev = rigids_t.device
        t7 = rigids_t.detach().float().contiguous()
        N = t7.shape[-2]
        rows = t7.numel() // (7 * N)
        shp3 = tuple(t7.shape[:-1]) + (3,)
        if z_rot is None:
            z_rot = rng.normal(shp3) if rng is not None else np.random.normal(size=shp3)
        if z_trans is None:
            z_trans = rng.normal(shp3) if rng is not None else np.random.normal(size=shp3)
        as_dev = lambda x, dt_: (x if torch.is_tensor(x) else torch.as_tensor(np.asarray(x))).to(device=dev, dtype=dt_).contiguous()
        zr, zt = as_dev(z_rot, torch.float64), as_dev(z_trans, torch.float64)
        rs, ts = as_dev(rot_score, torch.float64), as_dev(trans_score, torch.float32)
        if not self._diffuse_rot:
            rs, zr = torch.zeros_like(rs), torch.zeros_like(zr)
        mask = as_dev(diffuse_mask, torch.float32) if diffuse_mask is not None else None
        out = torch.empty_like(t7)
        r3 = self._r3_diffuser
        g_rot = float(self._so3_diffuser.diffusion_coef(t)) if self._diffuse_rot else 0.0
        _lib.check(_lib.lib().dfold_se3_reverse(
            _p(t7), _p(rs), _p(ts), _p(zr), _p(zt), _p(mask), _p(out), c_int64(rows), c_int32(N), c_double(g_rot),
            c_double(float(r3.b_t(t))), c_double(float(dt)), c_double(float(noise_scale)),
            c_double(float(r3._r3_conf.coordinate_scaling)), c_int32(1 if center else 0), _lib.stream()), "dfold_se3_reverse")
        if not self._diffuse_trans:
            out[..., 4:] = t7[..., 4:]
        return out

    def forward_marginal_t7(self, rigids_0, t, diffuse_mask=None, u=None, z_dir=None, z_trans=None, rng=None):
        """Device form of `forward_marginal` on tensor_7 frames: [F,N,7] with scalar t (the reference's per-item call,
        Dfold_data_loader_dynamic.py:336-345) or [B,F,N,7] with t [B] (one t per window).  One HIP launch for the sampling
        (csrc/diffusion.hip) + the IGSO(3) series launch for the rotation score.  Draws default to numpy's global RNG in
        the reference's order per window (direction normals, then the uniform of the inverse CDF, so3_diffuser.py:233-248;
        then the translation normals, r3_diffuser.py:96-99); pass u / z_dir / z_trans to inject them.  The quaternion of
        rigids_t is q_0 (x) exp(v): the same rotation as the reference's matrix product, sign not canonicalised (the
        reference's comes out of eigh, rigid_utils.py:226)."""
        from ctypes import c_double, c_int32, c_int64
        from .. import _lib
        from ..ops import _p
        dev = rigids_0.device
        if dev.type != 'cuda':
            raise RuntimeError("forward_marginal_t7 needs device tensors; use forward_marginal for host Rigid objects")
        t7 = rigids_0.detach().float().contiguous()
        batched = t7.dim() == 4
        t_np = np.atleast_1d(t.detach().cpu().numpy() if torch.is_tensor(t) else np.asarray(t, dtype=np.float64)).astype(np.float64)
        W = t7.shape[0] if batched else 1
        if len(t_np) != W:
            raise ValueError(f'{t} must hold one value per window.')
        if np.any(t_np < 0) or np.any(t_np > 1):
            raise ValueError(f'Invalid t={t}')
        P = t7.numel() // 7
        per_window = P // W
        lead = tuple(t7.shape[:-1])
        if rng is not None:            # device draws (DeviceRNG): same order as the host path, one subsequence each
            z_dir = rng.normal((W, per_window, 3)) if z_dir is None else z_dir
            u = rng.uniform((W, per_window)) if u is None else u
            z_trans = rng.normal((W, per_window, 3)) if z_trans is None else z_trans
        if u is None or z_dir is None or z_trans is None:
            zd, uu, zt = [], [], []
            for _ in range(W):
                zd.append(np.random.randn(per_window, 3))
                uu.append(np.random.rand(per_window))
                zt.append(np.random.normal(size=(per_window, 3)))
            z_dir = np.stack(zd) if z_dir is None else z_dir
            u = np.stack(uu) if u is None else u
            z_trans = np.stack(zt) if z_trans is None else z_trans
        as_dev = lambda x, dt_: (x if torch.is_tensor(x) else torch.as_tensor(np.asarray(x))).to(device=dev, dtype=dt_).contiguous()
        u_d, zd_d, zt_d = as_dev(u, torch.float64), as_dev(z_dir, torch.float64), as_dev(z_trans, torch.float64)
        so3, r3 = self._so3_diffuser, self._r3_diffuser
        tabs = so3.device_tables(dev)
        idx = so3.t_to_idx(t_np)
        idx_d = torch.as_tensor(np.asarray(idx, dtype=np.int32)).to(dev)
        bt_d = torch.as_tensor(r3.marginal_b_t(t_np)).to(device=dev, dtype=torch.float64)
        mask = as_dev(diffuse_mask, torch.float32) if diffuse_mask is not None else None
        out = torch.empty_like(t7)
        rotvec = torch.empty(lead + (3,), dtype=torch.float64, device=dev)
        trans_score = torch.empty(lead + (3,), dtype=torch.float32, device=dev)
        _lib.check(_lib.lib().dfold_se3_forward_marginal(
            _p(t7), _p(u_d), _p(zd_d), _p(zt_d), _p(mask), _p(tabs['cdf']), _p(tabs['omega']), _p(idx_d), _p(bt_d), _p(out),
            _p(rotvec), _p(trans_score), c_int64(P), c_int64(per_window), c_int32(tabs['cdf'].shape[1]),
            c_double(float(r3._r3_conf.coordinate_scaling)), _lib.stream()), "dfold_se3_forward_marginal")
        from ..model import score_heads
        sigma = np.atleast_1d(so3.discrete_sigma[idx])
        rv = rotvec.float()
        rot_score = score_heads.igso3_score(rv if batched else rv[None], sigma)
        rot_score = rot_score if batched else rot_score[0]
        if mask is not None:
            rot_score = rot_score * mask.view(lead)[..., None]
        if not self._diffuse_rot:
            out[..., :4] = t7[..., :4]
            rot_score = torch.zeros_like(rot_score)
        if not self._diffuse_trans:
            out[..., 4:] = t7[..., 4:]
            trans_score = torch.zeros_like(trans_score)
        rss = so3.score_scaling(t_np) if self._diffuse_rot else np.ones_like(t_np)
        tss = r3.score_scaling(t_np) if self._diffuse_trans else np.ones_like(t_np)
        if not batched:
            rss, tss = rss[0], tss[0]
        return {'rigids_t': out, 'trans_score': trans_score, 'rot_score': rot_score,
                'trans_score_scaling': tss, 'rot_score_scaling': rss}

    def sample_ref(self, n_samples: int, impute: Rigid = None, diffuse_mask=None, as_tensor_7=False):
        if impute is not None:
            assert impute.shape[0] == n_samples
            trans_impute, rot_impute = _extract_trans_rots(impute)
            trans_impute = self._r3_diffuser._scale(trans_impute.reshape((n_samples, 3)))
            rot_impute = rot_impute.reshape((n_samples, 3))
        if diffuse_mask is not None and impute is None:
            raise ValueError('Must provide imputation values.')
        if (not self._diffuse_rot) and impute is None:
            raise ValueError('Must provide imputation values.')
        if (not self._diffuse_trans) and impute is None:
            raise ValueError('Must provide imputation values.')
        rot_ref = self._so3_diffuser.sample_ref(n_samples=n_samples) if self._diffuse_rot else rot_impute
        trans_ref = self._r3_diffuser.sample_ref(n_samples=n_samples) if self._diffuse_trans else trans_impute
        if diffuse_mask is not None:
            rot_ref = self._apply_mask(rot_ref, rot_impute, diffuse_mask[..., None])
            trans_ref = self._apply_mask(trans_ref, trans_impute, diffuse_mask[..., None])
        trans_ref = self._r3_diffuser._unscale(trans_ref)
        if not self._se3_conf.dynamics:
            rigids_t = _assemble_rigid(rot_ref, trans_ref)
            if as_tensor_7:
                rigids_t = rigids_t.to_tensor_7()
        else:
            F = self._se3_conf.frame_time
            rot_ref, trans_ref = rot_ref.reshape(F, -1, 3), trans_ref.reshape(F, -1, 3)
            rigids_t = torch.stack([_assemble_rigid(rot_ref[i], trans_ref[i]).to_tensor_7() for i in range(F)], dim=0)
        return {'rigids_t': rigids_t}
