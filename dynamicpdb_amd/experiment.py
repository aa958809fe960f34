"""Training-step host logic: the slice of the reference's Experiment (train_DFOLD_dynamics.py:343-1568) that is
on the hot path -- loss_fn (:1182-1400, live terms only), update_fn (:660-667) and data-parallel gradient
averaging (DDP at :615) -- over a real window batch axis.  One process per GPU; gradients are averaged with
torch.distributed (backend "nccl" = RCCL over xGMI on MI355X, "gloo" in the CPU tests)."""
import torch
import torch.distributed as dist


def torsion_angle_loss(a, a_gt, a_alt_gt, mask):
    """openfold/utils/loss.py:52-76 as called at train_DFOLD_dynamics.py:1219-1224 (angle-norm weight 0)."""
    norm = torch.linalg.norm(a, dim=-1)
    a = a / (norm.unsqueeze(-1) + 1e-8)
    d_gt = ((a - a_gt) ** 2).sum(-1)
    d_alt = ((a - a_alt_gt) ** 2).sum(-1)
    m = torch.minimum(d_gt, d_alt)
    return (m * mask).sum(dim=(-1, -2)) / (mask.sum(dim=(-1, -2)) + 1e-2)


class LossLastFrameFn(torch.autograd.Function):
    """loss_fn on the device as ONE HIP launch (csrc/loss.hip, dfold_loss_last_frame): the live terms read the last frame of a
    window only (train_DFOLD_dynamics.py:1219-1340), so the node takes the last-frame slices, gets the per-window terms and the
    gradients w.r.t. (angles, x0 translations, rotation scores) back from the kernel and scatters the gradients into zero
    tensors in its backward.  The aten graph it replaces: ~55 launches forward, ~80 backward."""

    @staticmethod
    def forward(ctx, angles, rigids, rot_score, batch, trans_w, rot_w, torsion_w, rot_t_threshold):
        from . import _lib
        from ctypes import c_float, c_int32, c_void_p
        L = _lib.lib()
        B, Fr, N = angles.shape[0], angles.shape[1], angles.shape[2]
        p = lambda x: c_void_p(x.data_ptr())
        f32 = lambda x: x.detach().to(torch.float32).contiguous()
        f64 = lambda x: x.detach().to(torch.float64).contiguous()
        bb = batch['res_mask'].to(torch.float32)
        dm = 1 - batch['fixed_mask'].to(torch.float32)
        ang = f32(angles[:, -1])
        tr = f32(rigids[:, -1, :, 4:])
        rot = f64(rot_score[:, -1])
        a_gt, a_alt = f32(batch['torsion_angles_sin_cos'][:, -1]), f32(batch['alt_torsion_angles_sin_cos'][:, -1])
        a_m = f32(batch['torsion_angles_mask'][:, -1])
        tr_gt, rot_gt = f32(batch['rigids_0'][:, -1, :, 4:]), f64(batch['rot_score'][:, -1])
        dml, lml = dm[:, -1].contiguous(), (bb[:, -1] * dm[:, -1]).contiguous()
        rss, tt = f64(batch['rot_score_scaling'].reshape(B)), f32(batch['t'].reshape(B))
        live = torch.any(bb > 0, dim=-1).sum(-1).to(torch.float32).contiguous()
        dev = angles.device
        terms = torch.empty((B, 4), dtype=torch.float64, device=dev)
        d_ang, d_tr, d_rot = torch.empty_like(ang), torch.empty_like(tr), torch.empty_like(rot)
        _lib.check(L.dfold_loss_last_frame(p(ang), p(a_gt), p(a_alt), p(a_m), p(tr), p(tr_gt), p(rot), p(rot_gt), p(dml), p(lml), p(rss),
                                           p(tt), p(live), p(terms), p(d_ang), p(d_tr), p(d_rot), c_int32(B), c_int32(N), c_int32(Fr),
                                           c_float(trans_w), c_float(rot_w), c_float(torsion_w), c_float(rot_t_threshold),
                                           _lib.stream()), "dfold_loss_last_frame")
        ctx.save_for_backward(d_ang, d_tr, d_rot)
        ctx.shapes = (angles.shape, angles.dtype, rigids.shape, rigids.dtype, rot_score.shape, rot_score.dtype)
        m = terms.mean(0)
        ctx.mark_non_differentiable(m)
        return m[0].clone(), m

    @staticmethod
    def backward(ctx, g_loss, _g_terms):
        d_ang, d_tr, d_rot = ctx.saved_tensors
        sa, ta, sr, tr_, so, to = ctx.shapes
        g_a = g_r = g_o = None
        if ctx.needs_input_grad[0]:
            g_a = torch.zeros(sa, dtype=ta, device=d_ang.device)
            g_a[:, -1] = (d_ang * g_loss).to(ta)
        if ctx.needs_input_grad[1]:
            g_r = torch.zeros(sr, dtype=tr_, device=d_tr.device)
            g_r[:, -1, :, 4:] = (d_tr * g_loss).to(tr_)
        if ctx.needs_input_grad[2]:
            g_o = torch.zeros(so, dtype=to, device=d_rot.device)
            g_o[:, -1] = (d_rot * g_loss).to(to)
        return g_a, g_r, g_o, None, None, None, None, None


_LOSS_FUSED = __import__("os").environ.get("DFOLD_LOSS_FUSED", "1") != "0"


def loss_fn(out, batch, trans_w=100.0, rot_w=7.0, torsion_w=1.0, rot_t_threshold=0.0):
    """Live loss terms for a [B,F,N,..] batch: last-frame torsion / translation-x0 / rotation-score losses, each
    repeated over the F frames of its window, gated by trans_loss < 100 (train_DFOLD_dynamics.py:1210-1400).
    Returns (mean over windows of the reference's per-window loss, aux dict).  Device tensors: one HIP launch
    (LossLastFrameFn; DFOLD_LOSS_FUSED=0 keeps the aten graph below, which is also what host tensors -- the gloo tests, the
    oracle comparisons -- run)."""
    if _LOSS_FUSED and out['angles'].is_cuda:
        loss, m = LossLastFrameFn.apply(out['angles'], out['rigids'], out['rot_score'], batch, float(trans_w), float(rot_w),
                                        float(torsion_w), float(rot_t_threshold))
        return loss, dict(rot_loss=m[1], trans_loss=m[2], torsion_loss=m[3])
    dt = out['rigids'].dtype
    bb_mask = batch['res_mask'].to(dt)                       # [B,F,N]
    diffuse_mask = 1 - batch['fixed_mask'].to(dt)
    loss_mask = bb_mask * diffuse_mask
    B, Fr, _ = bb_mask.shape
    tl = torsion_angle_loss(out['angles'], batch['torsion_angles_sin_cos'].to(dt),
                            batch['alt_torsion_angles_sin_cos'].to(dt), batch['torsion_angles_mask'].to(dt)) * torsion_w
    torsion = tl[:, -1:].expand(B, Fr)
    gt_x0, pr_x0 = batch['rigids_0'][..., 4:].to(dt), out['rigids'][..., 4:]
    trans = ((gt_x0[:, -1:] - pr_x0[:, -1:]) ** 2).mean(dim=(-1, -2)).expand(B, Fr) * trans_w
    pr_rot = out['rot_score'] * diffuse_mask[..., None]
    rot_mse = (batch['rot_score'] - pr_rot) ** 2 * loss_mask[..., None]
    rss = batch['rot_score_scaling'].reshape(B, 1, 1, 1)
    rot = (rot_mse / rss ** 2).sum(dim=(-1, -2)) / (loss_mask.sum(-1) + 1e-10)
    rot = rot * rot_w
    rot = rot * (batch['t'].reshape(B, 1) > rot_t_threshold)
    rot = rot[:, -1:].expand(B, Fr)
    gate = (trans < 100.0)
    rot = rot * gate.to(rot.dtype)
    trans_g = trans * gate.to(trans.dtype)
    torsion = torsion * (trans_g < 100.0).to(torsion.dtype)
    final = rot + trans_g + torsion                          # [B,F]
    bmask = torch.any(bb_mask > 0, dim=-1)                   # [B,F]
    norm = lambda x: (x.sum(-1) / (bmask.sum(-1) + 1e-10)).mean()
    return norm(final), dict(rot_loss=norm(rot), trans_loss=norm(trans_g), torsion_loss=norm(torsion))


class Trainer:
    """update_fn of the reference (zero_grad, loss_fn, backward, optimizer step; train_DFOLD_dynamics.py:660-667) for one
    rank's shard of windows, with gradient averaging across ranks overlapped with backward (dp.GradReducer; the reference
    uses DistributedDataParallel, :615).  Adam(amsgrad=True, lr) as train_DFOLD_dynamics.py:412."""

    def __init__(self, model, lr=1e-4, loss_kwargs=None, bucket_bytes=32 << 20, last_frame_only=True, force_reduce=False,
                 sync_params=True, grad_payload_dtype=None):
        """last_frame_only: run the model in its training-step mode (the live loss terms of loss_fn and the frame
        updates read the last frame of each window only, so the conv tower evaluates just that frame's dependency
        cone; identical loss and gradients, see DFOLDIpaScore.forward).  False = every frame, as the reference.
        force_reduce: run the gradient collectives even in a single-rank world (coverage of the multi-GPU path on one
        GPU).  sync_params: in a multi-rank world, start from rank 0's parameters and buffers like the reference's
        DistributedDataParallel wrap does (train_DFOLD_dynamics.py:615; every rank is seeded differently, :419).
        grad_payload_dtype: torch.bfloat16 halves the gradient bytes on the wire (dp.GradReducer, opt-in; default fp32)."""
        from .dp import GradReducer, broadcast_parameters
        self.model = model
        self.bytes_broadcast = 0
        if sync_params:
            self.bytes_broadcast = broadcast_parameters(list(model.parameters()) + list(model.buffers()))
        self.last_frame_only = last_frame_only
        self.params = [p for p in model.parameters() if p.requires_grad]
        if self.params and self.params[0].is_cuda:
            from .optim import FusedAdam       # same update rule and state layout, one HIP launch per step
            self.opt = FusedAdam(self.params, lr=lr)
        else:
            self.opt = torch.optim.Adam(self.params, lr=lr, amsgrad=True)
        self.loss_kwargs = loss_kwargs or {}
        self.reducer = GradReducer(self.params, bucket_bytes=bucket_bytes, force=force_reduce,
                                   payload_dtype=grad_payload_dtype).attach(model)
        self.world = self.reducer.world

    def begin_step(self):
        """zero_grad + re-arm the per-step state (gradient buckets, the shared conv tower's accumulators)"""
        for m in self.model.modules():
            t = getattr(m, "_tower", None)
            if t is not None:
                t.reset_step()
        self.reducer.begin_step()

    def update_fn(self, batch, step_optimizer=True):
        self.begin_step()
        out = self.model(batch, last_frame_only=self.last_frame_only)
        loss, aux = loss_fn(out, batch, **self.loss_kwargs)
        loss.backward()                 # buckets are reduced as their gradients complete (hooks / conv tower)
        self.reducer.finish()
        if step_optimizer:
            self.opt.step()
        return loss.detach(), aux

    def sync_from_rank0(self):
        """Every rank takes rank 0's parameters, buffers AND optimizer state (exp_avg / exp_avg_sq / max_exp_avg_sq /
        step): call after checkpoint.resume() on rank 0 only, or whenever ranks may have diverged.  The optimizer state
        must exist on every rank first (a resumed rank has it; a fresh one gets zero state of the right shapes)."""
        from .dp import broadcast_parameters
        ts = list(self.model.parameters()) + list(self.model.buffers())
        steps = []
        for p in self.params:
            st = self.opt.state[p]
            if len(st) == 0:
                st["step"] = torch.tensor(0.0, dtype=torch.float32)
                for k in ("exp_avg", "exp_avg_sq", "max_exp_avg_sq"):
                    st[k] = torch.zeros_like(p, memory_format=torch.preserve_format)
            steps.append(float(st["step"]))
            ts += [st["exp_avg"], st["exp_avg_sq"], st["max_exp_avg_sq"]]
        dev = self.params[0].device if self.params else torch.device("cpu")
        step_t = torch.tensor(steps, dtype=torch.float64, device=dev)     # the step counters live on the host: one vector
        n = broadcast_parameters(ts + [step_t])
        for p, v in zip(self.params, step_t.cpu().tolist()):
            st = self.opt.state[p]
            if torch.is_tensor(st["step"]):
                st["step"].fill_(v)
            else:
                st["step"] = v
        if self.params and self.params[0].is_cuda:
            torch.autograd.graph.increment_version(self.params)    # the bf16 weight caches key on the version counter
        return n


def set_t_feats(diffuser, feats, t, like):
    """Experiment._set_t_feats (train_DFOLD_dynamics.py:1408-1413) for a batch of windows sharing one t."""
    import numpy as np
    rs, ts = diffuser.score_scaling(t)
    feats['t'] = torch.full_like(like, float(t))
    feats['t_host'] = np.full(tuple(like.shape), float(t))      # (the forward then needs no device -> host copy of t)
    from .model.ipa_pytorch_dynamic import stamp_t_host
    stamp_t_host(feats)            # valid for exactly this 't' tensor: a later `feats['t'] = ...` or in-place write voids the copy
    feats['rot_score_scaling'] = torch.full_like(like, float(rs))
    feats['trans_score_scaling'] = torch.full_like(like, float(ts))
    return feats


def inference_fn(model, diffuser, data_init, num_t=10, min_t=0.01, center=True, aux_traj=False, self_condition=True,
                 noise_scale=1.0, cfg_drop_rate=0.0, cfg_gamma=2.0, z_draws=None, rng=None):
    """Reverse-diffusion sampler, device resident (reference Experiment.inference_fn, train_DFOLD_dynamics.py:1425-1547):
    num_t model forwards; between them one dfold_se3_reverse launch instead of the reference's host round trip
    (D2H, scipy, numpy RNG, H2D, CPU eigh).  `data_init` holds [B,F,N,..] (or reference-shaped [F,N,..]) device tensors
    incl. the prior sample 'rigids_t'.  `z_draws`: optional iterable of (z_rot, z_trans) per reverse step (parity tests);
    `rng`: a dynamicpdb_amd.rng.DeviceRNG -- the draws are generated on the device (Philox4x32-10), nothing crosses
    PCIe inside the loop; default = numpy global RNG in the reference's draw order (a seeded run then consumes the
    reference's random stream).  Returns numpy trajectories flipped to start at t = 0."""
    import numpy as np
    model.eval()
    feats = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in data_init.items()}
    like = feats['t'].float() if 't' in feats else torch.ones(1, device=feats['rigids_t'].device)
    reverse_steps = np.linspace(min_t, 1.0, num_t)[::-1]
    dt = 1.0 / num_t
    all_rigids, all_bb_prots, all_trans_0_pred, all_bb_0_pred = [], [], [], []
    z_iter = iter(z_draws) if z_draws is not None else None
    angles = rigid_pred = None
    with torch.no_grad():
        if self_condition:
            feats = set_t_feats(diffuser, feats, reverse_steps[0], like)
            feats['sc_ca_t'] = model(feats)['rigids'][..., 4:]
        for t in reverse_steps:
            if t > min_t:
                feats = set_t_feats(diffuser, feats, t, like)
                model_out = model(feats)
                rot_score, trans_score, rigid_pred = model_out['rot_score'], model_out['trans_score'], model_out['rigids']
                if cfg_drop_rate > 0.01:
                    unref = model(feats, drop_ref=True)['trans_score']
                    trans_score = unref + cfg_gamma * (trans_score - unref)
                feats['sc_ca_t'] = rigid_pred[..., 4:]
                diffuse_mask = (1 - feats['fixed_mask'].float()) * feats['res_mask'].float()
                zr, zt = next(z_iter) if z_iter is not None else (None, None)
                feats['rigids_t'] = diffuser.reverse_t7(feats['rigids_t'], rot_score, trans_score, float(t), dt,
                                                        diffuse_mask=diffuse_mask, center=center, noise_scale=noise_scale,
                                                        z_rot=zr, z_trans=zt, rng=rng)
            else:
                # last step (t == min_t): the state becomes the network's own frame prediction (:1502-1504).  Like the
                # reference, `rigid_pred` keeps the PREVIOUS step's prediction here (it is only assigned in the branch
                # above), which is what the x0 traces below are built from.
                model_out = model(feats)
                feats['rigids_t'] = model_out['rigids'].clone()
            fixed_mask = feats['fixed_mask'].float() * feats['res_mask'].float()
            diffuse_mask = (1 - feats['fixed_mask'].float()) * feats['res_mask'].float()
            angles = model_out['angles']
            if aux_traj:
                all_rigids.append(model_out['rigids'].cpu().numpy())
                trans_pred_0 = diffuse_mask[..., None] * rigid_pred[..., 4:] + fixed_mask[..., None] * feats['rigids_t'][..., 4:]
                all_trans_0_pred.append(trans_pred_0.cpu().numpy())
                # all_atom.compute_backbone_atom37(rigid_pred, aatype, angles) (:1515-1520): all atoms of the predicted frames
                from .model import geometry as G
                all_bb_0_pred.append(G.frames_to_atoms_hip(rigid_pred, angles, feats['aatype'])[1].cpu().numpy())
            all_bb_prots.append(model_out['atom37'].cpu().numpy())
    flip = lambda x: np.flip(np.stack(x), (0,))
    ret = {'prot_traj': flip(all_bb_prots)}
    if aux_traj:
        ret.update(rigid_traj=flip(all_rigids), trans_traj=flip(all_trans_0_pred), psi_pred=angles,
                   rigid_0_traj=flip(all_bb_0_pred))
    return ret
