"""Functional fp32 torch-CPU restatement of the grasp-generation graph.
TEST INFRASTRUCTURE: see oracle/__init__.py.

Everything here is a pure function of (state_dict, inputs): no nn.Module, no
hidden state.  `sd` is a flat {key: tensor} dict using the reference's
checkpoint key names, `p` a key prefix ending in ".".
Each function cites the reference lines it restates.  Point ops go through
oracle/cpu_backend.py.  Pinned by tests/golden/*.npz, which were captured from
the reference's own Python in the build container (oracle/make_golden.py).

fp32 only: the reference's eps in weight standardisation / LayerNorm is
dtype-dependent (resnets.py:86,110) and only the fp32 branch is restated.
"""
import math

import torch
import torch.nn.functional as F

from .cpu_backend import _backend as B
from . import schedulers

# --------------------------------------------------------------------------
# 1-D ResNet (denoiser / decoder core)          grasp_ldm/models/modules/resnets.py
# --------------------------------------------------------------------------


def _ws_conv1d(x, w, b, padding):
    """WeightStandardizedConv2d.forward, resnets.py:85-101 (fp32: eps 1e-5)."""
    mean = w.mean(dim=(1, 2), keepdim=True)
    var = w.var(dim=(1, 2), unbiased=False, keepdim=True)
    return F.conv1d(x, (w - mean) * (var + 1e-5).rsqrt(), b, padding=padding)


def _chan_layer_norm(x, g):
    """LayerNorm over the channel axis, gain only; resnets.py:104-113."""
    var = x.var(dim=1, unbiased=False, keepdim=True)
    mean = x.mean(dim=1, keepdim=True)
    return (x - mean) * (var + 1e-5).rsqrt() * g


def _block(sd, p, x, groups, scale_shift=None):
    """Block.forward, resnets.py:134-177."""
    x = _ws_conv1d(x, sd[p + "proj.weight"], sd[p + "proj.bias"], padding=1)
    x = F.group_norm(x, groups, sd[p + "norm.weight"], sd[p + "norm.bias"], eps=1e-5)
    if scale_shift is not None:
        scale, shift = scale_shift  # [B, C, R]
        if scale.shape[-1] == 1:
            x = x * (scale + 1) + shift
        else:  # tile over the R conditioning rows and sum (resnets.py:171-175)
            x = (x.unsqueeze(-1) * (scale.unsqueeze(-2) + 1) + shift.unsqueeze(-2)).sum(-1)
    return F.silu(x)


def _resnet_block(sd, p, x, emb, groups):
    """ResnetBlock.forward, resnets.py:193-208 (dim == dim_out -> identity skip)."""
    ss = None
    if emb is not None and (p + "mlp.1.weight") in sd:
        e = F.linear(F.silu(emb), sd[p + "mlp.1.weight"], sd[p + "mlp.1.bias"])
        e = e.unsqueeze(-1) if e.ndim == 2 else e.transpose(1, 2)  # [B, 2C, R]
        ss = e.chunk(2, dim=1)
    h = _block(sd, p + "block1.", x, groups, ss)
    h = _block(sd, p + "block2.", h, groups)
    if (p + "res_conv.weight") in sd:
        x = F.conv1d(x, sd[p + "res_conv.weight"], sd[p + "res_conv.bias"])
    return h + x


def _linear_attention(sd, p, x, heads=4):
    """Residual(PreNorm(LinearAttention)), resnets.py:59-65,116-124,211-235.
    p addresses the Residual module ("blocks.i.2.")."""
    b, c, n = x.shape
    y = _chan_layer_norm(x, sd[p + "fn.norm.g"])
    qkv = F.conv1d(y, sd[p + "fn.fn.to_qkv.weight"])
    q, k, v = (t.reshape(b, heads, -1, n) for t in qkv.chunk(3, dim=1))
    d = q.shape[2]
    q = q.softmax(dim=-2) * (d ** -0.5)
    k = k.softmax(dim=-1)
    context = torch.einsum("bhdn,bhen->bhde", k, v)
    out = torch.einsum("bhde,bhdn->bhen", context, q).reshape(b, heads * d, n)
    out = F.conv1d(out, sd[p + "fn.fn.to_out.0.weight"], sd[p + "fn.fn.to_out.0.bias"])
    return _chan_layer_norm(out, sd[p + "fn.fn.to_out.1.g"]) + x


def time_embedding(sd, p, time):
    """time_mlp = RandomOrLearnedSinusoidalPosEmb -> Linear -> GELU -> Linear,
    resnets.py:44-56,517-522.  `time` is int64 [B]; f = ((t*w)*2)*pi in f32."""
    t = time.reshape(-1, 1)
    freqs = t * sd[p + "time_mlp.0.weights"].reshape(1, -1) * 2 * math.pi
    four = torch.cat((t, freqs.sin(), freqs.cos()), dim=-1)
    h = F.linear(four, sd[p + "time_mlp.1.weight"], sd[p + "time_mlp.1.bias"])
    return F.linear(F.gelu(h), sd[p + "time_mlp.3.weight"], sd[p + "time_mlp.3.bias"])


def resnet1d_forward(sd, p, x, z_cond=None, time=None, groups=4, cls_cond=None):
    """TimeConditionedResNet1D.forward (resnets.py:558-616) when `time` is given,
    ResNet1D.forward (resnets.py:373-424) otherwise.  x [B,1,D]; z_cond [B,R,Dc]
    or [B,Dc]; eval mode (dropout = identity).  cls_cond [B,1]: the class-conditioned variant
    (class_conditioned_resnet.py:48-122): latent_emb += SiLU(Linear(1, emb)(cls_cond))."""
    x = F.conv1d(x, sd[p + "init_conv.weight"], sd[p + "init_conv.bias"], padding=3)
    emb = time_embedding(sd, p, time) if time is not None else None
    if cls_cond is not None:
        emb = emb + F.silu(F.linear(cls_cond.reshape(-1, 1).float(), sd[p + "cls_embed.0.weight"],
                                    sd[p + "cls_embed.0.bias"]))
    if (p + "input_emb_layers.0.weight") in sd:
        ie = F.silu(F.linear(z_cond, sd[p + "input_emb_layers.0.weight"], sd[p + "input_emb_layers.0.bias"]))
        if emb is not None and ie.ndim == 3:
            emb = emb.unsqueeze(-2).repeat(1, ie.shape[1], 1)
        emb = ie if emb is None else emb + ie
    i = 0
    while (p + f"blocks.{i}.3.weight") in sd:
        q = p + f"blocks.{i}."
        x = _resnet_block(sd, q + "0.", x, emb, groups)
        x = _resnet_block(sd, q + "1.", x, emb, groups)
        x = _linear_attention(sd, q + "2.", x)
        x = F.conv1d(x, sd[q + "3.weight"], sd[q + "3.bias"], padding=1)
        i += 1
    x = _resnet_block(sd, p + "final_res_block.", x, emb, groups)
    return F.conv1d(x, sd[p + "final_conv.weight"], sd[p + "final_conv.bias"])


def dpmpp_sample(sd, p, z_cond, noise, num_sample_steps=20, clamp=False, groups=4, sigma_min=0.002, sigma_max=80,
                 sigma_data=0.5, rho=7):
    """ElucidatedDiffusion.sample_using_dpmpp, elucidated_diffusion.py:259-313 (schedule :149-162,
    preconditioned_network_forward :117-139 with c_skip/c_out/c_in/c_noise :103-115).  noise = the unit-normal
    draw that the reference scales by sigmas[0] (:281)."""
    N = num_sample_steps
    inv_rho = 1 / rho
    steps = torch.arange(N, dtype=torch.float32)
    sigmas = (sigma_max ** inv_rho + steps / (N - 1) * (sigma_min ** inv_rho - sigma_max ** inv_rho)) ** rho
    sigmas = F.pad(sigmas, (0, 1), value=0.0)
    x = sigmas[0] * noise
    batch = x.shape[0]

    def precond(xx, sigma):
        sg = torch.full((batch,), sigma)
        ps = sg.reshape(-1, 1, 1)
        c_in = 1 * (ps ** 2 + sigma_data ** 2) ** -0.5
        c_skip = (sigma_data ** 2) / (ps ** 2 + sigma_data ** 2)
        c_out = ps * sigma_data * (sigma_data ** 2 + ps ** 2) ** -0.5
        c_noise = torch.log(sg.clamp(min=1e-20)) * 0.25
        net = resnet1d_forward(sd, p, c_in * xx, z_cond=z_cond, time=c_noise, groups=groups)
        out = c_skip * xx + c_out * net
        return out.clamp(-1.0, 1.0) if clamp else out

    sigma_fn = lambda t: t.neg().exp()
    t_fn = lambda sigma: sigma.log().neg()
    old = None
    for i in range(len(sigmas) - 1):
        den = precond(x, sigmas[i].item())
        t, t_next = t_fn(sigmas[i]), t_fn(sigmas[i + 1])
        h = t_next - t
        if old is None or sigmas[i + 1] == 0:
            d = den
        else:
            h_last = t - t_fn(sigmas[i - 1])
            r = h_last / h
            gamma = -1 / (2 * r)
            d = (1 - gamma) * den + gamma * old
        x = (sigma_fn(t_next) / sigma_fn(t)) * x - (-h).expm1() * d
        old = den
    return x


def heun_sample(sd, p, z_cond, noise, step_noise, num_sample_steps=8, clamp=False, groups=4, sigma_min=0.002,
                sigma_max=80, sigma_data=0.5, rho=7, S_churn=80, S_tmin=0.05, S_tmax=50, S_noise=1.003):
    """ElucidatedDiffusion.sample_normal, elucidated_diffusion.py:177-257, with the noise draws handed in
    (noise = the initial unit-normal draw, step_noise[i] = the draw of step i)."""
    from math import sqrt
    N = num_sample_steps
    inv_rho = 1 / rho
    steps = torch.arange(N, dtype=torch.float32)
    sigmas = (sigma_max ** inv_rho + steps / (N - 1) * (sigma_min ** inv_rho - sigma_max ** inv_rho)) ** rho
    sigmas = F.pad(sigmas, (0, 1), value=0.0)
    gammas = torch.where((sigmas >= S_tmin) & (sigmas <= S_tmax), min(S_churn / N, sqrt(2) - 1), 0.0)
    batch = noise.shape[0]

    def precond(xx, sigma):
        sg = torch.full((batch,), sigma)
        ps = sg.reshape(-1, 1, 1)
        c_in = 1 * (ps ** 2 + sigma_data ** 2) ** -0.5
        c_skip = (sigma_data ** 2) / (ps ** 2 + sigma_data ** 2)
        c_out = ps * sigma_data * (sigma_data ** 2 + ps ** 2) ** -0.5
        c_noise = torch.log(sg.clamp(min=1e-20)) * 0.25
        net = resnet1d_forward(sd, p, c_in * xx, z_cond=z_cond, time=c_noise, groups=groups)
        out = c_skip * xx + c_out * net
        return out.clamp(-1.0, 1.0) if clamp else out

    x = sigmas[0] * noise
    for i in range(N):
        sigma, sigma_next, gamma = sigmas[i].item(), sigmas[i + 1].item(), gammas[i].item()
        eps = S_noise * step_noise[i]
        sigma_hat = sigma + gamma * sigma
        x_hat = x + sqrt(sigma_hat ** 2 - sigma ** 2) * eps
        out = precond(x_hat, sigma_hat)
        d_over = (x_hat - out) / sigma_hat
        x_next = x_hat + (sigma_next - sigma_hat) * d_over
        if sigma_next != 0:
            out2 = precond(x_next, sigma_next)
            d_prime = (x_next - out2) / sigma_next
            x_next = x_hat + 0.5 * (sigma_next - sigma_hat) * (d_over + d_prime)
        x = x_next
    return x


def decoder_forward(sd, p, z_h, cond, groups=4):
    """ConditionalGraspPoseDecoder.forward, grasp_vae.py:401-436 (no qualities)."""
    h = F.linear(z_h, sd[p + "in_layer.weight"], sd[p + "in_layer.bias"]).unsqueeze(-2)
    h = resnet1d_forward(sd, p + "net.", h, z_cond=cond, groups=groups).squeeze(-2)
    tmrp = F.linear(h, sd[p + "tmrp.weight"], sd[p + "tmrp.bias"])
    logit = F.linear(h, sd[p + "class_logits.weight"], sd[p + "class_logits.bias"])
    return tmrp, logit


# --------------------------------------------------------------------------
# PVCNN pieces                          grasp_ldm/models/modules/ext/pvcnn/**
# --------------------------------------------------------------------------


def _swish(x):
    return x * torch.sigmoid(x)


def _bn_eval(x, sd, p):
    return F.batch_norm(x, sd[p + "running_mean"], sd[p + "running_var"], sd[p + "weight"], sd[p + "bias"],
                        training=False, eps=1e-5)


def shared_mlp(sd, p, x):
    """SharedMLP (Conv k=1 + BatchNorm(eval) + ReLU)*, shared_mlp.py:6-35.
    p addresses the SharedMLP ("...layers." is appended here)."""
    i = 0
    while (p + f"layers.{i}.weight") in sd:
        w = sd[p + f"layers.{i}.weight"]
        conv = F.conv1d if w.ndim == 3 else F.conv2d
        x = conv(x, w, sd[p + f"layers.{i}.bias"])
        x = F.relu(_bn_eval(x, sd, p + f"layers.{i + 1}."))
        i += 3
    return x


def voxelize(features, coords, r, normalize, eps=0.0):
    """Voxelization.forward, modules/voxelization.py:16-35."""
    nc = coords - coords.mean(2, keepdim=True)
    if normalize:
        nc = nc / (nc.norm(dim=1, keepdim=True).max(dim=2, keepdim=True).values * 2.0 + eps) + 0.5
    else:
        nc = (nc + 1) / 2.0
    nc = torch.clamp(nc * r, 0, r - 1)
    vox = torch.round(nc).to(torch.int32)
    out, _, _ = B.avg_voxelize_forward(features.contiguous(), vox.contiguous(), r)
    return out.view(features.shape[0], features.shape[1], r, r, r), nc


def pvconv(sd, p, features, coords, r, normalize, se_relu):
    """PVConv.forward, modules/pvconv.py:76-84 with voxel_layers of :47-74 and
    SE3d (se.py:12-25); eval mode, no attention."""
    vox, nc = voxelize(features, coords, r, normalize)
    v = p + "voxel_layers."
    h = F.conv3d(vox, sd[v + "0.weight"], sd[v + "0.bias"], padding=1)
    h = _swish(F.group_norm(h, 8, sd[v + "1.weight"], sd[v + "1.bias"], eps=1e-5))
    h = F.conv3d(h, sd[v + "4.weight"], sd[v + "4.bias"], padding=1)
    h = _swish(F.group_norm(h, 8, sd[v + "5.weight"], sd[v + "5.bias"], eps=1e-5))
    if (v + "7.fc.0.weight") in sd:
        s = h.mean(-1).mean(-1).mean(-1)
        s = F.linear(s, sd[v + "7.fc.0.weight"])
        s = F.relu(s) if se_relu else _swish(s)
        s = torch.sigmoid(F.linear(s, sd[v + "7.fc.2.weight"]))
        h = h * s.view(h.shape[0], h.shape[1], 1, 1, 1)
    dv = B.trilinear_devoxelize_forward(r, False, nc.contiguous(), h.contiguous().view(h.shape[0], h.shape[1], -1))[0]
    return dv + shared_mlp(sd, p + "point_features.", features)


def pvcnn_block_spec(scale_channels, scale_voxel_resolution, num_blocks=(1, 1, 1, 1), extra_block_channels=None):
    """PVCNN.get_blocks_spec, pvcnn_base.py:81-112."""
    c = [int(64 * scale_channels), int(128 * scale_channels), int(1024 * scale_channels), int(2048 * scale_channels)]
    r = [int(32 * scale_voxel_resolution), int(16 * scale_voxel_resolution), None, None]
    spec = [(c[i], num_blocks[i], r[i]) for i in range(4)]
    if extra_block_channels:
        spec += [(ch, 1, None) for ch in extra_block_channels]
    return spec


def pvcnn_forward(sd, p, inputs, spec):
    """PVCNN.forward, pvcnn_base.py:114-140 (is_conditioned=False);
    layers from create_pointnet_components, utils.py:65-94 (with_se, Swish SE,
    normalize=False)."""
    features, coords = inputs, inputs[:, :3, :]
    i = 0
    for _, nb, res in spec:
        for _ in range(nb):
            q = p + f"point_features.{i}."
            if res is None:
                features = shared_mlp(sd, q, features)
            else:
                features = pvconv(sd, q, features, coords, res, normalize=False, se_relu=False)
            i += 1
    return features


def pvcnn_encoder_forward(sd, p, pc, spec):
    """PVCNNEncoder.forward, pc_encoders.py:87-115 (use_global_attention=False)."""
    x = pc.transpose(1, 2).contiguous()
    x = pvcnn_forward(sd, p + "pvcnn_modules.", x, spec)
    x = F.conv1d(x, sd[p + "conv_downscale.weight"], sd[p + "conv_downscale.bias"])
    x = F.conv1d(x, sd[p + "out_layer.0.weight"], sd[p + "out_layer.0.bias"])
    x = F.linear(x, sd[p + "out_layer.1.weight"], sd[p + "out_layer.1.bias"])
    return x.squeeze(1) if x.shape[-2] == 1 else x


# --------------------------------------------------------------------------
# PointNet++ set abstraction / feature propagation     modules/pointnet.py
# --------------------------------------------------------------------------


def ball_group(points, centers, feats, radius, k, include_coordinates=True):
    """BallQuery.forward, modules/ball_query.py:16-34."""
    idx = B.ball_query(centers.contiguous(), points.contiguous(), radius, k)
    nc = B.grouping_forward(points.contiguous(), idx) - centers.unsqueeze(-1)
    if feats is None:
        return nc
    nf = B.grouping_forward(feats.contiguous(), idx)
    return torch.cat([nc, nf], dim=1) if include_coordinates else nf


def furthest_point_sample(coords, m):
    """functional/sampling.py:39-50."""
    coords = coords.contiguous()
    return B.gather_features_forward(coords, B.furthest_point_sampling(coords, m))


def sa_module(sd, p, feats, coords, num_centers, radii, ks):
    """PointNetSAModule.forward, modules/pointnet.py:100-111.  feats may be None
    (features with 0 channels are passed as a [B,0,N] tensor by PointNet2)."""
    centers = furthest_point_sample(coords, num_centers)
    if feats is not None and feats.shape[1] == 0:
        feats = None
    outs = []
    for g, (rad, k) in enumerate(zip(radii, ks)):
        grouped = ball_group(coords, centers, feats, rad, k)
        outs.append(shared_mlp(sd, p + f"mlps.{g}.", grouped).max(dim=-1).values)
    return (torch.cat(outs, dim=1) if len(outs) > 1 else outs[0]), centers


def a_module(sd, p, feats, coords):
    """PointNetAModule.forward, modules/pointnet.py:34-46 (single MLP)."""
    x = torch.cat([feats, coords], dim=1)
    out = shared_mlp(sd, p + "mlps.0.", x).max(dim=-1, keepdim=True).values
    return out, torch.zeros((coords.size(0), 3, 1))


def fp_module(sd, p, points_coords, centers_coords, centers_feats, points_feats):
    """PointNetFPModule.forward, modules/pointnet.py:122-135."""
    interp = B.three_nearest_neighbors_interpolate_forward(
        points_coords.contiguous(), centers_coords.contiguous(), centers_feats.contiguous())[0]
    if points_feats is not None:
        interp = torch.cat([interp, points_feats], dim=1)
    return shared_mlp(sd, p + "mlp.", interp), points_coords


SSG_SA = [(512, 0.2, 64), (128, 0.4, 64), None]  # pointnet2.py:99-103


def pointnet2_ssg_forward(sd, p, inputs):
    """PointNet2.forward with the PointNet2SSG tables, pointnet2.py:64-123
    (extra_feature_channels=0: inputs [B,3,N])."""
    coords, feats = inputs[:, :3, :].contiguous(), inputs[:, 3:, :].contiguous()
    coords_list, feats_list = [], []
    for i, cfg in enumerate(SSG_SA):
        feats_list.append(feats)
        coords_list.append(coords)
        q = p + f"sa_layers.{i}."
        if cfg is None:
            feats, coords = a_module(sd, q, feats, coords)
        else:
            feats, coords = sa_module(sd, q, feats, coords, cfg[0], [cfg[1]], [cfg[2]])
    feats_list[0] = inputs.contiguous()
    for j in range(3):
        feats, coords = fp_module(sd, p + f"fp_layers.{j}.", coords_list[-1 - j], coords, feats, feats_list[-1 - j])
    return feats


# --------------------------------------------------------------------------
# Sampler, VAE / LDM generation, pose epilogue
# --------------------------------------------------------------------------


def make_scheduler(kind, num_steps=1000, beta_start=5e-5, beta_end=1e-3, variance_type="fixed_large"):
    """GaussianDiffusion1D.configure_noise_scheduler, gaussian_diffusion.py:124-164."""
    kw = dict(num_train_timesteps=num_steps, beta_start=beta_start, beta_end=beta_end,
              beta_schedule="linear", prediction_type="epsilon", clip_sample=True)
    if kind == "ddim":
        return schedulers.DDIMScheduler(**kw)
    return schedulers.DDPMScheduler(variance_type=variance_type, **kw)


def sample_latents(sd, p, z_cond, sched, n_dims, num_train_steps=1000, x_T=None, step_noise=None,
                   return_all=False, groups=4):
    """GaussianDiffusion1D.sample, gaussian_diffusion.py:232-277.  p addresses the
    denoiser ("diffusion_model.model.").  x_T / step_noise default to the global
    CPU RNG in the reference's draw order (x_T first, then one draw per step t>0)."""
    bsz = z_cond.shape[0]
    x = torch.randn((bsz, 1, n_dims)) if x_T is None else x_T.clone()
    n_inf = sched.num_inference_steps if sched.num_inference_steps is not None else num_train_steps
    trace = [x] if return_all else []
    for i, t in enumerate(reversed(range(0, num_train_steps, num_train_steps // n_inf))):
        tb = torch.full((bsz,), t, dtype=torch.long)
        eps = resnet1d_forward(sd, p, x, z_cond=z_cond, time=tb, groups=groups)
        if step_noise is not None and isinstance(sched, schedulers.DDPMScheduler):
            x = sched.step(eps, t, x, noise=step_noise[i] if t > 0 else None).prev_sample
        else:
            x = sched.step(eps, t, x).prev_sample
        if return_all:
            trace.append(x)
    return x, trace


def ldm_generate(sd, pc, num_grasps, sched, spec, n_dims=4, x_T=None, step_noise=None, groups=4):
    """GraspLatentDDM.generate_grasps, grasp_ldm.py:189-233 -> (tmrp, logit)."""
    z = pvcnn_encoder_forward(sd, "vae_model.encoder.pc_encoder.", pc, spec)
    z = z.repeat_interleave(num_grasps, dim=0)
    x, _ = sample_latents(sd, "diffusion_model.model.", z, sched, n_dims, x_T=x_T, step_noise=step_noise, groups=groups)
    return decoder_forward(sd, "vae_model.decoder.", x.squeeze(-2), z, groups=groups)


def vae_generate(sd, pc, num_grasps, spec, latent=4, z_h=None, prefix="", groups=4):
    """GraspCVAE.generate_grasps, grasp_vae.py:226-255."""
    z = pvcnn_encoder_forward(sd, prefix + "encoder.pc_encoder.", pc, spec).repeat_interleave(num_grasps, dim=0)
    if z_h is None:
        z_h = torch.randn(pc.shape[0] * num_grasps, latent)
    return decoder_forward(sd, prefix + "decoder.", z_h, z, groups=groups)


def tmrp_to_H(tmrp):
    """tmrp_to_H, utils/rotations.py:298-302 via mrp_to_quat :218-252,
    quat_to_rotmat :171-215 (SciPy convention), Rt_to_H :255-274."""
    t, m = tmrp[..., :3], tmrp[..., 3:6]
    magsq = (m * m).sum(-1, keepdim=True)
    q = (2 * m) / (1 + magsq)
    w = ((1 - magsq) / (1 + magsq))[..., 0]
    x, y, z = q[..., 0], q[..., 1], q[..., 2]
    H = torch.zeros(tmrp.shape[:-1] + (4, 4), dtype=tmrp.dtype)
    x2, y2, z2, w2 = x * x, y * y, z * z, w * w
    xy, zw, xz, yw, yz, xw = x * y, z * w, x * z, y * w, y * z, x * w
    H[..., 0, 0] = x2 - y2 - z2 + w2
    H[..., 1, 0] = 2 * (xy + zw)
    H[..., 2, 0] = 2 * (xz - yw)
    H[..., 0, 1] = 2 * (xy - zw)
    H[..., 1, 1] = -x2 + y2 - z2 + w2
    H[..., 2, 1] = 2 * (yz + xw)
    H[..., 0, 2] = 2 * (xz + yw)
    H[..., 1, 2] = 2 * (yz - xw)
    H[..., 2, 2] = -x2 - y2 + z2 + w2
    H[..., :3, 3] = t
    H[..., 3, 3] = 1
    return H


def pose_epilogue(tmrp, logit, metas, num_clouds, num_grasps):
    """InferenceLDM.generate_grasps tail, tools/inference.py:628-656:
    unnormalise (:64-94), tmrp->H, sigmoid confidence."""
    tm = tmrp.view(num_clouds, num_grasps, 6)
    un = tm * metas["grasp_std"].unsqueeze(-2) + metas["grasp_mean"].unsqueeze(-2)
    conf = torch.sigmoid(logit.view(num_clouds, num_grasps, 1))
    return dict(grasps=tmrp_to_H(un), grasp_tmrp=un, confidence=conf)
