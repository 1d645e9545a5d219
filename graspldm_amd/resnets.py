"""1-D ResNet family of GraspLDM (denoiser, pose-decoder core, grasp-encoder core):
host-side mirror of `grasp_ldm/models/modules/resnets.py` -- same class names,
constructor arguments and state_dict keys -- whose forward is the fused HIP engine
(csrc/resnet1d.hip).  The nn.Modules below only HOLD parameters in the reference's
layout; weights are packed for the engine lazily (and re-packed when they change).
Inference only; fp32 only (the reference's norm eps is dtype dependent: resnets.py:86,110).
"""
import math
from functools import partial
from typing import Sequence

import torch
from torch import nn

from .r1d import R1dEngine, pack_resnet1d
from .r1d_pack import SCHED_NONE


def default(val, d):
    return val if val is not None else (d() if callable(d) else d)


class RandomOrLearnedSinusoidalPosEmb(nn.Module):
    """resnets.py:44-56 (parameter holder; evaluated into the [T, E] table at pack time)."""

    def __init__(self, dim, is_random=False):
        super().__init__()
        assert (dim % 2) == 0
        self.weights = nn.Parameter(torch.randn(dim // 2), requires_grad=not is_random)


class WeightStandardizedConv2d(nn.Conv1d):
    """resnets.py:79-101 (standardisation happens once at pack time)."""


class LayerNorm(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.g = nn.Parameter(torch.ones(1, dim, 1))


class PreNorm(nn.Module):
    def __init__(self, dim, fn):
        super().__init__()
        self.fn = fn
        self.norm = LayerNorm(dim)


class Residual(nn.Module):
    def __init__(self, fn):
        super().__init__()
        self.fn = fn


class Block(nn.Module):
    def __init__(self, dim, dim_out, groups=8):
        super().__init__()
        self.proj = WeightStandardizedConv2d(dim, dim_out, 3, padding=1)
        self.norm = nn.GroupNorm(groups, dim_out)
        self.act = nn.SiLU()


class ResnetBlock(nn.Module):
    def __init__(self, dim, dim_out, *, emb_dim=None, groups=8):
        super().__init__()
        self.mlp = nn.Sequential(nn.SiLU(), nn.Linear(emb_dim, dim_out * 2)) if emb_dim is not None else None
        self.block1 = Block(dim, dim_out, groups=groups)
        self.block2 = Block(dim_out, dim_out, groups=groups)
        self.res_conv = nn.Conv1d(dim, dim_out, 1) if dim != dim_out else nn.Identity()


class LinearAttention(nn.Module):
    def __init__(self, dim, heads=4, dim_head=32):
        super().__init__()
        self.scale = dim_head ** -0.5
        self.heads = heads
        hidden = dim_head * heads
        self.to_qkv = nn.Conv1d(dim, hidden * 3, 1, bias=False)
        self.to_out = nn.Sequential(nn.Conv1d(hidden, dim, 1), LayerNorm(dim))


class _ResNet1DBase(nn.Module):
    """Shared construction + engine plumbing of ResNet1D / TimeConditionedResNet1D."""

    def _build(self, dim, init_dim, out_channels, block_channels, channels, input_conditioning_dims,
               is_self_conditioned, resnet_block_groups, learned_variance, dropout):
        if is_self_conditioned:
            raise NotImplementedError("self-conditioning is not on the generation hot path")
        self.channels = channels
        self.is_self_conditioned = False
        init_dim = default(init_dim, dim)
        self.init_conv = nn.Conv1d(channels, init_dim, 7, padding=3)
        dims = (dim,) + tuple(block_channels)
        self.in_features = self.out_features = dim
        self.groups = resnet_block_groups
        block_klass = partial(ResnetBlock, groups=resnet_block_groups)
        self.dropout = nn.Dropout(p=dropout, inplace=True) if dropout is not None else None
        self.emb_dim = emb_dim = dim * 4
        return dims, emb_dim, block_klass

    def _finish(self, dims, emb_dim, block_klass, input_conditioning_dims, channels, out_channels, learned_variance):
        if input_conditioning_dims is None:
            raise NotImplementedError("the HIP engine expects an input-conditioned network (z_cond)")
        self.is_input_conditioned = True
        self.input_emb_layers = nn.Sequential(nn.Linear(input_conditioning_dims, emb_dim), nn.SiLU())
        self.blocks = nn.ModuleList([])
        for dim_in, dim_out in zip(dims[:-1], dims[1:]):
            self.blocks.append(nn.ModuleList([
                block_klass(dim_in, dim_in, emb_dim=emb_dim),
                block_klass(dim_in, dim_in, emb_dim=emb_dim),
                Residual(PreNorm(dim_in, LinearAttention(dim_in))),
                nn.Conv1d(dim_in, dim_out, 3, padding=1),
            ]))
        self.out_channels = default(out_channels, channels * (1 if not learned_variance else 2))
        if self.out_channels != 1:
            raise NotImplementedError("learned variance (2 output channels) is not on the generation hot path")
        self.final_res_block = block_klass(dims[-1], dims[-1], emb_dim=emb_dim)
        self.final_conv = nn.Conv1d(dims[-1], self.out_channels, 1)
        self._engine = None
        self._engine_key = None
        self.max_timesteps = 1000

    # ---- engine plumbing
    def _param_key(self, device):
        from ._cache import params_key
        return params_key(self.parameters(), device, self.max_timesteps)

    def engine(self, device, decoder=None, prefix=""):
        key = self._param_key(device)
        if self._engine is None or self._engine_key != key:
            sd = {k: v.detach().float().cpu() for k, v in self.state_dict().items()}
            has_time = getattr(self, "time_mlp", None) is not None
            packed = pack_resnet1d(sd, "", groups=self.groups, seq_len=self.in_features,
                                   num_steps=self.max_timesteps if has_time else None, decoder=decoder,
                                   cond_rows=getattr(self, "cond_rows", 3))
            self._engine = R1dEngine(packed, device)
            self._engine_key = key
            from ._cache import publish
            publish(device)
        return self._engine

    def _cond_rows_of(self, z_cond):
        rows = 1 if z_cond.ndim == 2 else z_cond.shape[1]
        if getattr(self, "cond_rows", None) != rows:
            self.cond_rows = rows
            self._engine = None
        return rows


class ResNet1D(_ResNet1DBase):
    """resnets.py:263-424"""

    def __init__(self, dim: int, init_dim: int = None, out_channels: int = None,
                 block_channels: Sequence = (16, 64, 128, 64, 16), channels: int = 1,
                 input_conditioning_dims: int = None, is_self_conditioned: bool = False,
                 resnet_block_groups: int = 8, learned_variance: bool = False, dropout=None) -> None:
        super().__init__()
        dims, emb_dim, bk = self._build(dim, init_dim, out_channels, block_channels, channels,
                                        input_conditioning_dims, is_self_conditioned, resnet_block_groups,
                                        learned_variance, dropout)
        self._finish(dims, emb_dim, bk, input_conditioning_dims, channels, out_channels, learned_variance)

    @torch.no_grad()
    def forward(self, x, *, z_cond=None, x_self_cond=None):
        if not x.is_cuda:
            raise RuntimeError("x must be a CUDA tensor (graspldm_amd has no CPU path)")
        self._cond_rows_of(z_cond)
        eng = self.engine(x.device)
        return eng.denoise(x, eng.cond_embed(z_cond), 1, sched_kind=SCHED_NONE)


class TimeConditionedResNet1D(_ResNet1DBase):
    """resnets.py:427-616"""

    def __init__(self, dim: int, init_dim: int = None, out_channels: int = None,
                 block_channels: Sequence = (16, 64, 128, 64, 16), channels: int = 1,
                 input_conditioning_dims: int = None, is_self_conditioned: bool = False,
                 resnet_block_groups: int = 8, learned_variance: bool = False, dropout=None,
                 is_time_conditioned: bool = True, learned_sinusoidal_cond: bool = False,
                 random_fourier_features: bool = False, learned_sinusoidal_dim: int = 16) -> None:
        super().__init__()
        dims, emb_dim, bk = self._build(dim, init_dim, out_channels, block_channels, channels,
                                        input_conditioning_dims, is_self_conditioned, resnet_block_groups,
                                        learned_variance, dropout)
        self.random_or_learned_sinusoidal_cond = learned_sinusoidal_cond or random_fourier_features
        if not is_time_conditioned:
            raise NotImplementedError("use ResNet1D for a network without time conditioning")
        if not self.random_or_learned_sinusoidal_cond:
            raise NotImplementedError("plain SinusoidalPosEmb is not used by the shipped configs")
        self.is_time_conditioned = True
        self.time_mlp = nn.Sequential(
            RandomOrLearnedSinusoidalPosEmb(learned_sinusoidal_dim, random_fourier_features),
            nn.Linear(learned_sinusoidal_dim + 1, emb_dim), nn.GELU(), nn.Linear(emb_dim, emb_dim))
        self._finish(dims, emb_dim, bk, input_conditioning_dims, channels, out_channels, learned_variance)

    @torch.no_grad()
    def forward(self, x, *, time=None, z_cond=None, x_self_cond=None, **kwargs):
        """eps = model(x [B,1,D], time int64 [B], z_cond [B,R,Dc]); unknown kwargs (e.g. `metas`)
        are accepted and ignored like the reference does (resnets.py:565)."""
        assert time is not None
        if not x.is_cuda:
            raise RuntimeError("x must be a CUDA tensor (graspldm_amd has no CPU path)")
        tmax = int(time.max())
        if tmax >= self.max_timesteps:
            self.max_timesteps = tmax + 1
        self._cond_rows_of(z_cond)
        eng = self.engine(x.device)
        return eng.denoise(x, eng.cond_embed(z_cond), 1, sample_t=time.to(torch.int32).contiguous(),
                           sched_kind=SCHED_NONE)


class ClassTimeConditionedResNet1D(TimeConditionedResNet1D):
    """grasp_ldm/models/modules/class_conditioned_resnet.py:9-122: the time-conditioned denoiser with a class
    embedding (Linear(1, emb) + SiLU of the class label) added to the time embedding of every sample before the
    conditioning embedding.  In the fused engine that is one extra [n, emb] operand of gldm_denoise."""

    def __init__(self, *args, **kwargs) -> None:
        super().__init__(*args, **kwargs)
        self.cls_embed = nn.Sequential(nn.Linear(1, self.emb_dim), nn.SiLU())

    def class_embedding(self, cls_cond=None, n=None, **kwargs):
        """cls_embed(cls_cond) as [n, emb] (class_conditioned_resnet.py:72-82,99-100).  The label comes from
        `cls_cond` or, like in the reference, from kwargs["metas"]["mode_cls"]."""
        if cls_cond is None:
            metas = kwargs.get("metas")
            assert metas is not None and "mode_cls" in metas, "Class conditioning tensor is required"
            cls_cond = metas["mode_cls"]
        lin = self.cls_embed[0]
        cls_cond = cls_cond.to(device=lin.weight.device, dtype=torch.float32).unsqueeze(-1).reshape(-1, 1)
        if not cls_cond.is_cuda:
            raise RuntimeError("the model must be on the GPU (graspldm_amd has no CPU path)")
        if n is not None and cls_cond.shape[0] != n:
            raise RuntimeError(f"class labels for {cls_cond.shape[0]} samples, batch of {n}")
        from . import _lib as L
        m = cls_cond.shape[0]
        out = torch.empty((m, self.emb_dim), dtype=torch.float32, device=cls_cond.device)
        w, b = lin.weight.detach().contiguous().float(), lin.bias.detach().contiguous().float()
        with torch.cuda.device(out.device):  # Linear(1, E) + SiLU = the conditioning-embedding kernel with Dc = 1
            L.call("gldm_r1d_cond_embed", L.ptr(cls_cond.contiguous()), L.ptr(w), L.ptr(b), m, 1, 1, self.emb_dim,
                   L.ptr(out), L.current_stream(out.device))
        return out

    @torch.no_grad()
    def forward(self, x, *, time=None, z_cond=None, x_self_cond=None, cls_cond=None, **kwargs):
        assert time is not None
        if not x.is_cuda:
            raise RuntimeError("x must be a CUDA tensor (graspldm_amd has no CPU path)")
        tmax = int(time.max())
        if tmax >= self.max_timesteps:
            self.max_timesteps = tmax + 1
        self._cond_rows_of(z_cond)
        eng = self.engine(x.device)
        semb = self.class_embedding(cls_cond, n=x.shape[0], **kwargs)
        return eng.denoise(x, eng.cond_embed(z_cond), 1, sample_t=time.to(torch.int32).contiguous(),
                           sched_kind=SCHED_NONE, sample_emb=semb)


__all__ = ["ResNet1D", "TimeConditionedResNet1D", "ClassTimeConditionedResNet1D", "ResnetBlock", "LinearAttention", "LayerNorm",
           "WeightStandardizedConv2d", "RandomOrLearnedSinusoidalPosEmb"]
_ = math
