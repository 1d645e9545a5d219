"""Point-cloud encoder heads: mirror of `grasp_ldm/models/modules/pc_encoders.py`
(PVCNNEncoder :8-136, PVCNN2Encoder :139-197) with the same constructor arguments and
state_dict keys."""
import torch
from torch import nn

from . import dense
from .pvcnn import PVCNN, PVCNN2


class PVCNNEncoder(nn.Module):
    def __init__(self, in_features=3, out_features=32, n_points=1024, extra_feature_channels=0, scale_channels=0.25,
                 scale_voxel_resolution=0.75, num_blocks=(1, 1, 1, 1), is_conditioned=False, cond_dims=None,
                 extra_block_channels=None, use_global_attention=False, out_channels=1, load_from_ckpt_path=None):
        super().__init__()
        if use_global_attention:
            raise NotImplementedError("global attention is off in the shipped configs and not on the hot path")
        self.pvcnn_modules = PVCNN(extra_feature_channels=extra_feature_channels, scale_channels=scale_channels,
                                   scale_voxel_resolution=scale_voxel_resolution, num_blocks=num_blocks,
                                   is_conditioned=is_conditioned, cond_dims=cond_dims,
                                   extra_block_channels=extra_block_channels)
        self._finish(in_features, out_features, n_points, out_channels)
        if load_from_ckpt_path is not None:
            self.load_ckpt_and_freeze(load_from_ckpt_path)

    def _finish(self, in_features, out_features, n_points, out_channels):
        self.in_features, self.out_features = in_features, out_features
        mid = int(self.pvcnn_modules.out_channels / 2)
        self.conv_downscale = nn.Conv1d(self.pvcnn_modules.out_channels, mid, kernel_size=1)
        self.global_attention = None
        self.out_layer = nn.Sequential(nn.Conv1d(mid, out_channels, kernel_size=1),
                                       nn.Linear(n_points, self.out_features))

    @torch.no_grad()
    def forward(self, out, cond=None):
        """pc [B, N, 3] -> latent [B, C_out, out_features] (pc_encoders.py:87-115)."""
        if not out.is_cuda:
            raise RuntimeError("pointcloud must be a CUDA tensor (graspldm_amd has no CPU path)")
        x = torch.transpose(out, 1, 2).contiguous()
        w, b = self._folded_head()
        x = self._backbone_and_head(x, cond, w, b)
        x = dense.linear(x, self.out_layer[1])
        return x.squeeze(1) if x.shape[-2] == 1 else x

    def _backbone_and_head(self, x, cond, w, b):
        """backbone -> head.  When the backbone ends in wide SharedMLP layers (PVCNN: 96 -> 768 -> 1536), they and the
        folded head run as ONE launch: the 768-row tile is produced in LDS, the head product is taken on the last
        layer's accumulators, and neither [B, 768, N] (0.8 GB per 256 clouds) nor [B, 1536, N] (1.6 GB) is written."""
        from .pvcnn import PVCNN, SharedMLP
        bb = self.pvcnn_modules
        layers = list(bb.point_features) if isinstance(bb, PVCNN) else []
        last = layers[-1] if layers else None
        if isinstance(last, SharedMLP) and len(last.layers) == 3:
            conv, bn = last.layers[0], last.layers[1]
            cin, cout = conv.weight.shape[1], conv.weight.shape[0]
            feats = x[:, : bb.in_channels, :]
            coords = feats[:, :3, :].contiguous()
            prev = layers[-2] if len(layers) > 1 else None
            two = isinstance(prev, SharedMLP) and len(prev.layers) == 3  # 96 -> 768 -> 1536 -> head in one launch
            for layer in layers[: -2 if two else -1]:
                feats, _ = layer((feats, coords))
            feats = feats.contiguous().float()
            if two:
                conv0, bn0 = prev.layers[0], prev.layers[1]
                if w.shape[0] <= 16 and dense.fused_mlp2_supported(feats, conv0.weight.shape[1], cin, cout):
                    _, b0, wp0, ws0 = dense.folded_conv_bn(conv0, bn0, feats.device)
                    _, bf, wp, ws = dense.folded_conv_bn(conv, bn, feats.device)
                    if ws is not None and ws0 is not None and dense.split_supported(cin, conv0.weight.shape[1]):
                        # both GEMMs on the f16 matrix pipe (split-f32 operands)
                        return dense.pointwise_mlp(feats, ws, bf, cout, True, head=self._packed_head(w, b), keep_y=False,
                                                   front=(ws0, b0, cin, dense.folded_range_gain(conv0)), split=True)[1]
                    return dense.pointwise_mlp(feats, wp, bf, cout, True, head=self._packed_head(w, b), keep_y=False,
                                               front=(wp0, b0, cin))[1]
                feats = prev(feats).contiguous()
            if w.shape[0] <= 16 and dense.fused_mlp_supported(feats, cin, cout):
                _, bf, wp, ws = dense.folded_conv_bn(conv, bn, feats.device)
                if ws is not None:
                    return dense.pointwise_mlp(feats, ws, bf, cout, True, head=self._packed_head(w, b), keep_y=False,
                                               split=True)[1]
                return dense.pointwise_mlp(feats, wp, bf, cout, True, head=self._packed_head(w, b), keep_y=False)[1]
            feats = last(feats)
            return dense.pointwise_gemm(feats, w, b)
        return dense.pointwise_gemm(bb(x, cond=cond), w, b)

    def _packed_head(self, w, b):
        from ._cache import publish
        hit = self.__dict__.get("_head_packed")
        if hit is None or hit[0] is not w:
            hit = (w, dense.pack_head(w).to(w.device), b.contiguous(), int(w.shape[0]))
            self.__dict__["_head_packed"] = hit
            publish(w.device)
        return hit[1], hit[2], hit[3]

    def _folded_head(self):
        """conv_downscale (C -> C/2, k=1) and out_layer[0] (C/2 -> out_channels, k=1) have nothing between them
        when global attention is off (pc_encoders.py:104-111): W = W_out W_down, b = W_out b_down + b_out, one
        [out_channels x C] GEMM over the points instead of a 2 C^2/2 FLOP-per-point one.  Folded in f64 once per
        weight version."""
        cd, o0 = self.conv_downscale, self.out_layer[0]
        from ._cache import params_key, publish
        key = params_key([cd.weight, cd.bias, o0.weight, o0.bias], cd.weight.device)
        hit = self.__dict__.get("_head_cache")
        if hit is None or hit[0] != key:
            # folded on the HOST (f64 there; on the device this product was the path's last library GEMM, once per weight version)
            dev = cd.weight.device
            wo, wd = o0.weight[:, :, 0].detach().double().cpu(), cd.weight[:, :, 0].detach().double().cpu()
            w = (wo @ wd).float().contiguous().to(dev)
            b = (wo @ cd.bias.detach().double().cpu() + o0.bias.detach().double().cpu()).float().contiguous().to(dev)
            hit = (key, w, b)
            self.__dict__["_head_cache"] = hit
            publish(cd.weight.device)
        return hit[1], hit[2]

    def load_ckpt_and_freeze(self, ckpt_path, fine_tune=False):
        ckpt = torch.load(ckpt_path, map_location="cpu")
        self.load_state_dict(ckpt["state_dict"] if "state_dict" in ckpt else ckpt)
        for p in self.parameters():
            p.requires_grad = False


class PVCNN2Encoder(PVCNNEncoder):
    """The reference's PVCNN2Encoder cannot be constructed (it forwards scale_channels /
    num_blocks / ... to PVCNN2.__init__, which does not accept them: pc_encoders.py:188-197 vs
    pvcnn_base.py:204-212).  This is the repaired form: same head, PVCNN2 (set-abstraction +
    feature-propagation) backbone.  The two scale arguments mean what they mean for PVCNNEncoder and are mapped onto
    the arguments PVCNN2 does take (scale_channels -> width_multiplier, scale_voxel_resolution ->
    voxel_resolution_multiplier); the ones PVCNN2 has no counterpart for are rejected instead of silently dropped
    (its block counts live in the class tables sa_blocks / fp_blocks, and it has no conditioning inputs)."""

    def __init__(self, in_features=3, out_features=32, n_points=1024, extra_feature_channels=0, scale_channels=0.25,
                 scale_voxel_resolution=0.75, num_blocks=(1, 1, 1, 1), is_conditioned=False, cond_dims=None,
                 extra_block_channels=None, use_global_attention=False, use_local_attention=False, out_channels=1):
        nn.Module.__init__(self)
        if use_global_attention or use_local_attention:
            raise NotImplementedError("attention variants are not on the hot path")
        if tuple(num_blocks) != (1, 1, 1, 1):
            raise NotImplementedError("PVCNN2 fixes its block counts in sa_blocks / fp_blocks (pvcnn_base.py:186-202); "
                                      f"num_blocks={tuple(num_blocks)} cannot be honoured")
        if is_conditioned or cond_dims is not None or extra_block_channels is not None:
            raise NotImplementedError("PVCNN2 takes no conditioning / extra block channels (pvcnn_base.py:204-212)")
        self.pvcnn_modules = PVCNN2(extra_feature_channels=extra_feature_channels, width_multiplier=scale_channels,
                                    voxel_resolution_multiplier=scale_voxel_resolution, use_attention=False)
        self._finish(in_features, out_features, n_points, out_channels)
