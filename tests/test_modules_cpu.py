"""CPU: host logic of the reference-interface mirror -- registry/config builder,
state_dict schema compatibility with the reference (fixtures captured from it),
schedule tables, weight packing layout, and loud failure without a GPU."""
import json
import os

import pytest
import torch

from conftest import GOLDEN, load_schema

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def fpc_config(n_points=1024, scheduler="ddpm"):
    """The shipped fpc experiment (configs/generation/fpc/fpc_1a_latentc3_z4_pc64_180k.py:25-153) as data."""
    rn = dict(block_channels=(32, 64, 128, 256), input_conditioning_dims=64, resnet_block_groups=4, dropout=0.1)
    vae = dict(model=dict(type="GraspCVAE", args=dict(
        grasp_latent_size=4, pc_latent_size=64,
        pc_encoder_config=dict(type="PVCNNEncoder", args=dict(in_features=3, n_points=n_points, scale_channels=0.75,
                                                              scale_voxel_resolution=0.75, num_blocks=(1, 1, 1, 1),
                                                              out_channels=3, use_global_attention=False)),
        grasp_encoder_config=dict(type="ResNet1D", args=dict(in_features=7, **rn)),
        decoder_config=dict(type="ResNet1D", args=dict(**rn)),
        loss_config=dict(reconstruction_loss=dict(type="x"), latent_loss=dict(type="y")),
        num_output_qualities=0, intermediate_feature_resolution=16)))
    ddm = dict(model=dict(type="GraspLatentDDM", args=dict(
        model=dict(type="TimeConditionedResNet1D", args=dict(dim=4, channels=1, is_time_conditioned=True,
                                                             learned_variance=False, learned_sinusoidal_cond=False,
                                                             random_fourier_features=True, **rn)),
        latent_in_features=4, diffusion_timesteps=1000, noise_scheduler_type=scheduler, diffusion_loss="l2",
        beta_schedule="linear", is_conditioned=True, joint_training=False, denoising_loss_weight=1,
        variance_type="fixed_large", elucidated_diffusion=False, beta_start=0.00005, beta_end=0.001)))
    return dict(vae=vae, ddm=ddm)


def build_fpc(n_points=1024, scheduler="ddpm"):
    from graspldm_amd.builder import build_model_from_cfg
    cfg = fpc_config(n_points, scheduler)
    ldm = build_model_from_cfg(cfg["ddm"])
    ldm.set_vae_model(build_model_from_cfg(cfg["vae"]))
    return ldm.eval()


@pytest.mark.parametrize("n_points,schema", [(1024, "schema_fpc_ldm.json"), (64, "schema_fpc_ldm_n64.json")])
def test_state_dict_schema_matches_reference(n_points, schema):
    ref = load_schema(schema)
    sd = build_fpc(n_points).state_dict()
    assert set(sd) == set(ref), (sorted(set(ref) - set(sd))[:5], sorted(set(sd) - set(ref))[:5])
    for k, (shape, dtype) in ref.items():
        assert tuple(sd[k].shape) == shape and sd[k].dtype == dtype, k


def test_ppc_state_dict_schema_matches_reference():
    """The reference's partial-cloud experiment (partial_pc/ppc_1a_..._z16_pc256: 16-dim grasp latent, 3 x 256 cloud
    latent): same keys, shapes and dtypes as the reference's module tree."""
    from graspldm_amd.builder import build_model_from_cfg
    from graspldm_amd.pipeline import fpc_model_config
    cfg = fpc_model_config(scheduler="ddpm", latent=16, pc_latent=256)
    ldm = build_model_from_cfg(cfg["ddm"])
    ldm.set_vae_model(build_model_from_cfg(cfg["vae"]))
    ref, sd = load_schema("schema_ppc_ldm.json"), ldm.state_dict()
    assert set(sd) == set(ref)
    for k, (shape, dtype) in ref.items():
        assert tuple(sd[k].shape) == shape and sd[k].dtype == dtype, k


def test_strict_load_of_synthetic_reference_weights(fpc_state_dict):
    ldm = build_fpc()
    missing, unexpected = ldm.load_state_dict(fpc_state_dict, strict=True)
    assert not missing and not unexpected


def test_pointnet_schemas_match_reference():
    from graspldm_amd.pvcnn import PVCNN2, PointNet2SSG, PointNetSAModule
    sa1 = PointNetSAModule(num_centers=512, radius=0.2, num_neighbors=64, in_channels=0, out_channels=(64, 64, 128))
    ssg = PointNet2SSG(extra_feature_channels=0)
    for mod, schema in ((sa1, "schema_sa1.json"), (ssg, "schema_pointnet2_ssg.json")):
        ref = load_schema(schema)
        sd = mod.state_dict()
        assert set(sd) == set(ref)
        assert all(tuple(sd[k].shape) == ref[k][0] for k in ref)
    assert sum(p.numel() for p in PVCNN2().parameters()) == 11370080


def test_shipped_config_file_builds_when_reference_is_present():
    path = "/root/reference/configs/generation/fpc/fpc_1a_latentc3_z4_pc64_180k.py"
    if not os.path.exists(path):
        pytest.skip("reference tree not present (GPU box)")
    from graspldm_amd.builder import build_model_from_cfg
    from graspldm_amd.config import Config
    cfg = Config.fromfile(path)
    ldm = build_model_from_cfg(cfg.model.ddm)
    ldm.set_vae_model(build_model_from_cfg(cfg.model.vae))
    ldm2 = build_model_from_cfg(cfg.model.ddm)  # configs are reusable here (single-use in the reference)
    assert sum(p.numel() for p in ldm.parameters()) == 5913693
    assert type(ldm2).__name__ == "GraspLatentDDM"
    assert ldm.diffusion_model.beta_start == 5e-5 and ldm.diffusion_model.beta_end == 1e-3


def test_inference_timesteps_follow_reference_loop():
    from graspldm_amd.diffusion import inference_timesteps
    assert inference_timesteps(1000, 100)[:3] == [990, 980, 970] and inference_timesteps(1000, 100)[-1] == 0
    assert inference_timesteps(1000, None)[0] == 999 and len(inference_timesteps(1000, None)) == 1000


def test_mfma_fragment_packing_layout():
    from graspldm_amd.r1d_pack import conv_as_gemm, mfma_a_fragments
    w = torch.arange(20 * 3 * 3, dtype=torch.float32).view(20, 3, 3)  # Cout 20, Cin 3, taps 3
    g = conv_as_gemm(w)
    assert g[5, 1 * 3 + 2] == w[5, 2, 1]  # k = tap * Cin + ci
    frag = mfma_a_fragments(g).view(2, 1, 64, 4)  # 2 m-tiles, 1 k-block
    for lane in (0, 17, 63):
        for j in range(4):
            m, k = (lane & 15), 4 * j + (lane >> 4)
            assert frag[0, 0, lane, j] == (g[m, k] if k < 9 else 0)
            assert frag[1, 0, lane, j] == (g[16 + m, k] if (16 + m < 20 and k < 9) else 0)


def test_packed_descriptor_of_denoiser(fpc_state_dict):
    from graspldm_amd.r1d_pack import pack_resnet1d
    p = pack_resnet1d(fpc_state_dict, "diffusion_model.model.", groups=4, seq_len=4, num_steps=1000)
    d = p["desc"]
    assert list(d.dims)[:5] == [4, 32, 64, 128, 256] and d.n_levels == 4 and d.emb_dim == 16
    assert d.ss_rows == 2 * 256
    assert p["temb"].shape == (1000, 16) and p["weights"].numel() % 4 == 0
    import ctypes
    from graspldm_amd import _lib
    ptr = ctypes.cast(ctypes.pointer(d), ctypes.c_void_p)
    # 256-byte header + one 8-byte hand-off granule per column of ceil(20 / 16) position-major tiles of 64 columns
    # (no park scratch since round 5: with two f16 planes per value the 256-channel level's f32 rows stay in LDS)
    assert _lib.lib().gldm_r1d_workspace_bytes(ptr, 20) == 256 + 2 * 64 * 8
    d.emb_dim = 32  # not a shape of the position-major engine -> sample-major tiles: ceil(20 / 8) of 32 columns
    assert _lib.lib().gldm_r1d_workspace_bytes(ptr, 20) == 256 + 3 * 32 * 8
    d.emb_dim = 16
    keep = d.lv[1].qkvn_w
    d.lv[1].qkvn_w = 0  # a descriptor without the folded qkv block (ABI 3 packers) -> sample-major tiles as well
    assert _lib.lib().gldm_r1d_workspace_bytes(ptr, 20) == 256 + 3 * 32 * 8
    d.lv[1].qkvn_w = keep
    d.dims[1] = 200  # attention level wider than the LDS plan -> refused, not mis-run
    assert _lib.lib().gldm_r1d_workspace_bytes(ptr, 20) == -1


def test_decoder_workspace_holds_the_scale_shift_table():
    """gldm_r1d_workspace_bytes of the pose decoder (16-position 64-column engine): header + hand-off granules (4 samples
    per 64-column tile), rounded to 256 bytes, + one row set per sample (upper bound of one grasp per cloud) of 2 C floats
    per ResnetBlock, rounded again, + 64 KiB per workgroup where the 256-channel level parks its residual stream."""
    import ctypes
    from graspldm_amd import _lib
    from graspldm_amd.pipeline import build_fpc_ldm
    from graspldm_amd.r1d_pack import pack_resnet1d
    dec = build_fpc_ldm(device=None).vae_model.decoder
    sd = {k: v.detach().float().cpu() for k, v in dec.state_dict().items()}
    packed = pack_resnet1d(sd, "net.", groups=dec.net.groups, seq_len=dec.feature_resolution, cond_rows=3, decoder=dict(
        in_w=sd["in_layer.weight"], in_b=sd["in_layer.bias"], tmrp_w=sd["tmrp.weight"], tmrp_b=sd["tmrp.bias"],
        cls_w=sd["class_logits.weight"], cls_b=sd["class_logits.bias"]))
    d = packed["desc"]
    ptr = ctypes.cast(ctypes.pointer(d), ctypes.c_void_p)
    dims = list(d.dims)[: d.n_levels + 1]
    assert dims == [16, 32, 64, 128, 256] and d.emb_dim == 64 and d.seq_len == 16
    rows = sum(2 * dims[i // 2] for i in range(2 * d.n_levels)) + 2 * dims[d.n_levels]
    n = 50
    assert _lib.lib().gldm_r1d_tile_columns(ptr) == 64
    tiles = (n + 3) // 4
    base = (256 + tiles * 64 * 8 + 255) // 256 * 256 + n * rows * 4
    assert _lib.lib().gldm_r1d_workspace_bytes(ptr, n) == (base + 255) // 256 * 256 + tiles * 65536
    d.emb_dim = 32   # not a shape of the 64-column engine -> the sample-major one: 2 samples per 32-column tile, no park
    assert _lib.lib().gldm_r1d_tile_columns(ptr) == 32
    base = 256 + ((n + 1) // 2) * 32 * 8
    assert _lib.lib().gldm_r1d_workspace_bytes(ptr, n) == (base + 255) // 256 * 256 + n * rows * 4
    d.emb_dim = 64


def test_folded_prenorm_qkv_block_of_the_descriptor(fpc_state_dict):
    """ABI 4: per level, to_qkv with the PreNorm LayerNorm gain folded in (W' = W diag(g), to_qkv's own row order,
    MFMA fragment order) and its row sums s = W' 1, so that W LN(x) = rstd (W' x - mean s) (csrc/resnet1d.hip:
    qkv_ln_pm).  Checked entry by entry against the state dict, and the identity itself against torch's LayerNorm form
    (resnets.py:104-124)."""
    from graspldm_amd.r1d_pack import pack_resnet1d
    pre = "diffusion_model.model."
    p = pack_resnet1d(fpc_state_dict, pre, groups=4, seq_len=4, num_steps=1000)
    d, w = p["desc"], p["weights"]
    for lv, C in enumerate([4, 32, 64, 128]):
        q = pre + f"blocks.{lv}.2."
        wq = fpc_state_dict[q + "fn.fn.to_qkv.weight"][:, :, 0].float()
        g = fpc_state_dict[q + "fn.norm.g"].float().reshape(-1)
        wn = (wq.double() * g.double().reshape(1, -1)).float()
        kb = (C + 15) // 16
        frag = w[d.lv[lv].qkvn_w:d.lv[lv].qkvn_w + 24 * kb * 256].view(24, kb, 64, 4)
        for r, k in ((0, 0), (37, C - 1), (383, C // 2), (200, 1)):
            assert frag[r // 16, k // 16, (r % 16) + 16 * (k % 4), (k % 16) // 4] == wn[r, k]
        s = w[d.lv[lv].qkvn_s:d.lv[lv].qkvn_s + 384]
        assert torch.allclose(s, wn.sum(dim=1), rtol=0, atol=1e-5)
        x = torch.randn(C, 7, generator=torch.Generator().manual_seed(lv)) * 2.0 + 3.0   # columns with an offset
        mean, var = x.mean(dim=0, keepdim=True), x.var(dim=0, unbiased=False, keepdim=True)
        ref = wq @ ((x - mean) * (var + 1e-5).rsqrt() * g.reshape(-1, 1))
        fold = (wn @ x - mean * s.reshape(-1, 1)) * (var + 1e-5).rsqrt()
        assert (ref - fold).abs().max() < 2e-5 * max(1.0, ref.abs().max().item())


def test_modules_refuse_cpu_tensors():
    ldm = build_fpc()
    with pytest.raises(RuntimeError, match="CUDA tensor"):
        ldm.vae_model.encode_pc(torch.zeros(1, 1024, 3))
    with pytest.raises(RuntimeError, match="CUDA tensor|GPU only"):
        ldm.generate_grasps(torch.zeros(1, 1024, 3), num_grasps=2)


def test_split_f16_fragments_are_tight_and_in_mfma_order():
    """mfma_a_fragments_f16x2 (include/gldm.h, "Split-f16 weight fragments"): hi + lo == w up to 2^-22 |w| (and 2^-25
    where the lo part is an f16 subnormal), and lane l of fragment (mt, kb, plane) holds
    W[16 mt + (l & 15)][32 kb + 8 (l >> 4) + j].  Values beyond the f16 range are refused."""
    from graspldm_amd.r1d_pack import mfma_a_fragments_f16x2, split_f16x2
    g = torch.Generator().manual_seed(0)
    w = torch.randn(40, 96, generator=g) * torch.logspace(-6, 3, 96).unsqueeze(0)   # wide dynamic range
    hi, lo = split_f16x2(w)
    assert hi.dtype == torch.float16 and lo.dtype == torch.float16
    err = (hi.float() + lo.float() - w).abs()
    assert (err <= torch.maximum(w.abs() * 2.0 ** -22, torch.full_like(w, 2.0 ** -25))).all()
    assert (lo.float().abs() <= hi.float().abs() * 2.0 ** -10 + 2.0 ** -24).all()
    f = mfma_a_fragments_f16x2(w)
    assert f.dtype == torch.float32 and f.numel() == 3 * 3 * 2 * 64 * 4          # 3 m-tiles (40 -> 48 rows) x 3 k-blocks
    fb = f.view(torch.float16).view(3, 3, 2, 64, 8)
    planes = (hi, lo)
    for mt, kb, pl, lane, j in [(0, 0, 0, 0, 0), (1, 2, 1, 37, 5), (2, 1, 1, 63, 7), (2, 0, 0, 8, 3)]:
        row, k = 16 * mt + (lane & 15), 32 * kb + 8 * (lane >> 4) + j
        exp = planes[pl][row, k].item() if row < 40 else 0.0
        assert fb[mt, kb, pl, lane, j].item() == exp, (mt, kb, pl, lane, j)
    with pytest.raises(ValueError):
        mfma_a_fragments_f16x2(torch.randn(16, 48))
    with pytest.raises(ValueError, match="65504"):
        mfma_a_fragments_f16x2(torch.full((16, 32), 7.0e4))


def test_conv3d_split_packing_walks_tap_pairs():
    """pack_conv3d_f16x2: k = ((16-channel block) * 14 + pair) * 32 + 16 (tap - 2 pair) + channel; tap 27 is zero."""
    from graspldm_amd.voxel import pack_conv3d_f16x2, split_conv_supported
    from graspldm_amd.r1d_pack import split_f16x2
    g = torch.Generator().manual_seed(1)
    w = torch.randn(48, 32, 3, 3, 3, generator=g)
    f = pack_conv3d_f16x2(w).view(torch.float16).view(3, 2 * 14, 2, 64, 8)       # [mt][cb * 14 + pair][plane][lane][j]
    hi = split_f16x2(w.reshape(48, 32, 27))[0]
    for co, ci, tap in [(0, 0, 0), (17, 21, 13), (47, 31, 26), (5, 16, 1)]:
        cb, c16, pair, half = ci // 16, ci % 16, tap // 2, tap % 2
        k = 16 * half + c16
        lane, j = (co % 16) + 16 * (k // 8), k % 8
        assert f[co // 16, cb * 14 + pair, 0, lane, j].item() == hi[co, ci, tap].item()
    # the empty half of the last pair
    assert (f[:, 13, :, 32:, :] == 0).all() and (f[:, 27, :, 32:, :] == 0).all()
    assert split_conv_supported(48, 48, 24) and split_conv_supported(96, 96, 12) and not split_conv_supported(3, 32, 32)


def test_conv3d_few_channel_split_packing_is_tap_major_without_padding():
    """pack_conv3d_fewch_f16x2 (the encoder's first voxel conv, 3 -> 48 @ 24^3): k = tap * cin + ci, 81 real columns in
    three 32-deep blocks, zero beyond."""
    from graspldm_amd.voxel import pack_conv3d_fewch_f16x2, split_conv_supported
    from graspldm_amd.r1d_pack import split_f16x2
    g = torch.Generator().manual_seed(2)
    w = torch.randn(48, 3, 3, 3, 3, generator=g)
    f = pack_conv3d_fewch_f16x2(w).view(torch.float16).view(3, 3, 2, 64, 8)       # [mt][kb][plane][lane][j]
    planes = split_f16x2(w.reshape(48, 3, 27))
    for co, ci, tap in [(0, 0, 0), (17, 2, 13), (47, 1, 26), (5, 0, 10), (31, 2, 26)]:
        k = tap * 3 + ci
        kb, g8, j = k // 32, (k % 32) // 8, k % 8
        for pl in range(2):
            assert f[co // 16, kb, pl, (co % 16) + 16 * g8, j].item() == planes[pl][co, ci, tap].item()
    # columns 81 .. 95: lane groups 2 (j >= 1) and 3 of the last block
    assert (f[:, 2, :, 48:, :] == 0).all() and (f[:, 2, :, 32:48, 1:] == 0).all()
    assert split_conv_supported(3, 48, 24)


def test_pad_cin32_and_16_position_descriptors():
    """16-channel levels of the 64-column engines read one zero-padded 32-channel block of planes: the packer pads every
    tap's channels to 32 with zero weights (r1d_pack.pad_cin32) and provides every split-f16 copy for both 16-position
    nets of the shipped experiments (pose decoder, ppc denoiser), which then run on 64-column tiles."""
    import ctypes
    from graspldm_amd import _lib
    from graspldm_amd.pipeline import build_fpc_ldm
    from graspldm_amd.r1d_pack import pack_resnet1d, pad_cin32
    w = torch.arange(2 * 3 * 16, dtype=torch.float32).reshape(2, 48)          # [cout 2, taps 3 x cin 16]
    p = pad_cin32(w, 16, 3)
    assert p.shape == (2, 96)
    assert torch.equal(p.reshape(2, 3, 32)[:, :, :16], w.reshape(2, 3, 16)) and p.reshape(2, 3, 32)[:, :, 16:].abs().sum() == 0
    assert pad_cin32(torch.ones(4, 64), 64, 1).shape == (4, 64)                # multiples of 32 pass through
    ldm = build_fpc_ldm(device=None, latent=16, pc_latent=256)
    den = ldm.diffusion_model.model
    sd = {k: v.detach().float().cpu() for k, v in den.state_dict().items()}
    d = pack_resnet1d(sd, "", groups=den.groups, seq_len=16, num_steps=1000)["desc"]
    assert list(d.dims)[:5] == [16, 32, 64, 128, 256] and d.emb_dim == 64 and d.seq_len == 16
    assert all(d.rb[i].c1_w3 > 0 and d.rb[i].c2_w3 > 0 for i in range(2 * d.n_levels + 1))
    assert all(d.lv[i].qkvn_w3 > 0 and d.lv[i].out_w3 > 0 and d.lv[i].down_w3 > 0 for i in range(d.n_levels))
    ptr = ctypes.cast(ctypes.pointer(d), ctypes.c_void_p)
    assert _lib.lib().gldm_r1d_tile_columns(ptr) == 64
    n = 10   # 3 tiles of 4 samples; no decoder table; park scratch for the 256-channel level
    assert _lib.lib().gldm_r1d_workspace_bytes(ptr, n) == (256 + 3 * 64 * 8 + 255) // 256 * 256 + 3 * 65536
    keep = d.rb[0].c1_w3
    d.rb[0].c1_w3 = 0   # a packer without the padded 16-channel copies -> the sample-major f32 engine
    assert _lib.lib().gldm_r1d_tile_columns(ptr) == 32
    d.rb[0].c1_w3 = keep


def test_f32_only_switch_changes_what_is_packed_and_chosen(fpc_state_dict):
    """numerics.f32_only(): descriptors name no split copy (-> sample-major f32 engine), the split shape predicates say no
    wherever an f32 kernel exists, and the switch restores itself."""
    from graspldm_amd import numerics, voxel, dense
    from graspldm_amd.r1d_pack import pack_resnet1d
    from graspldm_amd.sa_pack import split_plan_ok
    sd = fpc_state_dict
    on = pack_resnet1d(sd, "diffusion_model.model.", groups=4, seq_len=4, num_steps=1000)["desc"]
    assert on.rb[2].c1_w3 > 0 and on.rb[2].c1_wq > 0 and on.lv[1].qkvn_wq > 0
    assert voxel.split_conv_supported(48, 48, 24) and dense.split_supported(768, 96) and split_plan_ok([131], [128, 128, 256], 64)
    with numerics.f32_only():
        assert not numerics.split_enabled()
        off = pack_resnet1d(sd, "diffusion_model.model.", groups=4, seq_len=4, num_steps=1000)["desc"]
        assert all(rb.c1_w3 == rb.c2_w3 == rb.c1_wq == rb.c2_wq == 0 for rb in off.rb)
        assert all(lv.qkvn_w3 == lv.out_w3 == lv.down_w3 == lv.qkvn_wq == lv.out_wq == lv.down_wq == 0 for lv in off.lv)
        assert not voxel.split_conv_supported(48, 48, 24) and not dense.split_supported(768, 96)
        assert not split_plan_ok([131], [128, 128, 256], 64)
        assert not voxel.split_conv_supported(3, 48, 24)      # round 6: the 3-channel conv runs its f32 form too (K padded)
        assert voxel.conv_supported(48, 24)
    assert numerics.split_enabled()


def test_sixteen_position_nets_carry_the_wave_local_copies(fpc_state_dict):
    """csrc/quad16_narrow.h reads the 16 / 32 / 64-channel levels of a 16-position net through quad-ordered split copies with K
    padded to whole 32-channel blocks (r1d_pack: quad_perm32(pad_cin32(...))): present for the pose decoder's three narrow
    levels, absent at 128 channels, sized as the kernel's fragment list expects, and the padded half of a 16-channel level's
    K is zero in the stored fragments."""
    import torch
    from graspldm_amd.r1d_pack import pack_resnet1d, quad_perm32, pad_cin32, mfma_a_fragments_f16x2
    sd = fpc_state_dict
    p = "vae_model.decoder."
    pk = pack_resnet1d(sd, p + "net.", groups=4, seq_len=16, decoder=dict(
        in_w=sd[p + "in_layer.weight"], in_b=sd[p + "in_layer.bias"], tmrp_w=sd[p + "tmrp.weight"],
        tmrp_b=sd[p + "tmrp.bias"], cls_w=sd[p + "class_logits.weight"], cls_b=sd[p + "class_logits.bias"]))
    d, w = pk["desc"], pk["weights"]
    assert [d.dims[i] for i in range(5)] == [16, 32, 64, 128, 256]
    for i in range(6):
        assert d.rb[i].c1_wq > 0 and d.rb[i].c2_wq > 0
    for i in range(3):
        assert d.lv[i].qkvn_wq > 0 and d.lv[i].out_wq > 0 and d.lv[i].down_wq > 0
    assert d.rb[6].c1_wq == 0 and d.lv[3].qkvn_wq == 0
    # a 16 -> 16 conv: one m-tile x 3 taps x one padded block = 3 fragments of 512 floats; slot 8 g + j of a block holds
    # channel 16 (j >> 2) + 4 g + (j & 3): j >= 4 are the zero-padded channels 16 .. 31
    frag = w[d.rb[0].c1_wq:d.rb[0].c1_wq + 3 * 512].view(torch.float16).reshape(3, 2, 64, 8)   # (fragment, plane, lane, j)
    assert float(frag[:, :, :, 4:].abs().max()) == 0.0 and float(frag[:, 0, :, :4].abs().max()) > 0.0
    x = torch.arange(2 * 96, dtype=torch.float32).reshape(2, 96)
    assert torch.equal(quad_perm32(x)[:, 8 * 1 + 5], x[:, 16 * 1 + 4 * 1 + 1])      # block 0, g = 1, j = 5
    assert mfma_a_fragments_f16x2(quad_perm32(pad_cin32(torch.ones(16, 48), 16, 3))).numel() == 3 * 512
