"""Host-side weight preparation for the fused 1-D ResNet engine
(csrc/resnet1d.hip, C ABI `gldm_r1d_desc` in include/gldm.h).

Runs once at model-load time (step-invariant work hoisted out of the denoise
loop): weight standardisation of every `Block.proj` (resnets.py:85-101, fp32
eps 1e-5), re-layout of every conv / 1x1 weight into v_mfma_f32_16x16x4_f32
A-fragment order, the combined scale/shift bias, and the time_mlp table
[T, E] (resnets.py:44-56,517-522), evaluated with the same torch CPU ops the
reference module uses so that large-argument sin/cos are identical.
"""
import ctypes
import math

import torch
import torch.nn.functional as F

MAX_LEVELS = 6
MAX_RESBLOCKS = 2 * MAX_LEVELS + 1
SCHED_NONE, SCHED_DDIM, SCHED_DDPM = 0, 1, 2
SCHED_COEF_STRIDE = 8


class R1dResblock(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int32) for n in
                ("c1_w", "c1_b", "n1_w", "n1_b", "c2_w", "c2_b", "n2_w", "n2_b", "ss_w", "ss_b", "c1_w3", "c2_w3",
                 "c1_wq", "c2_wq")]


class R1dLevel(ctypes.Structure):
    _fields_ = [("ln_g", ctypes.c_int32), ("qkv_w", ctypes.c_int32 * 2), ("out_w", ctypes.c_int32),
                ("out_b", ctypes.c_int32), ("ln2_g", ctypes.c_int32), ("down_w", ctypes.c_int32),
                ("down_b", ctypes.c_int32), ("qkvn_w", ctypes.c_int32), ("qkvn_s", ctypes.c_int32),
                ("qkvn_w3", ctypes.c_int32), ("out_w3", ctypes.c_int32), ("down_w3", ctypes.c_int32),
                ("qkvn_wq", ctypes.c_int32), ("out_wq", ctypes.c_int32), ("down_wq", ctypes.c_int32)]


class R1dDesc(ctypes.Structure):
    """Mirror of `gldm_r1d_desc` (include/gldm.h)."""
    _fields_ = [("seq_len", ctypes.c_int32), ("n_levels", ctypes.c_int32),
                ("dims", ctypes.c_int32 * (MAX_LEVELS + 1)),
                ("emb_dim", ctypes.c_int32), ("cond_rows", ctypes.c_int32), ("groups", ctypes.c_int32),
                ("init_w", ctypes.c_int32), ("init_b", ctypes.c_int32),
                ("ss_rows", ctypes.c_int32),
                ("rb", R1dResblock * MAX_RESBLOCKS), ("lv", R1dLevel * MAX_LEVELS),
                ("final_w", ctypes.c_int32), ("final_b", ctypes.c_int32),
                ("latent_dim", ctypes.c_int32), ("in_w", ctypes.c_int32), ("in_b", ctypes.c_int32),
                ("head_w", ctypes.c_int32), ("head_b", ctypes.c_int32), ("n_head", ctypes.c_int32)]


def mfma_a_fragments(w2d):
    """[M, K] -> flat buffer in fragment order [M/16][K/16][lane 64][4]: lane l of
    k-step j inside block kb holds W[16 mt + (l & 15)][16 kb + 4 j + (l >> 4)]."""
    m, k = w2d.shape
    mt, kb = (m + 15) // 16, (k + 15) // 16
    wp = torch.zeros(mt * 16, kb * 16, dtype=torch.float32)
    wp[:m, :k] = w2d
    wp = wp.view(mt, 16, kb, 4, 4)            # (mt, i, kb, j, kq):  k = 16 kb + 4 j + kq
    return wp.permute(0, 2, 4, 1, 3).contiguous().reshape(-1)   # (mt, kb, kq, i, j): lane = 16 kq + i


class SplitRangeError(ValueError):
    """A weight does not fit the f16 split (|w| >= 65504): the caller packs the layer for its f32-pipe kernel instead."""


def split_f16x2(w):
    """f32 -> (hi, lo) f16, round to nearest at each stage: hi + lo == w up to 2^-22 |w| while |w| >= 2^-3; below that lo
    is an f16 SUBNORMAL (the matrix pipe keeps those: tools/micro/mfma_f16_split) and the error is 2^-25 ABSOLUTE -- a
    weight of 0.03 keeps ~20 bits, one of 1e-3 ~15 (what the CPU restatement measures: 1.7e-6 on a denoiser forward against
    5.4e-7 for three bf16 pieces; the bar is 2e-5).  Values beyond the f16 range (|w| >= 65504) raise SplitRangeError."""
    w = w.detach().float()
    if w.numel() and not float(w.abs().max()) < 65504.0:
        raise SplitRangeError("split-f16 operands must stay below 65504 in magnitude (got %g)" % float(w.abs().max()))
    hi = w.to(torch.float16)
    lo = (w - hi.float()).to(torch.float16)
    return hi, lo


def mfma_a_fragments_f16x2(w2d):
    """[M, K] (K % 32 == 0) -> f32-typed bit container of the split-f16 A fragments of v_mfma_f32_16x16x32_f16:
    [M/16][K/32][plane hi|lo][lane 64][8 f16], lane l = W[16 mt + (l & 15)][32 kb + 8 (l >> 4) + j]
    (include/gldm.h).  512 floats per (m-tile, 32-deep k-block)."""
    m, k = w2d.shape
    if k % 32:
        raise ValueError("split-f16 fragments need K % 32 == 0")
    mt = (m + 15) // 16
    wp = torch.zeros(mt * 16, k, dtype=torch.float32)
    wp[:m] = w2d
    planes = torch.stack(split_f16x2(wp))                          # [2, M, K] f16
    planes = planes.view(2, mt, 16, k // 32, 4, 8)                 # (plane, mt, i, kb, g, j): k = 32 kb + 8 g + j
    frag = planes.permute(1, 3, 0, 4, 2, 5).contiguous()           # (mt, kb, plane, g, i, j): lane = 16 g + i
    return frag.reshape(-1, 8).view(torch.float32).reshape(-1)     # bit pattern kept: 8 f16 = 4 floats


def quad_perm32(w2d):
    """Columns of every 32-channel block in the order the wave-local engine of the narrow levels reads them
    (include/gldm.h, "Quad column order"): new[:, 32 b + 8 g + j] = old[:, 32 b + 16 (j >> 2) + 4 g + (j & 3)]."""
    m, k = w2d.shape
    if k % 32:
        raise ValueError("quad order needs K % 32 == 0")
    s = torch.arange(32)
    g, j = s // 8, s % 8
    src = 16 * (j // 4) + 4 * g + (j % 4)
    return w2d.reshape(m, k // 32, 32)[:, :, src].reshape(m, k)


def conv_as_gemm(w):
    """Conv1d weight [Cout, Cin, taps] -> [Cout, taps*Cin] with k = tap*Cin + ci."""
    return w.permute(0, 2, 1).reshape(w.shape[0], -1)


def pad_cin32(w2d, cin, taps):
    """[Cout, taps * cin] (k = tap * cin + ci) -> [Cout, taps * 32] with every tap's channels zero-padded to 32: a
    16-channel level of the 64-column engines reads its activations as one 32-channel block of split planes whose upper
    half is never written (finite leftovers), so the weights of those channels are zero."""
    if cin % 32 == 0:
        return w2d
    cp = (cin + 31) // 32 * 32
    out = torch.zeros(w2d.shape[0], taps, cp, dtype=w2d.dtype)
    out[:, :, :cin] = w2d.reshape(w2d.shape[0], taps, cin)
    return out.reshape(w2d.shape[0], taps * cp)


def weight_standardize(w):
    mean = w.mean(dim=(1, 2), keepdim=True)
    var = w.var(dim=(1, 2), unbiased=False, keepdim=True)
    return (w - mean) * (var + 1e-5).rsqrt()


class _Buf:
    def __init__(self):
        self.parts, self.n = [], 0

    def add(self, t):
        t = t.detach().to(torch.float32).reshape(-1)
        off = self.n
        pad = (-t.numel()) % 4          # keep every section 16-byte aligned
        self.parts.append(t)
        if pad:
            self.parts.append(torch.zeros(pad))
        self.n += t.numel() + pad
        return off

    def tensor(self):
        return torch.cat(self.parts) if self.parts else torch.zeros(0)


def time_embedding_table(sd, p, num_steps):
    """time_mlp evaluated for t = 0..T-1 exactly like the module (int64 time,
    f = ((t*w)*2)*pi in f32, cat(t, sin, cos), Linear, GELU(erf), Linear)."""
    t = torch.arange(num_steps, dtype=torch.long).reshape(-1, 1)
    freqs = t * sd[p + "time_mlp.0.weights"].reshape(1, -1) * 2 * math.pi
    four = torch.cat((t, freqs.sin(), freqs.cos()), dim=-1)
    h = F.linear(four, sd[p + "time_mlp.1.weight"], sd[p + "time_mlp.1.bias"])
    return F.linear(F.gelu(h), sd[p + "time_mlp.3.weight"], sd[p + "time_mlp.3.bias"]).contiguous()


def pack_resnet1d(sd, p, groups, seq_len, cond_rows=3, num_steps=None, decoder=None):
    """sd: flat state dict (CPU f32), p: prefix of the ResNet1D / TimeConditionedResNet1D.
    decoder: None or dict(in_w, in_b, tmrp_w, tmrp_b, cls_w, cls_b) for the pose decoder.
    Returns dict(desc=R1dDesc, weights=f32 tensor, temb=[T,E] or None,
                 cond_w=[E,Dc], cond_b=[E])."""
    sd = {k: v.detach().float().cpu() for k, v in sd.items() if k.startswith(p)}
    d = R1dDesc()
    buf = _Buf()
    init_w = sd[p + "init_conv.weight"]
    if init_w.shape[1] != 1 or init_w.shape[2] != 7:
        raise ValueError("init_conv must be Conv1d(1 -> C0, k=7) (self-conditioning is off on the hot path)")
    n_levels = 0
    while (p + f"blocks.{n_levels}.3.weight") in sd:
        n_levels += 1
    if n_levels > MAX_LEVELS:
        raise ValueError("too many levels")
    dims = [init_w.shape[0]] + [sd[p + f"blocks.{i}.3.weight"].shape[0] for i in range(n_levels)]
    emb = sd[p + "input_emb_layers.0.weight"].shape[0]
    d.n_levels, d.emb_dim, d.cond_rows, d.groups = n_levels, emb, cond_rows, groups
    for i, c in enumerate(dims):
        d.dims[i] = c
    d.init_w = buf.add(init_w.reshape(dims[0], 7))
    d.init_b = buf.add(sd[p + "init_conv.bias"])

    def resblock(q, c, slot):
        rb = d.rb[slot]
        mw, mb = sd[q + "mlp.1.weight"], sd[q + "mlp.1.bias"]          # [2C, E], [2C]
        comb = cond_rows * mb
        comb[:c] = comb[:c] + cond_rows                                 # sum_r (scale_r + 1)
        rb.ss_w = buf.add(mfma_a_fragments(mw))
        rb.ss_b = buf.add(comb)
        w1, w2 = (conv_as_gemm(weight_standardize(sd[q + f"block{i}.proj.weight"])) for i in (1, 2))
        if c % 16 == 0:   # split-f16 copies for the 64-column engines (per tap a multiple of 32 channels: 16 is padded)
            rb.c1_w3 = buf.add(mfma_a_fragments_f16x2(pad_cin32(w1, c, 3)))
            rb.c2_w3 = buf.add(mfma_a_fragments_f16x2(pad_cin32(w2, c, 3)))
            if seq_len == 4 and c in (32, 64):   # the quad engine's copies (csrc/quad_narrow.h)
                rb.c1_wq = buf.add(mfma_a_fragments_f16x2(quad_perm32(w1)))
                rb.c2_wq = buf.add(mfma_a_fragments_f16x2(quad_perm32(w2)))
            elif seq_len == 16 and c in (16, 32, 64):   # csrc/quad16_narrow.h (16 channels: one zero-padded block per tap)
                rb.c1_wq = buf.add(mfma_a_fragments_f16x2(quad_perm32(pad_cin32(w1, c, 3))))
                rb.c2_wq = buf.add(mfma_a_fragments_f16x2(quad_perm32(pad_cin32(w2, c, 3))))
        rb.c1_w = buf.add(mfma_a_fragments(conv_as_gemm(weight_standardize(sd[q + "block1.proj.weight"]))))
        rb.c1_b = buf.add(sd[q + "block1.proj.bias"])
        rb.n1_w = buf.add(sd[q + "block1.norm.weight"])
        rb.n1_b = buf.add(sd[q + "block1.norm.bias"])
        rb.c2_w = buf.add(mfma_a_fragments(conv_as_gemm(weight_standardize(sd[q + "block2.proj.weight"]))))
        rb.c2_b = buf.add(sd[q + "block2.proj.bias"])
        rb.n2_w = buf.add(sd[q + "block2.norm.weight"])
        rb.n2_b = buf.add(sd[q + "block2.norm.bias"])
        if (q + "res_conv.weight") in sd:
            raise ValueError("res_conv (dim != dim_out) does not occur in ResNet1D")

    slot = 0
    for i in range(n_levels):
        c = dims[i]
        q = p + f"blocks.{i}."
        resblock(q + "0.", c, slot)
        resblock(q + "1.", c, slot + 1)
        slot += 2
        lv = d.lv[i]
        lv.ln_g = buf.add(sd[q + "2.fn.norm.g"])
        wqkv = sd[q + "2.fn.fn.to_qkv.weight"][:, :, 0]                 # [384, C]
        hid = wqkv.shape[0] // 3
        if hid != 128:
            raise ValueError("LinearAttention with heads=4, dim_head=32 expected")
        for pr in range(2):
            rows = torch.cat([wqkv[o + 64 * pr:o + 64 * pr + 64] for o in (0, hid, 2 * hid)])
            lv.qkv_w[pr] = buf.add(mfma_a_fragments(rows))
        # PreNorm LayerNorm folded into to_qkv (csrc/resnet1d.hip: qkv_ln_pm / qkv4_pm), to_qkv's own row order:
        # W LN(x) = rstd (W' x - mean s),  W' = W diag(g), s = W' 1; products and sums in f64, rounded once
        wn = (wqkv.double() * sd[q + "2.fn.norm.g"].double().reshape(1, -1)).float()
        lv.qkvn_w = buf.add(mfma_a_fragments(wn))
        lv.qkvn_s = buf.add(wn.double().sum(dim=1).float())
        lv.out_w = buf.add(mfma_a_fragments(sd[q + "2.fn.fn.to_out.0.weight"][:, :, 0]))
        lv.out_b = buf.add(sd[q + "2.fn.fn.to_out.0.bias"])
        lv.ln2_g = buf.add(sd[q + "2.fn.fn.to_out.1.g"])
        lv.down_w = buf.add(mfma_a_fragments(conv_as_gemm(sd[q + "3.weight"])))
        if c % 16 == 0:
            lv.qkvn_w3 = buf.add(mfma_a_fragments_f16x2(pad_cin32(wn, c, 1)))
            lv.down_w3 = buf.add(mfma_a_fragments_f16x2(pad_cin32(conv_as_gemm(sd[q + "3.weight"]), c, 3)))
        lv.out_w3 = buf.add(mfma_a_fragments_f16x2(sd[q + "2.fn.fn.to_out.0.weight"][:, :, 0]))
        if seq_len == 4 and c in (4, 32, 64):
            wo = sd[q + "2.fn.fn.to_out.0.weight"][:, :, 0]
            if c == 4:   # one value per lane: channel ch in row 4 ch of the m-tile
                wo4 = torch.zeros(16, wo.shape[1])
                wo4[0::4] = wo
                wo = wo4
            lv.out_wq = buf.add(mfma_a_fragments_f16x2(quad_perm32(wo)))
            if c != 4:
                # the quad engine multiplies these rows with the NORMALISED column, (x - mean) rstd, split on the spot (no
                # mean / rstd correction behind the GEMM), and exponentiates q and k with v_exp_f32 = 2^x: their rows
                # (the first 2 x 128) carry log2(e)
                wq = wn.double().clone()
                wq[:2 * hid] *= 1.4426950408889634
                lv.qkvn_wq = buf.add(mfma_a_fragments_f16x2(quad_perm32(wq.float())))
                lv.down_wq = buf.add(mfma_a_fragments_f16x2(quad_perm32(conv_as_gemm(sd[q + "3.weight"]))))
        if seq_len == 16 and c in (16, 32, 64):   # csrc/quad16_narrow.h: the same three copies, K padded per tap / to 32
            lv.out_wq = buf.add(mfma_a_fragments_f16x2(quad_perm32(sd[q + "2.fn.fn.to_out.0.weight"][:, :, 0])))
            wq = wn.double().clone()
            wq[:2 * hid] *= 1.4426950408889634
            lv.qkvn_wq = buf.add(mfma_a_fragments_f16x2(quad_perm32(pad_cin32(wq.float(), c, 1))))
            lv.down_wq = buf.add(mfma_a_fragments_f16x2(quad_perm32(pad_cin32(conv_as_gemm(sd[q + "3.weight"]), c, 3))))
        lv.down_b = buf.add(sd[q + "3.bias"])
    resblock(p + "final_res_block.", dims[-1], slot)
    fw = sd[p + "final_conv.weight"]
    if fw.shape[0] != 1:
        raise ValueError("final_conv with one output channel expected (learned variance is off)")
    d.final_w = buf.add(fw.reshape(-1))
    d.final_b = buf.add(sd[p + "final_conv.bias"])
    d.ss_rows = 2 * max(dims)

    if decoder is not None:
        if decoder["in_w"].shape[0] != seq_len:
            raise ValueError("in_layer must map the latent to seq_len features")
        d.latent_dim = decoder["in_w"].shape[1]
        d.in_w = buf.add(decoder["in_w"])
        d.in_b = buf.add(decoder["in_b"])
        d.head_w = buf.add(torch.cat([decoder["tmrp_w"], decoder["cls_w"]]))
        d.head_b = buf.add(torch.cat([decoder["tmrp_b"], decoder["cls_b"]]))
        d.n_head = 7
    else:
        d.latent_dim = 0
    d.seq_len = seq_len
    from .numerics import split_enabled
    if not split_enabled():   # numerics.f32_only(): the descriptor names no split copy -> sample-major engine, exact f32 products
        for rb in d.rb:
            rb.c1_w3 = rb.c2_w3 = rb.c1_wq = rb.c2_wq = 0
        for lv in d.lv:
            lv.qkvn_w3 = lv.out_w3 = lv.down_w3 = lv.qkvn_wq = lv.out_wq = lv.down_wq = 0
    temb = None
    if (p + "time_mlp.1.weight") in sd:
        if num_steps is None:
            raise ValueError("num_steps needed for the time-embedding table")
        temb = time_embedding_table(sd, p, num_steps)
    return dict(desc=d, weights=buf.tensor(), temb=temb,
                cond_w=sd[p + "input_emb_layers.0.weight"].contiguous(),
                cond_b=sd[p + "input_emb_layers.0.bias"].contiguous())
