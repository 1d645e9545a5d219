"""Device-side handle of one packed 1-D ResNet (denoiser or pose decoder):
owns the packed weight buffer on the GPU and calls the fused HIP engine through
the C ABI (gldm_r1d_cond_embed / gldm_denoise / gldm_decode)."""
import ctypes

import torch

from . import _lib as L
from .r1d_pack import SCHED_DDIM, SCHED_DDPM, SCHED_NONE, SCHED_COEF_STRIDE, pack_resnet1d  # noqa: F401


class R1dEngine:
    def __init__(self, packed, device):
        self.device = torch.device(device)
        self.desc = packed["desc"]
        self.weights = packed["weights"].to(self.device)
        self.temb = packed["temb"].to(self.device) if packed["temb"] is not None else None
        self.cond_w = packed["cond_w"].to(self.device)
        self.cond_b = packed["cond_b"].to(self.device)
        self.seq_len = int(self.desc.seq_len)
        self._ws = {}   # scratch per HIP stream: launches on different streams may run concurrently
        self._probe = {}  # per workspace: (pinned host copy of its hand-off error word, event of that copy)

    # -- helpers
    def _desc_ptr(self):
        return ctypes.cast(ctypes.pointer(self.desc), ctypes.c_void_p)

    def _workspace(self, n):
        with torch.cuda.device(self.device):   # the size depends on the device's compute-unit count (park scratch per workgroup)
            need = L.lib().gldm_r1d_workspace_bytes(self._desc_ptr(), int(n))
        if need < 0:
            raise L.GldmError("gldm_r1d_workspace_bytes: this ResNet1D configuration is not supported by the HIP engine")
        key = torch.cuda.current_stream(self.device).cuda_stream
        ws = self._ws.get(key)
        if ws is None or ws.numel() < need:
            # zeroed once (contract of gldm_r1d_workspace_bytes); the library re-arms it after every launch
            ws = self._ws[key] = torch.zeros(int(need), dtype=torch.uint8, device=self.device)
        return ws

    def workspace_errors(self):
        """Sum of the hand-off error words of every workspace (0 unless a bounded wait expired); synchronises."""
        return int(sum(int(ws[12:16].view(torch.int32).item()) for ws in self._ws.values()))

    # A step-segment hand-off whose bounded wait expires (csrc/resnet1d.hip: chain_take) poisons its tile with NaN
    # and sets the workspace's error word.  The word is copied to pinned host memory behind every multi-step launch
    # (4 bytes, asynchronous, same stream) and looked at without a host sync at the next launch on that workspace,
    # and with one in check(), which the inference harness calls before results leave the device.
    def _arm_probe(self, key, ws):
        host, _ = self._probe.get(key, (None, None))
        if host is None:
            host = torch.zeros(1, dtype=torch.int32).pin_memory()
        host.copy_(ws[12:16].view(torch.int32), non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(self.device))
        self._probe[key] = (host, ev)

    def _raise_if_failed(self, key, wait):
        host, ev = self._probe.get(key, (None, None))
        if ev is None or not (wait or ev.query()):
            return
        if wait:
            ev.synchronize()
        if int(host[0]) != 0:
            host[0] = 0
            self._ws[key][12:16].zero_()
            raise L.GldmError("gldm_denoise: a step-segment hand-off between workgroups timed out (workspace error word "
                              "set); the latents of that launch are invalid (NaN) and must be discarded")

    def check(self):
        """Raise GldmError if any launch so far lost a hand-off (waits for the pending error-word copies only)."""
        for key in list(self._probe):
            self._raise_if_failed(key, wait=True)

    def cond_embed(self, z_cond):
        """input_emb_layers (Linear + SiLU) on [n, R, Dc] (or [n, Dc]) -> [n, R, E]."""
        z = z_cond if z_cond.ndim == 3 else z_cond.unsqueeze(1)
        z = z.contiguous().float()
        n, r, dc = z.shape
        e = self.cond_w.shape[0]
        if dc != self.cond_w.shape[1]:   # the launch strides the weight rows by dc: another width reads across rows
            raise RuntimeError(f"z_cond has {dc} features per row; this network's conditioning Linear takes {self.cond_w.shape[1]}")
        out = torch.empty((n, r, e), dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            L.call("gldm_r1d_cond_embed", L.ptr(z), L.ptr(self.cond_w), L.ptr(self.cond_b), n, r, dc, e, L.ptr(out),
                   L.current_stream(self.device))
        return out

    def denoise(self, x_in, cemb, samples_per_cond, timesteps=None, sample_t=None, sched_kind=SCHED_NONE,
                clip_sample=True, coef=None, step_noise=None, sample_emb=None, temb=None):
        """x_in [n, 1, L] -> x after all steps (or eps when sched_kind == NONE).  sample_emb [n, E]: per-sample
        embedding added to the time embedding (class conditioning).  temb [T, E]: time-embedding table to use instead of
        the engine's integer-timestep one (continuous-time samplers: one row per step)."""
        n = x_in.shape[0]
        x_in = x_in.contiguous().float()
        if sample_emb is not None:
            sample_emb = sample_emb.reshape(n, -1).contiguous().float()
            if sample_emb.shape[1] != self.cond_w.shape[0]:
                raise RuntimeError(f"sample_emb must be [n, {self.cond_w.shape[0]}]")
        out = torch.empty_like(x_in)
        n_steps = 1 if timesteps is None else int(timesteps.numel())
        ws = self._workspace(n)
        key = torch.cuda.current_stream(self.device).cuda_stream
        self._raise_if_failed(key, wait=False)
        with torch.cuda.device(self.device):
            tab = self.temb if temb is None else temb.contiguous().float()
            L.call("gldm_denoise", self._desc_ptr(), L.ptr(self.weights), L.ptr(tab), L.ptr(cemb),
                   int(samples_per_cond), L.ptr(x_in), n, L.ptr(timesteps), L.ptr(sample_t), n_steps, int(sched_kind),
                   1 if clip_sample else 0, L.ptr(coef), L.ptr(step_noise), L.ptr(sample_emb), L.ptr(out), L.ptr(ws),
                   L.current_stream(self.device))
            if n_steps > 1:   # only multi-step launches can split a tile's steps over workgroups
                self._arm_probe(key, ws)
        return out

    def denoise_rng(self, x_in, cemb, samples_per_cond, timesteps, coef, noise_seed, noise_base=0, clip_sample=True,
                    sample_emb=None):
        """DDPM reverse loop with the per-step noise drawn inside the kernel (gldm_denoise_rng: Philox4x32-10 keyed on
        `noise_seed`, counter = (noise_base + latent index, position block, step)).  No [steps, n, 1, L] noise tensor
        exists; results do not depend on how a batch is split when each part passes its first latent's global index."""
        n = x_in.shape[0]
        x_in = x_in.contiguous().float()
        if sample_emb is not None:
            sample_emb = sample_emb.reshape(n, -1).contiguous().float()
            if sample_emb.shape[1] != self.cond_w.shape[0]:
                raise RuntimeError(f"sample_emb must be [n, {self.cond_w.shape[0]}]")
        out = torch.empty_like(x_in)
        n_steps = int(timesteps.numel())
        ws = self._workspace(n)
        key = torch.cuda.current_stream(self.device).cuda_stream
        self._raise_if_failed(key, wait=False)
        with torch.cuda.device(self.device):
            L.call("gldm_denoise_rng", self._desc_ptr(), L.ptr(self.weights), L.ptr(self.temb), L.ptr(cemb),
                   int(samples_per_cond), L.ptr(x_in), n, L.ptr(timesteps), n_steps, 1 if clip_sample else 0, L.ptr(coef),
                   int(noise_seed) & 0xFFFFFFFFFFFFFFFF, int(noise_base), L.ptr(sample_emb), L.ptr(out), L.ptr(ws),
                   L.current_stream(self.device))
            if n_steps > 1:
                self._arm_probe(key, ws)
        return out

    def decode(self, z_h, cemb, samples_per_cond):
        n = z_h.shape[0]
        z_h = z_h.contiguous().float()
        tmrp = torch.empty((n, 6), dtype=torch.float32, device=self.device)
        logit = torch.empty((n, 1), dtype=torch.float32, device=self.device)
        ws = self._workspace(n)
        with torch.cuda.device(self.device):
            L.call("gldm_decode", self._desc_ptr(), L.ptr(self.weights), L.ptr(cemb), int(samples_per_cond), L.ptr(z_h),
                   n, L.ptr(tmrp), L.ptr(logit), L.ptr(ws), L.current_stream(self.device))
        return tmrp, logit


def step_noise_rng(noise_seed, noise_base, step, n, seq_len, device):
    """[n, seq_len] unit normals: exactly what denoise_rng adds at `step` to latents noise_base .. noise_base + n - 1."""
    out = torch.empty((n, seq_len), dtype=torch.float32, device=device)
    with torch.cuda.device(device):
        L.call("gldm_step_noise_rng", int(noise_seed) & 0xFFFFFFFFFFFFFFFF, int(noise_base), int(step), int(n), int(seq_len),
               L.ptr(out), L.current_stream(device))
    return out


def _per_cloud_rows(t, n_clouds, name):
    """[6] / [1,6] / [n_clouds,6] -> contiguous f32 [n_clouds,6] (the reference broadcasts a
    [1,6] std against a [B,6] mean through unsqueeze(-2): tools/inference.py:64-94)."""
    t = t.reshape(-1, 6).float()
    if t.shape[0] == 1 and n_clouds > 1:
        t = t.expand(n_clouds, 6)
    if t.shape[0] != n_clouds:
        raise RuntimeError(f"{name} has {t.shape[0]} rows; expected 1 or {n_clouds} (one per cloud)")
    return t.contiguous()


def pose_epilogue(tmrp, logit, grasp_mean, grasp_std, grasps_per_cloud):
    """unnormalise + tmrp_to_H + sigmoid in one launch (tools/inference.py:628-647).
    grasp_mean / grasp_std: [n_clouds,6] or broadcastable [1,6] / [6], independently."""
    n = tmrp.shape[0]
    dev = tmrp.device
    gpc = int(grasps_per_cloud)
    if gpc <= 0 or n % gpc:
        raise RuntimeError(f"{n} grasps do not split into clouds of {gpc}")
    n_clouds = n // gpc
    # every pointer handed to the launch stays referenced by a local until the call returns
    tm = tmrp.contiguous().float()
    lg = logit.contiguous().float() if logit is not None else None
    gm = _per_cloud_rows(grasp_mean.to(dev), n_clouds, "grasp_mean")
    gs = _per_cloud_rows(grasp_std.to(dev), n_clouds, "grasp_std")
    H = torch.empty((n, 4, 4), dtype=torch.float32, device=dev)
    un = torch.empty((n, 6), dtype=torch.float32, device=dev)
    conf = torch.empty((n, 1), dtype=torch.float32, device=dev) if lg is not None else None
    with torch.cuda.device(dev):
        L.call("gldm_pose_epilogue", L.ptr(tm), L.ptr(lg), L.ptr(gm), L.ptr(gs), n, gpc, n_clouds,
               L.ptr(H), L.ptr(un), L.ptr(conf), L.current_stream(dev))
    return H, un, conf
