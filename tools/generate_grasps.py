#!/usr/bin/env python3
"""Grasp generation CLI: same flags as the reference's `tools/generate_grasps.py:14-61`
(--exp_path --data_root --mode --split --num_grasps --visualize --no_ema --num_samples
--conditioning --condition_value --inference_steps) on the MI355X path.

Additive flags: --pc_file FILE [--num_points N] (generate on a sensor cloud read from .npy / .npz / .ply / .xyz:
the reference's `generate_on_pointcloud`, grasp_ldm/inference/inference_base.py:161-212, which its own CLI does not
reach -- it only iterates ACRONYM items, tools/generate_grasps.py:109-131), --device, --seed, --synthetic N (run on N-point synthetic object clouds with
the synthetic weight recipe when no experiment directory / ACRONYM data is available; there is
no network here for either), --out FILE.npz.  `--inference_steps` is honoured (the reference
silently ignores it: it passes use_fast_sampler=False, tools/generate_grasps.py:69-79).
"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from graspldm_amd.inference import Conditioning, InferenceLDM, InferenceVAE  # noqa: E402


def parse_args(argv=None):
    p = argparse.ArgumentParser(description="Grasp Generation Script (MI355X)")
    p.add_argument("--exp_path", type=str, default=None, help="Path to experiment checkpoint")
    p.add_argument("--data_root", type=str, default="data/ACRONYM", help="Root directory for data")
    p.add_argument("--mode", type=str, choices=["VAE", "LDM"], default="VAE", help="Model type to use")
    p.add_argument("--split", type=str, default="test", help="Data split to use")
    p.add_argument("--num_grasps", type=int, default=20, help="Number of grasps to generate")
    p.add_argument("--visualize", action="store_true", help="Enable visualization")
    p.add_argument("--no_ema", action="store_false", dest="use_ema_model", help="Disable EMA model usage")
    p.add_argument("--num_samples", type=int, default=11, help="Number of samples to generate")
    p.add_argument("--conditioning", type=str, choices=["unconditional", "class", "region"], default="unconditional")
    p.add_argument("--condition_value", type=int, help="Value for conditioning (class label or region ID)")
    p.add_argument("--inference_steps", type=int, default=100, help="Number of inference steps for LDM")
    p.add_argument("--device", type=str, default="cuda:0")
    p.add_argument("--seed", type=int, default=None)
    p.add_argument("--synthetic", type=int, default=0, metavar="N",
                   help="use N-point synthetic clouds and synthetic weights (no experiment dir needed)")
    p.add_argument("--pc_file", type=str, action="append", default=None, metavar="FILE",
                   help="generate on this raw cloud ([N,3], metres; .npy .npz .ply .xyz); may be repeated. The cloud is "
                        "brought to the encoder's point count and normalised like generate_on_pointcloud does")
    p.add_argument("--num_points", type=int, default=None,
                   help="encoder point count for --pc_file (default: the model's pc encoder n_points)")
    p.add_argument("--random_resample", action="store_true",
                   help="with --pc_file: random subsampling instead of farthest-point selection")
    p.add_argument("--out", type=str, default=None, help="write results of all samples to this .npz")
    return p.parse_args(argv)


def setup_model(args):
    if args.synthetic:
        from graspldm_amd.pipeline import build_fpc_ldm
        ldm = build_fpc_ldm(n_points=args.synthetic, scheduler="ddim")
        if args.mode == "LDM":
            return InferenceLDM(model=ldm, num_inference_steps=args.inference_steps, device=args.device)
        return InferenceVAE(model=ldm.vae_model, device=args.device)
    if not args.exp_path:
        raise SystemExit("--exp_path is required (or use --synthetic N)")
    exp_name, exp_root = os.path.basename(args.exp_path.rstrip("/")), os.path.dirname(args.exp_path.rstrip("/"))
    if args.mode == "LDM":
        model = InferenceLDM(exp_name=exp_name, exp_out_root=exp_root, data_root=args.data_root,
                             num_inference_steps=args.inference_steps, use_fast_sampler=True,
                             data_split=args.split, use_ema_model=args.use_ema_model, device=args.device)
        dm = model.model.diffusion_model
        print(f"Trained using noise schedule: beta0 = {dm.beta_start} ; betaT = {dm.beta_end}")
        return model
    return InferenceVAE(exp_name=exp_name, exp_out_root=exp_root, data_root=args.data_root, data_split=args.split,
                        use_ema_model=args.use_ema_model, device=args.device)


def main(argv=None):
    args = parse_args(argv)
    if args.conditioning != "unconditional":
        raise SystemExit("class / region conditioned models are not shipped with the reference (out of scope)")
    if args.visualize:
        print("visualisation (trimesh/pyrender) is out of scope on this path; ignoring --visualize")
    if args.seed is not None:
        torch.manual_seed(args.seed)
        np.random.seed(args.seed)
    model = setup_model(args)
    from graspldm_amd.synthetic import normalize_cloud, synthetic_cloud
    results = []
    if args.pc_file:
        from graspldm_amd.pointcloud import read_cloud_file
        n_pts = args.num_points or encoder_points(model.model)
        for path in args.pc_file:
            pc = torch.from_numpy(read_cloud_file(path))
            res = model.infer_on_pointcloud(pc, num_grasps=args.num_grasps, num_points=n_pts,
                                            use_farthest_point=not args.random_resample)
            conf = res["confidence"].flatten()
            print(f"{path}: {pc.shape[0]} points -> {n_pts}; grasps {tuple(res['grasps'].shape)}  "
                  f"confidence mean {conf.mean().item():.3f}  best {conf.max().item():.3f}")
            results.append(res)
        return finish(args, results)
    for i in range(args.num_samples):
        if args.synthetic:
            idx = int(np.random.randint(0, 1 << 20))
            pc, metas = normalize_cloud(synthetic_cloud(idx, args.synthetic))
            metas = {k: (v.unsqueeze(0) if isinstance(v, torch.Tensor) else v) for k, v in metas.items()}
        else:
            raise SystemExit("ACRONYM dataset loading is out of scope: pass the object's cloud with --pc_file FILE "
                             "(.npy / .ply / ...), or run on synthetic clouds with --synthetic N")
        res = model.generate_grasps(pc, metas, num_grasps=args.num_grasps)
        conf = res["confidence"].flatten()
        print(f"sample {i}: cloud #{idx}  grasps {tuple(res['grasps'].shape)}  "
              f"confidence mean {conf.mean().item():.3f}  best {conf.max().item():.3f}")
        results.append(res)
    return finish(args, results)


def encoder_points(model):
    """n_points of the model's cloud encoder (its out_layer[1] is a Linear over the point axis: N is fixed)."""
    vae = getattr(model, "vae_model", model)
    enc = vae.encoder.pc_encoder if hasattr(vae, "encoder") else vae.pc_encoder
    try:
        return int(enc.out_layer[1].in_features)
    except (AttributeError, IndexError, TypeError):
        raise SystemExit("cannot infer the encoder's point count; pass --num_points")


def finish(args, results):
    if args.out:
        np.savez_compressed(args.out, grasps=torch.cat([r["grasps"] for r in results]).cpu().numpy(),
                            grasp_tmrp=torch.cat([r["grasp_tmrp"] for r in results]).cpu().numpy(),
                            confidence=torch.cat([r["confidence"] for r in results]).cpu().numpy())
        print("wrote", args.out)
    return results


if __name__ == "__main__":
    main()
