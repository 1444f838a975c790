#!/usr/bin/env python3
"""Experiment (round 5): the headline pass as TWO half batches on two HIP streams (two model instances: a plan owns its workspace).
Question: do the second stream's workgroups fill the ramps / tails / one-workgroup-per-CU level-2 launches of the first?
    python benchmarks/dual_stream.py [--mode split3|bf16] [--batch 32] [--parts 2]"""
import argparse
import copy
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import k_diffusion_amd as K  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mode", default="split3")
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--parts", type=int, default=2)
    ap.add_argument("--passes", type=int, default=4)
    ap.add_argument("--config", default="configs/config_oxford_flowers.json")
    args = ap.parse_args()
    os.environ["KDIFF_GEMM"] = args.mode
    dev = torch.device("cuda")
    cfg = K.config.load_config(args.config)
    mc = cfg["model"]
    model = K.config.make_model(cfg).eval().requires_grad_(False)
    model.load_state_dict(K.synth.synth_state_dict(model.state_dict(), seed=0))
    model = model.to(dev)
    models = [model] + [copy.deepcopy(model) for _ in range(args.parts - 1)]
    dens = [K.Denoiser(m, sigma_data=mc["sigma_data"]) for m in models]
    shape = (mc["input_channels"], *mc["input_size"])
    x0 = K.synth.synth_noise_batch(shape, 0, 0, args.batch, mc["sigma_max"]).to(dev)
    sigmas = K.sampling.get_sigmas_karras(50, mc["sigma_min"], mc["sigma_max"], rho=7., device=dev)
    streams = [torch.cuda.Stream() for _ in range(args.parts)]
    per = args.batch // args.parts
    xs = [x0[i * per:(i + 1) * per].contiguous() for i in range(args.parts)]

    def single():
        return K.sampling.sample_dpmpp_2m(dens[0], x0, sigmas, disable=True)

    def split():
        outs = []
        cur = torch.cuda.current_stream()
        for s in streams:
            s.wait_stream(cur)
        for i, s in enumerate(streams):
            with torch.cuda.stream(s):
                outs.append(K.sampling.sample_dpmpp_2m(dens[i], xs[i], sigmas, disable=True))
        for s in streams:
            cur.wait_stream(s)
        return torch.cat(outs)

    for name, fn in (("single stream, batch %d" % args.batch, single), ("%d streams x batch %d" % (args.parts, per), split), ("single again", single)):
        y = fn()
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(args.passes):
            y = fn()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t) / args.passes
        print(f"{args.mode}: {name}: {dt * 1e3:.1f} ms per pass, {args.batch / dt:.1f} images/s", flush=True)
        if name.startswith("single stream"):
            ref = y
        else:
            print("   max rel diff vs single:", float((y - ref).abs().max() / ref.abs().max()))


if __name__ == "__main__":
    main()
