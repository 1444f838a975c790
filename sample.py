#!/usr/bin/env python3
"""Samples from k-diffusion image_transformer_v2 models on MI355X (drop-in for the reference's sample.py).

Keeps the reference CLI (sample.py:19-30: --batch-size --checkpoint --config -n --prefix --steps, output
``{prefix}_{i:05}.png``) and its flow (load_config -> make_model -> load safetensors -> Denoiser ->
get_sigmas_karras -> compute_features(gather) -> PNG).  One process per GPU: run it plainly for one GPU or under
``python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 sample.py ...`` for N (RCCL
all-gather of the finished images, k_diffusion/evaluation.py:87).

Additive flags (the reference hard-codes sample_lms, an unseeded rank-local randn and no class conditioning,
sample.py:59-60, and therefore cannot run its own class-conditional configs):
  --sampler NAME       any K.sampling.sample_* (default: lms, like the reference)
  --seed S             per-image noise from (seed, global image index): image i is the same for any batch size / GPU count
  --class-cond C       class id for every image (-1: image index mod num_classes) for class-conditional configs
  --random-weights     no checkpoint: synthetic weights (K.synth), for smoke runs and benchmarking
  --no-png             skip PNG encoding (timing runs)
  --noise host|device  where --seed's per-image noise is drawn.  host (default: what --seed S has always produced, and the recipe of the
                       committed fixtures): one CPU torch.Generator per global image index (K.synth.synth_noise), drawn by worker threads
                       AHEAD of the sampler and handed over as an asynchronous pinned copy on a side stream, so the draw runs beside the
                       previous batch's GPU pass (job rate = loop rate, bench.py --job); device: the keyed Philox generator of the HIP
                       library (kd_randn_f32; seed, global index, draw number -> values), no host work at all -- also for the ancestral
                       samplers' per-step noise; other images than host, reproducible to ~1e-5 across GPU generations (hardware log2 / cos)
  --gather-uint8       8-bit conversion on the GPU before the all-gather of finished images (same PNG bytes, 4x less xGMI traffic).
                       The DEFAULT whenever there is a gather (more than one process) and PNG files are written -- the writer needs
                       nothing else; --gather-fp32 keeps the reference's fp32 gather (main() then returns fp32 images)
With --seed the stochastic samplers are index-addressed too: one Brownian tree per global image index for the SDE samplers,
a per-(index, call) stream for the ancestral ones (the reference draws both from rank-local global RNG state).
"""
import argparse
import sys
import time
from pathlib import Path

import torch
from tqdm import tqdm

import k_diffusion_amd as K


LAST_RUN = {}        # statistics of the most recent main() (see run())


def parse(argv=None):
    p = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    p.add_argument('--batch-size', type=int, default=64, help='the batch size')
    p.add_argument('--checkpoint', type=Path, help='the checkpoint to use (safetensors)')
    p.add_argument('--config', type=Path, help='the model config')
    p.add_argument('-n', type=int, default=64, help='the number of images to sample')
    p.add_argument('--prefix', type=str, default='out', help='the output prefix')
    p.add_argument('--steps', type=int, default=50, help='the number of denoising steps')
    p.add_argument('--sampler', type=str, default='lms', help='K.sampling.sample_<name>')
    p.add_argument('--seed', type=int, default=None, help='seed for the per-image initial noise')
    p.add_argument('--class-cond', type=int, default=None, help='class id for all images; -1 = index mod num_classes')
    p.add_argument('--random-weights', action='store_true', help='synthetic weights instead of a checkpoint')
    p.add_argument('--no-png', action='store_true', help='do not write PNG files')
    p.add_argument('--noise', choices=['device', 'host'], default='host',
                   help="where --seed's per-image noise is drawn: per-image CPU generators ahead of the sampler (default: the recipe of every earlier "
                        "round and of the committed fixtures) or the library's keyed device generator")
    p.add_argument('--gather-uint8', action='store_true',
                   help='convert finished images to uint8 on the GPU before the all-gather (what the PNG writer needs; 4x less xGMI traffic)')
    p.add_argument('--gather-fp32', action='store_true', help='all-gather the finished images as fp32 even when only PNG files are wanted')
    args = p.parse_args(argv)
    if args.gather_uint8 and args.gather_fp32:
        p.error('--gather-uint8 and --gather-fp32 exclude each other')
    if args.checkpoint is None and not args.random_weights:
        p.error('--checkpoint is required (or pass --random-weights)')
    if args.checkpoint is None and args.config is None:
        p.error('--random-weights needs --config')
    return args


def resolve_sampler(name):
    fn = getattr(K.sampling, name if name.startswith('sample_') else 'sample_' + name, None)
    if fn is None:
        raise SystemExit(f'unknown sampler {name!r}; available: ' +
                         ', '.join(sorted(n[7:] for n in dir(K.sampling) if n.startswith('sample_'))))
    return fn


def class_ids(args, num_classes, indices, device):
    """class_cond for the given global image indices, or None for unconditional models."""
    if not num_classes:
        return None
    if args.class_cond is None:
        raise SystemExit(f'this config is class-conditional ({num_classes} classes): pass --class-cond C (or -1)')
    if args.class_cond >= num_classes:
        raise SystemExit(f'--class-cond {args.class_cond} is out of range: this config has {num_classes} classes (0..{num_classes - 1})')
    if args.class_cond < 0:
        return to_device_async(indices % num_classes, device)
    return torch.full([len(indices)], args.class_cond, dtype=torch.int64, device=device)


def brownian_seeds(seed, indices):
    """One Brownian-tree seed per GLOBAL image index (sampling.BrownianTreeNoiseSampler takes a list: one tree per batch item),
    so an SDE sampler's noise for image i does not depend on the batch it is drawn in or on the number of GPUs."""
    return [((int(seed) * 0x9E3779B97F4A7C15) ^ (int(g) * 0xD1B54A32D192ED03 + 0x2545F4914F6CDD1D)) & 0x7FFFFFFFFFFFFFFF for g in indices]


def noise_seeds(seed, indices):
    """One key of the device generator (kd_randn_f32) per GLOBAL image index; a different mix from ``brownian_seeds``, so an SDE run's
    start noise and its Brownian trees are unrelated streams."""
    return [((int(seed) * 0xD6E8FEB86659FD93) ^ (int(g) * 0x9E3779B97F4A7C15 + 0x632BE59BD9B4E019)) & 0x7FFFFFFFFFFFFFFF for g in indices]


def to_device_async(t, device):
    """A small host tensor (class ids, seeds) onto the device without the stream synchronisation a pageable ``.to(device)`` ends with --
    that wait would hold the host until the previous batch's GPU pass has drained."""
    device = torch.device(device)
    if device.type != 'cuda':
        return t.to(device)
    return t.pin_memory().to(device, non_blocking=True)


def device_noise(seed, indices, shape, device, draw=0, scale=1.0):
    """[len(indices), *shape] normals * scale drawn on the device: a function of (seed, global index, draw number) only."""
    keys = to_device_async(torch.tensor(noise_seeds(seed, indices), dtype=torch.int64), device)
    return K.ops.randn_indexed(torch.empty((len(indices), *shape), device=device, dtype=torch.float32), keys, draw=draw, scale=scale)


def indexed_noise_sampler(seed, indices, shape, device, where='host'):   # (the CLI passes --noise; the default here is the fixtures' recipe)
    """noise_sampler(sigma, sigma_next) for the ancestral samplers (default: randn_like on the global stream, sampling.py:73-75)
    whose draw for image i is a function of (seed, i, call number) only.  ``where``: 'host' = one CPU generator per (image, call)
    (the recipe of the round-1 fixtures; slow -- every call draws on the host), 'device' = kd_randn_f32 with the call number as
    its draw counter (draw 0 is the start noise)."""
    calls = [0]
    keys = None

    def noise_sampler(sigma, sigma_next):
        nonlocal keys
        k = calls[0]
        calls[0] += 1
        if where == 'device':
            if keys is None:
                keys = to_device_async(torch.tensor(noise_seeds(seed, indices), dtype=torch.int64), device)
            return K.ops.randn_indexed(torch.empty((len(indices), *shape), device=device, dtype=torch.float32), keys, draw=k + 1)
        return torch.stack([K.synth.synth_noise(shape, int(seed) + 7919 * (k + 1), int(g), 1.0) for g in indices]).to(device)
    return noise_sampler


def seeded_noise_args(sampler, seed, indices, x, sigma_min, sigma_max, where='host'):
    """extra keyword arguments that make a stochastic sampler's noise index-addressed when --seed is given."""
    import inspect
    if seed is None or 'noise_sampler' not in inspect.signature(sampler).parameters:
        return {}
    if sampler.__name__ in ('sample_dpmpp_sde', 'sample_dpmpp_2m_sde', 'sample_dpmpp_3m_sde'):      # default: Brownian tree (sampling.py:548,615,661)
        return {'noise_sampler': K.sampling.BrownianTreeNoiseSampler(x, sigma_min, sigma_max, seed=brownian_seeds(seed, indices))}
    return {'noise_sampler': indexed_noise_sampler(seed, indices, tuple(x.shape[1:]), x.device, where)}


def main(argv=None):
    args = parse(argv)
    config = K.config.load_config(args.config if args.config else args.checkpoint)
    model_config = config['model']
    # TODO (as in the reference): non-square input sizes
    assert len(model_config['input_size']) == 2 and model_config['input_size'][0] == model_config['input_size'][1]
    size = model_config['input_size']

    accelerator = K.distributed.RankContext()
    device = accelerator.device
    print('Using device:', device, flush=True)
    if device.type != 'cuda':
        raise SystemExit('sample.py runs the HIP hot path: no ROCm device is visible (there is no CPU fallback)')

    inner_model = K.config.make_model(config).eval().requires_grad_(False)
    if args.random_weights:
        inner_model.load_state_dict(K.synth.synth_state_dict(inner_model.state_dict(), seed=args.seed or 0))
    else:
        inner_model.load_state_dict(K.checkpoint.load_inference_checkpoint(args.checkpoint))
    inner_model = inner_model.to(device)
    accelerator.print('Parameters:', K.utils.n_params(inner_model))
    model = K.Denoiser(inner_model, sigma_data=model_config['sigma_data'])

    sigma_min, sigma_max = model_config['sigma_min'], model_config['sigma_max']
    sampler = resolve_sampler(args.sampler)
    num_classes = config['dataset']['num_classes']
    shape = (model_config['input_channels'], size[0], size[1])

    @torch.no_grad()
    @K.utils.eval_mode(model)
    def run():
        if accelerator.is_local_main_process:
            tqdm.write('Sampling...')
        sigmas = K.sampling.get_sigmas_karras(args.steps, sigma_min, sigma_max, rho=7., device=device)
        host_noise = K.synth.NoisePrefetcher(shape, args.seed, sigma_max, device, width=min(args.batch_size, args.n)) if (args.seed is not None and args.noise == 'host') else None
        rounds_done, batch_marks = [0], []

        def sample_fn(indices):
            """Images of the given GLOBAL indices (this rank's share of one round; may be empty)."""
            n = len(indices)
            k = rounds_done[0]
            rounds_done[0] += 1
            batch_marks.append(time.perf_counter())
            if host_noise is not None:
                x = host_noise.take(k)                            # drawn ahead by worker threads, asynchronous pinned copy
            if n == 0:
                return torch.empty([0, *shape], device=device)
            if args.seed is None:
                x = torch.randn([n, *shape], device=device) * sigma_max
            elif host_noise is None:
                x = device_noise(args.seed, indices, shape, device, draw=0, scale=sigma_max)
            extra = {}
            cc = class_ids(args, num_classes, indices, device)
            if cc is not None:
                extra['class_cond'] = cc
            quiet = not accelerator.is_local_main_process
            if sampler is K.sampling.sample_dpm_fast:           # these two take the sigma range, not a schedule
                return sampler(model, x, sigma_min, sigma_max, args.steps, extra_args=extra, disable=quiet)
            if sampler is K.sampling.sample_dpm_adaptive:
                return sampler(model, x, sigma_min, sigma_max, extra_args=extra, disable=quiet)
            noise = seeded_noise_args(sampler, args.seed, indices, x, sigma_min, sigma_max, args.noise)
            return sampler(model, x, sigmas, extra_args=extra, disable=quiet, **noise)

        t0 = time.perf_counter()
        # the reference's compute_features (evaluation.py:80-90) with images addressed by global index: out[i] is image i for
        # any batch size / GPU count (see compute_features_indexed)
        # [-1, 1] fp32 -> uint8 on the device before the gather: 4x fewer bytes over xGMI, the same PNG bytes
        as_u8 = args.gather_uint8 or (accelerator.num_processes > 1 and not args.no_png and not args.gather_fp32)
        post = K.ops.to_uint8 if as_u8 else None
        try:
            x_0 = K.evaluation.compute_features_indexed(accelerator, sample_fn, args.n, args.batch_size, post=post,
                                                        on_schedule=host_noise.schedule if host_noise is not None else None)
            torch.cuda.synchronize()
        finally:
            if host_noise is not None:
                host_noise.close()
        seconds = time.perf_counter() - t0
        accelerator.print(f'{args.n} images in {seconds:.2f} s')
        # what a caller that times the job (bench.py's `job` block) reads: the sampling region above -- noise draw -> finished
        # (gathered) images on the device, PNG encoding excluded -- and when each round was handed to the sampler
        LAST_RUN.clear()
        LAST_RUN.update({'n': args.n, 'seconds': seconds, 'rounds': len(batch_marks), 'round_starts': [m - t0 for m in batch_marks],
                         'world': accelerator.num_processes, 'noise': args.noise if args.seed is not None else 'device (unseeded torch.randn)'})
        if accelerator.is_main_process and not args.no_png:
            for i, out in enumerate(x_0):
                K.utils.to_pil_image(out).save(f'{args.prefix}_{i:05}.png')
        return x_0

    try:
        return run()
    except KeyboardInterrupt:
        pass
    finally:
        accelerator.shutdown()


if __name__ == '__main__':
    main()
    sys.exit(0)
