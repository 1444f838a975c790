#!/usr/bin/env python3
"""Benchmark of the k-diffusion sampling hot path on MI355X (contract: see the task statement).

    python bench.py --gpus 1 --steps 3 --warmup 1
    python bench.py --gpus N --steps K --warmup W          (starts its own ranks: re-executes itself under the launcher below)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over one batch: a complete 50-step ``sample_dpmpp_2m`` run
(50 denoiser evaluations of the 256x256 image_transformer_v2 + the fused solver steps) for
``--batch`` images per GPU, followed -- for N > 1 -- by the path's one exchange step, the RCCL
all-gather of the finished images (k_diffusion/evaluation.py:87; uint8 as the CLI gathers them, ``--gather fp32`` for the reference's).
Initial noise, weights and the sigma table are resident in HBM before the timed region.

OUTPUT (round 6).  Rank 0 prints ONE compact JSON line (< 4 KiB, the LAST line of stdout; nothing else goes to stdout and stderr stays
quiet) with the contract's keys -- metric / value / unit / n_gpus / steps / warmup / ms_per_step / dtype / config / roofline /
cpu_baseline -- plus ``parity`` (distance of this build on this box to the reference's 50-step batch-32 golden of this workload),
``mode_values`` and ``detail_file``.  Everything longer (per-mode entries with their per-family rooflines, and the opt-in blocks) goes to
``bench_detail.json`` (``--detail-file``).

Arithmetic mode (``--mode``, default split3): ``value`` / ``dtype`` / ``roofline`` belong to that mode.  The default is the
fp32-PARITY mode -- the reference samples in fp32 (sample.py:39-47, no mixed precision) and north_star asks for images within
1e-3 of it, which only the fp32 modes meet.  At N = 1 the bf16 and fp8 modes are measured right after it on the same box with the SAME
--steps / --warmup and their own HIP-event pass (``mode_values``; full entries in the detail file):
  split3  fp32 activations, 3 split-bf16 MFMA terms per product: the fp32-parity mode (< 5e-4 from the fp32 reference end to end)
  bf16    bf16 activations, one bf16 MFMA per product, fp32 accumulate / statistics (the reference under autocast(bfloat16);
          1.1e-2 from the fp32 reference after 50 steps: NOT parity-grade, reported for what it is)
  fp8     the bf16 mode with the norm -> qkv / norm -> GEGLU projections of the width-256 / 512 levels as e4m3 x e4m3 products on the block-scaled
          fp8 matrix instruction (BASELINE configs[4]'s arithmetic; NOT parity-grade)
  exact   fp32 activations, fp32-input MFMA, bit-for-bit an fmaf chain (``--detail`` or ``--modes split3,bf16,exact``)
Opt-in blocks (``--detail`` = all of them; they lengthen the run from ~1 to ~3 minutes and only ever write to the detail file):
``--other-configs`` (all five BASELINE configurations with their parity), ``--small-batch`` (batch 1 / 4 latency), ``--job`` (the
sample.py CLI job timed end to end), ``--power`` (benchmarks/power_report.py).
"""
import argparse
import json
import os
import sys
import time

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: what RCCL needs on this driver (must precede HIP init)

import torch  # noqa: E402

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

import k_diffusion_amd as K  # noqa: E402

# /opt/skills/guides/MI355X_MICROARCH.md (chip-level parameters)
FP8_MFMA_PEAK_TFLOPS = 5000.0      # v_mfma_scale_f32_32x32x64_f8f6f4 dense peak (KDIFF_GEMM=fp8: the gemm_mx8 kernels)
FP32_MFMA_PEAK_TFLOPS = 157.3      # v_mfma_f32_32x32x2_f32 dense peak (KDIFF_GEMM=exact)
BF16_MFMA_PEAK_TFLOPS = 2500.0     # v_mfma_f32_32x32x16_bf16 dense peak (split3 mode executes 3 bf16 products per fp32 product)
BF16_MFMA_SUSTAINED_TFLOPS = 2000.0  # measured: a pure MFMA kernel on all 256 CUs (profiles/r02_pipe_overlap.md); informational only
HBM_PEAK_GBS = 8000.0              # HBM3E spec (6.3 TB/s is the measured streaming ceiling)


def parse(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=3, help="timed passes (each = one full sampling run of a batch)")
    p.add_argument("--warmup", type=int, default=1)
    p.add_argument("--config", default="configs/config_oxford_flowers.json")
    p.add_argument("--batch", type=int, default=32, help="images per GPU per pass")
    p.add_argument("--sampler", default="sample_dpmpp_2m")
    p.add_argument("--sampler-steps", type=int, default=50)
    p.add_argument("--seed", type=int, default=0)
    p.add_argument("--mode", default=os.environ.get("KDIFF_GEMM", "split3"), choices=["bf16", "split3", "exact", "fp8"], help="arithmetic mode of `value`")
    p.add_argument("--modes", default=None, help="modes measured at N = 1 (same steps / warm-up, own roofline); default: split3,bf16,fp8 (+ exact with --detail)")
    p.add_argument("--no-other-modes", action="store_true", help="measure --mode only")
    p.add_argument("--gather", default="uint8", choices=["uint8", "fp32"],
                   help="what the N > 1 exchange step moves: uint8 images (what sample.py gathers when it writes PNGs) or the reference's fp32")
    p.add_argument("--detail-file", default=os.path.join(REPO, "bench_detail.json"), help="where the long form of the result goes")
    p.add_argument("--detail", action="store_true", help="all opt-in blocks: exact mode, --other-configs, --small-batch, --job")
    p.add_argument("--other-configs", action="store_true", help="all five BASELINE configurations, each with its parity (detail file)")
    p.add_argument("--other-passes", type=int, default=5, help="timed passes of each `other_configs` entry (1 warm-up)")
    p.add_argument("--small-batch", action="store_true", help="batch-1 / batch-4 latency entries (detail file)")
    p.add_argument("--job", action="store_true", help="the sample.py CLI job timed end to end: noise draw -> finished images (detail file)")
    p.add_argument("--job-images", type=int, default=512, help="images of the timed CLI job (`job`)")
    p.add_argument("--job-repeats", type=int, default=3, help="repeats of each timed CLI job (the median counts)")
    p.add_argument("--power", action="store_true", help="socket power / shader clock / throttle accumulators over extra untimed passes (detail file)")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--cpu-seconds", type=float, default=20.0, help="target CPU time for the baseline sample")
    p.add_argument("--cpu-batch", type=int, default=2, help="images of the CPU baseline's sample (BASELINE.md section 4: 2 - 4)")
    p.add_argument("--no-kernel-events", action="store_true", help="do not record per-launch HIP events in the extra pass behind the timed region")
    p.add_argument("--kernel-table", default=None, help="write the per-kernel event timings to this JSON file")
    p.add_argument("--no-parity", action="store_true", help="skip the golden-vector parity block (the 50-step batch-32 golden of this workload per measured mode)")
    p.add_argument("--backend", default=None, choices=["nccl", "gloo"], help="process-group backend (default: nccl = RCCL on a GPU box; gloo is for the "
                                                                              "launcher's CPU test together with --stub-workload)")
    p.add_argument("--force-collectives", action="store_true",
                   help="with --gpus 1: create the one-rank RCCL group and run the N-GPU pass (sampling + uint8 conversion + all-gather) anyway -- "
                        "how a 1-GPU box exercises the calls of the 8-GPU job (tests/test_model_gpu.py)")
    p.add_argument("--stub-workload", action="store_true",
                   help="launcher / rank-plumbing test: a tiny CPU tensor op per pass instead of the sampler (no GPU, no kernels; the line says so and is no measurement)")
    for gone in ("--no-other-configs", "--no-power", "--no-small-batch", "--no-job"):     # rounds 2 - 5: these blocks were on by default; the command
        p.add_argument(gone, action="store_true", help=argparse.SUPPRESS)                 # lists of benchmarks/profile_round*.sh still name the flags
    args = p.parse_args(argv)
    if args.detail:
        args.other_configs = args.small_batch = args.job = True
    if args.modes is None:
        args.modes = "split3,bf16,fp8,exact" if args.detail else "split3,bf16,fp8"
    return args


def build_model(cfg, device, seed):
    model = K.config.make_model(cfg).eval().requires_grad_(False)
    model.load_state_dict(K.synth.synth_state_dict(model.state_dict(), seed=seed))
    return model.to(device)


def kernel_table():
    """Per-launch HIP-event timings recorded by the library -> grouped per kernel."""
    import ctypes as C
    lib = K._native.lib()
    groups = {}
    name = C.create_string_buffer(128)
    ms, fl, by = C.c_float(), C.c_double(), C.c_double()
    for i in range(lib.kd_prof_count()):
        K._native.check(lib.kd_prof_get(i, name, 128, C.byref(ms), C.byref(fl), C.byref(by)), "kd_prof_get")
        g = groups.setdefault(name.value.decode(), {"launches": 0, "ms": 0.0, "flops": 0.0, "bytes": 0.0})
        g["launches"] += 1
        g["ms"] += ms.value
        g["flops"] += fl.value
        g["bytes"] += by.value
    lib.kd_prof_reset()
    return groups


SIDE_STREAM = ("gemm_skinny", "fourier", "cond_sum", "rmsnorm")     # the conditioning chain: once per sigma table, ahead of the loop


def family_roofline(name, g, mode, total_ms):
    """Roofline entry of one kernel family from its summed ALGORITHMIC work (flops = 2 M N K of the products, bytes = every
    operand once at its storage width) and HIP-event time.  `bound` is decided on the algorithmic work; `frac` is the useful
    fraction (algorithmic / peak).  The split-bf16x3 kernels execute 3 bf16 MFMA products per algorithmic product: their
    `executed_frac` (matrix-pipe occupancy) is reported beside it."""
    sec = g["ms"] * 1e-3
    gbs = g["bytes"] / sec / 1e9
    is_bf16 = "bf16" in name and "bf16x3" not in name
    is_x3 = "bf16x3" in name or "_x3" in name or name in ("gemm_astat", "attn_na2d") or (mode == "split3" and name.startswith("gemm_bf16x3"))
    has_mfma = name.startswith("gemm") or name.startswith("attn")
    if name.startswith("gemm_mx8"):
        mult, peak = 1.0, FP8_MFMA_PEAK_TFLOPS
    elif is_bf16 or is_x3:
        mult, peak = (3.0 if is_x3 else 1.0), BF16_MFMA_PEAK_TFLOPS
    elif has_mfma and not name.startswith("gemm_skinny"):
        mult, peak = 1.0, FP32_MFMA_PEAK_TFLOPS          # exact fp32-input MFMA kernels
    else:
        mult, peak = 0.0, BF16_MFMA_PEAK_TFLOPS
    tfl_alg = g["flops"] / sec / 1e12
    # time the algorithmic work needs at the roof of each resource (matrix flops priced at the instruction the kernel issues)
    t_hbm, t_mfma = g["bytes"] / (HBM_PEAK_GBS * 1e9), (g["flops"] * mult / (peak * 1e12) if mult else 0.0)
    common = {"kernel": name, "launches": g["launches"], "avg_launch_ms": round(g["ms"] / max(g["launches"], 1), 5),
              "algorithmic_bytes_per_launch": int(g["bytes"] / max(g["launches"], 1)), "algorithmic_flops_per_launch": int(g["flops"] / max(g["launches"], 1)),
              "share_of_kernel_time": None if name.startswith(SIDE_STREAM) or not total_ms else round(g["ms"] / total_ms, 4),
              "algorithmic_gbs": round(gbs, 1), "algorithmic_tflops": round(tfl_alg, 2)}
    if name.startswith(SIDE_STREAM):
        common["overlapped"] = ("conditioning chain: one launch set per sigma table of the run (steps x batch rows), ahead of the "
                                "solver loop; not part of a forward's main chain")
    if mult:
        common["mfma_useful_frac"] = round(tfl_alg / peak, 4)
        common["mfma_executed_frac"] = round(tfl_alg * mult / peak, 4)
        if peak == BF16_MFMA_PEAK_TFLOPS:
            # what a register-only bf16 MFMA loop sustains chip-wide on these boxes (power limited: 1.9-2.0 GHz, ~1300 W):
            # benchmarks/probe/power_probe.cpp, profiles/r02_pipe_overlap.md
            common["mfma_executed_frac_of_sustained_2000TF"] = round(tfl_alg * mult / BF16_MFMA_SUSTAINED_TFLOPS, 4)
    if t_hbm >= t_mfma:
        return {"bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4), **common,
                "note": "algorithmic bytes (inputs once + outputs once at their storage width, DESIGN.md section 4) of every launch of this "
                        "family in the profiled pass / sum of their HIP-event durations"}
    if mult > 1:
        # fp32-parity kernels: one fp32 product = 3 bf16 MFMA terms, so the roof of THIS arithmetic is the bf16 dense peak / 3 (833 TFLOP/s
        # of fp32-grade products; the fp32-input MFMA the dtype would otherwise use peaks at 157.3 TFLOP/s).  `frac` = achieved / that roof
        # = the fraction of the matrix pipe's cycles the kernel keeps busy (mfma_executed_frac); against the raw bf16 peak: mfma_useful_frac
        roof = peak / mult
        return {"bound": "mfma", "achieved": round(tfl_alg, 2), "peak": round(roof, 1), "unit": "TFLOP/s", "frac": round(tfl_alg / roof, 4), **common,
                "frac_of_f32_mfma_peak": round(tfl_alg / FP32_MFMA_PEAK_TFLOPS, 3),
                "note": f"algorithmic matrix flops 2*M*N*K (fp32-grade products) / sum of HIP-event durations; peak = {peak:g} TFLOP/s dense bf16 MFMA / "
                        f"{mult:g} MFMA terms per product (split-bf16x3); fp32-input MFMA peak {FP32_MFMA_PEAK_TFLOPS} TFLOP/s for comparison"}
    return {"bound": "mfma", "achieved": round(tfl_alg, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(tfl_alg / peak, 4), **common,
            "note": "algorithmic matrix flops 2*M*N*K / sum of HIP-event durations against the dense MFMA peak of the instruction issued"}


def pmc_traffic(kernel_family):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC summary (separate FETCH_SIZE / WRITE_SIZE
    passes, gfx950 FETCH x2 correction: profiles/summarize_pmc.py); None when the summary has no matching entry."""
    if not kernel_family:
        return None
    for name in ("r06_pmc_traffic.json", "r06_pmc_traffic_bf16.json", "r06_pmc_traffic_fp8.json", "r05_pmc_traffic.json", "r05_pmc_traffic_bf16.json", "r04_pmc_traffic.json", "r03_pmc_traffic.json", "r02_pmc_traffic.json"):        # newest committed summary that knows the kernel
        path = os.path.join(REPO, "profiles", name)
        try:
            table = json.load(open(path))
        except Exception:
            continue
        ent = table.get(kernel_family.split(" ")[0].split("<")[0])
        if ent and ent.get("hbm_bytes_per_launch"):
            return ent["hbm_bytes_per_launch"]
    return None


def reference_cpu_figure():
    """The REFERENCE's own wall time for this workload -- config_oxford_flowers.json, sample_dpmpp_2m x 50, batch 32, fp32 -- recorded while
    oracle/make_golden_r6.py generated the headline golden from the imported reference in the build container (the reference cannot travel
    to the GPU box): read from that file's metadata.  None if the file is missing."""
    from safetensors import safe_open
    from tests.golden import cases
    try:
        with safe_open(os.path.join(cases.GOLDEN_DIR, "samples_r6.safetensors"), "pt") as f:
            md = f.metadata() or {}
        return {"value": round(float(md["reference_images_per_s"]), 4), "unit": "images/sec", "cores": int(md["threads"]), "kind": "reference",
                "seconds": float(md["reference_seconds"]), "torch": md.get("torch"), "cpu": md.get("cpu"),
                "sample": "k_diffusion (the reference, imported) sample_dpmpp_2m x 50, THIS config, batch 32, fp32, build container "
                          "(tests/golden/samples_r6.safetensors metadata)"}
    except Exception:
        return None


def cpu_baseline(cfg, seed, sampler_steps, target_seconds, batch=2):
    """The CPU oracle (a port of the reference's algorithm: oracle/hdit.py + oracle/solvers.py) timed on
    this host's cores for a bounded sample of the same workload (same weights / noise recipe)."""
    from oracle import hdit, solvers
    na2d_recorded, hdit.na2d = hdit.na2d, hdit.na2d_shifted     # same op, one window offset at a time (no 49x gather): ~5x faster on the CPU
    try:
        return _cpu_baseline(hdit, solvers, cfg, seed, sampler_steps, target_seconds, batch)
    finally:
        hdit.na2d = na2d_recorded


CPU_THREAD_CAP = 32


def _cpu_baseline(hdit, solvers, cfg, seed, sampler_steps, target_seconds, batch):
    mc = cfg["model"]
    avail = os.cpu_count()
    cores = min(avail, CPU_THREAD_CAP)   # torch's intra-op pool stops scaling (and thrashes) far below 256 threads on these op sizes
    torch.set_num_threads(cores)
    model = K.config.make_model(cfg)
    sd = K.synth.synth_state_dict(model.state_dict(), seed=seed)
    den = solvers.denoiser(lambda x, s, **kw: hdit.forward(sd, mc, x, s, **kw), mc["sigma_data"])
    shape = (mc["input_channels"], *mc["input_size"])
    x = K.synth.synth_noise_batch(shape, seed, 0, batch, mc["sigma_max"])
    sig = solvers.sigmas_karras(sampler_steps, mc["sigma_min"], mc["sigma_max"])
    t0 = time.perf_counter()
    den(x, sig[:1].expand(batch))
    one = time.perf_counter() - t0
    n = max(2, min(sampler_steps, int(target_seconds / max(one, 1e-3))))
    t0 = time.perf_counter()
    with torch.no_grad():
        solvers.sample_dpmpp_2m(den, x, torch.cat([sig[:n], sig[-1:]]))
    dt = time.perf_counter() - t0
    per_batch = dt / n * sampler_steps
    capped = f" of {avail} available (capped: torch's intra-op pool gets slower beyond {CPU_THREAD_CAP} threads at these op sizes)" if avail > cores else ""
    return {"value": round(batch / per_batch, 5), "unit": "images/sec", "cores": cores, "cores_available": avail, "kind": "port",
            "sample": f"batch {batch}, {n} of {sampler_steps} DPM++2M steps of THIS workload with the oracle port (torch CPU fp32, {cores} threads{capped}), "
                      f"{dt:.1f} s measured, scaled to {sampler_steps} steps",
            "reference": reference_cpu_figure()}


PARITY_CASE = "smp32_flowers_na_2m50"    # tests/golden/cases.HEADLINE_CASE: THIS workload -- config, sampler, 50 steps, batch 32 -- recorded from the reference


def golden_parity(dev, modes, golden_file, case, cfgname, sampler, steps, batch, keep, autocast_file=None):
    """Distance of this build, on this box, to the REFERENCE's own images of one committed golden case (tests/golden/*.safetensors, recorded from
    the imported reference by oracle/make_golden*.py; the oracle is not involved): the same run (weights seed, noise, class ids, sigmas) goes
    through the HIP path in every mode; rel_err = max|got - ref| / max|ref|, north_star's measure.  The bf16 mode is also compared with the
    reference's own torch.autocast(bfloat16) run where one was recorded."""
    from safetensors.torch import load_file
    from tests.golden import cases
    ref = load_file(os.path.join(cases.GOLDEN_DIR, golden_file))[case]
    ref16 = None
    if autocast_file and os.path.exists(os.path.join(cases.GOLDEN_DIR, autocast_file)):
        t16 = load_file(os.path.join(cases.GOLDEN_DIR, autocast_file))       # (round-1 autocast file: same case names; later ones: case + "_bf16")
        ref16 = t16.get(case + "_bf16", t16.get(case) if autocast_file != golden_file else None)      # (absent: the bf16 entry stands alone)
    cfg = K.config.load_config(cases.raw_config(cfgname))
    mc = cfg["model"]
    model = K.config.make_model(cfg).eval().requires_grad_(False)
    model.load_state_dict(K.synth.synth_state_dict(model.state_dict(), seed=cases.WEIGHT_SEED))
    den = K.Denoiser(model.to(dev), sigma_data=mc["sigma_data"])
    x, cls = cases.sample_inputs(cfg, batch)
    x = x.to(dev)
    extra = {"class_cond": cls.to(dev)} if cls is not None else {}
    sigmas = K.sampling.get_sigmas_karras(steps, mc["sigma_min"], mc["sigma_max"], rho=7., device=dev)
    rel = lambda a, b: float((a - b).abs().max() / b.abs().max())
    out, saved = {}, os.environ.get("KDIFF_GEMM")
    try:
        for m in modes:
            os.environ["KDIFF_GEMM"] = m
            y = getattr(K.sampling, sampler)(den, x, sigmas, extra_args=extra, disable=True)
            y = (y if keep is None else y[keep]).float().cpu()
            gate = 1e-3 if m in ("split3", "exact") else None         # north_star's tolerance applies to the fp32-parity modes
            ent = {"rel_err_vs_reference_golden": round(rel(y, ref), 7), "gate": gate}
            if gate is not None:
                ent["pass"] = ent["rel_err_vs_reference_golden"] < gate
            elif ref16 is not None:
                ent["rel_err_vs_reference_autocast_bf16"] = round(rel(y, ref16), 7)
                ent["reference_autocast_bf16_vs_reference_fp32"] = round(rel(ref16, ref), 7)
            out[m] = ent
    finally:
        os.environ["KDIFF_GEMM"] = saved if saved is not None else "split3"
    what = CONFIG_OF.get(cfgname, cfgname)
    return {"case": f"{case}: {os.path.basename(what)}, {sampler} {steps} steps, batch {batch}, images {keep if keep is not None else 'all'} "
                    f"against the reference's fp32 run (tests/golden/{golden_file})",
            "measure": "max|got - ref| / max|ref|", **out}


def parity_vs_reference_golden(dev, modes):
    """The headline workload's parity case IS the headline workload: the reference's fp32 output of the 50-step DPM++2M run of
    config_oxford_flowers.json at batch 32 (samples_r6.safetensors, images cases.B32_KEEP, oracle/make_golden_r6.py; the bf16 mode also
    against the reference's autocast run of the same case where recorded)."""
    from tests.golden import cases
    case, cfgname, sampler, steps, batch = cases.HEADLINE_CASE
    assert case == PARITY_CASE
    return golden_parity(dev, modes, "samples_r6.safetensors", case, cfgname, sampler, steps, batch, cases.B32_KEEP, "samples_r6.safetensors")


def small_batch_latency(cfg, model, dev, args, measured_modes, batches=(1, 4)):
    """Latency regime of the same path (review item: batch 1 - 4): the headline sampler at small batches, 2 warm-up + 3 timed runs each,
    launch lists issued directly.  ms per forward = run time / model calls.  Not part of `value`."""
    mc = cfg["model"]
    den = K.Denoiser(model, sigma_data=mc["sigma_data"])
    sampler = getattr(K.sampling, args.sampler)
    sigmas = K.sampling.get_sigmas_karras(args.sampler_steps, mc["sigma_min"], mc["sigma_max"], rho=7., device=dev)
    shape = (mc["input_channels"], *mc["input_size"])
    nfe = {"sample_dpmpp_sde": 2 * args.sampler_steps - 1, "sample_heun": 2 * args.sampler_steps - 1}.get(args.sampler, args.sampler_steps)
    out, saved = {}, os.environ.get("KDIFF_GEMM")
    try:
        for m in measured_modes:
            os.environ["KDIFF_GEMM"] = m
            ent = {}
            for b in batches:
                x = K.synth.synth_noise_batch(shape, args.seed, 0, b, mc["sigma_max"]).to(dev)
                extra = {"class_cond": (torch.arange(b) % cfg["dataset"]["num_classes"]).to(dev)} if cfg["dataset"]["num_classes"] else {}
                for _ in range(2):
                    sampler(den, x, sigmas, extra_args=extra, disable=True)
                torch.cuda.synchronize()
                t = time.perf_counter()
                for _ in range(3):
                    sampler(den, x, sigmas, extra_args=extra, disable=True)
                torch.cuda.synchronize()
                dt = (time.perf_counter() - t) / 3
                ent[f"batch_{b}"] = {"ms_per_forward": round(dt * 1e3 / nfe, 4), "images_per_s": round(b / dt, 2)}
            out[m] = ent
    finally:
        os.environ["KDIFF_GEMM"] = saved if saved is not None else "split3"
    return {"workload": f"{os.path.basename(args.config)}, {args.sampler} {args.sampler_steps} steps, 3 timed runs per entry", **out}


def job_rate(args, modes, dev):
    """The job the CLI runs ("noise draw -> final image on device" over a full sampling run; /root/reference sample.py:52-66 +
    k_diffusion/evaluation.py:80-90), timed end to end: ``sample.main([... --random-weights --seed S -n N --batch-size B --steps 50 --sampler dpmpp_2m
    --no-png])`` builds its own model, draws every batch's noise, runs the sampler over N / B batches and assembles the N finished images on the
    device; ``seconds`` is sample.py's own "N images in ... s" region (the job's first batch builds its model's plans inside it) and images / seconds
    is the job rate, measured -- like `value` -- after one warm-up run of the same job (``first_job_seconds``: what the first job of a process pays).
    Both noise sources of --seed are timed: `device` (the CLI's default: kd_randn_f32, keyed by (seed, global index)) and `host` (per-image CPU
    generators drawn ahead by worker threads -- the recipe of the committed fixtures; its rate depends on what else the host's cores do).  `value` above times the sampler on a resident x0; `ratio_to_value`
    says how much of that the whole job keeps."""
    import contextlib
    import sample as cli
    quiet = open(os.devnull, "w")                                               # the CLI's own prints and progress bars go nowhere: the driver reads our tails
    os.environ["TQDM_DISABLE"] = "1"
    cfgpath = os.path.join(REPO, args.config) if not os.path.isabs(args.config) else args.config
    out, saved = {}, os.environ.get("KDIFF_GEMM")
    base = ["--config", cfgpath, "--random-weights", "--seed", str(args.seed), "--batch-size", str(args.batch), "--steps", str(args.sampler_steps),
            "--sampler", args.sampler.replace("sample_", ""), "--no-png"]
    try:
        for m in modes:
            os.environ["KDIFF_GEMM"] = m
            ent = {}
            with contextlib.redirect_stdout(quiet), contextlib.redirect_stderr(quiet):
                # warm-up, like the W passes in front of `value`: the same job once.  The FIRST job of a process in a mode is 3 - 5 % slower than
                # every later one whichever noise source it uses (first growth of the caching allocator by the 400 MB result + the batches'
                # workspaces, first plans of the mode): reported as first_job_seconds, not timed as the job rate
                cli.main(base + ["-n", str(args.job_images)])
                first = dict(cli.LAST_RUN)
            for noise in ("device", "host"):
                runs = []
                for _ in range(args.job_repeats):
                    with contextlib.redirect_stdout(quiet), contextlib.redirect_stderr(quiet):
                        cli.main(base + ["-n", str(args.job_images), "--noise", noise])
                    runs.append(dict(cli.LAST_RUN))
                secs = sorted(r["seconds"] for r in runs)
                med = secs[len(secs) // 2]
                st = runs[0]
                # consecutive identical jobs differ by +-3 % on these boxes (clock / allocator state): the MEDIAN of the repeats is the rate, all are shown
                ent[noise] = {"value": round(st["n"] / med, 3), "unit": "images/sec", "images": st["n"], "batches": st["rounds"],
                              "seconds": round(med, 4), "seconds_of_each_run": [round(r["seconds"], 4) for r in runs]}
            ent["first_job_seconds"] = round(first["seconds"], 4)
            out[m] = ent
    finally:
        os.environ["KDIFF_GEMM"] = saved if saved is not None else "split3"
    return {"workload": f"sample.py --config {os.path.basename(args.config)} --random-weights --seed {args.seed} -n {args.job_images} --batch-size {args.batch} "
                        f"--steps {args.sampler_steps} --sampler {args.sampler.replace('sample_', '')} --no-png [--noise device|host]",
            "timed_region": "sample.py's own 'N images in ... s': schedule, every batch's noise draw, sampler, assembly of the N finished fp32 images on "
                            "the device, final synchronize (model construction / weight upload happen before it); median of --job-repeats identical jobs after one warm-up job",
            "modes": out}


CONFIG_OF = {"flowers_na": "configs/config_oxford_flowers.json", "flowers_sw": "configs/config_oxford_flowers_shifted_window.json",
             "mnist": "configs/config_mnist_transformer.json", "cifar": "configs/config_cifar10_transformer.json"}


MODE_DTYPE = {
    "bf16": ("bf16", "bf16 activations in HBM (residual stream, qkv, attention out, FF hidden), one bf16 MFMA per product, fp32 accumulation, fp32 RMS "
                     "statistics / softmax / GELU / RoPE; fp32 image, solver state and conditioning chain -- the arithmetic of the reference under "
                     "torch.autocast(bfloat16); 1.1e-2 from the fp32 reference after 50 steps (gates 2e-2 per forward / 1.5e-2 end to end, "
                     "tests/test_model_gpu.py): narrower than the reference's own fp32 sampling path, NOT parity-grade"),
    "split3": ("f32", "fp32-parity mode: fp32 in HBM, fp32 accumulation everywhere; matrix products as 3 split-bf16 MFMA terms per fp32 product "
                      "(hi*hi + hi*lo + lo*hi, per-product error <= ~2^-15): < 5e-4 from the fp32 reference end to end, inside north_star's 1e-3"),
    "exact": ("f32", "exact fp32-input MFMA (bit-for-bit an fmaf chain)"),
    "fp8": ("fp8", "the bf16 mode with the AdaRMSNorm -> qkv / -> GEGLU projections of the width-256 / 512 levels as e4m3 x e4m3 products on the block-scaled "
                   "fp8 matrix instruction (weights: one power-of-two scale per output channel; activations: one per 32-k block), fp32 accumulation: "
                   "BASELINE configs[4]'s arithmetic, NOT parity-grade"),
}


def global_attention_block(groups, mode, cfg, batch, model_calls):
    """north_star's one named efficiency figure: the GLOBAL-ATTENTION level of the network (the last level of the hourglass: norm -> qkv ->
    attention -> out projection -> norm -> GEGLU up -> down, image_transformer_v2.py:370-396,487-493) as a fraction of the dense bf16 MFMA
    peak, from this pass's per-launch HIP events: every launch whose row count is that level's (batch x its tokens) except the token merge
    into it and the split out of it, plus the attention cores of that level.  flops = 2 M N K of the products + 4 T^2 d of the cores (the
    split-bf16x3 kernels EXECUTE three times that)."""
    mc = cfg["model"]
    n_lv = len(mc["depths"])
    if mc["self_attns"][n_lv - 1]["type"] != "global":
        return None
    ph, pw = mc["patch_size"]
    tokens = (mc["input_size"][0] // ph >> (n_lv - 1)) * (mc["input_size"][1] // pw >> (n_lv - 1))
    tag = f" M={batch * tokens} "
    picked = {}
    for name, g in groups.items():
        level_gemm = tag in name + " " and "<a1," not in name and ",e3>" not in name and "e3," not in name
        if level_gemm or name.startswith(("attn_global", "attn_block")):
            picked[name] = g
    if not picked:
        return None
    layers = mc["depths"][n_lv - 1] * model_calls
    ms = sum(g["ms"] for g in picked.values())
    fl = sum(g["flops"] for g in picked.values())
    tf = fl / (ms * 1e-3) / 1e12
    return {"level_width": mc["widths"][n_lv - 1], "tokens_per_sample": tokens, "layers_timed": layers,
            "launches_per_layer": round(sum(g["launches"] for g in picked.values()) / layers, 2),
            "us_per_layer": round(ms * 1e3 / layers, 2), "gflop_per_layer": round(fl / layers / 1e9, 2), "algorithmic_tflops": round(tf, 1),
            "frac_of_bf16_mfma_peak": round(tf / BF16_MFMA_PEAK_TFLOPS, 4),
            "executed_frac_of_bf16_mfma_peak": round(tf * (3.0 if mode == "split3" else 1.0) / BF16_MFMA_PEAK_TFLOPS, 4) if mode != "exact" else None,
            "kernels_us": {n: round(g["ms"] * 1e3 / max(g["launches"], 1), 2) for n, g in sorted(picked.items(), key=lambda kv: -kv[1]["ms"])}}


def roofline_of(groups, mode, pass_seconds):
    """`roofline` object of one arithmetic mode from its own HIP-event pass (kernel_table())."""
    fam = {}
    for name, g in groups.items():
        f = fam.setdefault(name.split(" ")[0].split("<")[0], {"launches": 0, "ms": 0.0, "flops": 0.0, "bytes": 0.0})
        for k in f:
            f[k] += g[k]
    if not fam:
        fam = {"(no kernel events recorded)": {"launches": 0, "ms": 0.0, "flops": 0.0, "bytes": 0.0}}
    main_ms = sum(g["ms"] for n, g in fam.items() if not n.startswith(SIDE_STREAM))
    rooflines = {name: family_roofline(name, g, mode, main_ms) for name, g in fam.items() if g["ms"] > 0}
    main_fams = [n for n in rooflines if not n.startswith(SIDE_STREAM)]
    dom_name = max(main_fams, key=lambda n: fam[n]["ms"]) if main_fams else None
    roofline = dict(rooflines[dom_name]) if dom_name else {"bound": "hbm", "achieved": 0.0, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": 0.0}
    roofline["traffic"] = pmc_traffic(dom_name)
    roofline["measured_on"] = "one extra identical pass right after this mode's timed region, a HIP event pair on the launch stream around every launch"
    keys = ("bound", "achieved", "unit", "frac", "share_of_kernel_time", "avg_launch_ms", "mfma_useful_frac", "mfma_executed_frac", "overlapped")
    roofline["other_kernels"] = {n: {k: r[k] for k in keys if k in r}
                                 for n, r in sorted(rooflines.items(), key=lambda kv: -fam[kv[0]]["ms"]) if n != dom_name}
    # whole path against its own rooflines: the algorithmic bytes / flops of every main-chain launch of the profiled pass
    tot_b = sum(g["bytes"] for n, g in fam.items() if not n.startswith(SIDE_STREAM))
    tot_f = sum(g["flops"] for n, g in fam.items() if not n.startswith(SIDE_STREAM))
    roofline["whole_path"] = {"algorithmic_gb_per_pass": round(tot_b / 1e9, 2), "algorithmic_tflop_per_pass": round(tot_f / 1e12, 3),
                              "hbm_frac_of_8TBs": round(tot_b / pass_seconds / (HBM_PEAK_GBS * 1e9), 4),
                              "mfma_useful_frac_of_bf16_peak": round(tot_f / pass_seconds / (BF16_MFMA_PEAK_TFLOPS * 1e12), 4),
                              "kernel_time_share_of_pass": round(main_ms * 1e-3 / pass_seconds, 4)}
    return roofline, fam


class Timed:
    """Warm-up, a barrier-bracketed timed region of exactly `steps` passes (max over ranks), then -- on rank 0 -- one more
    identical pass with a HIP event pair around every launch for the per-kernel table."""

    def __init__(self, ctx, steps, warmup, events):
        self.ctx, self.steps, self.warmup, self.events = ctx, steps, warmup, events

    def run(self, one_pass, event_pass=None):
        ctx = self.ctx
        for _ in range(self.warmup):
            one_pass()
        ctx.wait_for_everyone()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(self.steps):
            out = one_pass()
        ctx.wait_for_everyone()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if ctx.collectives:
            t = torch.tensor([dt], device=ctx.device, dtype=torch.float64)
            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
            dt = t.item()
        groups = {}
        if ctx.is_main_process and self.events:
            # Per-kernel durations: kept out of the timed region because ~9k event records per pass cost ~5 % throughput.
            K._native.lib().kd_prof_reset()
            K._native.prof_enable(True)
            (event_pass or one_pass)()
            torch.cuda.synchronize()
            K._native.prof_enable(False)
            groups = kernel_table()
        assert torch.isfinite(out.float()).all()
        return dt, out, groups


def mode_entry(mode, dt, steps, warmup, n_img, groups, n_gpus):
    ent = {"value": round(n_img / dt, 3), "unit": "images/sec", "steps": steps, "warmup": warmup, "ms_per_step": round(dt / steps * 1e3, 2),
           "dtype": MODE_DTYPE[mode][0], "dtype_note": MODE_DTYPE[mode][1], "parity_grade": mode in ("split3", "exact")}
    fam = None
    if groups:
        ent["roofline"], fam = roofline_of(groups, mode, dt / steps)
    return ent, fam


def other_config(name, path, dev, args, sampler_name, mode, fp8=False, brownian=False, batch=None, steps=None, passes=None):
    """A secondary BASELINE configuration on the same box: 1 warm-up + --other-passes timed passes + one event pass.  Class-conditional
    configs get class id = image index mod num_classes."""
    os.environ["KDIFF_GEMM"] = mode
    cfg = K.config.load_config(os.path.join(REPO, path))
    mc = cfg["model"]
    model = K.config.make_model(cfg).eval().requires_grad_(False)
    sd = K.synth.synth_state_dict(model.state_dict(), seed=args.seed)
    if fp8:       # what a `convert_for_inference.py --dtype fp8` checkpoint loads to (e4m3 + power-of-two channel scales: exact in bf16)
        sd = K.checkpoint.fp8_state_dict(sd)
    model.load_state_dict(sd)
    model = model.to(dev)
    den = K.Denoiser(model, sigma_data=mc["sigma_data"])
    shape = (mc["input_channels"], *mc["input_size"])
    B = batch or args.batch
    n_steps = steps or args.sampler_steps
    n_passes = passes or args.other_passes
    x0 = K.synth.synth_noise_batch(shape, args.seed, 0, B, mc["sigma_max"]).to(dev)
    sigmas = K.sampling.get_sigmas_karras(n_steps, mc["sigma_min"], mc["sigma_max"], rho=7., device=dev)
    sampler = getattr(K.sampling, sampler_name)
    nc = cfg["dataset"]["num_classes"]
    extra = {"class_cond": (torch.arange(B) % nc).to(dev)} if nc else {}
    import sample as cli      # the CLI's per-image Brownian seeds: image i's noise path is a function of (seed, global index i)

    def one_pass():
        kw = {"extra_args": extra} if extra else {}
        if brownian:
            kw["noise_sampler"] = K.sampling.BrownianTreeNoiseSampler(x0, mc["sigma_min"], mc["sigma_max"], seed=cli.brownian_seeds(args.seed, range(B)))
        return sampler(den, x0, sigmas, disable=True, **kw)
    ctx1 = K.distributed.RankContext.__new__(K.distributed.RankContext)
    ctx1.num_processes, ctx1.process_index, ctx1.local_process_index, ctx1.device, ctx1._owns_group = 1, 0, 0, dev, False
    dt, out, groups = Timed(ctx1, n_passes, 1, not args.no_kernel_events).run(one_pass)
    nfe = {"sample_dpmpp_2m": n_steps, "sample_euler": n_steps, "sample_dpmpp_sde": 2 * n_steps - 1, "sample_heun": 2 * n_steps - 1}.get(sampler_name)
    ent = {"value": round(n_passes * B / dt, 3), "unit": "images/sec", "steps": n_passes, "warmup": 1, "batch": B, "mode": mode,
           "ms_per_step": round(dt / n_passes * 1e3, 2), "model_calls_per_pass": nfe, "ms_per_model_call": round(dt / n_passes * 1e3 / nfe, 4) if nfe else None,
           "workload": f"{os.path.basename(path)} {mc['input_size'][0]}x{mc['input_size'][1]}, {sampler_name} {n_steps} steps"
                       + (", BrownianTreeNoiseSampler (one tree per image)" if brownian else "") + (", fp8-stored weights (e4m3, exact in bf16)" if fp8 else "")}
    if groups:
        tot = sum(g["ms"] for n, g in groups.items() if not n.startswith(SIDE_STREAM))
        share = lambda pre: round(sum(g["ms"] for n, g in groups.items() if n.startswith(pre)) / tot, 4) if tot else None
        ent["kernel_time_share"] = {"brownian": share("brownian"), "sampler_step": share("sampler_step"), "denoiser": round(1.0 - (share("brownian") or 0) - (share("sampler_step") or 0), 4)}
    os.environ["KDIFF_GEMM"] = args.mode
    return ent


def run_guarded(result, key, fn, mode):
    """An optional block must never cost the line itself (the contract's keys are complete before the first of them runs): its result goes
    under ``key``; a failure -- an exception, or the SystemExit sample.main raises on a refused argument / missing device -- is recorded there
    instead.  The arithmetic mode of the process is put back to ``mode`` either way."""
    try:
        out = fn()
        if out is not None:
            result[key] = out
    except (Exception, SystemExit) as e:
        result[key] = {"error": f"{type(e).__name__}: {e}"[:600]}
    finally:
        os.environ["KDIFF_GEMM"] = mode


LINE_LIMIT = 3800        # bytes of the one stdout line: the driver reads a 2 000-character tail of the log and an 8 000-character tail of stdout


def _cut(text, n):
    text = str(text)
    return text if len(text) <= n else text[:n - 3] + "..."


def compact_line(result, detail_file=None):
    """The driver's line: the contract's keys and the few figures that make the number mean something, every string bounded.  The long form
    (`result` itself) goes to the detail file.  Never longer than LINE_LIMIT: optional parts are dropped in a fixed order if it ever is."""
    keep = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "mode", "data")
    line = {k: (_cut(result[k], 120) if isinstance(result[k], str) else result[k]) for k in keep if k in result}
    cfg = dict(result.get("config") or {})
    if isinstance(cfg.get("gather"), dict):
        cfg["gather"] = {k: v for k, v in cfg["gather"].items() if k != "note"}
    line["config"] = {k: (_cut(v, 160) if isinstance(v, str) else v) for k, v in cfg.items() if v is not None}
    roof = result.get("roofline") or {}
    line["roofline"] = {k: roof[k] for k in ("bound", "kernel", "launches", "avg_launch_ms", "achieved", "peak", "unit", "frac", "traffic",
                                             "algorithmic_bytes_per_launch", "algorithmic_flops_per_launch", "share_of_kernel_time",
                                             "mfma_useful_frac", "mfma_executed_frac") if k in roof}
    line["roofline"].setdefault("traffic", None)
    gab = {}
    per_mode = {m: e for m, e in (result.get("modes") or {}).items() if isinstance(e, dict)}
    per_mode[result.get("mode")] = {"roofline": roof}               # (the headline mode's entry under `modes` only points at the top-level roofline)
    for m, ent in per_mode.items():
        g = (ent.get("roofline") or {}).get("global_attention_block") if isinstance(ent.get("roofline"), dict) else None
        if g:
            gab[m] = {k: g[k] for k in ("us_per_layer", "launches_per_layer", "frac_of_bf16_mfma_peak", "executed_frac_of_bf16_mfma_peak") if g.get(k) is not None}
    if gab:
        line["roofline"]["global_attention_block"] = gab
    cb = result.get("cpu_baseline")
    if isinstance(cb, dict):
        line["cpu_baseline"] = {k: (_cut(cb[k], 220) if isinstance(cb[k], str) else cb[k]) for k in ("value", "unit", "cores", "cores_available", "kind", "sample", "error") if k in cb}
        if isinstance(cb.get("reference"), dict):
            line["cpu_baseline"]["reference"] = {k: cb["reference"].get(k) for k in ("value", "unit", "cores", "kind", "seconds")}
    par = result.get("parity")
    if isinstance(par, dict):
        line["parity"] = {k: (_cut(par[k], 160) if isinstance(par[k], str) else par[k]) for k in ("case", "rel_err", "gate", "pass", "error") if k in par}
        if isinstance(par.get("modes"), dict):
            line["parity"]["rel_err_by_mode"] = {m: e.get("rel_err_vs_reference_golden") for m, e in par["modes"].items()}
    if "mode_values" in result:
        line["mode_values"] = result["mode_values"]
    if isinstance(result.get("job"), dict) and "value" in result["job"]:
        line["job"] = {"value": result["job"]["value"], "ratio_to_value": result["job"].get("ratio_to_value")}
    if detail_file:
        line["detail_file"] = os.path.basename(detail_file)
    for drop in (("parity", "case"), ("cpu_baseline", "sample"), ("roofline", "global_attention_block"), ("config", "gather"), ("job",), ("parity",), ("mode_values",)):
        if len(json.dumps(line)) <= LINE_LIMIT:
            break
        node = line
        for k in drop[:-1]:
            node = node.get(k, {})
        node.pop(drop[-1], None)
    return line


def emit(result, detail_file):
    """Long form -> the detail file; the compact line -> stdout, alone and last."""
    if detail_file:
        try:
            with open(detail_file, "w") as f:
                json.dump(result, f, indent=1)
        except OSError as e:
            result = dict(result, detail_file_error=str(e))
            detail_file = None
    sys.stderr.flush()
    print(json.dumps(compact_line(result, detail_file)), file=LINE_OUT or sys.stdout, flush=True)


LINE_OUT = None


def claim_stdout():
    """The JSON line is the ONLY thing a rank of this job writes to the stdout it was started with: the line goes to a duplicate of that
    descriptor and descriptor 1 itself is pointed at stderr for everything else.  RCCL announces itself with a C-level printf
    ("Librccl path : ...") whose stdio buffer is flushed at process exit -- i.e. AFTER the line, on every rank -- and gloo prints its
    connection notes the same way; a reader that takes the last line of stdout would get those instead
    (tests/test_model_gpu.py::test_bench_runs_the_n_gpu_pass_over_rccl_on_one_gpu caught it on the hardware)."""
    global LINE_OUT
    if LINE_OUT is None:
        sys.stdout.flush()
        LINE_OUT = os.fdopen(os.dup(1), "w")
        os.dup2(2, 1)


def launch_ranks(args):
    """``python bench.py --gpus N`` typed plainly (no WORLD_SIZE in the environment): replace this process by the launcher the contract
    names -- ``python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py <same
    arguments>`` -- which starts one rank per GPU; rank 0 of that run prints the JSON line.  The port is a free one asked of the kernel."""
    import socket
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    os.environ["KDIFF_BENCH_LAUNCHER"] = "self"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.stdout.flush()
    sys.stderr.flush()
    os.execv(sys.executable, cmd)


def stub_line(args, ctx):
    """--stub-workload: the rank plumbing of this file (launcher, process group, barrier-bracketed timed region, max over ranks, the
    all-gather, ONE line from rank 0) around a tiny CPU tensor op -- what the launcher's CPU test drives over gloo.  Not a measurement."""
    B = 2
    x0 = torch.full((B, 3, 8, 8), float(ctx.process_index))

    def one_pass():
        return ctx.gather(x0 * 1.0)
    for _ in range(args.warmup):
        one_pass()
    ctx.wait_for_everyone()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = one_pass()
    ctx.wait_for_everyone()
    dt = time.perf_counter() - t0
    if ctx.num_processes > 1:
        t = torch.tensor([dt], dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = t.item()
    ranks_seen = sorted({int(v) for v in out[::B, 0, 0, 0].tolist()})
    if ctx.is_main_process:
        emit({"metric": "STUB (launcher / rank plumbing test, no GPU work)", "value": round(args.gpus * B * args.steps / dt, 3), "unit": "stub items/sec",
              "n_gpus": args.gpus, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
              "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "stub",
              "config": {"workload": "stub", "global_batch": B * args.gpus, "nranks": torch.distributed.get_world_size() if args.gpus > 1 else 1,
                         "backend": torch.distributed.get_backend() if args.gpus > 1 else None, "ranks_in_gather": ranks_seen,
                         "launcher": os.environ.get("KDIFF_BENCH_LAUNCHER", "external")}}, None)
    ctx.wait_for_everyone()
    ctx.shutdown()


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        launch_ranks(args)                         # does not return
    claim_stdout()
    os.environ["KDIFF_GEMM"] = args.mode
    if args.stub_workload:
        ctx = K.distributed.RankContext(device="cpu", backend=args.backend or "gloo")
    else:
        ctx = K.distributed.RankContext(backend=args.backend, force_collectives=args.force_collectives)
    if ctx.num_processes != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={ctx.num_processes}: start it plainly (python bench.py --gpus N launches its own ranks) "
                         f"or under torch.distributed.run with --nproc-per-node {args.gpus}")
    if args.stub_workload:
        return stub_line(args, ctx)
    if ctx.collectives:
        assert torch.distributed.get_world_size() == args.gpus and torch.distributed.get_backend() == "nccl", "one rank per GPU over RCCL"
    if ctx.device.type != "cuda":
        raise SystemExit("bench.py needs an MI355X: no ROCm device visible (there is no CPU fallback for the hot path)")
    dev = ctx.device
    cfg = K.config.load_config(os.path.join(REPO, args.config) if not os.path.isabs(args.config) else args.config)
    mc = cfg["model"]
    model = build_model(cfg, dev, args.seed)
    den = K.Denoiser(model, sigma_data=mc["sigma_data"])
    B = args.batch
    shape = (mc["input_channels"], *mc["input_size"])
    lo = ctx.process_index * B
    x0 = K.synth.synth_noise_batch(shape, args.seed, lo, B, mc["sigma_max"]).to(dev)
    extra = {}
    if cfg["dataset"]["num_classes"]:
        extra["class_cond"] = (torch.arange(lo, lo + B) % cfg["dataset"]["num_classes"]).to(dev)
    sigmas = K.sampling.get_sigmas_karras(args.sampler_steps, mc["sigma_min"], mc["sigma_max"], rho=7., device=dev)
    sampler = getattr(K.sampling, args.sampler)
    gather_events, last_gathered = [], [None]
    post = K.ops.to_uint8 if args.gather == "uint8" else (lambda t: t)

    def one_pass():
        imgs = sampler(den, x0, sigmas, extra_args=extra, disable=True)
        if not ctx.collectives:
            return imgs
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        gathered = ctx.gather(post(imgs))            # the path's one exchange step: RCCL all-gather over xGMI (evaluation.py:87), as sample.py does it
        e1.record()
        gather_events.append((e0, e1))
        last_gathered[0] = gathered
        return imgs

    def compute_only():
        return sampler(den, x0, sigmas, extra_args=extra, disable=True)

    timed = Timed(ctx, args.steps, args.warmup, not args.no_kernel_events)
    dt, out, groups = timed.run(one_pass, compute_only)
    gather_ms = sum(a.elapsed_time(b) for a, b in gather_events[-args.steps:]) / args.steps if gather_events else 0.0
    if gather_events:
        g = last_gathered[0]
        assert g.shape[0] == args.gpus * B and (g.dtype == torch.uint8) == (args.gather == "uint8")

    if ctx.is_main_process:
        n_img = args.gpus * B * args.steps
        head, fam = mode_entry(args.mode, dt, args.steps, args.warmup, n_img, groups, args.gpus)
        nfe_of = {"sample_dpmpp_2m": args.sampler_steps, "sample_euler": args.sampler_steps, "sample_lms": args.sampler_steps,
                  "sample_heun": 2 * args.sampler_steps - 1, "sample_dpmpp_sde": 2 * args.sampler_steps - 1}.get(args.sampler)
        if groups and nfe_of and "roofline" in head:
            head["roofline"]["global_attention_block"] = global_attention_block(groups, args.mode, cfg, B, nfe_of)
        if args.kernel_table and groups:
            with open(args.kernel_table, "w") as f:
                json.dump({"mode": args.mode, "families": fam, "kernels": groups, "timed_seconds": dt}, f, indent=1)
        mac = K.models.flops.forward_cost_mac(mc)["total"]
        nfe = args.sampler_steps if args.sampler == "sample_dpmpp_2m" else None
        px = B * shape[0] * shape[1] * shape[2]
        result = {
            "metric": "images/sec, 256x256 image_transformer_v2, 50-step DPM++2M (whole job)",
            "value": head["value"], "unit": "images/sec", "n_gpus": args.gpus, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": head["ms_per_step"], "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": MODE_DTYPE[args.mode][0], "mode": args.mode, "dtype_note": MODE_DTYPE[args.mode][1], "parity_grade": args.mode in ("split3", "exact"),
            "data": "synthetic (seeded noise, random-init weights incl. re-randomised zero-init projections)",
            "per_gpu": round(n_img / dt / args.gpus, 3),
            "config": {"workload": f"{os.path.basename(args.config)} {mc['input_size'][0]}x{mc['input_size'][1]}, {args.sampler} "
                                   f"{args.sampler_steps} steps, batch {B}/GPU, all-gather of finished images",
                       "global_batch": B * args.gpus, "parallelism": f"dp{args.gpus} (independent images, one final all-gather)",
                       "rccl_nranks": torch.distributed.get_world_size() if ctx.collectives else 1,
                       "launcher": os.environ.get("KDIFF_BENCH_LAUNCHER", "external") if args.gpus > 1 else None,
                       "gather": {"ms_per_step": round(gather_ms, 3), "dtype": args.gather, "bytes_per_rank": px * (1 if args.gather == "uint8" else 4),
                                  "bytes_per_rank_uint8": px, "bytes_per_rank_fp32": 4 * px,
                                  "note": "all_gather_into_tensor of the finished images as sample.py gathers them (uint8 conversion on the GPU first; "
                                          "--gather fp32 = the reference's fp32 gather), HIP events on the compute stream"}},
            "algorithmic_tflops": round(n_img * 2 * mac * (nfe or 0) / dt / 1e12, 2) if nfe else None,
            "roofline": head.get("roofline", {"bound": "hbm", "achieved": 0.0, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": 0.0, "traffic": None}),
        }
        if args.gpus == 1 and not args.no_other_modes:
            # the other arithmetic modes of the same build on the same box: same --steps / --warmup, own event pass, own roofline
            result["modes"] = {args.mode: {k: v for k, v in head.items() if k != "roofline"}}
            result["modes"][args.mode]["roofline"] = "see the top-level `roofline`"
            for m in [m for m in args.modes.split(",") if m and m != args.mode]:
                os.environ["KDIFF_GEMM"] = m
                try:
                    dt_m, out_m, groups_m = Timed(ctx, args.steps, args.warmup, not args.no_kernel_events).run(compute_only)
                except Exception as e:                    # a secondary mode must not cost the line
                    result["modes"][m] = {"error": f"{type(e).__name__}: {e}"[:600]}
                    continue
                ent, fam_m = mode_entry(m, dt_m, args.steps, args.warmup, B * args.steps, groups_m, 1)
                if groups_m and nfe_of and "roofline" in ent:
                    ent["roofline"]["global_attention_block"] = global_attention_block(groups_m, m, cfg, B, nfe_of)
                ent["max_rel_diff_vs_" + args.mode] = round(float((out_m - out[:B]).abs().max() / out[:B].abs().max()), 6)
                result["modes"][m] = ent
                if args.kernel_table and groups_m:
                    with open(args.kernel_table.replace(".json", f"_{m}.json"), "w") as f:
                        json.dump({"mode": m, "families": fam_m, "kernels": groups_m, "timed_seconds": dt_m}, f, indent=1)
            os.environ["KDIFF_GEMM"] = args.mode
        guarded = lambda key, fn: run_guarded(result, key, fn, args.mode)      # noqa: E731

        def block_other_configs():
            sw, na = "configs/config_oxford_flowers_shifted_window.json", "configs/config_oxford_flowers.json"
            from tests.golden import cases
            c64 = cases.SAMPLE_B64_CASE
            return {
                # BASELINE configs[0]: the reference's CPU-runnable plumbing case, here on the GPU (batch 4: the latency regime)
                "configs[0] mnist, euler x 10, batch 4": {
                    **{m: other_config("mnist", CONFIG_OF["mnist"], dev, args, "sample_euler", m, batch=4, steps=10, passes=20) for m in ("split3", "bf16")},
                    "parity": golden_parity(dev, ("split3", "bf16"), "samples.safetensors", "smp_mnist_euler10", "mnist", "sample_euler", 10, 4, None,
                                            "samples_bf16.safetensors")},
                # BASELINE configs[1]: global attention only, sample_heun x 50 (99 model calls) at its stated batch 64
                "configs[1] cifar, heun x 50, batch 64": {
                    **{m: other_config("cifar", CONFIG_OF["cifar"], dev, args, "sample_heun", m, batch=64) for m in ("bf16", "split3")},
                    "parity": golden_parity(dev, ("split3", "bf16"), "samples_r5.safetensors", *c64, cases.B64_KEEP, "samples_r5.safetensors")},
                # BASELINE configs[2]: the single-GPU 256x256 DPM++2M case with shifted-window attention
                "configs[2] shifted-window, dpmpp_2m": {
                    **{m: other_config("sw", sw, dev, args, "sample_dpmpp_2m", m) for m in ("split3", "bf16")},
                    "parity": golden_parity(dev, ("split3", "bf16"), "samples_r3.safetensors", "smp32_flowers_sw_2m5", "flowers_sw", "sample_dpmpp_2m", 5, 32,
                                            cases.B32_KEEP, "samples_r4.safetensors")},
                # BASELINE configs[3] is the headline workload of this line (`value`, `modes`, `parity`, `job`): its per-GPU share of the 8-GPU batch
                "configs[3] neighbourhood, dpmpp_2m": "the headline workload: see value / modes / parity / job of this line (32 images per GPU of the 8 x 32 batch)",
                # BASELINE configs[4]: neighbourhood attention, sample_dpmpp_sde x 50 (99 model calls, 98 Brownian queries of one tree per
                # image), fp8-stored weights where the arithmetic can hold them exactly (bf16 mode); the fp32-parity mode runs fp32 weights
                "configs[4] neighbourhood, dpmpp_sde + Brownian tree": {
                    "split3": other_config("sde", na, dev, args, "sample_dpmpp_sde", "split3", brownian=True),
                    "bf16+fp8w": other_config("sde", na, dev, args, "sample_dpmpp_sde", "bf16", fp8=True, brownian=True),
                    # round 6: the fp8 ARITHMETIC mode on the same fp8-stored weights (kd_gemm_mx8: e4m3 x e4m3 on the block-scaled matrix instruction)
                    "fp8": other_config("sde", na, dev, args, "sample_dpmpp_sde", "fp8", fp8=True, brownian=True)},
            }

        good_modes = [m for m, e in result.get("modes", {}).items() if "value" in e]

        def block_parity():
            # parity magnitudes where the driver's record shows them: after the timed regions, every measured mode against the reference's golden
            measured = [args.mode] + [m for m in good_modes if m != args.mode]
            par = parity_vs_reference_golden(dev, measured)
            head_par = par[args.mode]
            return {"case": par["case"], "measure": par["measure"], "mode": args.mode,
                    "rel_err": head_par["rel_err_vs_reference_golden"], "gate": head_par["gate"], "pass": head_par.get("pass"),
                    "modes": {m: par[m] for m in measured}}

        def block_job():
            measured = [args.mode] + [m for m in good_modes if m not in (args.mode, "exact")]
            job = job_rate(args, measured, dev)
            for m, ent in job["modes"].items():
                ref_value = head["value"] if m == args.mode else result["modes"][m]["value"]
                for e in ent.values():
                    if isinstance(e, dict):
                        e["ratio_to_value"] = round(e["value"] / ref_value, 4)
            job["value"] = job["modes"][args.mode]["host"]["value"]            # the CLI's default noise source
            job["ratio_to_value"] = job["modes"][args.mode]["host"]["ratio_to_value"]
            return job

        def block_power():
            sys.path.insert(0, os.path.join(REPO, "benchmarks"))
            import power_report
            return power_report.power_and_clock(compute_only)

        headline_cfg = os.path.basename(args.config) == "config_oxford_flowers.json"
        if args.gpus == 1 and not args.no_parity and headline_cfg and args.sampler == "sample_dpmpp_2m" and args.sampler_steps == 50 and B == 32:
            guarded("parity", block_parity)
        if "modes" in result:
            result["mode_values"] = {m: e.get("value") for m, e in result["modes"].items()}
        if not args.no_cpu_baseline and args.gpus == 1:
            guarded("cpu_baseline", lambda: cpu_baseline(cfg, args.seed, args.sampler_steps, args.cpu_seconds, args.cpu_batch))
        if args.gpus == 1 and args.other_configs and headline_cfg:
            guarded("other_configs", block_other_configs)
        if args.gpus == 1 and args.small_batch:
            guarded("small_batch", lambda: small_batch_latency(cfg, model, dev, args, measured_modes=[args.mode] + [m for m in good_modes if m != args.mode and m != "exact"]))
        if args.gpus == 1 and args.job:
            guarded("job", block_job)
        if args.gpus == 1 and args.power:
            guarded("power", block_power)
        emit(result, args.detail_file)
    ctx.wait_for_everyone()
    ctx.shutdown()


if __name__ == "__main__":
    main()
