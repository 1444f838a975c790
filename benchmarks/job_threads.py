import os, sys, time, contextlib
sys.path.insert(0, os.getcwd())
import sample as cli
mode = sys.argv[1]
os.environ["KDIFF_GEMM"] = mode
base = ["--config", "configs/config_oxford_flowers.json", "--random-weights", "--seed", "0", "--batch-size", "32", "--steps", "50", "--sampler", "dpmpp_2m", "--no-png"]
for noise, thr in (("device", "0"), ("host", "2"), ("host", "4"), ("host", "8"), ("host", "16"), ("host", "32"), ("device", "0"), ("host", "4"), ("host", "16")):
    os.environ["KDIFF_NOISE_THREADS"] = thr
    with contextlib.redirect_stdout(sys.stderr):
        cli.main(base + ["-n", "32", "--noise", noise])
        cli.main(base + ["-n", "512", "--noise", noise])
    st = cli.LAST_RUN
    print(mode, noise, "threads", thr, f"{st['n'] / st['seconds']:.1f} images/s", f"{st['seconds']:.3f} s", flush=True)
