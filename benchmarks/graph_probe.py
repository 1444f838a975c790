"""Probe: eager launch list vs hipGraph replay of the main chain, by batch size (launch-bound regime)."""
import sys, time, json, torch
sys.path.insert(0, ".")
import k_diffusion_amd as K
cfg = K.config.load_config(sys.argv[1] if len(sys.argv) > 1 else "configs/config_oxford_flowers.json")
torch.manual_seed(0)
inner = K.config.make_model(cfg).eval().requires_grad_(False).to("cuda")
den = K.Denoiser(inner, sigma_data=cfg["model"]["sigma_data"])
size = cfg["model"]["input_size"]
for B in (1, 2, 4, 8, 32):
    x = torch.randn(B, cfg["model"]["input_channels"], *size, device="cuda")
    sig = torch.full((B,), 1.5, device="cuda")
    extra = {}
    if cfg.get("dataset", {}).get("num_classes", 0):
        extra["class_cond"] = torch.zeros(B, dtype=torch.long, device="cuda")
    for _ in range(3):
        y = den(x, sig, **extra)
    torch.cuda.synchronize()
    n = 30
    t = time.perf_counter()
    for _ in range(n):
        y = den(x, sig, **extra)
    t_issue = (time.perf_counter() - t) / n
    torch.cuda.synchronize()
    t_eager = (time.perf_counter() - t) / n
    plan = [p for k, p in inner._plans.items() if k[0] == B][0]
    out = torch.empty_like(x)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        plan.run(x, out, cfg["model"]["sigma_data"], plan.last_buf)
    g.replay(); torch.cuda.synchronize()
    ok = torch.equal(out, y)
    t = time.perf_counter()
    for _ in range(n):
        g.replay()
    torch.cuda.synchronize()
    t_graph = (time.perf_counter() - t) / n
    print(json.dumps({"B": B, "launches": len(plan.launches), "eager_ms": round(t_eager * 1e3, 3), "eager_issue_ms": round(t_issue * 1e3, 3),
                      "graph_main_chain_ms": round(t_graph * 1e3, 3), "same": ok}))
