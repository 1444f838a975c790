"""Host-side tests that run WITHOUT a GPU: the C-ABI library loads and exports every symbol that
include/kdiff_hip.h declares (no compute calls), the hot path refuses to run on the CPU (there is no
fallback), config/CLI plumbing, and the multi-rank shard -> sample -> all-gather logic over gloo with
world_size 2 (the RCCL path of k_diffusion/evaluation.py:80-90, exercised on CPU tensors)."""
import ctypes
import os
import re
import subprocess
import sys

import pytest
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))



def _free_port():
    """A port asked of the kernel: fixed rendezvous ports collide when two runs of the suite share a host."""
    import socket
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        return sock.getsockname()[1]


def header_symbols():
    src = open(os.path.join(REPO, "include", "kdiff_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(kd_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol(KD):
    names = header_symbols()
    assert len(names) >= 20
    assert os.path.exists(KD._native.LIB_PATH), "build first: python -c 'import __graft_entry__ as g; g.build()'"
    handle = ctypes.CDLL(KD._native.LIB_PATH)
    missing = [n for n in names if not hasattr(handle, n)]
    assert not missing, f"libkdiff_hip.so does not export {missing}"
    # the ctypes table binds exactly the declared ABI (no stale / undeclared entry points)
    assert sorted(KD._native.SIGNATURES) == names
    lib = KD._native.lib()
    assert lib.kd_version() >= 100


def test_ctypes_structs_mirror_the_header(KD):
    """The descriptor structs of the C ABI are declared twice (include/kdiff_hip.h, k-diffusion_amd/_native.py): same field names in the
    same order with the same kind (pointer / int / float), or every call through them reads garbage."""
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    text = open(os.path.join(root, "include", "kdiff_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    for name in ("KdGemm", "KdFfn"):
        end = re.search(r"\}\s*" + name + r"\s*;", text).start()
        body = text[text.rindex("typedef struct", 0, end):end].split("{", 1)[1]
        fields = []
        for decl in body.split(";"):
            decl = decl.strip()
            if not decl:
                continue
            kind = "p" if "*" in decl else ("f" if decl.startswith("float") else "i")
            names = decl.replace("*", " ").split(None, 2 if decl.startswith("const") else 1)[-1]
            fields += [(n.strip(), kind) for n in names.split(",")]
        mirror = []
        for fname, ftype in getattr(KD._native, name)._fields_:
            mirror.append((fname, "p" if ftype is ctypes.c_void_p else ("f" if ftype is ctypes.c_float else "i")))
        assert fields == mirror, (name, [a for a, b in zip(fields, mirror) if a != b][:3], len(fields), len(mirror))


def _call_args(text, start):
    """Top-level arguments of the call whose opening parenthesis is at ``text[start]`` (balanced (), [] and {})."""
    depth, args, cur = 0, [], []
    for ch in text[start:]:
        if ch in "([{":
            depth += 1
            if depth == 1:
                continue
        elif ch in ")]}":
            depth -= 1
            if depth == 0:
                break
        if ch == "," and depth == 1:
            args.append("".join(cur).strip())
            cur = []
        else:
            cur.append(ch)
    else:
        raise AssertionError("unbalanced call")
    last = "".join(cur).strip()
    return args + ([last] if last else [])


def header_prototypes():
    """name -> argument count of every entry point include/kdiff_hip.h declares."""
    src = open(os.path.join(REPO, "include", "kdiff_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    out = {}
    for m in re.finditer(r"\b(kd_[a-z0-9_]+)\s*\(", src):
        args = _call_args(src, m.end() - 1)
        out[m.group(1)] = 0 if args in ([], ["void"]) else len(args)
    return out


def test_integration_stubs_match_the_abi(KD):
    """INTEGRATION.md shows the ctypes stubs a reference maintainer would paste.  Every ``_lib.kd_*(...)`` call in its code blocks, and
    every prototype-style ``kd_*(a, b, ...)`` line in their comments, must pass exactly as many arguments as the header declares (and the
    ctypes table binds): a missing argument shifts the stream pointer into the wrong slot without any error at the call site."""
    protos = header_prototypes()
    assert {k: len(v) for k, v in KD._native.SIGNATURES.items()} == protos        # the ctypes table agrees with the header
    text = open(os.path.join(REPO, "INTEGRATION.md")).read()
    blocks = re.findall(r"```python\n(.*?)```", text, flags=re.S)
    assert len(blocks) >= 4
    checked = {}
    for code in blocks:
        for m in re.finditer(r"_lib\.(kd_[a-z0-9_]+)\(", code):
            args = _call_args(code, m.end() - 1)
            assert len(args) == protos[m.group(1)], (m.group(1), len(args), protos[m.group(1)], args)
            checked[m.group(1)] = checked.get(m.group(1), 0) + 1
        for line in code.splitlines():                       # "# global:  kd_attn_global_bf16(qkv, out, n, h*w, nh, stream)  (remark)"
            if line.lstrip().startswith("#"):
                for m in re.finditer(r"(?<![._a-z])(kd_[a-z0-9_]+)\(", line):
                    args = _call_args(line, m.end() - 1)
                    if args and "..." not in args[-1]:
                        assert len(args) == protos[m.group(1)], (line.strip(), len(args), protos[m.group(1)])
                        checked[m.group(1)] = checked.get(m.group(1), 0) + 1
    for name in ("kd_attn_na2d_f32", "kd_attn_global_f32", "kd_attn_window_f32", "kd_attn_na2d_bf16", "kd_rmsnorm_f32", "kd_sampler_step_f32"):
        assert checked.get(name), f"INTEGRATION.md no longer shows a stub for {name}"
    # the argument in front of the stream of the fp32 attention cores is the arithmetic mode, by name
    for name in ("kd_attn_na2d_f32", "kd_attn_global_f32", "kd_attn_window_f32"):
        call = re.search(r"_lib\." + name + r"\(", text)
        args = _call_args(text, call.end() - 1)
        assert args[-2].startswith("KD_PREC_") and args[-1] == "_s()", (name, args[-2:])


def test_schedule_rows_are_recognised_by_storage_and_version(KD):
    """prefetch_schedule bookkeeping (host logic, no GPU): a model call is served from a schedule's scale table only when its sigma
    argument IS a row of the hinted table -- same storage, same version counter, whole contiguous fp32 row."""
    from importlib import import_module
    mod = import_module(KD.__name__ + ".models.image_transformer_v2")
    table = torch.linspace(80.0, 0.1, 6)[:, None].expand(6, 4).contiguous()
    sch = mod._Schedule(table, (None, None, None), (None, None, None), torch.zeros(6, 4, 8), None)
    assert [sch.row_of(table[i]) for i in range(6)] == list(range(6))
    assert sch.row_of(table[2].clone()) is None                       # same values, other storage
    assert sch.row_of(table[1, 1:]) is None and sch.row_of(table[:, 0]) is None and sch.row_of(table.view(-1)[2:6]) is None
    assert sch.row_of(table[3].double()) is None and sch.row_of(table[0:2]) is None
    other = torch.empty(7, 4)
    assert sch.row_of(other[0]) is None
    table[4] *= 0.5                                                   # in place: every row of the record is stale now
    assert all(sch.row_of(table[i]) is None for i in range(6))
    # the record keeps the hinted tensors alive (their addresses cannot be recycled while it exists)
    assert sch.keep[0] is table
    # tensors made under torch.inference_mode() have no version counter (reading it raises): they are never recognised as unchanged,
    # so such a table is simply not used and its calls take the per-step chain
    with torch.inference_mode():
        itab = torch.linspace(80.0, 0.1, 6)[:, None].expand(6, 4).contiguous()
    isch = mod._Schedule(itab, (None, None, None), (None, None, None), torch.zeros(6, 4, 8), None)
    assert all(isch.row_of(itab[i]) is None for i in range(6))
    assert mod._ver(itab) != mod._ver(itab) and mod._ver(table) == mod._ver(table)


def test_every_option_name_used_in_the_sources_is_registered():
    """kd_set_option refuses unknown names; every name the kernels' dispatch code reads (option("name", default)) and every name
    the Python layer maps an environment variable to must therefore be in the library's table (checked in a child process: a set
    option overrides the call sites' defaults for the rest of the process)."""
    csrc = os.path.join(REPO, "k-diffusion_amd", "csrc")
    names = set()
    for fn in os.listdir(csrc):
        if fn.endswith((".hip", ".cpp", ".h")):
            names |= set(re.findall(r'option\("([a-z0-9_]+)"', open(os.path.join(csrc, fn)).read()))
    names.discard("name")                      # the usage example in kd_common.h
    assert {"code_warm", "patch_fast", "ffn_fused", "wstat", "astat"} <= names
    code = (
        "import sys; sys.path.insert(0, %r)\n"
        "import k_diffusion_amd as K\n"
        "lib = K._native.lib()\n"
        "names = %r + [v[0] for v in K._native._ENV_OPTIONS.values()]\n"
        "bad = [n for n in names if lib.kd_set_option(n.encode(), 1) != 0]\n"
        "assert not bad, bad\n"
        "assert lib.kd_set_option(b'no_such_option', 1) == -1\n"
    ) % (REPO, sorted(names))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr


def test_code_objects_end_with_the_text_pad():
    """Kernels that warm their own code read up to 32 KiB behind their entry point: every such code object must end with the
    36 KiB .kd_text_pad section right behind .text (csrc/check_code_objects.py, also run by every build)."""
    import subprocess
    import sys
    csrc = os.path.join(REPO, "k-diffusion_amd", "csrc")
    objs = [os.path.join(csrc, f) for f in ("gemm_bf16.o", "patch_bf16.o", "ffn_bf16.o", "attn_bf16.o", "gemm.o", "gemm_astat.o", "attn_f32.o")]
    if not all(os.path.exists(o) for o in objs):
        pytest.skip("object files not in the tree (library built elsewhere)")
    out = subprocess.run([sys.executable, os.path.join(csrc, "check_code_objects.py"), *objs], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert out.stdout.count(": ok") == len(objs)
    # a code object without the pad is refused
    bad = subprocess.run([sys.executable, os.path.join(csrc, "check_code_objects.py"), os.path.join(csrc, "elementwise.o")], capture_output=True, text=True)
    assert bad.returncode != 0 and "kd_text_pad" in (bad.stdout + bad.stderr)


def test_bad_arguments_are_rejected_without_a_gpu(KD):
    """Argument validation happens before any launch: error code + message, never a throw."""
    lib = KD._native.lib()
    assert lib.kd_gemm_f32(None, None) == -1
    assert b"null descriptor" in lib.kd_last_error()
    d = KD._native.KdGemm()
    d.M, d.N, d.K = 8, 8, 6          # K % 4 != 0
    assert lib.kd_gemm_f32(ctypes.byref(d), None) == -1
    assert lib.kd_attn_window_f32(1, 1, 1, 16, 16, 1, 7, 0, 0, None, None, None, 1e-6, 1, None) == -1
    assert b"window_size" in lib.kd_last_error()
    assert lib.kd_attn_na2d_f32(1, 1, 1, 16, 16, 1, 5, 0, None, None, None, 1e-6, 1, None) == -1
    assert lib.kd_attn_global_f32(1, 1, 1, 0, 1, 0, None, None, None, 1e-6, 1, None) == -1
    assert lib.kd_attn_global_f32(1, 1, 1, 64, 1, 1, None, None, None, 1e-6, 1, None) == -1          # prep without its tables
    assert b"prep" in lib.kd_last_error()
    assert lib.kd_attn_global_f32(1, 1, 1, 64, 1, 0, None, None, None, 1e-6, 2, None) == -1          # bf16 is not an fp32-core precision
    assert b"precision" in lib.kd_last_error()
    # bf16 mode entry points
    d = KD._native.KdGemm()
    d.M, d.N, d.K, d.precision = 8, 8, 8, 1
    assert lib.kd_gemm_bf16(ctypes.byref(d), None) == -1
    assert b"KD_PREC_BF16" in lib.kd_last_error()
    assert lib.kd_attn_na2d_bf16(1, 1, 1, 16, 16, 1, 15, None) == -1          # kernel sizes 3 .. 13 (odd)
    assert b"kernel_size" in lib.kd_last_error()
    assert lib.kd_attn_window_bf16(1, 1, 1, 16, 16, 1, 7, 0, None) == -1
    assert lib.kd_attn_global_bf16(None, 1, 1, 16, 1, None) == -1
    assert lib.kd_set_option(None, 1) == -1 and lib.kd_set_option(b"wstat", 1) == 0 and lib.kd_get_option(b"wstat", 7) == 1
    assert lib.kd_brownian_cached_f32(1, None, None, 1, 0, 1, 1, 8, 0.0, 1.0, 0.2, 0.4, 1.0, 36, None) == -1
    assert b"cached end point" in lib.kd_last_error()
    assert lib.kd_brownian_f32(1, 1, 1, 8, 0.0, 1.0, 0.5, 0.4, 1.0, 36, None) == -1            # t0 > t1


def test_fused_ff_is_advised_only_where_it_is_the_faster_form(KD):
    """kd_ffn_f32_supported (no launch, runs without a GPU): the width-128 fused FF block from 16 row panels on; the width-256 one (one workgroup per CU
    for ~120 us whatever the grid) only from a chip-filling grid on -- at batch 4 of the headline config it ran 32 workgroups for 95 us where the
    two-launch form takes ~35 (profiles/r04_small_batch.log)."""
    lib = KD._native.lib()
    assert lib.kd_ffn_f32_supported(2048, 128, 384) == 1 and lib.kd_ffn_f32_supported(1024, 128, 384) == 0
    assert lib.kd_ffn_f32_supported(32768, 256, 768) == 1            # batch 32 of the headline config: 256 panels
    assert lib.kd_ffn_f32_supported(4096, 256, 768) == 0 and lib.kd_ffn_f32_supported(16384, 256, 768) == 0
    assert lib.kd_ffn_f32_supported(32768, 256, 100) == 0 and lib.kd_ffn_f32_supported(32768, 512, 1536) == 0


def test_hot_path_has_no_cpu_fallback(KD):
    cfg = KD.config.load_config(os.path.join(REPO, "configs", "config_mnist_transformer.json"))
    model = KD.config.make_model(cfg)
    x = torch.zeros(1, 1, 28, 28)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        model(x, torch.ones(1), class_cond=torch.zeros(1, dtype=torch.int64))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        KD.sampling.sample_dpmpp_2m(lambda x, s: x, x, KD.sampling.get_sigmas_karras(3, 0.01, 80.))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        KD.ops.rms_norm(torch.zeros(4, 8), torch.ones(8))
    with pytest.raises(ValueError, match="class_cond"):
        model(x, torch.ones(1))


def test_unsupported_head_size_is_refused_with_a_reason(KD):
    """The reference takes any d_head (image_transformer_v2.py:355-363); the HIP kernels of both arithmetic modes are built for 64 (the
    only value the shipped configs use).  Anything else is refused at construction, by name, not misread by a kernel."""
    raw = {"model": {"type": "image_transformer_v2", "input_channels": 3, "input_size": [32, 32], "patch_size": [2, 2], "depths": [1],
                     "widths": [128], "self_attns": [{"type": "global", "d_head": 32}], "sigma_data": 0.5, "sigma_min": 1e-2, "sigma_max": 80}}
    with pytest.raises(ValueError, match="d_head must be 64"):
        KD.config.make_model(KD.config.load_config(raw))


def test_product_package_never_imports_the_oracle():
    """oracle/ is test infrastructure: nothing under k-diffusion_amd/ or sample.py may import it."""
    offenders = []
    for root, _, files in os.walk(os.path.join(REPO, "k-diffusion_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h")):
                if re.search(r"^\s*(from|import)\s+oracle\b", open(os.path.join(root, f)).read(), flags=re.M):
                    offenders.append(f)
    if re.search(r"^\s*(from|import)\s+oracle\b", open(os.path.join(REPO, "sample.py")).read(), flags=re.M):
        offenders.append("sample.py")
    # the benchmark scripts and the CLI shims neither; bench.py only inside cpu_baseline() (the timed CPU leg the contract allows)
    for root, _, files in os.walk(os.path.join(REPO, "benchmarks")):
        for f in files:
            if f.endswith((".py", ".sh", ".cpp")) and re.search(r"^\s*(from|import)\s+oracle\b", open(os.path.join(root, f)).read(), flags=re.M):
                offenders.append("benchmarks/" + f)
    for f in ("convert_for_inference.py", "config_from_inference.py", "k_diffusion_amd.py"):
        if re.search(r"^\s*(from|import)\s+oracle\b", open(os.path.join(REPO, f)).read(), flags=re.M):
            offenders.append(f)
    bench = open(os.path.join(REPO, "bench.py")).read()
    sites = [m.start() for m in re.finditer(r"^\s*(from|import)\s+oracle\b", bench, flags=re.M)]
    body = bench[bench.index("def cpu_baseline("):bench.index("CPU_THREAD_CAP")]
    assert len(sites) == 1 and re.search(r"^\s*from oracle import", body, flags=re.M), "bench.py may import the oracle in cpu_baseline() only"
    assert not offenders, offenders


def test_configs_merge_like_the_reference(KD, golden):
    """load_config's defaulting (config.py:58-146) against the merged dicts recorded from the reference."""
    recorded = golden["kat"].get("merged_configs")
    if not recorded:
        pytest.skip("golden set carries no merged configs")
    for name, ref in recorded.items():
        got = KD.config.load_config(os.path.join(REPO, "configs", name))
        # the repo's configs carry the model-defining keys only (no training / dataset-location entries)
        for k in ("type", "input_channels", "input_size", "patch_size", "depths", "widths", "d_ffs", "self_attns",
                  "mapping_width", "mapping_depth", "mapping_d_ff", "mapping_cond_dim", "sigma_data", "sigma_min",
                  "sigma_max", "has_variance", "loss_config"):
            assert got["model"][k] == ref["model"][k], (name, k)
        assert got["dataset"]["num_classes"] == ref["dataset"]["num_classes"], name


def test_checkpoint_tools_round_trip(KD, tmp_path):
    """Training checkpoint -> slim inference checkpoint (safetensors + config metadata) -> config / weights back
    (convert_for_inference.py, config_from_inference.py, config.py:113-115, sample.py:33,44)."""
    import json
    import safetensors.torch as safetorch
    sys.path.insert(0, REPO)
    import config_from_inference
    import convert_for_inference
    raw = json.load(open(os.path.join(REPO, "configs", "config_mnist_transformer.json")))
    model = KD.config.make_model(KD.config.load_config(raw))
    sd = KD.synth.synth_state_dict(model.state_dict(), seed=3)
    pth = tmp_path / "run_00001000.pth"
    torch.save({"config": raw, "model_ema": sd, "step": 1000}, pth)
    convert_for_inference.main([str(pth), "--dtype", "bf16"])
    slim = pth.with_suffix(".safetensors")
    assert slim.exists()
    cfg = KD.config.load_config(slim)                                 # config straight from the checkpoint's metadata
    assert cfg["model"]["widths"] == raw["model"]["widths"]
    loaded = safetorch.load_file(str(slim))
    assert set(loaded) == set(sd) and all(v.dtype == torch.bfloat16 for k, v in loaded.items() if sd[k].is_floating_point())
    model.load_state_dict(loaded)                                     # upcast to the fp32 parameters on load
    assert model.patch_in.proj.weight.dtype == torch.float32
    config_from_inference.main([str(slim), "-o", str(tmp_path / "c.json")])
    assert json.load(open(tmp_path / "c.json")) == raw
    with pytest.raises(ValueError, match="No configuration"):
        KD.checkpoint.write_inference_checkpoint(sd, None, tmp_path / "x.safetensors")


def test_fp8_weight_checkpoint(KD, tmp_path):
    """convert_for_inference.py --dtype fp8 (BASELINE configs[4]; no reference counterpart, convert_for_inference.py:23): projection
    weights stored as e4m3 + one power-of-two scale per output channel, everything else fp32; the loader's expansion is bit for
    bit ``fake_quantize_fp8`` of the original weight and EXACT in bf16 (so the bf16 arithmetic mode runs on the fp8 weights)."""
    import json
    import safetensors.torch as safetorch
    sys.path.insert(0, REPO)
    import convert_for_inference
    ck = KD.checkpoint
    raw = json.load(open(os.path.join(REPO, "configs", "config_mnist_transformer.json")))
    model = KD.config.make_model(KD.config.load_config(raw))
    sd = KD.synth.synth_state_dict(model.state_dict(), seed=5)
    pth = tmp_path / "run.pth"
    torch.save({"config": raw, "model_ema": sd}, pth)
    convert_for_inference.main([str(pth), "--dtype", "fp8", "-o", str(tmp_path / "m8.safetensors")])
    stored = safetorch.load_file(str(tmp_path / "m8.safetensors"))
    q_names = [k for k, v in sd.items() if ck.is_fp8_weight(k, v)]
    assert len(q_names) > 20 and "patch_in.proj.weight" in q_names and "class_emb.weight" not in q_names and "time_emb.weight" not in q_names
    for k in q_names:
        assert stored[k].dtype == torch.float8_e4m3fn and stored[k + ck.FP8_SCALE_SUFFIX].shape == (sd[k].shape[0],)
        sc = stored[k + ck.FP8_SCALE_SUFFIX]
        assert torch.equal(torch.log2(sc), torch.log2(sc).round())                       # powers of two
        assert (sd[k].abs().amax(1) / sc <= ck.FP8_MAX).all() and (sd[k].abs().amax(1) / sc > ck.FP8_MAX / 2 - 1e-3).all()   # the tightest such scale
    assert all(stored[k].dtype == torch.float32 for k, v in sd.items() if v.is_floating_point() and k not in q_names)
    loaded = ck.load_inference_checkpoint(tmp_path / "m8.safetensors")
    ref = ck.fp8_state_dict(sd)
    assert set(loaded) == set(sd)
    for k in sd:
        assert torch.equal(loaded[k], ref[k]), k
    for k in q_names:
        w8 = loaded[k]
        assert torch.equal(w8.to(torch.bfloat16).float(), w8)                            # exact in bf16: 4 significand bits x 2^k
        rel = (w8 - sd[k]).abs() / sd[k].abs().amax(1, keepdim=True)
        assert rel.max() <= 2.0 ** -4 + 1e-6                                             # half an e4m3 step at the top binade, relative to the row maximum
    model.load_state_dict(loaded)
    assert KD.config.load_config(tmp_path / "m8.safetensors")["model"]["widths"] == raw["model"]["widths"]
    assert os.path.getsize(tmp_path / "m8.safetensors") < 0.4 * sum(v.numel() * 4 for v in sd.values())
    # a scale tensor that lost its weight (or the reverse) is an error, not a silent fp8 -> fp32 cast
    del stored[q_names[0] + ck.FP8_SCALE_SUFFIX]
    safetorch.save_file(stored, str(tmp_path / "bad.safetensors"))
    with pytest.raises(ValueError, match="fp8_scale"):
        ck.load_inference_checkpoint(tmp_path / "bad.safetensors")


def test_sample_cli_contract():
    sys.path.insert(0, REPO)
    import sample
    a = sample.parse(["--checkpoint", "m.safetensors"])
    assert (a.batch_size, a.n, a.prefix, a.steps, a.sampler) == (64, 64, "out", 50, "lms")      # reference defaults
    a = sample.parse(["--config", "c.json", "--random-weights", "--sampler", "dpmpp_2m", "--seed", "3", "-n", "8"])
    assert a.random_weights and a.seed == 3 and a.n == 8
    with pytest.raises(SystemExit):
        sample.parse([])                                    # --checkpoint required, as in the reference
    assert sample.resolve_sampler("dpmpp_2m").__name__ == "sample_dpmpp_2m"
    assert sample.resolve_sampler("sample_heun").__name__ == "sample_heun"
    with pytest.raises(SystemExit):
        sample.resolve_sampler("nope")


def test_shard_range_covers_everything(KD):
    for n in (1, 7, 64, 100, 256):
        for world in (1, 2, 3, 8):
            spans = [KD.distributed.shard_range(n, world, r) for r in range(world)]
            got = [g for lo, hi in spans for g in range(lo, hi)]
            assert got == list(range(n))


_WORKER = r"""
import os, sys, torch
sys.path.insert(0, {repo!r})
import k_diffusion_amd as K
ctx = K.distributed.RankContext(device="cpu", backend="gloo")
assert ctx.num_processes == 2
n, bs, shape = 11, 3, (1, 4, 4)
per_rank = -(-n // ctx.num_processes)
cursor = [ctx.process_index * per_rank]
def sample_fn(k):                       # stands in for the sampler: a pure function of the GLOBAL image index
    lo = cursor[0]; cursor[0] += k
    return torch.stack([K.synth.synth_noise(shape, 5, lo + g, 1.0) for g in range(k)])
out = K.evaluation.compute_features(ctx, sample_fn, lambda x: x, n, bs)
g = ctx.gather(torch.full((2, 3), float(ctx.process_index)))
assert g.shape == (4, 3) and g[:2].eq(0).all() and g[2:].eq(1).all()
ctx.wait_for_everyone()
if ctx.is_main_process:
    torch.save(out, {out!r})
ctx.shutdown()
"""


_WORKER_INDEXED = r"""
import os, sys, torch
sys.path.insert(0, {repo!r})
import k_diffusion_amd as K
import sample
ctx = K.distributed.RankContext(device="cpu", backend="gloo")
assert ctx.num_processes == {world}
args = sample.parse(["--config", "c.json", "--random-weights", "--seed", "9", "--class-cond", "-1", "-n", "{n}", "--batch-size", "{bs}"])
shape, calls = (1, 4, 4), []
def sample_fn(idx):                     # stands in for model + sampler: a pure function of the GLOBAL image indices it is given
    calls.append(len(idx))
    if len(idx) == 0:
        return torch.empty(0, 3, 4, 4)
    x0 = torch.stack([K.synth.synth_noise(shape, args.seed, int(g), 1.0) for g in idx])
    cc = sample.class_ids(args, 10, idx, "cpu")
    bseed = torch.tensor(sample.brownian_seeds(args.seed, idx), dtype=torch.int64)
    noise = sample.indexed_noise_sampler(args.seed, idx, shape, "cpu")
    n1, n2 = noise(None, None), noise(None, None)
    feat = torch.cat([x0, (cc.float() / 10)[:, None, None, None].expand_as(x0) + n1 * 1e-3, ((bseed % 1000).float() / 1000)[:, None, None, None] + n2 * 1e-3], dim=1)
    return feat.clamp(-1, 1)
lo, hi = K.distributed.shard_range({n}, ctx.num_processes, ctx.process_index)
out = K.evaluation.compute_features_indexed(ctx, sample_fn, {n}, {bs})
out8 = K.evaluation.compute_features_indexed(ctx, sample_fn, {n}, {bs}, post=lambda x: (((x + 1) / 2) * 255).to(torch.uint8))
assert sum(calls) == 2 * (hi - lo), (calls, lo, hi)            # every rank sampled exactly its own shard, nothing else
ctx.wait_for_everyone()
if ctx.is_main_process:
    torch.save((out, out8), {out!r})
ctx.shutdown()
"""


@pytest.mark.parametrize("world,n,bs,port", [(8, 21, 2, 29641), (2, 7, 4, 29642), (8, 5, 3, 29643)])
def test_indexed_gather_many_ranks_over_gloo(KD, tmp_path, world, n, bs, port):
    """The data-parallel sampling schedule of sample.py on 8 (and 2) gloo ranks with an n the ranks do not divide (the last
    ranks get short or EMPTY shards): out[i] is image i -- start noise, class id, Brownian-tree seed and ancestral noise stream
    all functions of the global index -- identical to a single-process run, also through the uint8 gather."""
    sys.path.insert(0, REPO)
    import sample
    out = str(tmp_path / "gathered.pt")
    script = tmp_path / "worker.py"
    script.write_text(_WORKER_INDEXED.format(repo=REPO, out=out, world=world, n=n, bs=bs))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
                        "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), str(script)],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    got, got8 = torch.load(out)
    shape = (1, 4, 4)
    idx = torch.arange(n)
    x0 = torch.stack([KD.synth.synth_noise(shape, 9, int(g), 1.0) for g in idx])
    noise = sample.indexed_noise_sampler(9, idx, shape, "cpu")
    n1, n2 = noise(None, None), noise(None, None)
    bseed = torch.tensor(sample.brownian_seeds(9, idx), dtype=torch.int64)
    assert len(set(bseed.tolist())) == n and (bseed >= 0).all()
    expect = torch.cat([x0, ((idx % 10).float() / 10)[:, None, None, None].expand_as(x0) + n1 * 1e-3,
                        ((bseed % 1000).float() / 1000)[:, None, None, None] + n2 * 1e-3], dim=1).clamp(-1, 1)
    assert got.shape == (n, 3, 4, 4) and torch.equal(got, expect)
    assert got8.dtype == torch.uint8 and torch.equal(got8, (((expect + 1) / 2) * 255).to(torch.uint8))


def test_two_rank_gather_over_gloo(KD, tmp_path):
    """world_size 2 over gloo: every rank samples its shard in batches, batches are all-gathered, and the
    result holds each rank's images (keyed by GLOBAL index, so it does not depend on the world size)."""
    out = str(tmp_path / "gathered.pt")
    script = tmp_path / "worker.py"
    script.write_text(_WORKER.format(repo=REPO, out=out))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                        "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), str(script)],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    got = torch.load(out)
    n, bs, shape = 11, 3, (1, 4, 4)
    per_rank = 6
    # reference gather order (evaluation.py:84-88): for each batch step, rank 0's batch then rank 1's, cut to n
    expect = []
    for i in range(0, per_rank, bs):
        for rank in range(2):
            expect += [KD.synth.synth_noise(shape, 5, rank * per_rank + i + g, 1.0) for g in range(min(n - i, bs))]
    expect = torch.stack(expect)[:n]
    assert got.shape == (n, *shape)
    assert torch.equal(got, expect)


def test_option_sync_is_incremental():
    """KDIFF_* / KDIFF_OPTIONS -> kd_set_option (round-2 advisor): only what changed is re-applied, an entry that disappears from
    KDIFF_OPTIONS goes back to the built-in default, and a value set through set_option survives changes of OTHER variables.
    Runs in a child process: options are process-global."""
    code = (
        "import os, sys; sys.path.insert(0, %r)\n"
        "import k_diffusion_amd as K\n"
        "nat = K._native\n"
        "get = lambda n, d: nat.lib().kd_get_option(n.encode(), d)\n"
        "os.environ['KDIFF_OPTIONS'] = 'wstat=0,tiled_bm=256'\n"
        "assert get('wstat', 1) == 0 and get('tiled_bm', 0) == 256\n"
        "nat.set_option('ffn_fused', 0)\n"
        "os.environ['KDIFF_OPTIONS'] = 'tiled_bm=128'\n"          # wstat disappears: default again; ffn_fused (programmatic) untouched
        "assert get('wstat', 1) == 1 and get('wstat', 5) == 5 and get('tiled_bm', 0) == 128 and get('ffn_fused', 1) == 0\n"
        "os.environ['KDIFF_OPTIONS'] = 'tiled_bm=128,skinny=0'\n"
        "assert get('skinny', 1) == 0 and get('ffn_fused', 1) == 0 and get('tiled_bm', 0) == 128\n"
        "os.environ['KDIFF_OPTIONS'] = 'tiled_bm=128'\n"
        "assert get('skinny', 1) == 1\n"
        "os.environ['KDIFF_OPTIONS'] = 'ffn_fused=1'\n"          # the environment names it anew: it wins again
        "assert get('ffn_fused', 0) == 1 and get('tiled_bm', 7) == 7\n"
        "e = nat.option_epoch; nat.lib(); nat.lib(); assert nat.option_epoch == e\n"     # nothing changed: nothing re-applied
        "os.environ['KDIFF_OPTIONS'] = 'no_such=1'\n"
        "try:\n    nat.lib(); raise SystemExit('accepted an unknown option')\nexcept ValueError:\n    pass\n"
        "print('ok')\n") % REPO
    env = {k: v for k, v in os.environ.items() if not k.startswith("KDIFF_") and k != "KD_GEMM_DEBUG"}
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stdout + r.stderr


def test_oracle_brownian_tree_stands_in_for_torchsde():
    """oracle.brownian.OracleBrownianTree (what make_golden_r3.py installs as torchsde.BrownianTree under the REFERENCE's
    BatchedBrownianTree): call surface, additivity over adjacent intervals, unit variance per unit time, determinism per seed."""
    from oracle import brownian as ob
    w0 = torch.zeros(3, 64, 64)
    tree = ob.OracleBrownianTree(0.01, w0, 160.0, entropy=1234)
    ab, bc, ac = tree(1.0, 2.0), tree(2.0, 5.5), tree(1.0, 5.5)
    assert ab.shape == w0.shape and ab.dtype == torch.float32
    assert (ab + bc - ac).abs().max() < 1e-5
    assert abs(float(ac.var()) / 4.5 - 1.0) < 0.08 and abs(float(tree(0.02, 0.03).var()) / 0.01 - 1.0) < 0.08
    assert torch.equal(ob.OracleBrownianTree(0.01, w0, 160.0, entropy=1234)(1.0, 2.0), ab)
    assert not torch.equal(ob.OracleBrownianTree(0.01, w0, 160.0, entropy=1235)(1.0, 2.0), ab)
    assert torch.equal(tree(2.0, 1.0), -ab)
    if ob_ref_available():
        from oracle import ref_import
        K = ref_import.load()
        K.sampling.torchsde.BrownianTree = ob.OracleBrownianTree
        x = torch.zeros(2, 3, 8, 8)
        ns = K.sampling.BrownianTreeNoiseSampler(x, 0.01, 160.0, seed=[7, 8])            # the reference's own class
        n = ns(torch.tensor(4.0), torch.tensor(1.0))                                      # descending sigmas, as the samplers ask
        inc = ob.brownian_increment([7, 8], 192, 0.01, 160.0, 1.0, 4.0)
        want = -torch.from_numpy(inc).view(2, 3, 8, 8) / (torch.tensor(1.0) - torch.tensor(4.0)).abs().sqrt()
        assert torch.allclose(n, want, atol=1e-4)          # (the reference hands the tree fp32 end points: 0.01 -> 0.0099999998)


def ob_ref_available():
    from oracle import ref_import
    return ref_import.available()


def test_run_list_entries_are_packed_by_the_rule_the_library_unpacks_by():
    """kd_run_list (include/kdiff_hip.h): pointer arguments in declaration order in p[], int arguments in i[], the float in f; a descriptor
    goes in by address; an argument patched per run is told its slot.  Host logic only: no launch."""
    import ctypes as C
    import k_diffusion_amd as K
    nat = K._native
    assert C.sizeof(nat.KdCall) == 80 and nat.KdCall.p.offset == 8 and nat.KdCall.i.offset == 48        # the C struct's layout
    c = nat.KdCall()
    assert nat.encode_call(c, "kd_attn_window_f32", (C.c_void_p(16), C.c_void_p(32), 2, 8, 16, 4, 8, 4, 2, None, None, None, C.c_float(1e-6), 1))
    assert c.op == 5 and list(c.i) == [2, 8, 16, 4, 8, 4, 2, 1] and [c.p[k] for k in range(5)] == [16, 32, None, None, None] and abs(c.f - 1e-6) < 1e-12
    d = nat.KdGemm()
    assert nat.encode_call(c, "kd_gemm_bf16", (d,)) and c.op == 1 and c.p[0] == C.addressof(d)

    class Ref:
        scale = 4096

        def bind_call(self, call, index):
            self.where = (call, index)
    r = Ref()
    assert nat.encode_call(c, "kd_norm_split_f32", (64, r, 128, 256, 512, 1024, 256, 128, 1e-6))
    assert c.op == 10 and r.where[1] == 1 and [c.p[k] for k in range(4)] == [64, 4096, 512, 1024] and list(c.i)[:4] == [128, 256, 256, 128]
    assert not nat.encode_call(c, "kd_rmsnorm_f32", (1, 2, 3, 4, 5, 1e-6))             # not an entry point a list can name
    with pytest.raises(ValueError):
        nat.encode_call(c, "kd_gemm_f32", (d, 1))
    # every name in the table is a declared entry point whose last parameter is the stream
    for name in nat.RUN_LIST_OPS:
        assert nat.SIGNATURES[name][-1] is C.c_void_p


def test_weights_fingerprint_keeps_its_tensor_list_and_still_sees_every_change():
    """The per-call weights check of the model: the list of parameters / buffers is kept between calls -- walking the module tree cost
    0.3 - 0.4 ms per model call -- and what keeps it honest is a per-call identity check of every (module dict, name) slot of the model's
    OWN tree (no process-wide hooks).  Host logic only (no device): every way a weight can change must move the fingerprint -- including
    direct writes into ``module._parameters`` (torch.func.functional_call swaps parameters that way, past all registration hooks) -- and
    an untouched model must not walk the tree again."""
    import torch
    import k_diffusion_amd as K
    from k_diffusion_amd.models import image_transformer_v2 as itv2
    cfg = K.config.load_config(os.path.join(REPO, "configs", "config_mnist_transformer.json"))
    m = K.config.make_model(cfg).eval().requires_grad_(False)
    f0 = m._weights_fingerprint()
    assert len(f0) == 1 + len(list(m.parameters())) + len(list(m.buffers()))      # the epoch, then one entry per tensor
    walked = []
    orig = type(m).modules
    try:
        type(m).modules = lambda self, *a, **k: (walked.append(1), orig(self, *a, **k))[1]
        assert m._weights_fingerprint() == f0 and not walked                      # unchanged: the kept list, no traversal
        m.load_state_dict(K.synth.synth_state_dict(m.state_dict(), seed=1))     # in-place copies: versions move (and the epoch)
        f1 = m._weights_fingerprint()
        assert f1 != f0 and not walked
        with torch.no_grad():
            m.out_norm.scale.mul_(2.0)
        f2 = m._weights_fingerprint()
        assert f2 != f1 and not walked
        m.patch_in.proj.weight = torch.nn.Parameter(torch.zeros_like(m.patch_in.proj.weight), requires_grad=False)      # a new tensor object
        f3 = m._weights_fingerprint()
        assert f3 != f2 and walked and any(t is m.patch_in.proj.weight for t in m._fp_tensors)
        del walked[:]
        p = m.mapping.out_norm.scale
        p.data = p.data.clone()                                                   # same Parameter, other storage (what Module.to() does)
        f4 = m._weights_fingerprint()
        assert f4 != f3 and not walked
        # a write straight into the dict, past __setattr__ and every registration hook
        old = m.patch_out.proj._parameters["weight"]
        m.patch_out.proj._parameters["weight"] = torch.nn.Parameter(old.detach().clone() + 1.0, requires_grad=False)
        f5 = m._weights_fingerprint()
        assert f5 != f4 and walked and any(t is m.patch_out.proj._parameters["weight"] for t in m._fp_tensors)
        del walked[:]
        # torch.func.functional_call swaps parameters through the dicts for the duration of the call: the model must see the swapped
        # tensors INSIDE the call (plans / packed images built from the real weights must not serve it) and the real ones after it
        inside = []
        real_forward = type(m).forward
        type(m).forward = lambda self, *a, **k: inside.append(self._weights_fingerprint())
        try:
            before = m._weights_fingerprint()
            swapped = {k: v.detach().clone() + 0.5 for k, v in m.named_parameters()}
            torch.func.functional_call(m, swapped, ())
            assert inside and inside[0] != before
            addrs = {t.data_ptr() for t in swapped.values()}
            assert {a for a, _ in inside[0][1:]} >= addrs                          # the fingerprint read inside the call is of the SWAPPED tensors
            after = m._weights_fingerprint()
            assert after == before                                                # restored: same objects, addresses and versions as before the call
        finally:
            type(m).forward = real_forward
        del walked[:]
        # a replaced sub-module (same name, other object)
        m.out_norm = type(m.out_norm)(scale=torch.nn.Parameter(torch.ones_like(m.out_norm.scale)))
        f6 = m._weights_fingerprint()
        assert f6 != after and walked
        del walked[:]
        m.double()                                                                # a conversion: epoch bumped, addresses move
        f7 = m._weights_fingerprint()
        assert f7 != f6
        # weights made under inference_mode carry no version counter: load_state_dict (post-hook) and invalidate() still move the fingerprint
        with torch.inference_mode():
            mi = K.config.make_model(cfg).eval()
        g0 = mi._weights_fingerprint()
        with torch.inference_mode():                                              # (in-place edits of inference tensors are only legal here)
            mi.load_state_dict({k: v + 1 for k, v in mi.state_dict().items()})
        g1 = mi._weights_fingerprint()
        assert g1 != g0
        mi.invalidate()
        assert mi._weights_fingerprint() != g1
        # no import side effect on unrelated modules: the package installs no process-wide registration hooks
        from torch.nn.modules import module as tmod
        for table in (tmod._global_parameter_registration_hooks, tmod._global_buffer_registration_hooks, tmod._global_module_registration_hooks):
            assert not any(getattr(h, "__module__", "").startswith("k-diffusion_amd") or getattr(h, "__module__", "").startswith("k_diffusion_amd") for h in table.values())
        assert not hasattr(itv2, "_registration_epoch")
    finally:
        type(m).modules = orig


def test_public_signatures_match_the_reference(KD):
    """Drop-in, checked: every public function, class and public method the reference's hot-path modules define
    (k_diffusion/{sampling,layers,external,config}.py, recorded by oracle/make_golden_signatures.py into tests/golden/signatures.json) exists
    here with the same parameter names, order, kinds and defaults (the product may ADD trailing keyword parameters); what is absent is exactly the
    out-of-scope list of DESIGN.md section 8 (training, the U-Net family, likelihood evaluation)."""
    import inspect
    import json
    with open(os.path.join(REPO, "tests", "golden", "signatures.json")) as f:
        golden = json.load(f)
    out_of_scope = {
        "sampling": {"log_likelihood"},                                          # needs torchdiffeq + autograd through the model
        "config": {"make_sample_density", "round_to_power_of_two"},              # training-time sigma densities
        "layers": {"AdaGN", "ConditionedModule", "ConditionedResidualBlock", "ConditionedSequential", "CrossAttention2d", "Downsample2d",
                   "ResidualBlock", "SelfAttention2d", "UNet", "UnconditionedModule", "Upsample2d", "dct"},      # the U-Net (image_v1) family
        "external": set(),
    }
    training_methods = {"loss"}

    def params(f):
        out = []
        for p in inspect.signature(f).parameters.values():
            d = p.default
            out.append([p.name, p.kind.name, "<required>" if d is inspect.Parameter.empty else "<callable>" if callable(d) else repr(d)])
        return out

    def same(ref, got, where):
        assert got[:len(ref)] == ref, (where, ref, got)
        assert all(kind in ("KEYWORD_ONLY", "VAR_KEYWORD") or default != "<required>" for _, kind, default in got[len(ref):]), (where, got[len(ref):])

    checked = 0
    for modname, names in golden.items():
        mod = getattr(KD, modname)
        assert {n for n in names if not hasattr(mod, n)} == out_of_scope[modname], modname
        for name, ent in names.items():
            obj = getattr(mod, name, None)
            if obj is None:
                continue
            if ent["kind"] == "function":
                same(ent["params"], params(obj), f"{modname}.{name}")
                checked += 1
            else:
                for mname, ref in ent["methods"].items():
                    if mname in training_methods:
                        continue
                    meth = getattr(obj, mname, None)
                    assert meth is not None, f"{modname}.{name}.{mname}"
                    same(ref, params(meth), f"{modname}.{name}.{mname}")
                    checked += 1
    assert checked >= 60, checked


def test_bench_launches_its_own_ranks(tmp_path):
    """``python bench.py --gpus 2`` typed plainly (no WORLD_SIZE): bench.py re-executes itself under torch.distributed.run with one rank per
    device, rank 0 prints ONE JSON line.  Driven here over gloo with the stub workload (no GPU in this container); the rank plumbing --
    launcher, process group, barrier-bracketed timed region, max over ranks, the all-gather -- is the code the GPU run goes through."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "1"
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--backend", "gloo", "--stub-workload"],
                       env=env, capture_output=True, text=True, timeout=600, cwd=str(tmp_path))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    # what the driver reads: the line is the LAST thing on stdout, short enough for any tail, strict JSON; stderr stays quiet
    assert r.stdout.rstrip().splitlines()[-1] == lines[0] and len(lines[0]) < 4096 and len(r.stderr) < 2000, (len(lines[0]), r.stderr[-2000:])
    # ... and the ONLY thing: what the collective library prints to descriptor 1 from C (gloo's connection notes here, RCCL's "Librccl path"
    # on the GPU box -- flushed at process exit, i.e. behind the line) goes to stderr (bench.py: claim_stdout)
    assert r.stdout.strip() == lines[0], r.stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["steps"] == 3 and line["warmup"] == 1 and line["data"] == "stub"
    assert line["config"]["nranks"] == 2 and line["config"]["backend"] == "gloo" and line["config"]["ranks_in_gather"] == [0, 1]
    assert line["config"]["launcher"] == "self"
    # started under the launcher by somebody else (what the driver's multi-GPU command does): no second launch, launcher reported as external
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
                        os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "0", "--backend", "gloo", "--stub-workload"],
                       env=env, capture_output=True, text=True, timeout=600, cwd=str(tmp_path))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    line = json.loads(r.stdout)                                          # nothing but the line
    assert line["config"]["launcher"] == "external" and line["config"]["nranks"] == 2
    # a world size that contradicts --gpus is refused with the way out spelled out
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--stub-workload"], env=dict(env, WORLD_SIZE="1", RANK="0"),
                       capture_output=True, text=True, timeout=300, cwd=str(tmp_path))
    assert r.returncode != 0 and "launches its own ranks" in r.stderr


def test_noise_prefetcher_draws_the_fixture_recipe_ahead(KD):
    """synth.NoisePrefetcher (sample.py --seed, --noise host): the batches of a job's schedule drawn by worker threads ahead of the sampler
    are value for value ``synth_noise(shape, seed, g, sigma_max)`` -- ragged and empty batches, slots reused, any thread count."""
    shape, seed, smax = (3, 8, 8), 11, 160.0
    plan = [torch.arange(0, 4), torch.arange(4, 8), torch.arange(8, 11), torch.arange(11, 11), torch.arange(11, 12)]
    for threads, depth in ((1, 1), (3, 2), (8, 4)):
        pf = KD.synth.NoisePrefetcher(shape, seed, smax, "cpu", depth=depth, threads=threads)
        pf.schedule(plan)
        try:
            for k, idx in enumerate(plan):
                x = pf.take(k)
                assert x.shape == (len(idx), *shape)
                for row, g in enumerate(idx):
                    assert torch.equal(x[row], KD.synth.synth_noise(shape, seed, int(g), smax)), (threads, k, row)
            with pytest.raises(KeyError):
                pf.take(len(plan))
        finally:
            pf.close()
    assert torch.equal(KD.synth.synth_noise_batch(shape, seed, 4, 3, smax), torch.stack([KD.synth.synth_noise(shape, seed, g, smax) for g in (4, 5, 6)]))


def test_indexed_rounds_and_the_schedule_callback(KD):
    """evaluation.indexed_rounds is a pure host function every rank evaluates for every rank (no index vector travels with the images), and
    compute_features_indexed tells the sample_fn its whole schedule before the first round."""
    ev = KD.evaluation
    for n, world, bs in ((21, 8, 2), (7, 2, 4), (5, 8, 3), (64, 1, 32), (1, 1, 64)):
        rounds = ev.indexed_rounds(n, world, bs)
        covered = sorted(g for width, shares in rounds for lo, cnt in shares for g in range(lo, lo + cnt))
        assert covered == list(range(n))
        assert all(cnt <= width <= bs for width, shares in rounds for _, cnt in shares)

    class One:
        num_processes, process_index, is_main_process, device = 1, 0, False, torch.device("cpu")
        gather = staticmethod(lambda t: t)
    # n = 0: no round, no call of the sample_fn, an empty result (`sample.py -n 0` writes nothing)
    assert ev.indexed_rounds(0, 4, 3) == [] and ev.compute_features_indexed(One(), lambda idx: 1 / 0, 0, 3).shape == (0,)
    seen, calls = [], []
    out = ev.compute_features_indexed(One(), lambda idx: (calls.append(idx.tolist()), idx.float()[:, None] * torch.ones(1, 2))[1], 7, 3, on_schedule=seen.append)
    assert [p.tolist() for p in seen[0]] == [[0, 1, 2], [3, 4, 5], [6]] == calls
    assert torch.equal(out, torch.arange(7.)[:, None] * torch.ones(1, 2))


def test_oracle_index_addressed_normals():
    """oracle.brownian.randn_indexed (the restatement of kd_randn_f32, the device noise source of a seeded job): N(0, 1) moments, a
    sample's values are a function of (seed, draw, element) only, draws and seeds give unrelated streams."""
    import numpy as np
    from oracle import brownian as ob
    z = ob.randn_indexed([5, 6], 200_000)
    assert z.shape == (2, 200_000) and z.dtype == np.float32
    for row in z:
        assert abs(row.mean()) < 8e-3 and abs(row.var() - 1) < 1.5e-2
        assert abs((row ** 4).mean() - 3) < 0.1 and abs((row ** 3).mean()) < 0.05
        assert abs(np.corrcoef(row[:-1], row[1:])[0, 1]) < 8e-3                  # neighbours (the cos / sin pair of a block) are uncorrelated
    assert abs(np.corrcoef(z[0], z[1])[0, 1]) < 8e-3
    assert np.array_equal(ob.randn_indexed([6], 1001)[0], z[1][:1001])           # ragged length: a prefix of the same stream
    assert np.array_equal(ob.randn_indexed([9, 5], 64)[1], z[0][:64])            # position in the batch does not matter
    d1 = ob.randn_indexed([5], 200_000, draw=1)[0]
    assert abs(np.corrcoef(d1, z[0])[0, 1]) < 8e-3 and not np.array_equal(d1, z[0])
    assert np.allclose(ob.randn_indexed([5], 64, scale=160.0)[0], 160.0 * z[0][:64], rtol=1e-6)


def test_product_host_functions_bit_equal_the_reference_hex(KD, golden):
    """The PRODUCT's host-side functions of the path -- K.sampling.get_sigmas_karras / _exponential / _polyexponential / _vp (sampling.py:17-43),
    get_ancestral_step (:51-58), Denoiser.get_scalings (layers.py:70-74), make_axial_pos (axial_rope.py:60-68), the AxialRoPE frequency ladder
    (image_transformer_v2.py:234-240) -- against the hex the REFERENCE produced (tests/golden/kat.json, oracle/make_golden.py): bit for bit,
    directly (not through the oracle), no GPU."""
    import struct
    kat = golden["kat"]
    unhex = lambda lst: torch.tensor([struct.unpack(">f", bytes.fromhex(h))[0] for h in lst], dtype=torch.float32)
    bits = lambda t: torch.as_tensor(t, dtype=torch.float32).detach().contiguous().view(torch.int32).reshape(-1)
    S = KD.sampling
    for key, hx in kat["sigmas_karras"].items():
        n, lo, hi, rho = key.split(",")
        assert torch.equal(bits(S.get_sigmas_karras(int(n), float(lo), float(hi), float(rho))), bits(unhex(hx))), key
    assert torch.equal(bits(S.get_sigmas_exponential(12, 0.01, 80)), bits(unhex(kat["sigmas_exponential"]["12,0.01,80"])))
    assert torch.equal(bits(S.get_sigmas_polyexponential(12, 0.01, 80, 2.0)), bits(unhex(kat["sigmas_polyexponential"]["12,0.01,80,2.0"])))
    assert torch.equal(bits(S.get_sigmas_vp(12)), bits(unhex(kat["sigmas_vp"]["12"])))
    assert [f"{v & 0xffffffff:08x}" for v in bits(S.get_sigmas_karras(50, 1e-2, 80))[:3].tolist()] == ["429ffffe", "42902fec", "4281bbba"]
    for key, hx in kat["ancestral_step"].items():
        a, b, eta = (float(v) for v in key.split(","))
        sd, su = S.get_ancestral_step(torch.tensor(a), torch.tensor(b), eta)
        assert torch.equal(bits(torch.stack([torch.as_tensor(sd), torch.as_tensor(su)])), bits(unhex(hx))), key
    den = KD.Denoiser(None, sigma_data=0.5)
    for s, hx in kat["scalings_sd0.5"].items():
        assert torch.equal(bits(torch.stack(den.get_scalings(torch.tensor(float(s))))), bits(unhex(hx))), s
    for key, hx in kat["axial_pos"].items():
        h, w = (int(v) for v in key.split("x"))
        assert torch.equal(bits(KD.models.axial_rope.make_axial_pos(h, w)), bits(unhex(hx))), key
    for nh, hx in kat["rope_freqs"].items():
        assert torch.equal(bits(KD.models.axial_rope.rope_freqs(32, int(nh))), bits(unhex(hx))), nh          # AxialRoPE(d_head // 2, n_heads)


def _load_bench():
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(REPO, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    argv, sys.argv = sys.argv, ["bench.py"]
    try:
        spec.loader.exec_module(mod)
    finally:
        sys.argv = argv
    return mod


def test_bench_global_attention_block_figure_from_a_recorded_kernel_table(KD):
    """bench.py's `roofline.global_attention_block` (north_star's one named efficiency figure) is derived from the per-launch event table of a pass:
    every launch at the last level's row count except the merge into / split out of it, plus that level's attention launches.  Checked here on
    the committed round-5 tables (no GPU): which kernels are picked, the per-layer arithmetic, useful vs executed fraction."""
    import json
    bench = _load_bench()
    cfg = KD.config.load_config(os.path.join(REPO, "configs", "config_oxford_flowers.json"))
    for mode, fname in (("bf16", "r05_kernel_table_bf16.json"), ("split3", "r05_kernel_table_split3.json")):
        table = json.load(open(os.path.join(REPO, "profiles", fname)))["kernels"]
        g = bench.global_attention_block(table, mode, cfg, 32, 50)
        assert g["level_width"] == 512 and g["tokens_per_sample"] == 256 and g["layers_timed"] == 4 * 50
        names = list(g["kernels_us"])
        assert all(" M=8192 " in n + " " or n.startswith(("attn_global", "attn_block")) for n in names)
        assert not any("<a1," in n or ",e3>" in n for n in names)                       # neither the merge into the level nor the split out of it
        assert abs(g["gflop_per_layer"] - 60.13) < 0.05                                   # 2 M N K of qkv / out / up / down + 4 T^2 d of the cores
        assert g["launches_per_layer"] == (4.0 if mode == "bf16" else 5.0)
        ms = sum(v["ms"] for n, v in table.items() if n in names)
        assert abs(g["us_per_layer"] - ms * 1e3 / 200) < 0.02
        assert abs(g["frac_of_bf16_mfma_peak"] - g["gflop_per_layer"] / g["us_per_layer"] * 1e3 / 2500.0) < 2e-4       # GFLOP / us = PFLOP/s
        assert g["executed_frac_of_bf16_mfma_peak"] == pytest.approx(g["frac_of_bf16_mfma_peak"] * (3 if mode == "split3" else 1), abs=2e-4)
    # a model whose last level is not global attention has no such figure
    sw = KD.config.load_config({"model": {"type": "image_transformer_v2", "input_channels": 3, "input_size": [32, 32], "patch_size": [2, 2], "depths": [2],
                                          "widths": [128], "self_attns": [{"type": "shifted-window", "d_head": 64, "window_size": 8}],
                                          "sigma_data": 0.5, "sigma_min": 1e-2, "sigma_max": 80}})
    assert bench.global_attention_block(table, "bf16", sw, 32, 50) is None


def test_bench_throttle_accumulators_and_the_reading(monkeypatch):
    """benchmarks/power_report.py (`bench.py --power`): amd-smi's throttle accumulators are parsed from its JSON whatever the nesting, a missing tool gives None, and
    the sentence it prints grades the power limiter's share of the ticks instead of asserting a cause."""
    import json
    import subprocess
    sys.path.insert(0, os.path.join(REPO, "benchmarks"))
    import power_report as bench                                   # (moved out of bench.py in round 6: opt-in `bench.py --power`)
    sample = {"gpu_data": [{"gpu": 0, "throttle": {"accumulation_counter": 1000, "prochot_accumulated": 0, "ppt_accumulated": {"value": 250, "unit": "ticks"},
                                                    "socket_thermal_accumulated": 0, "vr_thermal_accumulated": 0, "hbm_thermal_accumulated": 0,
                                                    "gfx_clk_below_host_limit_accumulated": "N/A"}}]}

    class Done:
        def __init__(self, out):
            self.stdout = out
    monkeypatch.setattr(subprocess, "run", lambda *a, **k: Done(json.dumps(sample)))
    acc = bench.throttle_accumulators()
    assert acc == {"accumulation_counter": 1000, "prochot_accumulated": 0, "ppt_accumulated": 250, "socket_thermal_accumulated": 0,
                   "vr_thermal_accumulated": 0, "hbm_thermal_accumulated": 0}
    monkeypatch.setattr(subprocess, "run", lambda *a, **k: Done("amd-smi: command not found"))
    assert bench.throttle_accumulators() is None

    def boom(*a, **k):
        raise FileNotFoundError("amd-smi")
    monkeypatch.setattr(subprocess, "run", boom)
    assert bench.throttle_accumulators() is None and bench._smi("--showpower") == ""


def test_bench_line_is_compact_and_the_long_form_goes_to_the_detail_file(tmp_path, capsys):
    """The round-5 line was 21 KB with the contract's keys at the front, and the driver's tails cut them off (BENCH_r05.parsed == null).
    bench.compact_line() turns that very result (profiles/r05_bench_line.json) into a line under 4 KiB that still carries every key the
    contract names plus roofline / cpu_baseline / parity / mode_values; emit() prints it alone on stdout and writes the long form beside it."""
    import json
    bench = _load_bench()
    big = json.load(open(os.path.join(REPO, "profiles", "r05_bench_line.json")))
    assert len(json.dumps(big)) > 15000
    detail = tmp_path / "bench_detail.json"
    bench.emit(big, str(detail))
    out = capsys.readouterr()
    assert out.err == "" and out.out.count("\n") == 1
    text = out.out.strip()
    assert len(text) <= bench.LINE_LIMIT < 4096
    line = json.loads(text)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
              "config", "roofline", "cpu_baseline", "parity", "mode_values"):
        assert k in line, k
    assert line["value"] == big["value"] and line["dtype"] == "f32" and "workload" in line["config"]
    for k in ("bound", "kernel", "avg_launch_ms", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in line["roofline"], k
    assert line["roofline"]["frac"] == big["roofline"]["frac"] and line["roofline"]["traffic"] == big["roofline"]["traffic"]
    assert {"split3", "bf16"} <= set(line["roofline"]["global_attention_block"])
    assert line["roofline"]["global_attention_block"]["bf16"]["frac_of_bf16_mfma_peak"] == big["modes"]["bf16"]["roofline"]["global_attention_block"]["frac_of_bf16_mfma_peak"]
    assert {"value", "unit", "cores", "kind"} <= set(line["cpu_baseline"]) and line["parity"]["pass"] is True
    assert line["detail_file"] == "bench_detail.json" and json.load(open(detail)) == big
    # however long the strings of a result get, the line stays under the limit (optional parts go first, never the contract's keys)
    fat = json.loads(json.dumps(big))
    fat["config"]["workload"] = "w" * 5000
    fat["cpu_baseline"]["sample"] = "s" * 5000
    fat["parity"]["case"] = "c" * 5000
    fat["mode_values"] = {f"mode{i}": 1.0 for i in range(400)}
    slim = bench.compact_line(fat, str(detail))
    assert len(json.dumps(slim)) <= bench.LINE_LIMIT and slim["value"] == big["value"] and "roofline" in slim and "cpu_baseline" in slim
    # an unwritable detail file costs nothing but the file
    bench.emit(big, str(tmp_path / "no" / "such" / "dir" / "d.json"))
    assert "detail_file" not in json.loads(capsys.readouterr().out)


def test_bench_optional_blocks_never_cost_the_line(monkeypatch):
    """bench.py runs its optional blocks (other configs, power, parity, small batches, the CLI job, the CPU baseline) behind run_guarded: a block
    that raises -- or exits, as sample.main does on a refused argument -- leaves an `error` entry under its key, the line is still printed, and
    the process's arithmetic mode is restored."""
    bench = _load_bench()
    monkeypatch.setenv("KDIFF_GEMM", "bf16")
    line = {"value": 1.0}
    bench.run_guarded(line, "ok", lambda: {"x": 1}, "split3")
    bench.run_guarded(line, "nothing", lambda: None, "split3")
    bench.run_guarded(line, "boom", lambda: 1 / 0, "split3")

    def exits():
        os.environ["KDIFF_GEMM"] = "exact"
        raise SystemExit("unknown sampler 'nope'")
    bench.run_guarded(line, "exit", exits, "split3")
    assert line["ok"] == {"x": 1} and "nothing" not in line and line["value"] == 1.0
    assert line["boom"]["error"].startswith("ZeroDivisionError") and "unknown sampler" in line["exit"]["error"]
    assert os.environ["KDIFF_GEMM"] == "split3"
    with pytest.raises(KeyboardInterrupt):                      # an interrupt is not swallowed
        bench.run_guarded(line, "int", lambda: (_ for _ in ()).throw(KeyboardInterrupt()), "split3")
