#!/usr/bin/env python3
"""MFMA-pipe throughput of the hand-written f16x3 dense-layer kernel (both variants) next to round 1's arrangement (one
library f16 GEMM over the interleaved 3x operand + the dca_act_split glue kernel) on the cube3 network's layer shapes.
TFLOP/s are ISSUED f16 MFMA flops (3 products per useful one); `useful` = /3.   python tools/gemm_bench.py [rows]"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from deepcubea_amd import _lib  # noqa: E402
from deepcubea_amd.utils.pytorch_models import _pow2_scale, _split_f16  # noqa: E402

m = int(sys.argv[1]) if len(sys.argv) > 1 else 204800


def timed(fn, iters=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


for n, k in ((1024, 1024), (1024, 5120)):
    g = torch.Generator().manual_seed(n + k)
    x = torch.randn(m, k, generator=g).cuda()
    w = torch.randn(n, k, generator=g) / k ** 0.5
    sc = _pow2_scale(w)
    wh, wl = _split_f16(w, sc)
    inv = (1.0 / sc).cuda()
    b = torch.randn(n, generator=g).cuda()
    skip = torch.randn(m, n, generator=g).cuda()
    planes = _lib.split_planes(x)
    whc, wlc = wh.cuda().contiguous(), wl.cuda().contiguous()
    w3 = torch.stack([wh, wh, wl], dim=2).reshape(n, -1).contiguous().cuda()
    a3, _ = _lib.act_split(x, None, None, 1.0, False, False)
    flops = 2.0 * m * n * k * 3
    row = {"m": m, "n": n, "k": k}
    for v in (3, 2, 1):
        _lib.f16x3_gemm_variant(v)
        ms = timed(lambda: _lib.f16x3_gemm(planes, whc, wlc, inv, 1.0, b, skip, True, True, True))
        row["hip_v%d_ms" % v] = round(ms, 4)
        row["hip_v%d_mfma_tflops" % v] = round(flops / ms / 1e9, 1)
    _lib.f16x3_gemm_variant(0)

    def lib_layer():
        y = torch.mm(a3, w3.t(), out_dtype=torch.float32)
        return _lib.act_split(y, b, skip, inv, True, True)

    ms = timed(lib_layer)
    row["library_gemm_plus_glue_ms"] = round(ms, 4)
    row["library_mfma_tflops_incl_glue"] = round(flops / ms / 1e9, 1)
    ms = timed(lambda: torch.mm(a3, w3.t(), out_dtype=torch.float32))
    row["library_gemm_only_ms"] = round(ms, 4)
    row["library_gemm_only_mfma_tflops"] = round(flops / ms / 1e9, 1)
    print(json.dumps(row))
    del x, planes, a3, skip
    torch.cuda.empty_cache()

# ---- the 16-bit (non-parity) layer: dca_gemm16 with its tail in the epilogue vs the library's addmm_activation (+ the
# separate ReLU pass a residual layer needs there)
for dt, nm in ((torch.bfloat16, "bf16"), (torch.float16, "fp16")):
    for n, k in ((1024, 1024), (1024, 5120)):
        g = torch.Generator().manual_seed(n + k)
        x = (torch.randn(m, k, generator=g) * 0.5).to(dt).cuda()
        w = (torch.randn(n, k, generator=g) / k ** 0.5).to(dt).cuda()
        b32 = torch.randn(n, generator=g).cuda()
        bdt = b32.to(dt)
        skip = torch.randn(m, n, generator=g).to(dt).cuda()
        flops = 2.0 * m * n * k
        row = {"dtype": nm, "m": m, "n": n, "k": k}
        out = skip.clone()
        for v, tag in ((1, "hip_two_stage"), (2, "hip")):
            _lib.gemm16_variant(v)
            ms = timed(lambda: _lib.gemm16(x, w, b32, None, True))
            row[tag + "_bias_relu_ms"], row[tag + "_bias_relu_tflops"] = round(ms, 4), round(flops / ms / 1e9, 1)
            ms = timed(lambda: _lib.gemm16(x, w, None, out, True, out=out))
            row[tag + "_skip_relu_ms"], row[tag + "_skip_relu_tflops"] = round(ms, 4), round(flops / ms / 1e9, 1)
        ms = timed(lambda: torch._addmm_activation(bdt, x, w.t()))
        row["library_bias_relu_ms"], row["library_bias_relu_tflops"] = round(ms, 4), round(flops / ms / 1e9, 1)
        out2 = skip.clone()
        ms = timed(lambda: out2.addmm_(x, w.t()).relu_())
        row["library_skip_relu_ms"], row["library_skip_relu_tflops"] = round(ms, 4), round(flops / ms / 1e9, 1)
        print(json.dumps(row))
        del x, skip, out, out2
        torch.cuda.empty_cache()
