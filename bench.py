#!/usr/bin/env python3
"""Headline benchmark: MXFP4 GEMM 4096x4096x4096 (BASELINE.json configs[1]) on N MI355X GPUs.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

A "step" is one matmul_mxf4_bf16_tn over pre-quantised, HBM-resident operands ("ideal" provider of the
reference's own harness, benchmarks/bench_mxfp4_sm120.py:27-31,109-127).  The path is a per-layer dense
contraction with no exchange step, so ranks are independent replicas over the same shapes: `value` is
the sum of the ranks' work over the max-over-ranks time ("scaling": "weak", no data-path collective).

Prints ONE JSON line on rank 0 with `roofline` (MFMA-bound: FLOP/s of the dominant kernel from HIP
events on the launch stream vs the FP4 dense peak of MI355X_MICROARCH.md) and `cpu_baseline` (the
reference's dequantise + torch.matmul oracle path timed on this host's cores, bounded sample).

Round 3: the same line also carries (i) `configs` -- the other BASELINE.json configs (C3 GEMM / C3 step / C4 / C5 TN / C5 NN), each
timed with HIP events on the launch stream after the headline's timed region, with its own roofline fraction; (ii) `power` -- socket
power and shader clock sampled through librocm_smi64 on a host thread WHILE the timed region runs (the headline kernel sits at the
socket power limit: the clock it gets is part of the result); (iii) `roofline.traffic` measured FRESH: after the timed region the
script runs its own hot loop twice more under `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` (separate passes, as
MI355X_MICROARCH.md prescribes; child processes of this run, same box, same build) -- null with the reason if rocprofv3 is missing or
fails, never a replay of an earlier run.

Timing protocol.  `value` / `ms_per_step` follow the driver's contract: EXACTLY K back-to-back steps between two
barrier + synchronize pairs, wall clock, max over ranks.  Next to it the reference's own protocol
(benchmarks/bench_mxfp4_sm120.py:109-125: warm-up, >= 200 individually timed repetitions, median with the 20th / 80th
percentile) runs as a SECOND, separate pass after the timed region -- one HIP event pair per launch -- and is reported as
`per_launch_us`; the event records cost a few hundred ns each, which is why they stay out of the K-step region.
"""
from __future__ import annotations

import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

FP4_DENSE_PEAK_TFLOPS = 10066.0  # 256 CU x 4 SIMD x 2048 MAC/clk x 2 x 2.4 GHz (MI355X_MICROARCH.md: ~10 PF dense)
# MFMA-only loop (no memory traffic) with RANDOM fp4 operands, measured on this part: the power limit holds the
# clock near 1.6 GHz (profiles/ubench_r1f_const_vs_random_operands.log).  Informational only.
FP4_SUSTAINED_RANDOM_TFLOPS = 6550.0
# HBM-side bytes per launch come from two rocprofv3 --pmc child runs of this script's own hot loop (fresh_traffic below).
# QAMD_PMC_TRAFFIC_JSON (set by tools/pmc_bench.sh for its final, un-profiled run) may point at counters that script took
# minutes earlier on the same box and build instead; a committed file of an earlier run is never quoted.
PMC_TRAFFIC_ENV = "QAMD_PMC_TRAFFIC_JSON"
M = N = K = 4096
CPU_ROWS = 4096  # cpu_baseline sample: the whole workload (measured 3.2 s per 1024 rows on the 256-thread host)


def hadamard(n, device):
    h = torch.ones(1, 1)
    while h.shape[0] < n:
        h = torch.cat([torch.cat([h, h], 1), torch.cat([h, -h], 1)], 0)
    return (h * n ** -0.5).to(torch.bfloat16).to(device)


def cpu_baseline(a_q, a_s, b_q, b_s, rows):
    """Reference oracle path (tests/mxfp4_test.py:84-120,229-231) restated in oracle/dequant_matmul.py: dequantise both
    packed operands (code -> value table x 2^(e8m0 - 127)) and a_dq @ b_dq.T -> bf16, on a `rows`-row slab of A against
    all of B, on the host cores.  fp64 is exactly what the reference's test does; fp32 is the cheaper variant BASELINE.md
    section 3 asks for next to it.  Dequantisation and matmul are timed separately."""
    from oracle import dequant_matmul as dm

    torch.set_num_threads(os.cpu_count() or 1)
    a_q, a_s = a_q[:rows].cpu(), a_s[:rows].cpu()
    b_q, b_s = b_q.cpu(), b_s.cpu()
    flops = 2.0 * rows * N * K
    res, out64 = {}, None
    for name, dt_ in (("fp64", torch.float64), ("fp32", torch.float32)):
        t0 = time.perf_counter()
        a = dm.dq_fp4(a_q, a_s, 32, dt_)
        b = dm.dq_fp4(b_q, b_s, 32, dt_)
        t1 = time.perf_counter()
        out = (a @ b.T).to(torch.bfloat16)
        t2 = time.perf_counter()
        res[name] = {"TFLOP/s": round(flops / (t2 - t0) / 1e12, 4), "dequant_s": round(t1 - t0, 3), "matmul_s": round(t2 - t1, 3),
                     "matmul_only_TFLOP/s": round(flops / (t2 - t1) / 1e12, 4)}
        if name == "fp64":
            out64 = out
        else:
            res[name]["bf16_equal_to_fp64_path"] = bool(torch.equal(out, out64))
        del a, b
    return {
        "value": res["fp64"]["TFLOP/s"],
        "unit": "TFLOP/s",
        "cores": torch.get_num_threads(),
        "os_cpu_count": os.cpu_count(),
        "kind": "port",
        "sample": f"dequantise(A[:{rows}], B) + torch.matmul -> bf16 on a {rows}x{N}x{K} slab of the same operands; value = the fp64 variant "
                  f"(what tests/mxfp4_test.py does), dequant + matmul, {res['fp64']['dequant_s'] + res['fp64']['matmul_s']:.2f} s",
        "fp64": res["fp64"],
        "fp32": res["fp32"],
    }, out64


# ---- socket power / shader clock while the timed region runs (librocm_smi64 through ctypes, host thread) ------------------------
class _RsmiFreq(ctypes.Structure):   # rsmi_frequencies_t (rocm_smi.h)
    _fields_ = [("has_deep_sleep", ctypes.c_bool), ("num_supported", ctypes.c_uint32), ("current", ctypes.c_uint32), ("frequency", ctypes.c_uint64 * 33)]


class PowerSampler:
    """rsmi_dev_power_get + rsmi_dev_gpu_clk_freq_get(RSMI_CLK_TYPE_SYS) every ~2 ms on a daemon thread (the firmware refreshes its
    table every ~12 ms).  `mark()` timestamps phase boundaries; `window(a, b)` averages the samples between two marks."""

    def __init__(self, dev_index: int = 0):
        import threading

        self.ok, self.samples, self.marks, self._stop, self._dev = False, [], {}, threading.Event(), dev_index
        self.t0 = time.perf_counter()
        try:
            self.lib = ctypes.CDLL("librocm_smi64.so")
            self.ok = self.lib.rsmi_init(ctypes.c_uint64(0)) == 0
        except OSError:
            self.lib = None
        if self.ok:
            self._th = threading.Thread(target=self._run, daemon=True)
            self._th.start()

    def _run(self):
        uw, pt, fr = ctypes.c_uint64(0), ctypes.c_int(0), _RsmiFreq()
        while not self._stop.is_set():
            w = mhz = None
            try:
                if self.lib.rsmi_dev_power_get(ctypes.c_uint32(self._dev), ctypes.byref(uw), ctypes.byref(pt)) == 0:
                    w = uw.value * 1e-6
            except AttributeError:   # older librocm_smi64: average socket power
                if self.lib.rsmi_dev_power_ave_get(ctypes.c_uint32(self._dev), ctypes.c_uint32(0), ctypes.byref(uw)) == 0:
                    w = uw.value * 1e-6
            if self.lib.rsmi_dev_gpu_clk_freq_get(ctypes.c_uint32(self._dev), ctypes.c_int(0), ctypes.byref(fr)) == 0 and fr.current < 33:
                mhz = fr.frequency[fr.current] * 1e-6
            self.samples.append((time.perf_counter() - self.t0, w, mhz))
            time.sleep(0.002)

    def mark(self, name):
        self.marks[name] = time.perf_counter() - self.t0

    def window(self, a, b):
        ta, tb = self.marks[a], self.marks[b]
        ws = [w for t, w, _ in self.samples if ta <= t <= tb and w]
        fs = [f for t, _, f in self.samples if ta <= t <= tb and f]
        return {"power_w": round(sum(ws) / len(ws), 1) if ws else None, "sclk_mhz": round(sum(fs) / len(fs), 0) if fs else None,
                "power_w_max": round(max(ws), 1) if ws else None, "samples": len(ws), "window_ms": round((tb - ta) * 1e3, 1)}

    def stop(self):
        self._stop.set()
        if self.ok:
            self._th.join(timeout=1.0)
            try:
                self.lib.rsmi_shut_down()
            except Exception:
                pass


def event_us(fn, iters, warm=5, min_ms=30.0, sampler=None, tag=None, timed_ms=60.0):
    """average device microseconds per call: back-to-back launches bracketed by HIP events on the current stream, after `warm` calls
    and at least `min_ms` of the same load (clock ramp).  With a PowerSampler the timed loop is stretched to >= `timed_ms` (the firmware
    refreshes its power / clock table every ~12 ms) and bracketed by marks `tag`:0 / `tag`:1."""
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0, n = time.perf_counter(), 0
    while (time.perf_counter() - t0) * 1e3 < min_ms:
        for _ in range(10):
            fn()
        torch.cuda.synchronize()
        n += 10
    if sampler is not None and tag:
        iters = max(iters, int(timed_ms * 1e-3 / max((time.perf_counter() - t0) / n, 1e-7)) + 1)
        sampler.mark(tag + ":0")
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    if sampler is not None and tag:
        sampler.mark(tag + ":1")
    return e0.elapsed_time(e1) * 1e3 / iters


def graph_us(fn, sampler=None, tag=None, timed_ms=250.0, warm_ms=40.0):
    """GPU-only microseconds per call for the SIDE configs: `fn` is captured `n` times into a HIP graph (n calls ~ 5 ms) and the graph is replayed for >=
    `timed_ms` between two events -- nothing but the GPU sits between two launches, so a path of two or three short launches is not charged the host's
    op-dispatch bubbles (C3_step_blocked read 139 vs 125 us through the Python loop on one box, 123 vs 124 on others), and the window is
    long enough for the firmware's ~12 ms power / clock table to mean something (weak #9).  Falls back to event_us when capture fails."""
    try:
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        one = max(e0.elapsed_time(e1) * 1e3, 1.0)
        n = max(2, min(200, int(5000.0 / one)))
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            fn(); fn()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            for _ in range(n):
                fn()
        reps_w = max(1, int(warm_ms * 1e3 / (one * n)))
        for _ in range(reps_w):
            gr.replay()
        torch.cuda.synchronize()
        reps = max(2, int(timed_ms * 1e3 / (one * n)) + 1)
        if sampler is not None and tag:
            sampler.mark(tag + ":0")
        e0.record()
        for _ in range(reps):
            gr.replay()
        e1.record(); torch.cuda.synchronize()
        if sampler is not None and tag:
            sampler.mark(tag + ":1")
        us = e0.elapsed_time(e1) * 1e3 / (reps * n)
        del gr
        return us, "hip-graph replays, >= %d ms" % int(timed_ms)
    except Exception as e:   # capture refused (an op that allocates outside the caching allocator): the stream protocol
        return event_us(fn, 100, sampler=sampler, tag=tag, timed_ms=timed_ms), "events around a Python loop (graph capture failed: %s)" % type(e).__name__


def side_configs(q, dev, h32, alpha, sampler=None):
    """BASELINE.json configs[2..4] at full size (synthetic operands, resident in HBM; parity of every one of them is what
    tests/test_gpu_baseline_configs.py checks).  Peaks: MI355X_MICROARCH.md dense figures for the MFMA the path computes on --
    FP4 10066, FP8 5033, f16 2516 TFLOP/s (NVFP4 keeps the reference's exact e4m3-per-16 semantics on the f16 MFMA)."""
    from qutlass_amd.utils import to_blocked

    out = {}

    def put(name, fn, iters, flops, peak, warm=5, **extra):
        us, how = graph_us(fn, sampler=sampler if (sampler is not None and sampler.ok) else None, tag=name)
        tf = flops / us * 1e-6
        out[name] = {"us": round(us, 2), "TFLOP/s": round(tf, 1), "roofline": {"bound": "mfma", "achieved": round(tf, 1), "peak": peak, "unit": "TFLOP/s",
                                                                              "frac": round(tf / peak, 4)}, "timing": how, **extra}
        if sampler is not None and sampler.ok:   # socket power / shader clock while this config ran (which of them sit at the power limit)
            w = sampler.window(name + ":0", name + ":1")
            out[name]["power_w"], out[name]["sclk_mhz"] = w["power_w"], w["sclk_mhz"]
            if w["power_w"]:   # [r6] joules per step: the launch sits at the socket power limit, so energy is what sets its time (profiles/energy_ubench_r6p.txt)
                out[name]["energy_mj"] = round(w["power_w"] * us * 1e-3, 2)

    # C3: fusedQuantizeMx(H32, abs_max) + MXFP4 GEMM, Llama-3-8B FFN M=4096 N=14336 K=4096
    m, n, k = 4096, 14336, 4096
    x = torch.randn(m, k, dtype=torch.bfloat16, device=dev) * 25.0
    w = torch.randn(n, k, dtype=torch.bfloat16, device=dev) * 25.0
    w_q, w_s = q.fusedQuantizeMx(w, h32, method="abs_max")
    w_sf = to_blocked(w_s)
    x_q, x_s = q.fusedQuantizeMx(x, h32, method="abs_max")
    x_sf = to_blocked(x_s)
    fl = 2.0 * m * n * k
    put("C3_gemm", lambda: q.matmul_mxf4_bf16_tn(x_q, w_q, x_sf, w_sf, alpha), 100, fl, FP4_DENSE_PEAK_TFLOPS, workload="matmul_mxf4_bf16_tn 4096x14336x4096")

    def step3():   # the reference's activation path: three launches (qutlass/__init__.py:149-180, utils.py:160-193)
        a_q, a_s = q.fusedQuantizeMx(x, h32, method="abs_max")
        return q.matmul_mxf4_bf16_tn(a_q, w_q, to_blocked(a_s), w_sf, alpha)

    def step2():   # quantizer with GEMM-ready scales: two launches
        a_q, a_sb = q.fusedQuantizeMxBlocked(x, h32, method="abs_max")
        return q.matmul_mxf4_bf16_tn(a_q, w_q, a_sb, w_sf, alpha)

    put("C3_step", step3, 100, fl, FP4_DENSE_PEAK_TFLOPS, workload="fusedQuantizeMx(H32, abs_max) + to_blocked + GEMM, weights pre-quantised (3 launches)")
    put("C3_step_blocked", step2, 100, fl, FP4_DENSE_PEAK_TFLOPS, workload="fusedQuantizeMxBlocked(H32, abs_max) + GEMM (2 launches; extension)")
    del x, w, w_q, w_s, x_q, x_s, w_sf, x_sf
    # C4: NVFP4 8192^3
    m = n = k = 8192
    gs = torch.tensor([1.0], device=dev)
    h16 = hadamard(16, dev)
    a = torch.randn(m, k, dtype=torch.bfloat16, device=dev) * 25.0
    a_q, a_s = q.fusedQuantizeNv(a, h16, gs)
    a.normal_()
    a *= 25.0
    b_q, b_s = q.fusedQuantizeNv(a, h16, gs)
    del a
    a_sf, b_sf = to_blocked(a_s), to_blocked(b_s)
    put("C4", lambda: q.matmul_nvf4_bf16_tn(a_q, b_q, a_sf, b_sf, alpha), 40, 2.0 * m * n * k, 2516.0, warm=3,
        workload="matmul_nvf4_bf16_tn 8192^3 (exact e4m3-per-16 semantics on the f16 MFMA: peak = the 16-bit dense peak)")
    del a_q, b_q, a_s, b_s, a_sf, b_sf
    # C5: MXFP8 4096^3 TN and NN
    m = n = k = 4096
    a8 = (torch.randn(m, k, device=dev) * 4).to(torch.float8_e4m3fn)
    b8 = (torch.randn(n, k, device=dev) * 4).to(torch.float8_e4m3fn)
    s8a = to_blocked(torch.randint(120, 131, (m, k // 32), dtype=torch.uint8, device=dev).view(torch.float8_e8m0fnu))
    s8b = to_blocked(torch.randint(120, 131, (n, k // 32), dtype=torch.uint8, device=dev).view(torch.float8_e8m0fnu))
    a8t = a8.view(torch.uint8).T.contiguous().view(torch.float8_e4m3fn)
    put("C5_tn", lambda: q.matmul_mxf8_bf16_tn(a8, b8, s8a, s8b, alpha), 200, 2.0 * m * n * k, 5033.0, workload="matmul_mxf8_bf16_tn 4096^3")
    put("C5_nn", lambda: q.matmul_mxf8_bf16_nn(a8t, b8, s8a, s8b, alpha), 200, 2.0 * m * n * k, 5033.0, workload="matmul_mxf8_bf16_nn 4096^3 (A stored (K, M))")
    return out


def fresh_traffic():
    """HBM-side bytes per launch of the headline kernel: two `rocprofv3 --pmc` child runs (FETCH_SIZE, WRITE_SIZE: separate passes)
    of this script's own hot loop (`--pmc-child`), on this box, right now.  Returns (dict | None, note)."""
    import glob
    import importlib.util
    import shutil
    import sqlite3
    import subprocess
    import tempfile

    rocprof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(rocprof):
        return None, "rocprofv3 not found"
    spec = importlib.util.spec_from_file_location("_rps", os.path.join(ROOT, "tools", "rocprof_summary.py"))
    rps = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(rps)
    tmp = tempfile.mkdtemp(prefix="qamd_pmc_", dir="/tmp")
    vals = {}
    try:
        for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
            cmd = [rocprof, "--pmc", ctr, "-d", os.path.join(tmp, ctr), "-o", "p", "--", sys.executable, os.path.abspath(__file__), "--pmc-child", "30"]
            r = subprocess.run(cmd, cwd="/tmp", env={**os.environ, "TMPDIR": "/tmp"}, capture_output=True, text=True, timeout=240)
            dbs = glob.glob(os.path.join(tmp, ctr, "**", "*results.db"), recursive=True)
            if r.returncode != 0 or not dbs:
                return None, f"rocprofv3 --pmc {ctr} failed (rc {r.returncode}): {(r.stderr or r.stdout)[-300:]}"
            for kname, cname, cnt, avg in rps.pmc_rows(sqlite3.connect(dbs[0])):
                if "gemm_mx_" in kname and cname == ctr:
                    vals[ctr] = (avg * 1024.0, cnt, kname)
        if len(vals) != 2:
            return None, f"counters missing from the rocprofv3 output ({sorted(vals)})"
        f, w = vals["FETCH_SIZE"][0], vals["WRITE_SIZE"][0]
        return {"traffic_bytes": 2 * f + w, "fetch_size_raw_bytes": f, "fetch_bytes_corrected_x2": 2 * f, "write_bytes": w,
                "kernel": vals["FETCH_SIZE"][2], "dispatches": vals["FETCH_SIZE"][1]}, "ok"
    except Exception as e:   # noqa: BLE001 -- the profile is optional evidence, never a reason to lose the bench line
        return None, f"{type(e).__name__}: {e}"
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def dominant_kernel_name():
    """The kernel the product library's dispatch picks for the headline shape, from its dry-run hook (no GPU touched)."""
    import ctypes

    from qutlass_amd import _lib

    f = _lib.load().qutlass_amd_debug_gemm_plan
    f.restype = ctypes.c_int
    f.argtypes = [ctypes.c_int] + [ctypes.c_int64] * 4 + [ctypes.POINTER(ctypes.c_int), ctypes.c_int]
    out = (ctypes.c_int * 24)()
    cnt = f(4, M, N, K, 0, out, 8)
    names = {90: "gemm_mx_deepp_kernel<GemmCfg<256,256,2,2,4>> (persistent deep schedule)", 30: "gemm_mx_kernel<GemmCfg<256,256,2,2,4>, SCHED_DEEP>"}
    plan = [int(out[3 * i]) for i in range(max(cnt, 0))]
    return names.get(plan[0], f"gemm variant {plan[0]}") if plan else "unknown", plan


def percentile(sorted_vals, p):
    if not sorted_vals:
        return None
    i = min(len(sorted_vals) - 1, max(0, int(round(p * (len(sorted_vals) - 1)))))
    return sorted_vals[i]


def max_over_ranks(wall: float, device=None) -> float:
    """The job's step time is the slowest rank's (replicas: no data-path collective, only this MAX-reduce and the
    barriers around the timed region go through torch.distributed -- RCCL on GPUs, gloo in the CPU test)."""
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return wall
    tw = torch.tensor([wall], dtype=torch.float64, device=device if device is not None else "cpu")
    dist.all_reduce(tw, op=dist.ReduceOp.MAX)
    return float(tw.item())


def aggregate_value(flop_per_step: float, steps: int, world: int, wall: float) -> float:
    """Whole-job TFLOP/s: every rank runs the same workload (weak scaling, independent replicas)."""
    return flop_per_step * steps * world / wall / 1e12


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-configs", action="store_true", help="skip the C3 / C4 / C5 side measurements")
    ap.add_argument("--no-pmc", action="store_true", help="skip the two rocprofv3 --pmc child runs (roofline.traffic = null)")
    ap.add_argument("--pmc-child", type=int, default=0, metavar="N",
                    help="internal: set the headline operands up, launch the headline op N times and exit (the command the --pmc passes profile)")
    ap.add_argument("--ramp-ms", type=float, default=250.0,
                    help="untimed load before the W warmup steps: an idle MI355X needs ~50 ms under load to leave its clock "
                         "ramp (tools/clock_ramp.py: 54 -> 42 -> 38 -> 36.7 us/step over the first 40 ms)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", 0))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    if args.gpus > 1 or world > 1:
        import torch.distributed as dist

        dist.init_process_group("nccl")  # RCCL; only used for the timing barrier / max-reduce
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    import qutlass_amd
    from qutlass_amd.utils import to_blocked

    # ---- synthetic operands (seed 0, test distribution randn*25, mxfp4_test.py:224-225) ----
    torch.manual_seed(0)
    a = torch.randn(M, K, dtype=torch.bfloat16, device=dev) * 25.0
    b = torch.randn(N, K, dtype=torch.bfloat16, device=dev) * 25.0
    h = hadamard(32, dev)
    alpha = torch.tensor([1.0], device=dev)
    a_q, a_s = qutlass_amd.fusedQuantizeMx(a, h, method="abs_max")
    b_q, b_s = qutlass_amd.fusedQuantizeMx(b, h, method="abs_max")
    a_sf, b_sf = to_blocked(a_s), to_blocked(b_s)
    torch.cuda.synchronize()

    def step():
        return qutlass_amd.matmul_mxf4_bf16_tn(a_q, b_q, a_sf, b_sf, alpha)

    if args.pmc_child > 0:   # profiled child of fresh_traffic(): the hot loop only
        for _ in range(args.pmc_child):
            step()
        torch.cuda.synchronize()
        return
    sampler = PowerSampler(local_rank) if rank == 0 else None

    # ---- clock ramp (untimed, not part of the W warmup steps): bring the part to its steady clock / power state ----
    ramp_steps = 0
    t_ramp = time.perf_counter()
    while (time.perf_counter() - t_ramp) * 1e3 < args.ramp_ms:
        for _ in range(200):
            out = step()
        torch.cuda.synchronize()
        ramp_steps += 200

    for _ in range(args.warmup):
        out = step()

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            import torch.distributed as dist

            dist.barrier()
        torch.cuda.synchronize()

    # ---- timed region: EXACTLY `steps` steps, barrier + synchronize on both sides -----------
    stream = torch.cuda.current_stream(dev)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    barrier()
    if sampler:
        sampler.mark("timed_start")
    t0 = time.perf_counter()
    ev0.record(stream)  # the ops launch on torch's current stream, so these events bracket the kernels
    for _ in range(args.steps):
        out = step()
    ev1.record(stream)
    barrier()
    wall = time.perf_counter() - t0
    if sampler:
        sampler.mark("timed_end")
    kernel_ms = ev0.elapsed_time(ev1) / args.steps  # average launch-to-launch duration of the GEMM kernel

    wall = max_over_ranks(wall, dev if world > 1 else None)

    # ---- the reference's protocol as a separate pass: individually timed launches, median / p20 / p80 ----------------
    nrep = max(200, min(args.steps, 1000))
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(nrep + 1)]
    for _ in range(25):
        out = step()
    evs[0].record(stream)
    for i in range(nrep):
        out = step()
        evs[i + 1].record(stream)
    torch.cuda.synchronize()
    per = sorted(evs[i].elapsed_time(evs[i + 1]) * 1e3 for i in range(nrep))
    if sampler:
        sampler.mark("per_launch_end")
    per_launch = {"n": nrep, "median": round(percentile(per, 0.5), 3), "p20": round(percentile(per, 0.2), 3), "p80": round(percentile(per, 0.8), 3),
                  "min": round(per[0], 3), "note": "one HIP event pair per launch, after the timed region (bench_mxfp4_sm120.py:109-125 protocol)"}

    kname, kplan = dominant_kernel_name()
    flop_per_step = 2.0 * M * N * K
    value = aggregate_value(flop_per_step, args.steps, world, wall)
    achieved = flop_per_step / (kernel_ms * 1e-3) / 1e12

    result = {
        "metric": "TFLOP/s & %FP4-MFMA-peak, MXFP4 GEMM 4096x4096x4096, 1 MI355X",
        "value": round(value, 2),
        "unit": "TFLOP/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(wall / args.steps * 1e3, 5),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "fp4 (e2m1 x e2m1, e8m0 block scales, fp32 accumulate, bf16 out)",
        "data": "synthetic",
        "config": {
            "workload": "matmul_mxf4_bf16_tn 4096x4096x4096, gs=32 e8m0 scales (BASELINE.json configs[1])",
            "operands": "randn*25 bf16, seed 0, fusedQuantizeMx(H32, abs_max) once, to_blocked scales, alpha=1",
            "parallelism": f"replicas x{world}" if world > 1 else "single GPU",
            "clock_ramp": f"{ramp_steps} untimed steps (~{args.ramp_ms:.0f} ms) before the {args.warmup} warmup steps",
            "pct_of_fp4_peak": round(100.0 * value / world / FP4_DENSE_PEAK_TFLOPS, 2),
            # `value` is the wall clock of the timed region: barrier + synchronize, K launches, synchronize + barrier.  With few steps the first launch after the barrier
            # and the final synchronize add a fixed ~40 us (7 % at K = 20); the launch-to-launch duration of the kernel itself is roofline.kernel_us
            "timed_region": "wall clock around the K steps incl. the first launch after the barrier and the final synchronize (a fixed ~40 us: 7 % at 20 steps); "
                            "kernel-only duration: roofline.kernel_us",
            "wall_minus_kernel_us_per_step": round((wall / args.steps - kernel_ms * 1e-3) * 1e6, 3),
        },
        "roofline": {
            "bound": "mfma",
            "achieved": round(achieved, 2),
            "peak": FP4_DENSE_PEAK_TFLOPS,
            "unit": "TFLOP/s",
            "frac": round(achieved / FP4_DENSE_PEAK_TFLOPS, 4),
            "traffic": None,
            "kernel": kname,
            "dispatch_plan": kplan,
            "kernel_us": round(kernel_ms * 1e3, 3),
            "per_launch_us": per_launch,
            "algorithmic_flop_per_launch": flop_per_step,
            "algorithmic_bytes_per_launch": M * K // 2 + N * K // 2 + (M + N) * K // 32 + 2 * M * N,
            "frac_of_sustained_random_operand_mfma_rate": round(achieved / FP4_SUSTAINED_RANDOM_TFLOPS, 4),
        },
    }
    # socket power / shader clock during the timed region (and over the longer window that also holds the per-launch pass: the
    # firmware refreshes its table every ~12 ms, the K-step region of the default run lasts ~70 ms)
    if sampler:
        if sampler.ok and sampler.samples:
            result["power"] = {"timed_region": sampler.window("timed_start", "timed_end"), "timed_region_plus_per_launch_pass": sampler.window("timed_start", "per_launch_end"),
                               "source": "librocm_smi64 rsmi_dev_power_get / rsmi_dev_gpu_clk_freq_get(SYS), host thread, 2 ms period"}
            # [r6] joules per launch = mean socket power x the kernel's duration (the wider window when the K-step region was too short for a sample)
            pw = result["power"]["timed_region"]["power_w"] or result["power"]["timed_region_plus_per_launch_pass"]["power_w"]
            if pw:
                result["roofline"]["energy_mj_per_launch"] = round(pw * kernel_ms, 2)
                result["roofline"]["energy_note"] = ("socket W x kernel_us; ~0.30 kW of it is the idle socket.  Priced piece by piece in profiles/energy_ubench_r6p.txt: "
                                                     "20.3 mJ for the MFMAs of 137.4 GFLOP on this data + 10-13 mJ of operand fill above idle")
        else:
            result["power"] = None
    # [r5] the same kernel, same shape, ALL-ZERO operands under unit scales (nothing toggles in the matrix pipe, the clock stays up): what the SCHEDULE delivers without the
    # socket's power cap.  A reported side figure, after the timed region, never `value` / `roofline.frac` (tools/power_data_probe.py has the longer form).
    if rank == 0 and world == 1 and not args.no_configs:
        try:
            zq_a, zq_b = torch.zeros_like(a_q), torch.zeros_like(b_q)
            one_a = torch.full_like(a_sf.view(torch.uint8), 127).view(a_sf.dtype)
            one_b = torch.full_like(b_sf.view(torch.uint8), 127).view(b_sf.dtype)
            zus, zhow = graph_us(lambda: qutlass_amd.matmul_mxf4_bf16_tn(zq_a, zq_b, one_a, one_b, alpha), sampler=sampler if (sampler and sampler.ok) else None, tag="zero_operands")
            ztf = flop_per_step / zus * 1e-6
            result["roofline"]["same_kernel_on_zero_operands"] = {"kernel_us": round(zus, 3), "achieved": round(ztf, 2), "frac": round(ztf / FP4_DENSE_PEAK_TFLOPS, 4), "timing": zhow,
                                                                   "note": "all-zero e2m1 codes, every scale 2^0: identical instruction stream, no data-dependent power -- the schedule at the clock the "
                                                                           "peak is quoted for; the gap to roofline.frac is the socket power cap under the bench operands"}
            if sampler and sampler.ok:
                zw = sampler.window("zero_operands:0", "zero_operands:1")
                result["roofline"]["same_kernel_on_zero_operands"].update({"power_w": zw["power_w"], "sclk_mhz": zw["sclk_mhz"]})
                if zw["power_w"]:
                    result["roofline"]["same_kernel_on_zero_operands"]["energy_mj_per_launch"] = round(zw["power_w"] * zus * 1e-3, 2)
            del zq_a, zq_b, one_a, one_b
        except Exception as e:   # noqa: BLE001 -- a side measurement must not cost the headline line
            result["roofline"]["same_kernel_on_zero_operands"] = {"error": f"{type(e).__name__}: {e}"}
    # HBM-side bytes per launch of this kernel: fresh PMC passes (see fresh_traffic), rank 0 of a single-GPU run only
    if rank == 0 and world == 1:
        tj, note = None, "skipped (--no-pmc)"
        fresh = os.environ.get(PMC_TRAFFIC_ENV)
        if fresh:
            try:
                with open(fresh) as f:
                    tj = json.load(f)
                note = "counters taken by tools/pmc_bench.sh minutes earlier on this box: " + os.path.relpath(fresh, ROOT)
            except (OSError, ValueError) as e:
                tj, note = None, f"{fresh}: {e}"
        elif not args.no_pmc:
            tj, note = fresh_traffic()
            if tj:
                note = "2 rocprofv3 --pmc child runs (FETCH_SIZE, WRITE_SIZE) of this script's hot loop, this box, after the timed region"
        if tj and tj.get("traffic_bytes"):
            result["roofline"]["traffic"] = tj["traffic_bytes"]
            result["roofline"]["traffic_detail"] = {k_: tj.get(k_) for k_ in ("fetch_size_raw_bytes", "fetch_bytes_corrected_x2", "write_bytes", "dispatches")}
            result["roofline"]["traffic_kernel"] = tj.get("kernel")
            result["roofline"]["traffic_over_algorithmic"] = round(tj["traffic_bytes"] / result["roofline"]["algorithmic_bytes_per_launch"], 3)
        result["roofline"]["traffic_source"] = note
        if not args.no_configs:
            try:
                result["configs"] = side_configs(qutlass_amd, dev, h, alpha, sampler)
            except Exception as e:   # noqa: BLE001 -- side measurements must not cost the headline line
                result["configs"] = {"error": f"{type(e).__name__}: {e}"}
    if sampler:
        sampler.stop()

    if rank == 0:
        if not args.no_cpu_baseline and world == 1:
            cb, ref = cpu_baseline(a_q, a_s, b_q, b_s, rows=CPU_ROWS)
            result["cpu_baseline"] = cb
            # parity of the measured op against the same slab (exact bf16 equality, as the reference asserts)
            got = out[:CPU_ROWS].cpu()
            result["config"]["parity_vs_cpu_oracle_slab"] = bool(torch.equal(got, ref))
        else:
            result["cpu_baseline"] = None
        print(json.dumps(result), flush=True)
    if world > 1 or args.gpus > 1:
        import torch.distributed as dist

        if dist.is_initialized():
            dist.destroy_process_group()


if __name__ == "__main__":
    main()
