"""Timing helpers shared by the calibration / check tools.  graph_us() is the one to trust for kernels shorter than ~15 us: it replays a HIP graph of `n` launches, so
nothing but the GPU is between two launches (a ctypes call costs ~9 us, and launches spaced by it leave the part in a different clock state than back-to-back work:
profiles/native_r3_ring_vs_pipelined_instream.txt)."""
import torch


def graph_us(call, n=40, warm_replays=3, timed_replays=4, repeats=2):
    """Microseconds per launch of `call` (a closure that launches on the current stream and allocates only through torch's caching allocator)."""
    call(); torch.cuda.synchronize()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(2): call()
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for _ in range(n): call()
    for _ in range(warm_replays): gr.replay()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(repeats):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(timed_replays): gr.replay()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / (n * timed_replays))
    del gr
    return best


def stream_us(call, reps):
    """The round-3 calibration protocol: `reps` ctypes calls in the stream between two events (Python-bound below ~10 us per call)."""
    for _ in range(max(3, reps // 4)): call()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(2):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): call()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / reps)
    return best
