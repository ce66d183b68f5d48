"""Can the weight-gradient GEMM (MFMA-bound) and the transposed aggregation (HBM-bound) of the collab step share the GPU?
Times them back to back on one stream, concurrently on two plain streams, and concurrently with the GEMM's stream
restricted to a subset of the CUs (hipExtStreamCreateWithCUMask)."""
import ctypes, json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import plnlp_amd as P
from plnlp_amd import synthetic

dev = torch.device("cuda", 0)
hip = ctypes.CDLL("libamdhip64.so")


def masked_stream(words):
    arr = (ctypes.c_uint32 * len(words))(*words)
    s = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(s), ctypes.c_uint32(len(words)), arr)
    if rc != 0:
        raise RuntimeError(f"hipExtStreamCreateWithCUMask -> {rc}")
    return torch.cuda.ExternalStream(s.value, device=dev)


def pattern(keep_of_8):
    """keep_of_8 of every 8 consecutive CU bits"""
    byte = (1 << keep_of_8) - 1
    return [byte * 0x01010101] * 8


g = synthetic.make_graph("collab", seed=2, device=dev, weighted=True)
adj_t = g["adj_t"].t()
n = g["num_nodes"]
T = 132224
torch.manual_seed(0)
dz = torch.randn(T, 256, device=dev) * 0.05
cat = torch.randn(T, 512, device=dev)
gw = torch.empty(256, 512, device=dev)
x = torch.randn(n, 256, device=dev)
out = torch.empty(n, 256, device=dev)


def gemm():
    P.ops.gemm([(dz, cat)], True, False, out=gw)


def agg():
    P.ops.csr_aggregate(adj_t, x, "sum", True, out=out)


def timed(fn, iters=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


def both_on(side):
    main = torch.cuda.current_stream()

    def run():
        ev = torch.cuda.Event()
        ev.record(main)
        side.wait_event(ev)
        with torch.cuda.stream(side):
            gemm()
            done = torch.cuda.Event()
            done.record(side)
        agg()
        main.wait_event(done)
    return run


def on(side, fn):
    def run():
        with torch.cuda.stream(side):
            fn()
        torch.cuda.current_stream().wait_stream(side)
    return run


res = {"gemm_us": timed(gemm), "agg_us": timed(agg), "serial_us": timed(lambda: (gemm(), agg()))}
plain = torch.cuda.Stream(device=dev, priority=-1)
res["plain_side_stream_us"] = timed(both_on(plain))
print(json.dumps(res), flush=True)
for keep in (2, 3, 4, 5, 6):
    try:
        s = masked_stream(pattern(keep))
    except Exception as ex:          # noqa: BLE001 -- a probe: report and stop
        print(json.dumps({"keep_of_8": keep, "error": str(ex)}))
        break
    r = {"keep_of_8": keep, "gemm_alone_masked_us": timed(on(s, gemm)), "agg_alone_masked_us": timed(on(s, agg)),
         "concurrent_us": timed(both_on(s))}
    for slots in (384, 256):
        old = P.ops.SPLIT_K_SLOTS["slots"]
        P.ops.SPLIT_K_SLOTS["slots"] = slots * keep // 8 if slots == 384 else slots
        r[f"concurrent_slots{P.ops.SPLIT_K_SLOTS['slots']}_us"] = timed(both_on(s))
        P.ops.SPLIT_K_SLOTS["slots"] = old
    print(json.dumps(r), flush=True)
