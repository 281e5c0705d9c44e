"""One launch over G x 4096 rows (the loader's group as ONE grid): us per 4096-row batch vs G and streams.

    python tools/exp_grid.py [scale=hg38] [seconds-per-leg=0.4]

Every leg rotates over >= 690 MB of windows + slot lines (256 batches' worth), like bench.py's cold default.
"""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from genvarloader_amd import HapsDevice, synth


def main():
    scale = sys.argv[1] if len(sys.argv) > 1 else "hg38"
    leg_s = float(sys.argv[2]) if len(sys.argv) > 2 else 0.4
    gs = [int(x) for x in os.environ.get("GS", "1,2,4,8,16").split(",")]
    sts = [int(x) for x in os.environ.get("STREAMS", "1,2,3").split(",")]
    ds = synth.make_genome(scale, os.environ.get("WORKLOAD", "cfg3"), device="cuda:0", seed=20260805)
    dev = HapsDevice(**ds.static_kwargs(), device="cuda:0")
    P, L = ds.ploidy, ds.length
    K = 4096
    fn = dev.lib.gvl_reconstruct
    dref = C.byref(dev.c)
    all_streams = [torch.cuda.current_stream()] + [torch.cuda.Stream() for _ in range(max(sts) - 1)]
    for G in gs:
        n_rot = max(1, int(os.environ.get("ROT", "256")) // G) if scale == "hg38" else 1
        qsets = ds.draw_batches(n_rot, G * K // P, seed=11 + G)
        batches = []
        for q in qsets:
            r = ds.request(q, rc=os.environ.get("WORKLOAD", "cfg3") == "cfg3")
            batches.append(dev.prepare_batch(r["regions"], r["shifts"], r["geno_offset_idx"], L, to_rc=r["to_rc"]))
        n_slots = max(sts) + 1
        slots = [dev.alloc_output(batches[0], G * K * L, haps=False, onehot=True) for _ in range(n_slots)]
        bref = [C.byref(b.c) for b in batches]
        sref = [C.byref(s[1]) for s in slots]
        for ns in sts:
            streams = all_streams[:ns]
            sp = [C.c_void_p(s.cuda_stream) for s in streams]
            est = 8e-6 * G
            n = max(3 * ns, int(leg_s / est))
            n -= n % ns
            res = []
            for rep in range(3):
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                start = torch.cuda.Event()
                start.record(streams[0])
                for s in streams[1:]:
                    s.wait_event(start)
                e0.record(streams[0])
                for i in range(n):
                    if fn(dref, bref[i % n_rot], sref[i % n_slots], sp[i % ns]):
                        raise RuntimeError("launch failed")
                for s in streams[1:]:
                    ev = torch.cuda.Event()
                    ev.record(s)
                    streams[0].wait_event(ev)
                e1.record(streams[0])
                torch.cuda.synchronize()
                res.append(e0.elapsed_time(e1) * 1e3 / (n * G))
            print(f"G={G:3d} rows/launch={G * K:6d} streams={ns}  us per 4096 rows: " + " ".join(f"{x:6.2f}" for x in res), flush=True)
        del slots, batches
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
