"""Timeline of bench.py's `host_boundary` double-buffered form: when each copy and each forward starts and ends (HIP events on the three
streams, relative to the first forward's start).  Shows whether the copies on their own streams really run under the forwards.

    python tools/host_boundary_timeline.py [batches]          (COPY_IN_AHEAD=0: the next batch's copy in submitted AFTER this batch's
                                                                copy out -- it then waits behind it in the copy queue)
"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402
from audiodenoiser_amd.weights import make_state_dict  # noqa: E402


def main():
    n_it = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    dev = torch.device("cuda", 0)
    net = bench.make_net(make_state_dict(1234), dev, "f32")
    shape = (64, 1, bench.F_BINS, bench.T_FRAMES)
    x_pin = (torch.rand(shape) * 4.0).pin_memory()
    y_pin = [torch.empty(shape).pin_memory() for _ in range(2)]
    xd = [torch.empty(shape, device=dev) for _ in range(2)]
    comp = torch.cuda.current_stream(dev)
    s_in, s_out = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
    ev_in = [torch.cuda.Event() for _ in range(2)]
    ev_comp = [torch.cuda.Event() for _ in range(2)]

    def ev():
        return torch.cuda.Event(enable_timing=True)
    ahead = os.environ.get("COPY_IN_AHEAD", "1") != "0"
    print("copy in submitted", "before the previous batch's copy out" if ahead else "after the previous batch's copy out")
    with torch.no_grad():
        for _ in range(2):
            net(xd[0])
        torch.cuda.synchronize(dev)
        marks = [{n: ev() for n in ("in0", "in1", "f0", "f1", "out0", "out1")} for _ in range(n_it)]

        def copy_in(i):
            k, m = i & 1, marks[i]
            with torch.cuda.stream(s_in):
                if i >= 2:
                    s_in.wait_event(ev_comp[k])            # batch i-2's forward has read xd[k]
                m["in0"].record(s_in)
                xd[k].copy_(x_pin, non_blocking=True)
                m["in1"].record(s_in)
                ev_in[k].record(s_in)
        copy_in(0)
        for i in range(n_it):
            k, m = i & 1, marks[i]
            if ahead and i + 1 < n_it:
                copy_in(i + 1)                             # submitted BEFORE this batch's copy out: the copy queue is served in order
            comp.wait_event(ev_in[k])
            m["f0"].record(comp)
            y = net(xd[k])
            m["f1"].record(comp)
            ev_comp[k].record(comp)
            with torch.cuda.stream(s_out):
                s_out.wait_event(ev_comp[k])
                m["out0"].record(s_out)
                y_pin[k].copy_(y, non_blocking=True)
                m["out1"].record(s_out)
                y.record_stream(s_out)
            if not ahead and i + 1 < n_it:
                copy_in(i + 1)
        torch.cuda.synchronize(dev)
    t0 = marks[0]["f0"]
    print("batch   copy in [start, end]    forward [start, end]    copy out [start, end]   (ms after the first forward's start)")
    for i, m in enumerate(marks):
        t = {n: t0.elapsed_time(e) for n, e in m.items()}
        print(f"{i:5d}   {t['in0']:9.2f} {t['in1']:9.2f}    {t['f0']:9.2f} {t['f1']:9.2f}    {t['out0']:9.2f} {t['out1']:9.2f}")


if __name__ == "__main__":
    main()
